"""scenedreamer_amd: MI355X-native implementation of SceneDreamer's inference hot path.

Ray-voxel intersection, positional encoding, the hash-grid encoder and the
fused field renderer are hand-written HIP kernels (csrc/) behind the C ABI in
include/sdnative.h; this package is the thin host side mirroring the
reference's Python op surface.
"""
import os
import sys

__all__ = ["install_shims", "SHIM_DIR"]

SHIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def install_shims():
    """Make `voxlib`, `_gridencoder`, `gridencoder`, `upfirdn2d_cuda`, `bias_act_cuda`
    importable as top-level modules so the unmodified reference generator
    (imaginaire.generators.scenedreamer) runs on this backend."""
    if SHIM_DIR not in sys.path:
        sys.path.insert(0, SHIM_DIR)
