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


def install_shims(fast=False):
    """Make `voxlib`, `_gridencoder`, `gridencoder`, `upfirdn2d_cuda`, `bias_act_cuda`
    importable as top-level modules so the unmodified reference generator
    (imaginaire.generators.scenedreamer) runs on this backend.

    fast=True additionally puts the fused fast path behind the generator's own surface (dropin.py): `imaginaire`'s
    LightningMLP / SKYMLP / RenderCNN come out of the import as subclasses whose forward runs on the MFMA kernels, and
    Generator._forward_perpix / _forward_global are bound to sdn_field_render / the MFMA render CNN -- the reference's files
    are imported unchanged, only names in the finished module objects are rebound."""
    if SHIM_DIR not in sys.path:
        sys.path.insert(0, SHIM_DIR)
    if fast:
        from . import dropin
        dropin.install_import_hook()
