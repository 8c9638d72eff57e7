"""numpy front-end of oracle/_ref: the reference's own native sources compiled for the host
(oracle/build_ref.py).  TEST INFRASTRUCTURE ONLY -- same call signatures as oracle/oracle.py so a
test can run both and demand identical bits.

One variant per process: the two builds export the same symbols and Python caches extension
modules by name, so `variant="fma"` must be used from a fresh interpreter (see `run_variant`).
"""
import importlib.machinery
import importlib.util
import os
import pickle
import subprocess
import sys

import numpy as np
import torch

from . import build_ref as BR

_loaded = {}
_variant = None


def available(variant="nofma"):
    return BR.built(variant)


def load(name, variant="nofma"):
    global _variant
    if _variant not in (None, variant):
        raise RuntimeError(f"oracle/_ref variant {_variant!r} already loaded in this process; use run_variant()")
    if name not in _loaded:
        path = BR.module_path(name, variant)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m oracle.build_ref` in the build container")
        loader = importlib.machinery.ExtensionFileLoader(name, path)
        spec = importlib.util.spec_from_file_location(name, path, loader=loader)
        m = importlib.util.module_from_spec(spec)
        loader.exec_module(m)
        _loaded[name] = m
        _variant = variant
    return _loaded[name]


def _t(a, dt=np.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dt)))


def rvip(vox, cam_ori, cam_dir, cam_up, cam_f, cam_c, img_dims, max_samples, variant="nofma"):
    """voxlib.ray_voxel_intersection_perspective of the reference build.  `vox`: int32 ndarray, any strides."""
    v = torch.as_strided(torch.from_numpy(np.lib.stride_tricks.as_strided(vox)), vox.shape,
                         [s // 4 for s in vox.strides]) if not vox.flags.c_contiguous else torch.from_numpy(vox)
    out = load("voxlib", variant).ray_voxel_intersection_perspective(
        v, _t(cam_ori), _t(cam_dir), _t(cam_up), float(cam_f), [float(c) for c in cam_c],
        [int(i) for i in img_dims], int(max_samples))
    return tuple(o.numpy() for o in out)


def posenc_fwd(x, ndegrees, dim=-1, incl_orig=False, variant="nofma"):
    return load("voxlib", variant).positional_encoding(_t(x), int(ndegrees), int(dim), bool(incl_orig)).numpy()


def posenc_bwd(out_grad, out, ndegrees, dim=-1, incl_orig=False, variant="nofma"):
    return load("voxlib", variant).positional_encoding_backward(_t(out_grad), _t(out), int(ndegrees), int(dim),
                                                                bool(incl_orig)).numpy()


def grid_encode_fwd(inputs, embeddings, offsets, S, H, calc_grad_inputs=False, gridtype=0, align_corners=False,
                    variant="nofma", dtype=np.float32):
    x = _t(inputs)
    emb = _t(embeddings, dtype)
    off = _t(offsets, np.int32)
    B, D = x.shape
    C = emb.shape[1]
    L = off.numel() - 1
    out = torch.empty(L, B, C, dtype=emb.dtype)
    dy_dx = torch.empty(B, L * D * C, dtype=emb.dtype) if calc_grad_inputs else torch.empty(1, dtype=emb.dtype)
    load("_gridencoder", variant).grid_encode_forward(x, emb, off, out, B, D, C, L, float(S), int(H),
                                                      bool(calc_grad_inputs), dy_dx, int(gridtype),
                                                      bool(align_corners))
    return (out.numpy(), dy_dx.numpy()) if calc_grad_inputs else out.numpy()


def grid_encode_bwd(grad, inputs, embeddings_shape, offsets, S, H, dy_dx=None, gridtype=0, align_corners=False,
                    variant="nofma", dtype=np.float32):
    g = _t(grad, dtype)
    x = _t(inputs)
    off = _t(offsets, np.int32)
    L, B, C = g.shape
    D = x.shape[1]
    emb = torch.zeros(tuple(embeddings_shape), dtype=g.dtype)  # only its dtype/ptr are consulted
    gg = torch.zeros(tuple(embeddings_shape), dtype=g.dtype)
    calc = dy_dx is not None
    dd = _t(dy_dx, dtype) if calc else torch.empty(1, dtype=g.dtype)
    gi = torch.zeros(B, D, dtype=g.dtype) if calc else torch.empty(1, dtype=g.dtype)
    load("_gridencoder", variant).grid_encode_backward(g, x, emb, off, gg, B, D, C, L, float(S), int(H), calc, dd,
                                                       gi, int(gridtype), bool(align_corners))
    return gg.numpy(), (gi.numpy() if calc else None)


def run_variant(variant, fn, *args):
    """Evaluate ref_native.<fn>(*args, variant=variant) in a fresh interpreter (numpy in / out)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import pickle,sys; sys.path.insert(0, %r); from oracle import ref_native as R; "
            "fn, args = pickle.load(sys.stdin.buffer); "
            "sys.stdout.buffer.write(pickle.dumps(getattr(R, fn)(*args, variant=%r)))" % (root, variant))
    r = subprocess.run([sys.executable, "-c", code], input=pickle.dumps((fn, args)), capture_output=True, check=False)
    if r.returncode != 0:
        raise RuntimeError(r.stderr.decode()[-2000:])
    return pickle.loads(r.stdout)
