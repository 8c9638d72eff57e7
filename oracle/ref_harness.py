"""Import the UNMODIFIED reference generator (Python side) -- from /root/reference in the build
container, from the staged archive oracle/_ref/pytree.zip (oracle/build_ref.stage_pytree: the same
files, unchanged, shipped like the prebuilt oracle/_ref/*.so) on the GPU box.

TEST INFRASTRUCTURE: used by oracle/make_golden.py and by the `needs_reference`
tests; nothing in the product package, smoke() or bench.py imports this module.

Six top-level modules the reference imports are absent here:
cv2, imageio, upfirdn2d_cuda, bias_act_cuda (never called on the path: inert
stubs) and the two native extensions voxlib / _gridencoder, which are shimmed
with the C oracle so that the reference's Python layers run on CPU tensors.
"""
import os
import sys
import types

import numpy as np
import torch

REFERENCE = "/root/reference"
_PYTREE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "pytree.zip")
_unpacked = None


def available():
    return os.path.isdir(os.path.join(REFERENCE, "imaginaire")) or os.path.exists(_PYTREE)


def root():
    """Directory that holds the reference's Python tree: /root/reference where it exists, else the staged archive
    unpacked into a temporary directory (once per process, removed at exit; never inside the repo)."""
    global _unpacked
    if os.path.isdir(os.path.join(REFERENCE, "imaginaire")):
        return REFERENCE
    if _unpacked is None:
        import atexit
        import shutil
        import tempfile
        import zipfile
        if not os.path.exists(_PYTREE):
            raise RuntimeError("neither /root/reference nor oracle/_ref/pytree.zip is present")
        d = tempfile.mkdtemp(prefix="sdn_ref_pytree_")
        with zipfile.ZipFile(_PYTREE) as z:
            z.extractall(d)
        atexit.register(shutil.rmtree, d, True)
        _unpacked = d
    return _unpacked


_installed = None


def _stub(name):
    m = types.ModuleType(name)
    m.__dict__["__getattr__"] = lambda attr: (_ for _ in ()).throw(NotImplementedError(f"{name}.{attr} (stub)"))
    return m


def install(native="oracle"):
    """Put the reference on sys.path with the stub / shim modules in place.

    native="oracle": voxlib/_gridencoder run the C oracle on CPU tensors.
    native="ref":    they are the reference's OWN sources compiled for the host (oracle/_ref,
                     oracle/build_ref.py) -- the whole reference, Python and native, on the CPU.
    native="hip":    they are scenedreamer_amd's HIP-backed shims (needs a GPU).
    native="hip-fast": the same with scenedreamer_amd.install_shims(fast=True): the generator's render networks and its
                     _forward_perpix / _forward_global run on the fused kernels (scenedreamer_amd/dropin.py).
    """
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    from . import oracle as O

    # modules bound to another backend by an earlier install() (the reference binds `voxlib` / `_gridencoder`
    # at import time: gancraft/voxlib/__init__.py:7, gridencoder/grid.py:9-12) must be re-imported
    global _installed
    if _installed != native:
        import scenedreamer_amd
        if scenedreamer_amd.SHIM_DIR in sys.path:
            sys.path.remove(scenedreamer_amd.SHIM_DIR)
        for name in list(sys.modules):
            if name.split(".")[0] in ("imaginaire", "gridencoder", "voxlib", "_gridencoder", "upfirdn2d_cuda",
                                      "bias_act_cuda"):
                del sys.modules[name]
    _installed = native

    for name in ("cv2", "imageio", "upfirdn2d_cuda", "bias_act_cuda"):
        if name not in sys.modules:
            sys.modules[name] = _stub(name)

    if native in ("hip", "hip-fast"):
        import scenedreamer_amd
        from scenedreamer_amd import dropin
        dropin.uninstall_import_hook()
        scenedreamer_amd.install_shims(fast=native == "hip-fast")
    elif native == "ref":
        from . import ref_native
        sys.modules["voxlib"] = ref_native.load("voxlib")
        sys.modules["_gridencoder"] = ref_native.load("_gridencoder")
    else:
        vox = types.ModuleType("voxlib")

        def ray_voxel_intersection_perspective(in_voxel, cam_ori, cam_dir, cam_up, cam_f, cam_c, img_dims, max_samples):
            a, b, c = O.rvip(in_voxel.cpu().numpy(), cam_ori.cpu().numpy(), cam_dir.cpu().numpy(),
                             cam_up.cpu().numpy(), float(np.asarray(cam_f).reshape(-1)[0]), list(cam_c),
                             list(img_dims), max_samples)
            return [torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(c)]

        def positional_encoding(in_feature, ndegrees, dim, incl_orig):
            return torch.from_numpy(O.posenc_fwd(in_feature.detach().cpu().numpy(), ndegrees, dim, incl_orig))

        def positional_encoding_backward(out_grad, out, ndegrees, dim, incl_orig):
            return torch.from_numpy(O.posenc_bwd(out_grad.cpu().numpy(), out.cpu().numpy(), ndegrees, dim, incl_orig))

        def _na(*a, **k):
            raise NotImplementedError("sp_trilinear_worldcoord: off path")

        vox.ray_voxel_intersection_perspective = ray_voxel_intersection_perspective
        vox.positional_encoding = positional_encoding
        vox.positional_encoding_backward = positional_encoding_backward
        vox.sp_trilinear_worldcoord = _na
        vox.sp_trilinear_worldcoord_backward = _na
        sys.modules["voxlib"] = vox

        ge = types.ModuleType("_gridencoder")

        def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx,
                                gridtype, align_corners):
            r = O.grid_encode_fwd(inputs.detach().numpy(), embeddings.detach().numpy(), offsets.numpy(), np.float32(S),
                                  H, bool(calc_grad_inputs), gridtype, bool(align_corners))
            if calc_grad_inputs:
                outputs.copy_(torch.from_numpy(r[0]))
                dy_dx.copy_(torch.from_numpy(r[1]))
            else:
                outputs.copy_(torch.from_numpy(r))

        def grid_encode_backward(*a, **k):
            raise NotImplementedError("oracle shim: backward not wired")

        ge.grid_encode_forward = grid_encode_forward
        ge.grid_encode_backward = grid_encode_backward
        sys.modules["_gridencoder"] = ge

    if root() not in sys.path:
        sys.path.insert(0, root())


def build_generator(weights=None, scene=None):
    """Construct imaginaire.generators.scenedreamer.Generator from the shipped inference config and
    (optionally) load our synthetic weights and attach a synthetic scene handle."""
    import warnings
    warnings.filterwarnings("ignore")
    from imaginaire.config import Config
    cfg = Config(os.path.join(root(), "configs/scenedreamer_inference.yaml"))
    from imaginaire.generators.scenedreamer import Generator
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        G = Generator(cfg.gen, cfg.data)
    G.eval()
    if weights is not None:
        sd = G.state_dict()
        missing = []
        with torch.no_grad():
            for k, v in weights.items():
                if k in sd:
                    assert tuple(sd[k].shape) == tuple(np.asarray(v).shape), (k, sd[k].shape, np.asarray(v).shape)
                    sd[k].copy_(torch.from_numpy(np.asarray(v)))
                else:
                    missing.append(k)
        assert not missing, f"names not in the reference state dict: {missing}"
        # every hot-path parameter of the reference must have been provided
        hot = [k for k in sd if k.split(".")[0] in ("hash_encoder", "render_net", "sky_net", "style_net",
                                                    "world_encoder", "denoiser")]
        absent = [k for k in hot if k not in weights]
        assert not absent, f"reference parameters not covered by synth.make_weights: {absent}"
    if scene is not None:
        # Generator.voxel is a registered nn.Module child: set attributes on it
        v = G.voxel
        v.voxel_t = scene.voxel_t
        v.heightmap = scene.heightmap
        v.current_height_map = scene.current_height_map
        v.current_semantic_map = scene.current_semantic_map
        v.trans_mat = scene.trans_mat
        v.sample_size = scene.sample_size
    return G, cfg


def set_inference_overrides(G, num_samples, resolution_hw, pad=30):
    """The attribute overrides of inference_givenstyle, scenedreamer.py:547-555."""
    G.pad = pad
    G.num_samples = num_samples
    G.num_blocks_early_stop = 6
    G.sample_depth = 3
    G.coarse_deterministic_sampling = True
    G.crop_size = resolution_hw
    G.cam_res = [resolution_hw[0] + pad, resolution_hw[1] + pad]
    G.use_label_smooth_pgt = False
