"""Tensor-level wrappers over the C ABI.

These functions have exactly the signatures of the reference's pybind modules
(`voxlib`: imaginaire/model_utils/gancraft/voxlib/voxlib.cpp:25-31;
`_gridencoder`: gridencoder/src/bindings.cpp:5-8): PyTorch owns every tensor
(allocation through its caching allocator), work is enqueued on PyTorch's
current stream, argument violations raise RuntimeError like TORCH_CHECK does.
"""
import ctypes

import numpy as np
import torch

from . import capi

_f3 = ctypes.c_float * 3
_f2 = ctypes.c_float * 2
_i2 = ctypes.c_int * 2
_l3 = ctypes.c_int64 * 3


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _host3(t, name):
    if isinstance(t, torch.Tensor):
        _require(t.dtype == torch.float32, f"{name} must be float32")
        _require(t.numel() == 3, f"{name} must have 3 elements")
        v = t.detach().cpu().reshape(3).tolist()  # like .cpu() in ray_voxel_intersection.cu:275-277
    else:
        v = [float(x) for x in np.asarray(t, dtype=np.float32).reshape(3)]
    return _f3(*v)


def _stream(t):
    return capi.current_stream(t.device)


def voxel_occupancy(in_voxel):
    """Empty-space acceleration grid of `in_voxel` for sdn_rvip (1 byte per 8x16x16 cells, one pass over the volume).
    It is cached ON the tensor object (an address-keyed cache would go stale when the allocator reuses the address)
    and rebuilt after in-place edits (tensor._version)."""
    tag = (in_voxel._version, in_voxel.data_ptr(), tuple(in_voxel.shape), tuple(in_voxel.stride()))
    cached = getattr(in_voxel, "_sdn_occupancy", None)
    if cached is not None and cached[0] == tag:
        return cached[1]
    lib = capi.lib()
    dims, strides = _l3(*in_voxel.shape), _l3(*in_voxel.stride())
    build = lib.sdn_rvip_build_occupancy_u8 if in_voxel.dtype == torch.uint8 else lib.sdn_rvip_build_occupancy
    with torch.cuda.device(in_voxel.device):
        occ = torch.empty(lib.sdn_rvip_occupancy_bytes(dims), dtype=torch.uint8, device=in_voxel.device)
        capi.check(build(in_voxel.data_ptr(), dims, strides, occ.data_ptr(), _stream(in_voxel)), "sdn_rvip_build_occupancy")
    in_voxel._sdn_occupancy = (tag, occ)
    return occ


def ray_voxel_intersection_perspective(in_voxel, cam_ori, cam_dir, cam_up, cam_f, cam_c, img_dims, max_samples,
                                       accelerate=True, palette=None):
    """voxlib.ray_voxel_intersection_perspective (ray_voxel_intersection.cu:253-325).

    Returns [voxel_id i32[H,W,M,1], depth2 f32[2,H,W,M,1], raydirs f32[H,W,1,3]] on in_voxel's device.
    `accelerate` (not in the reference signature) only selects exact empty-space skipping; results are identical.
    `palette` (not in the reference signature): with a uint8 `in_voxel` of palette indices and an int32[256] palette
    on the same device the compact volume is walked (scene.py); voxel_id still holds the int32 block ids.
    """
    _require(isinstance(in_voxel, torch.Tensor) and in_voxel.is_cuda, "in_voxel must be a CUDA tensor")
    if palette is not None:
        _require(in_voxel.dtype == torch.uint8, "in_voxel must be uint8 when a palette is given")
        _require(isinstance(palette, torch.Tensor) and palette.is_cuda and palette.dtype == torch.int32 and
                 palette.numel() == 256 and palette.is_contiguous(), "palette must be a contiguous CUDA int32[256] tensor")
    else:
        _require(in_voxel.dtype == torch.int32, "in_voxel must be int32")
    _require(in_voxel.dim() == 3, "in_voxel must be 3-D")
    _require(len(img_dims) == 2, "img_dims must have 2 entries")
    H, W, M = int(img_dims[0]), int(img_dims[1]), int(max_samples)
    cam_f = float(np.asarray(cam_f, dtype=np.float64).reshape(-1)[0])  # training passes a 1-element array
    dev = in_voxel.device
    with torch.cuda.device(dev):
        voxel_id = torch.empty((H, W, M, 1), dtype=torch.int32, device=dev)
        depth2 = torch.empty((2, H, W, M, 1), dtype=torch.float32, device=dev)
        raydirs = torch.empty((H, W, 1, 3), dtype=torch.float32, device=dev)
        if H * W * M == 0:
            return [voxel_id, depth2, raydirs]
        occ = voxel_occupancy(in_voxel) if accelerate and in_voxel.numel() > 0 else None
        tail = (_host3(cam_ori, "cam_ori"), _host3(cam_dir, "cam_dir"), _host3(cam_up, "cam_up"),
                cam_f, _f2(float(cam_c[0]), float(cam_c[1])), _i2(H, W), M, occ.data_ptr() if occ is not None else None,
                voxel_id.data_ptr(), depth2.data_ptr(), raydirs.data_ptr(), _stream(in_voxel))
        if palette is not None:
            rc = capi.lib().sdn_rvip_u8(in_voxel.data_ptr(), palette.data_ptr(), _l3(*in_voxel.shape), _l3(*in_voxel.stride()), *tail)
        else:
            rc = capi.lib().sdn_rvip(in_voxel.data_ptr(), _l3(*in_voxel.shape), _l3(*in_voxel.stride()), *tail)
    capi.check(rc, "sdn_rvip")
    return [voxel_id, depth2, raydirs]


def rvip_step_counts(in_voxel, cam_ori, cam_dir, cam_up, cam_f, cam_c, img_dims, max_samples, accelerate=True, palette=None):
    """Counters of the ray marcher's walk for one frame (sdn_rvip_debug_counts; measurement only): dict with `iterations`
    (accelerate=False: the cell-by-cell DDA steps of the reference's loop), `volume_reads`, `block_jumps`, `rays`."""
    H, W, M = int(img_dims[0]), int(img_dims[1]), int(max_samples)
    dev = in_voxel.device
    cam_f = float(np.asarray(cam_f, dtype=np.float64).reshape(-1)[0])
    with torch.cuda.device(dev):
        vid = torch.empty((H, W, M, 1), dtype=torch.int32, device=dev)
        d2 = torch.empty((2, H, W, M, 1), dtype=torch.float32, device=dev)
        rd = torch.empty((H, W, 1, 3), dtype=torch.float32, device=dev)
        cnt = torch.zeros(4, dtype=torch.int64, device=dev)
        occ = voxel_occupancy(in_voxel) if accelerate else None
        rc = capi.lib().sdn_rvip_debug_counts(in_voxel.data_ptr(), palette.data_ptr() if palette is not None else None,
                                              _l3(*in_voxel.shape), _l3(*in_voxel.stride()), _host3(cam_ori, "cam_ori"),
                                              _host3(cam_dir, "cam_dir"), _host3(cam_up, "cam_up"), cam_f,
                                              _f2(float(cam_c[0]), float(cam_c[1])), _i2(H, W), M, occ.data_ptr() if occ is not None else None,
                                              vid.data_ptr(), d2.data_ptr(), rd.data_ptr(), cnt.data_ptr(), _stream(in_voxel))
    capi.check(rc, "sdn_rvip_debug_counts")
    it, rd_, jp, rays = (int(v) for v in cnt.tolist())
    return {"iterations": it, "volume_reads": rd_, "block_jumps": jp, "rays": rays}


def sample_depth_batched(depth2, nsamples, deterministic=False, use_box_boundaries=True, sample_depth=4, rand=None,
                         division="reciprocal", boundary_rand=None):
    """mc_utils.sample_depth_batched (imaginaire/model_utils/gancraft/mc_utils.py:82-151), same signature, defaults and
    return values: depth2 [N,2,H,W,M,1] -> (rand_depth [N,H,W,K,1], new_dists likewise, idx int64 likewise), K = nsamples - 1
    (use_box_boundaries=False: SceneDreamer's configs, scenedreamer.py:346-348) or nsamples + M (True).

    Not in the reference signature:
    rand: the uniform randoms of the stochastic branch, [N,H,W,nsamples,1]; by default drawn with torch.rand exactly as the
      reference does (:121), so a seeded call consumes the generator like the reference's call.
    division: how `rand_samples / nsamples` (:123, tensor / Python scalar) is evaluated.  "reciprocal" (default): as PyTorch
      evaluates it on a CUDA tensor -- the reference's GPU path -- a multiplication by the float32 reciprocal
      (BinaryDivTrueKernel.cu, CPU-scalar fast path); "ieee": as on a CPU tensor, a true division -- what the goldens recorded
      from the reference's CPU run contain.  At most 1 ulp apart, and only when nsamples is not a power of two.
    boundary_rand: the filler draw of the use_box_boundaries branch (:111, torch.rand_like(accu_depth)), [N,H,W,M,1].

    use_box_boundaries=False runs in one HIP kernel (sdn_sample_depth).  use_box_boundaries=True (GANcraft's option; no
    SceneDreamer config uses it, scenedreamer_train.yaml:121) needs a per-ray sort of nsamples + M + 1 positions and is
    composed from PyTorch ops on the device, in the reference's order of operations and of generator draws."""
    _require(isinstance(depth2, torch.Tensor) and depth2.is_cuda and depth2.dtype == torch.float32, "depth2 must be a CUDA float32 tensor")
    _require(depth2.dim() == 6 and depth2.shape[1] == 2 and depth2.shape[-1] == 1, "depth2 must be [N,2,H,W,M,1]")
    _require(division in ("reciprocal", "ieee"), "division must be 'reciprocal' or 'ieee'")
    N, _, H, W, M, _ = depth2.shape
    dev = depth2.device
    if use_box_boundaries:
        return _sample_depth_with_boundaries(depth2, nsamples, deterministic, sample_depth, rand, division, boundary_rand)
    R = N * H * W
    d2 = depth2.permute(1, 0, 2, 3, 4, 5).reshape(2, R, M).contiguous()
    if deterministic:
        lin = torch.linspace(0, 1, nsamples + 2)[1:-1].contiguous().to(dev)
        u = None
    else:
        lin = torch.linspace(0, 1, nsamples + 1)[:-1].contiguous().to(dev)
        if rand is None:
            rand = torch.rand([N, H, W, nsamples, 1], dtype=depth2.dtype, device=dev)
        u = rand.reshape(R, nsamples).contiguous()
    with torch.cuda.device(dev):
        depth = torch.empty((R, nsamples - 1), dtype=torch.float32, device=dev)
        dists = torch.empty_like(depth)
        idx = torch.empty((R, nsamples - 1), dtype=torch.int64, device=dev)
        capi.check(capi.lib().sdn_sample_depth(d2.data_ptr(), lin.data_ptr(), u.data_ptr() if u is not None else None, R, M,
                                               nsamples, float(sample_depth), depth.data_ptr(), dists.data_ptr(),
                                               idx.data_ptr(), 1 if division == "ieee" else 0, _stream(depth2)),
                   "sdn_sample_depth")
    shape = (N, H, W, nsamples - 1, 1)
    return depth.view(shape), dists.view(shape), idx.view(shape)


def _sample_depth_with_boundaries(depth2, nsamples, deterministic, sample_depth, rand, division, boundary_rand):
    """The use_box_boundaries=True form of mc_utils.sample_depth_batched (:108-114, :127-131): the in-range box exits join
    the samples (out-of-range ones are replaced by uniform fillers), plus a sample at depth 0, before the sort."""
    N, _, H, W, M, _ = depth2.shape
    dev = depth2.device
    t, t2 = depth2[:, 0], depth2[:, 1]
    d = t2 - t
    d = torch.where(torch.isnan(d), torch.zeros_like(d), d)      # only NaNs are zeroed (mc_utils.py:101), infinities stay
    accu = torch.cumsum(d, dim=-2)
    total = accu[..., -1:, :].clamp(max=sample_depth)
    if boundary_rand is None:
        boundary_rand = torch.rand_like(accu)                           # drawn before the stratified randoms, :111 / :121
    bad = (accu > sample_depth) | (d == 0)
    bnd = torch.where(bad, boundary_rand * total, accu)
    if deterministic:
        s = torch.linspace(0, 1, nsamples + 2)[1:-1].to(dev).view(1, 1, 1, nsamples, 1).expand(N, H, W, nsamples, 1)
    else:
        if rand is None:
            rand = torch.rand([N, H, W, nsamples, 1], dtype=depth2.dtype, device=dev)
        s = (rand * (1.0 / nsamples) if division == "reciprocal" else rand / torch.tensor(float(nsamples), device=dev)) \
            + torch.linspace(0, 1, nsamples + 1, device=dev)[:-1].view(1, 1, 1, nsamples, 1)
    s = torch.cat([s * total, bnd, torch.zeros([N, H, W, 1, 1], dtype=depth2.dtype, device=dev)], dim=-2)
    s, _ = torch.sort(s, dim=-2)
    mid = (s[..., 1:, :] + s[..., :-1, :]) / 2
    new_dists = s[..., 1:, :] - s[..., :-1, :]
    idx = (mid.unsqueeze(-3) > accu.unsqueeze(-2)).sum(dim=-3)
    gaps = torch.cumsum(t[..., 1:, :] - t2[..., :-1, :], dim=-2)
    heads = torch.cat([t[..., :1, :], gaps + t[..., :1, :]], dim=-2)
    return torch.gather(heads, -2, idx) + mid, new_dists, idx


def _pe_sizes(t, dim):
    if dim < 0:
        dim = t.dim() + dim
    _require(0 <= dim < t.dim(), "dim out of range")
    pre = 1
    for i in range(dim):
        pre *= t.size(i)
    post = 1
    for i in range(dim, t.dim()):
        post *= t.size(i)
    return dim, pre, post


def positional_encoding(in_feature, ndegrees, dim, incl_orig):
    """voxlib.positional_encoding (positional_encoding_kernel.cu:129-197)."""
    _require(isinstance(in_feature, torch.Tensor) and in_feature.is_cuda, "in_feature must be a CUDA tensor")
    _require(in_feature.dtype == torch.float32, "in_feature must be float32")
    _require(in_feature.is_contiguous(), "in_feature must be contiguous")
    dim, pre, post = _pe_sizes(in_feature, dim)
    mult = ndegrees * 2 + (1 if incl_orig else 0)
    shape = list(in_feature.shape)
    shape[dim] *= mult
    with torch.cuda.device(in_feature.device):
        out = torch.empty(shape, dtype=torch.float32, device=in_feature.device)
        rc = capi.lib().sdn_posenc_fwd(in_feature.data_ptr(), out.data_ptr(), pre, post, int(ndegrees),
                                       int(bool(incl_orig)), _stream(in_feature))
    capi.check(rc, "sdn_posenc_fwd")
    # provenance for modules.SKYMLPNative: a consumer that is handed THIS tensor object, unmodified, may evaluate the encoding
    # of `in_feature` itself (the sky kernel does) instead of reading it back
    out._sdn_pe_src = (in_feature, in_feature._version, int(ndegrees), dim, bool(incl_orig), out._version)
    return out


def positional_encoding_backward(out_feature_grad, out_feature, ndegrees, dim, incl_orig):
    """voxlib.positional_encoding_backward (positional_encoding_kernel.cu:209-285)."""
    _require(out_feature_grad.is_cuda and out_feature.is_cuda, "out_feature_grad must be a CUDA tensor")
    _require(out_feature_grad.dtype == torch.float32 and out_feature.dtype == torch.float32, "float32 expected")
    _require(out_feature.is_contiguous(), "out_feature must be contiguous")
    out_feature_grad = out_feature_grad.contiguous()
    mult = ndegrees * 2 + (1 if incl_orig else 0)
    if dim < 0:
        dim = out_feature.dim() + dim
    shape = list(out_feature.shape)
    _require(shape[dim] % mult == 0, "encoded dim is not a multiple of 2*ndegrees(+1)")
    shape[dim] //= mult
    pre = 1
    for i in range(dim):
        pre *= shape[i]
    post = 1
    for i in range(dim, len(shape)):
        post *= shape[i]
    with torch.cuda.device(out_feature.device):
        in_grad = torch.empty(shape, dtype=torch.float32, device=out_feature.device)
        rc = capi.lib().sdn_posenc_bwd(out_feature_grad.data_ptr(), out_feature.data_ptr(), in_grad.data_ptr(), pre,
                                       post, int(ndegrees), int(bool(incl_orig)), _stream(out_feature))
    capi.check(rc, "sdn_posenc_bwd")
    return in_grad


def sp_trilinear_worldcoord(*args, **kwargs):
    raise NotImplementedError("sp_trilinear_worldcoord is GANcraft-only and off SceneDreamer's path "
                              "(scenedreamer.py:285 overrides its only caller)")


def sp_trilinear_worldcoord_backward(*args, **kwargs):
    raise NotImplementedError("sp_trilinear_worldcoord_backward is GANcraft-only and off SceneDreamer's path")


def _dtype_code(t, name):
    if t.dtype == torch.float32:
        return capi.SDN_F32
    if t.dtype == torch.float16:
        return capi.SDN_F16
    raise RuntimeError(f"{name} must be a float32 or float16 tensor")


def _check_dev(t, name, contiguous=True):
    _require(isinstance(t, torch.Tensor) and t.is_cuda, f"{name} must be a CUDA tensor")
    if contiguous:
        _require(t.is_contiguous(), f"{name} must be a contiguous tensor")


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx, gridtype,
                        align_corners):
    """_gridencoder.grid_encode_forward (gridencoder.cu:423-446): writes `outputs` [L,B,C] in place."""
    for t, n in ((inputs, "inputs"), (embeddings, "embeddings"), (offsets, "offsets"), (outputs, "outputs"),
                 (dy_dx, "dy_dx")):
        _check_dev(t, n)
    _require(inputs.dtype == torch.float32, "inputs must be a float32 tensor")
    _require(offsets.dtype == torch.int32, "offsets must be an int tensor")
    code = _dtype_code(embeddings, "embeddings")
    _require(outputs.dtype == embeddings.dtype and dy_dx.dtype == embeddings.dtype,
             "outputs/dy_dx must have the dtype of embeddings")
    with torch.cuda.device(inputs.device):
        rc = capi.lib().sdn_grid_encode_fwd(inputs.data_ptr(), embeddings.data_ptr(), code, offsets.data_ptr(),
                                            outputs.data_ptr(), int(B), int(D), int(C), int(L), float(S), int(H),
                                            int(bool(calc_grad_inputs)), dy_dx.data_ptr(), int(gridtype),
                                            int(bool(align_corners)), _stream(inputs))
    capi.check(rc, "sdn_grid_encode_fwd")


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, dy_dx,
                         grad_inputs, gridtype, align_corners):
    """_gridencoder.grid_encode_backward (gridencoder.cu:448-478): accumulates into `grad_embeddings`."""
    for t, n in ((grad, "grad"), (inputs, "inputs"), (embeddings, "embeddings"), (offsets, "offsets"),
                 (grad_embeddings, "grad_embeddings"), (dy_dx, "dy_dx"), (grad_inputs, "grad_inputs")):
        _check_dev(t, n)
    code = _dtype_code(grad, "grad")
    with torch.cuda.device(inputs.device):
        rc = capi.lib().sdn_grid_encode_bwd(grad.data_ptr(), inputs.data_ptr(), embeddings.data_ptr(), code,
                                            offsets.data_ptr(), grad_embeddings.data_ptr(), int(B), int(D), int(C),
                                            int(L), float(S), int(H), int(bool(calc_grad_inputs)), dy_dx.data_ptr(),
                                            grad_inputs.data_ptr(), int(gridtype), int(bool(align_corners)),
                                            _stream(inputs))
    capi.check(rc, "sdn_grid_encode_bwd")
