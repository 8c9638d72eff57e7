"""Host side of the fused field renderer (csrc/field.hip): per-scene / per-style preparation and the
per-frame encode -> mlp launches.  PyTorch only provides device memory and the stream."""
import ctypes
import math
import os

import numpy as np
import torch

from . import capi

_f2 = ctypes.c_float * 2
_f3 = ctypes.c_float * 3
_i6 = ctypes.c_int32 * 6


AUX_OUTPUTS = ("weights", "depth", "sigma", "colour", "sky_blended", "nosky")


class Window:
    """Which rays of the frame-wide ray arrays (voxel_id [n_src,M], depth2 [2,n_src,M], raydirs [n_src,3], sky_c [n_src,64])
    a field launch evaluates: ray (y, x) of a rows x cols window is source ray first + y * pitch + x.  The kernels read
    through the window (include/sdnative.h, `window_host`), so no strided-slice copies are made on the host side."""

    def __init__(self, n_src, pitch=0, first=0, rows=None, cols=0):
        self.n_src, self.pitch, self.first, self.cols = int(n_src), int(pitch), int(first), int(cols)
        self.rows = rows
        self.n_rays = int(rows * cols) if cols else int(n_src)

    @classmethod
    def crop(cls, H0, W0, o):
        """The H0 x W0 frame without its outer o rows / columns."""
        return cls(H0 * W0, W0, o * W0 + o, H0 - 2 * o, W0 - 2 * o)

    def blocked(self, ray0=0, n_rays=None, ragged=False):
        """Whether a launch over this window orders its rays in 8 x 4 pixel blocks (csrc/field.hip RayWindow): the launch is the
        whole window, of 8k columns x 4m rows -- or, with ragged=True (sdn_field_render only), of any size: the block grid then
        covers the window and the positions outside it are no rays.  SDN_RAY_BLOCKS=0 keeps the row-major order (A/B)."""
        if not self.cols or ray0 or (n_rays is not None and n_rays != self.n_rays) or os.environ.get("SDN_RAY_BLOCKS", "1") in ("0", ""):
            return False
        rows = self.n_rays // self.cols
        if rows * self.cols != self.n_rays:
            return False
        return ragged or (self.cols % 8 == 0 and rows % 4 == 0)

    def n_groups(self, ragged=False):
        """32-ray groups of a whole-window launch (the length of its `passes` / `colour_passes` arrays)."""
        if self.blocked(0, self.n_rays, ragged):
            rows = self.n_rays // self.cols
            return -(-self.cols // 8) * -(-rows // 4)
        return (self.n_rays + 31) // 32

    def groups(self, per_ray, ragged=False):
        """per_ray [n_rays, ...] (window row-major) -> [n_groups, 32, ...]: the rays of every 32-ray group of a whole-window launch
        in the launch's ray order (padding rays / block positions outside a ragged window: zeros)."""
        n = self.n_rays
        if self.blocked(0, n, ragged):
            rows = n // self.cols
            R4, C8 = -(-rows // 4) * 4, -(-self.cols // 8) * 8
            v = per_ray.reshape(rows, self.cols, *per_ray.shape[1:])
            if (R4, C8) != (rows, self.cols):
                full = per_ray.new_zeros((R4, C8) + tuple(per_ray.shape[1:]))
                full[:rows, :self.cols] = v
                v = full
            v = v.reshape(R4 // 4, 4, C8 // 8, 8, *per_ray.shape[1:])
            return v.transpose(1, 2).reshape(-1, 32, *per_ray.shape[1:])
        pad = (-n) % 32
        if pad:
            per_ray = torch.cat([per_ray, per_ray.new_zeros((pad,) + tuple(per_ray.shape[1:]))], dim=0)
        return per_ray.reshape(-1, 32, *per_ray.shape[1:])

    def host(self, ray0=0, n_rays=None, ragged=False):
        mode = 0
        if self.blocked(ray0, n_rays, ragged):
            rows = self.n_rays // self.cols
            mode = 1 if (self.cols % 8 == 0 and rows % 4 == 0) else 2
        return _i6(self.n_src, self.pitch, self.first, self.cols, int(ray0), mode)


def _lib():
    return capi.lib()


def _stream(dev):
    return capi.current_stream(dev)


def prepare_scene(R):
    """Collapse the 5-D hash table with this scene's global_enc (once per scene)."""
    lib = _lib()
    offs = R.w["hash_encoder.offsets"].cpu().numpy().astype(np.int32)
    L = offs.size - 1
    T = int(offs[1] - offs[0])
    genc = R.global_enc.detach().cpu().numpy().astype(np.float32).reshape(2)
    table3 = torch.empty((L, T, 8), dtype=torch.float32, device=R.dev)
    with torch.cuda.device(R.dev):
        rc = lib.sdn_field_collapse_table(R.w["hash_encoder.embeddings"].data_ptr(), offs.ctypes.data, L,
                                          float(np.float32(R.grid_S)), 16, genc.ctypes.data, table3.data_ptr(),
                                          _stream(R.dev))
    capi.check(rc, "sdn_field_collapse_table")
    scales = np.empty(L, np.float32)
    capi.check(lib.sdn_grid_level_scales(L, float(np.float32(R.grid_S)), 16, scales.ctypes.data, None))
    # the kernel indexes a 1024-entry table with id & 1023; ids the reference's LUT does not cover would raise
    # there (mc_utils.py:241-246) -- refuse them here instead of mapping them silently
    vmax = R.max_block_id
    if vmax >= min(R.lut.numel(), 1024):
        raise RuntimeError(f"scene holds voxel id {vmax}, outside the label table ({R.lut.numel()} entries)")
    lut = torch.full((1024,), 3, dtype=torch.uint8)
    n = min(R.lut.numel(), 1024)
    lut[:n] = R.lut[:n].to(torch.uint8).cpu()
    R._fused_scene = dict(table3=table3, T=T, genc=genc, scales=torch.from_numpy(scales).to(R.dev),
                          lut=lut.to(R.dev), dims=np.asarray([float(v) for v in R.voxel_dims], np.float32))
    return R._fused_scene


TRUNK_F16_HEADROOM = 32768.0   # largest |packed trunk weight| accepted (f16 overflows at 65504)


class TrunkRangeError(RuntimeError):
    """prepare_style: this style's trunk weights do not fit the packed f16 stream.  The explicit Renderer API lets it
    propagate; the drop-in surfaces (modules.py, dropin.py) catch it and evaluate the call with the reference's own method."""


def check_trunk_range(w1, trunk_hidden, shift):
    """The packed trunk weights (fc_1 .. fc_4) carry 2^shift (field.hip pack_kernel): refuse a style whose weights would
    leave f16's range there instead of rendering infinities.  One device->host read per style."""
    tops = [float(w1.abs().max())] + [float(t.abs().max()) * 0.4 for t in trunk_hidden]
    m = max(tops) if all(math.isfinite(v) for v in tops) else float("nan")
    if not math.isfinite(m) or m * 2.0 ** shift >= TRUNK_F16_HEADROOM:
        raise TrunkRangeError(f"field MLP trunk weights reach {m:.4g}: times 2^{shift} (the packed stream's scale) that leaves "
                              f"f16's range; this style cannot be rendered by the MFMA field kernel")
    return m


def prepare_style(R):
    """Pack the folded MLP weights into MFMA fragment order + build the fp32 constant block (once per style).  Two images
    of the stream: the 3-term f16 split everywhere, and the one with the colour layers as f16 + fp6 (colour_terms = 6)."""
    lib = _lib()
    w = R.w
    packed = torch.empty(lib.sdn_field_packed_weight_bytes(), dtype=torch.uint8, device=R.dev)
    packed_mx = torch.empty_like(packed)
    wh = [R.mod[i][0].contiguous() for i in (2, 3, 4, 5, 6)]
    ptrs = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in wh])
    w1 = w["render_net.fc_1.weight"].contiguous()
    wc = w["render_net.fc_out_c.weight"].contiguous()
    shift = lib.sdn_field_trunk_shift()
    check_trunk_range(w1, wh[:3], shift)
    with torch.cuda.device(R.dev):
        rc = lib.sdn_field_pack_weights(w1.data_ptr(), ptrs, wc.data_ptr(), packed.data_ptr(), _stream(R.dev))
        capi.check(rc, "sdn_field_pack_weights")
        rc = lib.sdn_field_pack_weights_mx(w1.data_ptr(), ptrs, wc.data_ptr(), packed_mx.data_ptr(), _stream(R.dev))
    capi.check(rc, "sdn_field_pack_weights_mx")
    consts = torch.zeros(lib.sdn_field_consts_floats(), dtype=torch.float32, device=R.dev)
    off = [lib.sdn_field_const_offset(i) for i in range(6)]
    consts[off[0]:off[0] + 12 * 256] = R.label_bias.reshape(-1)
    consts[off[1]:off[1] + 5 * 256] = torch.stack([R.mod[i][1] for i in (2, 3, 4, 5, 6)]).reshape(-1)
    # the MLP kernel's activations are LeakyReLU(x) / 0.4 (see field.hip act_stage): the density head absorbs the 0.4
    consts[off[2]:off[2] + 256] = w["render_net.fc_sigma.weight"].reshape(-1) * 0.4
    consts[off[3]:off[3] + 64] = w["render_net.fc_out_c.bias"]
    consts[off[4]] = w["render_net.fc_sigma.bias"].reshape(-1)[0]
    R._fused_style = dict(packed=packed, packed_mx=packed_mx, consts=consts, sky_off=off[5], keep=wh, trunk_shift=shift)
    return R._fused_style


def _buffers(R, n_rays, ns, slot=0):
    key = (n_rays, ns, slot)
    cache = R.__dict__.setdefault("_fused_buf", {})
    if key not in cache:
        lib = _lib()
        while len(cache) >= 4:      # two apron settings x two pipeline slots of one resolution; older shapes are dropped
            cache.pop(next(iter(cache)))
        aux = lib.sdn_field_aux_elems(n_rays, ns)
        cache[key] = dict(
            feat=torch.empty(lib.sdn_field_feat_bytes(n_rays, ns) // 4, dtype=torch.float32, device=R.dev),
            dist=torch.empty(aux, dtype=torch.float32, device=R.dev),
            label=torch.empty(aux, dtype=torch.uint8, device=R.dev),
            rayflag=torch.empty(n_rays, dtype=torch.uint8, device=R.dev),
            lin=torch.linspace(0, 1, ns + 3)[1:-1].contiguous().to(R.dev),  # mc_utils.py:120
        )
    return cache[key]


def encode(R, vid, d2, rd, cam_ori, ns, buf=None, u=None, window=None, ray0=0, n_rays=None, division="reciprocal"):
    """Sample placement + hash-grid lookup for n_rays rays of the ray arrays vid / d2 / rd: all of them (window None), or
    rays ray0 .. ray0 + n_rays - 1 of `window` (a Window over the frame-wide arrays).
    u: None = deterministic sampling (inference); f32 [n_rays, ns + 1] uniform randoms (the caller's
    torch.rand(..., ns + 1) draw, mc_utils.py:121) = the training-time stratified sampling; `division` as in
    ops.sample_depth_batched ("reciprocal": the reference on a CUDA tensor, "ieee": on a CPU tensor)."""
    sc = R._fused_scene or prepare_scene(R)
    if window is None:
        window = Window(vid.shape[0])
    assert vid.is_contiguous() and d2.is_contiguous() and rd.is_contiguous() and vid.shape[0] == window.n_src
    if n_rays is None:
        n_rays = window.n_rays - ray0
    buf = buf or _buffers(R, n_rays, ns)
    lin = buf["lin"]
    if u is not None:
        assert u.is_cuda and u.dtype == torch.float32 and tuple(u.shape) == (n_rays, ns + 1) and u.is_contiguous()
        lin = buf.get("lin_strat")
        if lin is None:
            lin = buf["lin_strat"] = torch.linspace(0, 1, ns + 2)[:-1].contiguous().to(R.dev)   # mc_utils.py:124
    ori = np.asarray(cam_ori.detach().cpu().numpy() if isinstance(cam_ori, torch.Tensor) else cam_ori, np.float32)
    with torch.cuda.device(R.dev):
        rc = _lib().sdn_field_encode(vid.data_ptr(), d2.data_ptr(), rd.data_ptr(), sc["lut"].data_ptr(),
                                     sc["table3"].data_ptr(), sc["T"], sc["scales"].data_ptr(),
                                     sc["genc"].ctypes.data, ori.ctypes.data, sc["dims"].ctypes.data,
                                     lin.data_ptr(), u.data_ptr() if u is not None else None, n_rays, R.M, ns,
                                     R.sample_depth, R.dists_scale,
                                     buf["feat"].data_ptr(), buf["dist"].data_ptr(), buf["label"].data_ptr(),
                                     buf["rayflag"].data_ptr(), window.host(ray0, n_rays), {"reciprocal": 0, "ieee": 1}[division],
                                     _stream(R.dev))
    capi.check(rc, "sdn_field_encode")
    return buf


FEATURE_BUFFER_BYTES = 32 << 30   # rays are processed in chunks whose encode -> mlp feature buffer stays below this


def precision_profile(R):
    """(colour_terms, term_eps) of the field MLP for this renderer.

    colour_terms: how the products of the colour layers fc_5 / fc_6 are evaluated (nothing amplifies their error,
    tools/precision_study.py): 6 (default) = Whi.Xhi in f16 + the two correction terms as block-scaled fp6 products (error
    equal to the full split's to 1e-6 on the goldens, half the MFMA issue slots); 3 = the 3-term f16 split like every other
    layer; 2 = without Whi.Xlo (4-7e-4 on net_out: opt-in only).
    term_eps: early ray termination once the transmittance of all 32 rays of a workgroup is below it (0 = off);
    bounds the change of net_out by 2 * term_eps.  Default TERM_EPS_DEFAULT (below); set_precision(term_eps=0) / SDN_TERM_EPS=0
    evaluate every sample like the reference."""
    ct = getattr(R, "colour_terms", None)
    if ct is None:
        if "SDN_MLP_COLOUR_TERMS" in os.environ:
            ct = int(os.environ["SDN_MLP_COLOUR_TERMS"])
        else:       # the per-style decision of Renderer.calibrate_field (6 unless the fp6 corrections cost more than its bound)
            ct = getattr(R, "colour_terms_auto", None) or 6
    eps = getattr(R, "term_eps", None)
    if eps is None:
        eps = float(os.environ.get("SDN_TERM_EPS", TERM_EPS_DEFAULT))
    return ct, eps


# Early ray termination is ON by default: a 32-ray group stops sampling once the transmittance of every one of its rays is below
# this (wavefront ballots, field.hip).  It moves net_out by at most 2 x eps = 1e-4 of the 1e-3 tolerance (typically far less: the
# bound assumes all of the remaining mass sits in the skipped samples); the renderer's per-style calibration measures the path
# WITH it, so the charge is inside the measured error.  On the synthetic benchmark weights it removes 2 - 6 % of the field
# kernel's passes, on an opaque-surface weight set 5 of 6 (tests/test_render_gpu.py, bench.py `early_termination`).
TERM_EPS_DEFAULT = "5e-5"


def _launch_mlp(R, buf, st, sky_c, sky_avg, net_out, n_rays, ns, passes=None, window=None, ray0=0, dynamic=True):
    """sky_c [n_src,64] is indexed through `window` like the ray arrays; sky_avg dev f32 [64] (the frame mean, straight from
    sky_kernel); net_out [n_rays,64] is local.  dynamic: the persistent workgroups draw their 32-ray groups from a
    ticket counter (one per renderer: launches of one renderer are serialized on its stream) instead of a static stride."""
    ct, eps = precision_profile(R)
    assert sky_c.is_contiguous() and sky_avg.is_contiguous() and sky_avg.numel() == 64 and sky_avg.dtype == torch.float32
    # both are dereferenced by the kernel: a host tensor here would be a wild device pointer (memory fault), not an error code
    if not (sky_c.is_cuda and sky_avg.is_cuda and sky_c.device == R.dev == sky_avg.device and net_out.device == R.dev):
        raise ValueError(f"sky_c / sky_avg / net_out must live on {R.dev} (got {sky_c.device}, {sky_avg.device}, {net_out.device})")
    if window is None:
        window = Window(sky_c.shape[0])
    if "ticket" not in st:
        st["ticket"] = torch.zeros(2, dtype=torch.int32, device=R.dev)      # the kernel leaves it at zero
    with torch.cuda.device(R.dev):
        rc = _lib().sdn_field_mlp(buf["feat"].data_ptr(), buf["dist"].data_ptr(), buf["label"].data_ptr(),
                                  buf["rayflag"].data_ptr(), st["packed_mx" if ct == 6 else "packed"].data_ptr(),
                                  st["consts"].data_ptr(), sky_c.data_ptr(), net_out.data_ptr(), n_rays, ns, ct, eps,
                                  passes.data_ptr() if passes is not None else None, 0, window.host(ray0, n_rays),
                                  sky_avg.data_ptr(), st["ticket"].data_ptr() if dynamic else None, _stream(R.dev))
    capi.check(rc, "sdn_field_mlp")


def mlp_from(R, buf, sky_c, sky_avg, n_rays, ns, window=None):
    """Second half of field_fused for an already encoded ray set (the pipelined trajectory path)."""
    st = R._fused_style or prepare_style(R)
    net_out = torch.empty((n_rays, 64), dtype=torch.float32, device=R.dev)
    _launch_mlp(R, buf, st, sky_c, sky_avg, net_out, n_rays, ns, window=window)
    return net_out


def single_kernel(R):
    """Whether the field runs as ONE kernel (sdn_field_render: every pass gathers its own features) or as encode_kernel +
    mlp_kernel with the features handed over through HBM.  Renderer.field_single_kernel, else SDN_FIELD_SINGLE_KERNEL, else
    the default below."""
    v = getattr(R, "field_single_kernel", None)
    if v is None:
        v = os.environ.get("SDN_FIELD_SINGLE_KERNEL", SINGLE_KERNEL_DEFAULT) not in ("0", "", "false")
    return bool(v)


SINGLE_KERNEL_DEFAULT = "1"   # same frame time as the two-kernel sequence (A/B), without its 10.8 GB/frame of HBM hand-off


def colour_skip(R):
    """Whether field_kernel skips the colour branch of passes whose 128 samples all have volume-rendering weight exactly zero
    (csrc/field.hip; bit-identical output).  Renderer.colour_skip, else SDN_COLOUR_SKIP, else on."""
    v = getattr(R, "colour_skip", None)
    if v is None:
        v = os.environ.get("SDN_COLOUR_SKIP", "1") not in ("0", "", "false")
    return bool(v)


def field_render(R, vid, d2, rd, cam_ori, sky_c, sky_avg, ns, passes=None, u=None, window=None, division="reciprocal",
                 net_out=None, aux=None, colour_passes=None):
    """The whole field of a ray set in one launch (csrc/field.hip field_kernel): sample placement + hash-grid lookup + render
    MLP + compositing.  Same arguments and the same bits as field_fused's encode -> mlp sequence; no feature buffer, so no
    ray chunking at any frame size.
    vid / d2 / rd / sky_c: tensors, or -- with a `window` -- plain device addresses (int) of frame-wide arrays of window.n_src
    rays (Generator._forward_perpix's tile views of the per-frame voxlib outputs are evaluated in place that way).
    cam_ori: host values, or a CUDA tensor -- then the kernel reads it from device memory (no device -> host copy).
    aux: None, or a dict: its keys (any of AUX_OUTPUTS: "weights", "depth", "sigma" [n_rays, ns], "colour" [n_rays, ns, 64],
    "sky_blended" [n_rays, 64], "nosky" u8 [n_rays]; an empty dict = "weights" + "depth") name the other return values of
    Generator._forward_perpix the launch should also produce; the dict receives the tensors.
    passes / colour_passes: optional u8 [ceil(n_rays / 32)] -- passes every 32-ray group went through / how many of them
    evaluated the colour branch (see colour_skip)."""
    sc = R._fused_scene or prepare_scene(R)
    st = R._fused_style or prepare_style(R)
    ct, eps = precision_profile(R)
    if aux is not None:
        eps = 0.0       # the per-sample outputs cover every sample: no early termination for this launch
    if ct == 2:
        raise ValueError("the single-kernel field supports colour_terms 3 and 6")
    if window is None:
        window = Window(vid.shape[0])
    n_rays = window.n_rays

    def addr(t, rows, what):
        if isinstance(t, int):
            return t
        assert t.is_cuda and t.device == R.dev and t.is_contiguous() and t.shape[rows] == window.n_src, what
        return t.data_ptr()

    if not isinstance(sky_c, int):
        sky_c = sky_c.contiguous()
    p_vid, p_d2, p_rd, p_sky = addr(vid, 0, "voxel_id"), addr(d2, 1, "depth2"), addr(rd, 0, "raydirs"), addr(sky_c, 0, "sky_c")
    sky_avg = torch.as_tensor(sky_avg).reshape(-1).to(device=R.dev, dtype=torch.float32).contiguous()
    assert sky_avg.numel() == 64
    if net_out is None:
        net_out = torch.empty((n_rays, 64), dtype=torch.float32, device=R.dev)
    buf = R.__dict__.setdefault("_fused_lin", {})
    if u is None:
        lin = buf.get(("det", ns))
        if lin is None:
            lin = buf[("det", ns)] = torch.linspace(0, 1, ns + 3)[1:-1].contiguous().to(R.dev)        # mc_utils.py:120
    else:
        assert u.is_cuda and u.dtype == torch.float32 and tuple(u.shape) == (n_rays, ns + 1) and u.is_contiguous()
        lin = buf.get(("strat", ns))
        if lin is None:
            lin = buf[("strat", ns)] = torch.linspace(0, 1, ns + 2)[:-1].contiguous().to(R.dev)      # mc_utils.py:124
    if "ticket" not in st:
        st["ticket"] = torch.zeros(2, dtype=torch.int32, device=R.dev)      # the kernel leaves it at zero
    ori_dev = None
    if isinstance(cam_ori, torch.Tensor) and cam_ori.is_cuda:
        ori_dev = cam_ori.detach().reshape(-1).to(torch.float32).contiguous()
        assert ori_dev.numel() == 3 and ori_dev.device == R.dev
        ori = np.zeros(3, np.float32)
    else:
        ori = np.asarray(cam_ori.detach().cpu().numpy() if isinstance(cam_ori, torch.Tensor) else cam_ori, np.float32).reshape(3)
    aux_c = None
    if aux is not None:
        shapes = {"weights": ((n_rays, ns), torch.float32), "depth": ((n_rays, ns), torch.float32), "sigma": ((n_rays, ns), torch.float32),
                  "colour": ((n_rays, ns, 64), torch.float32), "sky_blended": ((n_rays, 64), torch.float32), "nosky": ((n_rays,), torch.uint8)}
        want = list(aux) or ["weights", "depth"]
        assert all(k in shapes for k in want), want
        for k in want:
            aux[k] = torch.empty(shapes[k][0], dtype=shapes[k][1], device=R.dev)     # (with aux the kernel visits every ray and sample)
        aux_c = capi.FieldAux(**{k: aux[k].data_ptr() for k in want})
    # ray order: 8 x 4 pixel blocks whenever the launch is a whole window -- also a ragged one (990 columns: the reference's padded
    # frame), unless the caller's per-group arrays are sized for the row-major groups or the sampling is stochastic
    ragged = u is None and all(t is None or t.numel() >= window.n_groups(ragged=True) for t in (passes, colour_passes))
    skip = colour_skip(R)
    if colour_passes is not None or not skip:
        aux_c = aux_c or capi.FieldAux()
        aux_c.colour_passes = colour_passes.data_ptr() if colour_passes is not None else None
        aux_c.flags = 0 if skip else capi.FIELD_NO_COLOUR_SKIP
    with torch.cuda.device(R.dev):
        rc = _lib().sdn_field_render(p_vid, p_d2, p_rd, sc["lut"].data_ptr(), sc["table3"].data_ptr(),
                                     sc["T"], sc["scales"].data_ptr(), sc["genc"].ctypes.data, ori.ctypes.data,
                                     sc["dims"].ctypes.data, lin.data_ptr(), u.data_ptr() if u is not None else None, n_rays, R.M,
                                     ns, R.sample_depth, R.dists_scale, st["packed_mx" if ct == 6 else "packed"].data_ptr(),
                                     st["consts"].data_ptr(), p_sky, sky_avg.data_ptr(), net_out.data_ptr(), ct, eps,
                                     passes.data_ptr() if passes is not None else None, 0, window.host(0, n_rays, ragged),
                                     {"reciprocal": 0, "ieee": 1}[division], st["ticket"].data_ptr(),
                                     ori_dev.data_ptr() if ori_dev is not None else None,
                                     ctypes.byref(aux_c) if aux_c is not None else None, _stream(R.dev))
    capi.check(rc, "sdn_field_render")
    return net_out


def _per_ray_feat_bytes(ns):
    return _lib().sdn_field_feat_bytes(32, ns) // 32


def single_chunk(n_rays, ns):
    return n_rays * _per_ray_feat_bytes(ns) <= FEATURE_BUFFER_BYTES


def field_fused(R, vid, d2, rd, cam_ori, sky_c, sky_avg, ns, passes=None, u=None, window=None, division="reciprocal"):
    """net_out [n,64] for the rays of `window` (default: all) of the intersections vid [n_src,M] / d2 [2,n_src,M] /
    raydirs rd [n_src,3], sky features sky_c [n_src,64] and their frame mean sky_avg [64].
    Rays are independent, so very large frames (4K x 40 samples = 174 GB of features) go through in ray chunks that
    reuse one feature buffer; the headline frame (6.9 GB) is a single chunk."""
    st = R._fused_style or prepare_style(R)
    if window is None:
        window = Window(vid.shape[0])
    n_rays = window.n_rays
    vid, d2, rd, sky_c = vid.contiguous(), d2.contiguous(), rd.contiguous(), sky_c.contiguous()
    if single_kernel(R) and precision_profile(R)[0] != 2:
        return field_render(R, vid, d2, rd, cam_ori, sky_c, sky_avg, ns, passes=passes, u=u, window=window, division=division)
    sky_avg = torch.as_tensor(sky_avg).reshape(-1).to(device=R.dev, dtype=torch.float32).contiguous()   # (a host mean is accepted)
    net_out = torch.empty((n_rays, 64), dtype=torch.float32, device=R.dev)
    chunk = max(32, (FEATURE_BUFFER_BYTES // _per_ray_feat_bytes(ns)) // 32 * 32)     # whole 32-ray groups
    for r0 in range(0, n_rays, chunk):
        n = min(n_rays, r0 + chunk) - r0
        uc = u[r0:r0 + n].contiguous() if u is not None else None
        buf = encode(R, vid, d2, rd, cam_ori, ns, u=uc, window=window, ray0=r0, n_rays=n, division=division)
        _launch_mlp(R, buf, st, sky_c, sky_avg, net_out[r0:r0 + n], n, ns, passes[r0 // 32:] if passes is not None else None,
                    window=window, ray0=r0)
    return net_out


def time_mlp_kernel(R, vid, d2, rd, cam_ori, sky_c, sky_avg, ns, reps=5):
    """(samples per launch, avg ms) of mlp_kernel alone (features already encoded)."""
    from .renderer import _time_ms
    st = R._fused_style or prepare_style(R)
    n = vid.numel() // R.M
    vid, d2, rd = vid.reshape(n, R.M).contiguous(), d2.reshape(2, n, R.M).contiguous(), rd.reshape(n, 3).contiguous()
    buf = encode(R, vid, d2, rd, cam_ori, ns)
    sky_avg = sky_avg.reshape(-1).to(torch.float32).contiguous()
    net_out = torch.empty((n, 64), dtype=torch.float32, device=R.dev)
    sky_c = sky_c.contiguous()

    ms = _time_ms(lambda: _launch_mlp(R, buf, st, sky_c, sky_avg, net_out, n, ns), reps)
    hit = float((vid[:, 0] != 0).float().mean())
    # samples the kernel actually evaluates: it skips 32-ray groups (4 tiles of 8 consecutive rays) that hit nothing,
    # and the passes early termination removes
    g = torch.nn.functional.pad((vid[:, 0] != 0), (0, (-n) % 32)).view(-1, 32).any(dim=1)
    passes = torch.zeros(g.numel(), dtype=torch.uint8, device=R.dev)
    _launch_mlp(R, buf, st, sky_c, sky_avg, net_out, n, ns, passes)
    executed = int(passes.sum(dtype=torch.int64))
    nch = -(-ns // 4)
    return n * ns, ms, hit, dict(group_hit_fraction=float(g.float().mean()), evaluated_samples=executed * 128,
                                 passes_skipped_by_termination=int(g.sum()) * nch - executed)


def time_encode_kernel(R, vid, d2, rd, cam_ori, ns, reps=5):
    """(samples per launch, avg ms, algorithmic bytes per sample, kernel name) for the roofline record."""
    from .renderer import _time_ms
    n = vid.numel() // R.M
    vid, d2, rd = vid.reshape(n, R.M).contiguous(), d2.reshape(2, n, R.M).contiguous(), rd.reshape(n, 3).contiguous()
    buf = _buffers(R, n, ns)
    ms = _time_ms(lambda: encode(R, vid, d2, rd, cam_ori, ns, buf), reps)
    return n * ns, ms, 16404, "encode_kernel (collapsed 3-D table: 4096 B/sample actually gathered)"


def prepare_sky(R):
    """Pack the sky MLP for the current style code (fc_z_a(z) folded into fc1's bias)."""
    lib = _lib()
    w = R.w
    packed = torch.empty(lib.sdn_sky_packed_weight_bytes(), dtype=torch.uint8, device=R.dev)
    wh = [w[f"sky_net.fc{i}.weight"].contiguous() for i in (2, 3, 4, 5)]
    ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in wh])
    w1 = w["sky_net.fc1.weight"].contiguous()
    wc = w["sky_net.fc_out_c.weight"].contiguous()
    packed_mx = torch.empty_like(packed)
    with torch.cuda.device(R.dev):
        capi.check(lib.sdn_sky_pack_weights(w1.data_ptr(), ptrs, wc.data_ptr(), packed.data_ptr(), _stream(R.dev)))
        capi.check(lib.sdn_sky_pack_weights_mx(w1.data_ptr(), ptrs, wc.data_ptr(), packed_mx.data_ptr(), _stream(R.dev)))
    consts = torch.cat([w["sky_net.fc1.bias"] + R.sky_z.reshape(-1)] + [w[f"sky_net.fc{i}.bias"] for i in (2, 3, 4, 5)] +
                       [w["sky_net.fc_out_c.bias"]]).contiguous()
    assert consts.numel() == lib.sdn_sky_consts_floats()
    R._fused_sky = dict(packed=packed, packed_mx=packed_mx, consts=consts, keep=(wh, w1, wc))
    return R._fused_sky


def sky_terms(R):
    """Products of the sky MLP's hidden layers fc2..fc5: 3 = 3-term f16 split; 6 = f16 Whi.Xhi + block-scaled fp6 corrections (the
    colour layers' scheme: ~15 % less sky time, errors of the four layers stack to ~1e-4 on sky_c).  Renderer.sky_terms, else
    SDN_SKY_TERMS, else the per-style decision of Renderer.calibrate_style (`sky_terms_auto`), else 3."""
    t = getattr(R, "sky_terms", None)
    if t is None and "SDN_SKY_TERMS" in os.environ:
        t = int(os.environ["SDN_SKY_TERMS"])
    return t or getattr(R, "sky_terms_auto", None) or 3


def sky_fused(R, rd, encoded=False):
    """sky_c [R,64] and the frame mean sky_avg [1,64] for ray directions rd [R,3] (the mean is finished inside the kernel by
    its last workgroup: fixed summation order, no host-side reduction).
    encoded: rd is [R,33], rows that are already positional-encoded (SKYMLP.forward's own argument)."""
    sk = getattr(R, "_fused_sky", None) or prepare_sky(R)
    rd = rd.contiguous()
    n = rd.shape[0]
    assert rd.dim() == 2 and rd.shape[1] == (33 if encoded else 3) and rd.dtype == torch.float32 and rd.device == R.dev
    sky_c = torch.empty((n, 64), dtype=torch.float32, device=R.dev)
    part = torch.empty((_lib().sdn_sky_partial_rows(n, 0), 64), dtype=torch.float32, device=R.dev)
    sky_avg = torch.empty((1, 64), dtype=torch.float32, device=R.dev)
    if "counter" not in sk:
        sk["counter"] = torch.zeros(1, dtype=torch.int32, device=R.dev)     # the kernel leaves it at zero
    # hidden layers fc2..fc5: 3 = 3-term f16 split (default); 6 = f16 + fp6 corrections: 1.02 -> 0.87 ms per frame, but all
    # four hidden layers stack their ~2^-17 errors (1.1e-4 max on sky_c against 4.5e-6): opt-in
    terms = 3 if encoded else sky_terms(R)
    with torch.cuda.device(R.dev):
        capi.check(_lib().sdn_sky_mlp(rd.data_ptr(), sk["packed_mx" if terms == 6 else "packed"].data_ptr(), sk["consts"].data_ptr(),
                                      sky_c.data_ptr(), part.data_ptr(), n, 0, sky_avg.data_ptr(), sk["counter"].data_ptr(),
                                      terms, 1 if encoded else 0, _stream(R.dev)), "sdn_sky_mlp")
    return sky_c, sky_avg
