"""Frame renderer: the host-side mirror of Generator.inference_givenstyle's per-frame body
(imaginaire/generators/scenedreamer.py:573-628) on top of libsdnative.

Two interchangeable per-pixel paths produce net_out [1, Hp, Wp, 64] for the padded frame:

  * "unfused": the reference's op sequence -- voxlib.ray_voxel_intersection_perspective ->
    sample placement -> GridEncoder.forward -> render MLP -> volume rendering -- with the three
    native ops served by the drop-in HIP kernels and everything else by PyTorch ops on the GPU
    (exactly what the unmodified reference generator does when our shim modules are installed);
  * "fused": sample placement, hash-grid lookup, MLP and compositing inside libsdnative's field
    kernels (see csrc/field.hip), selected with mode="fused".

Differences from the reference's loop that do not change the result: every ray is evaluated once on
the full padded frame instead of in 40 overlapping 158-px tiles (the per-pixel field has no spatial
coupling), and the render CNN runs once on the padded frame and is cropped by pad/2 afterwards (its
receptive radius of 4 px is smaller than the 15-px crop, gancraft_base.py:180-190).  Per-style
constants (W * alpha, beta of every ModLinear; the label-bias table replacing the one-hot matmul)
are folded once per style code instead of once per tile.
"""
import json
import os
import warnings

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .camera import frame_intrinsics

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_label_lut():
    """minecraft block id -> reduced label (12 classes), mc_lbl_reduction.py:36-43 (data file)."""
    return json.load(open(os.path.join(_DATA, "mc2reduced.json")))


def _lrelu(x):
    return F.leaky_relu(x, 0.2)


# ---- per-style constants of the three networks.  `R` is a Renderer or any object with `w` (the reference's parameter names ->
# ---- device tensors), e.g. the module-level backends of modules.py that read a live nn.Module's parameters.
def fold_render_net(R, z):
    """LightningMLP for one style code z [1, style_dim]: ModLinear with N = 1 is a plain layer W' = W * alpha (per input
    channel) with bias beta (layers.py:247-269); one-hot(label) @ fc_m_a^T + fc_1.bias is a row lookup in a [12, 256] table
    (use_seg = False: every row is fc_1.bias)."""
    w = R.w
    with torch.no_grad():
        R.mod = {}
        for i in (2, 3, 4, 5, 6):
            n = f"render_net.fc_{i}"
            alpha = F.linear(z, w[n + ".weight_alpha"], w[n + ".bias_alpha"])       # [1, in]
            beta = F.linear(z, w[n + ".weight_beta"], w[n + ".bias_beta"])          # [1, out]
            R.mod[i] = ((w[n + ".weight"] * alpha).contiguous(), beta[0].contiguous())
        b1 = w["render_net.fc_1.bias"][None, :]
        if "render_net.fc_m_a.weight" in w:
            R.label_bias = (w["render_net.fc_m_a.weight"].t() + b1).contiguous()
        else:
            R.label_bias = b1.expand(12, -1).contiguous()
    R._fused_style = None


def fold_sky_net(R, z):
    """SKYMLP: the style term fc_z_a(z) [1, 256] joins fc1's bias (gancraft_base.py:158-162)."""
    with torch.no_grad():
        R.sky_z = F.linear(z, R.w["sky_net.fc_z_a.weight"])
    R._fused_sky = None


def fold_denoiser(R, z):
    """RenderCNN: the four FiLM vectors fc_z_cond(z) [1, 1024] (gancraft_base.py:203-204)."""
    with torch.no_grad():
        R.cnn_adapt = F.linear(z, R.w["denoiser.fc_z_cond.weight"], R.w["denoiser.fc_z_cond.bias"])
    R.cnn_calibration = None      # the FiLM vectors changed: the render CNN's precision gate is re-evaluated


class Renderer:
    def __init__(self, weights, scene, device="cuda", num_blocks_early_stop=6, sample_depth=3.0, dists_scale=0.25,
                 pad=30):
        self.dev = torch.device(device)
        if self.dev.type == "cuda" and self.dev.index is None:      # "cuda" -> "cuda:<current>": tensor.device carries the index
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.w = {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))).to(self.dev)
                  for k, v in weights.items()}
        lut = load_label_lut()
        t = torch.tensor(lut["lut"], dtype=torch.long)
        t[t == lut["ignore_id"]] = lut["dirt_id"]  # mc2reduced(ign2dirt=True), mc_utils.py:241-246
        self.lut = t.to(self.dev)
        self.M = num_blocks_early_stop
        self.sample_depth = float(sample_depth)
        self.dists_scale = float(dists_scale)
        self.pad = pad
        offs = self.w["hash_encoder.offsets"]
        self.grid_L = offs.numel() - 1
        self.grid_S = float(np.log2(np.exp2(np.log2(2048 / 16) / (self.grid_L - 1))))
        self.timings = {}
        self.set_scene(scene)

    # ------------------------------------------------------------------ once per scene / style
    def set_scene(self, scene):
        self.scene = scene
        # compact scenes (scene.CompactScene: uint8 palette indices + int32 palette, 4x smaller) are walked as they are;
        # the reference's int32 volume otherwise.  `volume` is what the ray marcher gets, `voxel_dims` its extent.
        self.palette = None
        if getattr(scene, "voxel_u8", None) is not None:
            self.volume = scene.voxel_u8.to(self.dev)
            self.palette = scene.palette.to(self.dev).contiguous()
            self.max_block_id = int(self.palette.max())
        else:
            self.volume = scene.voxel_t.to(self.dev)
            self.max_block_id = int(self.volume.max()) if self.volume.numel() else 0
            if self.volume.numel() and int(self.volume.min()) < 0:
                raise RuntimeError("negative block id in the scene volume")
        self.voxel_dims = tuple(int(v) for v in self.volume.shape)
        if self.volume.is_cuda:
            ops.voxel_occupancy(self.volume)   # built here, on the caller's stream, before any side-stream ray casting
        w = self.w
        # (deterministic convolution algorithms: global_enc feeds every sample of every frame, and every rank of a sharded
        # trajectory computes it for itself -- MIOpen's default solver choice is not reproducible from call to call)
        with torch.no_grad(), warnings.catch_warnings(), \
                torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True):  # ConditionalHashGrid.forward, layers.py:40-55
            warnings.simplefilter("ignore")
            h = _lrelu(F.conv2d(scene.current_height_map.to(self.dev), w["world_encoder.hconv_head.weight"],
                                w["world_encoder.hconv_head.bias"], stride=2, padding=1))
            s = _lrelu(F.conv2d(scene.current_semantic_map.to(self.dev), w["world_encoder.sconv_head.weight"],
                                w["world_encoder.sconv_head.bias"], stride=2, padding=1))
            x = torch.cat([h, s], dim=1)
            for i in range(5):
                x = F.relu(F.conv2d(x, w[f"world_encoder.conv_blocks.{i}.layers.0.weight"], None, stride=1, padding=1))
                x = F.relu(F.conv2d(x, w[f"world_encoder.conv_blocks.{i}.layers.2.weight"], None, stride=2, padding=1))
                x = _lrelu(x)
            x = x.permute(0, 2, 3, 1)
            x = x.reshape(x.shape[0], -1, x.shape[-1]).mean(dim=1)
            x = _lrelu(F.linear(x, w["world_encoder.fc1.weight"], w["world_encoder.fc1.bias"]))
            self.global_enc = torch.tanh(F.linear(x, w["world_encoder.fc2.weight"], w["world_encoder.fc2.bias"]))
        self._fused_scene = None
        self.field_gate = None           # (the collapsed table changes with global_enc)
        self.colour_terms_auto = None
        self.sky_terms_auto = None

    def set_style(self, style):
        w = self.w
        with torch.no_grad():
            z = F.normalize(torch.as_tensor(style, dtype=torch.float32, device=self.dev), p=2, dim=-1)
            for i in range(5):  # StyleMLP.forward, gancraft_base.py:113-126
                z = _lrelu(F.linear(z, w[f"style_net.fc_layers.{i}.weight"], w[f"style_net.fc_layers.{i}.bias"]))
            z = _lrelu(F.linear(z, w["style_net.fc_out.weight"], w["style_net.fc_out.bias"]))
        self.set_style_code(z)

    def set_style_code(self, z):
        """Fold an intermediate style code z [1,256] (= style_net(style)) into per-style constants."""
        with torch.no_grad():
            z = torch.as_tensor(z, dtype=torch.float32, device=self.dev).reshape(1, -1)
            self.z = z
            fold_render_net(self, z)
            fold_sky_net(self, z)
            fold_denoiser(self, z)
        self.field_gate = None           # the per-style precision gates of the field (calibrate_field) are re-evaluated
        self.colour_terms_auto = None
        self.sky_terms_auto = None

    # ------------------------------------------------------------------ stages
    def cast_rays(self, pose, resolution_hw):
        cam_ori, cam_dir, cam_up, cam_f = pose
        f, c, cam_res = frame_intrinsics(cam_f, resolution_hw, self.pad)
        vid, d2, rd = ops.ray_voxel_intersection_perspective(self.volume, cam_ori, cam_dir, cam_up, f, c, cam_res,
                                                             self.M, palette=self.palette)
        return vid, d2, rd, cam_res

    @property
    def voxel_t(self):
        """The int32 block-id volume (expanded on demand for a compact scene)."""
        return self.volume if self.palette is None else self.scene.voxel_t

    def sky_features(self, raydirs):
        """sky_net(PE(raydirs)) for every ray: [R,3] -> [R,64] (scenedreamer.py:368-370, gancraft_base.py:150-169)."""
        w = self.w
        pe = ops.positional_encoding(raydirs.contiguous(), 5, -1, True)
        y = _lrelu(F.linear(pe, w["sky_net.fc1.weight"], w["sky_net.fc1.bias"]) + self.sky_z)
        for i in (2, 3, 4, 5):
            y = _lrelu(F.linear(y, w[f"sky_net.fc{i}.weight"], w[f"sky_net.fc{i}.bias"]))
        return F.linear(y, w["sky_net.fc_out_c.weight"], w["sky_net.fc_out_c.bias"])

    def place_samples(self, depth2, ns):
        """sample_depth_batched, deterministic, no box boundaries (mc_utils.py:82-151).
        depth2 [2,R,M] -> depth [R,ns], dists [R,ns], box index [R,ns]."""
        t, t2 = depth2[0], depth2[1]
        d = t2 - t
        d = torch.where(torch.isnan(d), torch.zeros_like(d), d)
        accu = torch.cumsum(d, dim=-1)
        total = accu[:, -1:].clamp(max=self.sample_depth)
        lin = torch.linspace(0, 1, ns + 3)[1:-1].to(self.dev)   # nsamples = ns+1 stratified points
        s = lin[None, :] * total
        mid = (s[:, 1:] + s[:, :-1]) / 2
        nd = s[:, 1:] - s[:, :-1]
        idx = (mid[:, None, :] > accu[:, :, None]).sum(dim=1)
        gaps = torch.cumsum(t[:, 1:] - t2[:, :-1], dim=-1)
        heads = torch.cat([t[:, :1], gaps + t[:, :1]], dim=-1)
        depth = torch.gather(heads, 1, idx) + mid
        return depth, nd, idx

    def field_unfused(self, voxel_id, depth2, raydirs, cam_ori, sky_c, sky_avg, ns, placement="torch"):
        """Per-ray feature net_out [R,64] from [R,M] intersections (scenedreamer.py:313-430).
        placement: "torch" = sample placement by PyTorch ops on the GPU, what the unmodified reference does on the op shims
        (torch.cumsum accumulates in float32 on the GPU, in double on the CPU: the two reference paths place some samples 1 ulp
        apart, which the fine grid levels and the density head turn into net_out differences of a few 1e-4 at isolated rays);
        "kernel" = placement by sdn_sample_depth, the device function the fused kernel uses (the CPU reference's arithmetic) -- the
        fp32 twin calibrate_style compares with, so that what it measures is the reduced-precision ARITHMETIC of the fused path
        and not the reference's own placement chaos."""
        w = self.w
        if placement == "kernel":
            R_ = depth2.shape[1]
            depth, nd, idx = ops.sample_depth_batched(depth2.reshape(2, R_, 1, self.M, 1).unsqueeze(0).contiguous(), ns + 1, deterministic=True,
                                                      use_box_boundaries=False, sample_depth=self.sample_depth)
            depth, nd, idx = depth.reshape(R_, ns), nd.reshape(R_, ns), idx.reshape(R_, ns).clamp(max=self.M - 1)
        else:
            depth, nd, idx = self.place_samples(depth2, ns)
        depth = torch.where(torch.isnan(depth) | torch.isinf(depth), torch.zeros_like(depth), depth)
        wc = raydirs[:, None, :] * depth[:, :, None] + cam_ori[None, None, :]
        lab = torch.gather(self.lut[voxel_id.long()], 1, idx)
        delim = torch.tensor([float(v) for v in self.voxel_dims], device=self.dev)
        n = wc / delim * 2 - 1
        x5 = torch.cat([n, self.global_enc[:, None, :].expand(n.shape[0], n.shape[1], 2)], dim=-1)
        x5 = ((x5 + 1) / 2).reshape(-1, 5).contiguous()           # GridEncoder.forward, grid.py:144
        B = x5.shape[0]
        feats = torch.empty(self.grid_L, B, 8, device=self.dev)
        ops.grid_encode_forward(x5, w["hash_encoder.embeddings"], w["hash_encoder.offsets"], feats, B, 5, 8,
                                self.grid_L, self.grid_S, 16, False, torch.empty(1, device=self.dev), 0, False)
        feats = feats.permute(1, 0, 2).reshape(B, self.grid_L * 8)
        f = _lrelu(F.linear(feats, w["render_net.fc_1.weight"]) + self.label_bias[lab.reshape(-1)])
        for i in (2, 3, 4):
            f = _lrelu(torch.addmm(self.mod[i][1], f, self.mod[i][0].t()))
        sigma = F.linear(f, w["render_net.fc_sigma.weight"], w["render_net.fc_sigma.bias"]).reshape(-1, ns)
        for i in (5, 6):
            f = _lrelu(torch.addmm(self.mod[i][1], f, self.mod[i][0].t()))
        color = F.linear(f, w["render_net.fc_out_c.weight"], w["render_net.fc_out_c.bias"]).reshape(-1, ns, 64)
        # volum_rendering_relu (mc_utils.py:154-161) + compositing (scenedreamer.py:373-413)
        fe = F.relu(sigma) * (nd * self.dists_scale)
        # exclusive cumsum as roll(cumsum) with a zero head (mc_utils.py:75-79)
        excl = torch.cat([torch.zeros_like(fe[:, :1]), torch.cumsum(fe, dim=-1)[:, :-1]], dim=-1)
        wts = (1 - torch.exp(-fe)) * torch.exp(-excl)
        sky_only = voxel_id[:, :1] == 0
        wts = wts * (~sky_only).float()
        T = wts.sum(dim=-1, keepdim=True)
        is_gnd = (wc[:, :, 0] <= 1.0).any(dim=-1, keepdim=True)
        nosky = ((voxel_id[:, -1:] != 0) | is_gnd).float()
        sky = sky_c * (1.0 - nosky) + sky_avg * nosky
        rgb = torch.clamp(color, -1, 1) + 1
        rgb_sky = torch.clamp(sky, -1, 1) + 1
        return (wts[:, :, None] * rgb).sum(dim=1) + (1.0 - T) * rgb_sky - 1

    def render_cnn(self, net_out):
        """_forward_global + RenderCNN (gancraft_base.py:588-603, :202-225): [1,Hp,Wp,64] -> [1,3,Hp,Wp]."""
        w = self.w
        a = torch.chunk(self.cnn_adapt, 4, dim=-1)
        mod = lambda v, s, b: v * (s[..., None, None] + 1) + b[..., None, None]
        cv = lambda v, n, p: F.conv2d(v, w[f"denoiser.{n}.weight"], w.get(f"denoiser.{n}.bias"), padding=p)
        x = net_out.permute(0, 3, 1, 2).contiguous()
        y = _lrelu(cv(x, "conv1", 0))
        y = y + cv(_lrelu(cv(y, "conv2a", 1)), "conv2b", 1)
        y = _lrelu(mod(y, a[0], a[1]))
        y = y + cv(_lrelu(cv(y, "conv3a", 1)), "conv3b", 1)
        y = _lrelu(mod(y, a[2], a[3]))
        y = y + cv(_lrelu(cv(y, "conv4a", 0)), "conv4b", 0)
        y = _lrelu(y)
        return torch.tanh(cv(y, "conv4", 0))

    # ------------------------------------------------------------------ measurement
    def set_precision(self, cnn_terms3x3=None, colour_terms=None, term_eps=None):
        """Precision profile of the MFMA kernels (None = the default of the environment / library):
        cnn_terms3x3: f16 product terms of the four 3x3 convolutions: 1, 3, a per-layer form like "1113" (cnn.CNN_LADDER), or
                      None = "auto" (the cheapest rung of the ladder that passes the per-style calibration -- see mfma_cnn);
        colour_terms: products of the colour layers fc_5 / fc_6: 6 (default: f16 + fp6 corrections), 3 or 2 (fused.precision_profile);
        term_eps: early ray termination threshold on the transmittance, 0 = off (default)."""
        self.cnn_terms3x3, self.colour_terms, self.term_eps = cnn_terms3x3, colour_terms, term_eps
        self._mfma_cnns = {}
        self.cnn_calibration = None
        self.field_gate = None
        self.colour_terms_auto = None
        self.sky_terms_auto = None

    # ------------------------------------------------------------------ per-style precision gates
    def calibrate_style(self, pose, resolution_hw, num_samples, more_poses=()):
        """calibrate_one on `pose` and on every pose of `more_poses` (the trajectory loop adds the middle pose of the trajectory:
        the errors depend on what the camera sees), the measurements combined with MAX, then adopt_precision -- the decision a
        multi-rank job reaches by reducing the same measurements over its ranks (dist.agree_precision)."""
        meas = self.calibrate_one(pose, resolution_hw, num_samples)
        for p2 in more_poses:
            m2 = self.calibrate_one(p2, resolution_hw, num_samples)
            for k, v in m2.items():
                if isinstance(v, dict):
                    meas[k] = {kk: max(vv, meas[k].get(kk, vv)) if isinstance(vv, float) else vv for kk, vv in v.items()}
                elif isinstance(v, float):
                    meas[k] = max(v, meas.get(k, v))
            meas["poses"] = meas.get("poses", 1) + 1
        return self.adopt_precision(meas)

    def calibrate_one(self, pose, resolution_hw, num_samples, crop_px=None):
        """Measure END TO END, for the CURRENT weights and style, what the reduced-precision choices of the fused path cost, and
        decide.  A window of one frame (`pose`) is rendered by the reference's op sequence in fp32 (field_unfused + render_cnn:
        PyTorch fp32 + the drop-in HIP ops -- the path the CPU-oracle tests validate; samples placed by the fused kernel's own
        device function, see field_unfused) and by the candidates; the cheapest candidate inside the bounds is adopted:

          colour layers fc_5 / fc_6: f16 + MX-fp6 corrections (colour_terms 6) if net_out stays within COLOUR_AUTO_BOUND of the
            3-term evaluation, else the 3-term split;
          the fused field (3-term f16 split, f32 accumulate): net_out against the fp32 net_out, bound FIELD_AUTO_BOUND -- above
            it the style is served by the fp32 op sequence (`path: "unfused"`): slow, but inside the tolerance;
          render CNN 3x3 layers: the cheapest rung of cnn.CNN_LADDER -- all four layers ONE f16 product; conv3b 3-term ("1113");
            conv3a + conv3b 3-term ("1133"); all 3-term -- whose image stays within CNN_AUTO_BOUND of the 3-term image AND whose
            MEASURED total error against the fp32 image (field error included) stays within IMAGE_AUTO_BOUND; if not even the
            3-term image is within IMAGE_AUTO_BOUND, the fp32 path.

        The window (round 6; `crop_px`, default CAL_CROP = 256 output pixels square, 0 = the whole frame as in rounds 4-5): the fp32
        twin of a whole 960x540x24 frame is 0.25 s of GPU time per pose -- with two poses more than half of a 40-frame trajectory
        (0.72 s).  The field is evaluated per ray and the CNN's receptive radius is 4 px, so any window of the frame is a valid
        sample of both; the window is put where the frame's content changes most from pixel to pixel (box sum of first-hit block-id
        changes and depth steps, straight from the ray caster's output: silhouettes and material boundaries, where net_out -- and
        with it the f16 rounding of the one-product 3x3 layers -- varies most).  A maximum over 1/8 of the pixels under-estimates
        the frame's (extreme-value growth ~ sqrt(2 ln N): 1.08 here), so every window-measured maximum is charged times
        CAL_CROP_FACTOR = 1.15 before it meets a bound (`raw` keeps the measured values).  The fused sky MLP runs on every ray of
        the padded frame (its frame mean needs them), its fp32 twin on the window's rays.

        The errors depend on the loaded weights (the density head amplifies hidden-activation error; 3x3 gains compound over
        four layers): tests/test_precision_gates_gpu.py scales them until every gate closes.  Explicit settings (set_precision,
        SDN_MLP_COLOUR_TERMS, SDN_CNN_TERMS) are measured but not overridden.  Returns the measurements; calibrate_style turns
        them into the records `field_gate`, `cnn_calibration` (bench.py writes both to bench_detail.json)."""
        from . import fused
        H, W = resolution_hw
        if H * W > CAL_MAX_PIXELS:
            f = (CAL_MAX_PIXELS / float(H * W)) ** 0.5
            H, W = max(8, int(H * f)), max(8, int(W * f))
        if crop_px is None:
            crop_px = int(os.environ.get("SDN_CAL_CROP", CAL_CROP))
        crop = self.pad // 2
        import time as _time
        phases, _t = {}, [None]

        def tick(name):          # SDN_CAL_TIMING=1: wall clock per phase (synchronised) -> meas["timing_ms"] (tools/cal_timing.py)
            if os.environ.get("SDN_CAL_TIMING"):
                torch.cuda.synchronize()
                now = _time.perf_counter()
                if _t[0] is not None and name:
                    phases[name] = phases.get(name, 0.0) + 1000.0 * (now - _t[0])
                _t[0] = now
        with torch.no_grad():
            tick(None)
            vid, d2, rd, (H0, W0) = self.cast_rays(pose, (H, W))
            n = H0 * W0
            vid, d2, rd = vid.view(n, self.M), d2.view(2, n, self.M), rd.view(n, 3)
            ori = torch.as_tensor(pose[0], dtype=torch.float32).reshape(3)
            tick("cast rays")
            inner = (lambda im: im[:, :, crop:-crop, crop:-crop]) if crop else (lambda im: im)
            # ---- the window: where the frame's content changes most from pixel to pixel -- silhouettes, material boundaries, depth
            #      steps of the first hit (from the ray caster's output: no field evaluation needed) -- is where net_out varies most
            explicit_ct = getattr(self, "colour_terms", None)
            if explicit_ct is None and "SDN_MLP_COLOUR_TERMS" in os.environ:
                explicit_ct = int(os.environ["SDN_MLP_COLOUR_TERMS"])
            saved = getattr(self, "colour_terms", None)
            Hc, Wc, r0, c0 = H0, W0, 0, 0
            windowed = bool(crop_px) and (H0 > crop_px + self.pad + 32 or W0 > crop_px + self.pad + 32)
            if windowed:
                Hc, Wc = min(H0, crop_px + self.pad), min(W0, crop_px + self.pad)
                v0 = vid[:, 0].view(H0, W0)
                t0 = torch.nan_to_num(d2[0][:, 0], nan=-64.0).view(H0, W0)
                g = torch.zeros(H0, W0, device=self.dev)
                g[1:] += (v0[1:] != v0[:-1]).float() + ((t0[1:] - t0[:-1]).abs() > 1.0).float()
                g[:, 1:] += (v0[:, 1:] != v0[:, :-1]).float() + ((t0[:, 1:] - t0[:, :-1]).abs() > 1.0).float()
                g += 1e-3 * (v0 != 0).float()           # (ties: prefer ground to sky)
                r0, c0 = _busiest_window(g, Hc, Wc)
                del g
            tick("window choice")
            nc = Hc * Wc
            cut = lambda t, last: t.view(H0, W0, last)[r0:r0 + Hc, c0:c0 + Wc].reshape(nc, last).contiguous()
            if windowed:
                vid_c, rd_c = cut(vid, self.M), cut(rd, 3)
                d2_c = torch.stack([cut(d2[0], self.M), cut(d2[1], self.M)]).contiguous()
            else:
                vid_c, rd_c, d2_c = vid, rd, d2
            # ---- the sky MLP: hidden layers as f16 + fp6 corrections if its features stay within SKY_AUTO_BOUND of the fp32 ones.
            #      The fused forms run on every ray of the padded frame (the frame mean needs them; 0.8 ms each); the fp32 twin on the
            #      window's rays, its frame mean taken from the 3-term evaluation (4e-6 per feature before averaging 564 k of them)
            ori_dev = ori.to(self.dev)
            explicit_sky = getattr(self, "sky_terms", None) or (int(os.environ["SDN_SKY_TERMS"]) if "SDN_SKY_TERMS" in os.environ else None)
            sky32_c = self.sky_features(rd_c)
            self.sky_terms_auto = None
            sky_c, sky_avg = fused.sky_fused(self, rd)
            savg32 = sky_avg.reshape(1, 64) if (windowed and fused.sky_terms(self) == 3) else None
            cut_sky = (lambda t: cut(t, 64)) if windowed else (lambda t: t)
            sky_err = {fused.sky_terms(self): float((cut_sky(sky_c) - sky32_c).abs().max()) * (CAL_CROP_FACTOR if windowed else 1.0)}
            if explicit_sky is None:
                self.sky_terms_auto = 6
                c6, a6 = fused.sky_fused(self, rd)
                sky_err[6] = float((cut_sky(c6) - sky32_c).abs().max()) * (CAL_CROP_FACTOR if windowed else 1.0)
                if sky_err[6] <= SKY_AUTO_BOUND:
                    sky_c, sky_avg = c6, a6
                else:
                    self.sky_terms_auto = None
            if savg32 is None:      # whole frame (or an explicit fp6 sky): the fp32 mean over every ray
                savg32 = (sky32_c if not windowed else self.sky_features(rd)).mean(dim=0, keepdim=True)
            skyc_c = cut_sky(sky_c)
            tick("sky (fp32 twin on the window, 2 fused forms on the frame)")
            # ---- the fp32 twin of the window
            ref_no = torch.cat([self.field_unfused(vid_c[r:r + CAL_CHUNK], d2_c[:, r:r + CAL_CHUNK].contiguous(), rd_c[r:r + CAL_CHUNK], ori_dev,
                                                   sky32_c[r:r + CAL_CHUNK], savg32, num_samples, placement="kernel")
                                for r in range(0, nc, CAL_CHUNK)], dim=0)
            tick("fp32 field twin")
            ref_img = inner(self.render_cnn(ref_no.view(1, Hc, Wc, 64)))
            tick("fp32 CNN twin")
            # ---- the fused field on the window's rays
            no = {}
            try:
                for ct in ((explicit_ct,) if explicit_ct is not None else (6, 3)):
                    self.colour_terms = ct
                    no[ct] = fused.field_fused(self, vid_c, d2_c, rd_c, ori, skyc_c, sky_avg, num_samples)
            finally:
                self.colour_terms = saved
            k_ev = CAL_CROP_FACTOR if windowed else 1.0       # window maxima are charged with the extreme-value factor
            raw = {"field_err": {ct: float((v - ref_no).abs().max()) for ct, v in no.items()}}
            meas = {"field_err": {ct: e * k_ev for ct, e in raw["field_err"].items()}, "sky_err": sky_err, "explicit_sky": explicit_sky}
            if explicit_ct is None:
                raw["colour_diff"] = float((no[6] - no[3]).abs().max())
                meas["colour_diff"] = raw["colour_diff"] * k_ev
            ct = explicit_ct if explicit_ct is not None else (6 if meas["colour_diff"] <= COLOUR_AUTO_BOUND else 3)
            tick("fused field, 2 colour forms")
            # ---- the render CNN on the chosen field's output
            from .cnn import CNN_LADDER, form_key
            explicit_t = getattr(self, "cnn_terms3x3", None)
            if explicit_t is None and "SDN_CNN_TERMS" in os.environ:
                explicit_t = form_key(os.environ["SDN_CNN_TERMS"])       # "1", "3" or a per-layer form like "1113"
            x = no[ct].view(1, Hc, Wc, 64)
            if explicit_t is not None:
                explicit_t = form_key(explicit_t)
            # (every rung is measured, whichever is adopted: adopt_precision must be a function of `meas` alone, so that the ranks of
            #  a multi-GPU job can reduce the measurements and reach the same decision)
            imgs = {t: inner(self._cnn_form(t)(x)).clone() for t in ((explicit_t,) if explicit_t is not None else CNN_LADDER)}
            raw["image_err"] = {t: float((im - ref_img).abs().max()) for t, im in imgs.items()}
            meas["image_err"] = {t: e * k_ev for t, e in raw["image_err"].items()}
            if explicit_t is None:
                raw["cnn_diffs"] = {t: float((imgs[t] - imgs[3]).abs().max()) for t in CNN_LADDER if t != 3}
                meas["cnn_diffs"] = {t: e * k_ev for t, e in raw["cnn_diffs"].items()}
                meas["cnn_diff"] = meas["cnn_diffs"][1]
            tick("MFMA CNN rungs")
            if windowed:        # (a window's activation planes are not the frame's: drop them, the packed weights stay)
                for c in self.__dict__.get("_mfma_cnns", {}).values():
                    c._planes.pop((Hc, Wc), None)
        meas.update(explicit_colour=explicit_ct, explicit_cnn=explicit_t, pixels=int((Hc - 2 * crop) * (Wc - 2 * crop)), rays=int(nc), samples_per_ray=int(num_samples),
                    frame=f"{W}x{H} (+{self.pad}-px apron), {num_samples} samples/ray" +
                          (f"; window {Wc - 2 * crop}x{Hc - 2 * crop} at ({r0},{c0}), maxima x {CAL_CROP_FACTOR}" if windowed else ""),
                    window=([r0, c0, Hc, Wc] if windowed else None), raw=raw if windowed else None)
        if phases:
            meas["timing_ms"] = phases
        return meas

    def adopt_precision(self, meas):
        """Decisions that follow from calibrate_style's measurements (a pure function of `meas`: dist.agree_precision reduces the
        measurements over the ranks with MAX and lets every rank adopt the same ones)."""
        ect, et = meas["explicit_colour"], meas["explicit_cnn"]
        ct = ect if ect is not None else (6 if meas["colour_diff"] <= COLOUR_AUTO_BOUND else 3)
        ferr = meas["field_err"][ct]
        path = "fused" if ferr <= FIELD_AUTO_BOUND else "unfused"
        bound = float(getattr(self, "cnn_auto_bound", None) or CNN_AUTO_BOUND)
        ierr = meas["image_err"]
        cal = None
        if et is None:
            from .cnn import CNN_LADDER
            diffs = dict(meas.get("cnn_diffs") or {1: meas["cnn_diff"]})
            t = 3
            for cand in CNN_LADDER:         # cheapest first
                if cand != 3 and cand in diffs and cand in ierr and diffs[cand] <= bound and ierr[cand] <= IMAGE_AUTO_BOUND:
                    t = cand
                    break
            if t == 3 and ierr[3] > IMAGE_AUTO_BOUND:
                path = "unfused"
            cal = {"terms3x3": t, "max_abs_diff_1term_vs_3term": meas["cnn_diff"], "bound": bound,
                   "max_abs_diff_vs_3term": {str(k): v for k, v in diffs.items()},
                   "image_err_vs_fp32": {("1-term" if k == 1 else "3-term" if k == 3 else str(k)): v for k, v in ierr.items()},
                   "image_bound": IMAGE_AUTO_BOUND, "ladder": [str(k) for k in CNN_LADDER],
                   "pixels": CNN_CAL_PIXELS, "pixels_measured": meas["pixels"], "calls": 1, "frame": meas["frame"], "measured": "end to end (calibrate_style)"}
        self.field_gate = {
            "path": path, "max_abs_err_vs_fp32": ferr, "bound": FIELD_AUTO_BOUND, "quantity": "net_out (per-ray feature, range [-1, 1])",
            "colour": ({"terms": ct, "set_explicitly": True} if ect is not None else
                       {"terms": ct, "max_abs_diff_fp6_vs_3term": meas["colour_diff"], "bound": COLOUR_AUTO_BOUND}),
            "image_err_vs_fp32": ierr[et if et is not None else cal["terms3x3"]], "image_bound": IMAGE_AUTO_BOUND,
            "sky": {"hidden_terms": (meas.get("explicit_sky") or (6 if meas.get("sky_err", {}).get(6, 1.0) <= SKY_AUTO_BOUND else 3)),
                    "max_abs_err_vs_fp32": meas.get("sky_err"), "bound": SKY_AUTO_BOUND, "set_explicitly": meas.get("explicit_sky") is not None},
            "rays": meas["rays"], "samples_per_ray": meas["samples_per_ray"], "frame": meas["frame"], "measurements": meas}
        self.colour_terms_auto = ct if ect is None else None
        if "sky_err" in meas:
            self.sky_terms_auto = 6 if (meas.get("explicit_sky") is None and meas["sky_err"].get(6, 1.0) <= SKY_AUTO_BOUND) else None
        if cal is not None:
            self.cnn_calibration = cal
            self._drop_other_cnn_planes(cal["terms3x3"])
        return self.field_gate

    def recheck_cnn(self, net_out):
        """Once per style, on a LATER frame than the ones calibrate_style saw (the trajectory loop passes its last frame's
        net_out [1,Hp,Wp,64]): the adopted 3x3 rung against the 3-term form on the window where this frame's net_out varies
        most, maximum charged like calibrate_one's.  Two calibration poses decide for a whole trajectory and the margins are thin
        by construction (a style may adopt a rung at 4.97e-4 against 5e-4): if the later pose disagrees, warn and step up the
        ladder for the rest of the style.  ~1.5 ms + one host read, once per style."""
        cal = getattr(self, "cnn_calibration", None)
        if (not cal or cal.get("recheck") is not None or cal["terms3x3"] == 3 or getattr(self, "cnn_terms3x3", None) is not None
                or "SDN_CNN_TERMS" in os.environ or os.environ.get("SDN_CNN_RECHECK", "1") == "0"):
            return None
        from .cnn import CNN_LADDER, form_key
        bound = float(cal.get("bound") or CNN_AUTO_BOUND)
        _, Hp, Wp, _ = net_out.shape
        side = CAL_CROP + 2 * CNN_HALO
        Hc, Wc = min(Hp, side), min(Wp, side)
        with torch.no_grad():
            v = net_out[0]
            g = torch.zeros(Hp, Wp, device=net_out.device)
            g[1:] += (v[1:] - v[:-1]).abs().sum(dim=-1)
            g[:, 1:] += (v[:, 1:] - v[:, :-1]).abs().sum(dim=-1)
            r0, c0 = _busiest_window(g, Hc, Wc)
            x = net_out[:, r0:r0 + Hc, c0:c0 + Wc].contiguous()
            ref3 = self._cnn_form(3)(x).clone()
            ladder = list(CNN_LADDER)
            start = ladder.index(form_key(cal["terms3x3"]))
            seen = {}
            adopted = 3
            for cand in ladder[start:]:
                if cand == 3:
                    break
                seen[str(cand)] = float((self._cnn_form(cand)(x) - ref3).abs().max()) * CAL_CROP_FACTOR
                if seen[str(cand)] <= bound:
                    adopted = cand
                    break
            for c in self.__dict__.get("_mfma_cnns", {}).values():      # the window's planes are not the frame's
                c._planes.pop((Hc, Wc), None)
        cal["recheck"] = {"window": [r0, c0, Hc, Wc], "max_abs_diff_vs_3term_charged": seen, "bound": bound, "adopted_before": cal["terms3x3"],
                          "adopted_after": adopted}
        if adopted != cal["terms3x3"]:
            warnings.warn(f"render CNN: rung {cal['terms3x3']} adopted on the calibration poses measures {seen} > {bound:g} on a later frame of the "
                          f"style; stepping up to {adopted}")
            cal["terms3x3"] = adopted
        return cal["recheck"]

    def _drop_other_cnn_planes(self, keep):
        """The forms not chosen keep their packed weights (9 MB each), not their activation planes (1.2 GB at 960x540)."""
        for k, c in self.__dict__.get("_mfma_cnns", {}).items():
            if k != keep:
                c._planes.clear()

    def field_falls_back(self):
        g = getattr(self, "field_gate", None)
        return bool(g) and g.get("path") == "unfused"

    def _cnn_form(self, terms3x3):
        from .cnn import MfmaCNN, form_key
        cache = self.__dict__.setdefault("_mfma_cnns", {})
        terms3x3 = form_key(terms3x3)
        if terms3x3 not in cache:
            cache[terms3x3] = MfmaCNN(self, terms3x3)
        return cache[terms3x3]

    def mfma_cnn(self, net_out):
        """The MFMA render CNN (cnn.MfmaCNN) for the current precision profile.

        The four 3x3 convolutions can run as ONE f16 product (operands rounded to nearest: a third of the MFMAs, 2.9 ms
        instead of 7.3 ms per 960x540 frame) or as the 3-term f16 split (agrees with the fp32 CNN to < 2e-5).  The 1-term form
        is LOSSY -- its error grows with the activations' magnitude, i.e. it depends on the loaded weights and the style -- so
        it is not a blind default.  Who decides (cnn_terms3x3 = None, "auto"):
          * Renderer.calibrate_style, end to end, on the style's first frame (the record `cnn_calibration` then says
            `measured: end to end`): 1-term only if its image is within CNN_AUTO_BOUND of the 3-term image AND within
            IMAGE_AUTO_BOUND of the fp32 image;
          * where no fp32 twin is at hand (modules.Backend: the drop-in binding; bands rendered without dist.agree_precision), the
            window below: every net_out presented until CNN_CAL_PIXELS pixels of the style have been seen goes through both
            forms; the 1-term image is used while every comparison stayed within CNN_AUTO_BOUND and the charged field error plus
            that difference within IMAGE_BUDGET; the first violation closes the gate for the style.
        An explicit cnn_terms3x3 (set_precision, or SDN_CNN_TERMS in the environment) bypasses the gate."""
        cache = self.__dict__.setdefault("_mfma_cnns", {})
        get = self._cnn_form

        want = getattr(self, "cnn_terms3x3", None)
        if want is None and "SDN_CNN_TERMS" in os.environ:
            want = os.environ["SDN_CNN_TERMS"]          # form_key (in _cnn_form) reads "1", "3" and per-layer forms like "1113"
        if want is not None:
            return get(want)
        cal = getattr(self, "cnn_calibration", None)
        if cal is None or (cal["terms3x3"] != 3 and cal["pixels"] < CNN_CAL_PIXELS):
            # calibration window: every net_out presented until CNN_CAL_PIXELS pixels of the style have been seen (one 960x540
            # frame; the first ~20 tiles of the reference's tiled loop) goes through the 3-term form AND every cheaper rung of
            # cnn.CNN_LADDER that has not failed yet.  The cheapest rung is used whose every comparison so far stayed inside the
            # bound AND inside the image budget left by the field's own measured error (field_gate); a rung that violates either
            # once is out for the style.
            from .cnn import CNN_LADDER
            bound = float(getattr(self, "cnn_auto_bound", None) or CNN_AUTO_BOUND)
            fg = getattr(self, "field_gate", None)
            field_err = float(fg["max_abs_err_vs_fp32"]) if fg else FIELD_NOMINAL_ERR
            worst = dict(cal["max_abs_diff_vs_3term"]) if cal else {}
            fits = lambda v: v <= bound and field_err + v <= IMAGE_BUDGET
            with torch.no_grad():
                ref3 = get(3)(net_out)
                for cand in CNN_LADDER:
                    if cand != 3 and fits(worst.get(str(cand), 0.0)):
                        worst[str(cand)] = max(worst.get(str(cand), 0.0), float((ref3 - get(cand)(net_out)).abs().max()))
            t = next((cand for cand in CNN_LADDER if cand != 3 and fits(worst.get(str(cand), float("inf")))), 3)
            px = int(net_out.shape[1] * net_out.shape[2])
            cal = self.cnn_calibration = {
                "terms3x3": t, "max_abs_diff_1term_vs_3term": worst.get("1"), "max_abs_diff_vs_3term": worst, "bound": bound,
                "ladder": [str(k) for k in CNN_LADDER],
                "field_err_charged": field_err, "image_budget": IMAGE_BUDGET, "pixels": (cal["pixels"] if cal else 0) + px,
                "calls": (cal["calls"] if cal else 0) + 1,
                "frame": f"first {(cal['calls'] if cal else 0) + 1} net_out(s) of the style, {(cal['pixels'] if cal else 0) + px} px "
                         f"(window {CNN_CAL_PIXELS} px)"}
            if t == 3 or cal["pixels"] >= CNN_CAL_PIXELS:
                self._drop_other_cnn_planes(t)
        return get(cal["terms3x3"])

    def compute_dtype(self, mode):
        if mode == "unfused":
            return "f32"
        from . import fused
        ct, _ = fused.precision_profile(self)
        cal = getattr(self, "cnn_calibration", None)
        t3 = getattr(self, "cnn_terms3x3", None) or os.environ.get("SDN_CNN_TERMS")
        if t3 is None:
            t3 = (f"{cal['terms3x3']}-term (auto: 1-term vs 3-term image differed by {cal['max_abs_diff_1term_vs_3term']:.1e} <= "
                  f"{cal['bound']:.0e} on the style's first frame)" if cal and cal["terms3x3"] == 1 else
                  f"3-term (auto: the 1-term form differed by {cal['max_abs_diff_1term_vs_3term']:.1e} > {cal['bound']:.0e})" if cal and cal["terms3x3"] == 3
                  else f"per layer (conv2a, conv2b, conv3a, conv3b) = {cal['terms3x3']} terms (auto: the cheapest rung inside the gate; "
                       f"all-1-term differed by {cal['max_abs_diff_1term_vs_3term']:.1e} > {cal['bound']:.0e})" if cal
                  else "auto (1-term if within 5e-4 of the 3-term image, not yet calibrated)")
        else:
            t3 = f"{t3}-term (set explicitly)"
        eps = fused.precision_profile(self)[1]
        return (f"f32 (hash grid) + f16 MFMA with f32 accumulate{f' (early ray termination at transmittance {eps:g})' if eps > 0 else ''}: field/sky MLP 3-term split"
                f"{' (colour layers 2-term)' if ct == 2 else ' (colour layers: f16 Whi.Xhi + MX-fp6 corrections)' if ct == 6 else ''}"
                f"{' (sky hidden layers: f16 + MX-fp6 corrections)' if fused.sky_terms(self) == 6 else ''}"
                f", render CNN 1x1 3-term / 3x3 {t3}")

    def measure_roofline(self, pose, resolution_hw, num_samples, mode, hbm_peak_gbps=8000.0, mfma_peak_tflops=2500.0):
        """Roofline records, timed with events on the launch stream (PyTorch's current stream).
        Algorithmic work per sample: SURVEY.md 8(d) -- 754 176 FLOP (render MLP), 16 404 B (grid gather, fused)
        / 16 916 B (un-fused).  Returns (dominant-kernel record, grid-sampler record)."""
        with torch.no_grad():
            vid, d2, rd, cam_res = self.cast_rays(pose, resolution_hw)
            R = cam_res[0] * cam_res[1]
            cam_ori = torch.as_tensor(pose[0], dtype=torch.float32).to(self.dev)
            if mode == "unfused":
                n = min(R, 1 << 16)
                depth, _, _ = self.place_samples(d2.view(2, R, self.M)[:, :n], num_samples)
                depth = torch.nan_to_num(depth, nan=0.0, posinf=0.0, neginf=0.0)
                wc = rd.view(R, 3)[:n, None, :] * depth[:, :, None] + cam_ori
                delim = torch.tensor([float(v) for v in self.voxel_dims], device=self.dev)
                x5 = torch.cat([wc / delim * 2 - 1, self.global_enc[:, None, :].expand(n, num_samples, 2)], dim=-1)
                x5 = ((x5 + 1) / 2).reshape(-1, 5).contiguous()
                B = x5.shape[0]
                feats = torch.empty(self.grid_L, B, 8, device=self.dev)
                dummy = torch.empty(1, device=self.dev)
                w = self.w
                ms = _time_ms(lambda: ops.grid_encode_forward(x5, w["hash_encoder.embeddings"], w["hash_encoder.offsets"],
                                                              feats, B, 5, 8, self.grid_L, self.grid_S, 16, False, dummy,
                                                              0, False))
                achieved = B * 16916 / (ms * 1e-3) / 1e9
                # 32 corner rows x 32 B x 16 levels per sample really are requested, but from a 268 MB table whose coarse levels
                # stay in L2 / Infinity Cache: the rate is an on-die gather rate, bounded by the aggregate L2 bandwidth -- not
                # by HBM (dividing it by the HBM peak gave a "fraction" above 1)
                grid = {"bound": "l2", "kernel": "grid_fwd_quad_kernel<5,8> (drop-in GridEncoder.forward, 32-corner 5-D gather)", "achieved": achieved,
                        "peak": L2_PEAK_GBPS, "unit": "GB/s", "frac": achieved / L2_PEAK_GBPS, "traffic": None,
                        "effective_over_hbm_peak": achieved / hbm_peak_gbps,
                        "samples_per_launch": B, "algorithmic_bytes_per_sample": 16916, "avg_launch_ms": ms,
                        "note": "achieved = samples x 16 916 B (SURVEY 8(d), un-fused) / launch time: L2-level gather rate; peak = "
                                "aggregate L2 bandwidth (MI355X_MICROARCH.md: 34.5 TB/s); DRAM traffic not profiled for this kernel"}
                return grid, grid
            from . import fused
            vid, d2, rd = vid.view(R, self.M), d2.view(2, R, self.M), rd.view(R, 3)
            sky_c = self.sky_features(rd)
            sky_avg = sky_c.mean(dim=0, keepdim=True)
            B, ms_enc, per_sample, kernel = fused.time_encode_kernel(self, vid, d2, rd, cam_ori, num_samples)
            _, ms_mlp, hit, ev = fused.time_mlp_kernel(self, vid, d2, rd, cam_ori, sky_c, sky_avg, num_samples)
        return self.roofline_records(B, ms_enc, ms_mlp, hit, ev, kernel, hbm_peak_gbps, mfma_peak_tflops,
                                     "HIP events around 5 back-to-back launches of each kernel on the whole padded frame, "
                                     "outside the timed region")

    def field_work(self, poses, resolution_hw, num_samples, apron="minimal"):
        """What the field kernel of the fused frame loop processes for these poses, averaged per frame (outside any timed
        region): samples per launch, fraction of rays that hit something, and the samples the kernel EVALUATES -- it visits only
        32-ray groups with a hit, and with early termination on (the default) it drops a group's remaining passes once all its
        rays are opaque: the executed passes are then counted by launching the kernel once per pose with a `passes` buffer."""
        from . import fused
        crop = self.pad // 2
        o = crop - CNN_HALO if (apron == "minimal" and crop > CNN_HALO) else 0
        nch = -(-num_samples // 4)
        eps = fused.precision_profile(self)[1]
        B = hits = groups = evald = skipped = coloured = 0.0
        with torch.no_grad():
            for pose in poses:
                vid, d2, rd, (H0, W0) = self.cast_rays(pose, resolution_hw)
                hit = (vid.view(H0, W0, self.M)[o:H0 - o, o:W0 - o, 0] != 0).reshape(-1)
                n = hit.numel()
                g = fused.Window.crop(H0, W0, o).groups(hit, ragged=True).any(dim=1)          # the 32-ray groups as the launch forms them
                B += n * num_samples
                hits += float(hit.float().mean())
                groups += float(g.float().mean())
                if (eps > 0 or fused.colour_skip(self)) and fused.single_kernel(self):
                    n0 = H0 * W0
                    v, d, r = vid.view(n0, self.M), d2.view(2, n0, self.M), rd.view(n0, 3)
                    sky_c, sky_avg = fused.sky_fused(self, r)
                    win = fused.Window.crop(H0, W0, o)
                    pa = torch.zeros(win.n_groups(ragged=True), dtype=torch.uint8, device=self.dev)
                    cp = torch.zeros_like(pa)
                    fused.field_render(self, v, d, r, torch.as_tensor(pose[0], dtype=torch.float32), sky_c, sky_avg, num_samples,
                                       passes=pa, window=win, colour_passes=cp)
                    executed = int(pa.sum(dtype=torch.int64))
                    evald += executed * 128
                    coloured += int(cp.sum(dtype=torch.int64)) * 128
                    skipped += int((pa > 0).sum()) * nch - executed
                else:
                    evald += int(g.sum()) * 32 * nch * 4
                    coloured += int(g.sum()) * 32 * nch * 4
        k = max(1, len(poses))
        return B / k, hits / k, dict(group_hit_fraction=groups / k, evaluated_samples=evald / k, passes_skipped_by_termination=skipped / k,
                                     passes_of_visited_groups=(evald / 128 + skipped) / k, colour_samples=coloured / k)

    def roofline_records(self, B, ms_enc, ms_mlp, hit, ev, kernel, hbm_peak_gbps=8000.0, mfma_peak_tflops=2500.0, timing="",
                         field_kernel=False):
        """(field-MLP record, grid-sampler record) from per-launch work (B samples, hit fraction, evaluated samples in
        `ev`) and average launch durations.  field_kernel: ms_mlp is the duration of the single-kernel field (its launches
        contain the encode stage as well: the MLP's algorithmic FLOPs are divided by the WHOLE launch time)."""
        from . import fused
        traffic, traffic_src = _profiled_traffic()
        ct, eps = fused.precision_profile(self)
        # ---- grid sampler (encode_kernel).  SURVEY 8(d): effective gather bandwidth = samples x 16 404 B / time.  The
        # gathers are served on-die (collapsed table: 8 x 32 B per level instead of 32 x 32 B, L2 / Infinity-Cache
        # hits), so that figure exceeds the HBM peak many times over: HBM does not bound this kernel, the L2-level
        # gather rate does.  All three rates are reported; `frac` is against the bound that applies (aggregate L2).
        n_gather = B * hit                                   # samples of rays that hit something: the others issue no gathers
        eff = B * 16404 / (ms_enc * 1e-3) / 1e9              # SURVEY 8(d) definition, every sample of the frame
        coll = n_gather * (4096 + 20 + 512) / (ms_enc * 1e-3) / 1e9   # bytes the kernel really moves at L2 level
        dram = traffic.get("encode_kernel")
        grid = {"bound": "l2", "kernel": kernel, "achieved": coll, "peak": L2_PEAK_GBPS, "unit": "GB/s",
                "frac": coll / L2_PEAK_GBPS, "avg_launch_ms": ms_enc, "samples_per_launch": B, "timing": timing,
                "samples_with_gathers": n_gather,
                "effective_GBps": eff, "effective_bytes_per_sample": 16404, "effective_over_hbm_peak": eff / hbm_peak_gbps,
                "collapsed_GBps": coll, "collapsed_bytes_per_sample": 4096 + 20 + 512,
                "dram_GBps_from_profile": (dram / (ms_enc * 1e-3) / 1e9) if dram else None,
                "dram_frac_of_hbm_peak_from_profile": (dram / (ms_enc * 1e-3) / 1e9 / hbm_peak_gbps) if dram else None,
                "traffic": dram, "traffic_source": traffic_src,
                "note": "effective = SURVEY 8(d): samples x 16 404 B (reference's 32-corner 5-D gather) / launch time -- "
                        "served on-die, hence far above the 8 TB/s HBM peak; collapsed = what this kernel moves at L2 level "
                        "(8 corners x 32 B x 16 levels + 20 B coords + 512 B feature write, only for rays that hit); "
                        "dram = FETCH+WRITE bytes of the PMC profile named in traffic_source / launch time (mostly the "
                        "feature write); peak = aggregate L2 bandwidth (MI355X_MICROARCH.md: 34.5 TB/s)"}
        # ---- field MLP.  Algorithmic FLOPs are counted on the samples the kernel EVALUATES (it skips 32-ray groups that
        # hit nothing and the passes early termination removes): samples of skipped groups are not work done.
        n_eval = ev["evaluated_samples"]
        # ... and the colour branch (fc_5, fc_6, fc_out_c: 294 912 of the 754 176 FLOP) only on the passes that ran it: passes whose
        # 128 samples all have volume-rendering weight exactly zero skip it (field.hip), and work not done is not counted
        n_col = ev.get("colour_samples", n_eval)
        flop_launch = n_eval * (754176 - 294912) + n_col * 294912
        ach_m = flop_launch / (ms_mlp * 1e-3) / 1e12
        # MFMA issue slots per pass / algorithmic (one f16 MFMA per product tile): 2208 for the 3-term split everywhere;
        # colour layers 2-term: 2 x 128 fewer; colour layers f16 + fp6: 2 x (384 - 192) fewer (an fp6 K = 64 MFMA takes the
        # issue time of one K = 16 f16 MFMA)
        issued = (2208 - (256 if ct == 2 else 384 if ct == 6 else 0)) / 736.0
        colour = {2: "2-term", 3: "3-term", 6: "f16 + MX-fp6 corrections"}[ct]
        name = ("field_kernel = mlp_kernel<FUSED>: sample placement + collapsed hash-grid lookup + MLP + compositing in ONE launch"
                if field_kernel else "mlp_kernel")
        mlp = {"bound": "mfma", "kernel": f"{name} (f16 MFMA, 3-term split, colour layers {colour}, f32 accumulate)",
               "achieved": ach_m, "peak": mfma_peak_tflops, "unit": "TFLOP/s", "frac": ach_m / mfma_peak_tflops,
               "traffic": traffic.get("field_kernel (mlp_kernel<0, 6, 1>)" if field_kernel else "mlp_kernel"), "traffic_source": traffic_src,
               "samples_per_launch": B, "samples_evaluated": n_eval, "algorithmic_flop_per_sample": 754176,
               "samples_with_colour_branch": n_col, "colour_branch_flop_per_sample": 294912, "algorithmic_flop_per_launch": flop_launch,
               "colour_passes_skipped_fraction": 1.0 - n_col / max(n_eval, 1.0),
               "achieved_counting_skipped_colour_branch": n_eval * 754176 / (ms_mlp * 1e-3) / 1e12,
               "frac_counting_skipped_colour_branch": n_eval * 754176 / (ms_mlp * 1e-3) / 1e12 / mfma_peak_tflops,
               "accounting": "`achieved` / `frac` count the FLOPs the launch EXECUTES; `*_counting_skipped_colour_branch` is SURVEY 8(d)'s "
                             "754 176 FLOP x every sample the launch finishes (a skipped colour branch is a finished sample: its colour is "
                             "multiplied by a weight that is exactly zero) -- the figure comparable with earlier rounds' `frac`",
               "avg_launch_ms": ms_mlp, "ray_hit_fraction": hit, "group_hit_fraction": ev["group_hit_fraction"],
               "early_termination_eps": eps, "passes_skipped_by_termination": ev["passes_skipped_by_termination"],
               "issued_over_algorithmic": issued, "issued_frac_of_peak": ach_m * issued / mfma_peak_tflops,
               "timing": timing,
               "achieved_counting_skipped_samples": B * 754176 / (ms_mlp * 1e-3) / 1e12,
               "note": ("the launch ALSO contains the encode stage of its samples (sample placement + 8-corner gathers of 16 levels, "
                        "the work of the former encode_kernel): its time is in the denominator, its bytes are not in the numerator; "
                        if field_kernel else "") +
                       "achieved = (samples evaluated x 459 264 FLOP of trunk + density head + samples whose pass ran the colour branch x "
                       "294 912 FLOP) / launch time (skipped sky groups, terminated passes and skipped colour branches are not "
                       "counted as work); the kernel issues `issued_over_algorithmic` MFMA slots per algorithmic product (hi*hi + "
                       "lo*hi + hi*lo: plain f16 misses the 1e-3 bound 17x; in the colour layers the two corrections run as "
                       "block-scaled fp6 at 4x the rate); traffic = HBM bytes per launch from the "
                       "PMC profile named in traffic_source (a separate rocprofv3 --pmc run, not this process)"}
        return mlp, grid

    # ------------------------------------------------------------------ row bands (tile-parallel single frame)
    def row_costs(self, pose, resolution_hw, scale=4):
        """Relative cost of every OUTPUT row of the frame, for cutting it into bands of equal work (dist.balanced_row_bands):
        the field kernel visits only rays that hit something (sky rows cost almost nothing there), every ray costs ray casting,
        sky MLP and CNN.  Estimated from a 1/scale-resolution ray cast of the padded frame (1/16 of the rays; deterministic and
        bit-identical on every rank, so all ranks cut the same bands without talking): cost(row) = hits(row) + MISS_COST * width.
        The result is kept per (pose, resolution): a trajectory that is rendered again -- or the stats frame of bench.py -- pays
        the low-resolution ray cast and its device -> host read once."""
        cam_ori, cam_dir, cam_up, cam_f = pose
        H, W = resolution_hw
        key = (tuple(np.asarray(cam_ori, np.float64).reshape(-1).tolist()), tuple(np.asarray(cam_dir, np.float64).reshape(-1).tolist()),
               tuple(np.asarray(cam_up, np.float64).reshape(-1).tolist()), float(cam_f), int(H), int(W), int(scale), id(self.volume))
        cache = self.__dict__.setdefault("_row_cost_cache", {})
        if key in cache:
            return cache[key]
        f, c, cam_res = frame_intrinsics(cam_f, resolution_hw, self.pad)
        Hq, Wq = -(-cam_res[0] // scale), -(-cam_res[1] // scale)
        off = (scale - 1) / 2.0
        with torch.no_grad():
            vid, _, _ = ops.ray_voxel_intersection_perspective(self.volume, cam_ori, cam_dir, cam_up, f / scale,
                                                               [(c[0] - off) / scale, (c[1] - off) / scale], [Hq, Wq], 1, palette=self.palette)
            hits_q = (vid.view(Hq, Wq) != 0).sum(dim=1).cpu().numpy().astype(np.float64) * scale      # hits per padded row, estimated
        pad_rows = np.minimum(np.arange(H) + self.pad // 2, cam_res[0] - 1)       # the padded row at the centre of output row r's apron
        while len(cache) >= 512:
            cache.pop(next(iter(cache)))
        cache[key] = hits_q[pad_rows // scale] + MISS_COST * cam_res[1]
        return cache[key]

    def band_prepare(self, pose, resolution_hw, row0, row1, mode="fused", apron="minimal"):
        """Cast the rays needed for output rows [row0,row1) and evaluate the sky MLP on them.  Returns a handle with the band's
        share of the frame-wide sky sum (sky_avg is the mean over ALL rays of the padded frame, scenedreamer.py:592-598: every
        padded row is owned by exactly one band).
        apron: the reference's tiles carry 15 px of apron per side (pad / 2); only CNN_HALO = 4 px can reach a kept pixel.
        "minimal" (fused mode): the band casts padded rows [row0 + 11, row1 + 19) -- the first / last band additionally the
        frame's top / bottom rows, which only the sky mean needs -- and evaluates the field and the CNN on its 4-px apron;
        "reference": padded rows [row0, row1 + 30), everything evaluated (the un-fused path always does)."""
        cam_ori, cam_dir, cam_up, cam_f = pose
        H, W = resolution_hw
        f, c, cam_res = frame_intrinsics(cam_f, resolution_hw, self.pad)
        Wp = cam_res[1]
        crop = self.pad // 2
        o = crop - CNN_HALO if (mode == "fused" and apron == "minimal" and crop > CNN_HALO) else 0
        # padded rows this band casts / owns for the sky sum / evaluates the field on
        p0 = 0 if row0 == 0 else row0 + o
        p1 = cam_res[0] if row1 == H else row1 + self.pad - o
        own0 = 0 if row0 == 0 else row0 + crop
        own1 = cam_res[0] if row1 == H else row1 + crop
        if o == 0:                       # reference apron: the ownership of rounds 2-3 (rows [row0, row1) + the trailing pad)
            p0, own0, own1 = row0, row0, (cam_res[0] if row1 == H else row1)
        e0, e1 = row0 + o, row1 + self.pad - o
        # same rays as the full frame: ndc_y = c0 - row_global = (c0 - p0) - row_local, exact in float32
        vid, d2, rd = ops.ray_voxel_intersection_perspective(self.volume, cam_ori, cam_dir, cam_up, f, [c[0] - p0, c[1]],
                                                             [p1 - p0, Wp], self.M, palette=self.palette)
        n = (p1 - p0) * Wp
        vid, d2, rd = vid.view(n, self.M), d2.view(2, n, self.M), rd.view(n, 3)
        with torch.no_grad():
            if mode == "fused":
                from . import fused
                sky_c, _ = fused.sky_fused(self, rd)
            else:
                sky_c = self.sky_features(rd)
            sky_sum = sky_c[(own0 - p0) * Wp:(own1 - p0) * Wp].sum(dim=0, dtype=torch.float64)
        return dict(vid=vid, d2=d2, rd=rd, sky_c=sky_c, sky_sum=sky_sum, sky_cnt=(own1 - own0) * Wp, cast_rows=(p1 - p0), Wp=Wp,
                    rows=(e1 - e0), cols=Wp - 2 * o, first=(e0 - p0) * Wp + o, halo=crop - o,
                    cam_ori=(torch.as_tensor(cam_ori, dtype=torch.float32) if mode == "fused"
                             else torch.as_tensor(cam_ori, dtype=torch.float32).to(self.dev)), mode=mode)

    def band_finish(self, hd, sky_avg, num_samples, cnn_mode=None):
        """Field + CNN for a prepared band given the frame-wide sky_avg [1,64]; returns image rows [1,3,row1-row0,W]."""
        mode = hd["mode"]
        with torch.no_grad():
            sky_avg = sky_avg.to(torch.float32).reshape(1, 64)
            full = hd["rows"] == hd["cast_rows"] and hd["cols"] == hd["Wp"]
            if mode == "fused" and self.field_falls_back():      # (the job-wide decision of dist.agree_precision)
                mode, hd["cam_ori"] = "unfused", hd["cam_ori"].to(self.dev)
            if mode == "fused":
                from . import fused
                win = None if full else fused.Window(hd["cast_rows"] * hd["Wp"], hd["Wp"], hd["first"], hd["rows"], hd["cols"])
                net_out = fused.field_fused(self, hd["vid"], hd["d2"], hd["rd"], hd["cam_ori"], hd["sky_c"], sky_avg,
                                            num_samples, window=win)
            else:
                vid, d2, rd, sky_c = hd["vid"], hd["d2"], hd["rd"], hd["sky_c"]
                if not full:             # (a fused band that fell back: cut the evaluated window out of the cast block)
                    y0, x0 = divmod(hd["first"], hd["Wp"])
                    cut = lambda t: t.view(hd["cast_rows"], hd["Wp"], -1)[y0:y0 + hd["rows"], x0:x0 + hd["cols"]].reshape(hd["rows"] * hd["cols"], -1)
                    vid, rd, sky_c = cut(vid), cut(rd), cut(sky_c)
                    d2 = torch.stack([cut(d2[0]), cut(d2[1])])
                net_out = self.field_unfused(vid.contiguous(), d2.contiguous(), rd.contiguous(), hd["cam_ori"], sky_c.contiguous(), sky_avg,
                                             num_samples)
            net_out = net_out.view(1, hd["rows"], hd["cols"], 64)
            if cnn_mode is None:
                cnn_mode = "mfma" if mode == "fused" else "torch"
            if cnn_mode == "mfma":
                img = self.mfma_cnn(net_out)(net_out)
            else:
                img = self.render_cnn(net_out)
            p = hd["halo"]
            return img[:, :, p:-p, p:-p] if p else img

    # ------------------------------------------------------------------ frame
    def render_frame(self, pose, resolution_hw=(540, 960), num_samples=24, mode="unfused", cnn=True,
                     ray_chunk=1 << 16, timers=None, cnn_mode=None, apron="minimal", _precast=None):
        """One frame of the trajectory.  Returns image [1,3,H,W] (or net_out [1,Hp,Wp,64] if cnn=False).

        apron: the reference evaluates every ray of the frame padded by 15 px per side (its tile scheme,
        scenedreamer.py:573-628) and crops the image afterwards.  Only CNN_HALO = 4 px of that apron can reach a kept
        pixel (RenderCNN has four 3x3 convolutions, gancraft_base.py:175-225; their zero padding at the padded frame's
        border is 15 px away).  "minimal" (fused path, default) evaluates the field and the CNN on the 4-px apron; the
        sky MLP still sees every ray of the padded frame, because its frame mean does (scenedreamer.py:592-598).  The
        image is bit-identical to "reference" (full apron) when every sample is evaluated (term_eps = 0); with early ray
        termination (the default) the 32-ray groups that stop together differ between the two windows, and the images agree
        to the termination bound (each net_out within 2 eps = 1e-4 of the untruncated one) -- tests/test_render_gpu.py."""
        ev = _Stamps(timers)
        with torch.no_grad():
            ev.mark("start")
            if _precast is None:
                vid, d2, rd, cam_res = self.cast_rays(pose, resolution_hw)
            else:
                vid, d2, rd, cam_res = _precast
            ev.mark("rvip")
            Hp, Wp = cam_res
            R = Hp * Wp
            vid = vid.view(R, self.M)
            d2 = d2.view(2, R, self.M)
            rd = rd.view(R, 3)
            # host value for the fused path (its C entry points take host floats): a device copy here and the .cpu() that
            # would undo it are two host<->device synchronisations per frame, each draining the launch queue
            cam_ori = torch.as_tensor(pose[0], dtype=torch.float32)
            if mode != "fused":
                cam_ori = cam_ori.to(self.dev)
            if mode == "fused":
                # per-style precision gates (once per style; a host synchronisation on the style's first frame) -- before the sky
                # MLP of this frame, whose hidden-layer form is one of the decisions
                if getattr(self, "field_gate", None) is None and FIELD_GATE:
                    self.calibrate_style(pose, resolution_hw, num_samples)
                if self.field_falls_back():      # this style / these weights are outside the fused path's tolerance: the fp32 op
                    mode, cam_ori = "unfused", cam_ori.to(self.dev)       # sequence, all of it (sky MLP and CNN included)
                    if cnn_mode is None:
                        cnn_mode = "torch"
            if mode == "fused":
                from . import fused
                sky_c, sky_avg = fused.sky_fused(self, rd)
            else:
                sky_c = self.sky_features(rd)
                sky_avg = sky_c.mean(dim=0, keepdim=True)    # full-frame mean, scenedreamer.py:592-598
            ev.mark("sky")
            crop = self.pad // 2
            window = None
            if mode == "fused" and cnn and apron == "minimal" and crop > CNN_HALO:
                # rows / columns of the padded frame that cannot influence the cropped image are not evaluated: the field
                # kernels read the frame-wide ray arrays through a window (no strided-slice copies)
                from . import fused
                o = crop - CNN_HALO
                window = fused.Window.crop(Hp, Wp, o)
                Hp, Wp, crop = Hp - 2 * o, Wp - 2 * o, CNN_HALO
            elif mode == "fused":       # the whole padded frame, as a window too: its launch takes the 8 x 4-pixel ray blocks
                from . import fused
                window = fused.Window.crop(Hp, Wp, 0)
            if mode == "unfused":
                outs = []
                for r0 in range(0, R, ray_chunk):
                    r1 = min(r0 + ray_chunk, R)
                    outs.append(self.field_unfused(vid[r0:r1], d2[:, r0:r1], rd[r0:r1], cam_ori, sky_c[r0:r1],
                                                   sky_avg, num_samples))
                net_out = torch.cat(outs, dim=0)
            elif mode == "fused":
                from . import fused
                net_out = fused.field_fused(self, vid, d2, rd, cam_ori, sky_c, sky_avg, num_samples, window=window)
            else:
                raise ValueError(mode)
            net_out = net_out.view(1, Hp, Wp, 64)
            ev.mark("field")
            if not cnn:
                ev.done()
                return net_out
            if cnn_mode is None:
                cnn_mode = "mfma" if mode == "fused" else "torch"
            if cnn_mode == "mfma":
                img = self.mfma_cnn(net_out)(net_out)
            else:
                img = self.render_cnn(net_out)
            if crop:
                img = img[:, :, crop:-crop, crop:-crop]
            ev.mark("cnn")
            ev.done()
            return img


def _render_frames(self, poses, resolution_hw=(540, 960), num_samples=24, mode="fused", apron="minimal", probe=None, **kw):
    """Generator over the frames of a trajectory, software-pipelined over two streams: the front half of frame i+1
    (ray casting, sky MLP, sample encode) is issued on a second stream while the back half of frame i (field MLP, render
    CNN) runs.  rvip_kernel (20 registers, no LDS) co-resides with the one-workgroup-per-CU MFMA kernels; the sky and
    encode kernels fill the CUs that idle at the tails and launch boundaries of the MFMA kernels.  Images are bit-identical
    to render_frame (tests/test_fullsize_gpu.py).
    probe: optional dict; gets lists of (start, end) timing events around the dominant kernels' launches, recorded on the
    stream they are launched on ("mlp_kernel": main stream, "encode_kernel": side stream) -- bench.py's roofline record is
    computed from the launches of the timed region itself."""
    from . import fused
    poses = list(poses)
    if not poses:
        return
    main = torch.cuda.current_stream(self.dev)
    side = getattr(self, "_side_stream", None)
    if side is None:
        side = self._side_stream = torch.cuda.Stream(self.dev)
    if mode == "fused":
        if getattr(self, "field_gate", None) is None and FIELD_GATE:
            self.calibrate_style(poses[0], resolution_hw, num_samples, more_poses=poses[len(poses) // 2:len(poses) // 2 + 1] if len(poses) > 2 else ())
        if self.field_falls_back():
            mode = "unfused"
    f0, c0, cam_res = frame_intrinsics(poses[0][3], resolution_hw, self.pad)
    crop = self.pad // 2
    o = crop - CNN_HALO if (apron == "minimal" and crop > CNN_HALO) else 0
    Hp, Wp = cam_res[0] - 2 * o, cam_res[1] - 2 * o
    one = mode == "fused" and fused.single_kernel(self) and fused.precision_profile(self)[0] != 2   # lookup + MLP in ONE kernel
    deep = mode == "fused" and not kw and (one or fused.single_chunk(Hp * Wp, num_samples))

    def front(pose, slot):
        start = torch.cuda.Event()
        start.record(main)                      # everything of the frame that used this slot before has been enqueued
        with torch.cuda.stream(side), torch.no_grad():
            side.wait_event(start)
            vid, d2, rd, res = self.cast_rays(pose, resolution_hw)
            out = (vid, d2, rd, res)
            if deep:
                H0, W0 = res
                n0 = H0 * W0
                vid, d2, rd = vid.view(n0, self.M), d2.view(2, n0, self.M), rd.view(n0, 3)
                sky_c, sky_avg = fused.sky_fused(self, rd)
                win = fused.Window.crop(H0, W0, o)      # the kernels read the frame-wide arrays through the window
                if one:     # the field kernel gathers for itself: the front half is ray casting + sky MLP only
                    out = ((vid, d2, rd), sky_c, sky_avg, win)
                    keep = (vid, d2, rd, sky_c, sky_avg)
                else:
                    if probe is not None:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(side)
                    buf = fused.encode(self, vid, d2, rd, torch.as_tensor(pose[0], dtype=torch.float32), num_samples,
                                       fused._buffers(self, win.n_rays, num_samples, slot), window=win)
                    if probe is not None:
                        e1.record(side)
                        probe.setdefault("encode_kernel", []).append((e0, e1))
                    out = (buf, sky_c, sky_avg, win)
                    keep = (sky_c, sky_avg)     # vid / d2 / rd are only read on the side stream (by encode)
            else:
                keep = out[:3]
            done = torch.cuda.Event()
            done.record(side)
        for t in keep:
            t.record_stream(main)               # allocated on the side stream, consumed on the main stream
        return out, done

    # (Issuing the next front half only behind this frame's MLP was measured: mlp_kernel 16.2 -> 15.8 ms without the ray caster
    # beside its start, but the frame 22.4 -> 22.8 ms, because the ray caster then lands in the CNN phase too.)
    # where the next frame's front half (ray casting + sky MLP [+ encode]) is released: "early" = as soon as the previous
    # frame's CNN has been enqueued, i.e. beside this frame's field kernel; "late" = behind this frame's field kernel, i.e. beside
    # its CNN.  (SDN_FRONT=late|early; measured in DESIGN.md section 6.)
    late = deep and os.environ.get("SDN_FRONT", FRONT_DEFAULT) == "late"
    # The render CNN of frame i on a THIRD stream, beside the field kernel of frame i+1 (SDN_CNN_STREAM=1): both are one-workgroup-
    # per-CU kernels, so they cannot share a CU, but the CNN's six dependent launches leave CUs idle at every launch boundary
    # and the field kernel's persistent workgroups retire over the length of a 32-ray group -- with both in flight whichever has
    # workgroups ready takes the idle CUs.  The image of frame i is handed out one iteration later (after the field kernel of
    # frame i+1 has been enqueued), so the consumer's wait for it does not order the main stream behind the CNN.
    cnn_side = deep and os.environ.get("SDN_CNN_STREAM", CNN_STREAM_DEFAULT) == "1" and getattr(self, "field_gate", None) is not None
    cstream = None
    if cnn_side:
        cstream = getattr(self, "_cnn_stream", None)
        if cstream is None:
            cstream = self._cnn_stream = torch.cuda.Stream(self.dev)
    pending = None          # (image, its completion event) of the previous frame
    nxt = front(poses[0], 0)
    try:
        for i, pose in enumerate(poses):
            cur, done = nxt
            if not late:
                nxt = front(poses[i + 1], (i + 1) & 1) if i + 1 < len(poses) else None
            main.wait_event(done)
            if not deep:
                yield self.render_frame(pose, resolution_hw, num_samples, mode=mode, apron=apron, _precast=cur, **kw)
                continue
            buf, sky_c, sky_avg, win = cur
            with torch.no_grad():
                if probe is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(main)
                if one:
                    net_out = fused.field_render(self, *buf, torch.as_tensor(pose[0], dtype=torch.float32), sky_c, sky_avg, num_samples,
                                                 window=win).view(1, Hp, Wp, 64)
                else:
                    net_out = fused.mlp_from(self, buf, sky_c, sky_avg.reshape(-1), win.n_rays, num_samples, window=win).view(1, Hp, Wp, 64)
                if probe is not None:
                    e1.record(main)
                    probe.setdefault("mlp_kernel", []).append((e0, e1))
                if late:
                    nxt = front(poses[i + 1], (i + 1) & 1) if i + 1 < len(poses) else None
                c = crop - o
                if cnn_side:
                    f_done = torch.cuda.Event()
                    f_done.record(main)
                    net_out.record_stream(cstream)              # allocated on the main stream, read on the CNN stream
                    with torch.cuda.stream(cstream):
                        cstream.wait_event(f_done)
                        cnn = self.mfma_cnn(net_out)            # (decided by calibrate_style: no calibration launches here)
                        if probe is not None:
                            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            c0.record(cstream)
                        img = cnn(net_out)
                        if probe is not None:
                            c1.record(cstream)
                            probe.setdefault("render_cnn", []).append((c0, c1))
                        c_done = torch.cuda.Event()
                        c_done.record(cstream)
                    img.record_stream(main)                     # allocated on the CNN stream, consumed on the main stream
                    if pending is not None:
                        main.wait_event(pending[1])
                        yield pending[0]
                    pending = (img[:, :, c:-c, c:-c] if c else img, c_done)
                    continue
                cnn = self.mfma_cnn(net_out)         # (first frame of a style: calibrates the 3x3 precision, see mfma_cnn)
                if probe is not None:
                    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    c0.record(main)
                img = cnn(net_out)
                if probe is not None:
                    c1.record(main)
                    probe.setdefault("render_cnn", []).append((c0, c1))
                if i == len(poses) - 1 and len(poses) >= RECHECK_MIN_FRAMES:
                    self.recheck_cnn(net_out)        # (once per style: the adopted 3x3 rung on a pose the calibration did not see)
                yield img[:, :, c:-c, c:-c] if c else img
        if pending is not None:
            last, pending = pending, None
            main.wait_event(last[1])
            yield last[0]
    finally:
        if pending is not None:      # the consumer stopped early: the main stream still has to be ordered behind the CNN stream's work
            main.wait_event(pending[1])


Renderer.render_frames = _render_frames

RECHECK_MIN_FRAMES = 2     # trajectories at least this long re-check the adopted CNN rung on their last frame (Renderer.recheck_cnn)
FRONT_DEFAULT = "early"
CNN_STREAM_DEFAULT = "0"   # render CNN of frame i on its own stream beside the field kernel of frame i+1 (see _render_frames)
CNN_AUTO_BOUND = 5e-4   # mfma_cnn: largest image difference (max abs) at which the 1-term 3x3 convolutions are accepted
CNN_CAL_PIXELS = 400_000   # ... measured on every net_out of a style until this many pixels have been compared
IMAGE_BUDGET = 8e-4        # ... and only while (field error charged) + (that difference) stays below this (north star: 1e-3)
FIELD_NOMINAL_ERR = 2e-4   # field error charged to the budget when no field_gate was measured (goldens: 1.0 - 1.6e-4)
# calibrate_style (the renderer's end-to-end gates; the north star's tolerance is 1e-3 abs on radiance and on the image):
COLOUR_AUTO_BOUND = 1e-4   # largest net_out difference fp6-corrected vs 3-term colour layers (goldens: 4e-5)
FIELD_AUTO_BOUND = 1e-3    # largest net_out error of the fused field vs the fp32 op sequence, whole frame: the north star's radiance
                           # tolerance itself.  Measured on the synthetic weights (tools/dbg_field_err.py): max over the 36 M values of
                           # a 960x540 frame 5 - 6e-5 (rms 4e-6) without early termination, 9e-5 with the default term_eps -- since the
                           # trunk weights are packed times 2^8 (field.hip TRUNK_SHIFT; before that 5.6 - 8.2e-4, profiles/
                           # r04_gate_survey.jsonl: the lo halves of the split sat in f16's subnormal range, ~20 significant bits, and
                           # the density head sums ~2e3 x its result in cancelling terms).  The kernel's sigma is now as close to an
                           # fp64 evaluation as PyTorch's fp32 one is (1e-4 both).
IMAGE_AUTO_BOUND = 8e-4    # largest image error of the whole fused path vs the fp32 path, whole frame
SKY_AUTO_BOUND = 2e-4      # largest sky_c error (vs PyTorch fp32) at which the sky MLP's hidden layers run as f16 + fp6 corrections
CAL_MAX_PIXELS = 1 << 22   # frames above this many pixels (1920x1080 is below: calibrated at its own resolution) are calibrated at a reduced resolution (same pose)
CAL_CHUNK = 1 << 16        # rays per launch group of the fp32 field
CAL_CROP = 256             # calibrate_one: side of the window (output pixels) the fp32 twin and the candidates are evaluated on (0: whole frame)
CAL_CROP_FACTOR = 1.15     # ... and what a maximum measured on that window is multiplied by before it meets a bound
MISS_COST = 0.2            # row_costs: cost of a ray that hits nothing relative to one that does (ray casting + sky MLP + CNN vs + field)
FIELD_GATE = os.environ.get("SDN_FIELD_GATE", "1") != "0"   # (0: no field calibration -- kernel timing experiments only)
CNN_HALO = 4   # receptive-field radius of RenderCNN: four 3x3 convolutions (conv2a, conv2b, conv3a, conv3b)


L2_PEAK_GBPS = 34500.0   # MI355X_MICROARCH.md: 4 MiB per XCD, ~34.5 TB/s aggregate
PMC_PROFILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")   # newest first


def _profiled_traffic():
    """(HBM bytes per launch by kernel, source label) from the newest committed PMC profile: bench.py cannot run rocprofv3 on
    itself, so `traffic` in the roofline records is NOT measured in the bench process -- the label says so.  A profile is
    used only if it was taken on THESE kernel sources: tools/pmc_traffic.py stores the digest of csrc/*.hip + the header
    (build._digest(), the same value as lib/libsdnative.stamp) and a profile whose digest differs from the current
    sources' -- or that predates the digest field -- yields no traffic figure and a label that says why."""
    import json
    from . import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cur = build._digest()
    why = None
    for name in PMC_PROFILES:
        try:
            with open(os.path.join(root, "profiles", name)) as f:
                d = json.load(f)
            per = {k: v["traffic"] for k, v in d["per_launch_bytes"].items()}
        except (OSError, KeyError, ValueError):
            continue
        if d.get("csrc_digest") != cur:
            why = why or (f"profiles/{name} is STALE (taken at build {d.get('commit', 'unrecorded')}, kernel-source digest "
                          f"{str(d.get('csrc_digest'))[:12]} != current {cur[:12]}): no traffic figure reported")
            continue
        return (per, f"profiles/{name} (rocprofv3 --pmc passes of tools/frame_once.py, build {d.get('commit', 'unrecorded')}, "
                     f"kernel-source digest {cur[:12]} = this build; not measured in this run)")
    return {}, why


def _busiest_window(g, Hc, Wc, stride=8):
    """(row, column) of the Hc x Wc window of the score map g [H, W] with the largest sum, on a grid of `stride` pixels: box sums from
    a summed-area table (a pooling kernel with a 286 x 286 window took 34 ms of a 75-ms calibration; this takes 0.3)."""
    H, W = g.shape
    sat = F.pad(g.double().cumsum(0).cumsum(1), (1, 0, 1, 0))
    ys = torch.arange(0, H - Hc + 1, stride, device=g.device)
    xs = torch.arange(0, W - Wc + 1, stride, device=g.device)
    box = sat[ys + Hc][:, xs + Wc] - sat[ys][:, xs + Wc] - sat[ys + Hc][:, xs] + sat[ys][:, xs]
    k = int(box.argmax())
    return int(ys[k // xs.numel()]), int(xs[k % xs.numel()])


def _time_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs]))


class _Stamps:
    """Optional per-stage GPU timing with events on the current stream."""

    def __init__(self, sink):
        self.sink = sink
        self.ev = []

    def mark(self, name):
        if self.sink is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.ev.append((name, e))

    def done(self):
        if self.sink is not None and self.ev:
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(self.ev[:-1], self.ev[1:]):
                self.sink.setdefault(n1, []).append(e0.elapsed_time(e1))
