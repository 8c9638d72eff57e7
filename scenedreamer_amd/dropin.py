"""The fused fast path behind the reference generator's OWN surface.

`imaginaire.generators.scenedreamer.Generator.inference_givenstyle` (scenedreamer.py:479-632) calls, per frame and per tile,

    voxlib.ray_voxel_intersection_perspective   -> shims/voxlib.py            (sdn_rvip)
    voxlib.positional_encoding + self.sky_net   -> modules.SKYMLPNative       (sdn_sky_mlp)
    self._forward_perpix(...)                   -> fast_forward_perpix below  (sdn_field_render: sample placement + hash
                                                   grid + LightningMLP + volume rendering + sky compositing in ONE kernel)
    self._forward_global(net_out, z)            -> fast_forward_global below  (cnn.MfmaCNN: the MFMA render CNN)

With `scenedreamer_amd.install_shims(fast=True)` these bindings are put in place FROM OUTSIDE while the unmodified
`imaginaire` package is imported (a sys.meta_path hook that post-processes three modules; the reference's files stay
byte-identical), or -- for a generator that already exists -- by `accelerate(G)`.

What the bound methods promise: for every call the reference's own method would serve, the same `net_out` / images within
the float tolerance of the native kernels (1e-3 abs, tests/test_dropin_gpu.py).  Calls the fused kernel does not implement
(autograd, batch > 1, box-boundary sampling, a view-direction input, ...) are handed to the reference's own method (kept as
`_forward_perpix_reference` / `_forward_global_reference`), which then runs on the module-level drop-ins (modules.py) and
the HIP ops -- never on a CPU path.  `binding(G).stats` counts which way every call went and why.

Tiles: inference_givenstyle evaluates a frame in overlapping tiles (its default: 40 tiles of 158 x 158 rays for 960x540).
The per-pixel field has no coupling between rays and the CNN's receptive radius (4 px) is below the 15 px the loop crops off
every inner tile edge, so the binding evaluates the WHOLE frame once -- one field launch, one CNN pass -- when the first tile
of a frame arrives, and serves the frame's tiles as views of that result (GeneratorBinding.frame_field / frame_image_tile;
`binding(G).coalesce = False` evaluates tile by tile, in place, instead).  A frame is recognised through object identities of
the sky pre-pass / camera tensors and the version counters of the frame arrays, never through bare addresses.

_forward_perpix returns the reference's 12-tuple.  `net_out` is always there.  With `aux=False` (default: the inference
loop and Generator.forward only read `net_out`) the other eleven are None; with `binding(G, aux=True)` ALL of them are
produced (the field kernel's MODE_FUSED_AUX instantiation writes weights, rand_depth, net_out_s, net_out_c, the blended
skynet_out_c and nosky_mask; new_dists / new_idx come from sdn_sample_depth, the sky masks from voxel_id) -- what
inference_givenstyle_depth reads (scenedreamer.py:812-817); tile by tile then, without the one-evaluation-per-frame shortcut.
"""
import importlib
import importlib.abc
import importlib.util
import sys
import types

import numpy as np
import torch

from . import fused, modules, ops
from .renderer import fold_denoiser, fold_render_net

PERPIX_OUTPUTS = ("net_out", "new_dists", "weights", "total_weights_raw", "rand_depth", "net_out_s", "net_out_c", "skynet_out_c",
                  "nosky_mask", "sky_mask", "sky_only_mask", "new_idx")


class GeneratorBinding:
    def __init__(self, aux=False):
        self.B = modules.Backend()
        self.aux = bool(aux)
        self.stats = {"perpix_fast": 0, "perpix_reference": 0, "global_fast": 0, "global_reference": 0, "tiles_in_place": 0,
                      "tiles_copied": 0, "sky_reused": 0, "sky_evaluated": 0, "frames_coalesced": 0, "tiles_from_frame": 0,
                      "cnn_tiles_from_frame": 0, "why": {}}
        self._scene_key = None
        self._frame = None          # the frame evaluated once for all of its tiles (frame_field)
        self.coalesce = True
        self._refused = None        # (style key, message): prepare_style refused this style (TrunkRangeError) -- not retried per tile
        self.on_frame = None        # optional callback(frame record) when a coalesced frame's image has been produced (tests / bench:
                                    # record["img"] is the float image [1,3,H0,W0] of the padded frame the loop's tiles are views of)

    def release(self, G=None):
        """Drop the frame-sized device buffers this binding pins between calls (net_out / image of the last coalesced frame
        and, with `G`, the sky features its sky_net keeps for the frame's tiles): ~150 MB each at 960x540.  They are
        replaced, never accumulated, while rendering; call this when a generator stays alive after its last frame."""
        self._frame = None
        if G is not None:
            G.sky_net.__dict__.pop("_sdn_last_frame", None)

    # ------------------------------------------------------------------ what the fused kernel implements
    def why_not_perpix(self, G, voxel_id, depth2, raydirs, cam_ori_t, z, global_enc):
        if not (isinstance(voxel_id, torch.Tensor) and voxel_id.is_cuda and voxel_id.dtype == torch.int32):
            return "voxel_id is not a CUDA int32 tensor"
        if not modules._cuda_f32(depth2, raydirs, cam_ori_t, z, global_enc):
            return "inputs are not CUDA float32 tensors"
        if modules._wants_grad(G.render_net, z, global_enc) or modules._wants_grad(G.hash_encoder):
            return "autograd requested (the kernels are forward-only)"
        if voxel_id.dim() != 5 or voxel_id.shape[0] != 1 or z.shape[0] != 1 or voxel_id.shape[-1] != 1:
            return "batch size > 1"
        if not (1 <= voxel_id.shape[3] <= 8):
            return "more than 8 intersections per ray"
        if G.sample_use_box_boundaries:
            return "sample_use_box_boundaries"
        if not (0 < G.num_samples <= 78):
            return "more than 78 samples per ray"
        if G.raw_noise_std > 0:
            return "raw_noise_std > 0"
        if not (G.keep_sky_out and G.keep_sky_out_avgpool and (hasattr(G, "sky_avg") or G.sky_global_avgpool)):
            return "sky blending other than keep_sky_out_avgpool with a global average"
        if G.clip_feat_map is not True:
            return "clip_feat_map is not True"
        if not (G.pe_params[2] == 0 and G.pe_params[3] is False):
            return "view-direction input to the render MLP"
        if list(G.pe_params_sky) != [5, True]:
            return "sky positional encoding other than (5, incl_orig)"
        he = G.hash_encoder
        if not (getattr(he, "input_dim", 0) == 5 and getattr(he, "level_dim", 0) == 8 and getattr(he, "num_levels", 0) == 16 and
                getattr(he, "gridtype", "") == "hash" and not getattr(he, "align_corners", True) and
                he.embeddings.dtype == torch.float32):
            return "hash grid other than 5-D / 16 levels / 8 channels / hashed"
        why = _render_net_reason(G.render_net)
        if why:
            return "render_net: " + why
        if tuple(G.sky_net.fc1.weight.shape) != (256, 33) or tuple(G.sky_net.fc_out_c.weight.shape) != (64, 256):
            return "sky_net sizes"
        return None

    # ------------------------------------------------------------------ per-call state
    def sync(self, G, z, global_enc):
        """Bring the backend in line with the generator's live state; everything is cached on identities / version counters,
        so a steady-state call costs a few tuple comparisons and no device synchronisation."""
        B = self.B
        B.bind("render_net.", G.render_net)
        B.bind("sky_net.", G.sky_net)
        grid_changed = B.bind("hash_encoder.", G.hash_encoder)
        B.M = int(G.num_blocks_early_stop)
        B.sample_depth, B.dists_scale = float(G.sample_depth), float(G.dists_scale)
        he = G.hash_encoder
        B.grid_L = int(he.num_levels)
        B.grid_S = float(np.log2(he.per_level_scale))
        vt = G.voxel.voxel_t
        skey = (vt.data_ptr(), vt._version, tuple(vt.shape), global_enc.data_ptr(), global_enc._version)
        if grid_changed or skey != self._scene_key:
            lt = G.label_trans
            lut = lt.mcid2rdid_lut.clone()
            lut[lut == lt.ignore_id] = lt.dirt_id                    # mc2reduced(ign2dirt=True), mc_utils.py:241-246
            B.lut = lut.to(B.dev)
            B.voxel_dims = tuple(int(v) for v in vt.shape)
            B.max_block_id = int(vt.max()) if vt.numel() else 0      # (one synchronisation per scene)
            B.global_enc = global_enc.detach().reshape(1, 2)
            B._fused_scene = None
            self._scene_key = skey
        B.style("render_net.", z, 0, fold_render_net)
        B.style("sky_net.", z, 0, modules.fold_sky_net)

    def style_refusal(self):
        """None, or why the packed weight stream cannot hold the current style (fused.TrunkRangeError: trunk weights beyond
        f16's range once scaled).  Checked after sync(); one device -> host read per STYLE (prepare_style), then cached."""
        B = self.B
        key = (B._zkey.get("render_net."), B._bound["render_net."][1])
        if self._refused is not None and self._refused[0] == key:
            return self._refused[1]
        self._refused = None
        if B._fused_style is None:
            try:
                fused.prepare_style(B)
            except fused.TrunkRangeError as e:
                self._refused = (key, "trunk weights outside the packed f16 range")
                self._refused_detail = str(e)
                return self._refused[1]
        return None

    # ------------------------------------------------------------------ rays of the call
    @staticmethod
    def frame_window(voxel_id, depth2, raydirs):
        """If the three arguments are views of contiguous frame-wide arrays voxel_id_all [1,H0,W0,M,1], depth2_all
        [1,2,H0,W0,M,1], raydirs_all [1,H0,W0,1,3] cut to the same tile -- how inference_givenstyle produces them,
        scenedreamer.py:600-616 -- return (Window, base addresses): the kernel then reads the frame-wide arrays in place."""
        _, h, w, M, _ = voxel_id.shape
        if tuple(depth2.shape) != (1, 2, h, w, M, 1) or tuple(raydirs.shape) != (1, h, w, 1, 3) or h == 0 or w == 0:
            return None
        sv, sd, sr = voxel_id.stride(), depth2.stride(), raydirs.stride()
        # (strides of size-1 dimensions are arbitrary: a constraint is only read off a dimension that has more than one entry)
        if (M > 1 and (sv[3] != 1 or sd[4] != 1)) or sr[4] != 1:
            return None
        if w > 1 and (sv[2] != M or sd[3] != M or sr[2] != 3):
            return None
        if sd[1] % M:
            return None
        n_src = sd[1] // M                                   # rays per plane of depth2_all = rays of the frame
        if h > 1:
            if sv[1] % M or sd[2] != sv[1] or sr[1] * M != sv[1] * 3:
                return None
            pitch = sv[1] // M
        else:
            pitch = w                                        # a one-row tile does not show the frame's row pitch: rays() looks it up
        ov, od, orr = voxel_id.storage_offset(), depth2.storage_offset(), raydirs.storage_offset()
        if ov % M or od != ov or orr * M != ov * 3:          # the same tile of all three frames
            return None
        first = ov // M
        if pitch < w or first + (h - 1) * pitch + w > n_src:
            return None
        # the frame-wide arrays must lie inside the storages the views belong to
        if (voxel_id.untyped_storage().nbytes() < n_src * M * 4 or depth2.untyped_storage().nbytes() < 2 * n_src * M * 4 or
                raydirs.untyped_storage().nbytes() < n_src * 3 * 4):
            return None
        bases = (voxel_id.data_ptr() - ov * 4, depth2.data_ptr() - od * 4, raydirs.data_ptr() - orr * 4)
        return fused.Window(n_src, pitch, first, h, w), bases

    def rays(self, G, voxel_id, depth2, raydirs):
        """(window, voxel_id, depth2, raydirs, sky_c) for sdn_field_render: frame-wide arrays read in place through a window when
        the sky features of the whole frame are at hand (SKYMLPNative keeps those of the pre-pass), else tile-local copies."""
        B = self.B
        _, h, w, M, _ = voxel_id.shape
        last = G.sky_net.__dict__.get("_sdn_last_frame") if modules.is_native(G.sky_net) else None
        fw = self.frame_window(voxel_id, depth2, raydirs) if last is not None else None
        if fw is not None:
            win, (pv, pd, pr) = fw
            if (last["rd_ptr"] == pr and last["n_rays"] == win.n_src and last["rd_version"] == raydirs._version and
                    last["zkey"] is not None and last["zkey"] == B._zkey.get("sky_net.") and
                    last["wkey"] == modules.Backend.tensors_key(G.sky_net)):
                if h == 1:
                    # the frame's row pitch is read off the pre-pass's ray directions ([1,H0,W0,1,3], scenedreamer.py:592-598);
                    # unknown -> the tile is evaluated in place but never treated as a window of a coalesced frame
                    shp = tuple(last["rd_ref"].shape)
                    W0 = shp[2] if len(shp) == 5 and shp[0] == 1 and shp[1] * shp[2] == win.n_src else 0
                    win.pitch_known = W0 >= w and (win.first % W0) + w <= W0 if W0 else False
                    if win.pitch_known:
                        win.pitch = W0
                self.stats["tiles_in_place"] += 1
                self.stats["sky_reused"] += 1
                return win, pv, pd, pr, last["sky_c"], last
        n = h * w
        vid = voxel_id.reshape(n, M).contiguous()
        d2 = depth2.reshape(2, n, M).contiguous()
        rd = raydirs.reshape(n, 3).contiguous()
        sky_c, sky_mean = fused.sky_fused(B, rd)
        self.stats["tiles_copied"] += 1
        self.stats["sky_evaluated"] += 1
        return fused.Window(n), vid, d2, rd, sky_c, sky_mean

    # ------------------------------------------------------------------ one field / CNN evaluation per FRAME
    def frame_field(self, G, last, win, bases, views, cam_ori_t, sky_avg, ns):
        """net_out [1,H0,W0,64] of the WHOLE frame the tile belongs to, evaluated once and kept for the frame's other tiles.

        inference_givenstyle cuts the padded frame into overlapping tiles (128 px + a 30-px apron, scenedreamer.py:600-616) and
        calls _forward_perpix per tile: 828 000 tile-rays for the 564 300 rays of a 960x540 frame, in 40 launches that each fill
        the GPU for a few rounds.  _forward_perpix is a per-ray function of the frame-wide arrays (no coupling between rays), so
        a tile's result IS the tile-shaped window of the frame's result: the first tile of a frame evaluates all rays in one
        launch, the others are views of it.  What identifies "the same frame" are object identities held here (the sky
        pre-pass record, the camera tensor) plus the version counters of the three frame arrays -- never bare addresses, which
        the caching allocator reuses from frame to frame."""
        fr = self._frame
        key = (bases, tuple(v._version for v in views), cam_ori_t._version, ns, self.B._zkey.get("render_net."), self._scene_key,
               self.B._bound["render_net."][1], float(self.B.sample_depth), float(self.B.dists_scale), int(self.B.M))
        if (fr is not None and fr["last"] is last and fr["cam"] is cam_ori_t and fr["sky_avg"] is sky_avg and fr["key"] == key):
            return fr["net_out"]
        H0, W0 = win.n_src // win.pitch, win.pitch
        net_out = fused.field_render(self.B, bases[0], bases[1], bases[2], cam_ori_t, last["sky_c"], sky_avg, ns,
                                     window=fused.Window.crop(H0, W0, 0))     # the whole frame as a window: 8 x 4-pixel ray blocks (ragged: 990 columns)
        self._frame = dict(last=last, cam=cam_ori_t, sky_avg=sky_avg, key=key, net_out=net_out.view(1, H0, W0, 64), img=None, raw=None,
                           keep=views)
        self.stats["frames_coalesced"] += 1
        return self._frame["net_out"]

    def frame_image_tile(self, G, net_out, z):
        """If `net_out` is a tile-shaped view of the frame evaluated by frame_field, the same window of the frame's image
        (MfmaCNN on the whole frame, once): RenderCNN's receptive radius is 4 px and the reference crops 15 px off every tile
        edge that is not a frame edge (scenedreamer.py:623-624), so every pixel it keeps is the pixel of the frame-wide
        evaluation.  None if net_out is something else."""
        fr = self._frame
        if fr is None or net_out.dim() != 4:
            return None
        full = fr["net_out"]
        _, H0, W0, _ = full.shape
        _, h, w, c = net_out.shape
        if (net_out.untyped_storage().data_ptr() != full.untyped_storage().data_ptr() or c != 64 or net_out._version != full._version or
                (w > 1 and net_out.stride(2) != 64) or (h > 1 and net_out.stride(1) != W0 * 64) or net_out.stride(3) != 1):
            return None
        off = net_out.storage_offset() - full.storage_offset()
        if off < 0 or off % 64:
            return None
        hb, wb = divmod(off // 64, W0)
        if hb + h > H0 or wb + w > W0:
            return None
        B = self.B
        zkey = (z.data_ptr(), z._version)       # (fr["img_z"] holds the keyed tensor: its address cannot be recycled meanwhile)
        if fr["img"] is None or fr.get("img_z") is None or fr.get("img_key") != (zkey, B._bound.get("denoiser.", (None, None))[1]):
            B.bind("denoiser.", G.denoiser)
            B.style("denoiser.", z, 0, fold_denoiser)
            raw = torch.empty((1, 3, H0, W0), dtype=torch.float32, device=full.device)
            fr["img"] = B.mfma_cnn(full)(full, raw=raw)
            fr["raw"] = raw
            fr["img_key"] = (zkey, B._bound["denoiser."][1])
            fr["img_z"] = z
            if self.on_frame is not None:
                self.on_frame(fr)
        self.stats["cnn_tiles_from_frame"] += 1
        return fr["img"][:, :, hb:hb + h, wb:wb + w], fr["raw"][:, :, hb:hb + h, wb:wb + w]


def _render_net_reason(net):
    """Why the fused kernel cannot stand in for this LightningMLP (None: it can) -- the size / structure part of
    modules.LightningMLPNative.native_reason, for the reference's plain class as well."""
    try:
        if net.fc_viewdir is not None:
            return "viewdir_dim > 0"
        if not (tuple(net.fc_1.weight.shape) == (256, 128) and tuple(net.fc_out_c.weight.shape) == (64, 256) and
                net.fc_sigma.weight.shape[0] == 1 and (not net.use_seg or net.fc_m_a.weight.shape[1] == 12)):
            return "layer sizes other than 128 -> 256 x6 -> 1 + 64 with 12 labels"
        for i in (2, 3, 4, 5, 6):
            f = getattr(net, f"fc_{i}")
            if not (f.output_mode and f.mod_bias and f.bias is None):
                return f"fc_{i} is not a bias-free output-mode ModLinear"
    except AttributeError as e:
        return f"not a LightningMLP ({e})"
    return None


def binding(G, aux=None, term_eps=None):
    """The generator's GeneratorBinding (created on first use).  aux: also produce the other eleven return values of
    _forward_perpix.  term_eps: early ray termination threshold of the field kernel for this generator (0 = evaluate every
    sample exactly like the reference; default fused.TERM_EPS_DEFAULT, which moves net_out by at most 2 x eps = 1e-4)."""
    b = G.__dict__.get("_sdn_binding")
    if b is None:
        b = G.__dict__["_sdn_binding"] = GeneratorBinding()
    if aux is not None:
        b.aux = bool(aux)
    if term_eps is not None:
        b.B.term_eps = float(term_eps)
        b._frame = None
    return b


def _count(b, key, why=None):
    b.stats[key] += 1
    if why:
        b.stats["why"][why] = b.stats["why"].get(why, 0) + 1


# ---------------------------------------------------------------------------------------------------------------------
# the two methods
# ---------------------------------------------------------------------------------------------------------------------
def fast_forward_perpix(self, blk_feats, voxel_id, depth2, raydirs, cam_ori_t, z, global_enc):
    """Generator._forward_perpix (scenedreamer.py:313-430) on sdn_field_render.  Same arguments; returns the same 12-tuple
    (see the module docstring for which entries are filled)."""
    b = binding(self)
    why = b.why_not_perpix(self, voxel_id, depth2, raydirs, cam_ori_t, z, global_enc)
    if why is not None:
        _count(b, "perpix_reference", why)
        return self._forward_perpix_reference(blk_feats, voxel_id, depth2, raydirs, cam_ori_t, z, global_enc)
    _count(b, "perpix_fast")
    B = b.B
    with torch.no_grad():
        b.sync(self, z, global_enc)
        why = b.style_refusal()
        if why is not None:
            b.stats["perpix_fast"] -= 1
            _count(b, "perpix_reference", why)
    if why is not None:
        return self._forward_perpix_reference(blk_feats, voxel_id, depth2, raydirs, cam_ori_t, z, global_enc)
    with torch.no_grad():
        _, h, w, M, _ = voxel_id.shape
        ns = int(self.num_samples)
        win, vid, d2, rd, sky_c, sky_mean = b.rays(self, voxel_id, depth2, raydirs)
        if hasattr(self, "sky_avg"):                          # the frame-wide pre-pass of inference_givenstyle, :592-598
            sky_avg = self.sky_avg
        else:                                                 # sky_global_avgpool over the rays of this call, :392-393
            sky_avg = sky_mean if isinstance(sky_mean, torch.Tensor) else sky_c.mean(dim=0)
        u = None
        if not self.coarse_deterministic_sampling:            # the reference's draw, mc_utils.py:121 (nsamples = num_samples + 1)
            u = torch.rand([1, h, w, ns + 1, 1], dtype=depth2.dtype, device=depth2.device).reshape(h * w, ns + 1)
        aux = dict.fromkeys(fused.AUX_OUTPUTS) if b.aux else None
        out = [None] * len(PERPIX_OUTPUTS)
        if (b.coalesce and isinstance(sky_mean, dict) and u is None and aux is None and hasattr(self, "sky_avg") and
                win.rows * win.cols < win.n_src and getattr(win, "pitch_known", True)):
            # a tile of a frame whose arrays (and sky features) are all at hand: the frame is evaluated once, tiles are views
            full = b.frame_field(self, sky_mean, win, (vid, d2, rd), (voxel_id, depth2, raydirs), cam_ori_t, sky_avg, ns)
            hb, wb = divmod(win.first, win.pitch)
            out[0] = full[:, hb:hb + h, wb:wb + w, :]
            b.stats["tiles_from_frame"] += 1
            return tuple(out)
        net_out = fused.field_render(B, vid, d2, rd, cam_ori_t, sky_c, sky_avg, ns, u=u, window=win, aux=aux)
        out[0] = net_out.view(1, h, w, 64)
        if aux is not None:
            # the same values the reference returns (scenedreamer.py:335-352, :373-377); samples come from the stand-alone op
            rand_depth, new_dists, new_idx = ops.sample_depth_batched(
                depth2, ns + 1, deterministic=u is None, use_box_boundaries=False, sample_depth=self.sample_depth,
                rand=u.view(1, h, w, ns + 1, 1) if u is not None else None)
            rand_depth = torch.where(torch.isnan(rand_depth) | torch.isinf(rand_depth), torch.zeros_like(rand_depth), rand_depth)
            weights = aux["weights"].view(1, h, w, ns, 1)
            out[1], out[2], out[3], out[4] = new_dists, weights, weights.sum(dim=-2, keepdim=True), rand_depth
            out[5], out[6] = aux["sigma"].view(1, h, w, ns, 1), aux["colour"].view(1, h, w, ns, 64)       # net_out_s, net_out_c
            out[7] = aux["sky_blended"].view(1, h, w, 1, 64)                                              # skynet_out_c (blended, :401)
            out[8] = aux["nosky"].view(1, h, w, 1, 1).to(torch.float32)                                   # nosky_mask (float, :383)
            out[9] = voxel_id[:, :, :, [-1], :] == 0
            out[10] = voxel_id[:, :, :, [0], :] == 0
            out[11] = new_idx
    return tuple(out)


def fast_forward_global(self, net_out, z):
    """Base3DGenerator._forward_global (gancraft_base.py:588-603) on the MFMA render CNN: net_out [1,H,W,64] -> (tanh image,
    conv4 output) [1,3,H,W] each."""
    b = binding(self)
    den = self.denoiser
    why = None
    if not modules._cuda_f32(net_out, z):
        why = "inputs are not CUDA float32 tensors"
    elif modules._wants_grad(den, net_out, z):
        why = "autograd requested (the kernels are forward-only)"
    elif net_out.dim() != 4 or net_out.shape[0] != 1 or net_out.shape[-1] != 64 or z.shape[0] != 1:
        why = "batch size > 1 or a feature width other than 64"
    elif tuple(den.conv1.weight.shape) != (256, 64, 1, 1):
        why = "denoiser sizes"
    if why is not None:
        _count(b, "global_reference", why)
        return self._forward_global_reference(net_out, z)
    _count(b, "global_fast")
    B = b.B
    with torch.no_grad():
        if b.coalesce:
            got = b.frame_image_tile(self, net_out, z)
            if got is not None:
                return got
        B.bind("denoiser.", den)
        B.style("denoiser.", z, 0, fold_denoiser)
        x = net_out.contiguous()
        raw = torch.empty((1, 3, x.shape[1], x.shape[2]), dtype=torch.float32, device=x.device)
        img = B.mfma_cnn(x)(x, raw=raw)
    return img, raw


# ---------------------------------------------------------------------------------------------------------------------
# installation
# ---------------------------------------------------------------------------------------------------------------------
def accelerate(G, aux=False, term_eps=None):
    """Bind the fast path to an existing reference generator: the classes of its render_net / sky_net / denoiser get the
    native forwards (parameters untouched), its _forward_perpix / _forward_global become the methods above."""
    for name in ("render_net", "sky_net", "denoiser"):
        modules.make_fast(type(getattr(G, name)))
    cls = type(G)
    if "_forward_perpix_reference" not in G.__dict__ and not hasattr(cls, "_forward_perpix_reference"):
        G._forward_perpix_reference = types.MethodType(cls._forward_perpix, G)
        G._forward_global_reference = types.MethodType(cls._forward_global, G)
        G._forward_perpix = types.MethodType(fast_forward_perpix, G)
        G._forward_global = types.MethodType(fast_forward_global, G)
    binding(G, aux=aux, term_eps=term_eps)
    return G


def _patch_layers(mod):
    mod.LightningMLP = modules.make_fast(mod.LightningMLP)


def _patch_base(mod):
    mod.SKYMLP = modules.make_fast(mod.SKYMLP)
    mod.RenderCNN = modules.make_fast(mod.RenderCNN)


def _patch_generator(mod):
    G = mod.Generator
    if "_forward_perpix_reference" not in G.__dict__:
        G._forward_perpix_reference = G._forward_perpix
        G._forward_global_reference = G._forward_global
        G._forward_perpix = fast_forward_perpix
        G._forward_global = fast_forward_global


_TARGETS = {"imaginaire.model_utils.layers": _patch_layers, "imaginaire.generators.gancraft_base": _patch_base,
            "imaginaire.generators.scenedreamer": _patch_generator}


class _PatchingLoader(importlib.abc.Loader):
    def __init__(self, inner, patch):
        self.inner, self.patch = inner, patch

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        self.patch(module)

    def __getattr__(self, name):          # get_code / get_source / is_package ...: the real loader answers
        return getattr(self.inner, name)


class _Finder(importlib.abc.MetaPathFinder):
    """Post-processes the three modules of _TARGETS right after the import system has executed them: the reference's
    source files are read and executed unchanged; only names in the finished module objects are rebound."""
    def find_spec(self, name, path, target=None):
        if name not in _TARGETS:
            return None
        spec = None
        for finder in sys.meta_path:                      # whoever would have found the module without this hook
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(name, path, target)
            if spec is not None:
                break
        if spec is None or spec.loader is None:
            return None
        spec.loader = _PatchingLoader(spec.loader, _TARGETS[name])
        return spec


_finder = None


def install_import_hook():
    """Arrange for the reference's classes to come out of `import imaginaire...` with the native forwards (idempotent).
    Modules that were imported before the hook are patched in place."""
    global _finder
    if _finder is None:
        _finder = _Finder()
        sys.meta_path.insert(0, _finder)
    for name, patch in _TARGETS.items():
        if name in sys.modules:
            patch(sys.modules[name])
    # `from imaginaire.model_utils.layers import LightningMLP` in an already imported scenedreamer module bound the old class
    sd, ly = sys.modules.get("imaginaire.generators.scenedreamer"), sys.modules.get("imaginaire.model_utils.layers")
    if sd is not None and ly is not None:
        sd.LightningMLP = ly.LightningMLP


def uninstall_import_hook():
    global _finder
    if _finder is not None:
        if _finder in sys.meta_path:
            sys.meta_path.remove(_finder)
        _finder = None
