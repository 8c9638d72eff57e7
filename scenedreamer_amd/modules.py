"""The render networks of the hot path as drop-in nn.Modules on libsdnative's MFMA kernels.

`LightningMLP`, `SKYMLP` and `RenderCNN` keep the reference's constructor arguments, parameter names and forward
signatures (imaginaire/model_utils/layers.py:60-126, imaginaire/generators/gancraft_base.py:132-169, :172-225), so a
reference state dict loads unchanged and `imaginaire.generators.scenedreamer.Generator` builds and calls them as it
calls its own ("Render MLP -- an nn.Module boundary").  Their forward runs

    LightningMLP.forward  ->  sdn_render_mlp   (field.hip mlp_kernel<MODE_RAW>: the fused field kernel's layer machinery)
    SKYMLP.forward        ->  sdn_sky_mlp      (sky_kernel; positional-encoded rows in, or -- when the argument is the
                                                tagged output of this package's voxlib.positional_encoding -- the ray
                                                directions themselves, the encoding then runs inside the kernel)
    RenderCNN.forward     ->  sdn_conv_head / sdn_conv x4 / sdn_conv_chain (cnn.MfmaCNN)

whenever the call is one the kernels implement: CUDA float32 tensors, no gradient requested, the layer sizes of the
SceneDreamer configuration.  Any other call (training with autograd, CPU tensors, other sizes) is evaluated by
`_forward_composite`: the same arithmetic as plain PyTorch ops (for the classes `make_fast` derives from the reference's
own classes it is the reference's own forward).  There is no silent CPU substitute for the kernels: a CUDA call with
libsdnative missing raises (capi.lib()).

Two ways in:
  * `make_fast(RefClass)`: the REFERENCE's own class (its constructor, its parameters, its forward kept as the composite
    path) gets the native forward.  dropin.install_import_hook() does this while the unmodified `imaginaire` package is
    being imported.
  * the stand-alone classes below, for use without the reference's Python tree.
"""
import itertools
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import capi
from .renderer import Renderer, fold_denoiser, fold_render_net, fold_sky_net


class Backend:
    """What fused.* and cnn.MfmaCNN expect of a renderer (`w`: the reference's parameter names -> device tensors, `dev`, the
    per-style constants, the packed-weight caches), fed from LIVE nn.Module parameters: `bind` aliases them, and notices
    in-place updates (load_state_dict, an optimizer step) and re-allocations (.cuda()) through the tensors' version counters
    and addresses, so packed weights are rebuilt exactly when they are stale."""
    mfma_cnn = Renderer.mfma_cnn
    _cnn_form = Renderer._cnn_form
    _drop_other_cnn_planes = Renderer._drop_other_cnn_planes
    set_precision = Renderer.set_precision

    def __init__(self):
        self.dev = None
        self.w = {}
        self.M, self.sample_depth, self.dists_scale, self.pad = 6, 3.0, 0.25, 0
        self._fused_scene = self._fused_style = self._fused_sky = None
        self.cnn_calibration = None
        self._bound = {}
        self._zkey = {}
        self._zref = {}     # the style tensor each _zkey was taken from, kept ALIVE: see style()
        self.sky_terms_auto = self._sky_gate_key = self.sky_gate = None

    @staticmethod
    def tensors_key(module):
        return tuple((t.data_ptr(), t._version) for t in itertools.chain(module.parameters(), module.buffers()))

    def bind(self, prefix, module):
        """Alias module's parameters / buffers as w[prefix + name].  True when anything changed since the last call."""
        key = self.tensors_key(module)
        old = self._bound.get(prefix)
        if old is not None and old[0] is module and old[1] == key:
            return False
        for k in [k for k in self.w if k.startswith(prefix)]:
            del self.w[k]
        for k, v in module.state_dict().items():
            self.w[prefix + k] = v                  # state_dict() hands out detached aliases of the live tensors
        dev = next(itertools.chain(module.parameters(), module.buffers())).device
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if self.dev is not None and self.dev != dev:
            self.w = {k: v for k, v in self.w.items() if k.startswith(prefix)}
            self._bound.clear()
            self._fused_scene = self._fused_style = self._fused_sky = None
            self.__dict__.pop("_mfma_cnns", None)
        self.dev = dev
        self._bound[prefix] = (module, key)
        self._zkey.pop(prefix, None)
        self._zref.pop(prefix, None)
        if prefix == "denoiser.":
            self.__dict__.pop("_mfma_cnns", None)   # packed convolution weights
            self.cnn_calibration = None
        if prefix == "hash_encoder.":
            self._fused_scene = None
        if prefix == "sky_net.":
            self._reset_sky_gate()
        return True

    def _reset_sky_gate(self):
        """The sky MLP's hidden-layer form is a per-style, per-weights MEASUREMENT (SKYMLPNative.forward): new weights or a new
        style start from the 3-term form until it has been re-measured."""
        self.sky_terms_auto = None
        self._sky_gate_key = None
        self.sky_gate = None

    def style(self, prefix, z, item, fold):
        """Fold style code z[item] for the network under `prefix` unless that very tensor content was folded already.

        "Already" = same address, same version counter, same shape -- which only means "same content" while the tensor the
        key was taken from is still alive: the reference builds `z = self.style_net(style)` afresh per call
        (scenedreamer.py:505) and the caching allocator hands a freed z's address (version 0 again) to the next one.  So the
        keyed tensor is HELD here (`_zref`); while it lives its storage cannot be recycled, and a tensor with the same address
        can only be a view of the same storage, whose writes bump the shared version counter."""
        key = (z.data_ptr(), z._version, tuple(z.shape), tuple(z.stride()), item)
        if self._zkey.get(prefix) != key or self._zref.get(prefix) is None:
            fold(self, z[item:item + 1].detach().to(torch.float32).reshape(1, -1))
            self._zkey[prefix] = key
            self._zref[prefix] = z
            if prefix == "sky_net.":
                self._reset_sky_gate()


def _backend(module):
    b = module.__dict__.get("_sdn_backend")
    if b is None:
        b = module.__dict__["_sdn_backend"] = Backend()
    return b


def _wants_grad(module, *tensors):
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())


def _cuda_f32(*tensors):
    return all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 for t in tensors)


# ---------------------------------------------------------------------------------------------------------------------
# native forwards (mixins: they only rely on the reference's attribute names)
# ---------------------------------------------------------------------------------------------------------------------
class LightningMLPNative:
    """forward(x [N,H,W,M,128], raydir (unused: viewdir_dim = 0), z [N,style], m [N,H,W,M,12] one-hot) -> (sigma [..,1], c [..,64])."""
    _sdn_native = True

    def native_reason(self, x, raydir, z, m):
        """None when the MFMA kernel serves this call, else why not."""
        if not _cuda_f32(x, z) or (self.use_seg and not _cuda_f32(m)):
            return "needs CUDA float32 tensors"
        if _wants_grad(self, x, z):
            return "autograd requested (the kernels are forward-only)"
        if self.fc_viewdir is not None:
            return "viewdir_dim > 0"
        if x.dim() != 5 or tuple(self.fc_1.weight.shape) != (256, 128) or x.shape[-1] != 128:
            return "needs in_channels = 128, hidden_channels = 256"
        if tuple(self.fc_out_c.weight.shape) != (64, 256) or self.fc_sigma.weight.shape[0] != 1:
            return "needs out_channels_c = 64, out_channels_s = 1"
        if self.use_seg and (self.fc_m_a.weight.shape[1] != 12 or m.shape[-1] != 12):
            return "needs mask_dim = 12"
        for i in (2, 3, 4, 5, 6):
            f = getattr(self, f"fc_{i}")
            if not (getattr(f, "output_mode", False) and getattr(f, "mod_bias", False) and getattr(f, "bias", None) is None):
                return f"fc_{i} is not a bias-free output-mode ModLinear"
        if z.dim() != 2 or z.shape[0] != x.shape[0]:
            return "z must be [N, style_dim]"
        if self.use_seg and m.numel():
            # the kernel replaces fc_m_a(m) by ONE row of fc_m_a^T (the label bias): only a one-hot m means that.  Anything
            # else the reference accepts (soft / smoothed labels, an all-zero row) goes to the composite path -- one fused
            # reduction and one device -> host read per call; the generator-level binding (dropin.py) builds the labels
            # itself and does not come through here.
            mf = m.reshape(-1, m.shape[-1])
            if not bool(((mf.max(dim=-1).values == 1) & (mf.sum(dim=-1) == 1)).all()):
                return "m is not one-hot"
        return None

    def forward(self, x, raydir, z, m):
        why = self.native_reason(x, raydir, z, m)
        if why is not None:
            self.__dict__["_sdn_composite_reason"] = why
            return self._forward_composite(x, raydir, z, m)
        from . import fused
        lib = capi.lib()
        B = _backend(self)
        B.bind("render_net.", self)
        n, h, w_, ms, _ = x.shape
        rows = h * w_ * ms
        sigma = torch.empty((n, h, w_, ms, 1), dtype=torch.float32, device=x.device)
        c = torch.empty((n, h, w_, ms, 64), dtype=torch.float32, device=x.device)
        if rows == 0:
            return sigma, c
        with torch.no_grad():
            for i in range(n):
                B.style("render_net.", z, i, fold_render_net)
                try:
                    st = B._fused_style or fused.prepare_style(B)
                except fused.TrunkRangeError as e:        # weights the packed f16 stream cannot hold: the reference's arithmetic
                    self.__dict__["_sdn_composite_reason"] = str(e)
                    return self._forward_composite(x, raydir, z, m)
                ct, _ = fused.precision_profile(B)
                xi = x[i].reshape(rows, 128).contiguous()
                if self.use_seg:
                    # fc_m_a(m) for a ONE-HOT row is row `argmax` of fc_m_a^T (what scenedreamer.py:357-363 builds m for)
                    mi = m[i].reshape(rows, 12)                       # one-hot: native_reason checked it
                    lab = mi.argmax(dim=-1).to(torch.uint8)
                else:
                    lab = torch.zeros(rows, dtype=torch.uint8, device=x.device)
                if "ticket" not in st:
                    st["ticket"] = torch.zeros(2, dtype=torch.int32, device=B.dev)
                with torch.cuda.device(B.dev):
                    capi.check(lib.sdn_render_mlp(xi.data_ptr(), lab.data_ptr(), st["packed_mx" if ct == 6 else "packed"].data_ptr(),
                                                  st["consts"].data_ptr(), sigma[i].data_ptr(), c[i].data_ptr(), rows, 6 if ct == 6 else 3,
                                                  0, st["ticket"].data_ptr(), capi.current_stream(B.dev)), "sdn_render_mlp")
        return sigma, c


class SKYMLPNative:
    """forward(x [N,...,33] positional-encoded ray directions, z [N,style]) -> c [N,...,64]."""
    _sdn_native = True

    def native_reason(self, x, z):
        if not _cuda_f32(x, z):
            return "needs CUDA float32 tensors"
        if _wants_grad(self, x, z):
            return "autograd requested (the kernels are forward-only)"
        if tuple(self.fc1.weight.shape) != (256, 33) or tuple(self.fc_out_c.weight.shape) != (64, 256) or x.shape[-1] != 33:
            return "needs in_channels = 33, hidden_channels = 256, out_channels_c = 64"
        if not isinstance(self.act, nn.LeakyReLU) or self.act.negative_slope != 0.2:
            return "needs LeakyReLU(0.2)"
        if z.dim() != 2 or z.shape[0] != x.shape[0]:
            return "z must be [N, style_dim]"
        return None

    def forward(self, x, z):
        why = self.native_reason(x, z)
        if why is not None:
            self.__dict__["_sdn_composite_reason"] = why
            return self._forward_composite(x, z)
        from . import fused
        B = _backend(self)
        B.bind("sky_net.", self)
        shape = tuple(x.shape[:-1]) + (64,)
        per = x[0].numel() // 33
        if per == 0:
            return torch.empty(shape, dtype=torch.float32, device=x.device)
        src = getattr(x, "_sdn_pe_src", None)    # set by ops.positional_encoding: (input, its version, ndegrees, dim, incl_orig, out version)
        results = []
        with torch.no_grad():
            for i in range(x.shape[0]):
                B.style("sky_net.", z, i, fold_sky_net)
                rd = None
                if (src is not None and x.shape[0] == 1 and src[2] == 5 and src[4] and src[0].shape[-1] == 3 and
                        src[3] in (-1, src[0].dim() - 1) and src[0]._version == src[1] and x._version == src[5] and
                        src[0].is_contiguous() and src[0].numel() == per * 3):
                    rd = src[0].detach().reshape(per, 3)
                if rd is not None:
                    # the argument IS positional_encoding(rd, 5, -1, True) (this package's op produced it and nobody wrote to it
                    # since): the kernel evaluates the encoding itself, and the per-ray result is kept for
                    # Generator._forward_perpix, which asks for sky_net of the very same rays again for every tile
                    # (scenedreamer.py:368-370 after the frame-wide pre-pass :592-598)
                    # The renderer's per-style sky gate (Renderer.calibrate_style) for this surface: on the first frame of a style the
                    # hidden layers are evaluated as f16 + fp6 corrections AND as the 3-term split (4e-6 from fp32); the cheap form is
                    # kept for the style if the two stay within SKY_AUTO_BOUND of each other.  One extra launch per style.
                    gate_key = (B._zkey.get("sky_net."), B._bound["sky_net."][1])
                    if getattr(B, "_sky_gate_key", None) != gate_key and "SDN_SKY_TERMS" not in os.environ and getattr(B, "sky_terms", None) is None:
                        from .renderer import SKY_AUTO_BOUND
                        B.sky_terms_auto = None
                        c3, _ = fused.sky_fused(B, rd)
                        B.sky_terms_auto = 6
                        c6, _ = fused.sky_fused(B, rd)
                        d = float((c6 - c3).abs().max())
                        if not d <= SKY_AUTO_BOUND:
                            B.sky_terms_auto = None
                        B._sky_gate_key = gate_key
                        B.sky_gate = {"hidden_terms": 6 if B.sky_terms_auto == 6 else 3, "max_abs_diff_fp6_vs_3term": d, "bound": SKY_AUTO_BOUND}
                    sky_c, _ = fused.sky_fused(B, rd)
                    # (rd_ref keeps the ray-direction storage alive: while this record exists its address cannot be handed to
                    #  another tensor, so "same address + same version counter" below means "same content")
                    self.__dict__["_sdn_last_frame"] = dict(rd_ref=src[0], rd_ptr=rd.data_ptr(), rd_version=src[1], n_rays=per, sky_c=sky_c,
                                                            zkey=B._zkey.get("sky_net."), wkey=B._bound["sky_net."][1])
                else:
                    sky_c, _ = fused.sky_fused(B, x[i].reshape(per, 33), encoded=True)
                results.append(sky_c)
        return (results[0] if len(results) == 1 else torch.stack(results)).view(shape)


class RenderCNNNative:
    """forward(x [N,64,H,W], z [N,style]) -> conv4 output [N,3,H,W] (before tanh, like the reference)."""
    _sdn_native = True

    def native_reason(self, x, z):
        if not _cuda_f32(x, z):
            return "needs CUDA float32 tensors"
        if _wants_grad(self, x, z):
            return "autograd requested (the kernels are forward-only)"
        if x.dim() != 4 or tuple(self.conv1.weight.shape) != (256, 64, 1, 1) or x.shape[1] != 64:
            return "needs in_channels = 64, hidden_channels = 256"
        if not isinstance(self.act, nn.LeakyReLU) or self.act.negative_slope != 0.2:
            return "needs LeakyReLU(0.2)"
        if z.dim() != 2 or z.shape[0] != x.shape[0]:
            return "z must be [N, style_dim]"
        return None

    def forward(self, x, z):
        why = self.native_reason(x, z)
        if why is not None:
            self.__dict__["_sdn_composite_reason"] = why
            return self._forward_composite(x, z)
        B = _backend(self)
        B.bind("denoiser.", self)
        n, _, H, W = x.shape
        raw = torch.empty((n, 3, H, W), dtype=torch.float32, device=x.device)
        with torch.no_grad():
            for i in range(n):
                B.style("denoiser.", z, i, fold_denoiser)
                net_out = x[i:i + 1].permute(0, 2, 3, 1).contiguous()      # channels-last rows [1,H,W,64]
                B.mfma_cnn(net_out)(net_out, raw=raw[i:i + 1])
        return raw


def make_fast(ref_cls):
    """Give the reference's LightningMLP / SKYMLP / RenderCNN class the native forward (its own forward stays reachable as
    `_forward_composite`).  The class object itself is patched, not subclassed: the reference's constructors call
    `super(SKYMLP, self).__init__()` through the module-global name, which must keep meaning the class they are defined in."""
    mixin = {"LightningMLP": LightningMLPNative, "SKYMLP": SKYMLPNative, "RenderCNN": RenderCNNNative}[ref_cls.__name__]
    if not is_native(ref_cls):
        ref_cls._forward_composite = ref_cls.forward
        ref_cls.forward = mixin.forward
        ref_cls.native_reason = mixin.native_reason
        ref_cls._sdn_native = True
    return ref_cls


def is_native(module_or_cls):
    cls = module_or_cls if isinstance(module_or_cls, type) else type(module_or_cls)
    return bool(getattr(cls, "_sdn_native", False))


# ---------------------------------------------------------------------------------------------------------------------
# stand-alone classes (no reference tree needed): the same parameters, the composite forward written out
# ---------------------------------------------------------------------------------------------------------------------
def _lrelu(v):
    return F.leaky_relu(v, 0.2)


class ModLinear(nn.Module):
    """Linear layer whose weight columns are scaled by a style-dependent alpha and whose bias gets a style-dependent beta
    (layers.py:198-271): parameters `weight` [out,in], `bias` (optional), `weight_alpha` / `bias_alpha` [in,style] / [in],
    `weight_beta` / `bias_beta` over the outputs (output_mode) or the inputs."""

    def __init__(self, in_features, out_features, style_features, bias=True, mod_bias=True, output_mode=False, weight_gain=1,
                 bias_init=0):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_features, in_features) * (weight_gain / np.sqrt(in_features)))
        self.bias = nn.Parameter(torch.full((out_features,), float(bias_init))) if bias else None
        self.weight_alpha = nn.Parameter(torch.randn(in_features, style_features) / np.sqrt(style_features))
        self.bias_alpha = nn.Parameter(torch.ones(in_features))
        self.mod_bias, self.output_mode = mod_bias, output_mode
        self.weight_beta = self.bias_beta = None
        if mod_bias:
            d = out_features if output_mode else in_features
            self.weight_beta = nn.Parameter(torch.randn(d, style_features) / np.sqrt(style_features))
            self.bias_beta = nn.Parameter(torch.zeros(d))

    def forward(self, x, z):
        shape = x.shape
        x = x.reshape(shape[0], -1, shape[-1])
        z = z.reshape(z.shape[0], 1, z.shape[-1]).to(x.dtype)
        alpha = z @ self.weight_alpha.t().to(x.dtype) + self.bias_alpha.to(x.dtype)            # [N,1,in]
        wmod = self.weight.to(x.dtype)[None] * alpha                                            # [N,out,in]
        add = self.bias.to(x.dtype)[None, None, :] if self.bias is not None else None
        if self.mod_bias:
            beta = z @ self.weight_beta.t().to(x.dtype) + self.bias_beta.to(x.dtype)
            if self.output_mode:
                add = beta if add is None else add + beta
            else:
                x = x + beta
        y = torch.bmm(x, wmod.transpose(1, 2))
        if add is not None:
            y = y + add
        return y.reshape(*shape[:-1], y.shape[-1])


class AffineMod(nn.Module):
    """x * alpha(z) + beta(z) (layers.py:127-176)."""

    def __init__(self, in_features, style_features, mod_bias=True):
        super().__init__()
        self.weight_alpha = nn.Parameter(torch.randn(in_features, style_features) / np.sqrt(style_features))
        self.bias_alpha = nn.Parameter(torch.ones(in_features))
        self.mod_bias = mod_bias
        self.weight_beta = self.bias_beta = None
        if mod_bias:
            self.weight_beta = nn.Parameter(torch.randn(in_features, style_features) / np.sqrt(style_features))
            self.bias_beta = nn.Parameter(torch.zeros(in_features))

    def forward(self, x, z):
        shape = x.shape
        x = x.reshape(shape[0], -1, shape[-1])
        z = z.reshape(z.shape[0], 1, z.shape[-1]).to(x.dtype)
        x = x * (z @ self.weight_alpha.t().to(x.dtype) + self.bias_alpha.to(x.dtype))
        if self.mod_bias:
            x = x + (z @ self.weight_beta.t().to(x.dtype) + self.bias_beta.to(x.dtype))
        return x.reshape(shape)


class LightningMLP(LightningMLPNative, nn.Module):
    """layers.py:60-126 -- same constructor, parameter names and forward."""

    def __init__(self, in_channels, style_dim, viewdir_dim, mask_dim=680, out_channels_s=1, out_channels_c=3,
                 hidden_channels=256, use_seg=True):
        super().__init__()
        self.use_seg = use_seg
        if use_seg:
            self.fc_m_a = nn.Linear(mask_dim, hidden_channels, bias=False)
        self.fc_viewdir = nn.Linear(viewdir_dim, hidden_channels, bias=False) if viewdir_dim > 0 else None
        self.fc_1 = nn.Linear(in_channels, hidden_channels)
        mk = lambda: ModLinear(hidden_channels, hidden_channels, style_dim, bias=False, mod_bias=True, output_mode=True)
        self.fc_2, self.fc_3, self.fc_4 = mk(), mk(), mk()
        self.fc_sigma = nn.Linear(hidden_channels, out_channels_s)
        if viewdir_dim > 0:
            self.fc_5 = nn.Linear(hidden_channels, hidden_channels, bias=False)
            self.mod_5 = AffineMod(hidden_channels, style_dim, mod_bias=True)
        else:
            self.fc_5 = mk()
        self.fc_6 = mk()
        self.fc_out_c = nn.Linear(hidden_channels, out_channels_c)
        self.act = nn.LeakyReLU(negative_slope=0.2)

    def _forward_composite(self, x, raydir, z, m):
        z = z[:, None, None, None, :]
        f = self.fc_1(x)
        if self.use_seg:
            f = f + self.fc_m_a(m)
        f = self.act(f)
        for layer in (self.fc_2, self.fc_3, self.fc_4):
            f = self.act(layer(f, z))
        sigma = self.fc_sigma(f)
        if self.fc_viewdir is not None:
            f = self.act(self.mod_5(self.fc_5(f) + self.fc_viewdir(raydir), z))
        else:
            f = self.act(self.fc_5(f, z))
        f = self.act(self.fc_6(f, z))
        return sigma, self.fc_out_c(f)


class SKYMLP(SKYMLPNative, nn.Module):
    """gancraft_base.py:129-169 -- same constructor, parameter names and forward."""

    def __init__(self, in_channels, style_dim, out_channels_c=3, hidden_channels=256, leaky_relu=True):
        super().__init__()
        self.fc_z_a = nn.Linear(style_dim, hidden_channels, bias=False)
        self.fc1 = nn.Linear(in_channels, hidden_channels)
        self.fc2 = nn.Linear(hidden_channels, hidden_channels)
        self.fc3 = nn.Linear(hidden_channels, hidden_channels)
        self.fc4 = nn.Linear(hidden_channels, hidden_channels)
        self.fc5 = nn.Linear(hidden_channels, hidden_channels)
        self.fc_out_c = nn.Linear(hidden_channels, out_channels_c)
        self.act = nn.LeakyReLU(negative_slope=0.2) if leaky_relu else nn.ReLU()

    def _forward_composite(self, x, z):
        s = self.fc_z_a(z)
        s = s.reshape(s.shape[0], *([1] * (x.dim() - 2)), s.shape[-1])
        y = self.act(self.fc1(x) + s)
        for layer in (self.fc2, self.fc3, self.fc4, self.fc5):
            y = self.act(layer(y))
        return self.fc_out_c(y)


class RenderCNN(RenderCNNNative, nn.Module):
    """gancraft_base.py:172-225 -- same constructor, parameter names and forward (the output is conv4's, before tanh)."""

    def __init__(self, in_channels, style_dim, hidden_channels=256, leaky_relu=True):
        super().__init__()
        self.fc_z_cond = nn.Linear(style_dim, 2 * 2 * hidden_channels)
        self.conv1 = nn.Conv2d(in_channels, hidden_channels, 1)
        self.conv2a = nn.Conv2d(hidden_channels, hidden_channels, 3, padding=1)
        self.conv2b = nn.Conv2d(hidden_channels, hidden_channels, 3, padding=1, bias=False)
        self.conv3a = nn.Conv2d(hidden_channels, hidden_channels, 3, padding=1)
        self.conv3b = nn.Conv2d(hidden_channels, hidden_channels, 3, padding=1, bias=False)
        self.conv4a = nn.Conv2d(hidden_channels, hidden_channels, 1)
        self.conv4b = nn.Conv2d(hidden_channels, hidden_channels, 1)
        self.conv4 = nn.Conv2d(hidden_channels, 3, 1)
        self.act = nn.LeakyReLU(negative_slope=0.2) if leaky_relu else nn.ReLU()

    def _forward_composite(self, x, z):
        a = torch.chunk(self.fc_z_cond(z), 4, dim=-1)
        film = lambda v, s, b: v * (s[..., None, None] + 1) + b[..., None, None]
        y = self.act(self.conv1(x))
        y = self.act(film(y + self.conv2b(self.act(self.conv2a(y))), a[0], a[1]))
        y = self.act(film(y + self.conv3b(self.act(self.conv3a(y))), a[2], a[3]))
        y = self.act(y + self.conv4b(self.act(self.conv4a(y))))
        return self.conv4(y)
