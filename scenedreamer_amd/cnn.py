"""Host side of the MFMA render CNN (csrc/cnn.hip): RenderCNN.forward + tanh
(imaginaire/generators/gancraft_base.py:202-225, :588-603).  Six launches: the head (`sdn_conv_head`, csrc/field.hip: net_out
rows -> conv1 -> LeakyReLU -> y planes in one kernel; `chain = False`: sdn_conv_planes_from_f32 + conv1 as a conv_kernel launch),
conv2a/2b/3a/3b (3x3), then the tail conv4a -> conv4b (+ residual) -> conv4 -> tanh as ONE register-resident chain on the
field MLP's layer machinery (`sdn_conv_chain`, csrc/field.hip: the 256-channel activation is read once and never written;
`chain = False`: conv4a / conv4b as conv_kernel launches with conv4 + tanh folded into conv4b's epilogue); activations
travel as f16 hi/lo planes.

Precision profile (`terms3x3`): the 1x1 layers always use the 3-term f16 split; the four 3x3 layers use either the
3-term split (< 2e-5 from the fp32 CNN) or ONE term (Whi.Xhi, both rounded to nearest: a third of the MFMAs; measured
image error vs the fp32 CNN ~7e-5 rms / < 5e-4 max on the synthetic weights, tools/precision_study.py).  The 1-term form
is lossy in a weight-dependent way, so which one runs is decided per style by Renderer.mfma_cnn (a measured gate), not
here; this class takes the decision as `terms3x3`."""
import ctypes
import os

import torch

from . import capi

_LAYERS = {"conv1": (64, 1), "conv2a": (256, 9), "conv2b": (256, 9), "conv3a": (256, 9), "conv3b": (256, 9),
           "conv4a": (256, 1), "conv4b": (256, 1)}


class MfmaCNN:
    def __init__(self, R, terms3x3, chain=None):
        self.R = R
        assert terms3x3 in (1, 3)
        if chain is None:
            chain = os.environ.get("SDN_CNN_CHAIN", "1") != "0"
        self.chain = bool(chain)
        self.terms = {n: (terms3x3 if taps == 9 else 3) for n, (_, taps) in _LAYERS.items()}
        lib = capi.lib()
        w = R.w
        self.packed = {}
        with torch.cuda.device(R.dev):
            for n, (cin, taps) in _LAYERS.items():
                wt = w[f"denoiser.{n}.weight"]
                assert tuple(wt.shape[:2]) == (256, cin) and wt.shape[2] * wt.shape[3] == taps, (n, tuple(wt.shape))
                buf = torch.empty(lib.sdn_conv_packed_weight_bytes(cin, taps, self.terms[n]), dtype=torch.uint8, device=R.dev)
                capi.check(lib.sdn_conv_pack_weights(wt.contiguous().data_ptr(), cin, taps, self.terms[n], buf.data_ptr(),
                                                     capi.current_stream(R.dev)), "sdn_conv_pack_weights")
                self.packed[n] = buf
        self.w4 = w["denoiser.conv4.weight"].reshape(3, 256).contiguous()
        self.b4 = w["denoiser.conv4.bias"].contiguous()
        self._planes = {}
        if self.chain:
            with torch.cuda.device(R.dev):
                self.chain_packed = torch.empty(lib.sdn_conv_chain_packed_weight_bytes(), dtype=torch.uint8, device=R.dev)
                w4a = w["denoiser.conv4a.weight"].reshape(256, 256).contiguous()
                w4b = w["denoiser.conv4b.weight"].reshape(256, 256).contiguous()
                capi.check(lib.sdn_conv_chain_pack_weights(w4a.data_ptr(), w4b.data_ptr(), self.w4.data_ptr(),
                                                           self.chain_packed.data_ptr(), capi.current_stream(R.dev)),
                           "sdn_conv_chain_pack_weights")
                c = torch.zeros(lib.sdn_conv_chain_consts_floats(), device=R.dev)
                c[0:256] = w["denoiser.conv4a.bias"]
                c[256:512] = w["denoiser.conv4b.bias"]
                c[512:515] = self.b4
                self.chain_consts = c
                self.head_packed = torch.empty(lib.sdn_conv_head_packed_weight_bytes(), dtype=torch.uint8, device=R.dev)
                w1 = w["denoiser.conv1.weight"].reshape(256, 64).contiguous()
                capi.check(lib.sdn_conv_head_pack_weights(w1.data_ptr(), self.head_packed.data_ptr(), capi.current_stream(R.dev)),
                           "sdn_conv_head_pack_weights")
                self.head_bias = w["denoiser.conv1.bias"].contiguous()

    def _buffers(self, H, W):
        key = (H, W)
        if key not in self._planes:
            hb, wb = ctypes.c_int(), ctypes.c_int()
            capi.lib().sdn_conv_plane_dims(H, W, ctypes.byref(hb), ctypes.byref(wb))
            n = hb.value * wb.value * 256
            mk = lambda: torch.zeros(n, dtype=torch.float16, device=self.R.dev)   # zero border / out-of-frame pixels
            while len(self._planes) >= 2:
                self._planes.pop(next(iter(self._planes)))
            self._planes[key] = dict(a=(mk(), mk()), b=(mk(), mk()))
        return self._planes[key]

    def _adapt(self):
        """The four FiLM vectors of the current style (gancraft_base.py:204), cached per cnn_adapt tensor."""
        t = self.R.cnn_adapt
        if getattr(self, "_adapt_src", None) is not t:
            self._adapt_src = t
            self._adapt_vecs = [c[0].contiguous() for c in torch.chunk(t, 4, dim=-1)]
        return self._adapt_vecs

    def _conv(self, src, name, H, W, bias=None, resid=None, resid_planes=None, mod=None, dst=None, out32=None, proj=None,
              img=None):
        p = lambda t: t.data_ptr() if t is not None else None
        cin, taps = _LAYERS[name]
        with torch.cuda.device(self.R.dev):
            capi.check(capi.lib().sdn_conv(src[0].data_ptr(), src[1].data_ptr(), cin, taps, self.terms[name],
                                           self.packed[name].data_ptr(),
                                           p(bias), p(resid), resid_planes[0].data_ptr() if resid_planes else None,
                                           resid_planes[1].data_ptr() if resid_planes else None,
                                           p(mod[0]) if mod else None, p(mod[1]) if mod else None,
                                           dst[0].data_ptr() if dst else None,
                                           dst[1].data_ptr() if dst and dst[1] is not None else None, p(out32),
                                           p(proj[0]) if proj else None, p(proj[1]) if proj else None, p(img),
                                           H, W, 0, capi.current_stream(self.R.dev)), "sdn_conv")

    def __call__(self, net_out, raw=None):
        """net_out [1,H,W,64] -> image [1,3,H,W] (tanh).  raw: optional f32 [1,3,H,W] that receives conv4's output before the
        tanh (RenderCNN.forward's own return value, gancraft_base.py:221-225; needs the chained tail)."""
        if raw is not None and not self.chain:
            raise ValueError("the pre-tanh output is produced by the chained tail (chain=True)")
        R, w = self.R, self.R.w
        _, H, W, _ = net_out.shape
        buf = self._buffers(H, W)
        A, B = buf["a"], buf["b"]
        a = self._adapt()
        bias = lambda n: w.get(f"denoiser.{n}.bias")
        x = net_out.reshape(H * W, 64).contiguous()
        img = torch.empty(1, 3, H, W, device=R.dev)
        # the running activation y lives in planes A (hi + lo f16 = y to 2^-22) and is updated in place by the
        # residual convolutions; planes B hold the inner activation of each residual block
        if self.chain:                                                                             # y = act(conv1(x))
            with torch.cuda.device(R.dev):
                capi.check(capi.lib().sdn_conv_head(x.data_ptr(), self.head_packed.data_ptr(), self.head_bias.data_ptr(),
                                                    A[0].data_ptr(), A[1].data_ptr(), H, W, 0, capi.current_stream(R.dev)),
                           "sdn_conv_head")
        else:
            with torch.cuda.device(R.dev):
                capi.check(capi.lib().sdn_conv_planes_from_f32(x.data_ptr(), 64, B[0].data_ptr(), B[1].data_ptr(), H, W,
                                                               capi.current_stream(R.dev)), "sdn_conv_planes_from_f32")
            self._conv(B, "conv1", H, W, bias=bias("conv1"), dst=A)
        # the inner activations are consumed only by conv2b / conv3b: no lo plane when those are 1-term
        inner = (B[0], None) if self.terms["conv2b"] == 1 else B
        self._conv(A, "conv2a", H, W, bias=bias("conv2a"), dst=inner)                              # act(conv2a(y))
        self._conv(B, "conv2b", H, W, bias=bias("conv2b"), resid_planes=A, mod=(a[0], a[1]), dst=A)
        self._conv(A, "conv3a", H, W, bias=bias("conv3a"), dst=inner)
        self._conv(B, "conv3b", H, W, bias=bias("conv3b"), resid_planes=A, mod=(a[2], a[3]), dst=A)
        if self.chain:
            with torch.cuda.device(R.dev):
                capi.check(capi.lib().sdn_conv_chain(A[0].data_ptr(), A[1].data_ptr(), self.chain_packed.data_ptr(),
                                                     self.chain_consts.data_ptr(), img.data_ptr(),
                                                     raw.data_ptr() if raw is not None else None, H, W, 0,
                                                     capi.current_stream(R.dev)), "sdn_conv_chain")
            return img
        self._conv(A, "conv4a", H, W, bias=bias("conv4a"), dst=B)
        self._conv(B, "conv4b", H, W, bias=bias("conv4b"), resid_planes=A, proj=(self.w4, self.b4), img=img)
        return img
