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
here; this class takes the decision as `terms3x3`: 1, 3, or one choice per layer (conv2a, conv2b, conv3a, conv3b) as a 4-tuple or
a string like "1113" -- the rungs between "all one product" and "all 3-term" (CNN_LADDER: the LAST 3x3 layers go 3-term first:
on the synthetic weights conv3b alone takes 20 - 30 % off the 1-term error for +0.6 ms, profiles/r05_cnn_ladder_study.txt; a
2-term split of all four layers costs twice that and corrects the weights only -- the activations' rounding is what matters)."""
import ctypes
import os

import torch

from . import capi

_LAYERS = {"conv1": (64, 1), "conv2a": (256, 9), "conv2b": (256, 9), "conv3a": (256, 9), "conv3b": (256, 9),
           "conv4a": (256, 1), "conv4b": (256, 1)}
_3X3 = ("conv2a", "conv2b", "conv3a", "conv3b")
CNN_LADDER = (1, "1113", "1133", 3)       # cheapest first: 4, 6, 8, 12 one-product layer equivalents


def form_key(terms3x3):
    """Canonical name of a 3x3 precision form: 1, 3 (all four layers alike) or a 4-character string of 1 / 3."""
    if isinstance(terms3x3, (tuple, list)):
        terms3x3 = "".join(str(int(t)) for t in terms3x3)
    if isinstance(terms3x3, str):
        terms3x3 = terms3x3.strip()
        if terms3x3 in ("1", "3"):          # (environment variables arrive as strings)
            return int(terms3x3)
        if not (len(terms3x3) == 4 and set(terms3x3) <= {"1", "3"}):
            raise ValueError(f"3x3 precision form {terms3x3!r}: expected 1, 3 or four characters of 1 / 3 (conv2a, conv2b, conv3a, conv3b), e.g. 1113")
        return 1 if terms3x3 == "1111" else 3 if terms3x3 == "3333" else terms3x3
    assert int(terms3x3) in (1, 3), terms3x3
    return int(terms3x3)


def form_terms(terms3x3):
    """(conv2a, conv2b, conv3a, conv3b) product terms of a form."""
    k = form_key(terms3x3)
    return (k,) * 4 if isinstance(k, int) else tuple(int(c) for c in k)


def form_cost(terms3x3):
    return sum(form_terms(terms3x3))


class MfmaCNN:
    def __init__(self, R, terms3x3, chain=None):
        self.R = R
        self.form = form_key(terms3x3)
        per3x3 = dict(zip(_3X3, form_terms(terms3x3)))
        if chain is None:
            chain = os.environ.get("SDN_CNN_CHAIN", "1") != "0"
        self.chain = bool(chain)
        self.terms = {n: per3x3.get(n, 3) for n in _LAYERS}
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

    FLOP_PER_PIXEL = {"head (conv1)": 2 * 64 * 256, "conv2a": 2 * 9 * 256 * 256, "conv2b": 2 * 9 * 256 * 256, "conv3a": 2 * 9 * 256 * 256,
                      "conv3b": 2 * 9 * 256 * 256, "chain (conv4a, conv4b, conv4)": 2 * (2 * 256 * 256 + 256 * 3)}    # sums to 5 015 040

    def __call__(self, net_out, raw=None, timers=None):
        """net_out [1,H,W,64] -> image [1,3,H,W] (tanh).  raw: optional f32 [1,3,H,W] that receives conv4's output before the
        tanh (RenderCNN.forward's own return value, gancraft_base.py:221-225; needs the chained tail).
        timers: optional dict; receives (start, end) event pairs per launch, keyed like FLOP_PER_PIXEL (bench.py's per-kernel records)."""
        last = [None]

        def tick(name=None):
            if timers is None:
                return
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            if name is not None:
                timers.setdefault(name, []).append((last[0], e))
            last[0] = e
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
        tick()
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
        # the inner activations are consumed only by conv2b / conv3b: no lo plane when that layer is 1-term
        inner = lambda consumer: (B[0], None) if self.terms[consumer] == 1 else B
        tick("head (conv1)")
        self._conv(A, "conv2a", H, W, bias=bias("conv2a"), dst=inner("conv2b"))                    # act(conv2a(y))
        tick("conv2a")
        self._conv(B, "conv2b", H, W, bias=bias("conv2b"), resid_planes=A, mod=(a[0], a[1]), dst=A)
        tick("conv2b")
        self._conv(A, "conv3a", H, W, bias=bias("conv3a"), dst=inner("conv3b"))
        tick("conv3a")
        self._conv(B, "conv3b", H, W, bias=bias("conv3b"), resid_planes=A, mod=(a[2], a[3]), dst=A)
        tick("conv3b")
        if self.chain:
            with torch.cuda.device(R.dev):
                capi.check(capi.lib().sdn_conv_chain(A[0].data_ptr(), A[1].data_ptr(), self.chain_packed.data_ptr(),
                                                     self.chain_consts.data_ptr(), img.data_ptr(),
                                                     raw.data_ptr() if raw is not None else None, H, W, 0,
                                                     capi.current_stream(R.dev)), "sdn_conv_chain")
            tick("chain (conv4a, conv4b, conv4)")
            return img
        self._conv(A, "conv4a", H, W, bias=bias("conv4a"), dst=B)
        self._conv(B, "conv4b", H, W, bias=bias("conv4b"), resid_planes=A, proj=(self.w4, self.b4), img=img)
        return img
