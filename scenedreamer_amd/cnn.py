"""Host side of the MFMA render CNN (csrc/cnn.hip): RenderCNN.forward + tanh
(imaginaire/generators/gancraft_base.py:202-225, :588-603) with the four 3x3 convolutions on libsdnative and the
1x1 convolutions as channels-last GEMMs through PyTorch."""
import ctypes

import torch
import torch.nn.functional as F

from . import capi


def _lrelu(x):
    return F.leaky_relu(x, 0.2)


class MfmaCNN:
    def __init__(self, R):
        self.R = R
        lib = capi.lib()
        w = R.w
        self.packed = {}
        nbytes = lib.sdn_conv_packed_weight_bytes()
        with torch.cuda.device(R.dev):
            for n in ("conv2a", "conv2b", "conv3a", "conv3b"):
                buf = torch.empty(nbytes, dtype=torch.uint8, device=R.dev)
                capi.check(lib.sdn_conv_pack_weights(w[f"denoiser.{n}.weight"].contiguous().data_ptr(), buf.data_ptr(),
                                                     capi.current_stream(R.dev)))
                self.packed[n] = buf
        self.w1 = w["denoiser.conv1.weight"].reshape(256, 64).contiguous()
        self.w4a = w["denoiser.conv4a.weight"].reshape(256, 256).contiguous()
        self.w4b = w["denoiser.conv4b.weight"].reshape(256, 256).contiguous()
        self.w4 = w["denoiser.conv4.weight"].reshape(3, 256).contiguous()
        self._planes = {}

    def _buffers(self, H, W):
        key = (H, W)
        if key not in self._planes:
            hb, wb = ctypes.c_int(), ctypes.c_int()
            capi.lib().sdn_conv_plane_dims(H, W, ctypes.byref(hb), ctypes.byref(wb))
            n = hb.value * wb.value * 256
            mk = lambda: torch.zeros(n, dtype=torch.float16, device=self.R.dev)   # zero border / out-of-frame pixels
            self._planes.clear()
            self._planes[key] = dict(a=(mk(), mk()), b=(mk(), mk()),
                                     y1=torch.empty(H * W, 256, device=self.R.dev),
                                     y2=torch.empty(H * W, 256, device=self.R.dev))
        return self._planes[key]

    def _conv(self, src, name, H, W, bias=None, resid=None, mod=None, dst=None, out32=None):
        p = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(self.R.dev):
            capi.check(capi.lib().sdn_conv3x3(src[0].data_ptr(), src[1].data_ptr(), self.packed[name].data_ptr(), p(bias),
                                              p(resid), p(mod[0]) if mod else None, p(mod[1]) if mod else None,
                                              dst[0].data_ptr() if dst else None, dst[1].data_ptr() if dst else None,
                                              p(out32), H, W, 0, capi.current_stream(self.R.dev)), "sdn_conv3x3")

    def __call__(self, net_out):
        """net_out [1,H,W,64] -> image [1,3,H,W] (tanh)."""
        R, w = self.R, self.R.w
        _, H, W, _ = net_out.shape
        buf = self._buffers(H, W)
        a = [t[0].contiguous() for t in torch.chunk(R.cnn_adapt, 4, dim=-1)]
        x = net_out.reshape(H * W, 64)
        y0 = _lrelu(torch.addmm(w["denoiser.conv1.bias"], x, self.w1.t()))                       # conv1 (1x1)
        with torch.cuda.device(R.dev):
            capi.check(capi.lib().sdn_conv_planes_from_f32(y0.data_ptr(), buf["a"][0].data_ptr(), buf["a"][1].data_ptr(),
                                                           H, W, capi.current_stream(R.dev)))
        self._conv(buf["a"], "conv2a", H, W, bias=w["denoiser.conv2a.bias"], dst=buf["b"])      # act(conv2a(y))
        self._conv(buf["b"], "conv2b", H, W, resid=y0, mod=(a[0], a[1]), dst=buf["a"], out32=buf["y1"])
        self._conv(buf["a"], "conv3a", H, W, bias=w["denoiser.conv3a.bias"], dst=buf["b"])
        self._conv(buf["b"], "conv3b", H, W, resid=buf["y1"], mod=(a[2], a[3]), out32=buf["y2"])
        y2 = buf["y2"]
        t = _lrelu(torch.addmm(w["denoiser.conv4a.bias"], y2, self.w4a.t()))                      # conv4a (1x1)
        y3 = _lrelu(y2 + torch.addmm(w["denoiser.conv4b.bias"], t, self.w4b.t()))                 # y + conv4b(...)
        img = torch.tanh(torch.addmm(w["denoiser.conv4.bias"], y3, self.w4.t()))                  # conv4 (1x1) + tanh
        return img.reshape(1, H, W, 3).permute(0, 3, 1, 2).contiguous()
