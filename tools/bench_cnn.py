#!/usr/bin/env python
"""Time one 3x3 conv launch of the MFMA render CNN on the padded headline frame.  env: SDN_CONV_DBG"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import synth
from scenedreamer_amd.renderer import Renderer, _time_ms
from scenedreamer_amd.cnn import MfmaCNN
dev = torch.device("cuda:0")
scene = synth.make_scene(256, 3407, device=dev)
R = Renderer(synth.make_weights(0, grid_log2_hashmap=10), scene, dev)
R.set_style(synth.make_style(8888))
cnn = MfmaCNN(R, int(os.environ.get("SDN_CNN_TERMS", "1")))
H, W = 548, 968   # the benchmark frame with its apron
x = torch.rand(1, H, W, 64, device=dev) * 2 - 1
buf = cnn._buffers(H, W)
t_all = _time_ms(lambda: cnn(x), 5)
t_conv = _time_ms(lambda: cnn._conv(buf["a"], "conv2a", H, W, bias=R.w["denoiser.conv2a.bias"], dst=buf["b"]), 5)
t_11 = _time_ms(lambda: cnn._conv(buf["a"], "conv4a", H, W, bias=R.w["denoiser.conv4a.bias"], dst=buf["b"]), 5)
t_1 = _time_ms(lambda: cnn._conv(buf["b"], "conv1", H, W, bias=R.w["denoiser.conv1.bias"], dst=buf["a"]), 5)
img = torch.empty(1, 3, H, W, device=dev)
t_4b = _time_ms(lambda: cnn._conv(buf["b"], "conv4b", H, W, bias=R.w.get("denoiser.conv4b.bias"), resid_planes=buf["a"], proj=(cnn.w4, cnn.b4), img=img), 5)
print(f"conv1 {t_1:.3f} ms  conv4a {t_11:.3f} ms  conv4b+proj {t_4b:.3f} ms")
if cnn.chain:
    from scenedreamer_amd import capi
    t_ch = _time_ms(lambda: capi.check(capi.lib().sdn_conv_chain(buf["a"][0].data_ptr(), buf["a"][1].data_ptr(), cnn.chain_packed.data_ptr(),
                                                                  cnn.chain_consts.data_ptr(), img.data_ptr(), None, H, W, 0,
                                                                  capi.current_stream(dev)), "sdn_conv_chain"), 5)
    print(f"conv4a -> conv4b -> conv4 as one chain {t_ch:.3f} ms (as launches: {t_11 + t_4b:.3f} ms)")
    xs = x.reshape(-1, 64).contiguous()
    t_hd = _time_ms(lambda: capi.check(capi.lib().sdn_conv_head(xs.data_ptr(), cnn.head_packed.data_ptr(), cnn.head_bias.data_ptr(),
                                                                buf["a"][0].data_ptr(), buf["a"][1].data_ptr(), H, W, 0,
                                                                capi.current_stream(dev)), "sdn_conv_head"), 5)
    t_pl = _time_ms(lambda: capi.check(capi.lib().sdn_conv_planes_from_f32(xs.data_ptr(), 64, buf["b"][0].data_ptr(), buf["b"][1].data_ptr(),
                                                                           H, W, capi.current_stream(dev)), "planes"), 5)
    print(f"head (rows -> conv1 -> planes) {t_hd:.3f} ms (as launches: planes {t_pl:.3f} + conv1 {t_1:.3f} ms)")
print(f"cnn total {t_all:.3f} ms   one conv3x3 {t_conv:.3f} ms   dbg={os.environ.get('SDN_CONV_DBG','0')}")
