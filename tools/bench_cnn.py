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
cnn = MfmaCNN(R)
H, W = 570, 990
x = torch.rand(1, H, W, 64, device=dev) * 2 - 1
buf = cnn._buffers(H, W)
t_all = _time_ms(lambda: cnn(x), 5)
t_conv = _time_ms(lambda: cnn._conv(buf["a"], "conv2a", H, W, bias=R.w["denoiser.conv2a.bias"], dst=buf["b"]), 5)
print(f"cnn total {t_all:.3f} ms   one conv3x3 {t_conv:.3f} ms   dbg={os.environ.get('SDN_CONV_DBG','0')}")
