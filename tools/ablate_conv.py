#!/usr/bin/env python
"""Timing ablations of the 1-term 3x3 convolution (build with SDN_MLP_ABLATION=1; results are wrong unless dbg = 0).
bits: 1 no DMA in the loop, 2 no barrier, 16 no MFMA, 64 no fragment reads from LDS (sums as listed in sdn_conv's switch)."""
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
H, W = 548, 968
x = torch.rand(1, H, W, 64, device=dev) * 2 - 1
buf = cnn._buffers(H, W)
cnn(x)
for rep in range(2):
    for dbg in (0, 1, 2, 3, 16, 17, 19, 64, 65, 66, 67):
        os.environ["SDN_CONV_DBG"] = str(dbg)
        t = _time_ms(lambda: cnn._conv(buf["a"], "conv2a", H, W, bias=R.w["denoiser.conv2a.bias"], dst=buf["b"]), 10)
        print(f"dbg {dbg:3d}: conv3x3 {t:.3f} ms", flush=True)
