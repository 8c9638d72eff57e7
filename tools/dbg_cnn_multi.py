#!/usr/bin/env python
"""MFMA CNN vs torch CNN on frames with more patches than workgroups (every workgroup runs several patches)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import synth
from scenedreamer_amd.renderer import Renderer
from scenedreamer_amd.cnn import MfmaCNN
dev = torch.device("cuda:0")
scene = synth.make_scene(256, 3407, device=dev)
R = Renderer(synth.make_weights(0, grid_log2_hashmap=10), scene, dev)
R.set_style(synth.make_style(8888))
torch.manual_seed(0)
out = []
for terms in (1, 3):
    for hw in ((128, 200), (272, 400), (300, 520)):
        x = torch.rand(1, hw[0], hw[1], 64, device=dev) * 2 - 1
        ref = R.render_cnn(x)
        got = MfmaCNN(R, terms)(x)
        out.append(f"t{terms} {hw[0]}x{hw[1]}: {(got - ref).abs().max().item():.1e}")
print("  ".join(out))
