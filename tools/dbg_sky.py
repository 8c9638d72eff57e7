"""Sky MLP: hidden layers as f16 + fp6 (SDN_SKY_TERMS=6) vs the 3-term split vs PyTorch fp32; kernel time."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import golden
from scenedreamer_amd import fused, synth
from scenedreamer_amd.renderer import Renderer, _time_ms
g = golden("field_a.npz")
scene = synth.make_scene(256, 3407, device="cuda")
R = Renderer(synth.make_weights(0), scene, "cuda")
R.set_style_code(g["z"])
rd = torch.nn.functional.normalize(torch.randn(564300, 3, device="cuda"), dim=-1)
ref = R.sky_features(rd)
for t in (3, 6):
    R.sky_terms = t
    got, avg = fused.sky_fused(R, rd)
    ms = _time_ms(lambda: fused.sky_fused(R, rd), 5)
    print("sky terms", t, "max err vs torch fp32", (got - ref).abs().max().item(), "mean", (got - ref).abs().mean().item(),
          "avg err", (avg - ref.mean(0, keepdim=True)).abs().max().item(), "ms", ms, flush=True)
