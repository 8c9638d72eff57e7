#!/usr/bin/env python
"""Render a few frames of the headline config (for rocprofv3 runs: no MIOpen search noise from setup
is avoidable, but the per-frame kernels dominate after the first frame)."""
import os
import sys

os.environ.setdefault("SDN_FIELD_GATE", "0")   # profiling runs: no calibration launches (fp32 frame, other precision forms) in the trace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dev = torch.device("cuda:0")
scene = synth.make_scene(S, 3407, device=dev)
R = Renderer(synth.make_weights(0), scene, dev)
R.set_style(synth.make_style(8888))
poses = camera.eval_camera_poses(scene, maxstep=40)
for k in range(frames):
    R.render_frame(poses[(2 * k) % 40], (540, 960), 24, mode=mode)
torch.cuda.synchronize()
print("done", mode, frames)
