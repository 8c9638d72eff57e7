#!/usr/bin/env python
"""Frames/s at the headline config on the opaque-surface variant of the benchmark weights (density head + 4000: what a trained field's
terrain looks like to the rays: they saturate inside the first voxels, early termination drops 5 of 6 passes), colour-branch
skipping on / off in one process (the decision costs its early sigma on every pass a group goes through there)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, scene as scene_mod, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer  # noqa: E402

bias = float(sys.argv[1]) if len(sys.argv) > 1 else 4000.0
w = dict(synth.make_weights(0))
w["render_net.fc_sigma.bias"] = np.asarray(w["render_net.fc_sigma.bias"]) + np.float32(bias)
scene = scene_mod.to_compact(synth.make_scene(2048, 3407, device="cuda"))
R = Renderer(w, scene, "cuda")
R.set_style(synth.make_style(8888))
poses = camera.eval_camera_poses(scene, maxstep=40)
sel = [poses[(2 * k) % 40] for k in range(25)]
for rep in range(2):
    for skip in (False, True):
        R.colour_skip = skip
        for _ in R.render_frames(sel[:5], (540, 960), 24, mode="fused"):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in R.render_frames(sel[5:], (540, 960), 24, mode="fused"):
            pass
        torch.cuda.synchronize()
        print(f"fc_sigma.bias + {bias:g}, colour skip {skip}: {1000.0 * (time.perf_counter() - t0) / 20:.3f} ms per frame", flush=True)
