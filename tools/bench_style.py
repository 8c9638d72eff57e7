#!/usr/bin/env python
"""Frames/s of the pipelined trajectory loop at the headline config for another weight seed / style, with the precision rungs the
per-style calibration picks (what a style that does not fit the cheapest forms costs):  python tools/bench_style.py [wseed] [style]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, scene as scene_mod, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer  # noqa: E402

wseed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
style = int(sys.argv[2]) if len(sys.argv) > 2 else 8888
scene = scene_mod.to_compact(synth.make_scene(2048, 3407, device="cuda"))
R = Renderer(synth.make_weights(wseed), scene, "cuda")
R.set_style(synth.make_style(style))
poses = camera.eval_camera_poses(scene, maxstep=40)
sel = [poses[(2 * k) % 40] for k in range(25)]
for _ in R.render_frames(sel[:5], (540, 960), 24, mode="fused"):
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in R.render_frames(sel[5:], (540, 960), 24, mode="fused"):
    pass
torch.cuda.synchronize()
ms = 1000.0 * (time.perf_counter() - t0) / 20
c, g = R.cnn_calibration, R.field_gate
print(json.dumps({"weights_seed": wseed, "style": style, "frames_per_s": 1000.0 / ms, "ms_per_frame": ms, "cnn_terms3x3": c["terms3x3"],
                  "cnn_diffs_vs_3term": c.get("max_abs_diff_vs_3term"), "image_err_vs_fp32": c["image_err_vs_fp32"], "path": g["path"],
                  "colour_terms": g["colour"]["terms"], "net_out_err": g["max_abs_err_vs_fp32"], "sky_terms": g["sky"]["hidden_terms"]}))
