"""Where a style's calibration spends its time (Renderer.calibrate_one with SDN_CAL_TIMING=1: synchronised wall clock per phase).
    python tools/cal_timing.py [crop_px ...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDN_CAL_TIMING"] = "1"
from scenedreamer_amd import camera, scene as scene_mod, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer  # noqa: E402

dev = torch.device("cuda:0")
scene = scene_mod.to_compact(synth.make_scene(2048, 3407, device=dev))
R = Renderer(synth.make_weights(0), scene, dev)
R.set_style(synth.make_style(8888))
poses = camera.eval_camera_poses(scene, maxstep=40)
for crop in [int(a) for a in sys.argv[1:]] or [256, 192, 0]:
    for rep in range(2):          # the second repetition is the warm one
        R.set_style(synth.make_style(8888))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = R.calibrate_one(poses[0], (540, 960), 24, crop_px=crop)
        torch.cuda.synchronize()
        ms = 1000.0 * (time.perf_counter() - t0)
    print(json.dumps({"crop_px": crop, "total_ms_with_phase_syncs": ms, "phases_ms": {k: round(v, 2) for k, v in m["timing_ms"].items()},
                      "image_err": {str(k): v for k, v in m["image_err"].items()}, "field_err": {str(k): v for k, v in m["field_err"].items()},
                      "window": m.get("window")}), flush=True)
