#!/usr/bin/env python
"""Per-style calibration measurements (Renderer.calibrate_style) over poses / resolutions of the benchmark scene: how close the
fused path's measured errors come to the gates' bounds.   python tools/gate_survey.py [scene_size]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
scene = synth.make_scene(S, 3407, device="cuda")
R = Renderer(synth.make_weights(0), scene, "cuda")
R.set_style(synth.make_style(8888))
poses = camera.eval_camera_poses(scene, maxstep=40)
for hw, ns, sel in (((540, 960), 24, (0, 2, 4, 8, 9, 11, 17, 23, 30, 36)), ((1080, 1920), 40, (12, 3))):
    for pi in sel:
        for eps in (None, 0.0):
            R.set_precision(term_eps=eps)
            g = R.calibrate_style(poses[pi], hw, ns)
            m = g["measurements"]
            print(json.dumps({"hw": hw, "ns": ns, "pose": pi, "term_eps": eps, "path": g["path"], "field_err": m["field_err"], "colour_diff": m.get("colour_diff"),
                              "image_err": m["image_err"], "cnn_diff": m.get("cnn_diff")}), flush=True)
R.set_precision()
