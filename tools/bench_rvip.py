#!/usr/bin/env python
"""Time sdn_rvip on the headline frame over the orbit, with and without exact empty-space skipping."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, ops, synth
from scenedreamer_amd.renderer import _time_ms
dev = torch.device("cuda:0")
scene = synth.make_scene(2048, 3407, device=dev)
poses = camera.eval_camera_poses(scene, maxstep=40)
tot = {True: 0.0, False: 0.0}
for pi in range(0, 40, 4):
    ori, d, up, cf = poses[pi]
    f, c, cam_res = camera.frame_intrinsics(cf, (540, 960), 30)
    r = {}
    for acc in (True, False):
        r[acc] = _time_ms(lambda: ops.ray_voxel_intersection_perspective(scene.voxel_t, ori, d, up, f, c, cam_res, 6, accelerate=acc), 5)
        tot[acc] += r[acc]
    print(f"pose {pi:2d}: skip {r[True]:.3f} ms   plain {r[False]:.3f} ms")
print(f"mean: skip {tot[True] / 10:.3f} ms   plain {tot[False] / 10:.3f} ms")
