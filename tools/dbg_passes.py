#!/usr/bin/env python
"""Debug: per-group pass counts written by mlp_kernel (sdn_field_mlp `passes`)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import synth, camera, fused
from scenedreamer_amd.renderer import Renderer
dev = torch.device("cuda:0")
scene = synth.make_scene(256, 3407, device=dev)
R = Renderer(synth.make_weights(0, grid_log2_hashmap=10), scene, dev)
R.set_style(synth.make_style(8888))
pose = camera.eval_camera_poses(scene, maxstep=8)[5]
for hw in ((96, 128), (540, 960)):
    vid, d2, rd, cam_res = R.cast_rays(pose, hw)
    n = cam_res[0] * cam_res[1]
    vid, d2, rd = vid.view(n, R.M), d2.view(2, n, R.M), rd.view(n, 3)
    ori = torch.as_tensor(pose[0], dtype=torch.float32)
    sky_c, sky_avg = fused.sky_fused(R, rd)
    passes = torch.full(((n + 31) // 32,), 77, dtype=torch.uint8, device=dev)
    fused.field_fused(R, vid, d2, rd, ori, sky_c, sky_avg, 24, passes=passes)
    torch.cuda.synchronize()
    g = torch.nn.functional.pad((vid[:, 0] != 0), (0, (-n) % 32)).view(-1, 32).any(dim=1)
    vals, cnt = torch.unique(passes, return_counts=True)
    print(hw, "groups", passes.numel(), "hit groups", int(g.sum()), "histogram", dict(zip(vals.tolist(), cnt.tolist())),
          "first 20", passes[:20].tolist(), "idx of first 77:", (passes == 77).nonzero()[:5].flatten().tolist())
