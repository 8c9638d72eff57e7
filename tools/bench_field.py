#!/usr/bin/env python
"""Time the field kernels (encode, mlp) alone on one frame of the headline config.
    python tools/bench_field.py [reps]      env: SDN_MLP_VARIANT, SDN_MLP_DBG"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, fused, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer, _time_ms  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
S = int(os.environ.get("SDN_SCENE", "2048"))
dev = torch.device("cuda:0")
scene = synth.make_scene(S, 3407, device=dev)
R = Renderer(synth.make_weights(0), scene, dev)
R.set_style(synth.make_style(8888))
poses = camera.eval_camera_poses(scene, maxstep=40)
for pi in (0, 10):
    pose = poses[pi]
    with torch.no_grad():
        vid, d2, rd, cam_res = R.cast_rays(pose, (540, 960))
        n = cam_res[0] * cam_res[1]
        vid, d2, rd = vid.view(n, R.M), d2.view(2, n, R.M), rd.view(n, 3)
        ori = torch.as_tensor(pose[0], dtype=torch.float32)
        sky_c = R.sky_features(rd)
        sky_avg = sky_c.mean(0, keepdim=True)
        hit = float((vid[:, 0] != 0).float().mean())
        t_enc = _time_ms(lambda: fused.encode(R, vid, d2, rd, ori, 24), reps)
        t_all = _time_ms(lambda: fused.field_fused(R, vid, d2, rd, ori, sky_c, sky_avg, 24), reps)
    print(f"pose {pi}: rays {n} hit-frac {hit:.3f} encode {t_enc:.3f} ms  encode+mlp {t_all:.3f} ms  mlp ~{t_all - t_enc:.3f} ms "
          f"variant={os.environ.get('SDN_MLP_VARIANT', 'lds')} dbg={os.environ.get('SDN_MLP_DBG', '0')}")
