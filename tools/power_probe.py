#!/usr/bin/env python
"""Run one kernel of the frame in a tight loop for a few seconds so that rocm-smi can sample power / clocks under it.
    python tools/power_probe.py mlp|conv|encode <marker file>"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, capi, fused, synth
from scenedreamer_amd.renderer import Renderer
from scenedreamer_amd.cnn import MfmaCNN
which, marker = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
scene = synth.make_scene(2048, 3407, device=dev)
R = Renderer(synth.make_weights(0), scene, dev)
R.set_style(synth.make_style(8888))
pose = camera.eval_camera_poses(scene, maxstep=40)[10]
with torch.no_grad():
    vid, d2, rd, cam_res = R.cast_rays(pose, (540, 960))
    n = cam_res[0] * cam_res[1]
    vid, d2, rd = vid.view(n, R.M).contiguous(), d2.view(2, n, R.M).contiguous(), rd.view(n, 3).contiguous()
    ori = torch.as_tensor(pose[0], dtype=torch.float32)
    sky_c, sky_avg = fused.sky_fused(R, rd)
    buf = fused.encode(R, vid, d2, rd, ori, 24)
    st = R._fused_style or fused.prepare_style(R)
    net_out = torch.empty((n, 64), device=dev)
    cnn = MfmaCNN(R, int(os.environ.get("SDN_CNN_TERMS", "1")))
    x = torch.rand(1, cam_res[0], cam_res[1], 64, device=dev) * 2 - 1
    fns = {
        "mlp": lambda: fused._launch_mlp(R, buf, st, sky_c, sky_avg.reshape(-1), net_out, n, 24),
        "conv": lambda: cnn(x),
        "encode": lambda: fused.encode(R, vid, d2, rd, ori, 24, buf),
    }
    fn = fns[which]
    fn(); torch.cuda.synchronize()
    open(marker, "w").write("go")
    t0 = time.time()
    k = 0
    while time.time() - t0 < 14:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        k += 10
    print(f"{which}: {k} launches, {(time.time() - t0) / k * 1e3:.3f} ms each")
