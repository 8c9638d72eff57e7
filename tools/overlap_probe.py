#!/usr/bin/env python
"""Does the front half of frame n+1 (rvip + encode: latency / HBM bound, few registers, no LDS) hide under the MFMA
kernels of frame n (mlp / conv: one workgroup per CU) when issued on a second stream?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, capi, fused, ops, synth
from scenedreamer_amd.renderer import Renderer
from scenedreamer_amd.cnn import MfmaCNN
dev = torch.device("cuda:0")
scene = synth.make_scene(2048, 3407, device=dev)
R = Renderer(synth.make_weights(0), scene, dev)
R.set_style(synth.make_style(8888))
poses = camera.eval_camera_poses(scene, maxstep=40)
ns = 24
pose, pose2 = poses[4], poses[6]
with torch.no_grad():
    vid, d2, rd, cam_res = R.cast_rays(pose, (540, 960))
    n = cam_res[0] * cam_res[1]
    vid, d2, rd = vid.view(n, R.M).contiguous(), d2.view(2, n, R.M).contiguous(), rd.view(n, 3).contiguous()
    ori = torch.as_tensor(pose[0], dtype=torch.float32)
    sky_c, sky_avg = fused.sky_fused(R, rd)
    buf = fused.encode(R, vid, d2, rd, ori, ns)
    buf2 = {k: v.clone() for k, v in buf.items()}
    st = R._fused_style or fused.prepare_style(R)
    st["consts"][st["sky_off"]:st["sky_off"] + 64] = sky_avg.reshape(-1)
    net_out = torch.empty((n, 64), device=dev)
    cnn = MfmaCNN(R, int(os.environ.get("SDN_CNN_TERMS", "1")))
    x = torch.rand(1, cam_res[0], cam_res[1], 64, device=dev) * 2 - 1
    ori2 = torch.as_tensor(pose2[0], dtype=torch.float32)
    f2, c2, _ = camera.frame_intrinsics(pose2[3], (540, 960), 30)

    def mlp():
        fused._launch_mlp(R, buf, st, sky_c, sky_avg.reshape(-1), net_out, n, ns)

    def front():
        v, dd, r = ops.ray_voxel_intersection_perspective(scene.voxel_t, pose2[0], pose2[1], pose2[2], f2, c2, cam_res, R.M)
        fused.encode(R, v.view(n, R.M), dd.view(2, n, R.M), r.view(n, 3), ori2, ns, buf2)

    side = torch.cuda.Stream()

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def seq_mlp():
        mlp(); front()

    def ovl_mlp():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            front()
            ev2 = torch.cuda.Event(); ev2.record()
        mlp()
        torch.cuda.current_stream().wait_event(ev2)

    def seq_cnn():
        cnn(x); front()

    def ovl_cnn():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            front()
            ev2 = torch.cuda.Event(); ev2.record()
        cnn(x)
        torch.cuda.current_stream().wait_event(ev2)

    print(f"mlp alone {timed(mlp):.2f} ms   front (rvip+encode) alone {timed(front):.2f} ms   cnn alone {timed(lambda: cnn(x)):.2f} ms")
    print(f"mlp ; front sequential {timed(seq_mlp):.2f} ms   overlapped {timed(ovl_mlp):.2f} ms")
    print(f"cnn ; front sequential {timed(seq_cnn):.2f} ms   overlapped {timed(ovl_cnn):.2f} ms")
