"""Work balance of the tile-parallel path (BASELINE config 5: 3840x2160x40 in N row bands) measured on ONE GPU, band after band.

N ranks sharing one GPU time-slice it, so their per-band device times say nothing about balance (profiles/r05_dist_config5_8ranks_
1gpu.json: 450 ms for six bands, 200-260 for two -- the latter simply ran while the others waited).  Here ONE process renders the
N bands of a frame one after the other exactly as rank k would (Renderer.band_prepare -> frame-wide sky mean -> band_finish), with
device events around each, for N = 2, 4, 8: that is the time each GPU of a node would spend.  Iteration 0 cuts the bands with the
static model (Renderer.row_costs); every further iteration applies dist.rebalance_scale to the measured times, as
render_frame_tile_parallel(balance="feedback") does between consecutive frames -- on the NEXT pose of the orbit, as a trajectory
would.  Prints one JSON record per (N, iteration): bands, band_ms, imbalance = max / mean, and the frame time a node would see
(max band) next to the one-band time.

    python tools/band_balance.py [--iters 4] [--worlds 2,4,8] > gpurun_out/band_balance.jsonl
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--samples", type=int, default=40)
    ap.add_argument("--pose0", type=int, default=6)
    args = ap.parse_args()
    from scenedreamer_amd import camera, dist as sdist, scene as scene_mod, synth
    from scenedreamer_amd.renderer import Renderer
    dev = torch.device("cuda", 0)
    scene = scene_mod.to_compact(synth.make_scene(2048, 3407, device=dev))
    R = Renderer(synth.make_weights(0), scene, dev)
    R.set_style(synth.make_style(8888))
    poses = camera.eval_camera_poses(scene, maxstep=40)
    hw, ns = (args.height, args.width), args.samples
    R.calibrate_style(poses[args.pose0], hw, ns)

    def run_bands(pose, bands):
        hds, ms = [], []
        for r0, r1 in bands:          # phase 1 of every rank: ray casting + sky MLP of its band
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            hds.append(R.band_prepare(pose, hw, r0, r1, mode="fused"))
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        tot, cnt = sum(h["sky_sum"] for h in hds), sum(h["sky_cnt"] for h in hds)
        sky_avg = (tot / cnt).to(torch.float32)
        for k, h in enumerate(hds):   # phase 2: field + CNN of the band
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            R.band_finish(h, sky_avg, ns)
            b.record()
            torch.cuda.synchronize()
            ms[k] += a.elapsed_time(b)
        return ms

    one = run_bands(poses[args.pose0], [(0, hw[0])])
    one = run_bands(poses[args.pose0], [(0, hw[0])])
    print(json.dumps({"world": 1, "band_ms": one}), flush=True)
    for world in [int(v) for v in args.worlds.split(",")]:
        scale = None
        for it in range(args.iters):
            pose = poses[(args.pose0 + it) % len(poses)]          # the next frame of the trajectory
            costs = np.asarray(R.row_costs(pose, hw), np.float64)
            bands = sdist.balanced_row_bands(costs * scale if scale is not None else costs, world)
            if it == 0:
                run_bands(pose, bands)                               # (allocator / plane warm-up for these band shapes)
            ms = run_bands(pose, bands)
            rec = {"world": world, "iteration": it, "pose": (args.pose0 + it) % len(poses), "bands": bands, "band_ms": [round(v, 2) for v in ms],
                   "imbalance_max_over_mean": max(ms) / (sum(ms) / len(ms)), "frame_ms_on_a_node": max(ms), "sum_band_ms": sum(ms),
                   "one_band_ms": one[0], "speedup_vs_one_gpu": one[0] / max(ms),
                   "cut": "static model (row_costs)" if it == 0 else f"static model x feedback of {it} measured frame(s)"}
            print(json.dumps(rec), flush=True)
            scale = sdist.rebalance_scale(costs, scale, bands, ms)


if __name__ == "__main__":
    main()
