#!/usr/bin/env python
"""Headline benchmark: rendered frames/s at 960x540, 24 samples/ray, scene_size 2048 (BASELINE.json).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = every rank renders ONE complete frame of the reference's pattern-0 camera orbit
(ray casting on the padded 570x990 frame -> sky MLP -> per-sample hash-grid + render MLP + volume
rendering -> render CNN -> 540x960 image), synthetic scene / random-init weights of the reference's
shapes (no checkpoint is available offline).  Frames are independent, so ranks shard the trajectory
with no data-path collective (weak scaling: K frames per rank); RCCL is only used before the timed
region to broadcast the scene volume / weights / style code from rank 0.

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` for the dominant
grid-sample kernel and, at N=1, `cpu_baseline` (the CPU oracle timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

# Host threads for the cpu_baseline leg: both OpenMP runtimes in the process (PyTorch's and the C oracle's)
# read this at load time.  Capped at 32: on the 256-thread GPU host two runtimes spinning 256 threads each
# made the oracle 10x SLOWER than on 8 cores.
CPU_THREADS = int(os.environ.get("SDN_CPU_THREADS", min(os.cpu_count() or 1, 32)))
os.environ.setdefault("OMP_NUM_THREADS", str(CPU_THREADS))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # the per-scene world-encoder convolutions: no exhaustive MIOpen search

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E spec
BYTES_PER_SAMPLE_UNFUSED = 16916  # 16384 gather + 20 coords in + 512 features out
BYTES_PER_SAMPLE_FUSED = 16404    # 16384 gather + 20 coords in


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default=os.environ.get("SDN_BENCH_MODE", "auto"), choices=["auto", "fused", "unfused"])
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--samples", type=int, default=24)
    ap.add_argument("--scene-size", type=int, default=2048)
    ap.add_argument("--apron", default="minimal", choices=["minimal", "reference"],
                    help="field/CNN evaluated on the 4-px apron the image can depend on (bit-identical image), or on the "
                         "reference's full 15-px apron")
    ap.add_argument("--no-overlap", action="store_true", help="do not cast the next frame's rays on a second stream")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra loops (other apron setting, delivered rate): for the very large configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    return ap.parse_args()


def cpu_baseline(args, weights, scene_cpu_small, budget_s):
    """The reference path as the CPU oracle executes it (reference-literal tiling, fp32, all host
    cores): oracle/sdn_oracle.c for the three native ops + oracle/field_ref.py for the Python layers.
    Bounded sample: ONE 158x158-ray tile (the reference's own tile size incl. apron) at 24 samples/ray
    on a 256^2 synthetic scene; frames/s is extrapolated by ray count to the 40 tiles = 828 000
    tile-rays the reference executes per 960x540 frame."""
    from oracle import field_ref as FR
    from oracle import oracle as O
    from scenedreamer_amd import camera, synth
    from scenedreamer_amd.renderer import load_label_lut
    cores = CPU_THREADS
    torch.set_num_threads(cores)
    lut = load_label_lut()["lut"]
    sc = scene_cpu_small
    pose = camera.eval_camera_poses(sc, maxstep=8)[2]
    z = FR.style_mlp(weights, synth.make_style(8888))
    genc = FR.world_encoder(weights, sc.current_height_map, sc.current_semantic_map)
    hw = (128, 128)  # + 30 px apron = one reference tile of 158 x 158 rays
    vox = sc.voxel_t.numpy()
    p = (pose[0].numpy(), pose[1].numpy(), pose[2].numpy(), pose[3])
    t0 = time.time()
    reps = 0
    while True:
        FR.render_frame_tiled(weights, lut, vox, p, hw, args.samples, z, genc)
        reps += 1
        if time.time() - t0 > budget_s or reps >= 16:      # ~10-30 s of host work (bounded sample)
            break
    dt = (time.time() - t0) / reps
    tile_rays = 158 * 158
    frame_tile_rays = 828000 if (args.height, args.width) == (540, 960) else None
    if frame_tile_rays is None:
        from scenedreamer_amd.camera import tile_grid
        tiles, _, _ = tile_grid([args.height + 30, args.width + 30], 30)
        frame_tile_rays = sum((a[1] - a[0]) * (a[3] - a[2]) for a in tiles)
    fps = 1.0 / (dt * frame_tile_rays / tile_rays)
    return {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"one 158x158-ray reference tile ({args.samples} samples/ray, 256^2 scene) rendered {reps}x in "
                      f"{dt:.2f} s each by the CPU oracle; extrapolated by ray count to the {frame_tile_rays} "
                      f"tile-rays of a {args.width}x{args.height} frame", "threads_oracle_c": O.num_threads(), "host_cpus": os.cpu_count()}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="env://")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from scenedreamer_amd import camera, capi, synth
    from scenedreamer_amd import dist as sdist
    from scenedreamer_amd.renderer import Renderer
    capi.lib()  # hard requirement: no fallback

    mode = args.mode
    if mode == "auto":
        try:
            from scenedreamer_amd import fused  # noqa: F401
            mode = "fused"
        except ImportError:
            mode = "unfused"

    # ---- scene / weights / style: built on rank 0, broadcast over RCCL ----------------------------
    t_setup = time.time()
    scene = synth.make_scene(args.scene_size, 3407, device=dev) if rank == 0 or world == 1 else None
    weights = synth.make_weights(0) if rank == 0 or world == 1 else None
    style = synth.make_style(8888) if rank == 0 or world == 1 else None
    if world > 1:
        scene, weights, style = sdist.broadcast_state(scene, weights, style, dev, src=0)
    R = Renderer(weights, scene, dev)
    R.set_style(style)
    maxstep = 40
    poses = camera.eval_camera_poses(scene, maxstep=maxstep)
    # every 2nd pose of the orbit, sharded round-robin over ranks (frame f -> rank f % world)
    order = [(2 * i) % maxstep for i in range(maxstep)]
    # weak scaling: step k renders global frames k*world .. k*world+world-1, rank r takes frame k*world + r
    frame_pose = lambda k: poses[order[sdist.shard_frames(range(k * world, (k + 1) * world), rank, world)[0] % len(order)]]
    hw = (args.height, args.width)
    setup_s = time.time() - t_setup

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for k in range(args.warmup):
        R.render_frame(frame_pose(k), hw, args.samples, mode=mode, apron=args.apron)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    timed_poses = [frame_pose(args.warmup + k) for k in range(args.steps)]
    if args.no_overlap or mode != "fused":
        frames = (R.render_frame(pz, hw, args.samples, mode=mode, apron=args.apron) for pz in timed_poses)
    else:   # all K frames are cast, evaluated and finished inside the timed region; frame k+1's ray casting runs beside frame k
        frames = R.render_frames(timed_poses, hw, args.samples, mode=mode, apron=args.apron)
    for k, img in enumerate(frames):
        marks[k + 1].record()       # no sync: per-frame device times for the p10 / p50 / p90 spread (DDA work is pose dependent)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    frame_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    pct = lambda q: frame_ms[min(len(frame_ms) - 1, int(round(q * (len(frame_ms) - 1))))]

    # ---- delivered rate: frame in host memory as uint8 HWC (async D2H, PNG/MP4 encoding excluded), outside the timed region
    from scenedreamer_amd.output import to_uint8_hwc
    n_del = 0 if args.no_extras else min(args.steps, 10)
    pinned = [torch.empty((hw[0], hw[1], 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for k in range(n_del):
        im = R.render_frame(frame_pose(args.warmup + k), hw, args.samples, mode=mode, apron=args.apron)
        pinned[k & 1].copy_(to_uint8_hwc(im), non_blocking=True)
    torch.cuda.synchronize()
    delivered_fps = n_del / (time.perf_counter() - t2) if n_del else None

    # ---- per-stage breakdown + roofline of the dominant kernel (outside the timed region) ----------
    stages = {}
    for k in range(min(args.steps, 5)):
        R.render_frame(frame_pose(args.warmup + k), hw, args.samples, mode=mode, timers=stages, apron=args.apron)
    stage_ms = {k: float(np.mean(v)) for k, v in stages.items()}
    # the same frames with the other apron setting (outside the timed region; reported for transparency)
    other = "reference" if args.apron == "minimal" else "minimal"
    n_other = 0 if args.no_extras else min(args.steps, 8)
    for k in range(2 if n_other else 0):
        R.render_frame(frame_pose(k), hw, args.samples, mode=mode, apron=other)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k in range(n_other):
        R.render_frame(frame_pose(args.warmup + k), hw, args.samples, mode=mode, apron=other)
    torch.cuda.synchronize()
    other_ms = 1000.0 * (time.perf_counter() - t1) / n_other if n_other else None
    roof, roof_grid = R.measure_roofline(frame_pose(args.warmup), hw, args.samples, mode)

    if rank == 0:
        fps = world * args.steps / elapsed
        out = {
            # BASELINE.json's metric verbatim ("...; HBM GB/s": value = frames/s, the grid sampler's GB/s is in roofline_grid_sampler)
            "metric": f"rendered frames/sec @{args.width}\u00d7{args.height}, {args.samples} samples/ray, scene_size {args.scene_size}; HBM GB/s",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": R.compute_dtype(mode), "data": "synthetic",
            "config": {"workload": f"{args.width}x{args.height}, num_samples={args.samples}, "
                                   f"scene_size={args.scene_size}, cam pattern 0 (every 2nd of 40 poses), "
                                   f"1 frame per rank per step", "path": mode, "apron": args.apron,
                       "ray_casting_overlap": not (args.no_overlap or mode != "fused"),
                       "padded_rays": (hw[0] + 30) * (hw[1] + 30),
                       "field_rays": (hw[0] + 8) * (hw[1] + 8) if (args.apron == "minimal" and mode == "fused") else (hw[0] + 30) * (hw[1] + 30),
                       "samples_per_frame": ((hw[0] + 8) * (hw[1] + 8) if (args.apron == "minimal" and mode == "fused") else (hw[0] + 30) * (hw[1] + 30)) * args.samples,
                       "parallelism": f"frames x{world}",
                       "apron_note": "ray casting and the sky MLP always cover the reference's padded frame (15-px apron); "
                                     "'minimal' evaluates the field MLP and the CNN on the 4-px apron that can reach a kept "
                                     "pixel -- the image is bit-identical (tests/test_render_gpu.py, test_fullsize_gpu.py)"},
            "frame_ms_p10_p50_p90": [pct(0.1), pct(0.5), pct(0.9)], "delivered_frames_per_s_uint8_host": delivered_fps,
            "stage_ms": stage_ms, "setup_s": setup_s, f"ms_per_step_apron_{other}": other_ms,
            "roofline": roof, "roofline_grid_sampler": roof_grid,
        }
        if world == 1 and not args.no_cpu_baseline:
            small = synth.make_scene(256, 3407)
            out["cpu_baseline"] = cpu_baseline(args, weights if isinstance(weights, dict) else None, small,
                                               args.cpu_budget_s)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
