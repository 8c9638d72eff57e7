#!/usr/bin/env python
"""Sample compaction for the colour branch (north star: "wavefront ballots for early termination AND sample compaction"; VERDICT r5
row n-3): what would a field kernel execute if it PACKED the live samples (relu(sigma) * dist > 0: the only ones whose colour is
multiplied by a non-zero weight, mc_utils.py:154-161) before fc_5 / fc_6 / fc_out_c, instead of skipping the colour branch only for
passes whose 128 samples are all dead (what csrc/field.hip ships)?  Model on the REAL sigma field of the benchmark frames (and of the
fog / surface weight sets), with the kernel's measured cycle costs.

  what is counted, per pose:
    live            fraction of the evaluated samples with weight != 0
    run_now         fraction of the evaluated passes whose colour branch the shipped kernel runs (its own `colour_passes` counter)
    run_ideal_skip  ... if every all-dead pass were skipped (no adaptive decision): the ceiling of SKIPPING at 128-sample granularity
    run_packed_G    colour batches of G samples a workgroup would run with a queue of live columns, / passes (G = 32, 64, 128; groups dealt
                    round-robin to 256 workgroups, queue flushed at the end) -- equals `live` up to the flush remainder
  cost model (cycles per 128-sample pass of one workgroup, profiles/r05_layers_cycles.txt, one-kernel build, timers inflate all alike):
    colour branch when it runs  C = 34 288  (fc_5 14 618 + fc_6 12 819 + fc_out_c 6 851)
    whole pass, average         P = 130 028
    exchange per live column: 256 activations as f16 hi (512 B) + fp6 lo + scales (200 B) written to and read back from an LDS queue
      in B-fragment order = 1 424 B of LDS traffic; + the weighted colour (64 f32) added to the ray's accumulator through LDS (the
      packed column no longer sits in its ray's quad): 256 B read-modify-write.  The LDS moves 128 B / clock / CU: X = (1424 + 512) / 128
      = 15.1 clocks per live column if nothing else used the LDS -- but the weight ring's fragment reads already use 0.47 of its cycles
      in the colour layers (3 ds_read_b128 per MFMA), so the exchange cannot hide behind them: it is charged in full.
    LDS budget: a batch of 128 columns x 712 B = 91 KB + up to 127 columns of overflow = 181 KB -- the CU has 160 KB and the weight
      ring holds 128 KB of it (4 slots x 32 KB; 3 slots is the least that keeps one DMA in flight per wave).  G = 32 per WAVE: 4 x
      (32 + 31) x 712 B = 179 KB.  An in-LDS queue does not fit at any granularity that keeps the ring; a queue in HBM / L2 moves
      live x 12.7 M samples x 712 B x 2 per frame (see `hbm_queue_ms`).
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, fused, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer  # noqa: E402

C_COLOUR, P_PASS = 34288.0, 130028.0
X_COL = (1424 + 512) / 128.0
QUEUE_BYTES = 712
HBM_EFF = 4.0e12        # bytes / s a streaming kernel sustains on this part (profiles: 3.7 - 4.4 TB/s)

dev = torch.device("cuda:0")
scene = synth.make_scene(2048, 3407, device=dev)
base_w = synth.make_weights(0)
poses = camera.eval_camera_poses(scene, maxstep=40)
ns, hw = 24, (540, 960)
surface = dict(base_w)
surface["render_net.fc_sigma.bias"] = np.asarray(base_w["render_net.fc_sigma.bias"]) + np.float32(4000.0)
out = []
for wname, w, pis in (("benchmark", base_w, (0, 10, 13, 27)), ("fog", synth.fog_weights(base_w), (10,)), ("surface", surface, (26,))):
    R = Renderer(w, scene, dev)
    R.set_style(synth.make_style(8888))
    for pi in pis:
        pose = poses[pi]
        with torch.no_grad():
            vid, d2, rd, (H0, W0) = R.cast_rays(pose, hw)
            n = H0 * W0
            vid, d2, rd = vid.view(n, R.M), d2.view(2, n, R.M), rd.view(n, 3)
            sky_c, sky_avg = fused.sky_fused(R, rd)
            win = fused.Window.crop(H0, W0, 11)
            aux = {"weights": None}
            fused.field_render(R, vid, d2, rd, torch.as_tensor(pose[0], dtype=torch.float32), sky_c, sky_avg, ns, aux=aux, window=win)
            pa = torch.zeros((win.n_rays + 31) // 32, dtype=torch.uint8, device=dev)
            cp = torch.zeros_like(pa)
            fused.field_render(R, vid, d2, rd, torch.as_tensor(pose[0], dtype=torch.float32), sky_c, sky_avg, ns, passes=pa, colour_passes=cp, window=win)
            wts = aux["weights"]                                   # [n_rays (window row-major), ns]; 0 for rays that hit nothing
            live = win.groups(wts != 0)                            # [groups, 32 rays, ns] in the launch's ray order (8 x 4 pixel blocks)
            ran = pa > 0                                           # groups the kernel visited
            lg = live[ran].view(-1, 32, ns // 4, 4).permute(0, 2, 1, 3).reshape(-1, ns // 4, 128)    # [groups, passes, 128 samples]
            executed = pa[ran].long()                              # passes each group went through (early termination cuts the rest)
            pass_ok = torch.arange(ns // 4, device=dev)[None, :] < executed[:, None]
            n_pass = int(pass_ok.sum())
            live_cnt = (lg.sum(dim=-1) * pass_ok).long()          # live columns per executed pass
            rec = {"weights": wname, "pose": pi, "passes": n_pass, "live": float(live_cnt.sum()) / (128.0 * n_pass),
                   "run_now": float(cp[ran].sum(dtype=torch.int64)) / n_pass, "run_ideal_skip": float(((live_cnt > 0) & pass_ok).sum()) / n_pass}
            G = live_cnt.shape[0]
            wg = torch.arange(G, device=dev) % 256                 # groups dealt round-robin to the 256 persistent workgroups
            per_wg = torch.zeros(256, dtype=torch.long, device=dev).index_add_(0, wg, live_cnt.sum(dim=1))
            for g in (32, 64, 128):
                batches = ((per_wg + g - 1) // g).sum() * (g / 128.0)          # in units of 128-sample colour branches
                rec[f"run_packed_{g}"] = float(batches) / n_pass
            # cycles per pass (average): now / packed-128 with the exchange charged per live column
            now = rec["run_now"] * C_COLOUR
            packed = rec["run_packed_128"] * C_COLOUR + rec["live"] * 128.0 * X_COL
            rec["colour_cycles_per_pass_now"] = now
            rec["colour_cycles_per_pass_packed"] = packed
            rec["kernel_time_saved_frac"] = (now - packed) / P_PASS
            rec["hbm_queue_ms"] = rec["live"] * 128.0 * n_pass * QUEUE_BYTES * 2 / HBM_EFF * 1e3
            rec["kernel_ms_saved_if_free_exchange"] = (now - rec["run_packed_128"] * C_COLOUR) / P_PASS * 14.4
            out.append(rec)
            print(json.dumps(rec), flush=True)
    del R
b = [r for r in out if r["weights"] == "benchmark"]
print(json.dumps({"summary": "benchmark weights, mean over poses", "live": float(np.mean([r["live"] for r in b])),
                  "run_now": float(np.mean([r["run_now"] for r in b])), "run_packed_128": float(np.mean([r["run_packed_128"] for r in b])),
                  "kernel_time_saved_frac_with_lds_exchange": float(np.mean([r["kernel_time_saved_frac"] for r in b])),
                  "hbm_queue_ms": float(np.mean([r["hbm_queue_ms"] for r in b])),
                  "kernel_ms_saved_if_free_exchange": float(np.mean([r["kernel_ms_saved_if_free_exchange"] for r in b])),
                  "lds_needed_bytes": (128 + 127) * QUEUE_BYTES, "lds_free_beside_the_ring_bytes": 160 * 1024 - 128 * 1024 - (151808 - 131072)}))
