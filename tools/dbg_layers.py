#!/usr/bin/env python
"""Where a pass of the field kernel goes: per-segment cycle counters (mlp_kernel<DBG = 512>, ablation build).
    SDN_MLP_ABLATION=1 python -m scenedreamer_amd.build && python tools/dbg_layers.py [512|515] [one|two]
512: colour layers f16 + fp6 (the default profile), 515: 3-term everywhere; one = field_kernel, two = encode_kernel + mlp_kernel.
513 / 514 / 516 / 520 (two only): 512 plus one ablation -- no ring DMA / no ring barrier / no activation VALU / no fragment reads."""
import os
import sys

os.environ["SDN_MLP_DBG"] = sys.argv[1] if len(sys.argv) > 1 else "512"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, fused, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer  # noqa: E402

dev = torch.device("cuda:0")
scene = synth.make_scene(2048, 3407, device=dev)
R = Renderer(synth.make_weights(0), scene, dev)
R.set_style(synth.make_style(8888))
R.field_single_kernel = (sys.argv[2] if len(sys.argv) > 2 else "one") == "one"
if os.environ["SDN_MLP_DBG"] == "515":
    R.set_precision(colour_terms=3)
names = ["inputs (encode stage / staging)", "fc_1", "fc_2", "fc_3", "fc_4", "fc_5", "fc_6", "fc_out_c", "volume rendering",
         "between passes / groups"]
mfma = [0, 192, 384, 384, 384, 384, 384, 96, 0, 0] if os.environ["SDN_MLP_DBG"] == "515" else [0, 192, 384, 384, 384, 192, 192, 96, 0, 0]
for pi in (0,):
    pose = camera.eval_camera_poses(scene, maxstep=40)[pi]
    with torch.no_grad():
        vid, d2, rd, cam_res = R.cast_rays(pose, (540, 960))
        n = cam_res[0] * cam_res[1]
        vid, d2, rd = vid.view(n, R.M), d2.view(2, n, R.M), rd.view(n, 3)
        sky_c, sky_avg = fused.sky_fused(R, rd)
        for _ in range(2):
            out = fused.field_fused(R, vid, d2, rd, torch.as_tensor(pose[0], dtype=torch.float32), sky_c, sky_avg, 24)
        torch.cuda.synchronize()
        t = out.view(-1, 64)[0:1024:4, 3:16].double().cpu()          # one row per workgroup
    passes = t[:, 10].sum()
    tot = t[:, :10].sum() + t[:, 11].sum()
    print(f"  colour branch skipped in {int(t[:, 12].sum())} of {int(passes)} passes; the decision (early sigma + ballot) costs "
          f"{t[:, 11].sum() / passes:.0f} cycles per pass")
    print(f"pose {pi}: {int(passes)} passes of {t.shape[0]} workgroups, {tot / passes:.0f} cycles per pass (thread 0 of each workgroup)")
    for k, nm in enumerate(names):
        c = t[:, k].sum() / passes
        extra = f"  = {c / mfma[k]:.1f} cycles per MFMA (32 = matrix pipe busy)" if mfma[k] else ""
        print(f"  {nm:34s} {c:8.0f} cycles  {100 * t[:, k].sum() / tot:5.1f} %{extra}")
