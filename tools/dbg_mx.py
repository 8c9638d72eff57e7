"""Debug: colour_terms = 6 (f16 + fp6) against 3 on golden inputs, per ablation (SDN_MLP_DBG with an SDN_MLP_ABLATION build)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import golden
from scenedreamer_amd import fused, synth
from scenedreamer_amd.renderer import Renderer
g = golden("field_a.npz")
scene = synth.make_scene(256, 3407, device="cuda")
R = Renderer(synth.make_weights(0), scene, "cuda")
R.set_style_code(g["z"]); R.global_enc = torch.from_numpy(g["global_enc"]).cuda()
M = g["voxel_id"].shape[-2]
vid = torch.from_numpy(g["voxel_id"]).cuda().reshape(-1, M); n = vid.shape[0]
d2 = torch.from_numpy(g["depth2"]).cuda().reshape(2, n, M); rd = torch.from_numpy(g["raydirs"]).cuda().reshape(n, 3)
ori = torch.from_numpy(g["cam_ori"])
ns = int(g["num_samples"])
sky_c = R.sky_features(rd); sky_avg = torch.from_numpy(g["sky_avg"]).cuda().reshape(1, 64)
ref = torch.from_numpy(g["net_out"]).cuda().reshape(n, 64)
for ct in (3, 6):
    R.colour_terms = ct
    no = fused.field_fused(R, vid, d2, rd, ori, sky_c, sky_avg, ns)
    d = (no - ref).abs()
    if ct == 6:
        bad = (d.max(dim=1).values > 5e-3).nonzero().reshape(-1)
        print("   bad rays", bad.numel(), "of", n, "; ray % 32 histogram", torch.bincount(bad % 32, minlength=32).tolist())
        print("   ray % 8 hist", torch.bincount(bad % 8, minlength=8).tolist(), "first", bad[:16].tolist())
        fb = (d > 5e-3).sum(dim=0)
        print("   bad feature histogram (64 outputs)", fb.tolist())
    print("SDN_MLP_DBG", os.environ.get("SDN_MLP_DBG"), "ct", ct, "max err", d.max().item(), "mean err", d.mean().item(), "nan", int(torch.isnan(no).sum()), flush=True)

