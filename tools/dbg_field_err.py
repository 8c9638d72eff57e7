#!/usr/bin/env python
"""Where does the fused field's largest deviation from its fp32 twin come from?  For one benchmark frame: the error
distribution of net_out, and for the worst ray the per-sample (sigma, colour) of the MFMA MLP (sdn_render_mlp on the twin's own
features) against PyTorch fp32 -- separates the MLP arithmetic from everything before it.   python tools/dbg_field_err.py [pose]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, capi, fused, ops, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer  # noqa: E402

pi = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scene = synth.make_scene(2048, 3407, device="cuda")
R = Renderer(synth.make_weights(0), scene, "cuda")
R.set_style(synth.make_style(8888))
R.set_precision(term_eps=0.0)
pose = camera.eval_camera_poses(scene, maxstep=40)[pi]
ns = 24
with torch.no_grad():
    vid, d2, rd, (H0, W0) = R.cast_rays(pose, (540, 960))
    n = H0 * W0
    vid, d2, rd = vid.view(n, R.M), d2.view(2, n, R.M), rd.view(n, 3)
    ori = torch.as_tensor(pose[0], dtype=torch.float32)
    sky32 = R.sky_features(rd)
    savg = sky32.mean(dim=0, keepdim=True)
    C = 1 << 16
    ref = torch.cat([R.field_unfused(vid[r:r + C], d2[:, r:r + C].contiguous(), rd[r:r + C], ori.cuda(), sky32[r:r + C], savg, ns, placement="kernel")
                     for r in range(0, n, C)])
    got = fused.field_fused(R, vid, d2, rd, ori, sky32, savg, ns)          # the SAME sky features: only the field differs
    err = (got - ref).abs()
    print(f"pose {pi}: net_out max abs err {float(err.max()):.3e}; values > 1e-4: {int((err > 1e-4).sum())}, > 3e-4: {int((err > 3e-4).sum())}, "
          f"> 5e-4: {int((err > 5e-4).sum())} of {err.numel()}; rms {float(err.pow(2).mean().sqrt()):.2e}")
    rerr = err.max(dim=1).values
    worst = torch.topk(rerr, 5).indices
    for ray in worst.tolist():
        # the twin's own per-sample quantities for this ray
        depth, nd, idx = ops.sample_depth_batched(d2[:, ray:ray + 1].reshape(2, 1, 1, R.M, 1).unsqueeze(0).contiguous(), ns + 1, deterministic=True,
                                                  use_box_boundaries=False, sample_depth=R.sample_depth)
        depth, nd, idx = depth.reshape(1, ns), nd.reshape(1, ns), idx.reshape(1, ns).clamp(max=R.M - 1)
        depth = torch.where(torch.isnan(depth) | torch.isinf(depth), torch.zeros_like(depth), depth)
        wc = rd[ray:ray + 1, None, :] * depth[:, :, None] + ori.cuda()[None, None, :]
        delim = torch.tensor([float(v) for v in R.voxel_dims], device="cuda")
        x5 = torch.cat([wc / delim * 2 - 1, R.global_enc[:, None, :].expand(1, ns, 2)], dim=-1)
        x5 = ((x5 + 1) / 2).reshape(-1, 5).contiguous()
        feats = torch.empty(R.grid_L, ns, 8, device="cuda")
        ops.grid_encode_forward(x5, R.w["hash_encoder.embeddings"], R.w["hash_encoder.offsets"], feats, ns, 5, 8, R.grid_L, R.grid_S, 16, False,
                                torch.empty(1, device="cuda"), 0, False)
        feats = feats.permute(1, 0, 2).reshape(ns, 128).contiguous()
        lab = torch.gather(R.lut[vid[ray:ray + 1].long()], 1, idx).reshape(-1)
        f = F.leaky_relu(F.linear(feats, R.w["render_net.fc_1.weight"]) + R.label_bias[lab], 0.2)
        for i in (2, 3, 4):
            f = F.leaky_relu(torch.addmm(R.mod[i][1], f, R.mod[i][0].t()), 0.2)
        sig32 = F.linear(f, R.w["render_net.fc_sigma.weight"], R.w["render_net.fc_sigma.bias"]).reshape(-1)
        for i in (5, 6):
            f = F.leaky_relu(torch.addmm(R.mod[i][1], f, R.mod[i][0].t()), 0.2)
        col32 = F.linear(f, R.w["render_net.fc_out_c.weight"], R.w["render_net.fc_out_c.bias"])
        # the same in float64 (what is the fp32 twin's own error?)
        d = lambda t: t.double()
        f64 = F.leaky_relu(F.linear(d(feats), d(R.w["render_net.fc_1.weight"])) + d(R.label_bias)[lab], 0.2)
        for i in (2, 3, 4):
            f64 = F.leaky_relu(torch.addmm(d(R.mod[i][1]), f64, d(R.mod[i][0]).t()), 0.2)
        sig64 = F.linear(f64, d(R.w["render_net.fc_sigma.weight"]), d(R.w["render_net.fc_sigma.bias"])).reshape(-1)
        # the MFMA MLP on the twin's features
        st = R._fused_style or fused.prepare_style(R)
        sg = torch.empty(ns, device="cuda")
        cg = torch.empty(ns, 64, device="cuda")
        capi.check(capi.lib().sdn_render_mlp(feats.data_ptr(), lab.to(torch.uint8).data_ptr(), st["packed_mx"].data_ptr(), st["consts"].data_ptr(),
                                             sg.data_ptr(), cg.data_ptr(), ns, 6, 0, None, capi.current_stream(R.dev)))
        torch.cuda.synchronize()
        fe = F.relu(sig32) * (nd.reshape(-1) * R.dists_scale)
        print(f" ray {ray} (row {ray // W0}, col {ray % W0}): net_out err {float(rerr[ray]):.2e} at feature {int(err[ray].argmax())}; boxes {vid[ray].tolist()}; "
              f"sigma32 range [{float(sig32.min()):.1f}, {float(sig32.max()):.1f}], optical depth per sample max {float(fe.max()):.2f}, total {float(fe.sum()):.2f}")
        print(f"   sigma: MFMA vs fp32 {float((sg - sig32).abs().max()):.2e}, fp32 vs fp64 {float((sig32.double() - sig64).abs().max()):.2e}, "
              f"MFMA vs fp64 {float((sg.double() - sig64).abs().max()):.2e};  colour: MFMA vs fp32 {float((cg - col32).abs().max()):.2e} (|c| max {float(col32.abs().max()):.2f})")
