#!/usr/bin/env python
"""Where the unmodified loop's float frame (drop-in binding, frame evaluated once) differs from Renderer.render_frame at config 2."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from loop_helpers import run_reference_loop
from oracle import ref_harness as RH
from scenedreamer_amd import camera, dropin, synth
from scenedreamer_amd.renderer import Renderer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
scene = synth.make_scene(S, 3407)
w = synth.make_weights(0)
RH.install("hip-fast")
G, _ = RH.build_generator(w, scene)
G = G.cuda()
for p in G.parameters(): p.requires_grad_(False)
G.voxel.voxel_t = scene.voxel_t.cuda(); G.voxel.current_height_map = scene.current_height_map.cuda(); G.voxel.current_semantic_map = scene.current_semantic_map.cuda()
b = dropin.binding(G)
hw, ns, steps = [540, 960], 24, 3
R = Renderer(w, scene, "cuda"); R.set_style(synth.make_style(8888))
poses = camera.eval_camera_poses(scene, maxstep=steps, pattern=0, cam_ang=72)
for coalesce in (True, False):
    b.coalesce = coalesce
    got = []
    def full(view, dtype):
        return torch.empty(0, dtype=dtype, device="cuda").set_(view.untyped_storage()).clone()
    caps = []
    def on_frame(fr):
        got.append((fr["img"].clone(), fr["net_out"].clone()))
        v, d, r = fr["keep"]
        caps.append(dict(vid=full(v, torch.int32), d2=full(d, torch.float32), rd=full(r, torch.float32), cam=fr["cam"].clone(), sky_avg=fr["sky_avg"].clone(),
                         sky_c=fr["last"]["sky_c"].clone()))
    b.on_frame = on_frame
    with tempfile.TemporaryDirectory() as tmp:
        frames = run_reference_loop(G, tmp, hw, ns, steps, tile_size=128)
    b.on_frame = None
    for apron in ("reference", "minimal"):
        for f in range(steps):
            mine = R.render_frame(poses[f], tuple(hw), ns, mode="fused", apron=apron)
            u8 = (np.clip(((mine[0].permute(1, 2, 0).cpu().numpy()) * 0.5 + 0.5) * 255, 0, 255)).astype(np.uint8).astype(np.int32)
            d8 = np.abs(u8 - frames[f].astype(np.int32))
            msg = f"coalesce {coalesce} apron {apron} frame {f}: uint8 max diff {d8.max()} ({100 * (d8 > 0).mean():.2f} % differ)"
            if got:
                img = got[f][0][:, :, 15:-15, 15:-15]
                e = (img - mine).abs()
                idx = np.unravel_index(int(e.argmax()), e.shape)
                msg += f"; float max diff {float(e.max()):.3e} at {idx}, > 1e-3: {int((e > 1e-3).sum())} values, rows with > 1e-3: {sorted(set(torch.nonzero(e[0].amax(0) > 1e-3)[:, 0].tolist()))[:12]}"
            print(msg, flush=True)
    if got:
        for f in range(steps):
            no_r = R.render_frame(poses[f], tuple(hw), ns, mode="fused", apron="reference", cnn=False)
            no_d = got[f][1]
            e = (no_d - no_r).abs().amax(dim=-1)[0]            # [H0, W0]
            bad = torch.nonzero(e > 1e-3)
            print(f"frame {f}: net_out shapes {tuple(no_d.shape)} {tuple(no_r.shape)}; rays with |diff| > 1e-3: {bad.shape[0]}, > 1e-4: {int((e > 1e-4).sum())}; max {float(e.max()):.3e}")
            vid_r, d2_r, rd_r, cam_res = R.cast_rays(poses[f], tuple(hw))
            import voxlib
            cf, cc, _ = camera.frame_intrinsics(poses[f][3], tuple(hw), 30)
            vid_s, d2_s, rd_s = voxlib.ray_voxel_intersection_perspective(G.voxel.voxel_t, poses[f][0], poses[f][1], poses[f][2], cf, cc, cam_res, 6)
            print("   shim rvip == renderer rvip:", torch.equal(vid_s.view(-1), vid_r.view(-1).to(vid_s.dtype)), torch.equal(d2_s.view(-1).view(torch.int32), d2_r.view(-1).view(torch.int32)),
                  torch.equal(rd_s.view(-1), rd_r.view(-1)))
            for y, x in bad[:6].tolist():
                print(f"   ray ({y},{x}): vid {vid_r.view(cam_res[0], cam_res[1], -1)[y, x].tolist()} net_out[0:4] drop-in {no_d[0, y, x, :4].tolist()} renderer {no_r[0, y, x, :4].tolist()}")
    if got:
        # the same comparison with the renderer on the LOOP's own codes (G.world_encoder / G.style_net as the unmodified generator
        # evaluates them: PyTorch convolutions, a few 1e-6 away from the renderer's own evaluation of the same layers)
        from scenedreamer_amd import fused as FU
        pz = poses[0]
        vid_r, d2_r, rd_r, cam_res = R.cast_rays(pz, tuple(hw))
        n = cam_res[0] * cam_res[1]
        v, d, r_ = vid_r.view(n, 6), d2_r.view(2, n, 6), rd_r.view(n, 3)
        base = R.render_frame(pz, tuple(hw), ns, mode="fused", apron="reference", cnn=False).view(n, 64)
        sky6, avg6 = FU.sky_fused(R, r_)
        sky3, avg3 = FU.sky_fused(b.B, r_)
        ori_h = torch.as_tensor(pz[0], dtype=torch.float32)
        cnt = lambda x: (int(((x - base).abs().amax(-1) > 1e-3).sum()), int(((x - base).abs().amax(-1) > 1e-4).sum()), float((x - base).abs().max()))
        print("bisect: R consts, host origin, R sky:", cnt(FU.field_render(R, v, d, r_, ori_h, sky6, avg6, ns)))
        print("bisect: R consts, DEVICE origin, R sky:", cnt(FU.field_render(R, v, d, r_, ori_h.cuda(), sky6, avg6, ns)))
        print("bisect: R consts, host origin, 3-term sky + torch.mean:", cnt(FU.field_render(R, v, d, r_, ori_h, sky3, sky3.mean(dim=0), ns)))
        print("bisect: B consts, host origin, R sky:", cnt(FU.field_render(b.B, v, d, r_, ori_h, sky6, avg6, ns)))
        print("bisect: B consts, device origin [1,3], B sky:", cnt(FU.field_render(b.B, v, d, r_, ori_h.cuda().reshape(1, 3), sky3, sky3.mean(dim=0), ns)))
        print("bisect: drop-in frame 0:", cnt(got[0][1].view(n, 64)))
        c0 = caps[0]
        print("captured loop inputs vs renderer: vid", c0["vid"].numel(), v.numel(), torch.equal(c0["vid"][:v.numel()], v.reshape(-1).to(torch.int32)),
              "d2", torch.equal(c0["d2"][:d.numel()].view(torch.int32), d.reshape(-1).view(torch.int32)), "rd", torch.equal(c0["rd"][:r_.numel()], r_.reshape(-1)),
              "cam", c0["cam"].flatten().tolist(), ori_h.tolist(), "sky_avg diff", float((c0["sky_avg"].flatten() - avg3.flatten()).abs().max()),
              "sky_c diff vs 3-term", float((c0["sky_c"] - sky3).abs().max()))
        dd = (c0["d2"][:d.numel()] - d.reshape(-1)); nz = torch.nonzero(~torch.eq(c0["d2"][:d.numel()].view(torch.int32), d.reshape(-1).view(torch.int32))).flatten()
        print("   d2 differing elements:", nz.numel(), nz[:8].tolist(), c0["d2"][nz[:4]].tolist(), d.reshape(-1)[nz[:4]].tolist())
        print("bisect: captured arrays through field_render:", cnt(FU.field_render(b.B, c0["vid"][:v.numel()].view(n, 6), c0["d2"][:d.numel()].view(2, n, 6), c0["rd"][:r_.numel()].view(n, 3),
                                                                                  c0["cam"], c0["sky_c"], c0["sky_avg"], ns)))
        sb, sr = b.B._fused_scene, R._fused_scene
        for k in ("table3", "scales", "lut"):
            print("scene", k, "equal:", torch.equal(sb[k], sr[k]), "" if torch.equal(sb[k], sr[k]) else float((sb[k].float() - sr[k].float()).abs().max()))
        print("scene genc", sb["genc"], sr["genc"], "dims", sb["dims"], sr["dims"], "T", sb["T"], sr["T"], "grid_S", b.B.grid_S, R.grid_S, "M", b.B.M, R.M,
              b.B.sample_depth, R.sample_depth, b.B.dists_scale, R.dists_scale)
        tb, tr = b.B._fused_style, R._fused_style
        for k in ("packed", "packed_mx", "consts"):
            print("style", k, "equal:", torch.equal(tb[k], tr[k]))
        kb, kr = b.B._fused_sky, R._fused_sky
        for k in ("packed", "consts"):
            print("sky", k, "equal:", torch.equal(kb[k], kr[k]))
        from scenedreamer_amd import fused as FU
        print("precision profile", FU.precision_profile(b.B), FU.precision_profile(R), "sky terms", FU.sky_terms(b.B), FU.sky_terms(R))
        print("global_enc: loop", b.B.global_enc.flatten().tolist(), "renderer", R.global_enc.flatten().tolist())
        zl = G.style_net(torch.from_numpy(np.asarray(synth.make_style(8888))).cuda())
        print("z max diff", float((zl - R.z).abs().max()))
        R.global_enc = b.B.global_enc.clone()
        R._fused_scene = None
        R.set_style_code(zl)
        for f in range(steps):
            no_r = R.render_frame(poses[f], tuple(hw), ns, mode="fused", apron="reference", cnn=False)
            e = (got[f][1] - no_r).abs().amax(dim=-1)[0]
            img = R.render_frame(poses[f], tuple(hw), ns, mode="fused", apron="reference")
            ei = (got[f][0][:, :, 15:-15, 15:-15] - img).abs()
            print(f"frame {f}, renderer on the loop's codes: net_out rays > 1e-3: {int((e > 1e-3).sum())}, > 1e-4: {int((e > 1e-4).sum())}, max {float(e.max()):.3e}; image max {float(ei.max()):.3e}")
    print("stats", {k: v for k, v in b.stats.items() if k != 'why'}, b.B.cnn_calibration and b.B.cnn_calibration.get("terms3x3"), "R:", R.cnn_calibration["terms3x3"], R.field_gate["colour"])
