#!/usr/bin/env python
"""How many samples of the headline frame have volume-rendering weight EXACTLY zero (relu(sigma) * dist == 0, mc_utils.py:154-161)
and how they cluster: per sample, per wave pass (8 rays x 4 samples = the 32 columns of one wave's MFMAs) and per workgroup pass
(32 rays x 4 samples).  A pass whose columns are all zero-weight does not need the colour branch fc_5 / fc_6 / fc_out_c at all
("measure, then build if > 5 %")."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, fused, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer  # noqa: E402

dev = torch.device("cuda:0")
scene = synth.make_scene(2048, 3407, device=dev)
R = Renderer(synth.make_weights(0), scene, dev)
R.set_style(synth.make_style(8888))
poses = camera.eval_camera_poses(scene, maxstep=40)
ns = 24
for pi in (0, 10, 13, 27):
    pose = poses[pi]
    with torch.no_grad():
        vid, d2, rd, cam_res = R.cast_rays(pose, (540, 960))
        n = cam_res[0] * cam_res[1]
        vid, d2, rd = vid.view(n, R.M), d2.view(2, n, R.M), rd.view(n, 3)
        sky_c, sky_avg = fused.sky_fused(R, rd)
        aux = {"sigma": None, "weights": None}
        fused.field_render(R, vid, d2, rd, torch.as_tensor(pose[0], dtype=torch.float32), sky_c, sky_avg, ns, aux=aux)
        sig, w = aux["sigma"], aux["weights"]
        hit = vid[:, 0] != 0
        pad = (-n) % 32
        z = (sig <= 0) | ~hit[:, None]                       # zero weight: density clipped, or a ray that hit nothing
        zp = torch.nn.functional.pad(z, (0, 0, 0, pad), value=True)
        hp = torch.nn.functional.pad(hit, (0, pad))
        grp_hit = hp.view(-1, 32).any(dim=1)                  # groups the kernel evaluates at all
        zg = zp.view(-1, 32, ns // 4, 4)[grp_hit]             # [groups, 32 rays, 6 passes, 4 samples]
        samples = zg.numel()
        wave = zg.view(-1, 4, 8, ns // 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 32).all(dim=1)     # (group, wave, pass) x 32 columns
        grp = zg.permute(0, 2, 1, 3).reshape(-1, 128).all(dim=1)
        tiny = ((w < 1e-6) | ~hit[:, None])
        tp = torch.nn.functional.pad(tiny, (0, 0, 0, pad), value=True).view(-1, 32, ns // 4, 4)[grp_hit]
        twave = tp.view(-1, 4, 8, ns // 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 32).all(dim=1)
        # would a 2-D pixel footprint of the 32-ray group (instead of 32 consecutive pixels of a row) be empty more often?
        H0, W0 = cam_res
        zi = z.view(H0, W0, ns // 4, 4).all(dim=-1)                 # [H0, W0, passes]: the ray's 4 samples of the pass are all zero-weight
        hi = hit.view(H0, W0)
        shapes = {}
        for bw, bh in ((32, 1), (16, 2), (8, 4), (4, 8)):
            Hc, Wc = H0 // bh * bh, W0 // bw * bw
            blk = zi[:Hc, :Wc].view(Hc // bh, bh, Wc // bw, bw, ns // 4).permute(0, 2, 4, 1, 3).reshape(-1, ns // 4, bh * bw)
            bhit = hi[:Hc, :Wc].view(Hc // bh, bh, Wc // bw, bw).permute(0, 2, 1, 3).reshape(-1, bh * bw).any(dim=1)
            shapes[f"{bw}x{bh}"] = (round(float(blk[bhit].all(dim=-1).float().mean()), 4), round(float(bhit.float().mean()), 4))
        print(f"pose {pi}: group footprint (w x h) -> (fraction of the evaluated groups' passes that are all zero-weight, fraction of groups evaluated): {shapes}")
        # how the all-zero passes of a 32-ray group follow each other along the ray (would an adaptive "test only when likely" pay?)
        gz = zg.permute(0, 2, 1, 3).reshape(zg.shape[0], ns // 4, 128)          # [groups, passes, 128 samples]
        allz = gz.all(dim=-1).float()                                            # [groups, passes]
        frac = gz.float().mean(dim=-1)
        per_pass = [round(float(allz[:, c].mean()), 3) for c in range(ns // 4)]
        prev1 = allz[:, :-1].reshape(-1) > 0.5
        nxt = allz[:, 1:].reshape(-1)
        fprev = frac[:, :-1].reshape(-1)
        cond = {"P(next all-zero | this all-zero)": float(nxt[prev1].mean()), "P(next | this not)": float(nxt[~prev1].mean())}
        for lo, hi in ((0.0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 0.999)):
            m = (fprev >= lo) & (fprev < hi) & ~prev1
            cond[f"P(next | zero fraction of this in [{lo},{hi}))"] = (round(float(nxt[m].mean()), 3), round(float(m.float().mean()), 3))
        print(f"pose {pi}: all-zero probability by pass index {per_pass}; {cond}")
        # ray level: how many hitting rays are empty over their WHOLE length, and does the block id of the first hit tell?
        ray_empty = z.all(dim=1) & hit
        first = vid[:, 0].long()
        lab = R._fused_scene["lut"].long()[first.clamp(0, 1023)] if getattr(R, "_fused_scene", None) else first
        by = {}
        for l in torch.unique(lab[hit]).tolist():
            m = hit & (lab == l)
            by[int(l)] = (round(float(m.float().sum() / hit.float().sum()), 3), round(float(ray_empty[m].float().mean()), 3))
        print(f"pose {pi}: hitting rays that are empty along their whole length {float(ray_empty.float().sum() / hit.float().sum()):.3f}; by label of the "
              f"first hit {{label: (share of hitting rays, empty fraction)}}: {by}")
        # what regrouping the rays would give (rays are independent: any 32 of them can share a workgroup).  Stable orders, raster
        # order inside each bucket; the figure is the fraction of (group, pass) pairs of groups with a hit that are all zero-weight
        zr = z.view(n, ns // 4, 4).all(dim=-1)                       # [rays, passes]
        def grouped(order, skip_first=0):
            zo = zr[order]
            ho = hit[order]
            padn = (-order.numel()) % 32
            zo = torch.nn.functional.pad(zo, (0, 0, 0, padn), value=True).view(-1, 32, ns // 4)
            ho = torch.nn.functional.pad(ho, (0, padn)).view(-1, 32).any(dim=1)
            return round(float(zo[ho][:, :, skip_first:].all(dim=1).float().mean()), 4), round(float(ho.float().mean()), 4)
        raster = torch.arange(n, device=dev)
        key_label = torch.where(hit, lab, torch.full_like(lab, 99))
        key_p0 = torch.where(hit, zr[:, 0].long(), torch.full_like(lab, 9))
        key_empty = torch.where(hit, ray_empty.long(), torch.full_like(lab, 9))
        res = {"raster": grouped(raster), "hit first": grouped(torch.sort((~hit).long(), stable=True).indices),
               "by first-hit label": grouped(torch.sort(key_label, stable=True).indices),
               "by pass-0 emptiness (passes 1..5 only)": grouped(torch.sort(key_p0, stable=True).indices, 1),
               "by whole-ray emptiness (oracle)": grouped(torch.sort(key_empty, stable=True).indices)}
        print(f"pose {pi}: regrouping -> (all-zero fraction of the evaluated group passes, fraction of groups evaluated): {res}")
        print(f"pose {pi}: evaluated samples {samples}; zero-weight samples {float(zg.float().mean()):.3f}; wave passes with all 32 columns zero "
              f"{float(wave.float().mean()):.4f}; workgroup passes all zero {float(grp.float().mean()):.4f}; "
              f"weight < 1e-6: samples {float(tp.float().mean()):.3f}, wave passes {float(twave.float().mean()):.4f}; "
              f"sigma: mean {float(sig[hit].mean()):.3f} std {float(sig[hit].std()):.3f}")
