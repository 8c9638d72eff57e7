#!/usr/bin/env python
"""Debug: which stage of the fused frame differs between two evaluations of the same pose (full-size frame)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, fused, synth  # noqa: E402
from scenedreamer_amd.renderer import Renderer  # noqa: E402


def main():
    scene = synth.make_scene(2048, 3407, device="cuda")
    R = Renderer(synth.make_weights(0), scene, "cuda")
    R.set_style(synth.make_style(8888))
    poses = camera.eval_camera_poses(scene, maxstep=40)
    hw, ns = (540, 960), 24
    for pi in (2, 11):
        pose = poses[pi]
        vid, d2, rd, (H0, W0) = R.cast_rays(pose, hw)
        n = H0 * W0
        vid, d2, rd = vid.view(n, R.M), d2.view(2, n, R.M), rd.view(n, 3)
        s1, a1 = fused.sky_fused(R, rd)
        s2, a2 = fused.sky_fused(R, rd)
        print(pi, "sky_c equal", torch.equal(s1, s2), "sky_avg equal", torch.equal(a1, a2), (a1 - a2).abs().max().item(),
              "vs f64 mean", (a1.reshape(-1) - (s1.sum(0, dtype=torch.float64) / n).float()).abs().max().item())
        win = fused.Window.crop(H0, W0, 11)
        ori = torch.as_tensor(pose[0], dtype=torch.float32)
        st = R._fused_style or fused.prepare_style(R)
        buf = fused.encode(R, vid, d2, rd, ori, ns, window=win)
        outs = []
        for use_wl in (False, True, True, False):
            no = torch.full((win.n_rays, 64), float("nan"), device="cuda")
            fused._launch_mlp(R, buf, st, s1, a1.reshape(-1), no, win.n_rays, ns, window=win, dynamic=use_wl)
            outs.append(no)
        for k in range(1, 4):
            d = (outs[0] - outs[k]).abs()
            bad = (d > 0).any(dim=1).nonzero().reshape(-1)
            print(pi, "mlp", k, "equal", torch.equal(outs[0], outs[k]), "max", d.max().item(), "rays differing", bad.numel(),
                  bad[:8].tolist(), "groups", sorted(set((bad // 32).tolist()))[:8])
        f1 = R.render_frame(pose, hw, ns, mode="fused")
        f2 = R.render_frame(pose, hw, ns, mode="fused")
        print(pi, "frame twice equal", torch.equal(f1, f2), (f1 - f2).abs().max().item())
    sel = [poses[i] for i in (2, 11, 23, 30)]
    piped = [im.clone() for im in R.render_frames(sel, hw, ns, mode="fused")]
    for pose, im in zip(sel, piped):
        f = R.render_frame(pose, hw, ns, mode="fused")
        print("piped vs seq equal", torch.equal(im, f), (im - f).abs().max().item())


if __name__ == "__main__":
    main()
