"""Per-layer precision study of the f16-split MFMA products (CPU emulation).

Every product W.X of the field MLP / sky MLP / render CNN is evaluated on the GPU as a sum of f16 x f16 MFMA
terms with f32 accumulation:  hh = Whi.Xhi,  lh = Wlo.Xhi,  hl = Whi.Xlo  (hi = f16 round-toward-zero,
lo = f16(x - hi)).  This script emulates any per-layer subset of the terms with fp32 matmuls on the CPU
(f16 x f16 products are exact in f32) inside the literal oracle (oracle/field_ref.py) and reports the max abs
error of net_out and of the image against the fp32 oracle, so the term sets can be chosen against the 1e-3
radiance bound before a kernel is touched.

    python tools/precision_study.py [--hw 48 64] [--ns 24]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import field_ref as FR  # noqa: E402


ROUND = {"x": "rtz", "w": "rtz"}     # how hi is rounded: rtz (v_cvt_pkrtz_f16_f32) or rtn (v_cvt_pk_f16_f32, gfx950)


def split(x, mode="rtz"):
    h = x.to(torch.float16)
    if mode == "rtz":
        over = h.float().abs() > x.abs()
        hv = h.view(torch.int16)
        hv = torch.where(over, hv - 1, hv)          # sign-magnitude: one step toward zero
        h = hv.view(torch.float16)
    hi = h.float()
    lo = (x - hi).to(torch.float16).float()
    return hi, lo


def mm(x, W, terms):
    """x [..., K], W [N, K] -> x W^T with the given subset of split terms ('f32' = exact)."""
    if terms == "f32":
        return x @ W.t()
    xh, xl = split(x, ROUND["x"])
    Wh, Wl = split(W, ROUND["w"])
    y = xh @ Wh.t()
    if "lh" in terms:
        y = y + xh @ Wl.t()
    if "hl" in terms:
        y = y + xl @ Wh.t()
    if "ll" in terms:
        y = y + xl @ Wl.t()
    return y


T3, LH, HL, T1 = ("hh", "lh", "hl"), ("hh", "lh"), ("hh", "hl"), ("hh",)


def make_render_mlp(cfg):
    def render_mlp(w, x, z, m, dtype=torch.float32):
        act = lambda v: F.leaky_relu(v, 0.2)
        Tn = lambda n: FR.T(w, n, dtype)
        zz = z.reshape(1, -1)

        def modlin(name, f):
            alpha = zz @ Tn(name + ".weight_alpha").t() + Tn(name + ".bias_alpha")
            beta = zz @ Tn(name + ".weight_beta").t() + Tn(name + ".bias_beta")
            Wp = Tn(name + ".weight") * alpha          # per-style constant W' = W (.) alpha
            return mm(f, Wp, cfg[name.split(".")[-1]]) + beta
        f = mm(x, Tn("render_net.fc_1.weight"), cfg["fc_1"]) + Tn("render_net.fc_1.bias")
        f = f + F.linear(m, Tn("render_net.fc_m_a.weight"))      # label bias row (exact f32 add in the kernel)
        f = act(f)
        for n in ("fc_2", "fc_3", "fc_4"):
            f = act(modlin("render_net." + n, f))
        sigma = mm(f, Tn("render_net.fc_sigma.weight"), cfg["fc_sigma"]) + Tn("render_net.fc_sigma.bias")
        for n in ("fc_5", "fc_6"):
            f = act(modlin("render_net." + n, f))
        c = mm(f, Tn("render_net.fc_out_c.weight"), cfg["fc_out_c"]) + Tn("render_net.fc_out_c.bias")
        return sigma, c
    return render_mlp


def render_cnn(w, net_out, z, cfg):
    act = lambda v: F.leaky_relu(v, 0.2)
    Tn = lambda n: FR.T(w, n)
    x = torch.as_tensor(net_out).permute(0, 3, 1, 2).contiguous()
    z = torch.as_tensor(z)
    cond = F.linear(z, Tn("denoiser.fc_z_cond.weight"), Tn("denoiser.fc_z_cond.bias"))
    adapt = torch.chunk(cond, 4, dim=-1)
    mod = lambda v, a, b: v * (a[..., None, None] + 1) + b[..., None, None]

    def cv(v, n, p):
        W = Tn(f"denoiser.{n}.weight")
        b = Tn(f"denoiser.{n}.bias") if f"denoiser.{n}.bias" in w else None
        terms = cfg[n]
        if terms == "f32":
            return F.conv2d(v, W, b, padding=p)
        vh, vl = split(v, ROUND["x"])
        Wh, Wl = split(W, ROUND["w"])
        y = F.conv2d(vh, Wh, None, padding=p)
        if "lh" in terms:
            y = y + F.conv2d(vh, Wl, None, padding=p)
        if "hl" in terms:
            y = y + F.conv2d(vl, Wh, None, padding=p)
        return y if b is None else y + b[None, :, None, None]
    y = act(cv(x, "conv1", 0))
    y = y + cv(act(cv(y, "conv2a", 1)), "conv2b", 1)
    y = act(mod(y, adapt[0], adapt[1]))
    y = y + cv(act(cv(y, "conv3a", 1)), "conv3b", 1)
    y = act(mod(y, adapt[2], adapt[3]))
    y = y + cv(act(cv(y, "conv4a", 0)), "conv4b", 0)
    y = act(y)
    y = cv(y, "conv4", 0)
    return torch.tanh(y)


MLP_LAYERS = ["fc_1", "fc_2", "fc_3", "fc_4", "fc_sigma", "fc_5", "fc_6", "fc_out_c"]
CNN_LAYERS = ["conv1", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "conv4"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, nargs=2, default=[40, 56])
    ap.add_argument("--ns", type=int, default=24)
    ap.add_argument("--poses", type=int, nargs="*", default=[1, 3, 6])
    args = ap.parse_args()
    torch.set_num_threads(8)
    from oracle import oracle as O
    from scenedreamer_amd import camera, synth
    from scenedreamer_amd.renderer import load_label_lut
    lut = load_label_lut()["lut"]
    scene = synth.make_scene(256, 3407)
    w = synth.make_weights(0)
    z = FR.style_mlp(w, synth.make_style(8888))
    genc = FR.world_encoder(w, scene.current_height_map, scene.current_semantic_map)
    poses = camera.eval_camera_poses(scene, maxstep=8)
    hw = tuple(args.hw)
    cases = []
    for pi in args.poses:
        p = poses[pi]
        f, c, cam_res = camera.frame_intrinsics(p[3], hw, 30)
        vid, d2, rd = O.rvip(scene.voxel_t.numpy(), p[0].numpy(), p[1].numpy(), p[2].numpy(), f, c, cam_res, 6)
        vid, d2, rd = torch.from_numpy(vid)[None], torch.from_numpy(d2)[None], torch.from_numpy(rd)[None]
        sky_avg = FR.sky_average(w, rd, z)
        cases.append((vid, d2, rd, p[0][None], sky_avg))
    orig = FR.render_mlp

    def field(cfg):
        FR.render_mlp = make_render_mlp(cfg) if cfg is not None else orig
        try:
            return [FR.forward_perpix(w, lut, scene.voxel_t.shape, vid, d2, rd, ori, z, genc, args.ns, sky_avg=sa)
                    for vid, d2, rd, ori, sa in cases]
        finally:
            FR.render_mlp = orig
    ref_no = field(None)
    ref_img = [FR.render_cnn(w, no, z) for no in ref_no]
    cnn3 = {n: T3 for n in CNN_LAYERS}

    def report(tag, mlp_cfg, cnn_cfg):
        nos = field(mlp_cfg) if mlp_cfg is not None else ref_no
        e_no = max(float((a - b).abs().max()) for a, b in zip(nos, ref_no))
        imgs = [render_cnn(w, no, z, cnn_cfg) for no in nos]
        e_img = max(float((a - b).abs().max()) for a, b in zip(imgs, ref_img))
        nmf = sum({3: 3, 2: 2, 1: 1}[len(v)] * k for v, k in zip([mlp_cfg[n] for n in MLP_LAYERS], MLP_MACS)) / (3 * sum(MLP_MACS)) if mlp_cfg else 1.0
        ncf = sum(len(cnn_cfg[n]) * k for n, k in zip(CNN_LAYERS, CNN_MACS)) / (3 * sum(CNN_MACS))
        print(f"{tag:58s} net_out err {e_no:.2e}   image err {e_img:.2e}   MFMAs: mlp x{nmf:.3f} cnn x{ncf:.3f}", flush=True)
    MLP_MACS = [128 * 256, 65536, 65536, 65536, 256, 65536, 65536, 256 * 64]
    global CNN_MACS
    CNN_MACS = [64 * 256, 9 * 65536, 9 * 65536, 9 * 65536, 9 * 65536, 65536, 65536, 256 * 3]
    m3 = {n: T3 for n in MLP_LAYERS}
    report("all 3-term (current kernels)", m3, cnn3)
    if os.environ.get("ROUND"):
        ROUND["x"], ROUND["w"] = os.environ["ROUND"].split(",")
        print("hi rounding:", ROUND)
        report("all 3-term", m3, cnn3)
    for name, t in (("lh (Whi.Xhi + Wlo.Xhi)", LH), ("hl (Whi.Xhi + Whi.Xlo)", HL), ("1-term", T1)):
        cfg = dict(m3, fc_5=t, fc_6=t, fc_out_c=t)
        report(f"mlp colour layers fc_5/fc_6/fc_out_c {name}", cfg, cnn3)
    for name, t in (("lh", LH), ("hl", HL)):
        report(f"mlp fc_out_c only {name}", dict(m3, fc_out_c=t), cnn3)
        report(f"mlp fc_1 only {name}", dict(m3, fc_1=t), cnn3)
        report(f"mlp fc_4 only {name}", dict(m3, fc_4=t), cnn3)
        report(f"mlp all layers {name}", {n: t for n in MLP_LAYERS}, cnn3)
    for name, t in (("lh", LH), ("hl", HL), ("1-term", T1)):
        report(f"cnn all layers {name} (mlp 3-term)", m3, {n: t for n in CNN_LAYERS})
    for name, t in (("lh", LH), ("hl", HL)):
        report(f"cnn all {name} + mlp colour layers {name}", dict(m3, fc_5=t, fc_6=t, fc_out_c=t), {n: t for n in CNN_LAYERS})
    report("cnn 3x3 lh, 1x1 3-term", m3, dict(cnn3, conv2a=LH, conv2b=LH, conv3a=LH, conv3b=LH))
    report("cnn 3x3 1-term, 1x1 3-term", m3, dict(cnn3, conv2a=T1, conv2b=T1, conv3a=T1, conv3b=T1))
    report("cnn conv2a/conv3a 1-term, conv2b/conv3b lh", m3, dict({n: LH for n in CNN_LAYERS}, conv2a=T1, conv3a=T1))


if __name__ == "__main__":
    main()
