"""Debug / check: decode the MX image of the packed weight stream (pack_mx_kernel) and compare with the folded weights."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import fused, synth
from scenedreamer_amd.renderer import Renderer


def dec6(c):
    c = np.asarray(c, np.int64)
    s, e, m = (c >> 5) & 1, (c >> 3) & 3, c & 7
    v = np.where(e == 0, m * 0.125, (1 + m * 0.125) * np.exp2(e - 1.0))
    return np.where(s == 1, -v, v)


def kmap_hidden(s, h, e):
    return 32 * (s >> 1) + 16 * (s & 1) + (e & 3) + 8 * (e >> 2) + 4 * h


def main():
    scene = synth.make_scene(256, 3407, device="cuda")
    R = Renderer(synth.make_weights(0), scene, "cuda")
    R.set_style(synth.make_style(8888))
    st = fused.prepare_style(R)
    torch.cuda.synchronize()
    raw = st["packed_mx"].cpu().numpy().view(np.uint32)          # dwords
    L0, LH = 8 * 8 * 2 * 64, 16 * 8 * 2 * 64                     # in half8 (16-byte) units
    worst = {0: 0.0, 1: 0.0, "hh": 0.0}
    for layer, name in ((0, 5), (1, 6)):
        W = (R.mod[name][0].cpu().numpy().astype(np.float32) * np.float32(0.4)).astype(np.float32)
        Whi = W.astype(np.float16).astype(np.float32)
        Wlo = W - Whi
        base = (L0 + (3 + layer) * LH) * 4                        # dword index
        for u in range(64):
            half, kb, sub, ib0 = u // 32, (u % 32) // 8, u % 8, 4 * (u // 32)
            ub = base + u * 4 * 64 * 4
            for lane in (0, 5, 31, 32, 40, 63):
                h = lane >> 5
                if sub < 4:
                    s = 4 * kb + sub
                    for f in range(4):
                        d = raw[ub + (f * 64 + lane) * 4: ub + (f * 64 + lane) * 4 + 4].view(np.float16).astype(np.float32)
                        row = 32 * (ib0 + f) + (lane & 31)
                        ref = np.array([Whi[row, kmap_hidden(s, h, e)] for e in range(8)])
                        worst["hh"] = max(worst["hh"], float(np.abs(d - ref).max()))
                    continue
                term, iba = (sub - 4) // 2, ib0 + 2 * ((sub - 4) % 2)
                for rb in range(2):
                    f0 = raw[ub + ((2 * rb) * 64 + lane) * 4: ub + ((2 * rb) * 64 + lane) * 4 + 4]
                    f1 = raw[ub + ((2 * rb + 1) * 64 + lane) * 4: ub + ((2 * rb + 1) * 64 + lane) * 4 + 4]
                    w6 = [int(x) for x in list(f0) + list(f1[:2])]
                    big = sum(v << (32 * i) for i, v in enumerate(w6))
                    codes = [(big >> (6 * i)) & 63 for i in range(32)]
                    scale = np.exp2(float(int(f1[2]) - 127))
                    val = dec6(codes) * scale
                    row = 32 * (iba + rb) + (lane & 31)
                    src = Wlo if term == 0 else Whi
                    ref = np.array([src[row, kmap_hidden(4 * kb + i // 8, h, i % 8)] for i in range(32)])
                    if term == 0:     # an exact tie of the f16 rounding may break the other way (fused multiply on the device): |lo| = half an ulp either sign
                        val, ref = np.abs(val), np.abs(ref)
                    err = np.abs(val - ref).max() / max(np.abs(ref).max(), 1e-30)
                    worst[term] = max(worst[term], float(err))
                    if err > 0.07:
                        print("BAD layer", name, "unit", u, "lane", lane, "rb", rb, "term", term, "scale byte", int(f1[2]), "err/max", err,
                              "ref[:4]", ref[:4], "got[:4]", val[:4])
                        np.set_printoptions(linewidth=200, precision=3)
                        print("ref/scale", ref / scale)
                        print("codes", codes)
                        print("got/scale", val / scale)
                        return
    print("pack_mx: hh fragments max abs err", worst["hh"], "; fp6 fragments max err / block max: Wlo", worst[0], "Whi", worst[1], "(fp6 step: 1/30 of the block max)")


if __name__ == "__main__":
    main()
