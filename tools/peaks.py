#!/usr/bin/env python
"""Re-verify the roofline peaks on this box: HBM streaming bandwidth (torch copy / read-reduce / fill on 4 GiB) and the
dense f16 MFMA rate (tools/mfma_ubench2, zero operands = best case clocks)."""
import os, subprocess, sys, time
import torch
dev = torch.device("cuda:0")
n = 1 << 30   # float32 elements = 4 GiB
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
print("| test | bytes moved | time | GB/s |"); print("|---|---|---|---|")
for name, fn, nbytes in (("copy (read + write)", lambda: b.copy_(a), 8 * n), ("fill (write)", lambda: a.fill_(1.0), 4 * n),
                         ("sum (read)", lambda: a.sum(), 4 * n)):
    s = t(fn); print(f"| {name} | {nbytes / 2**30:.0f} GiB | {s * 1e3:.2f} ms | {nbytes / s / 1e9:.0f} |")
ub = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_ubench2")
if os.path.exists(ub):
    out = subprocess.run([ub], capture_output=True, text=True).stdout
    print("\n```\n" + "\n".join(l for l in out.splitlines() if "valu/mfma=0 lds/mfma=0" in l) + "\n```")
