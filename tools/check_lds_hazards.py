#!/usr/bin/env python
"""Static check of the hand-counted LDS waits in the MFMA kernels (field.hip, cnn.hip).

The weight-fragment / bias reads are inline-asm ds_read_b128 whose completion hipcc does not track, so nothing but the
hand-written `s_waitcnt lgkmcnt(N)` keeps an instruction from reading a register whose data has not landed.  This
script compiles the kernels to ISA and replays every kernel linearly: each ds_read* pushes its destination registers
on an in-order queue, `s_waitcnt lgkmcnt(N)` retires all but the newest N entries (LDS returns in order; SMEM loads
are treated as queue entries too, conservatively), and any instruction that reads OR overwrites a register still in
the queue is reported.  Branch targets are handled conservatively: the queue is carried across labels as is (the
kernels' loops are straight-line bodies).

    python tools/check_lds_hazards.py            # exit status 1 if a hazard is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = {"field.hip": ["-fno-slp-vectorize"], "cnn.hip": []}
KERNELS = ("mlp_kernelILi0ELi3", "mlp_kernelILi0ELi2", "mlp_kernelILi0ELi6", "sky_kernelILi0ELi0", "sky_kernelILi0ELi1", "conv_kernelILi9ELi0ELi3", "conv_kernelILi9ELi0ELi1", "conv_kernelILi1ELi0ELi3")
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def check_kernel(name, lines):
    queue, problems, n_reads = [], [], 0
    for ln, raw in lines:
        line = raw.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        op, _, rest = line.partition(" ")
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                keep = int(m.group(1))
                queue = queue[len(queue) - keep:] if keep else []
            continue
        pending = set().union(*[q[0] for q in queue]) if queue else set()
        if op.startswith("ds_read") or op.startswith("ds_load"):
            dst, _, srcs = rest.partition(",")
            used = regs_of(srcs)
            if used & pending:
                problems.append((ln, raw.strip(), "address register pending"))
            d = regs_of(dst)
            if d & pending:
                problems.append((ln, raw.strip(), "overwrites a pending destination"))
            queue.append((d, ln))
            n_reads += 1
            continue
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            queue.append((set(), ln))   # counts on lgkmcnt, scalar destination
            continue
        if op.startswith("ds_"):        # ds_write / ds_bpermute ...: count, no vector destination tracked here
            if regs_of(rest) & pending:
                problems.append((ln, raw.strip(), "reads a pending register"))
            queue.append((regs_of(rest.split(",")[0]) if "permute" in op or "swizzle" in op else set(), ln))
            continue
        if regs_of(rest) & pending:
            problems.append((ln, raw.strip(), "touches a pending register"))
    return n_reads, problems


def check_prefetch_agprs(lines, lo=190, hi=255):
    """mlp_kernel's input prefetch lands in the physical AGPRs a[lo:hi] named in asm text.  They carry data from the
    prefetch loads (issued in front of the output layer) across the end of the pass to the v_accvgpr_read block at the
    start of the next pass -- a lifetime hipcc does not know about.  Allowed: the asm loads and reads themselves, and
    compiler-generated uses (spill space) ONLY in the part of the pass where the registers are dead: behind the last
    pass-start read and in front of the first prefetch load (the layers of the pass; the kernel's blocks are laid out in
    execution order, the clobber list of the load asm keeps hipcc's own values from living across it)."""
    pf = set(range(lo, hi + 1))
    touched, in_asm = [], False
    for ln, raw in lines:
        st = raw.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        t = raw.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        regs = set()
        for m in re.finditer(r"\ba\[(\d+):(\d+)(\+3)?\]|\ba(\d+)\b", t):
            if m.group(1):
                regs.update(range(int(m.group(1)), int(m.group(2)) + 1 + (3 if m.group(3) else 0)))
            else:
                regs.add(int(m.group(4)))
        if regs & pf:
            kind = ("load" if t.startswith("global_load") else "read" if t.startswith("v_accvgpr_read_b32") else "other") if in_asm else "foreign"
            touched.append((ln, t, kind))
    reads = [ln for ln, _, k in touched if k == "read"]
    loads = [ln for ln, _, k in touched if k == "load"]
    problems = [(ln, t, "unexpected asm instruction on a prefetch AGPR") for ln, t, k in touched if k == "other"]
    dead_from, dead_to = (max(reads), min(loads)) if reads and loads else (0, -1)
    for ln, t, k in touched:
        if k == "foreign" and not (dead_from < ln < dead_to):
            problems.append((ln, t, "prefetch AGPR touched by a foreign instruction while it carries prefetched data"))
    return len(touched), problems, sum(1 for _, _, k in touched if k == "foreign")


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src, flags in SRC.items():
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, f"-I{ROOT}/include",
                                   f"-I{ROOT}/scenedreamer_amd/csrc", "-S", "--cuda-device-only", "-o", out,
                                   os.path.join(ROOT, "scenedreamer_amd", "csrc", src)], stderr=subprocess.DEVNULL)
            text = open(out).read().split("\n")
            for k in KERNELS:
                start = next((i for i, l in enumerate(text) if l.startswith("_Z") and k in l and l.rstrip().endswith(":") or
                              (l.startswith("_Z") and k in l and ": " in l)), None)
                if start is None:
                    continue
                end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
                body = list(enumerate(text[start:end], start + 1))
                n, problems = check_kernel(k, body)
                print(f"{src}:{k}: {n} LDS reads replayed, {len(problems)} hazard(s)")
                if k.startswith("mlp_kernel"):
                    n2, p2, nf = check_prefetch_agprs(body)
                    print(f"{src}:{k}: {n2} instructions on the prefetch AGPRs a[190:255]: {nf} compiler-generated "
                          f"(spill space while the registers are dead), {len(p2)} violation(s)")
                    problems = problems + p2
                for ln, raw, why in problems[:10]:
                    print(f"    line {ln}: {why}: {raw}")
                bad += len(problems)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
