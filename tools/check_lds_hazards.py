#!/usr/bin/env python
"""Static check of the hand-counted LDS waits in the MFMA kernels (field.hip, cnn.hip).

The weight-fragment / bias reads are inline-asm ds_read_b128 whose completion hipcc does not track, so nothing but the
hand-written `s_waitcnt lgkmcnt(N)` keeps an instruction from reading a register whose data has not landed.  This
script compiles the kernels to ISA and replays every kernel linearly: each ds_read* pushes its destination registers
on an in-order queue, `s_waitcnt lgkmcnt(N)` retires all but the newest N entries (LDS returns in order; SMEM loads
are treated as queue entries too, conservatively), and any instruction that reads OR overwrites a register still in
the queue is reported.  The replay follows the control-flow graph: every basic block is replayed from every distinct
queue that can reach it (loops until the set of (block, queue) states stops growing).

    python tools/check_lds_hazards.py            # exit status 1 if a hazard is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = {"field.hip": ["-fno-slp-vectorize"], "cnn.hip": []}
# (mlp_kernel<DBG, CT, MODE>: MODE 0 = features from encode_kernel's buffer (the only one with the AGPR input prefetch), 1 = field_kernel,
#  2 = field_kernel + per-sample outputs, 3 = LightningMLP.forward as an op; sky_kernel<DBG, SMX, PRE>)
KERNELS = ("mlp_kernelILi0ELi3ELi0E", "mlp_kernelILi0ELi2ELi0E", "mlp_kernelILi0ELi6ELi0E", "mlp_kernelILi0ELi3ELi1E", "mlp_kernelILi0ELi6ELi1E",
           "mlp_kernelILi0ELi3ELi2E", "mlp_kernelILi0ELi6ELi2E", "mlp_kernelILi0ELi3ELi3E", "mlp_kernelILi0ELi6ELi3E",
           "sky_kernelILi0ELi0ELb0E", "sky_kernelILi0ELi1ELb0E", "sky_kernelILi0ELi0ELb1E", "chain_kernelENS_11ChainParamsE", "head_kernelENS_10HeadParamsE",
           "conv_kernelILi9ELi0ELi3ELi16E", "conv_kernelILi9ELi0ELi3ELi27E", "conv_kernelILi9ELi0ELi3ELi255E",
           "conv_kernelILi9ELi0ELi1ELi0E", "conv_kernelILi9ELi0ELi1ELi16E", "conv_kernelILi9ELi0ELi1ELi27E", "conv_kernelILi9ELi0ELi1ELi255E",
           "conv_kernelILi1ELi0ELi3ELi16E", "conv_kernelILi1ELi0ELi3ELi255E")
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def run_block(lines, queue, problems, linear=False):
    """Replay one basic block from the pending-read queue `queue` (list of (frozenset of registers, line)); returns the
    queue at its end.  Hazards are appended to `problems` (a dict keyed by line, so a line is reported once).
    linear: `lines` is a whole kernel in text order -- what follows an unconditional branch is not reached by falling through
    (it is a cold block entered by a jump, e.g. the placement-only tail of early ray termination), so the reads pending at the
    branch do not carry into it; such blocks only hold compiler-generated LDS reads, which the compiler waits for itself."""
    queue = list(queue)
    for ln, raw in lines:
        line = raw.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        op, _, rest = line.partition(" ")
        if linear and op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            queue = []
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                keep = int(m.group(1))
                queue = queue[len(queue) - keep:] if keep else []
            continue
        pending = set().union(*[q[0] for q in queue]) if queue else set()
        if op.startswith("ds_read") or op.startswith("ds_load"):
            dst, _, srcs = rest.partition(",")
            if regs_of(srcs) & pending:
                problems.setdefault((ln, "address register pending"), raw.strip())
            d = frozenset(regs_of(dst))
            if d & pending:
                problems.setdefault((ln, "overwrites a pending destination"), raw.strip())
            queue.append((d, ln))
            continue
        if op.startswith("s_load") or op.startswith("s_buffer_load") or op == "s_memtime":
            queue.append((frozenset(), ln))   # counts on lgkmcnt, scalar destination
            continue
        if op.startswith("ds_"):        # ds_write / ds_bpermute ...: count, no vector destination tracked here
            if regs_of(rest) & pending:
                problems.setdefault((ln, "reads a pending register"), raw.strip())
            queue.append((frozenset(regs_of(rest.split(",")[0])) if "permute" in op or "swizzle" in op else frozenset(), ln))
            continue
        if regs_of(rest) & pending:
            problems.setdefault((ln, "touches a pending register"), raw.strip())
    return queue


def check_kernel(name, lines):
    """field.hip's kernels: one linear replay in text order (their blocks are laid out in execution order and the loops are
    straight-line bodies; the CFG walk below does not finish on 1 400 reads).  cnn.hip's: every path through the kernel's
    control-flow graph (hipcc rotates its k loop, text order is not execution order)."""
    if not name.startswith("conv_kernel"):
        problems = {}
        run_block(lines, [], problems, linear=True)
        n = sum(1 for _, raw in lines if raw.split(";")[0].strip().startswith(("ds_read", "ds_load")))
        return n, [(ln, raw, why) for (ln, why), raw in sorted(problems.items())]
    return check_kernel_cfg(name, lines)


def check_kernel_cfg(name, lines):
    """Every path through the kernel's control-flow graph: basic blocks are replayed from each distinct pending-read queue
    that reaches them (work list over (block, queue) pairs; the queues are finite sequences and a loop reaches its
    steady state after a couple of trips, so this terminates quickly)."""
    # ---- basic blocks: a block starts at a label or behind a branch
    blocks, cur, label_of = [], [], {}
    def close():
        nonlocal cur
        if cur:
            blocks.append(cur)
            cur = []
    for ln, raw in lines:
        t = raw.split(";")[0].strip()
        if t.endswith(":") and not t.startswith("."):
            continue
        if t.endswith(":"):
            close()
            label_of[t[:-1]] = len(blocks)
            continue
        cur.append((ln, raw))
        op = t.split(" ")[0] if t else ""
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm"):
            close()
    close()
    def successors(i):
        last = next((r.split(";")[0].strip() for _, r in reversed(blocks[i]) if r.split(";")[0].strip()), "")
        op, _, tgt = last.partition(" ")
        out = []
        if op == "s_endpgm":
            return out
        if op == "s_branch":
            return [label_of[tgt.strip()]]
        if op.startswith("s_cbranch"):
            out.append(label_of[tgt.strip()])
        if i + 1 < len(blocks):
            out.append(i + 1)
        return out
    problems, seen, work = {}, set(), [(0, ())]
    n_reads = sum(1 for _, raw in lines if raw.split(";")[0].strip().startswith(("ds_read", "ds_load")))
    while work:
        i, q = work.pop()
        key = (i, tuple(r for r, _ in q))
        if key in seen:
            continue
        seen.add(key)
        if len(seen) > 200000:
            problems[(0, "state explosion")] = "more than 200000 (block, queue) states"
            break
        out = tuple(run_block(blocks[i], q, problems))
        for j in successors(i):
            if j < len(blocks):   # (a label behind the last instruction: the kernel's end)
                work.append((j, out))
    return n_reads, [(ln, raw, why) for (ln, why), raw in sorted(problems.items())]


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
                    if src == "field.hip" and not k.startswith("conv_kernel"):
                        print(f"{src}:{k}: NOT FOUND in the compiled ISA (kernel renamed? update KERNELS)")
                        bad += 1
                    continue
                end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
                body = list(enumerate(text[start:end], start + 1))
                n, problems = check_kernel(k, body)
                print(f"{src}:{k}: {n} LDS reads replayed, {len(problems)} hazard(s)")
                if k.startswith("mlp_kernel") and k.endswith("ELi0E"):
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
