#!/usr/bin/env python
"""Per-launch HBM traffic / MFMA busy of the hot kernels from the rocprofv3 PMC passes of tools/prof_round.sh.

    python tools/pmc_traffic.py <fetch.db> <write.db> <mfma.db> <commit> [<fetch2.db> <write2.db>] > profiles/rNN_pmc_traffic.json

(fetch2 / write2: the same two passes on the TWO-KERNEL form of the field, SDN_FIELD_SINGLE_KERNEL=0 -- the only launches of
encode_kernel, the stand-alone grid sampler, and of mlp_kernel<.., 0>.)

Corrections as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide (16 B/lane) coalesced streaming reads -> doubled for the kernels whose reads are such
streams (mlp_kernel on the feature buffer: feature + weight streams, conv_kernel: LDS-DMA streams); 32-B row gathers of the
collapsed hash table are not a calibrated pattern -> left raw: encode_kernel, and field_kernel (= mlp_kernel<.., 1>, whose HBM
fetches are those gathers -- its weight stream is L2-resident LDS-DMA); both figures (fetch_raw, fetch_x2) are kept so a reader
can apply the other convention; WRITE_SIZE raw (checks against encode_kernel's known 512 B/sample feature write).
"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (the render CNN's launches by template instance: <TAPS, DBG, TERMS, EPI>; "conv_kernel<9>" / "<1>" = the first instance that
# matches, kept for the records of earlier rounds)
KERNELS = {"field_kernel (mlp_kernel<0, 6, 1>)": ("mlp_kernel<0, 6, 1>", 1.0), "field_kernel 3-term colour (mlp_kernel<0, 3, 1>)": ("mlp_kernel<0, 3, 1>", 1.0),
           "mlp_kernel": ("mlp_kernel<0, 6, 0>", 2.0), "encode_kernel": ("encode_kernel", 1.0), "conv_kernel<9>": ("conv_kernel<9", 2.0),
           "conv_kernel<1>": ("conv_kernel<1", 2.0), "sky_kernel": ("sky_kernel", 2.0), "rvip_kernel": ("rvip_kernel", 1.0),
           "conv_kernel<9, 0, 1, 0> (conv2a/3a)": ("conv_kernel<9, 0, 1, 0>", 2.0),
           "conv_kernel<9, 0, 1, 27> (conv2b/3b)": ("conv_kernel<9, 0, 1, 27>", 2.0),
           "conv_kernel<1, 0, 3, 16> (conv1/conv4a)": ("conv_kernel<1, 0, 3, 16>", 2.0),
           "conv_kernel<1, 0, 3, 255> (conv4b + conv4)": ("conv_kernel<1, 0, 3, 255>", 2.0),
           "chain_kernel (conv4a -> conv4b -> conv4)": ("chain_kernel", 2.0), "head_kernel (rows -> conv1 -> planes)": ("head_kernel", 1.0)}


def counters(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    dur = dict((n, a) for n, a in cur.execute("select name, avg(end-start) from kernels group by name").fetchall())
    return rows, dur


def pick(rows, key, counter):
    for n, c, cnt, avg in rows:
        if key in n and c == counter:
            return avg, cnt
    return None, 0


def main(fetch_db, write_db, mfma_db, commit, fetch2_db=None, write2_db=None):
    f_rows, _ = counters(fetch_db)
    w_rows, _ = counters(write_db)
    if fetch2_db and write2_db:     # kernels that only the two-kernel form launches: appended, so the one-kernel passes win on a tie
        f_rows = list(f_rows) + list(counters(fetch2_db)[0])
        w_rows = list(w_rows) + list(counters(write2_db)[0])
    m_rows, m_dur = counters(mfma_db)
    from scenedreamer_amd import build
    out = {"commit": commit,
           "csrc_digest": build._digest(),    # renderer._profiled_traffic() refuses this profile once the kernel sources change
           "source": "rocprofv3 --kernel-trace --pmc <one counter set per pass> on tools/frame_once.py fused 3 (frames of poses 0, 2, 4 of "
                     "the headline config, field / CNN on the 4-px apron), tools/prof_round.sh; summarised by tools/pmc_traffic.py",
           "correction": "KiB -> bytes; FETCH_SIZE x2 for 16 B/lane streaming readers (mlp_kernel on the feature buffer, conv_kernel, "
                         "sky_kernel, chain_kernel) per MI355X_MICROARCH.md; raw for the 32-B table gathers of encode_kernel and of "
                         "field_kernel (mlp_kernel<.., 1>) and for rvip_kernel's 4-B / 1-B reads; WRITE_SIZE raw; every entry carries "
                         "fetch_raw and fetch_x2 besides the figure used (`fetch_corrected`, `traffic`)",
           "per_launch_bytes": {}, "mfma_busy": {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"},
           "clock_under_load_GHz": {"note": "GRBM_GUI_ACTIVE / 8 / kernel duration"}}
    for name, (key, fcorr) in KERNELS.items():
        fr, n = pick(f_rows, key, "FETCH_SIZE")
        wr, _ = pick(w_rows, key, "WRITE_SIZE")
        if fr is None or wr is None:
            continue
        out["per_launch_bytes"][name] = {"dispatches": n, "fetch_raw": fr * 1024, "fetch_x2": fr * 2048, "fetch_factor_used": fcorr,
                                         "fetch_corrected": fr * 1024 * fcorr,
                                         "write": wr * 1024, "traffic": fr * 1024 * fcorr + wr * 1024}
        busy, _ = pick(m_rows, key, "SQ_VALU_MFMA_BUSY_CYCLES")
        gui, _ = pick(m_rows, key, "GRBM_GUI_ACTIVE")
        if busy and gui:
            out["mfma_busy"][name] = busy / (1024 * gui / 8)
            d = next((v for k, v in m_dur.items() if key in k), None)
            if d:
                out["clock_under_load_GHz"][name] = gui / 8 / d
        insts, _ = pick(m_rows, key, "SQ_INSTS_MFMA")
        if insts:
            out.setdefault("mfma_instructions_per_launch", {})[name] = insts
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:7])
