#!/usr/bin/env python
"""Timeline of a rocprofv3 --kernel-trace database: the dispatches of a few frames in start order, with the queue they
ran on, start offset, duration and the overlap with the previous dispatch -- where a frame's time goes between kernels.

    python tools/rocpd_timeline.py results.db [first_mlp_launch] [n_frames] > profiles/rNN_timeline.md
A "frame" here = from the start of one mlp_kernel launch to the start of the next.
"""
import sqlite3
import sys


def short(n):
    for k in ("mlp_kernel", "sky_kernel", "encode_kernel", "rvip_kernel", "worklist_kernel", "planes_kernel", "occupancy_kernel", "chain_kernel", "head_kernel"):
        if k in n:
            return k
    if "conv_kernel" in n:
        return "conv_kernel" + n[n.index("conv_kernel") + 11:].split("(")[0]
    return n[:60]


def main(path, first=10, frames=2):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in cols), None)
    rows = cur.execute(f"select name, start, end, {qcol or 0} from kernels order by start").fetchall()
    mlp = [i for i, r in enumerate(rows) if "mlp_kernel" in r[0]]
    if len(mlp) < first + frames + 1:
        first = max(0, len(mlp) - frames - 1)
    i0, i1 = mlp[first], mlp[first + frames]
    # include the side-stream kernels that started shortly before the first mlp launch
    t0 = rows[i0][1]
    print(f"# timeline of {frames} frame(s) from mlp_kernel launch #{first} ({path}); columns: {cols}\n")
    print("| t start (ms) | dur (ms) | queue | kernel | gap to previous end on the same queue (us) |")
    print("|---|---|---|---|---|")
    last_end = {}
    for n, s, e, q in rows:
        if e < t0 or s > rows[i1][1]:
            last_end[q] = e
            continue
        gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = e
        if (e - s) < 20e3 and "at::native" in n:
            tag = "torch: " + n.split("<")[0][-40:]
        else:
            tag = short(n)
        print(f"| {(s - t0) / 1e6:8.3f} | {(e - s) / 1e6:7.3f} | {q} | `{tag}` | {gap:8.1f} |")
    span = (rows[i1][1] - t0) / 1e6
    print(f"\n{frames} frame(s) = {span:.3f} ms -> {span / frames:.3f} ms per frame\n")
    busy = {}
    for n, s, e, q in rows:
        if s >= t0 and s < rows[i1][1]:
            busy[short(n) if "at::native" not in n else "torch glue"] = busy.get(short(n) if "at::native" not in n else "torch glue", 0) + (e - s)
    print("| kernel | summed duration per frame (ms) |\n|---|---|")
    for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
        print(f"| `{k}` | {v / 1e6 / frames:.3f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10, int(sys.argv[3]) if len(sys.argv) > 3 else 2)
