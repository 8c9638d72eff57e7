#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database as markdown: per-kernel time table (--kernel-trace) and,
when counters were collected (--pmc), per-kernel counter averages.

    python tools/rocpd_stats.py results.db [top_n] [name_filter] > profiles/rNN_x.md
"""
import sqlite3
import sys


def main(path, top=30, filt=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    where = f"where name like '%{filt}%'" if filt else ""
    rows = cur.execute(f"select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       f"max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                       f"from kernels {where} group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 summary: {path}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds | scratch |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n, c, s, a, mn, mx, vg, ag, sg, lds, scr in rows[:top]:
        n = n if len(n) < 90 else n[:87] + "..."
        print(f"| `{n}` | {c} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / total:.1f} "
              f"| {vg} | {ag} | {sg} | {lds} | {scr} |")
    print(f"\ntotal kernel time {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches\n")
    try:
        w2 = f"where kernel_name like '%{filt}%'" if filt else ""
        pmc = cur.execute(f"select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection {w2} "
                          f"group by kernel_name, counter_name order by sum(value) desc").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("## counters (per dispatch average)\n")
        print("| kernel | counter | dispatches | avg per dispatch | total |")
        print("|---|---|---|---|---|")
        keep = {r[0] for r in rows[:top]}
        for n, cn, c, a, s in pmc:
            if n not in keep:
                continue
            n = n if len(n) < 70 else n[:67] + "..."
            print(f"| `{n}` | {cn} | {c} | {a:.4g} | {s:.4g} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30, sys.argv[3] if len(sys.argv) > 3 else None)
