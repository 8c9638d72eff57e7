"""Where a kernel of field.hip touches scratch memory (register spills): line numbers inside the kernel's ISA + the nearest labels.

    python tools/spills.py [mangled-name-fragment, default mlp_kernelILi0ELi6ELi1E = field_kernel]

tools/kernel_resources.py says HOW MUCH scratch a kernel has; this says where the spills sit (a spill inside a unit of a layer costs
matrix time, one between passes does not)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scenedreamer_amd import build as b  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("-")]
dump = [a[2:] for a in sys.argv[1:] if a.startswith("-o")]   # -o<file>: write the kernel's ISA there
verbose = "-v" in sys.argv
frag = args[0] if args else "mlp_kernelILi0ELi6ELi1E"
src = os.path.join(ROOT, "scenedreamer_amd/csrc/field.hip")
with tempfile.TemporaryDirectory() as d:
    cmd = [b._hipcc(), *b.COMMON, *b.SOURCES["field.hip"], "-c", src, "-o", os.path.join(d, "x.o"), "-save-temps"]
    r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-3000:])
    txt = open(os.path.join(d, [f for f in os.listdir(d) if f.endswith("gfx950.s")][0])).read()
m = re.search(rf"^(_Z\S*{re.escape(frag)}\S*):[^\n]*\n", txt, re.M)
if m is None:
    raise SystemExit("no such kernel; have: " + ", ".join(sorted(set(re.findall(r"^(_Z\S*kernel\S*):", txt, re.M)))))
body = txt[m.end():txt.index("s_endpgm", m.end())].split("\n")
print(m.group(1), len(body), "lines")
if dump:
    open(dump[0], "w").write("\n".join(f"{i:6d} {l}" for i, l in enumerate(body)))
mf = [i for i, l in enumerate(body) if "v_mfma" in l]
ops = [l for l in body if "scratch_" in l]
print(len(ops), "scratch instructions:", sum("load" in l for l in ops), "loads,", sum("store" in l for l in ops), "stores;",
      sum(1 for i, l in enumerate(body) if "scratch_" in l and mf and mf[0] < i < mf[-1]), "of them between the first and the last MFMA")
for i, l in enumerate(body):
    if verbose and "scratch_" in l:
        prev = max([k for k in mf if k < i], default=-1)
        nxt = min([k for k in mf if k > i], default=-1)
        print(f"{i:6d}  {l.strip():60s} MFMAs before: {sum(1 for k in mf if k < i):5d}; distance to previous / next MFMA: {i - prev} / {nxt - i}")
