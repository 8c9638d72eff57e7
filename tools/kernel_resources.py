"""Per-kernel register / scratch / LDS usage of a compiled HIP source (reads the AMDGPU metadata of `hipcc -save-temps`).

    python tools/kernel_resources.py [scenedreamer_amd/csrc/field.hip] [extra hipcc flags...]

Used to check that an edit of field.hip did not push a hand-scheduled kernel into scratch spills."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(src, extra=()):
    sys.path.insert(0, ROOT)
    from scenedreamer_amd import build as b
    name = os.path.basename(src)
    with tempfile.TemporaryDirectory() as d:
        cmd = [b._hipcc(), *b.COMMON, *b.SOURCES.get(name, []), *extra, "-c", os.path.abspath(src), "-o", os.path.join(d, "x.o"),
               "-save-temps"]
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-3000:])
        asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
        txt = open(os.path.join(d, asm)).read()
    out = []
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)
        nm = g("name").group(1)
        dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
        out.append(dict(kernel=re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0], agpr=int(blk.split()[0]),
                        vgpr=int(g("vgpr_count").group(1)), sgpr=int(g("sgpr_count").group(1)),
                        scratch=int(g("private_segment_fixed_size").group(1)), lds=int(g("group_segment_fixed_size").group(1))))
    return out


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else os.path.join(ROOT, "scenedreamer_amd/csrc/field.hip")
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    print(f"{'kernel':60s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'lds':>7s}")
    for k in resources(src, extra):
        print(f"{k['kernel'][:60]:60s} {k['vgpr']:5d} {k['agpr']:5d} {k['sgpr']:5d} {k['scratch']:8d} {k['lds']:7d}")
