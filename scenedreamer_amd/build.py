"""Build libsdnative.so (hand-written HIP kernels for gfx950 + the C ABI).

    python -m scenedreamer_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is built IN-TREE
(scenedreamer_amd/lib/libsdnative.so) so that it travels with the source
snapshot to the GPU box; it is git-ignored.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsdnative.so")
ARCH = "gfx950"

# file -> extra flags
SOURCES = {
    "capi.hip": [],
    "rvip.hip": ["-ffp-contract=off"],  # bit-exact vs the oracle: no FMA contraction
    "posenc.hip": [],
    "gridenc.hip": [],
    # no SLP vectorisation: hipcc pairs fp32 ops into v_pk_* and pays for it with v_mov shuffles and spills in the MLP kernel
    # (SDN_FIELD_CFLAGS: extra -D switches for timing ablations, tools/ab_libs.sh -- never set for a product build)
    "field.hip": ["-fno-slp-vectorize"] + (["-DSDN_MLP_ABLATION"] if os.environ.get("SDN_MLP_ABLATION") else [])
                 + os.environ.get("SDN_FIELD_CFLAGS", "").split(),
    "cnn.hip": (["-DSDN_MLP_ABLATION"] if os.environ.get("SDN_MLP_ABLATION") else []),
    "scene.hip": [],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256()
    h.update((os.environ.get("SDN_FIELD_CFLAGS", "") + ("|ablation" if os.environ.get("SDN_MLP_ABLATION") else "")).encode())
    names = sorted(n for n in os.listdir(CSRC) if n.endswith((".hip", ".h"))) + ["../../include/sdnative.h", "../build.py"]
    for n in names:
        p = os.path.join(CSRC, n)
        if os.path.isfile(p):
            h.update(n.encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libsdnative.stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *COMMON, *SOURCES[src], "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[sdnative build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    if verbose:
        print("[sdnative build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    open(stamp, "w").write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
