"""oracle/_ref: the reference's OWN native sources compiled for the host CPU.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.build_ref [--force]

Compiles, from where they lie under /root/reference (nothing is copied into the repo):
  imaginaire/model_utils/gancraft/voxlib/{voxlib.cpp, ray_voxel_intersection.cu,
      positional_encoding_kernel.cu, sp_trilinear_worldcoord_kernel.cu}   -> oracle/_ref/<variant>/voxlib.so
  gridencoder/src/{bindings.cpp, gridencoder.cu}                          -> oracle/_ref/<variant>/_gridencoder.so
as ordinary torch C++ extension modules (real torch/ATen/pybind11 headers, g++), with the CUDA
language features they use supplied by oracle/ref_shim/ (blockIdx/threadIdx as thread-locals,
__global__ etc. as empty macros, atomicAdd, __half2).  Two textual changes are applied to the
source STREAM on its way into g++ (stdin, never written to disk):
  * `kernel<<<grid, block, ...>>>(args)`  ->  `sdn_ref_shim::Launcher(grid, block, ...).bind(kernel)(args)`
    (g++ cannot parse the launch token);
  * `#define is_cuda is_cpu` after the last #include, so the sources' own CHECK_CUDA(x) accepts
    the CPU tensors the host build works on.
Every arithmetic statement of the kernels is the reference's, compiled as written.

Variants (floating-point contraction decides discrete outcomes):
  nofma : -ffp-contract=off            (the convention of oracle/sdn_oracle.c and of the HIP kernels)
  fma   : -mfma -ffp-contract=fast     (a*b+c fused where the compiler sees it, like nvcc's default -fmad=true)

The modules are what `oracle/ref_harness.install("ref")` puts under the unmodified Python
reference.  /root/reference exists only in the build container; on the GPU box the prebuilt
.so files (git-ignored, shipped by gpurun) are used as they are.

stage_pytree() stages the reference's PYTHON side the same way: the modules the generator imports
(imaginaire/**.py + its label tables, gridencoder/*.py, encoding.py, activation.py, the inference
config) are packed, unchanged, into ONE archive oracle/_ref/pytree.zip -- a built artefact like the
.so files: git-ignored, shipped by gpurun, never unpacked inside the repo.  oracle/ref_harness
unpacks it into a temporary directory when /root/reference is absent, which is how
tests/test_shim_replay_gpu.py::test_unmodified_generator_on_hip_shims gets the UNMODIFIED generator
next to a GPU.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
OUT = os.path.join(HERE, "_ref")
SHIM = os.path.join(HERE, "ref_shim")
VOXLIB = os.path.join(REFERENCE, "imaginaire/model_utils/gancraft/voxlib")
GRID = os.path.join(REFERENCE, "gridencoder/src")

MODULES = {
    "voxlib": [os.path.join(VOXLIB, f) for f in ("voxlib.cpp", "ray_voxel_intersection.cu",
                                                 "positional_encoding_kernel.cu",
                                                 "sp_trilinear_worldcoord_kernel.cu")],
    "_gridencoder": [os.path.join(GRID, f) for f in ("bindings.cpp", "gridencoder.cu")],
}
VARIANTS = {
    "nofma": ["-ffp-contract=off"],
    "fma": ["-mfma", "-ffp-contract=fast"],
}
_LAUNCH = re.compile(r"([A-Za-z_]\w*(?:<[^<>;(){}]*>)?)\s*<<<(.*?)>>>", re.S)


PYTREE = os.path.join(OUT, "pytree.zip")
_PY_TOP = ("encoding.py", "activation.py", "configs/scenedreamer_inference.yaml")
_PY_DIRS = {"imaginaire": (".py", ".csv", ".json", ".yaml"), "gridencoder": (".py",)}


def available():
    return os.path.isdir(VOXLIB) and os.path.isdir(GRID)


def stage_pytree(force=False, verbose=True):
    """Pack the reference's Python tree (unchanged files, paths relative to the reference root) into oracle/_ref/pytree.zip."""
    import zipfile
    if not available():
        if os.path.exists(PYTREE):
            return PYTREE
        raise RuntimeError("oracle/_ref: /root/reference is absent and no staged pytree.zip is present")
    files = [f for f in _PY_TOP if os.path.isfile(os.path.join(REFERENCE, f))]
    for d, exts in _PY_DIRS.items():
        for base, _, names in os.walk(os.path.join(REFERENCE, d)):
            if "__pycache__" in base:
                continue
            files += [os.path.relpath(os.path.join(base, n), REFERENCE) for n in names if n.endswith(exts)]
    files.sort()
    newest = max(os.path.getmtime(os.path.join(REFERENCE, f)) for f in files)
    if not force and os.path.exists(PYTREE) and os.path.getmtime(PYTREE) >= max(newest, os.path.getmtime(__file__)):
        return PYTREE
    os.makedirs(OUT, exist_ok=True)
    tmp = PYTREE + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for f in files:
            z.write(os.path.join(REFERENCE, f), f)
    os.replace(tmp, PYTREE)
    if verbose:
        print(f"[oracle/_ref] staged {len(files)} files of the reference's Python tree -> {os.path.relpath(PYTREE, HERE)}", flush=True)
    return PYTREE


def module_path(name, variant="nofma"):
    return os.path.join(OUT, variant, name + ".so")


def built(variant="nofma"):
    return all(os.path.exists(module_path(m, variant)) for m in MODULES)


def _stream(path):
    """The two textual changes described in the module docstring."""
    src = open(path).read()
    if not path.endswith(".cu"):
        return src
    src = _LAUNCH.sub(lambda m: f"sdn_ref_shim::Launcher({m.group(2)}).bind({m.group(1)})", src)
    lines = src.split("\n")
    last = max(i for i, l in enumerate(lines) if l.lstrip().startswith("#include"))
    # keep the line numbers of everything below intact for diagnostics
    lines[last] = lines[last] + "\n#define is_cuda is_cpu\n#line %d" % (last + 2)
    return "\n".join(lines)


def _flags():
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce
    inc = [SHIM] + ce.include_paths() + [sysconfig.get_paths()["include"]]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cflags = ["-O2", "-fPIC", "-std=c++17", "-fopenmp", "-fno-fast-math", "-w",
              f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cflags += [f"-I{i}" for i in inc]
    ldflags = ["-shared", "-fopenmp", f"-L{libdir}", f"-Wl,-rpath,{libdir}", "-lc10", "-ltorch_cpu", "-ltorch",
               "-ltorch_python"]
    return cflags, ldflags


def build(force=False, variants=("nofma", "fma"), verbose=True):
    if not available():
        if all(built(v) for v in variants):
            return OUT  # GPU box: prebuilt
        raise RuntimeError("oracle/_ref: /root/reference is absent and no prebuilt modules are present")
    cflags, ldflags = _flags()
    jobs = []
    deps = [os.path.join(SHIM, "cuda_runtime.h"), os.path.join(SHIM, "cuda_fp16.h"), os.path.abspath(__file__)]
    for variant in variants:
        vdir = os.path.join(OUT, variant)
        os.makedirs(os.path.join(vdir, "obj"), exist_ok=True)
        for mod, srcs in MODULES.items():
            target = module_path(mod, variant)
            newest = max(os.path.getmtime(s) for s in srcs + deps)
            if not force and os.path.exists(target) and os.path.getmtime(target) >= newest:
                continue
            jobs.append((variant, mod, srcs, target))

    def compile_one(args):
        variant, mod, src = args
        obj = os.path.join(OUT, variant, "obj", f"{mod}__{os.path.basename(src)}.o")
        cmd = ["g++", *cflags, *VARIANTS[variant], f"-DTORCH_EXTENSION_NAME={mod}", f"-I{os.path.dirname(src)}",
               "-x", "c++", "-", "-c", "-o", obj]
        if verbose:
            print(f"[oracle/_ref] {variant}: {os.path.relpath(src, REFERENCE)}", flush=True)
        r = subprocess.run(cmd, input=_stream(src), capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed on {src} ({variant}):\n{r.stderr[-4000:]}")
        return obj

    units = [(v, m, s) for v, m, srcs, _ in jobs for s in srcs]
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(units)))) as ex:
        objs = dict(zip(units, ex.map(compile_one, units)))
    for variant, mod, srcs, target in jobs:
        gl = os.path.join(OUT, variant, "obj", f"{mod}__shim_globals.o")
        subprocess.run(["g++", *cflags, "-c", os.path.join(SHIM, "shim_globals.cc"), "-o", gl], check=True)
        cmd = ["g++", *[objs[(variant, mod, s)] for s in srcs], gl, *ldflags, "-o", target]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed for {target}:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[oracle/_ref] built {os.path.relpath(target, HERE)}", flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    stage_pytree(force="--force" in sys.argv)
