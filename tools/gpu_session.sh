#!/bin/bash
# One parameterised GPU session script (replaces the per-run run_rNN_*.sh files of earlier rounds):
#   tools/gpu_session.sh <label> <step> [<step> ...]
# run on the GPU box through gpurun; every step logs under gpurun_out/<label>_<step>.*  Steps:
#   new        the tests added in round 4 (drop-in surface, precision gates)
#   core       render / fused / ops / dist tests
#   full       the whole `-m gpu` suite
#   smoke      __graft_entry__.smoke()
#   bench      bench.py headline (20 steps) without the CPU leg
#   benchfull  bench.py exactly as the driver runs it (defaults)
#   driver     bench.py as the driver runs it (--gpus 1 --steps 20 --warmup 5) + the parse check on the stdout tail
#   dropin     bench.py --only dropin
#   light      tools/prof_light.sh: test subset + kernel trace + PMC passes + the driver's bench command (evidence refresh)
#   prof       tools/prof_final.sh: full suite + driver bench + rocprofv3 kernel trace + PMC passes, summaries -> gpurun_out/<label>_*
#   any other word: executed as tools/<word>.sh if it exists
set -u
cd "$(dirname "$0")/.."
label=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${label}_build.log 2>&1 || { echo "BUILD FAILED"; tail -20 gpurun_out/${label}_build.log; }
for step in "$@"; do
  echo "=== $step ($(date +%T))"
  case $step in
    new)   timeout 1500 python -m pytest tests/test_dropin_gpu.py tests/test_precision_gates_gpu.py -q -m gpu -s --durations=10 > gpurun_out/${label}_new.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/${label}_new.log ;;
    core)  timeout 1500 python -m pytest tests/test_render_gpu.py tests/test_fused_gpu.py tests/test_ops_gpu.py tests/test_dist_gpu.py -q -m gpu --durations=10 > gpurun_out/${label}_core.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/${label}_core.log ;;
    full)  timeout 2400 python -m pytest tests -q -m gpu -s --durations=25 > gpurun_out/${label}_full.log 2>&1; echo "rc=$?"; tail -40 gpurun_out/${label}_full.log ;;
    smoke) timeout 600 python __graft_entry__.py smoke > gpurun_out/${label}_smoke.log 2>&1; echo "rc=$?"; grep "\[smoke\]" gpurun_out/${label}_smoke.log ;;
    bench) timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${label}_bench.json 2> gpurun_out/${label}_bench.err; echo "rc=$?"; tail -c 1500 gpurun_out/${label}_bench.json; tail -5 gpurun_out/${label}_bench.err ;;
    benchfull) timeout 900 python bench.py > gpurun_out/${label}_benchfull.json 2> gpurun_out/${label}_benchfull.err; echo "rc=$?"; tail -c 3000 gpurun_out/${label}_benchfull.json; tail -5 gpurun_out/${label}_benchfull.err ;;
    driver) # the driver's exact command; the check it applies: the LAST stdout line (inside an 8 KB tail) must parse as JSON
            ( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${label}_driver.out 2> gpurun_out/${label}_driver.err ); echo "rc=$?"
            cp bench_detail.json gpurun_out/${label}_bench_detail.json 2>/dev/null
            wc -l -c gpurun_out/${label}_driver.out
            tail -c 8192 gpurun_out/${label}_driver.out | tail -1 | python3 -m json.tool > gpurun_out/${label}_driver_parsed.json && echo "PARSED OK" || echo "PARSE FAILED"
            cat gpurun_out/${label}_driver.out; tail -5 gpurun_out/${label}_driver.err ;;
    dropin) timeout 900 python bench.py --only dropin > gpurun_out/${label}_dropin.json 2> gpurun_out/${label}_dropin.err; echo "rc=$?"; tail -c 3000 gpurun_out/${label}_dropin.json; tail -8 gpurun_out/${label}_dropin.err ;;
    light) bash tools/prof_light.sh "$(cat tools/.commit 2>/dev/null)" $label ;;
    prof)  bash tools/prof_final.sh "$(cat tools/.commit 2>/dev/null)" $label ;;
    *)     if [ -x tools/$step.sh ]; then tools/$step.sh $label; else echo "unknown step $step"; fi ;;
  esac
done
echo "=== done ($(date +%T))"
