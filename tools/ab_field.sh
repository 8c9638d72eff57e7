#!/bin/bash
# same-box A/B of the field kernel: library variants (tools/build_variant.sh) and early ray termination
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
label=${1:-ab}
{
bash tools/ab_libs.sh "python tools/bench_field.py 10" cur nognd
echo "== cur, SDN_TERM_EPS=5e-5"; SDN_TERM_EPS=5e-5 python tools/bench_field.py 10 2>&1 | grep pose
echo "== cur, SDN_TERM_EPS=1e-4"; SDN_TERM_EPS=1e-4 python tools/bench_field.py 10 2>&1 | grep pose
} > gpurun_out/${label}_ab_field.txt 2>&1
cat gpurun_out/${label}_ab_field.txt
