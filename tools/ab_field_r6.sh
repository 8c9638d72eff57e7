#!/bin/bash
# tools/gpu_session.sh step: same-box A/B of field-kernel library variants (scenedreamer_amd/lib/variants/*.so, tools/build_variant.sh)
label=$1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/ab_libs.sh "python tools/bench_field.py 10" $(cat tools/.variants 2>/dev/null || echo base flags) > gpurun_out/${label}_ab_field.txt 2>&1
grep -E "==|pose" gpurun_out/${label}_ab_field.txt
