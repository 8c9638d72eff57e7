#!/bin/bash
# same-box A/B of environment switches on the headline bench:  tools/gpu_session.sh <label> ab_env
label=$1
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-other-configs --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],3), [round(x,2) for x in d['frame_ms_p10_p50_p90']], round(d['roofline']['avg_launch_ms'],3), round(d['roofline_cnn']['avg_ms_in_timed_region'],3), d['precision']['gates']['cnn']['image_err_vs_fp32'])"; }
{
for rep in 1 2; do
run SDN_NOP=1
run SDN_FRONT=late
run SDN_SKY_TERMS=6
run SDN_TERM_EPS=1e-4
done
} > gpurun_out/${label}_ab_env.txt 2>&1
cat gpurun_out/${label}_ab_env.txt
