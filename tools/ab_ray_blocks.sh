#!/bin/bash
# same-box A/B of environment switches of the field kernel on the headline bench (tools/gpu_session.sh <label> ab_ray_blocks):
#   SDN_RAY_BLOCKS=0|1 (row-major vs 8 x 4-pixel ray groups), preceded by the tests that pin both orders to the same bits;
#   SDN_COLOUR_SKIP=0|1 works the same way (profiles/r05_ab_colour_skip.txt was taken like this)
label=$1
export TMPDIR=/tmp
echo "--- tests"; timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_render_gpu.py tests/test_fullsize_gpu.py tests/test_dist_gpu.py tests/test_dropin_gpu.py -q -m gpu > gpurun_out/${label}_tests.log 2>&1; tail -4 gpurun_out/${label}_tests.log
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-other-configs --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value'],2), round(d['ms_per_step'],3), [round(x,2) for x in d['frame_ms_p10_p50_p90']], 'field', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4), 'colour skipped', round(r.get('colour_passes_skipped_fraction',0),4), 'cnn', round(d['roofline_cnn']['avg_ms_in_timed_region'],3))"; }
{
for rep in 1; do
run SDN_RAY_BLOCKS=0
run SDN_RAY_BLOCKS=1
done
true
} > gpurun_out/${label}_ab_ray_blocks.txt 2>&1
cat gpurun_out/${label}_ab_ray_blocks.txt
