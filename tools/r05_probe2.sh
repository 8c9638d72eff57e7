#!/bin/bash
# round 5, second GPU session: colour-branch skipping -- exactness tests, same-box A/B on the headline bench, per-segment cycles
label=$1
export TMPDIR=/tmp
echo "--- tests"; timeout 1200 python -m pytest tests/test_fused_gpu.py tests/test_render_gpu.py -q -m gpu -x -s > gpurun_out/${label}_fused_tests.log 2>&1; tail -5 gpurun_out/${label}_fused_tests.log; grep "colour_terms" gpurun_out/${label}_fused_tests.log
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-other-configs --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value'],2), round(d['ms_per_step'],3), [round(x,2) for x in d['frame_ms_p10_p50_p90']], 'field', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4), 'colour skipped', round(r.get('colour_passes_skipped_fraction',0),4), 'cnn', round(d['roofline_cnn']['avg_ms_in_timed_region'],3))"; }
{
for rep in 1 2; do
run SDN_COLOUR_SKIP=0
run SDN_COLOUR_SKIP=1
done
} > gpurun_out/${label}_ab_colour_skip.txt 2>&1
cat gpurun_out/${label}_ab_colour_skip.txt
echo "--- sigma stats"; timeout 300 python tools/dbg_sigma_stats.py 2>&1 | grep -vE "Warning|warn|amdgpu.ids" > gpurun_out/${label}_sigma_stats.txt; cat gpurun_out/${label}_sigma_stats.txt
echo "--- layers"; bash tools/run_layers.sh > gpurun_out/${label}_layers_cycles.txt 2>&1; cat gpurun_out/${label}_layers_cycles.txt
