#!/bin/bash
# tools/gpu_session.sh step: the other BASELINE configs through bench.py's own flags (one GPU)
label=$1
for c in 3 4 5; do SDN_BENCH_DETAIL=gpurun_out/${label}_bench_config${c}_1gpu_detail.json timeout 400 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-dropin --no-other-configs > gpurun_out/${label}_bench_config${c}_1gpu.json 2> gpurun_out/${label}_c$c.err; echo "config $c rc=$?"; done
python - <<PY
import json
for c in (3,4,5):
    try:
        d=json.loads(open('gpurun_out/${label}_bench_config%d_1gpu.json' % c).read().strip().splitlines()[-1])
        print(c, round(d['value'],3), 'fps', round(d['ms_per_step'],2), 'ms', d['steps'], 'steps', d.get('band_ms'))
    except Exception as e: print(c, 'ERR', e)
PY
