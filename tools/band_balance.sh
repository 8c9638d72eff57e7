#!/bin/bash
# tools/gpu_session.sh step: tile-parallel work balance, band after band on one GPU (tools/band_balance.py)
label=$1
timeout 900 python tools/band_balance.py > gpurun_out/${label}_band_balance.jsonl 2> gpurun_out/${label}_band_balance.err; echo "rc=$?"
python - <<PY
import json
for ln in open('gpurun_out/${label}_band_balance.jsonl'):
    d = json.loads(ln)
    if d['world'] == 1: print('one band', d['band_ms']); continue
    print(d['world'], 'bands it', d['iteration'], 'imbalance %.3f' % d['imbalance_max_over_mean'], 'max %.1f ms' % d['frame_ms_on_a_node'], 'x%.2f' % d['speedup_vs_one_gpu'], d['band_ms'])
PY
tail -3 gpurun_out/${label}_band_balance.err
