#!/bin/bash
# same-box A/B: render CNN on the main stream (0) vs on its own stream beside the next frame's field kernel (1)
label=$1
for rep in 1 2; do for v in 0 1; do
  echo "== SDN_CNN_STREAM=$v (rep $rep)"
  SDN_CNN_STREAM=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-other-configs --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['frame_ms_p10_p50_p90'], d['roofline']['avg_launch_ms'], d['roofline_cnn']['avg_ms_in_timed_region'])"
done; done > gpurun_out/${label}_ab_cnn_stream.txt 2>&1
cat gpurun_out/${label}_ab_cnn_stream.txt
