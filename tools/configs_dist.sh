#!/bin/bash
# tools/gpu_session.sh step: the multi-rank paths of BASELINE configs 4 and 5 as dry runs on ONE GPU (no 8-GPU node is available to
# the builder): N ranks over gloo (host-staged collectives), one process per rank as on a node, all on cuda:0.  What the records
# prove: the whole launch / broadcast / shard / gather machinery, the work balance of the bands, the broadcast volume -- not speed
# (the ranks share one GPU: frames/s is N-way time-sliced).
label=$1
for spec in "4 2" "5 2" "5 8"; do set -- $spec; c=$1; n=$2
  SDN_BENCH_DETAIL=gpurun_out/${label}_dist_config${c}_${n}ranks_1gpu_detail.json timeout 600 python bench.py --gpus $n --backend gloo --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-dropin --no-other-configs --no-extras \
    > gpurun_out/${label}_dist_config${c}_${n}ranks_1gpu.json 2> gpurun_out/${label}_dist_c${c}_${n}.err; echo "config $c x $n ranks rc=$?"
done
python - <<PY
import json
for c, n in ((4, 2), (5, 2), (5, 8)):
    try:
        d = json.loads(open('gpurun_out/${label}_dist_config%d_%dranks_1gpu.json' % (c, n)).read().strip().splitlines()[-1])
        print(c, n, 'ranks:', round(d['value'], 3), 'fps', round(d['ms_per_step'], 2), 'ms;', 'broadcast_s', d.get('broadcast_s'), 'band_ms', d.get('band_ms'), 'imbalance', d.get('imbalance'))
    except Exception as e:
        print(c, n, 'ERR', e)
PY
