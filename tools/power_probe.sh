#!/bin/bash
# rocm-smi power / clock samples while one kernel runs in a loop.   bash tools/power_probe.sh mlp|conv|encode
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; M=/tmp/marker_$1; rm -f $M
python $GRAFT_REPO_ROOT/tools/power_probe.py $1 $M > $O/power_$1.log 2>&1 &
for i in $(seq 1 240); do [ -f $M ] && break; sleep 0.5; done
sleep 2
for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk|mclk|fclk" | sed "s/^/[$1 $i] /"; sleep 1.5; done
wait
tail -1 $O/power_$1.log
