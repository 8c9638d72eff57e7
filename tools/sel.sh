#!/bin/bash
# tools/gpu_session.sh step: a selection of tests named in tools/.sel (one pytest argument per line)
label=$1
timeout 2000 python -m pytest $(cat tools/.sel) -q -m gpu -s --durations=8 > gpurun_out/${label}_sel.log 2>&1; echo "rc=$?"; grep -v "Warning\|warn\|Rendering frame\|Saving to" gpurun_out/${label}_sel.log | tail -30
