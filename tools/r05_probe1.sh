#!/bin/bash
# round 5, first GPU session: (1) does the matrix pipe's power depend on zero sample columns, (2) how many samples / passes have
# zero volume-rendering weight, (3) per-segment cycles of a pass at HEAD, (4) the new drop-in tests
label=$1
export TMPDIR=/tmp
echo "--- zero-cols"; timeout 120 tools/mlp_shape_ubench --zero-cols > gpurun_out/${label}_zero_cols.txt 2>&1; cat gpurun_out/${label}_zero_cols.txt
echo "--- sigma stats"; timeout 300 python tools/dbg_sigma_stats.py 2>&1 | grep -vE "Warning|warn|amdgpu.ids" > gpurun_out/${label}_sigma_stats.txt; cat gpurun_out/${label}_sigma_stats.txt
echo "--- layers"; bash tools/run_layers.sh > gpurun_out/${label}_layers_cycles.txt 2>&1; cat gpurun_out/${label}_layers_cycles.txt
echo "--- tests"; timeout 900 python -m pytest tests/test_dropin_gpu.py -q -m gpu -x > gpurun_out/${label}_dropin_tests.log 2>&1; tail -5 gpurun_out/${label}_dropin_tests.log
