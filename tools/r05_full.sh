#!/bin/bash
# full GPU suite + per-segment cycles + the driver's bench command
label=$1
export TMPDIR=/tmp
echo "--- layers"; bash tools/run_layers.sh > gpurun_out/${label}_layers_cycles.txt 2>&1; grep -v "^  torch" gpurun_out/${label}_layers_cycles.txt | tail -16
