#!/bin/bash
# same-box A/B of the colour-skip variants of field_kernel: base (no skipping), v1a (workgroup-level skipping, adaptive decision),
# v1g (+ ghost waves); the exactness tests run on whatever libsdnative.so is the product build
label=$1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fused_gpu.py -q -m gpu -x -s -k "colour_branch or single_kernel or early_termination" 2>&1 | tail -8
bash tools/ab_libs.sh "python tools/bench_field.py 8" base v1a v1g 2>&1 | grep -E "==|pose" | tee gpurun_out/${label}_ab_skip_variants.txt
