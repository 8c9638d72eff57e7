#!/bin/bash
label=$1
timeout 1500 python -m pytest tests/test_dropin_gpu.py tests/test_render_gpu.py tests/test_dist_gpu.py "tests/test_config_parity_gpu.py::test_row_bands_equal_full_frame" "tests/test_fullsize_gpu.py::test_trajectory_to_png_and_mp4_keeps_pace" -q -m gpu -s --durations=8 > gpurun_out/${label}_sel.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/${label}_sel.log
