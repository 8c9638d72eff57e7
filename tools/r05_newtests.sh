#!/bin/bash
# the tests added in round 5 (parity thin spots, CNN ladder, ADVICE fixes)
label=$1
export TMPDIR=/tmp
timeout 2400 python -m pytest -q -m gpu -x -s --durations=8 \
  "tests/test_dropin_gpu.py::test_unmodified_loop_at_the_headline_config_against_oracle_tiles" "tests/test_render_gpu.py::test_mfma_cnn_matches_torch_cnn" "tests/test_render_gpu.py::test_cnn_ladder_trades_time_for_error" \
  "tests/test_render_gpu.py::test_cnn_precision_gate_is_measured_per_style" "tests/test_precision_gates_gpu.py" \
  "tests/test_config_parity_gpu.py::test_config2_surface_like_weights_against_oracle" "tests/test_fullsize_gpu.py::test_fused_and_unfused_frames_agree_at_full_size" \
  "tests/test_config_parity_gpu.py::test_config5_every_band_seam_against_oracle" "tests/test_dist_gpu.py" \
  > gpurun_out/${label}_newtests.log 2>&1
echo "rc=$?"; grep -E "passed|failed|error" gpurun_out/${label}_newtests.log | tail -3
grep -E "CNN ladder|surface-like|fused vs GPU-placement|unmodified inference_givenstyle, 960|rung|weights seed|x[24].0:|config 3840" gpurun_out/${label}_newtests.log | cut -c1-400
