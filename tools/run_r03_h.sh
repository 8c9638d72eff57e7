# same-box A/B of library variants + bitwise tests of the current library:  bash tools/run_r03_h.sh "<variants one-kernel>" "<variants two-kernel>"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_render_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/h_tests.log
cat gpurun_out/h_tests.log
bash tools/ab_libs.sh "python tools/bench_field.py 8" $1 2>&1 | grep -E "==|pose" | tee gpurun_out/h_ab_one.log
export SDN_FIELD_SINGLE_KERNEL=0
bash tools/ab_libs.sh "python tools/bench_field.py 8" $2 2>&1 | grep -E "==|pose" | tee gpurun_out/h_ab_two.log
