mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_shim_replay_gpu.py tests/test_dist_gpu.py -m gpu -q -s > gpurun_out/r03_pytest_gpu_b.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/r03_pytest_gpu_b.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_final.json 2> gpurun_out/bench_final.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_source'][:90], d['roofline_grid_sampler']['dram_frac_of_hbm_peak_from_profile'])
PY
