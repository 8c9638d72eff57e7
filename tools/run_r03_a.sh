mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03_pytest_gpu_a.log 2>&1; echo pytest rc=$?; tail -5 gpurun_out/r03_pytest_gpu_a.log
grep -E "max abs err|max abs diff|tile-parallel over|passed|failed|skipped" gpurun_out/r03_pytest_gpu_a.log | tail -40
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_a.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','frame_ms_p10_p50_p90','stage_ms','dtype')})
print('precision',d.get('precision'))
print('mlp',d['roofline']['avg_launch_ms'],d['roofline']['frac'],'enc',d['roofline_grid_sampler']['avg_launch_ms'],'cnn',d['roofline_cnn']['avg_ms_in_timed_region'],d['roofline_cnn']['alone_ms'], d['roofline_cnn'].get('precision_gate'))
print('cpu',d.get('cpu_baseline',{}).get('value'),d.get('cpu_baseline',{}).get('cores'))
PY
