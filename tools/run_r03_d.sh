mkdir -p gpurun_out
for rep in 1 2; do for v in 0 1; do SDN_FIELD_SINGLE_KERNEL=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r03_bench_sk$v.json 2>gpurun_out/sk$v.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_bench_sk$v.json').read().strip().splitlines()[-1])
    print('single_kernel=$v', round(d['value'],2),'fps', round(d['ms_per_step'],2),'ms; field/mlp', round(d['roofline']['avg_launch_ms'],2), 'frac', round(d['roofline']['frac'],4), 'cnn in frame', round(d['roofline_cnn']['avg_ms_in_timed_region'],2), 'enc', round(d['roofline_grid_sampler']['avg_launch_ms'],2), d['frame_ms_p10_p50_p90'])
except Exception as e:
    print('single_kernel=$v ERR', e); print(open('gpurun_out/sk$v.err').read()[-1500:])
PY
done; done
SDN_FIELD_SINGLE_KERNEL=1 timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_config_parity_gpu.py tests/test_render_gpu.py tests/test_scene_gpu.py tests/test_dist_gpu.py -m gpu -q -x -s 2>&1 | grep -E "passed|failed|max abs err vs oracle|Error" | tail -12
