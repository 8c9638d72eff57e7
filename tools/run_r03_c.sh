mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_gpu.py -m gpu -q -x -k "single_kernel or encode_matches or golden" 2>&1 | tail -5
for v in 0 1 0 1; do echo "== SDN_FIELD_SINGLE_KERNEL=$v"; SDN_FIELD_SINGLE_KERNEL=$v timeout 300 python tools/bench_field.py 6 2>&1 | grep pose; done
for v in 0 1; do SDN_FIELD_SINGLE_KERNEL=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r03_bench_sk$v.json 2>gpurun_out/sk$v.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r03_bench_sk$v.json').read().strip().splitlines()[-1])
print('single_kernel=$v', round(d['value'],2),'fps', round(d['ms_per_step'],2),'ms; mlp', round(d['roofline']['avg_launch_ms'],2), 'cnn in frame', round(d['roofline_cnn']['avg_ms_in_timed_region'],2), d['stage_ms'])
PY
done
