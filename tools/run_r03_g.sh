mkdir -p gpurun_out
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_final.json 2> gpurun_out/bench_final.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['frac'], d['config']['field'], d['cpu_baseline']['value'])
PY
