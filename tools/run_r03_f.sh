mkdir -p gpurun_out
for rep in 1 2; do for v in early late; do SDN_FRONT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r03_bench_front_$v.json 2>gpurun_out/front_$v.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_bench_front_$v.json').read().strip().splitlines()[-1])
    print('front=$v', round(d['value'],2),'fps', round(d['ms_per_step'],2),'ms; field', round(d['roofline']['avg_launch_ms'],2), 'cnn in frame', round(d['roofline_cnn']['avg_ms_in_timed_region'],2), [round(x,2) for x in d['frame_ms_p10_p50_p90']])
except Exception as e:
    print('front=$v ERR', e); print(open('gpurun_out/front_$v.err').read()[-1500:])
PY
done; done
