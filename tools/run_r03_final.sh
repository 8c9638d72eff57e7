bash tools/prof_final.sh "$1" r03
R=$GRAFT_REPO_ROOT; cd $R
for c in 3 4 5; do timeout 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_config${c}_1gpu.json 2> gpurun_out/c$c.err; echo config $c rc=$?; done
timeout 300 python bench.py --mode unfused --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r03_bench_unfused.json 2> gpurun_out/unf.err; echo unfused rc=$?
python - <<'PY'
import json
for n in ('config3_1gpu','config4_1gpu','config5_1gpu','unfused'):
    try:
        d=json.loads(open(f'gpurun_out/r03_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['value'],3), 'fps', round(d['ms_per_step'],2), 'ms', (d.get('roofline_grid_sampler') or {}).get('frac'))
    except Exception as e: print(n, 'ERR', e)
PY
