# kernel trace of the default bench command only (no counters): summary + two-frame timeline into gpurun_out/
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
MIOPEN_FIND_MODE=FAST timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_r02/k -o k -- python $R/bench.py --steps 20 --warmup 5 --profile --no-cpu-baseline > $R/gpurun_out/r02_bench_under_rocprof.json 2> $R/gpurun_out/prof.log
python $R/tools/rocpd_stats.py /tmp/prof_r02/k/k_results.db 16 > $R/gpurun_out/r02_kernel_stats.md
python $R/tools/rocpd_timeline.py /tmp/prof_r02/k/k_results.db 12 2 > $R/gpurun_out/r02_timeline.md
tail -12 $R/gpurun_out/r02_timeline.md
