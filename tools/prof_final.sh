#!/bin/bash
# Final evidence of a round, all on one box:  bash tools/prof_final.sh <commit> [tag]   (tag default r03)
#   GPU test suite, the driver's bench command, rocprofv3 kernel trace of the same command (--profile), three PMC passes
#   (FETCH_SIZE / WRITE_SIZE / MFMA + clock) and two stall passes on tools/frame_once.py; summaries -> gpurun_out/<tag>_*
T=${2:-r03}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/${T}_pytest_gpu_final.log 2>&1; echo pytest rc=$?; tail -2 gpurun_out/${T}_pytest_gpu_final.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_final.json 2> gpurun_out/bench_final.err; echo bench rc=$?
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$T
MIOPEN_FIND_MODE=FAST timeout 300 rocprofv3 --kernel-trace --stats -d $P/k -o k -- python $R/bench.py --steps 20 --warmup 5 --profile > $R/gpurun_out/${T}_bench_under_rocprof.json 2> $R/gpurun_out/prof.log
python $R/tools/rocpd_stats.py $P/k/k_results.db 16 > $R/gpurun_out/${T}_kernel_stats.md
python $R/tools/rocpd_timeline.py $P/k/k_results.db 12 2 > $R/gpurun_out/${T}_timeline.md
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/f -o f -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
python $R/tools/rocpd_stats.py $P/f/f_results.db 8 _kernel > $R/gpurun_out/${T}_pmc_fetch.md
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/w -o w -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
python $R/tools/rocpd_stats.py $P/w/w_results.db 8 _kernel > $R/gpurun_out/${T}_pmc_write.md
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $P/m -o m -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
python $R/tools/rocpd_stats.py $P/m/m_results.db 8 _kernel > $R/gpurun_out/${T}_pmc_mfma.md
# the stand-alone grid sampler (encode_kernel) only exists in the two-kernel form of the field: two more passes for its DRAM bytes
SDN_FIELD_SINGLE_KERNEL=0 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/f2 -o f2 -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
SDN_FIELD_SINGLE_KERNEL=0 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/w2 -o w2 -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
python $R/tools/pmc_traffic.py $P/f/f_results.db $P/w/w_results.db $P/m/m_results.db "$1" $P/f2/f2_results.db $P/w2/w2_results.db > $R/gpurun_out/${T}_pmc_traffic.json
cd $R && bash tools/prof_stall.sh $T > /dev/null 2>&1
tail -3 $R/gpurun_out/${T}_timeline.md
python - <<PY
import json
d=json.loads(open('$R/gpurun_out/${T}_bench_final.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','frame_ms_p10_p50_p90','stage_ms')})
print('precision',d.get('precision',{}).get('max_abs_err'),'cpu',d.get('cpu_baseline',{}).get('value'),d.get('cpu_baseline',{}).get('kind'),d.get('cpu_baseline',{}).get('cores'))
print('mlp',d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['roofline'].get('traffic'),'enc',d['roofline_grid_sampler']['avg_launch_ms'],'cnn',d['roofline_cnn']['avg_ms_in_timed_region'],d['roofline_cnn']['alone_ms'])
PY
