#!/bin/bash
# Evidence refresh after a small kernel change (the full set: tools/prof_final.sh):  bash tools/prof_light.sh <commit> [tag]
#   a test subset, kernel trace of the driver's bench command, the three PMC passes behind `roofline.traffic`, then the bench line
#   itself (it finds the fresh PMC profile).
T=${2:-r03}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_render_gpu.py -m gpu -x -q > gpurun_out/${T}_pytest_gpu_light.log 2>&1; echo pytest rc=$?; tail -1 gpurun_out/${T}_pytest_gpu_light.log
cd /tmp && export TMPDIR=/tmp
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
cd $R
cp gpurun_out/${T}_pmc_traffic.json profiles/${T}_pmc_traffic.json
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_final.json 2> gpurun_out/bench_final.err; echo bench rc=$?
python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_bench_final.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','stage_ms')}, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_cnn']['avg_ms_in_timed_region'], d['roofline_cnn']['alone_ms'])
PY
grep -E "sky_kernel" gpurun_out/${T}_kernel_stats.md | head -2
