#!/bin/bash
# Round-6 evidence, all on one box:  tools/gpu_session.sh <label> prof_r06
#   PMC passes first (so that the bench line of the same session finds a profile of THIS build for `roofline.traffic`), kernel
#   trace + timeline of the driver's bench command, stall passes, the driver's bench command itself, then the whole GPU suite.
T=$1
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
C="$(cat tools/.commit 2>/dev/null)"
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_$T
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/f -o f -- python $R/tools/frame_once.py fused 3 > $R/gpurun_out/prof.log 2>&1
python $R/tools/rocpd_stats.py $P/f/f_results.db 8 _kernel > $R/gpurun_out/${T}_pmc_fetch.md
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/w -o w -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
python $R/tools/rocpd_stats.py $P/w/w_results.db 8 _kernel > $R/gpurun_out/${T}_pmc_write.md
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $P/m -o m -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
python $R/tools/rocpd_stats.py $P/m/m_results.db 8 _kernel > $R/gpurun_out/${T}_pmc_mfma.md
SDN_FIELD_SINGLE_KERNEL=0 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/f2 -o f2 -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
SDN_FIELD_SINGLE_KERNEL=0 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/w2 -o w2 -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
python $R/tools/pmc_traffic.py $P/f/f_results.db $P/w/w_results.db $P/m/m_results.db "$C" $P/f2/f2_results.db $P/w2/w2_results.db > $R/gpurun_out/${T}_pmc_traffic.json
cp $R/gpurun_out/${T}_pmc_traffic.json $R/profiles/r06_pmc_traffic.json
echo "pmc done ($(date +%T))"
SDN_BENCH_DETAIL=$R/gpurun_out/${T}_bench_under_rocprof_detail.json MIOPEN_FIND_MODE=FAST timeout 300 rocprofv3 --kernel-trace --stats -d $P/k -o k -- python $R/bench.py --steps 20 --warmup 5 --profile > $R/gpurun_out/${T}_bench_under_rocprof.json 2>> $R/gpurun_out/prof.log
python $R/tools/rocpd_stats.py $P/k/k_results.db 16 > $R/gpurun_out/${T}_kernel_stats.md
python $R/tools/rocpd_timeline.py $P/k/k_results.db 12 2 > $R/gpurun_out/${T}_timeline.md
cd $R && bash tools/prof_stall.sh $T > /dev/null 2>&1
echo "traces done ($(date +%T))"
# the driver's exact command; stdout must be ONE parseable line; the full record beside it
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_final.json 2> gpurun_out/${T}_bench_final.err; echo "bench (driver command) rc=$?"
cp bench_detail.json gpurun_out/${T}_bench_detail.json 2>/dev/null
wc -l -c gpurun_out/${T}_bench_final.json
tail -c 8192 gpurun_out/${T}_bench_final.json | tail -1 | python3 -m json.tool > /dev/null && echo "PARSED OK" || echo "PARSE FAILED"
cat gpurun_out/${T}_bench_final.json
tail -3 gpurun_out/${T}_timeline.md
timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 > gpurun_out/${T}_pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_pytest_gpu_final.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?"; grep "\[smoke\]" gpurun_out/${T}_smoke.log
