#!/bin/bash
# Round profile on the GPU box: kernel trace + PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy), summarised to markdown.
#   bash tools/prof_round.sh <tag> [commit]     -> gpurun_out/<tag>_*.md, <tag>_pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; P=/tmp/prof_$1; T=$1
# the kernel trace is taken on bench.py itself (the command whose JSON line is reported); MIOPEN_FIND_MODE=FAST keeps MIOpen's
# one-off exhaustive search for the per-scene world-encoder convolutions (seconds of naive_conv kernels) out of the table
MIOPEN_FIND_MODE=FAST rocprofv3 --kernel-trace --stats -d $P/k -o k -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/${T}_bench_under_rocprof.json 2> $O/prof.log
python $R/tools/rocpd_stats.py $P/k/k_results.db 24 > $O/${T}_kernel_stats.md
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/f -o f -- python $R/tools/frame_once.py fused 3 >> $O/prof.log 2>&1
python $R/tools/rocpd_stats.py $P/f/f_results.db 8 _kernel > $O/${T}_pmc_fetch.md
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/w -o w -- python $R/tools/frame_once.py fused 3 >> $O/prof.log 2>&1
python $R/tools/rocpd_stats.py $P/w/w_results.db 8 _kernel > $O/${T}_pmc_write.md
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $P/m -o m -- python $R/tools/frame_once.py fused 3 >> $O/prof.log 2>&1
python $R/tools/rocpd_stats.py $P/m/m_results.db 8 _kernel > $O/${T}_pmc_mfma.md
python $R/tools/pmc_traffic.py $P/f/f_results.db $P/w/w_results.db $P/m/m_results.db "${2:-unrecorded}" > $O/${T}_pmc_traffic.json
ls -la $O
