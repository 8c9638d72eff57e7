mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_gpu_final.log 2>&1; echo pytest rc=$?; tail -2 gpurun_out/r02_pytest_gpu_final.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_final.json 2> gpurun_out/bench_final.err; echo bench rc=$?
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
MIOPEN_FIND_MODE=FAST timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r02/k -o k -- python $R/bench.py --steps 20 --warmup 5 --profile > $R/gpurun_out/r02_bench_under_rocprof.json 2> $R/gpurun_out/prof.log
python $R/tools/rocpd_stats.py /tmp/prof_r02/k/k_results.db 16 > $R/gpurun_out/r02_kernel_stats.md
python $R/tools/rocpd_timeline.py /tmp/prof_r02/k/k_results.db 12 2 > $R/gpurun_out/r02_timeline.md
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_r02/f -o f -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_r02/f/f_results.db 8 _kernel > $R/gpurun_out/r02_pmc_fetch.md
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_r02/w -o w -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_r02/w/w_results.db 8 _kernel > $R/gpurun_out/r02_pmc_write.md
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE -d /tmp/prof_r02/m -o m -- python $R/tools/frame_once.py fused 3 >> $R/gpurun_out/prof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_r02/m/m_results.db 8 _kernel > $R/gpurun_out/r02_pmc_mfma.md
python $R/tools/pmc_traffic.py /tmp/prof_r02/f/f_results.db /tmp/prof_r02/w/w_results.db /tmp/prof_r02/m/m_results.db "$1" > $R/gpurun_out/r02_pmc_traffic.json
tail -3 $R/gpurun_out/r02_timeline.md
