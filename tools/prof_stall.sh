#!/bin/bash
# Stall / co-execution counters of the MFMA kernels (two PMC passes on tools/frame_once.py) -> gpurun_out/<tag>_pmc_stall{1,2}.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r02}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA -d /tmp/prof_s/a -o a -- python $R/tools/frame_once.py fused 3 > $O/stall.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_s/a/a_results.db 6 _kernel > $O/${T}_pmc_stall1.md
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_INSTS_SALU -d /tmp/prof_s/b -o b -- python $R/tools/frame_once.py fused 3 >> $O/stall.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_s/b/b_results.db 6 _kernel > $O/${T}_pmc_stall2.md
grep -E "mlp_kernel" $O/${T}_pmc_stall1.md $O/${T}_pmc_stall2.md | cut -c1-200
