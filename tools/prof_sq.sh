#!/bin/bash
# SQ wait / issue counters for the MFMA kernels (one pass per small counter group).   bash tools/prof_sq.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; P=/tmp/profsq_$1; T=$1
rocprofv3 -L > $O/${T}_counters_avail.txt 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $P/g$i -o g -- python $R/tools/frame_once.py fused 2 >> $O/profsq.log 2>&1
  python $R/tools/rocpd_stats.py $P/g$i/g_results.db 6 _kernel > $O/${T}_sq_g$i.md 2>&1
done
ls -la $O
