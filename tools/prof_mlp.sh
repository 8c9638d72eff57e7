#!/bin/bash
# PMC breakdown of the field kernels (run on the GPU box): tools/prof_mlp.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; P=/tmp/prof_$1
run() { # name, counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $P/$n -o $n -- python $R/tools/bench_field.py 1 > $O/prof_$n.log 2>&1
  python $R/tools/rocpd_stats.py $P/$n/${n}_results.db 8 _kernel > $O/$1_tmp.md 2>/dev/null
  python $R/tools/rocpd_stats.py $P/$n/${n}_results.db 8 _kernel
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES | grep -E "mlp_kernel|encode_kernel" 
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA | grep -E "mlp_kernel|encode_kernel"
