#!/bin/bash
# SQ / cache counter passes over one micro-benchmark: tools/pmc_micro.sh <microbench name> [precision]
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-attn}
OUT=$R/gpurun_out/pmc_$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export KSR_PRECISION=${2:-bf16s}
timeout -s KILL 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/a -o a -- python3 $R/tools/microbench.py $B > $OUT/a.log 2>&1 || exit 1
timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/b -o b -- python3 $R/tools/microbench.py $B > $OUT/b.log 2>&1 || exit 1
timeout -s KILL 100 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/c -o c -- python3 $R/tools/microbench.py $B > $OUT/c.log 2>&1 || exit 1
echo done
