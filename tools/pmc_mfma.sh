#!/bin/bash
# MFMA counters (busy cycles, operation counts) per kernel for the inference and the training workload of bench.py;
# summarised by tools/summarize_mfma.py into profiles/r01_mfma_util.csv.  Run on the GPU box from the repo root.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_mfma
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/infer -o infer -- python3 $R/bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline > $OUT/infer.json 2> $OUT/infer.log || exit 1
echo "infer done"
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/train -o train -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > $OUT/train.json 2> $OUT/train.log || exit 1
echo "train done"
