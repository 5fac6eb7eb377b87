#!/bin/bash
# Vector-ALU and LDS issue counters per kernel of the default training workload (bench.py --headline-only);
# summarised by tools/summarize_valu.py into profiles/r02_valu_util.csv.  Run on the GPU box from the repo root.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_valu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export KSR_SIDE_WGRAD=0
timeout -s KILL 150 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/train -o train -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > $OUT/train.json 2> $OUT/train.log || { tail -5 $OUT/train.log; exit 1; }
echo "train done"
