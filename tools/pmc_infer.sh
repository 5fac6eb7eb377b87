#!/bin/bash
# SQ counters per kernel for the inference workload of bench.py (two passes): tools/pmc_infer.sh, then tools/pmc_show.py gpurun_out/pmc_infer
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_infer
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o a -- python3 $R/bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline > $OUT/a.json 2> $OUT/a.log || exit 1
echo done
