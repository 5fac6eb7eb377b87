#!/bin/bash
# rocprofv3 kernel trace + stats over the UNet-backbone training workload (run on the GPU box from the repo root).
# Summaries land under gpurun_out/prof_unet/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_unet
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py --workload train-unet --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log || exit 1
echo "trace done"
