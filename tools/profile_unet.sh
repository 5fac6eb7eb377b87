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
# HBM traffic: one PMC pass each for FETCH_SIZE and WRITE_SIZE, every kernel launched from the host (no graph replay)
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $R/bench.py --workload train-unet --no-graph --steps 2 --warmup 3 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.log || exit 1
echo "fetch done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 $R/bench.py --workload train-unet --no-graph --steps 2 --warmup 3 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.log || exit 1
echo "write done"
