#!/bin/bash
# rocprofv3 passes over the default bench (run on the GPU box from the repo root): kernel trace + stats, then
# one PMC pass each for FETCH_SIZE and WRITE_SIZE.  Summaries land under gpurun_out/prof_r01s/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r01s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# one stream for every launch: with the parameter-gradient reductions on their second stream (the default) they share
# the GPU with the data-gradient chain and the traced durations are no longer each kernel's own
export KSR_SIDE_WGRAD=0
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log || exit 1
echo "trace done"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.log || exit 1
echo "fetch done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.log || exit 1
echo "write done"
