#!/bin/bash
# rocprofv3 passes over bench.py workloads (run on the GPU box from the repo root): kernel trace + stats, then one PMC pass each
# for FETCH_SIZE and WRITE_SIZE.  Usage: tools/profile_r03.sh [workload ...]   (default: train infer train-attn)
# Output: gpurun_out/prof_r03/<workload>/.  The training passes hold the headline step only (--headline-only): 5 + 2 + 5 = 12
# steps in the trace, 2 + 1 + 3 = 6 in each PMC pass.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
# one stream for every launch: with the parameter-gradient reductions on their second stream (the default) they share
# the GPU with the data-gradient chain and the traced durations are no longer each kernel's own
export KSR_SIDE_WGRAD=0
WL="$@"; [ -z "$WL" ] && WL="train infer train-attn"
for W in $WL; do
  EXTRA=""; [ $W != infer ] && EXTRA="--headline-only"
  [ $W = train-unet ] && EXTRA="--headline-only --no-graph"
  OUT=$R/gpurun_out/prof_r03/$W
  mkdir -p $OUT
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline $EXTRA > $OUT/bench_under_rocprof.json 2> $OUT/trace.log || exit 1
  echo "$W trace done"
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.log || exit 1
  echo "$W fetch done"
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/pmc_write.json 2> $OUT/pmc_write.log || exit 1
  echo "$W write done"
done
