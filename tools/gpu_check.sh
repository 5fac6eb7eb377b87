#!/bin/bash
# One GPU round trip: the GPU test suite, then the headline bench (JSON line in gpurun_out/bench_quick.log).
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread ${KSR_PYTEST_ARGS} > gpurun_out/gputest.log 2>&1
rc=$?
tail -5 gpurun_out/gputest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --headline-only > gpurun_out/bench_quick.log 2>&1 || { tail -5 gpurun_out/bench_quick.log; exit 1; }
python tools/show_bench.py gpurun_out/bench_quick.log
