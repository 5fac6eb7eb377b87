#!/bin/bash
# UNet-backbone training bench (per-entry-point times), optionally preceded by a pytest selection: tools/quick_unet.sh ["-k expr"]
mkdir -p gpurun_out
if [ -n "$1" ]; then
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "$1" > gpurun_out/gputest.log 2>&1
  rc=$?; tail -3 gpurun_out/gputest.log; [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 300 python bench.py --workload train-unet --no-cpu-baseline --headline-only > gpurun_out/bench_unet.log 2>&1 || { tail -5 gpurun_out/bench_unet.log; exit 1; }
python tools/show_bench.py gpurun_out/bench_unet.log
