#!/bin/bash
# The secondary bench workloads in one GPU round trip: batch sweep (B = 8, 16), attention head, UNet backbone, inference.
mkdir -p gpurun_out
timeout -k 10 200 python tools/batch_sweep.py > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
cat gpurun_out/sweep.log
for W in train-attn train-unet infer; do
  timeout -k 10 300 python bench.py --workload $W --no-cpu-baseline --headline-only > gpurun_out/bench_$W.log 2>&1 || { tail -5 gpurun_out/bench_$W.log; exit 1; }
  echo "== $W"; python tools/show_bench.py gpurun_out/bench_$W.log
done
