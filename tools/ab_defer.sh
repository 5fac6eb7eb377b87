#!/bin/bash
# A/B of the deferred slab reductions: the headline step with KSR_DEFER_REDUCTIONS=1 / 0, at 64 and 8 patches per GPU
for b in 64 8; do for d in 1 0; do
  KSR_DEFER_REDUCTIONS=$d timeout -k 10 200 python bench.py --no-cpu-baseline --headline-only --per-gpu-batch $b --steps 40 > gpurun_out/ab_${b}_${d}.log 2>&1 || exit 1
  echo "B=$b defer=$d: $(python tools/show_bench.py gpurun_out/ab_${b}_${d}.log | head -1)"
done; done
