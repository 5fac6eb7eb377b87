#!/usr/bin/env python3
"""Runs the training step at a small per-GPU batch (default 8 patches of 128 x 128, bf16 storage, full loss) for a few
steps: the workload for `rocprofv3 --kernel-trace -- python3 tools/b8_trace.py [B]`, whose timeline tools/step_timeline.py reads."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from kan_sr.train import TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
m = bench.build_model().to(dev).train()
m.precision = "bf16s"
ts = TrainStep(m, lr=1e-3, loss="l1+ssim")
x, t = torch.rand(B, 3, 128, 128, device=dev), torch.rand(B, 3, 512, 512, device=dev)
for _ in range(8):
    ts.step(x, t)
torch.cuda.synchronize()

# host time of a step (enqueue only) against the time the GPU needs for it: with the host slower, the step is bound by the
# Python launch path, not by the kernels
import time  # noqa: E402
N = 50
t0 = time.perf_counter()
for _ in range(N):
    ts.step(x, t)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"B={B}: host enqueue {t_host / N * 1e3:.3f} ms/step, with the GPU drained {t_all / N * 1e3:.3f} ms/step", flush=True)
