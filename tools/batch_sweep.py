#!/usr/bin/env python3
"""Training-step time against the per-GPU batch (128x128 patches, bf16 storage mode, full loss): the strong-scaling
shards of a global batch of 64.  Usage: python tools/batch_sweep.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from kan_sr.train import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
for B, capture in ((8, False), (8, True), (16, False), (16, True), (32, False), (32, True), (64, False), (64, True), (128, False)):
    m = bench.build_model().to(dev).train()
    m.precision = "bf16s"
    ts = TrainStep(m, lr=1e-3, loss="l1+ssim", capture=capture)
    x, t = torch.rand(B, 3, 128, 128, device=dev), torch.rand(B, 3, 512, 512, device=dev)
    for _ in range(5):
        ts.step(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ts.step(x, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"B={B}{' (HIP graph)' if capture else ''}: {dt * 1e3:.3f} ms/step, {B * 512 * 512 / 1e6 / dt:.0f} output MP/s", flush=True)
    ts.close()
