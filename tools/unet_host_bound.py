#!/usr/bin/env python3
"""Is the UNet training step bound by the host's launch rate?  For per-GPU batches B the step time with the GPU kept
busy (wall), and the host's own enqueue time per step (timed without waiting for the GPU, queue drained before).
Both with every kernel launched from the host and with TrainStep(capture=True) (forward + loss + backward replayed from a
HIP graph).  Usage: python tools/unet_host_bound.py [B ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch  # noqa: E402

from kan_sr.init import init_like_training_setup  # noqa: E402
from kan_sr.module import ConvSepKan  # noqa: E402
from kan_sr.train import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
for B, capture in [(b, c) for b in ([int(a) for a in sys.argv[1:]] or [2, 4, 8, 16]) for c in (False, True)]:
    torch.manual_seed(0)
    m = init_like_training_setup(ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4, backbone="unet")).to(dev).train()
    ts = TrainStep(m, lr=1e-4, loss="l1+ssim", capture=capture)
    x, tgt = torch.rand(B, 3, 256, 256, device=dev), torch.rand(B, 3, 1024, 1024, device=dev)
    for _ in range(5):
        ts.step(x, tgt)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step(x, tgt)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    host = 0.0
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts.step(x, tgt)
        host += (time.perf_counter() - t0) / 5
    torch.cuda.synchronize()
    print(f"B={B:3d} {'graph   ' if capture else 'launched'}: step {wall * 1e3:7.3f} ms  host enqueue {host * 1e3:7.3f} ms  {B * 1.048576 / wall:7.0f} output MP/s", flush=True)
