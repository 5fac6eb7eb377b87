#!/usr/bin/env python3
"""Inference timing at the two readings of '1080p -> 4K': LR 960x540 -> 3840x2160 and LR 1920x1080 -> 7680x4320
(fp32, and bf16 MFMA operands for comparison).  Usage: python tools/infer_sizes.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model().to(dev).eval()
for prec in ("fp32", "bf16"):
    model.precision = prec
    for (h, w) in ((540, 960), (1080, 1920)):
        x = torch.rand(1, 3, h, w, device=dev)
        with torch.no_grad():
            for _ in range(3):
                y = model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                y = model(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{prec}: {w}x{h} -> {4 * w}x{4 * h}: {dt * 1e3:.3f} ms/frame, {16 * h * w / 1e6 / dt:.0f} output MP/s, "
              f"finite={bool(torch.isfinite(y).all())}", flush=True)
