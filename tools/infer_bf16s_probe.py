#!/usr/bin/env python3
"""One 960x540 frame through the model in eval mode in every precision mode: time per frame and the difference of the
bf16 modes' output from the fp32 output (cosine, relative L2 over the luma plane).  Usage: python tools/infer_bf16s_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = bench.build_model().to(dev).eval()
x = torch.rand(1, 3, 540, 960, device=dev)
ref = None
for prec in ("fp32", "bf16", "bf16s"):
    m.precision = prec
    with torch.no_grad():
        for _ in range(3):
            y = m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            y = m(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    line = f"{prec}: {dt * 1e3:.3f} ms/frame"
    if ref is None:
        ref = y[:, 0].double()
    else:
        d = y[:, 0].double()
        cos = float((d * ref).sum() / (d.norm() * ref.norm()))
        line += f"  cos {cos:.6f}  rel-L2 {float((d - ref).norm() / ref.norm()):.3e}  max|d| {float((d - ref).abs().max()):.3e}"
    print(line, flush=True)
