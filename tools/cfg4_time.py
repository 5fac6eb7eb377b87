#!/usr/bin/env python3
"""Timing of the attention-head configuration (G=16, ConvSepKanAttention(64, 64, 336) in place of feed): forward +
backward through autograd and the native training step, 64 patches of 128x128, per entry point.  python tools/cfg4_time.py [fp32|bf16|bf16s]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch  # noqa: E402

from kan_sr import _lib  # noqa: E402
from kan_sr.module import ConvSepKan  # noqa: E402

torch.manual_seed(0)
for prec in (sys.argv[1:] or ["fp32", "bf16", "bf16s"]):
    m = ConvSepKan([1], [16], [3], 16, 3, 0.1, [-1, 1], 4, attention=True).cuda().train()
    m.precision = prec
    x = torch.rand(64, 3, 128, 128, device="cuda")
    tgt = torch.rand(64, 3, 512, 512, device="cuda")

    def step():
        m.zero_grad(set_to_none=True)
        y = m(x)
        loss = (y[:, :1] - tgt[:, :1]).abs().mean()
        loss.backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{prec}: {dt * 1e3:.2f} ms/step, {64 * 512 * 512 / 1e6 / dt:.1f} output MP/s", flush=True)
    # the native training step (fused loss + Adam, tape backward into the flat gradient buffer)
    from kan_sr.train import TrainStep  # noqa: E402
    ts = TrainStep(m, lr=1e-3, loss="l1+ssim")
    for _ in range(3):
        ts.step(x, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ts.step(x, tgt)
    torch.cuda.synchronize()
    dts = (time.perf_counter() - t0) / 10
    print(f"{prec}: TrainStep(l1+ssim) {dts * 1e3:.2f} ms/step, {64 * 512 * 512 / 1e6 / dts:.1f} output MP/s", flush=True)
    _lib.PROFILE = {}
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    tot = {k: sum(s.elapsed_time(e) for s, e in v) / 3 for k, v in _lib.PROFILE.items()}
    _lib.PROFILE = None
    print({k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]}, flush=True)
