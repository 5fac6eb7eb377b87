#!/usr/bin/env python3
"""Time per launch of ksr_pointwise_fwd (64 -> n_out) from one 128-pixel tile (weight staging + one tile + launch) to 8 x 128 x
128 pixels, in the fp32 and the bf16-storage mode.  Back-to-back launches from Python: below ~6 us the host's launch rate
is what is measured; GPU-side durations come from `rocprofv3 --kernel-trace` (tools/step_timeline.py).
Usage: python tools/fixed_cost_probe.py"""
import os
import sys
from ctypes import c_int

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch  # noqa: E402

from kan_sr import engine  # noqa: E402
from kan_sr._lib import check, lib, ptr, stream  # noqa: E402

L = lib()
dev = torch.device("cuda:0")


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for prec in ("fp32", "bf16s"):
    engine.set_precision(prec)
    dt = engine.act_dtype()
    for n_out, act in ((64, 0), (128, 1), (192, 0)):            # the supported (outputs, activation) pairs
        for (B, H, W) in ((1, 4, 32), (1, 32, 32), (8, 128, 128)):
            N = B * H * W
            x = torch.randn(N, 64, device=dev).to(dt)
            w = torch.randn(n_out, 64, device=dev) * 0.1
            b = torch.zeros(n_out, device=dev)
            y = torch.empty(N, n_out, device=dev, dtype=dt)
            us = timeit(lambda: check(L.ksr_pointwise_fwd(ptr(x), ptr(w), ptr(b), ptr(y), c_int(n_out), c_int(act), c_int(B), c_int(H),
                                                            c_int(W), stream()), "pw"))
            print(f"{prec} pointwise 64->{n_out} N={N}: {us:.1f} us per launch", flush=True)
engine.set_precision("fp32")
