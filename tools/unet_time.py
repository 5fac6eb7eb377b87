#!/usr/bin/env python3
"""Time of ConvSepKan(backbone='unet') on 256x256 LR patches (x4, G=5): inference, and forward + backward in train mode,
with the per-entry-point breakdown.  Usage: python tools/unet_time.py [B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch  # noqa: E402

from kan_sr import _lib  # noqa: E402
from kan_sr.init import init_like_training_setup  # noqa: E402
from kan_sr.module import ConvSepKan  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
m = init_like_training_setup(ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4, backbone="unet")).to(dev)
x = torch.rand(B, 3, 256, 256, device=dev)
gy = torch.randn(B, 3, 1024, 1024, device=dev)


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def infer():
    with torch.no_grad():
        m(x)


def train():
    for p in m.parameters():
        p.grad = None
    m(x).backward(gy)


def breakdown(fn, label):
    _lib.PROFILE = {}
    fn()
    torch.cuda.synchronize()
    rows = sorted(((sum(s.elapsed_time(e) for s, e in v), len(v), k) for k, v in _lib.PROFILE.items()), reverse=True)
    if os.environ.get("KSR_PER_CALL"):
        for k in os.environ["KSR_PER_CALL"].split(","):
            print(f"   {k} calls (ms):", " ".join(f"{s.elapsed_time(e):.3f}" for s, e in _lib.PROFILE.get(k, [])))
    _lib.PROFILE = None
    print(f"-- {label}: kernel time by entry point (ms, calls)")
    for ms, n, k in rows:
        print(f"   {k:28s} {ms:8.3f} {n:4d}")
    print(f"   {'total':28s} {sum(r[0] for r in rows):8.3f} {sum(r[1] for r in rows):4d}")


m.eval()
t = timed(infer)
print(f"inference B={B}: {t * 1e3:.3f} ms, {B * 1024 * 1024 / 1e6 / t:.0f} output MP/s", flush=True)
breakdown(infer, "inference")
m.train()
t = timed(train)
print(f"train fwd+bwd B={B}: {t * 1e3:.3f} ms, {B * 1024 * 1024 / 1e6 / t:.0f} output MP/s", flush=True)
breakdown(train, "train fwd+bwd")
