#!/usr/bin/env python3
"""Batch-axis attention kernels alone (bf16 storage, S = 64 images, 128x128 pixels): time with and without dropout."""
import os, sys
from ctypes import c_float, c_int, c_uint64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch
from kan_sr import engine
from kan_sr._lib import lib, ptr, stream, check

S, HW = 64, 128 * 128
N = S * HW
L = lib()
engine.set_precision("bf16s")
qkv = torch.randn(N, 192, device="cuda").bfloat16()
dctx = torch.randn(N, 64, device="cuda").bfloat16()
ctx = torch.empty(N, 64, device="cuda").bfloat16()
dqkv = torch.empty(N, 192, device="cuda").bfloat16()
w_in = torch.randn(192, 64, device="cuda") * 0.2
b_in = torch.randn(192, device="cuda") * 0.1
x_in = torch.randn(N, 64, device="cuda").bfloat16()
qkv_o = torch.empty(N, 192, device="cuda").bfloat16()
for p in (0.0, 0.1):
    for name, fn in (("in_proj + fwd, qkv stored", lambda: L.ksr_batch_qkv_attn_fwd(ptr(x_in), ptr(w_in), ptr(b_in), ptr(qkv_o), ptr(ctx), c_int(S), c_int(HW), c_float(p), c_uint64(5), stream())),
                     ("in_proj + fwd, qkv not stored", lambda: L.ksr_batch_qkv_attn_fwd(ptr(x_in), ptr(w_in), ptr(b_in), None, ptr(ctx), c_int(S), c_int(HW), c_float(p), c_uint64(5), stream())),
                     ("fwd", lambda: L.ksr_batch_attn_fwd(ptr(qkv), ptr(ctx), c_int(S), c_int(HW), c_float(p), c_uint64(5), stream())),
                     ("bwd", lambda: L.ksr_batch_attn_bwd(ptr(qkv), ptr(ctx), ptr(dctx), ptr(dqkv), c_int(S), c_int(HW), c_float(p), c_uint64(5), stream()))):
        for _ in range(3):
            check(fn(), name)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        print(f"p_drop={p} {name}: {s.elapsed_time(e) / 20 * 1e3:.1f} us", flush=True)
