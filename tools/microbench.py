#!/usr/bin/env python3
"""Per-kernel micro-benchmarks through the C ABI (HIP events on the launch stream): achieved GB/s against
the algorithmic bytes of each call.  Usage: python tools/microbench.py [kan_cell] [wgrad] [gram]"""
import json
import os
import sys
from ctypes import c_int, c_int64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch  # noqa: E402

from kan_sr._lib import check, host_floats, lib, ptr, stream  # noqa: E402
from kan_sr.ops import make_knots  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def report(name, ms, nbytes, **kw):
    print(json.dumps({"kernel": name, "ms": round(ms, 4), "algorithmic_GB": round(nbytes / 1e9, 4),
                      "GB_per_s": round(nbytes / ms / 1e6, 1), "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000, 4), **kw}),
          flush=True)


def bench_kan_cell():
    L = lib()
    for (B, H, W, r, G) in [(8, 128, 128, 4, 5), (64, 128, 128, 4, 5), (64, 128, 128, 4, 16)]:
        k, O = 3, r * r
        Kc = G + k
        P = O * (Kc + 2)
        N = B * H * W
        knots = make_knots(G, k, [-1, 1])
        kn = host_floats(knots)
        x = torch.rand(B, 1, H, W, device="cuda")
        for nhwc in (0, 1):
            w = torch.randn(N * P, device="cuda")
            out = torch.empty(B, 1, H * r, W * r, device="cuda")
            gout = torch.randn_like(out)
            gw = torch.empty_like(w)

            def fwd():
                check(L.ksr_kan_cell_fwd(ptr(x), ptr(w), ptr(out), c_int(B), c_int(1), c_int(O), c_int(H), c_int(W), c_int(Kc),
                                         kn, c_int(len(knots)), c_int(k), c_int(nhwc), c_int(r), c_int64(0), stream()), "fwd")

            def bwd():
                check(L.ksr_kan_cell_bwd(ptr(x), ptr(w), ptr(gout), ptr(gw), ptr(None), c_int(B), c_int(1), c_int(O), c_int(H),
                                         c_int(W), c_int(Kc), kn, c_int(len(knots)), c_int(k), c_int(nhwc), c_int(r), c_int64(0),
                                         stream()), "bwd")
            lay = "nhwc" if nhwc else "nchw"
            report(f"kan_cell_fwd[{lay}]", timeit(fwd), N * (4 + 4 * P + 4 * O), B=B, P=P)
            report(f"kan_cell_bwd[{lay}]", timeit(bwd), N * (4 * O + 4 + 4 * P + 4 * P), B=B, P=P)
            del w, gw


def bench_wgrad():
    L = lib()
    N = 64 * 128 * 128
    for (J, K, a2) in [(160, 64, 0), (192, 64, 0), (64, 64, 0), (32, 64, 0), (64, 32, 0), (128, 64, 0), (64, 64, 1)]:
        dz = torch.randn(N, J, device="cuda")
        a1 = torch.randn(N, K, device="cuda")
        a2t = torch.randn(N, K, device="cuda") if a2 else None
        slab = torch.empty(int(L.ksr_wgrad_slab_floats(c_int(J), c_int(K))), device="cuda")
        dW, db = torch.empty(J, K, device="cuda"), torch.empty(J, device="cuda")

        def run():
            check(L.ksr_wgrad(ptr(dz), c_int(J), c_int(0), c_int(J), ptr(a1), c_int(K), c_int(0), c_int(K), ptr(a2t), c_int(K),
                              c_int(0), c_int64(N), ptr(slab), ptr(dW), ptr(db), c_int(0), c_int(0), stream()), "wgrad")
        ms = timeit(run)
        report(f"wgrad[J={J},K={K},a2={a2}]", ms, N * 4 * (J + K + (K if a2 else 0)),
               mfma_tflops=round(2 * J * K * N / ms / 1e9, 2))


def bench_pointwise():
    L = lib()
    B, H, W = 64, 128, 128
    N = B * H * W
    x = torch.randn(N, 64, device="cuda")
    for n_out, act in [(64, 0), (128, 1), (192, 0)]:
        w, b = torch.randn(n_out, 64, device="cuda") * 0.1, torch.randn(n_out, device="cuda")
        out = torch.empty(N, n_out, device="cuda")

        def run():
            check(L.ksr_pointwise_fwd(ptr(x), ptr(w), ptr(b), ptr(out), c_int(n_out), c_int(act), c_int(B), c_int(H), c_int(W),
                                      stream()), "pw")
        report(f"pointwise_fwd[64->{n_out}]", timeit(run), N * 4 * (64 + n_out))
    dz = torch.randn(N, 192, device="cuda")
    dx = torch.empty(N, 64, device="cuda")
    for n_out in (64, 192):
        w = torch.randn(n_out, 64, device="cuda") * 0.1

        def run():
            check(L.ksr_pointwise_bwd(ptr(dz), c_int(n_out), ptr(w), ptr(dx), c_int(B), c_int(H), c_int(W), stream()), "pwb")
        report(f"pointwise_bwd[{n_out}->64]", timeit(run), N * 4 * (64 + n_out))


def bench_attn():
    from kan_sr import engine
    L = lib()
    S, HW = 64, 128 * 128
    N = S * HW
    dt = engine.act_dtype()
    qkv = torch.randn(N, 192, device="cuda").to(dt)
    dctx = torch.randn(N, 64, device="cuda").to(dt)
    ctx = torch.empty(N, 64, device="cuda", dtype=dt)
    dqkv = torch.empty(N, 192, device="cuda", dtype=dt)
    esz = qkv.element_size()
    from ctypes import c_float, c_uint64
    for p_drop in (0.0, 0.1):
        def fwd():
            check(L.ksr_batch_attn_fwd(ptr(qkv), ptr(ctx), c_int(S), c_int(HW), c_float(p_drop), c_uint64(5), stream()), "fwd")

        def bwd():
            check(L.ksr_batch_attn_bwd(ptr(qkv), ptr(ctx), ptr(dctx), ptr(dqkv), c_int(S), c_int(HW), c_float(p_drop), c_uint64(5),
                                       stream()), "bwd")
        report(f"attn_fwd[S=64,p={p_drop}]", timeit(fwd), N * esz * (192 + 64))
        report(f"attn_bwd[S=64,p={p_drop}]", timeit(bwd), N * esz * (192 + 64 + 64 + 192))


def bench_gram():
    L = lib()
    B, HW = 64, 128 * 128
    N = B * HW
    qkv = torch.randn(N, 192, device="cuda")
    for (J, c0) in [(64, 64), (160, 128), (336, 128)]:
        dz = qkv if J == 64 else torch.randn(N, J, device="cuda")
        slab = torch.empty(int(L.ksr_gram_slab_floats(c_int(J), c_int(64), c_int(B), c_int64(HW))), device="cuda")
        out, tail = torch.empty(B, J, 64, device="cuda"), torch.empty(B, 2 * J + 64, device="cuda")

        def run():
            check(L.ksr_gram_batched(ptr(dz), c_int(dz.shape[1]), c_int(0), c_int(J), ptr(qkv), c_int(192), c_int(c0), c_int(64),
                                     c_int64(HW), c_int(B), ptr(slab), ptr(out), ptr(tail), c_int(int(J == 64)), stream()), "gram")
        report(f"gram_batched[J={J}]", timeit(run), N * 4 * (J + 64))


def bench_ssim():
    """SSIM forward (3 planes, with derivative maps) and luma gradient on 64 images of 512 x 512."""
    from ctypes import c_float
    L = lib()
    B, C, H, W = 64, 3, 512, 512
    dev = torch.device("cuda:0")
    pred, tgt = torch.rand(B, C, H, W, device=dev), torch.rand(B, C, H, W, device=dev)
    slab = torch.empty(int(L.ksr_ssim_slab_floats(c_int(C), c_int(B), c_int(H), c_int(W))), device=dev)
    maps = torch.empty(3 * B * (H - 10) * (W - 10), device=dev)
    val, gy = torch.zeros(1, device=dev), torch.zeros(B, H, W, device=dev)
    plane = H * W

    def fwd():
        check(L.ksr_ssim_fwd(ptr(pred), c_int64(3 * plane), ptr(tgt), c_int64(3 * plane), c_int(C), c_int(B), c_int(H),
                             c_int(W), ptr(maps), ptr(slab), ptr(val), stream()), "ssim_fwd")

    def grad():
        check(L.ksr_ssim_y_grad(ptr(maps), ptr(pred), c_int64(3 * plane), ptr(tgt), c_int64(3 * plane), ptr(gy),
                                c_int64(plane), c_int(B), c_int(H), c_int(W), c_float(1e-6), stream()), "ssim_y_grad")
    report("ssim_fwd", timeit(fwd), 4 * (2 * B * C * plane + 3 * B * (H - 10) * (W - 10)))
    report("ssim_y_grad", timeit(grad), 4 * (3 * B * (H - 10) * (W - 10) + 4 * B * plane))


if __name__ == "__main__":
    from kan_sr import engine
    engine.set_precision(os.environ.get("KSR_PRECISION", "fp32"))
    which = sys.argv[1:] or ["kan_cell", "wgrad", "gram", "pointwise", "attn"]
    for w in which:
        globals()["bench_" + w]()
