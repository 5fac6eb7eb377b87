"""Where the SSIM forward's time goes (GPU box): times ksr_ssim_fwd on the training shape with / without the derivative maps
and for one channel.  python tools/ssim_probe.py [libksr variant.so ...]"""
import ctypes, os, sys
from ctypes import c_int, c_int64, c_void_p
import torch

here = os.path.dirname(os.path.abspath(__file__))
libs = sys.argv[1:] or [os.path.join(here, "..", "super-resolution_amd", "kan_sr", "libksr.so")]
B, H, W = 64, 512, 512
pred, tgt = torch.rand(B, 3, H, W, device="cuda"), torch.rand(B, 3, H, W, device="cuda")
slab = torch.empty(1 << 20, device="cuda")
maps = torch.empty(3 * B * (H - 10) * (W - 10), device="cuda")
val = torch.zeros(1, device="cuda")
P = lambda t: c_void_p(t.data_ptr())


def run(L, C, with_maps, reps=10):
    def call():
        rc = L.ksr_ssim_fwd(P(pred), c_int64(3 * H * W), P(tgt), c_int64(3 * H * W), c_int(C), c_int(B), c_int(H), c_int(W),
                            P(maps) if with_maps else None, P(slab), P(val), c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for path in libs:
    L = ctypes.CDLL(path)
    print(os.path.basename(path), " ".join(f"C={C} maps={int(m)}: {run(L, C, m) * 1e3:.1f} us |" for C, m in ((3, True), (3, False), (1, False))),
          f"ssim={float(val):.6f}")
