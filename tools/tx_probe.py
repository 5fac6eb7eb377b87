"""Times ksr_batch_tx_layer_fwd in the single-image inference shape (fp32, 540 x 960 pixels) for one or more builds of the
library, in the exact fp32 form and with the products split over bf16 matrix instructions (ksr_set_f32_split), and
reports each form's error against a float64 evaluation: python tools/tx_probe.py [libksr variant.so ...]"""
import ctypes, os, sys
from ctypes import c_int, c_float, c_uint64, c_void_p
import torch

here = os.path.dirname(os.path.abspath(__file__))
libs = sys.argv[1:] or [os.path.join(here, "..", "super-resolution_amd", "kan_sr", "libksr.so")]
HW = 540 * 960
torch.manual_seed(0)
x, out = torch.rand(HW, 64, device="cuda"), torch.empty(HW, 64, device="cuda")
shapes = [(192, 64), (192,), (64, 64), (64,), (32, 64), (32,), (64, 32), (64,), (64,), (64,), (64,), (64,), (64,), (64,)]
params = [torch.randn(*s, device="cuda") * 0.1 for s in shapes]
arr = (c_void_p * 14)(*[c_void_p(t.data_ptr()) for t in params])


def reference64():
    """The single-image layer in float64 (softmax over one key = 1: ctx = V projection)."""
    in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b, nf_w, nf_b = [t.double() for t in params]
    xd = x.double()
    ln = lambda v, w, b: torch.nn.functional.layer_norm(v, (64,), w, b, 1e-5)
    ctx = xd @ in_w[128:].T + in_b[128:]
    s1 = ln(xd + ctx @ out_w.T + out_b, n1_w, n1_b)
    s2 = ln(s1 + torch.relu(s1 @ l1_w.T + l1_b) @ l2_w.T + l2_b, n2_w, n2_b)
    return ln(s2, nf_w, nf_b)


def run(L):
    def call():
        rc = L.ksr_batch_tx_layer_fwd(c_void_p(x.data_ptr()), c_void_p(out.data_ptr()), arr, None, c_int(1), c_int(HW), c_float(1e-5),
                                      c_float(0.0), c_uint64(0), None, c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


ref = reference64()
for path in libs:
    L = ctypes.CDLL(path)
    for split in (0, 1):
        if hasattr(L, "ksr_set_f32_split"):
            L.ksr_set_f32_split(c_int(split))
        elif split:
            continue
        us = run(L)
        err = (out.double() - ref).abs()
        print(os.path.basename(path), f"split={split}: {us:.1f} us  max |err| {float(err.max()):.3e}  mean |err| {float(err.mean()):.3e}  "
              f"(|ref| max {float(ref.abs().max()):.2f})")
