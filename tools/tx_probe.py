"""Times ksr_batch_tx_layer_fwd in the single-image inference shape (fp32, 540 x 960 pixels) for one or more builds of the
library: python tools/tx_probe.py [libksr variant.so ...]"""
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
for path in libs:
    L = ctypes.CDLL(path)

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
    print(os.path.basename(path), f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us  checksum {float(out.double().sum()):.3f}")
