"""Launch sequences for the ConvSepKAN hot path: which kernel runs when, on which buffers.

All activations between kernels are NHWC ([B*H*W, C]; fp32, or bf16 in the "bf16s" precision mode).
`forward_infer` is the no-grad path; `encoder_forward(saves=...)` is the forward that keeps what the backward
needs, and the backward itself lives in autograd.py (one torch.autograd.Function) and train.py (native step).
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_float, c_int, c_int64, c_uint64, c_void_p
from typing import Dict, Optional, Sequence

import torch

from ._lib import KsrError, check, host_floats, lib, ptr, stream
from .ops import bilinear_up_into, make_knots

LN_EPS = 1e-5
# local batch above which the transformer layer runs as QKV GEMM + MFMA attention + post kernel instead of the
# single fused kernel (whose VALU attention is cheaper for short sequences)
SPLIT_ATTENTION_ABOVE = 8
# bf16 storage: the split path's attention kernels stage their rows by LDS-DMA and are cheaper than the fused kernel
# from this local batch on (8 patches per GPU: 79 -> 44 us per layer forward, 51 -> 26 us attention backward)
SPLIT_ATTENTION_FROM_BF16S = 4


def split_attention(local_batch: int) -> bool:
    """Whether a transformer layer over `local_batch` images runs as in-projection + attention + post kernel."""
    if local_batch > SPLIT_ATTENTION_ABOVE:
        return True
    return getattr(_tls, "precision", "fp32") == "bf16s" and local_batch >= SPLIT_ATTENTION_FROM_BF16S
TX_PARAM_NAMES = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                  "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
                  "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


def _ptr_array(tensors) -> ctypes.Array:
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else ptr(t).value
    return arr


class Geometry:
    def __init__(self, x: torch.Tensor, upscale: int, grid_size: int, spline_order: int, grid_range):
        if x.dim() != 4 or x.shape[1] != 3:
            raise KsrError(f"expected input [B,3,H,W], got {tuple(x.shape)}")
        if x.dtype != torch.float32 or not x.is_cuda:
            raise KsrError("ConvSepKan (HIP path) needs a float32 GPU tensor")
        self.B, _, self.H, self.W = x.shape
        self.N = self.B * self.H * self.W
        self.r = upscale
        self.n_coef = grid_size + spline_order
        self.k = spline_order
        self.knots = make_knots(grid_size, spline_order, grid_range)
        self.knots_c = host_floats(self.knots)
        if self.B > 128:
            raise KsrError("local batch > 128 is not supported by the batch-axis attention kernel")
        if not lib().ksr_feed_kan_supported(c_int(self.r), c_int(self.n_coef), c_int(self.k)):
            raise KsrError(f"no fused feed+KAN kernel for upscale={self.r}, grid_size+spline_order="
                           f"{self.n_coef}, spline_order={self.k}")


def _dims(g: Geometry):
    return c_int(g.B), c_int(g.H), c_int(g.W)


FUSE_SINGLE_IMAGE_LAYERS = os.environ.get("KSR_FUSE_TX2", "1") != "0"
PRECISIONS = {"fp32": 0, "bf16": 1, "bf16s": 2}
_tls = threading.local()          # the library keeps the mode per calling thread; so does this mirror of it


def set_precision(name: str) -> None:
    """Precision mode of the kernels for the calls that follow ON THIS THREAD (autograd runs the backward on a thread
    of its own: every backward entry sets the mode again):
    "fp32"  exact fp32 MFMA, fp32 storage (the parity mode);
    "bf16"  bf16 MFMA operands, fp32 accumulation, fp32 storage;
    "bf16s" as "bf16", and the [N, C] activation / gradient tensors between kernels are stored as bf16
            (the image, the output, per-pixel scalars, parameters and their gradients stay fp32)."""
    if name not in PRECISIONS:
        raise KsrError(f"precision must be one of {sorted(PRECISIONS)}, got {name!r}")
    check(lib().ksr_set_precision(c_int(PRECISIONS[name])), "ksr_set_precision")
    _tls.precision = name


def set_f32_split(on) -> None:
    """fp32 mode, forward kernels with such an instance: products as six bf16 matrix instructions on three-term splits of
    both operands (fp32-accurate, not bit-equal to the fp32 matrix instruction).  True / False for this thread, None = the
    process default (on unless KSR_F32_SPLIT=0)."""
    check(lib().ksr_set_f32_split(c_int(-1 if on is None else int(bool(on)))), "ksr_set_f32_split")


def act_dtype() -> torch.dtype:
    """Storage type of the activation tensors in the current precision mode."""
    return torch.bfloat16 if getattr(_tls, "precision", "fp32") == "bf16s" else torch.float32


def new_seed() -> int:
    """Dropout stream id drawn from torch's CPU generator (follows torch.manual_seed, no GPU sync)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def encoder_forward(x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, g: Geometry,
                    saves: Optional[dict] = None, p_drop: float = 0.0, seed: int = 0) -> torch.Tensor:
    """Y plane of x -> final-LayerNormed 64-channel features [N,64] (input of `feed`).
    With `saves` (a dict) every tensor the backward needs is kept there."""
    L = lib()
    dev = x.device
    N = g.N
    adt = act_dtype()
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=adt)                  # activations [N, C]
    new32 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)      # per-pixel scalars
    keep = saves is not None
    # ffn1 + ffn2 + ffn3.pointwise1 in one kernel that reads only the Y plane: t2 = relu(fn_1(ffn1(Y))) is relu(A v + B) per
    # channel with one scalar v per pixel.  The [N,128] tensor itself is never needed: the backward rebuilds it from v and
    # the coefficients ab, as the forward does (saves["_keep_t2"] = True materialises it for inspection).
    t2 = new(N, 128) if (keep and (adt == torch.float32 or saves.get("_keep_t2"))) else None     # bf16 storage: rebuilt from v
    ab = new32(256) if keep else None
    v = new32(N) if keep else None
    g2 = new(N, 64) if keep else None
    x3 = new(N, 64) if keep else None
    t3 = new(N, 64)
    names14 = ("ffn1.pointwise1.weight", "ffn1.pointwise1.bias", "ffn1.depthwise.weight", "ffn1.depthwise.bias",
               "ffn1.pointwise2.weight", "ffn1.pointwise2.bias", "ffn2.fn_1.0.weight", "ffn2.fn_1.0.bias",
               "ffn2.gate.weight", "ffn2.gate.bias", "ffn2.fn_2.weight", "ffn2.fn_2.bias",
               "ffn3.pointwise1.weight", "ffn3.pointwise1.bias")
    check(L.ksr_stem_gate_fwd(ptr(x), c_int64(3 * g.H * g.W), _ptr_array([p[pre + n] for n in names14]), ptr(t3),
                              ptr(v), ptr(t2), ptr(g2), ptr(x3), ptr(ab), *_dims(g), stream()), "ksr_stem_gate_fwd")
    x4 = new(N, 64)
    a4 = new(N, 64) if keep else None
    check(L.ksr_dwrelu_pw_fwd(ptr(t3), ptr(p[pre + "ffn3.depthwise.weight"]), ptr(p[pre + "ffn3.depthwise.bias"]),
                              ptr(p[pre + "ffn3.pointwise2.weight"]), ptr(p[pre + "ffn3.pointwise2.bias"]),
                              ptr(x4), ptr(a4), *_dims(g), stream()), "ksr_dwrelu_pw_fwd")
    t5 = new(N, 128)
    check(L.ksr_pointwise_fwd(ptr(x4), ptr(p[pre + "ffn4.fn_1.0.weight"]), ptr(p[pre + "ffn4.fn_1.0.bias"]),
                              ptr(t5), c_int(128), c_int(1), *_dims(g), stream()), "ksr_pointwise_fwd")
    x6 = new(N, 64)
    # bf16 storage: the first transformer layer's residual + LayerNorm reads an fp32 copy of x6 (sigmoid outputs
    # sit at 0.5 +- 1e-3 at initialisation, below bf16 resolution)
    x6_f32 = new32(N, 64) if adt != torch.float32 else None
    g6 = new(N, 64) if keep else None
    check(L.ksr_gate_pw_fwd(ptr(t5), ptr(p[pre + "ffn4.gate.weight"]), ptr(p[pre + "ffn4.gate.bias"]),
                            ptr(p[pre + "ffn4.fn_2.weight"]), ptr(p[pre + "ffn4.fn_2.bias"]), ptr(x6), ptr(g6), ptr(x6_f32),
                            *_dims(g), stream()), "ksr_gate_pw_fwd")
    if keep:
        saves.update(v=v, ab=ab, t2=t2, g2=g2, x3=x3, t3=t3, a4=a4, x4=x4, t5=t5, g6=g6, x6=x6)
    s = x6
    if g.B == 1 and not keep and adt == torch.float32 and getattr(_tls, "precision", "fp32") == "fp32" and p_drop == 0.0 and FUSE_SINGLE_IMAGE_LAYERS:
        # one image, eval: both layers and the final norm in one launch (the tensor between the layers is never written)
        q0, q1 = f"{pre}encoder.layers.0.", f"{pre}encoder.layers.1."
        out = new(N, 64)
        check(L.ksr_tx2_single_fwd(ptr(s), ptr(out), _ptr_array([p[q0 + n] for n in TX_PARAM_NAMES]),
                                   _ptr_array([p[q1 + n] for n in TX_PARAM_NAMES]), ptr(p[pre + "encoder.norm.weight"]),
                                   ptr(p[pre + "encoder.norm.bias"]), c_int64(N), c_float(LN_EPS), stream()), "ksr_tx2_single_fwd")
        return out
    for li in range(2):
        q = f"{pre}encoder.layers.{li}."
        plist = [p[q + n] for n in TX_PARAM_NAMES]
        last = li == 1
        plist += [p[pre + "encoder.norm.weight"], p[pre + "encoder.norm.bias"]] if last else [None, None]
        out = new(N, 64)
        sv = None
        if keep:
            # bf16 storage: LayerNorm 1's output s1 is not kept (linear1's weight gradient takes xhat1 and the affine map)
            sv = dict(qkv=new(N, 192), ctx=new(N, 64), xhat1=new(N, 64), rstd1=new32(N),
                      s1=new(N, 64) if adt == torch.float32 else None,
                      hmid=new(N, 32), xhat2=new(N, 64), rstd2=new32(N),
                      xhatf=new(N, 64) if last else None, rstdf=new32(N) if last else None)
            sv["in"] = s
            saves[f"tx{li}"] = sv
        names = ("qkv", "ctx", "xhat1", "rstd1", "s1", "hmid", "xhat2", "rstd2", "xhatf", "rstdf")
        lseed = c_uint64((seed * 4 + li) & 0xFFFFFFFFFFFFFFFF)
        s_f32 = x6_f32 if li == 0 else None
        if split_attention(g.B):
            # large local batch: QKV projection, MFMA attention and the rest of the layer as three kernels
            qkv = sv["qkv"] if keep else new(N, 192)
            ctx = sv["ctx"] if keep else new(N, 64)
            check(L.ksr_batch_qkv_attn_fwd(ptr(s), ptr(plist[0]), ptr(plist[1]), ptr(qkv), ptr(ctx), c_int(g.B),
                                           c_int(g.H * g.W), c_float(p_drop), lseed, stream()), "ksr_batch_qkv_attn_fwd")
            sv_arr = _ptr_array([None, None] + [sv[k] for k in names[2:]]) if keep else None
            check(L.ksr_batch_tx_layer_post_fwd(ptr(s), ptr(ctx), ptr(out), _ptr_array(plist), sv_arr, c_int64(N),
                                                c_float(LN_EPS), c_float(p_drop), lseed, ptr(s_f32), stream()),
                  "ksr_batch_tx_layer_post_fwd")
        else:
            sv_arr = _ptr_array([sv[k] for k in names]) if keep else None
            check(L.ksr_batch_tx_layer_fwd(ptr(s), ptr(out), _ptr_array(plist), sv_arr, c_int(g.B),
                                           c_int(g.H * g.W), c_float(LN_EPS), c_float(p_drop), lseed, ptr(s_f32), stream()),
                  "ksr_batch_tx_layer_fwd")
        s = out
    if keep:
        saves["feat"] = s
    return s


def feed_kan_forward(feat: torch.Tensor, x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, g: Geometry,
                     out: torch.Tensor, save_us: Optional[torch.Tensor] = None) -> None:
    check(lib().ksr_feed_kan_fwd(ptr(feat), ptr(x), c_int64(3 * g.H * g.W), ptr(p[pre + "feed.weight"]),
                                 ptr(p[pre + "feed.bias"]), ptr(out), c_int64(out.shape[1] * out.shape[2] * out.shape[3]),
                                 ptr(save_us), *_dims(g), c_int(g.r), c_int(g.n_coef), g.knots_c,
                                 c_int(len(g.knots)), c_int(g.k), stream()), "ksr_feed_kan_fwd")


def forward_infer(x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, g: Geometry,
                  p_drop: float = 0.0, seed: int = 0) -> torch.Tensor:
    """[B,3,H,W] -> [B,3,rH,rW]: Y through encoder + fused feed/KAN/shuffle, Cb/Cr bilinear."""
    out = torch.empty(g.B, 3, g.H * g.r, g.W * g.r, device=x.device, dtype=torch.float32)
    bilinear_up_into(x, out, g.r, 1, 2, 1)
    feat = encoder_forward(x, p, pre, g, None, p_drop, seed)
    feed_kan_forward(feat, x, p, pre, g, out)
    return out
