"""Training path: one torch.autograd.Function whose forward and backward are sequences of libksr
kernels.  PyTorch only carries the tape node and owns the buffers."""
from __future__ import annotations

import contextlib
import os

from ctypes import c_float, c_int, c_int64, c_uint64, c_void_p
from typing import Dict, List, Optional

import torch

from . import engine
from ._lib import KsrError, check, lib, ptr, stream
from .engine import Geometry, TX_PARAM_NAMES, _dims, _ptr_array
from .ops import bilinear_up_into

# order in which the encoder's parameters enter (and their gradients leave) the autograd node
PARAM_ORDER = (
    ["ffn1.pointwise1.weight", "ffn1.pointwise1.bias", "ffn1.depthwise.weight", "ffn1.depthwise.bias",
     "ffn1.pointwise2.weight", "ffn1.pointwise2.bias"]
    + [f"{g}.{n}" for g in ("ffn2",) for n in ("fn_1.0.weight", "fn_1.0.bias", "fn_2.weight", "fn_2.bias",
                                              "gate.weight", "gate.bias")]
    + ["ffn3.pointwise1.weight", "ffn3.pointwise1.bias", "ffn3.depthwise.weight", "ffn3.depthwise.bias",
       "ffn3.pointwise2.weight", "ffn3.pointwise2.bias"]
    + [f"ffn4.{n}" for n in ("fn_1.0.weight", "fn_1.0.bias", "fn_2.weight", "fn_2.bias", "gate.weight", "gate.bias")]
    + [f"encoder.layers.{l}.{n}" for l in range(2) for n in TX_PARAM_NAMES]
    + ["encoder.norm.weight", "encoder.norm.bias", "feed.weight", "feed.bias"]
)


_SIDE_STREAMS: dict = {}
SIDE_STREAM_FOR_WGRAD = os.environ.get("KSR_SIDE_WGRAD", "1") != "0"
SIDE_STREAM_MIN_TOKENS = 1 << 19          # LR pixels per backward pass from which the second stream is used
# One launch for all slab reductions of a gradient bucket (ksr_reduce_defer / ksr_reduce_flush) instead of one per weight
# gradient.  Pays where the step is bound by its launch count (8 patches of 128 x 128 per GPU: 0.913 -> 0.884 ms); with many
# pixels the immediate reduction reads its slab while it is still in the caches and the deferred one re-reads all slabs
# (170 MB) from memory (64 patches: 4.48 -> 4.65 ms), so it is used below DEFER_BELOW_TOKENS pixels per pass.  Running a
# stream's recorded sums in groups of k (ksr_reduce_defer(k), KSR_DEFER_WINDOW=k) is slower there as well (4.34 ms immediate,
# 4.37 / 4.39 / 4.41 ms with k = 4 / 3 / 8): what the immediate form has is ONE slab that every producer rewrites and that
# therefore never leaves the caches, where every recorded sum needs a slab of its own.
# KSR_DEFER_REDUCTIONS=0 / 1 / k forces none / one launch per bucket / groups of k.
# bf16 storage: the per-token backward of a transformer layer as two kernels that accumulate linear2's / linear1's weight
# gradients themselves (ksr_batch_tx_layer_bwd_mlp_wg, ..._ln1_wg) instead of one kernel + two ksr_wgrad launches
SPLIT_TOKEN_BACKWARD = os.environ.get("KSR_SPLIT_TOKEN_BWD", "1") != "0"
DEFER_REDUCTIONS = os.environ.get("KSR_DEFER_REDUCTIONS", "auto")
DEFER_BELOW_TOKENS = 1 << 19
DEFER_WINDOW = int(os.environ.get("KSR_DEFER_WINDOW", "0"))


def _side_stream(dev):
    key = (dev.type, dev.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


class _Ws:
    """Scratch allocator for one backward pass.  With an `arena` (name -> view into a flat gradient
    buffer) every parameter gradient is written in place by the kernels.

    The parameter-gradient reductions (wgrad, depthwise wgrad, stem sums) hang off the data-gradient chain: they
    are enqueued on a second stream, so that they fill the ramp-up / tail of the chain's kernels instead of
    lengthening it; `join()` makes the main stream wait for them."""

    def __init__(self, dev, arena=None, n_tokens=0, allow_side_stream=True):
        self.dev = dev
        # Deferred reductions: every weight-gradient kernel leaves per-workgroup partial rows in a slab of its own and only
        # RECORDS the fixed-order sum that finishes it; flush() runs all recorded sums in one launch (21 launches of ~5 us
        # per step otherwise: a tenth of the step at 8 patches per GPU).  Same sums, same order, same bits.
        mode = 0
        if dev.type == "cuda" and DEFER_REDUCTIONS != "0":
            if DEFER_REDUCTIONS == "auto":
                mode = 1 if 0 < n_tokens < DEFER_BELOW_TOKENS else (DEFER_WINDOW if DEFER_WINDOW >= 2 else 0)
            else:
                mode = max(int(DEFER_REDUCTIONS), 0)
        self.defer = mode > 0
        self._held = []           # slabs of recorded reductions: alive until the flush
        self._after_flush = []    # launches that continue from a reduced gradient (the affine map of linear1's)
        if self.defer:
            check(lib().ksr_reduce_defer(c_int(mode)), "ksr_reduce_defer")
        self._slab = None
        self._ln_slab = None
        self._sums_slab = None
        self._fused_slab = None
        self.dpre = None
        self.arena = arena
        # small problems are bound by the host's launch rate, where the extra stream bookkeeping only costs
        self.wstream = _side_stream(dev) if (SIDE_STREAM_FOR_WGRAD and allow_side_stream and dev.type == "cuda"
                                             and n_tokens >= SIDE_STREAM_MIN_TOKENS) else None

    @contextlib.contextmanager
    def side(self, *tensors):
        """Runs the enclosed launches on the side stream, after everything enqueued so far on the current one."""
        if self.wstream is None:
            yield
            return
        self.wstream.wait_stream(torch.cuda.current_stream())
        for t in tensors:
            if t is not None:
                t.record_stream(self.wstream)       # allocated on the main stream, read / written on the side stream
        with torch.cuda.stream(self.wstream):
            yield

    def join(self):
        if self.wstream is not None:
            torch.cuda.current_stream().wait_stream(self.wstream)

    def flush(self):
        """Main stream waits for the second stream, then every recorded reduction runs (one launch) and what was waiting
        for their results follows."""
        self.join()
        if self.defer:
            check(lib().ksr_reduce_flush(stream()), "ksr_reduce_flush")
        for fn in self._after_flush:
            fn()
        self._after_flush.clear()
        self._held.clear()

    def after_reduction(self, fn):
        """Runs fn (launches that read a gradient some reduction writes) now, or after the flush when reductions are deferred."""
        if self.defer:
            self._after_flush.append(fn)
        else:
            fn()

    def close(self, ok=True):
        if self.defer:
            if ok:
                self.flush()
            check(lib().ksr_reduce_defer(c_int(0 if ok else -1)), "ksr_reduce_defer")
            self.defer = False

    def new(self, *shape):
        return torch.empty(*shape, device=self.dev, dtype=torch.float32)

    def act(self, *shape):
        """[N, C] activation-gradient tensor in the storage type of the current precision mode."""
        return torch.empty(*shape, device=self.dev, dtype=engine.act_dtype())

    def grad(self, name, *shape):
        if self.arena is None:
            return self.new(*shape)
        t = self.arena[name]
        assert t.numel() == torch.Size(shape).numel(), name
        return t

    def grad_pair(self, name_w, name_b, n):
        """[2, n] tensor whose rows are the gradients of two adjacent parameters (weight, bias)."""
        if self.arena is None:
            return self.new(2, n)
        w, b = self.arena[name_w], self.arena[name_b]
        assert w.numel() == n and b.numel() == n and b.data_ptr() == w.data_ptr() + 4 * n, (name_w, name_b)
        return torch.as_strided(w, (2, n), (n, 1))

    def fused_wgrad(self) -> bool:
        """bf16 storage mode: the data-gradient kernels of the 64 -> 64 convs accumulate the weight gradient themselves."""
        return engine.act_dtype() is torch.bfloat16

    def fused_slab(self):
        """Workspace of the weight gradients that data-gradient kernels accumulate themselves.  All of those launches are
        on the caller's stream, in order, and each call's reduction has consumed the rows before the next call's kernel
        starts; it must stay apart from slab(), which the reductions on the second stream use concurrently.
        With deferred reductions every call gets a workspace of its own, kept until the flush."""
        if self._fused_slab is None or self.defer:
            if getattr(_Ws, "_fused_n", None) is None:
                _Ws._fused_n = max(int(lib().ksr_fused_wgrad_slab_floats()), int(lib().ksr_in_proj_bwd_wg_slab_floats()),
                                   int(lib().ksr_tx_bwd_wg_slab_floats()))
            self._fused_slab = torch.empty(_Ws._fused_n, device=self.dev, dtype=torch.float32)
            if self.defer:
                self._held.append(self._fused_slab)
        return self._fused_slab

    def sums_slab(self):
        if self._sums_slab is None:
            self._sums_slab = torch.empty(3 * 512 * 128, device=self.dev, dtype=torch.float32)
        return self._sums_slab

    def ln_slab(self):
        """Workspace of the LayerNorm gain / shift sums; their closing sums are recorded with the others while deferring."""
        if self._ln_slab is None or self.defer:
            self._ln_slab = torch.empty(3 * 768 * 128, device=self.dev, dtype=torch.float32)
            if self.defer:
                self._held.append(self._ln_slab)
        return self._ln_slab

    def slab(self, nfloats):
        if self.defer:            # one workspace per call, kept until the flush
            t = torch.empty(max(nfloats, 1), device=self.dev, dtype=torch.float32)
            self._held.append(t)
            return t
        if self._slab is None or self._slab.numel() < nfloats:
            self._slab = torch.empty(max(nfloats, 1024 * (192 * 64 + 192)), device=self.dev, dtype=torch.float32)
        return self._slab


def _wgrad(ws: _Ws, dz, dz_ld, dz_c0, J, a1, a1_ld, a1_c0, K, N, dW, db, a2=None, a2_ld=0, a2_c0=0, kan=(0, 0)):
    L = lib()
    slab = ws.slab(int(L.ksr_wgrad_slab_floats(c_int(J), c_int(K))))
    with ws.side(dz, a1, a2, slab, dW, db):
        check(L.ksr_wgrad(ptr(dz), c_int(dz_ld), c_int(dz_c0), c_int(J), ptr(a1), c_int(a1_ld), c_int(a1_c0), c_int(K),
                          ptr(a2), c_int(a2_ld), c_int(a2_c0), c_int64(N), ptr(slab), ptr(dW), ptr(db),
                          c_int(kan[0]), c_int(kan[1]), stream()), "ksr_wgrad")


def _colreduce(ws: _Ws, a, a_ld, a_c0, b, b_scalar, N, out, out_c0=0):
    """out[0, out_c0 + c] = sum_n a[n, a_c0 + c] * b[n(, c)];  out[1, out_c0 + c] = sum_n a[n, a_c0 + c]."""
    slab = ws.slab(1024 * 128)
    check(lib().ksr_colreduce(ptr(a), c_int(a_ld), c_int(a_c0), ptr(b), c_int(int(b_scalar)), c_int64(N), ptr(slab),
                              c_void_p(out.data_ptr() + 4 * out_c0), c_int(out.shape[1]), stream()), "ksr_colreduce")
    return out


def _dwconv_wgrad(ws: _Ws, inp, in_ld, in_c0, ds, g: Geometry, name_w, name_b):
    dW = ws.grad(name_w, 64, 1, 3, 3)
    db = ws.grad(name_b, 64)
    slab = ws.slab(512 * 640)
    with ws.side(inp, ds, slab, dW, db):
        check(lib().ksr_dwconv_wgrad(ptr(inp), c_int(in_ld), c_int(in_c0), ptr(ds), c_int(64), c_int(0), ptr(slab), ptr(dW),
                                     ptr(db), *_dims(g), stream()), "ksr_dwconv_wgrad")
    return dW, db


def _tx_layer_backward(ws: _Ws, dy, li: int, p, pre, g: Geometry, sv, p_drop, seed, grads, y_in_sigmoid,
                       final_norm_done: bool = False):
    """Backward of transformer layer li.  dy: gradient of its output [N,64] - for the last layer, of the final norm's
    output, or with final_norm_done of the final norm's input.  Returns the gradient of its input (times sigmoid' of
    that input when y_in_sigmoid is given)."""
    L = lib()
    N = g.N
    q = f"{pre}encoder.layers.{li}."
    last = li == 1
    nf_w = p[pre + "encoder.norm.weight"] if last else None
    params = [nf_w, p[q + "norm2.weight"], p[q + "norm1.weight"], p[q + "linear2.weight"], p[q + "linear1.weight"],
              p[q + "self_attn.out_proj.weight"]]
    saved = [sv["xhatf"], sv["rstdf"], sv["xhat2"], sv["rstd2"], sv["hmid"], sv["xhat1"], sv["rstd1"]]
    fused = ws.fused_wgrad()
    assert not final_norm_done or (fused and SPLIT_TOKEN_BACKWARD and last)
    if fused and SPLIT_TOKEN_BACKWARD and (not last or final_norm_done):
        return _tx_layer_backward_split(ws, dy, li, p, pre, g, sv, p_drop, seed, grads, y_in_sigmoid)
    dm, dpre, du1, dctx = ws.act(N, 64), ws.act(N, 32), ws.act(N, 64), ws.act(N, 64)
    # bf16 storage: the token kernel stops at du1; the out-projection's backward (with out_proj's weight gradient) follows
    da = ws.act(N, 64) if (p_drop > 0 and not fused) else None
    lseed = (seed * 4 + li) & 0xFFFFFFFFFFFFFFFF
    # LayerNorm gains / shifts are summed inside the token kernel (ds2 / ds1 would only feed those sums)
    gn2 = ws.grad_pair(q + "norm2.weight", q + "norm2.bias", 64)
    gn1 = ws.grad_pair(q + "norm1.weight", q + "norm1.bias", 64)
    gnf = ws.grad_pair(pre + "encoder.norm.weight", pre + "encoder.norm.bias", 64) if last else None
    check(L.ksr_batch_tx_layer_bwd_token(ptr(dy), _ptr_array(params), _ptr_array(saved),
                                         _ptr_array([None, dm, dpre, None, du1, da, None if fused else dctx]), c_int64(N),
                                         c_float(p_drop), c_uint64(lseed), ptr(ws.ln_slab()), _ptr_array([gnf, gn2, gn1]),
                                         c_int(64), stream()), "ksr_batch_tx_layer_bwd_token")
    gw, gb = ws.grad(q + "linear2.weight", 64, 32), ws.grad(q + "linear2.bias", 64)
    _wgrad(ws, dm, 64, 0, 64, sv["hmid"], 32, 0, 32, N, gw, gb)
    grads[q + "linear2.weight"], grads[q + "linear2.bias"] = gw, gb
    if da is None:
        da = du1
    if last:
        grads[pre + "encoder.norm.weight"], grads[pre + "encoder.norm.bias"] = gnf[0], gnf[1]
    grads[q + "norm2.weight"], grads[q + "norm2.bias"] = gn2[0], gn2[1]
    grads[q + "norm1.weight"], grads[q + "norm1.bias"] = gn1[0], gn1[1]
    # linear2 / linear1 / out_proj weights
    gw, gb = ws.grad(q + "linear1.weight", 32, 64), ws.grad(q + "linear1.bias", 32)
    if sv.get("s1") is not None:
        _wgrad(ws, dpre, 32, 0, 32, sv["s1"], 64, 0, 64, N, gw, gb)
    else:
        # the forward did not keep LayerNorm 1's output s1 = xhat1 * gain + shift: product with xhat1, then the affine map
        _wgrad(ws, dpre, 32, 0, 32, sv["xhat1"], 64, 0, 64, N, gw, gb)

        def fold(gw=gw, gb=gb, n1w=p[q + "norm1.weight"], n1b=p[q + "norm1.bias"]):
            check(L.ksr_wgrad_affine_input(ptr(gw), ptr(gb), ptr(n1w), ptr(n1b), c_int(32), c_int(64), stream()),
                  "ksr_wgrad_affine_input")
        if ws.defer:
            ws.after_reduction(fold)          # the product's reduction is recorded, not run: the map follows the flush
        else:
            with ws.side(gw, gb):
                fold()
    grads[q + "linear1.weight"], grads[q + "linear1.bias"] = gw, gb
    gw, gb = ws.grad(q + "self_attn.out_proj.weight", 64, 64), ws.grad(q + "self_attn.out_proj.bias", 64)
    if fused:
        check(L.ksr_batch_tx_layer_bwd_out_wg(ptr(du1), ptr(sv["ctx"]), ptr(p[q + "self_attn.out_proj.weight"]), ptr(dctx),
                                              c_int64(N), c_float(p_drop), c_uint64(lseed), ptr(ws.fused_slab()), ptr(gw), ptr(gb),
                                              stream()), "ksr_batch_tx_layer_bwd_out_wg")
    else:
        _wgrad(ws, da, 64, 0, 64, sv["ctx"], 64, 0, 64, N, gw, gb)
    grads[q + "self_attn.out_proj.weight"], grads[q + "self_attn.out_proj.bias"] = gw, gb
    # attention
    dqkv = ws.act(N, 192)
    attn_bwd = L.ksr_batch_attn_bwd if engine.split_attention(g.B) else L.ksr_batch_tx_layer_bwd_attn
    check(attn_bwd(ptr(sv["qkv"]), ptr(sv["ctx"]), ptr(dctx), ptr(dqkv), c_int(g.B), c_int(g.H * g.W), c_float(p_drop),
                   c_uint64(lseed), stream()), "attention backward")
    gw, gb = ws.grad(q + "self_attn.in_proj_weight", 192, 64), ws.grad(q + "self_attn.in_proj_bias", 192)
    grads[q + "self_attn.in_proj_weight"], grads[q + "self_attn.in_proj_bias"] = gw, gb
    dsin = ws.act(N, 64)
    if ws.fused_wgrad():
        # bf16 storage: data gradient and weight gradient of the in-projection share one pass over dqkv
        assert y_in_sigmoid is None or y_in_sigmoid.data_ptr() == sv["in"].data_ptr()
        slab = ws.fused_slab()      # not ws.slab(): that one belongs to the reductions on the second stream
        check(L.ksr_in_proj_bwd_wg(ptr(dqkv), ptr(p[q + "self_attn.in_proj_weight"]), ptr(du1), ptr(sv["in"]),
                                   c_int(0 if y_in_sigmoid is None else 1), ptr(dsin), ptr(slab), ptr(gw), ptr(gb),
                                   c_int64(N), stream()), "ksr_in_proj_bwd_wg")
        return dsin
    _wgrad(ws, dqkv, 192, 0, 192, sv["in"], 64, 0, 64, N, gw, gb)
    check(L.ksr_in_proj_bwd(ptr(dqkv), ptr(p[q + "self_attn.in_proj_weight"]), ptr(du1), ptr(y_in_sigmoid), ptr(dsin),
                            *_dims(g), stream()), "ksr_in_proj_bwd")
    return dsin


def _tx_layer_backward_split(ws: _Ws, dy, li: int, p, pre, g: Geometry, sv, p_drop, seed, grads, y_in_sigmoid):
    """bf16 storage: the layer's backward with every weight gradient accumulated by the kernel that holds its operands:
    norm2 + linear2 (ksr_batch_tx_layer_bwd_mlp_wg), linear1 + norm1 (..._ln1_wg), out-projection (..._out_wg),
    attention, in-projection (ksr_in_proj_bwd_wg).  No ksr_wgrad launch, and dm is never written.  dy is the gradient of
    the layer's own output (for the last layer the final norm's backward has run already)."""
    L = lib()
    N = g.N
    q = f"{pre}encoder.layers.{li}."
    lseed = (seed * 4 + li) & 0xFFFFFFFFFFFFFFFF
    gn2 = ws.grad_pair(q + "norm2.weight", q + "norm2.bias", 64)
    gn1 = ws.grad_pair(q + "norm1.weight", q + "norm1.bias", 64)
    du2, dpre, du1, dctx = ws.act(N, 64), ws.act(N, 32), ws.act(N, 64), ws.act(N, 64)
    gw2, gb2 = ws.grad(q + "linear2.weight", 64, 32), ws.grad(q + "linear2.bias", 64)
    check(L.ksr_batch_tx_layer_bwd_mlp_wg(ptr(dy), ptr(sv["xhat2"]), ptr(sv["rstd2"]), ptr(sv["hmid"]), ptr(p[q + "norm2.weight"]),
                                          ptr(p[q + "linear2.weight"]), ptr(du2), ptr(dpre), c_int64(N), c_float(p_drop),
                                          c_uint64(lseed), ptr(ws.ln_slab()), ptr(gn2), c_int(64), ptr(ws.fused_slab()),
                                          ptr(gw2), ptr(gb2), stream()), "ksr_batch_tx_layer_bwd_mlp_wg")
    grads[q + "linear2.weight"], grads[q + "linear2.bias"] = gw2, gb2
    grads[q + "norm2.weight"], grads[q + "norm2.bias"] = gn2[0], gn2[1]
    gw1, gb1 = ws.grad(q + "linear1.weight", 32, 64), ws.grad(q + "linear1.bias", 32)
    check(L.ksr_batch_tx_layer_bwd_ln1_wg(ptr(du2), ptr(dpre), ptr(sv["xhat1"]), ptr(sv["rstd1"]), ptr(p[q + "norm1.weight"]),
                                          ptr(p[q + "linear1.weight"]), ptr(du1), c_int64(N), ptr(ws.ln_slab()), ptr(gn1), c_int(64),
                                          ptr(ws.fused_slab()), ptr(gw1), ptr(gb1), stream()), "ksr_batch_tx_layer_bwd_ln1_wg")
    grads[q + "norm1.weight"], grads[q + "norm1.bias"] = gn1[0], gn1[1]

    # the forward did not keep LayerNorm 1's output s1 = xhat1 * gain + shift: the product was taken with xhat1
    def fold(gw=gw1, gb=gb1, n1w=p[q + "norm1.weight"], n1b=p[q + "norm1.bias"]):
        check(L.ksr_wgrad_affine_input(ptr(gw), ptr(gb), ptr(n1w), ptr(n1b), c_int(32), c_int(64), stream()),
              "ksr_wgrad_affine_input")
    ws.after_reduction(fold)
    grads[q + "linear1.weight"], grads[q + "linear1.bias"] = gw1, gb1
    gw, gb = ws.grad(q + "self_attn.out_proj.weight", 64, 64), ws.grad(q + "self_attn.out_proj.bias", 64)
    check(L.ksr_batch_tx_layer_bwd_out_wg(ptr(du1), ptr(sv["ctx"]), ptr(p[q + "self_attn.out_proj.weight"]), ptr(dctx),
                                          c_int64(N), c_float(p_drop), c_uint64(lseed), ptr(ws.fused_slab()), ptr(gw), ptr(gb),
                                          stream()), "ksr_batch_tx_layer_bwd_out_wg")
    grads[q + "self_attn.out_proj.weight"], grads[q + "self_attn.out_proj.bias"] = gw, gb
    dqkv = ws.act(N, 192)
    attn_bwd = L.ksr_batch_attn_bwd if engine.split_attention(g.B) else L.ksr_batch_tx_layer_bwd_attn
    check(attn_bwd(ptr(sv["qkv"]), ptr(sv["ctx"]), ptr(dctx), ptr(dqkv), c_int(g.B), c_int(g.H * g.W), c_float(p_drop),
                   c_uint64(lseed), stream()), "attention backward")
    gw, gb = ws.grad(q + "self_attn.in_proj_weight", 192, 64), ws.grad(q + "self_attn.in_proj_bias", 192)
    grads[q + "self_attn.in_proj_weight"], grads[q + "self_attn.in_proj_bias"] = gw, gb
    dsin = ws.act(N, 64)
    assert y_in_sigmoid is None or y_in_sigmoid.data_ptr() == sv["in"].data_ptr()
    check(L.ksr_in_proj_bwd_wg(ptr(dqkv), ptr(p[q + "self_attn.in_proj_weight"]), ptr(du1), ptr(sv["in"]),
                               c_int(0 if y_in_sigmoid is None else 1), ptr(dsin), ptr(ws.fused_slab()), ptr(gw), ptr(gb),
                               c_int64(N), stream()), "ksr_in_proj_bwd_wg")
    return dsin


def _gated_backward(ws: _Ws, dz, name: str, p, pre, g: Geometry, t, gate_saved, grads, x_in, stem_saves=None, x_img=None,
                    stem_ab=None):
    """GatedFFN backward.  dz: gradient of fn_2's output (sigmoid' already applied) [N,64]; t = relu(fn_1(x_in))
    [N,128] (None for the first block, where it is rebuilt from stem_saves = v [N] and stem_ab = A|B [256]).  Returns
    the gradient of the block input x_in [N,64]."""
    L = lib()
    N = g.N
    q = pre + name + "."
    dt = ws.act(N, 128)
    dgate = ws.act(N, 64)
    gw, gb = ws.grad(q + "fn_2.weight", 64, 64, 1, 1), ws.grad(q + "fn_2.bias", 64)
    if stem_saves is not None and t is None:
        return _stem_gated_backward(ws, dz, dt, dgate, gw, gb, q, p, pre, g, gate_saved, grads, stem_saves, stem_ab, x_img)
    if ws.fused_wgrad():
        # bf16 storage: the kernel that turns dz into dt / dgate holds both operands of fn_2's weight gradient
        check(L.ksr_gate_pw_bwd_wg(ptr(dz), ptr(p[q + "fn_2.weight"]), ptr(gate_saved), ptr(t), ptr(dt), ptr(dgate),
                                   ptr(ws.fused_slab()), ptr(gw), ptr(gb), *_dims(g), stream()), "ksr_gate_pw_bwd_wg")
    else:
        check(L.ksr_gate_pw_bwd(ptr(dz), ptr(p[q + "fn_2.weight"]), ptr(gate_saved), ptr(t), ptr(dt), ptr(dgate),
                                *_dims(g), stream()), "ksr_gate_pw_bwd")
        _wgrad(ws, dz, 64, 0, 64, t, 128, 0, 64, N, gw, gb, a2=gate_saved, a2_ld=64, a2_c0=0)
    grads[q + "fn_2.weight"], grads[q + "fn_2.bias"] = gw, gb
    grads[q + "gate.weight"], grads[q + "gate.bias"] = _dwconv_wgrad(ws, t, 128, 64, dgate, g, q + "gate.weight",
                                                                   q + "gate.bias")
    if stem_saves is None:
        dx = ws.act(N, 64)
        check(L.ksr_fn1_bwd(ptr(dt), ptr(dgate), ptr(p[q + "gate.weight"]), ptr(t), ptr(p[q + "fn_1.0.weight"]), ptr(dx),
                            c_int(0), ptr(None), ptr(None), ptr(None), ptr(None), ptr(None), ptr(None), c_int(0),
                            *_dims(g), stream()), "ksr_fn1_bwd")
        gw, gb = ws.grad(q + "fn_1.0.weight", 128, 64, 1, 1), ws.grad(q + "fn_1.0.bias", 128)
        _wgrad(ws, dt, 128, 0, 128, x_in, 64, 0, 64, N, gw, gb)
        grads[q + "fn_1.0.weight"], grads[q + "fn_1.0.bias"] = gw, gb
        return dx
    # the block input is the first FFN: x1[n, k] = pw2_w[k] * v[n] + pw2_b[k]
    v = stem_saves
    f = pre + "ffn1."
    dpre = ws.new(N)
    # the kernel reduces the column sums of dt and dx (times v) that the stem's parameter gradients need; dx and the
    # gate half of dt, which nothing else reads, are not written
    rc = ws.new(2, 128)
    r2 = ws.grad_pair(f + "pointwise2.weight", f + "pointwise2.bias", 64)
    check(L.ksr_fn1_bwd(ptr(dt), ptr(dgate), ptr(p[q + "gate.weight"]), ptr(t), ptr(p[q + "fn_1.0.weight"]), ptr(None),
                        c_int(1), ptr(v), ptr(p[f + "pointwise2.weight"]), ptr(dpre), ptr(ws.sums_slab()), ptr(rc), ptr(r2),
                        c_int(64), *_dims(g), stream()), "ksr_fn1_bwd")
    return _stem_parameter_gradients(ws, q, p, pre, g, grads, v, dpre, rc, r2, x_img)


def _stem_gated_backward(ws: _Ws, dz, dt, dgate, gw, gb, q, p, pre, g: Geometry, gate_saved, grads, v, ab, x_img):
    """Backward of the first gated block (ffn2) and of ffn1 behind it.  The block's 128-channel tensor t is rebuilt
    inside the kernels from the per-pixel scalar v and the coefficients ab (t = relu(A v + B)); the block input is
    x1[n, k] = pw2_w[k] * v[n] + pw2_b[k]."""
    L = lib()
    N = g.N
    fused = ws.fused_wgrad()
    check(L.ksr_gate_pw_bwd_v(ptr(dz), ptr(p[q + "fn_2.weight"]), ptr(gate_saved), ptr(v), ptr(ab), ptr(dt), ptr(dgate),
                              ptr(ws.fused_slab() if fused else None), ptr(gw if fused else None), ptr(gb if fused else None),
                              *_dims(g), stream()), "ksr_gate_pw_bwd_v")
    assert fused, "the modes without bf16 storage keep t materialised (engine.encoder_forward) and take _gated_backward"
    grads[q + "fn_2.weight"], grads[q + "fn_2.bias"] = gw, gb
    dWg, dbg = ws.grad(q + "gate.weight", 64, 1, 3, 3), ws.grad(q + "gate.bias", 64)
    slab = ws.slab(512 * 640)
    with ws.side(v, ab, dgate, slab, dWg, dbg):
        check(L.ksr_dwconv_wgrad_v(ptr(v), ptr(ab), ptr(dgate), c_int(64), c_int(0), ptr(slab), ptr(dWg), ptr(dbg), *_dims(g),
                                   stream()), "ksr_dwconv_wgrad_v")
    grads[q + "gate.weight"], grads[q + "gate.bias"] = dWg, dbg
    f = pre + "ffn1."
    dpre = ws.new(N)
    # the kernel reduces the column sums of dt and dx (times v) that the stem's parameter gradients need; dx and the
    # gate half of dt, which nothing else reads, are not written
    rc = ws.new(2, 128)
    r2 = ws.grad_pair(f + "pointwise2.weight", f + "pointwise2.bias", 64)
    check(L.ksr_fn1_bwd_stem_v(ptr(dt), ptr(dgate), ptr(p[q + "gate.weight"]), ptr(v), ptr(ab), ptr(p[q + "fn_1.0.weight"]),
                               ptr(p[f + "pointwise2.weight"]), ptr(dpre), ptr(ws.sums_slab()), ptr(rc), ptr(r2), c_int(64),
                               *_dims(g), stream()), "ksr_fn1_bwd_stem_v")
    return _stem_parameter_gradients(ws, q, p, pre, g, grads, v, dpre, rc, r2, x_img)


def _stem_parameter_gradients(ws: _Ws, q, p, pre, g: Geometry, grads, v, dpre, rc, r2, x_img):
    """Parameter gradients of ffn2.fn_1 and ffn1 from the column sums rc / r2 and dpre that the fn_1 backward kernel
    produced (the input of fn_1 is u * v + w with one scalar v per pixel)."""
    L = lib()
    f = pre + "ffn1."
    gw, gb = ws.grad(q + "fn_1.0.weight", 128, 64, 1, 1), ws.grad(q + "fn_1.0.bias", 128)
    with ws.side(rc, gw, gb):
        check(L.ksr_outer2(ptr(gw), c_void_p(rc.data_ptr()), ptr(p[f + "pointwise2.weight"]),
                           c_void_p(rc.data_ptr() + 4 * 128), ptr(p[f + "pointwise2.bias"]),
                           c_int(128), c_int(64), ptr(gb), stream()), "ksr_outer2")
    grads[q + "fn_1.0.weight"], grads[q + "fn_1.0.bias"] = gw, gb
    grads[f + "pointwise2.weight"], grads[f + "pointwise2.bias"] = r2[0].reshape(64, 1, 1, 1), r2[1]
    if ws.arena is None:
        out12 = ws.new(12)
    else:   # the four tensors are adjacent in the flat buffer, in this order
        names12 = [f + "pointwise1.weight", f + "pointwise1.bias", f + "depthwise.weight", f + "depthwise.bias"]
        base = ws.arena[names12[0]]
        off = 0
        for nm in names12:
            assert ws.arena[nm].data_ptr() == base.data_ptr() + 4 * off, nm
            off += ws.arena[nm].numel()
        out12 = torch.as_strided(base, (12,), (1,))
    slab = ws.slab(256 * 12)
    with ws.side(dpre, slab, out12):
        check(L.ksr_stem_scalar_wgrad(ptr(x_img), c_int64(3 * g.H * g.W), ptr(dpre), ptr(p[f + "pointwise1.weight"]),
                                      ptr(p[f + "pointwise1.bias"]), ptr(p[f + "depthwise.weight"]), ptr(slab), ptr(out12),
                                      *_dims(g), stream()), "ksr_stem_scalar_wgrad")
    grads[f + "pointwise1.weight"] = out12[0:1].reshape(1, 1, 1, 1)
    grads[f + "pointwise1.bias"] = out12[1:2]
    grads[f + "depthwise.weight"] = out12[2:11].reshape(1, 1, 3, 3)
    grads[f + "depthwise.bias"] = out12[11:12]
    ws.dpre = dpre            # the input-image gradient (if asked for) continues from here
    return None


def backward_all(gout: torch.Tensor, gout_batch_stride: int, x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str,
                 g: Geometry, sv: dict, save_us: torch.Tensor, p_drop: float, seed: int, arena=None,
                 on_tail_done=None, allow_side_stream: bool = True, want_input_grad: bool = False,
                 gout_chroma=None) -> Dict[str, torch.Tensor]:
    """gout: gradient of the Y plane of the output (images gout_batch_stride elements apart).
    arena: optional name -> view map; gradients are then written in place.  on_tail_done() is called
    once the gradients of feed + both transformer layers + the final norm have been enqueued (the
    first all-reduce bucket), while the conv-trunk backward is still to come.
    want_input_grad: also return the gradient with respect to the input image under the key "__input__"
    (gout_chroma: the whole [B,3,rH,rW] output gradient, whose planes 1, 2 reach the input through the bilinear
    upsampling; None = those planes carry no gradient)."""
    L = lib()
    N = g.N
    ws = _Ws(x.device, arena, g.N, allow_side_stream)      # one stream while a HIP graph is being recorded
    grads: Dict[str, torch.Tensor] = {}
    try:
        # ---- feed + KAN ----
        lane_len = int(L.ksr_feed_kan_lane_len(c_int(g.r), c_int(g.n_coef)))
        gfeat = ws.act(N, 64)
        gparam = ws.act(N, 2 * lane_len)
        # bf16 storage: the final norm's backward runs inside the feed + KAN backward (gfeat = gradient of the norm's INPUT)
        final_norm_done = ws.fused_wgrad() and SPLIT_TOKEN_BACKWARD
        if final_norm_done:
            tx1 = sv["tx1"]
            gnf = ws.grad_pair(pre + "encoder.norm.weight", pre + "encoder.norm.bias", 64)
            check(L.ksr_feed_kan_bwd_lnf(ptr(gout), c_int64(gout_batch_stride), ptr(x), c_int64(3 * g.H * g.W),
                                         ptr(p[pre + "feed.weight"]), ptr(save_us), ptr(gfeat), ptr(gparam), *_dims(g),
                                         c_int(g.r), c_int(g.n_coef), g.knots_c, c_int(len(g.knots)), c_int(g.k),
                                         ptr(tx1["xhatf"]), ptr(tx1["rstdf"]), ptr(p[pre + "encoder.norm.weight"]),
                                         ptr(ws.ln_slab()), ptr(gnf), c_int(64), stream()), "ksr_feed_kan_bwd_lnf")
            grads[pre + "encoder.norm.weight"], grads[pre + "encoder.norm.bias"] = gnf[0], gnf[1]
        else:
            check(L.ksr_feed_kan_bwd(ptr(gout), c_int64(gout_batch_stride), ptr(x),
                                     c_int64(3 * g.H * g.W), ptr(p[pre + "feed.weight"]), ptr(save_us), ptr(gfeat), ptr(gparam),
                                     *_dims(g), c_int(g.r), c_int(g.n_coef), g.knots_c, c_int(len(g.knots)), c_int(g.k),
                                     stream()), "ksr_feed_kan_bwd")
        P = g.r * g.r * (g.n_coef + 2)
        gw, gb = ws.grad(pre + "feed.weight", P, 64, 1, 1), ws.grad(pre + "feed.bias", P)
        _wgrad(ws, gparam, 2 * lane_len, 0, 2 * lane_len, sv["feat"], 64, 0, 64, N, gw, gb, kan=(g.r, g.n_coef))
        grads[pre + "feed.weight"], grads[pre + "feed.bias"] = gw, gb
        trunk_backward(ws, gfeat, x, p, pre, g, sv, p_drop, seed, grads, on_tail_done, final_norm_done)
        ws.close()                # joins the second stream and runs the recorded reductions
        if want_input_grad:
            grads["__input__"] = _input_gradient(ws, gout, gout_batch_stride, gout_chroma, x, p, pre, g, sv)
    except BaseException:
        ws.close(ok=False)
        raise
    return grads


def _input_gradient(ws: _Ws, gout, gout_batch_stride, gout_full, x, p, pre, g: Geometry, sv):
    """Gradient with respect to the input image [B,3,H,W] (conv_sep_kan.py:29-42 under autograd): chroma planes through
    the bilinear upsampling, luma through the KAN evaluation's own use of the pixel value and through the encoder."""
    L = lib()
    gx = torch.empty_like(x)
    plane = g.H * g.W
    if gout_full is not None:
        check(L.ksr_bilinear_up_bwd(ptr(gout_full), ptr(gx), c_int(g.B), c_int(2), c_int(g.H), c_int(g.W), c_int(g.r),
                                    c_int64(3 * plane), c_int(1), c_int64(3 * plane * g.r * g.r), c_int(1), stream()),
              "ksr_bilinear_up_bwd")
    else:
        gx[:, 1:].zero_()
    check(L.ksr_feed_kan_dx(ptr(sv["feat"]), ptr(x), c_int64(3 * plane), ptr(p[pre + "feed.weight"]), ptr(p[pre + "feed.bias"]),
                            ptr(gout), c_int64(gout_batch_stride), ptr(gx), c_int64(3 * plane), *_dims(g), c_int(g.r),
                            c_int(g.n_coef), g.knots_c, c_int(len(g.knots)), c_int(g.k), stream()), "ksr_feed_kan_dx")
    f = pre + "ffn1."
    check(L.ksr_stem_dx(ptr(ws.dpre), ptr(p[f + "pointwise1.weight"]), ptr(p[f + "depthwise.weight"]), ptr(gx),
                        c_int64(3 * plane), *_dims(g), stream()), "ksr_stem_dx")
    return gx


def trunk_backward(ws: _Ws, gfeat: torch.Tensor, x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, g: Geometry,
                   sv: dict, p_drop: float, seed: int, grads: Dict[str, torch.Tensor], on_tail_done=None,
                   final_norm_done: bool = False) -> None:
    """Backward of engine.encoder_forward from gfeat [N,64] (gradient of the final-LayerNormed features, or with
    final_norm_done - the caller ran that norm's backward and filled its two gradients - of the norm's input):
    fills `grads` for every encoder parameter except `feed`."""
    L = lib()
    N = g.N
    # ---- transformer layers ----
    d = _tx_layer_backward(ws, gfeat, 1, p, pre, g, sv["tx1"], p_drop, seed, grads, None, final_norm_done)
    dz6 = _tx_layer_backward(ws, d, 0, p, pre, g, sv["tx0"], p_drop, seed, grads, sv["x6"])
    if on_tail_done is not None:
        ws.flush()                # the first bucket's gradients include second-stream and recorded reductions
        on_tail_done()
    # ---- ffn4 (gated) ----
    dx4 = _gated_backward(ws, dz6, "ffn4", p, pre, g, sv["t5"], sv["g6"], grads, sv["x4"])
    # ---- ffn3 ----
    q = pre + "ffn3."
    ds4 = ws.act(N, 64)
    gw, gb = ws.grad(q + "pointwise2.weight", 64, 64, 1, 1), ws.grad(q + "pointwise2.bias", 64)
    if ws.fused_wgrad():
        check(L.ksr_pw2_bwd_wg(ptr(dx4), ptr(p[q + "pointwise2.weight"]), ptr(sv["a4"]), ptr(ds4), ptr(ws.fused_slab()),
                               ptr(gw), ptr(gb), *_dims(g), stream()), "ksr_pw2_bwd_wg")
    else:
        check(L.ksr_pw2_bwd(ptr(dx4), ptr(p[q + "pointwise2.weight"]), ptr(sv["a4"]), ptr(ds4), *_dims(g), stream()),
              "ksr_pw2_bwd")
        _wgrad(ws, dx4, 64, 0, 64, sv["a4"], 64, 0, 64, N, gw, gb)
    grads[q + "pointwise2.weight"], grads[q + "pointwise2.bias"] = gw, gb
    grads[q + "depthwise.weight"], grads[q + "depthwise.bias"] = _dwconv_wgrad(ws, sv["t3"], 64, 0, ds4, g,
                                                                             q + "depthwise.weight", q + "depthwise.bias")
    dz2 = ws.act(N, 64)
    gw, gb = ws.grad(q + "pointwise1.weight", 64, 64, 1, 1), ws.grad(q + "pointwise1.bias", 64)
    if ws.fused_wgrad():
        # bf16 storage: pointwise1's weight gradient inside the kernel (the conv-transposed gradient dt3 is not written)
        check(L.ksr_pw1_bwd_wg(ptr(ds4), ptr(p[q + "depthwise.weight"]), ptr(p[q + "pointwise1.weight"]), ptr(sv["x3"]),
                               ptr(dz2), ptr(ws.fused_slab()), ptr(gw), ptr(gb), *_dims(g), stream()), "ksr_pw1_bwd_wg")
    else:
        dt3 = ws.act(N, 64)
        check(L.ksr_pw1_bwd(ptr(ds4), ptr(p[q + "depthwise.weight"]), ptr(p[q + "pointwise1.weight"]), ptr(sv["x3"]),
                            ptr(dt3), ptr(dz2), *_dims(g), stream()), "ksr_pw1_bwd")
        _wgrad(ws, dt3, 64, 0, 64, sv["x3"], 64, 0, 64, N, gw, gb)
    grads[q + "pointwise1.weight"], grads[q + "pointwise1.bias"] = gw, gb
    # ---- ffn2 (gated) + ffn1 ----
    _gated_backward(ws, dz2, "ffn2", p, pre, g, sv["t2"], sv["g2"], grads, None, stem_saves=sv["v"], x_img=x, stem_ab=sv["ab"])


class _ConvSepKanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pre, g, p_drop, seed, names, precision, *params):
        p = dict(zip(names, params))
        out = torch.empty(g.B, 3, g.H * g.r, g.W * g.r, device=x.device, dtype=torch.float32)
        bilinear_up_into(x, out, g.r, 1, 2, 1)
        sv: dict = {}
        feat = engine.encoder_forward(x, p, pre, g, sv, p_drop, seed)
        save_us = torch.empty(g.N, g.r * g.r, 2, device=x.device, dtype=torch.float32)
        engine.feed_kan_forward(feat, x, p, pre, g, out, save_us)
        ctx.sv, ctx.save_us, ctx.x, ctx.p = sv, save_us, x, p
        ctx.cfg = (pre, g, p_drop, seed, names, precision)
        return out

    @staticmethod
    def backward(ctx, gout):
        pre, g, p_drop, seed, names, precision = ctx.cfg
        engine.set_precision(precision)
        gout = gout.contiguous()
        want_gx = ctx.needs_input_grad[0]
        grads = backward_all(gout, gout.shape[1] * gout.shape[2] * gout.shape[3], ctx.x, ctx.p, pre, g, ctx.sv,
                             ctx.save_us, p_drop, seed, want_input_grad=want_gx, gout_chroma=gout if want_gx else None)
        ctx.sv = ctx.save_us = None
        out = []
        for n, t in zip(names, ctx.p.values()):
            gr = grads.get(n)
            out.append(None if gr is None else gr.reshape(t.shape))
        return (grads.get("__input__"), None, None, None, None, None, None, *out)


def conv_sep_kan_train(x, p: Dict[str, torch.Tensor], pre: str, g: Geometry, p_drop: float, seed: int,
                        precision: str = "fp32"):
    names = [pre + n for n in PARAM_ORDER]
    li_prefix = pre[: -len("encoder.encoder.")]
    # p is keyed "layers.L.encoder.encoder.<name>"
    params = [p[n] for n in names]
    for t in params:
        if not t.is_contiguous():
            raise KsrError("parameters must be contiguous")
    return _ConvSepKanFn.apply(x, pre, g, p_drop, seed, names, precision, *params)


class _TrunkFn(torch.autograd.Function):
    """Encoder trunk alone (everything before `feed`): x [B,3,H,W] -> features [N,64], for heads other
    than the fused feed+KAN kernel."""

    @staticmethod
    def forward(ctx, x, pre, g, p_drop, seed, names, precision, sink, *params):
        p = dict(zip(names, params))
        sv: dict = {}
        feat = engine.encoder_forward(x, p, pre, g, sv, p_drop, seed)
        ctx.sv, ctx.x, ctx.p = sv, x, p
        ctx.cfg = (pre, g, p_drop, seed, names, precision, sink)
        return feat

    @staticmethod
    def backward(ctx, gfeat):
        pre, g, p_drop, seed, names, precision, sink = ctx.cfg
        engine.set_precision(precision)
        grads: Dict[str, torch.Tensor] = {}
        ws = _Ws(ctx.x.device, sink, g.N)
        try:
            trunk_backward(ws, gfeat.contiguous(), ctx.x, ctx.p, pre, g, ctx.sv, p_drop, seed, grads)
            gx = None
            if ctx.needs_input_grad[0]:
                # the trunk reads the luma plane only (ffn1's front): w1 * convT(dpre) into plane 0
                gx = torch.zeros_like(ctx.x)
                f = pre + "ffn1."
                check(lib().ksr_stem_dx(ptr(ws.dpre), ptr(ctx.p[f + "pointwise1.weight"]), ptr(ctx.p[f + "depthwise.weight"]), ptr(gx),
                                        c_int64(3 * g.H * g.W), *_dims(g), stream()), "ksr_stem_dx")
            ws.close()
        except BaseException:
            ws.close(ok=False)
            raise
        ctx.sv = None
        out = []
        for n, t in zip(names, ctx.p.values()):
            gr = grads.get(n)
            # with a sink the kernels have written the gradient where the caller wants it: nothing for autograd to add
            out.append(None if gr is None or sink is not None else gr.reshape(t.shape))
        return (gx, None, None, None, None, None, None, None, *out)


def trunk_train(x, p: Dict[str, torch.Tensor], pre: str, g: Geometry, p_drop: float, seed: int, precision: str = "fp32",
                sink: Optional[Dict[str, torch.Tensor]] = None):
    """sink: optional name -> tensor map covering every trunk parameter (adjacent in memory in PARAM_ORDER, as in a flat
    buffer laid out in module order); the backward writes the parameter gradients there instead of returning them."""
    names = [pre + n for n in PARAM_ORDER if not n.startswith("feed.")]
    params = [p[n] for n in names]
    for t in params:
        if not t.is_contiguous():
            raise KsrError("parameters must be contiguous")
    if sink is not None and any(n not in sink for n in names):
        raise KsrError("trunk_train: the gradient sink must name every trunk parameter")
    return _TrunkFn.apply(x, pre, g, p_drop, seed, names, precision, sink, *params)
