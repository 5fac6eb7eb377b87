"""ConvSepKanAttention on the HIP path (reference: conv_sep_kan/conv_sep_kan_attention.py:9-71).

Same constructor, parameter names and forward signature as the reference module.  The forward and
backward are sequences of libksr kernels over NHWC pixels; PyTorch only owns the buffers, carries the
tape nodes and does the NCHW <-> NHWC relayout at the module boundary.  Only in = mid = 64 channels and
out in {40, 64, 160, 336} have kernels; anything else raises KsrError.
"""
from __future__ import annotations

from ctypes import c_int, c_int64
from typing import Sequence

import torch
import torch.nn as nn

from . import engine
from ._lib import KsrError, check, host_floats, lib, ptr, stream

SUPPORTED_OUT = (40, 64, 160, 336)


def _new(dev, *shape):
    return torch.empty(*shape, device=dev, dtype=torch.float32)


def _act(dev, *shape):
    """[N, C] activation (or activation-gradient) tensor in the storage type of the current precision mode: bfloat16 in
    the bf16-storage mode, which forward_kan's chain of kernels supports."""
    return torch.empty(*shape, device=dev, dtype=engine.act_dtype())


def _gram(dz, dz_c0, J, a, a_c0, HW, B, want_sq):
    """Per-image sum over pixels of dz[:, dz_c0:dz_c0+J]^T a[:, a_c0:a_c0+64]; returns (out [B,J,64], tail)."""
    L = lib()
    dev = dz.device
    slab = _new(dev, max(1, int(L.ksr_gram_slab_floats(c_int(J), c_int(64), c_int(B), c_int64(HW)))))
    out = _new(dev, B, J, 64)
    tail = _new(dev, B, 2 * J + 64 if want_sq else J)
    check(L.ksr_gram_batched(ptr(dz), c_int(dz.shape[1]), c_int(dz_c0), c_int(J), ptr(a), c_int(a.shape[1]), c_int(a_c0),
                             c_int(64), c_int64(HW), c_int(B), ptr(slab), ptr(out), ptr(tail), c_int(int(want_sq)),
                             stream()), "ksr_gram_batched")
    return out, tail


# Layout of the q | k | v branches' parameters in one buffer: [192][64] projection weights, [192] biases, [192][9] depthwise
# weights, [192] depthwise biases (offsets in floats; every block starts on a 16-byte boundary)
_QKV_CAT = (0, 192 * 64, 192 * 64 + 192, 192 * 64 + 192 + 192 * 9, 192 * 64 + 192 + 192 * 9 + 192)
_QKV_ROWS = (64 * 64, 64, 64 * 9, 64)          # floats per branch in each block


def _qkv_views(buf):
    o = _QKV_CAT
    return buf[o[0]:o[1]].view(192, 64), buf[o[1]:o[2]], buf[o[2]:o[3]].view(192, 9), buf[o[3]:o[4]]


def _qkv_pieces(buf):
    """The twelve per-parameter pieces of such a buffer, in the order of ConvSepKanAttention._qkv_params()."""
    return [buf[_QKV_CAT[k] + i * n:_QKV_CAT[k] + (i + 1) * n] for k, n in enumerate(_QKV_ROWS) for i in range(3)]


# bf16 storage: ksr_chan_dw_bwd and ksr_pointwise_bwd of the q | k | v branches as one kernel (False: the two launches)
FUSED_DW_BWD_DX = True


class _QkvFn(torch.autograd.Function):
    """qkv [N,192] = dw3x3(pw(x)) for the three branches at once (weights concatenated q | k | v)."""

    @staticmethod
    def forward(ctx, x, w, b, dw_w, dw_b, dims, precision, sink=None):
        B, H, W = dims
        engine.set_precision(precision)
        L = lib()
        N = x.shape[0]
        ctx.sink = sink
        pq = _act(x.device, N, 192)
        check(L.ksr_pointwise_fwd(ptr(x), ptr(w), ptr(b), ptr(pq), c_int(192), c_int(0), c_int(B), c_int(H), c_int(W),
                                  stream()), "ksr_pointwise_fwd")
        qkv = _act(x.device, N, 192)
        check(L.ksr_chan_dw_fwd(ptr(pq), ptr(dw_w), ptr(dw_b), ptr(qkv), c_int(192), c_int(3), c_int(B), c_int(H),
                                c_int(W), stream()), "ksr_chan_dw_fwd")
        ctx.save_for_backward(x, w, dw_w, pq)
        ctx.cfg = (dims, precision)
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        x, w, dw_w, pq = ctx.saved_tensors
        (B, H, W), precision = ctx.cfg
        engine.set_precision(precision)
        L = lib()
        dev = x.device
        N = x.shape[0]
        dqkv = dqkv.contiguous()
        dpq = _act(dev, N, 192)
        # bf16 storage: the projections' data gradient in the epilogue of the transposed depthwise conv (dpq is not read again
        # for it)
        dx = None
        fused_dx = FUSED_DW_BWD_DX and ctx.needs_input_grad[0] and dpq.dtype is torch.bfloat16
        if fused_dx:
            dx = _act(dev, N, 64)
            check(L.ksr_chan_dw_bwd_dx(ptr(dqkv), ptr(dw_w), ptr(w), ptr(dpq), ptr(dx), c_int(B), c_int(H), c_int(W), stream()),
                  "ksr_chan_dw_bwd_dx")
        else:
            check(L.ksr_chan_dw_bwd(ptr(dqkv), ptr(dw_w), ptr(dpq), c_int(192), c_int(3), c_int(B), c_int(H), c_int(W),
                                    stream()), "ksr_chan_dw_bwd")
        sink = ctx.sink
        if sink is not None:
            # the four concatenated gradients in one buffer, handed to the twelve parameters' tensors by one multi-tensor copy
            gcat = _new(dev, _QKV_CAT[-1])
            g_w, g_b, g_dw_w, g_dw_b = _qkv_views(gcat)
        else:
            g_w, g_b = _new(dev, 192, 64), _new(dev, 192)
            g_dw_w, g_dw_b = _new(dev, 192, 9), _new(dev, 192)
        slab = _new(dev, max(3 * 512 * 640, int(L.ksr_wgrad_slab_floats(c_int(192), c_int(64)))))
        # the three branches' depthwise weight gradients in one launch (channel groups q | k | v of the same two tensors)
        check(L.ksr_dwconv_wgrad_groups(ptr(pq), c_int(192), c_int(0), ptr(dqkv), c_int(192), c_int(0), c_int(3), ptr(slab),
                                        ptr(g_dw_w), ptr(g_dw_b), c_int(B), c_int(H), c_int(W), stream()), "ksr_dwconv_wgrad_groups")
        check(L.ksr_wgrad(ptr(dpq), c_int(192), c_int(0), c_int(192), ptr(x), c_int(64), c_int(0), c_int(64),
                          ptr(None), c_int(0), c_int(0), c_int64(N), ptr(slab), ptr(g_w), ptr(g_b), c_int(0), c_int(0),
                          stream()), "ksr_wgrad")
        if ctx.needs_input_grad[0] and not fused_dx:
            dx = _act(dev, N, 64)
            check(L.ksr_pointwise_bwd(ptr(dpq), c_int(192), ptr(w), ptr(dx), c_int(B), c_int(H), c_int(W), stream()),
                  "ksr_pointwise_bwd")
        if sink is not None:
            torch._foreach_copy_(sink, _qkv_pieces(gcat))
            return dx, None, None, None, None, None, None, None
        return dx, g_w, g_b, g_dw_w, g_dw_b, None, None, None


class _AttnCoreFn(torch.autograd.Function):
    """y [N,P] = a_proj(softmax(q^^T k^) v) from qkv [N,192]."""

    @staticmethod
    def forward(ctx, qkv, wa, ba, dims, precision):
        B, H, W = dims
        HW = H * W
        engine.set_precision(precision)
        L = lib()
        dev = qkv.device
        P = wa.shape[0]
        G, tail = _gram(qkv, 0, 64, qkv, 64, HW, B, True)
        A, S, nrm, M = _new(dev, B, 64, 64), _new(dev, B, 64, 64), _new(dev, B, 128), _new(dev, B, P, 64)
        check(L.ksr_chan_soft_fwd(ptr(G), ptr(tail), ptr(wa), ptr(A), ptr(S), ptr(nrm), ptr(M), c_int(P), c_int(B),
                                  stream()), "ksr_chan_soft_fwd")
        y = _new(dev, qkv.shape[0], P)
        check(L.ksr_chan_apply(ptr(qkv), c_int(192), c_int(128), c_int(64), ptr(M), c_int(0), ptr(ba), ptr(y), c_int(P),
                               c_int(0), c_int(P), c_int(B), c_int(HW), stream()), "ksr_chan_apply")
        ctx.save_for_backward(qkv, wa, A, S, nrm, M)
        ctx.cfg = (dims, precision)
        return y

    @staticmethod
    def backward(ctx, dy):
        qkv, wa, A, S, nrm, M = ctx.saved_tensors
        (B, H, W), precision = ctx.cfg
        HW = H * W
        engine.set_precision(precision)
        L = lib()
        dev = qkv.device
        P = wa.shape[0]
        dy = dy.contiguous()
        Gd, colsum = _gram(dy, 0, P, qkv, 128, HW, B, False)
        g_ba = _new(dev, P)
        check(L.ksr_sum_entries(ptr(colsum), c_int(B), c_int64(P), c_int(P), ptr(g_ba), stream()), "ksr_sum_entries")
        Mqk, part, g_wa = _new(dev, B, 128, 128), _new(dev, B, P, 64), _new(dev, P, 64)
        check(L.ksr_chan_soft_bwd(ptr(Gd), ptr(A), ptr(S), ptr(nrm), ptr(wa), ptr(Mqk), ptr(part), ptr(g_wa), c_int(P),
                                  c_int(B), stream()), "ksr_chan_soft_bwd")
        dqkv = _new(dev, qkv.shape[0], 192)
        check(L.ksr_chan_apply(ptr(dy), c_int(P), c_int(0), c_int(P), ptr(M), c_int(1), ptr(None), ptr(dqkv), c_int(192),
                               c_int(128), c_int(64), c_int(B), c_int(HW), stream()), "ksr_chan_apply")
        check(L.ksr_chan_apply(ptr(qkv), c_int(192), c_int(0), c_int(128), ptr(Mqk), c_int(0), ptr(None), ptr(dqkv),
                               c_int(192), c_int(0), c_int(128), c_int(B), c_int(HW), stream()), "ksr_chan_apply")
        return dqkv, g_wa, g_ba, None, None


class _AttnKanFn(torch.autograd.Function):
    """_AttnCoreFn followed by the KAN evaluation, pixel shuffle and chroma upsampling (conv_sep_kan.py:29-42 with the
    attention head producing the per-pixel parameters), without the [N, P] parameter tensor: the head's last step is a 1x1
    map with one matrix per image, M_b = W_a softmax(q^ k^T), which the fused feed+KAN kernel takes as per-image weights."""

    @staticmethod
    def forward(ctx, qkv, wa, ba, x, dims, precision, n_coef, knots, spline_order, r, sink=None):
        B, H, W = dims
        HW = H * W
        ctx.sink = sink
        engine.set_precision(precision)
        L = lib()
        dev = qkv.device
        P = wa.shape[0]
        G, tail = _gram(qkv, 0, 64, qkv, 64, HW, B, True)
        A, S, nrm, M = _new(dev, B, 64, 64), _new(dev, B, 64, 64), _new(dev, B, 128), _new(dev, B, P, 64)
        check(L.ksr_chan_soft_fwd(ptr(G), ptr(tail), ptr(wa), ptr(A), ptr(S), ptr(nrm), ptr(M), c_int(P), c_int(B),
                                  stream()), "ksr_chan_soft_fwd")
        out = _new(dev, B, 3, H * r, W * r)
        from .ops import bilinear_up_into
        bilinear_up_into(x, out, r, 1, 2, 1)
        save_us = _new(dev, B * HW, r * r, 2)
        kn = host_floats(knots)
        check(L.ksr_feed_kan_fwd_img(ptr(qkv), c_int(192), c_int(128), ptr(x), c_int64(3 * HW), ptr(M), ptr(ba), ptr(out),
                                     c_int64(3 * HW * r * r), ptr(save_us), c_int(B), c_int(H), c_int(W), c_int(r), c_int(n_coef),
                                     kn, c_int(len(knots)), c_int(spline_order), stream()), "ksr_feed_kan_fwd_img")
        ctx.save_for_backward(qkv, wa, A, S, nrm, M, save_us, x, ba)
        ctx.cfg = (dims, precision, n_coef, knots, spline_order, r)
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, wa, A, S, nrm, M, save_us, x, ba = ctx.saved_tensors
        (B, H, W), precision, n_coef, knots, spline_order, r = ctx.cfg
        HW = H * W
        engine.set_precision(precision)
        L = lib()
        dev = qkv.device
        P = wa.shape[0]
        gout = gout.contiguous()
        gx = None
        if ctx.needs_input_grad[3]:
            # the image's own paths into the output: chroma through the bilinear upsampling, luma through the KAN
            # evaluation's use of the pixel value (its path through the encoder trunk is the trunk function's gradient)
            gx = torch.empty_like(x)
            check(L.ksr_bilinear_up_bwd(ptr(gout), ptr(gx), c_int(B), c_int(2), c_int(H), c_int(W), c_int(r), c_int64(3 * HW),
                                        c_int(1), c_int64(3 * HW * r * r), c_int(1), stream()), "ksr_bilinear_up_bwd")
            check(L.ksr_feed_kan_dx_img(ptr(qkv), c_int(192), c_int(128), ptr(x), c_int64(3 * HW), ptr(M), ptr(ba), ptr(gout),
                                        c_int64(3 * HW * r * r), ptr(gx), c_int64(3 * HW), c_int(B), c_int(H), c_int(W), c_int(r),
                                        c_int(n_coef), host_floats(knots), c_int(len(knots)), c_int(spline_order), stream()),
                  "ksr_feed_kan_dx_img")
        J = 2 * int(L.ksr_feed_kan_lane_len(c_int(r), c_int(n_coef)))
        gparam = _act(dev, B * HW, J)
        dqkv = _act(dev, B * HW, 192)
        kn = host_floats(knots)
        # gradient of the per-pixel parameters (lane order) and, through M_b, of v: written into the v third of dqkv
        check(L.ksr_feed_kan_bwd_img(ptr(gout), c_int64(3 * HW * r * r), ptr(x), c_int64(3 * HW), ptr(M), ptr(save_us), ptr(dqkv),
                                     c_int(192), c_int(128), ptr(gparam), c_int(B), c_int(H), c_int(W), c_int(r), c_int(n_coef),
                                     kn, c_int(len(knots)), c_int(spline_order), stream()), "ksr_feed_kan_bwd_img")
        # per-image dM_b = dparam^T v and the column sums of dparam (bias gradient), rows back in channel order
        slab = _new(dev, max(1, int(L.ksr_gram_slab_floats(c_int(J), c_int(64), c_int(B), c_int64(HW)))))
        Gd, colsum = _new(dev, B, P, 64), _new(dev, B, J)
        check(L.ksr_gram_batched_kan(ptr(gparam), c_int(J), ptr(qkv), c_int(192), c_int(128), c_int(64), c_int64(HW), c_int(B),
                                     ptr(slab), ptr(Gd), ptr(colsum), c_int(r), c_int(n_coef), stream()), "ksr_gram_batched_kan")
        sink = ctx.sink                         # (gradient tensors of a_proj's weight [P,64] and bias [P]) or None
        g_ba = _new(dev, P) if sink is None else sink[1]
        check(L.ksr_sum_entries(ptr(colsum), c_int(B), c_int64(J), c_int(P), ptr(g_ba), stream()), "ksr_sum_entries")
        Mqk, part = _new(dev, B, 128, 128), _new(dev, B, P, 64)
        g_wa = _new(dev, P, 64) if sink is None else sink[0]
        check(L.ksr_chan_soft_bwd(ptr(Gd), ptr(A), ptr(S), ptr(nrm), ptr(wa), ptr(Mqk), ptr(part), ptr(g_wa), c_int(P),
                                  c_int(B), stream()), "ksr_chan_soft_bwd")
        check(L.ksr_chan_apply(ptr(qkv), c_int(192), c_int(0), c_int(128), ptr(Mqk), c_int(0), ptr(None), ptr(dqkv),
                               c_int(192), c_int(0), c_int(128), c_int(B), c_int(HW), stream()), "ksr_chan_apply")
        if sink is not None:
            g_wa = g_ba = None
        return dqkv, g_wa, g_ba, gx, None, None, None, None, None, None, None


def _to_nhwc(t):
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


class ConvSepKanAttention(nn.Module):
    """Drop-in for the reference class of the same name; `precision` ("fp32" | "bf16"; "bf16s" = bf16 activation storage,
    for forward_kan only) is the only addition."""

    def __init__(self, in_channels: int, mid_channels: int, out_channels: int):
        super().__init__()
        for n in ("q", "k", "v"):
            setattr(self, n + "_proj", nn.Conv2d(in_channels, mid_channels, kernel_size=1))
            setattr(self, n + "_dw", nn.Conv2d(mid_channels, mid_channels, kernel_size=3, stride=1, padding=1,
                                               groups=mid_channels))
        self.a_proj = nn.Conv2d(mid_channels, out_channels, kernel_size=1)
        # defined by the reference (conv_sep_kan_attention.py:41-46) and never used by its forward
        self.a_dw = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, groups=out_channels)
        self.in_channels, self.mid_channels, self.out_channels = in_channels, mid_channels, out_channels
        self.precision = "fp32"
        # optional name -> tensor map (this module's parameter names): forward_kan's backward then writes the parameter
        # gradients into those tensors and reports none to autograd (TrainStep: views of its flat gradient buffer)
        self.grad_sink = None

    def _qkv_params(self):
        """The q | k | v branches' parameters, kind by kind: projection weights, biases, depthwise weights, depthwise biases."""
        return [getattr(getattr(self, n + m), kind) for m, kind in (("_proj", "weight"), ("_proj", "bias"), ("_dw", "weight"),
                                                                    ("_dw", "bias")) for n in "qkv"]

    def forward_nhwc(self, x, dims, q=None, k=None, v=None):
        """x [N,64] (N = B*H*W pixels, NHWC order) -> [N,out]; q, k, v optional precomputed [N,64] branches."""
        if self.in_channels != 64 or self.mid_channels != 64 or self.out_channels not in SUPPORTED_OUT:
            raise KsrError(f"ConvSepKanAttention({self.in_channels}, {self.mid_channels}, {self.out_channels}): the HIP "
                           f"path has kernels for in = mid = 64 and out in {SUPPORTED_OUT}")
        if not x.is_cuda or x.dtype != torch.float32:
            raise KsrError("ConvSepKanAttention needs fp32 GPU tensors (the HIP path is the only path)")
        w = torch.cat([getattr(self, n + "_proj").weight.reshape(64, 64) for n in "qkv"])
        b = torch.cat([getattr(self, n + "_proj").bias for n in "qkv"])
        dw_w = torch.cat([getattr(self, n + "_dw").weight.reshape(64, 9) for n in "qkv"])
        dw_b = torch.cat([getattr(self, n + "_dw").bias for n in "qkv"])
        qkv = _QkvFn.apply(x.contiguous(), w, b, dw_w, dw_b, dims, self.precision)
        if q is not None or k is not None or v is not None:
            parts = [qkv[:, 0:64] if q is None else q, qkv[:, 64:128] if k is None else k,
                     qkv[:, 128:192] if v is None else v]
            qkv = torch.cat(parts, dim=1)
        return _AttnCoreFn.apply(qkv, self.a_proj.weight.reshape(self.out_channels, 64), self.a_proj.bias, dims,
                                 self.precision)

    def _qkv(self, x, dims):
        sink = self.grad_sink
        if sink is not None and torch.is_grad_enabled():
            # one concatenation for the four kernel operands; the gradients go straight to the sink
            with torch.no_grad():
                w, b, dw_w, dw_b = _qkv_views(torch.cat([t.reshape(-1) for t in self._qkv_params()]))
            names = [n + m + "." + kind for m, kind in (("_proj", "weight"), ("_proj", "bias"), ("_dw", "weight"), ("_dw", "bias"))
                     for n in "qkv"]
            return _QkvFn.apply(x.contiguous(), w, b, dw_w, dw_b, dims, self.precision, [sink[n].view(-1) for n in names])
        w = torch.cat([getattr(self, n + "_proj").weight.reshape(64, 64) for n in "qkv"])
        b = torch.cat([getattr(self, n + "_proj").bias for n in "qkv"])
        dw_w = torch.cat([getattr(self, n + "_dw").weight.reshape(64, 9) for n in "qkv"])
        dw_b = torch.cat([getattr(self, n + "_dw").bias for n in "qkv"])
        return _QkvFn.apply(x.contiguous(), w, b, dw_w, dw_b, dims, self.precision)

    def forward_kan(self, feat, dims, x_img, n_coef: int, knots: Sequence[float], spline_order: int, r: int):
        """Head + KAN evaluation + pixel shuffle + chroma in one tape node chain: feat [N,64] (NHWC pixels of x_img
        [B,3,H,W]) -> [B,3,rH,rW].  The [N, out] tensor of per-pixel KAN parameters is never formed (see _AttnKanFn).
        Needs a fused feed+KAN kernel for (r, n_coef, spline_order): ksr_feed_kan_supported."""
        if self.in_channels != 64 or self.mid_channels != 64 or self.out_channels != r * r * (n_coef + 2):
            raise KsrError(f"ConvSepKanAttention({self.in_channels}, {self.mid_channels}, {self.out_channels}) does not produce "
                           f"the parameters of a KAN cell with upscale {r} and {n_coef} coefficients")
        want = torch.bfloat16 if self.precision == "bf16s" else torch.float32
        if not feat.is_cuda or feat.dtype != want:
            raise KsrError(f"ConvSepKanAttention(precision={self.precision!r}) needs {want} GPU features (the HIP path is the "
                           "only path)")
        qkv = self._qkv(feat, dims)
        sink = self.grad_sink
        if sink is not None and torch.is_grad_enabled():
            return _AttnKanFn.apply(qkv, self.a_proj.weight.detach().reshape(self.out_channels, 64), self.a_proj.bias.detach(),
                                    x_img.contiguous(), dims, self.precision, n_coef, tuple(knots), spline_order, r,
                                    (sink["a_proj.weight"].view(self.out_channels, 64), sink["a_proj.bias"]))
        return _AttnKanFn.apply(qkv, self.a_proj.weight.reshape(self.out_channels, 64), self.a_proj.bias, x_img.contiguous(),
                                dims, self.precision, n_coef, tuple(knots), spline_order, r)

    def forward(self, x, q=None, k=None, v=None):
        B, C, H, W = x.shape
        y = self.forward_nhwc(_to_nhwc(x), (B, H, W), *(None if t is None else _to_nhwc(t) for t in (q, k, v)))
        return y.reshape(B, H, W, self.out_channels).permute(0, 3, 1, 2).contiguous()
