"""torch.autograd.Function wrappers around the libksr C ABI.

PyTorch is plumbing here: it owns device memory, streams and the autograd tape; every
forward and backward below is one or more hand-written HIP kernels.
"""
from __future__ import annotations

from ctypes import c_int, c_int64
from typing import Sequence

import torch

from . import _lib
from ._lib import check, host_floats, lib, ptr, stream


def make_knots(grid_size: int, spline_order: int, grid_range: Sequence[float]) -> list:
    """fp32 knots t_j = float32(j)*h + lo, j=-k..G+k (conv_sep_kan/kan_layer.py:23-25): the same
    expression in the same precision, so interval membership matches the reference bit for bit."""
    lo, hi = grid_range[0], grid_range[1]
    h = (hi - lo) / grid_size
    t = torch.arange(-spline_order, grid_size + spline_order + 1) * h + lo
    return [float(v) for v in t.to(torch.float32)]


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _lib.KsrError(f"expected float32, got {t.dtype}")
    return t.contiguous()


class _KanCell(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, out_dim, n_coef, knots, spline_order, shuffle_r, w_nhwc):
        x, w = _f32c(x), _f32c(w)
        B, I, H, W = x.shape
        O = out_dim
        if shuffle_r:
            out = torch.empty(B, O // (shuffle_r * shuffle_r), H * shuffle_r, W * shuffle_r,
                              device=x.device, dtype=torch.float32)
        else:
            out = torch.empty(B, O, H, W, device=x.device, dtype=torch.float32)
        kn = host_floats(knots)
        check(lib().ksr_kan_cell_fwd(ptr(x), ptr(w), ptr(out), c_int(B), c_int(I), c_int(O), c_int(H),
                                     c_int(W), c_int(n_coef), kn, c_int(len(knots)), c_int(spline_order),
                                     c_int(int(w_nhwc)), c_int(shuffle_r), c_int64(0), stream()),
              "ksr_kan_cell_fwd")
        ctx.save_for_backward(x, w)
        ctx.cfg = (O, n_coef, knots, spline_order, shuffle_r, w_nhwc)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, w = ctx.saved_tensors
        O, n_coef, knots, spline_order, shuffle_r, w_nhwc = ctx.cfg
        B, I, H, W = x.shape
        gout = _f32c(gout)
        gw = torch.empty_like(w)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        kn = host_floats(knots)
        check(lib().ksr_kan_cell_bwd(ptr(x), ptr(w), ptr(gout), ptr(gw), ptr(gx), c_int(B), c_int(I),
                                     c_int(O), c_int(H), c_int(W), c_int(n_coef), kn, c_int(len(knots)),
                                     c_int(spline_order), c_int(int(w_nhwc)), c_int(shuffle_r), c_int64(0),
                                     stream()),
              "ksr_kan_cell_bwd")
        return gx, gw, None, None, None, None, None, None


def kan_cell(x: torch.Tensor, w: torch.Tensor, out_dim: int, grid_size: int, spline_order: int,
             grid_range: Sequence[float], shuffle_r: int = 0, w_nhwc: bool = False) -> torch.Tensor:
    """x [B,I,H,W], w [B,P,H,W] (or [B*H*W,P] with w_nhwc) -> [B,O,H,W] (or pixel-shuffled)."""
    knots = make_knots(grid_size, spline_order, grid_range)
    return _KanCell.apply(x, w, out_dim, grid_size + spline_order, knots, spline_order, shuffle_r, w_nhwc)


class _KanHead(torch.autograd.Function):
    """Output assembly of ConvSepKan.forward (conv_sep_kan/conv_sep_kan.py:28-41) from a materialised
    per-pixel parameter tensor w [B*H*W, P]: Y' = PixelShuffle(KAN(Y; w)) into plane 0, bilinear Cb/Cr
    into planes 1, 2 of one [B,3,rH,rW] tensor."""

    @staticmethod
    def forward(ctx, x, w, n_coef, knots, spline_order, r):
        x, w = _f32c(x), _f32c(w)
        B, _, H, W = x.shape
        out = torch.empty(B, 3, H * r, W * r, device=x.device, dtype=torch.float32)
        bilinear_up_into(x, out, r, 1, 2, 1)
        y = x[:, :1].contiguous()
        kn = host_floats(knots)
        check(lib().ksr_kan_cell_fwd(ptr(y), ptr(w), ptr(out), c_int(B), c_int(1), c_int(r * r), c_int(H), c_int(W),
                                     c_int(n_coef), kn, c_int(len(knots)), c_int(spline_order), c_int(1), c_int(r),
                                     c_int64(3 * H * r * W * r), stream()), "ksr_kan_cell_fwd")
        ctx.save_for_backward(y, w)
        ctx.cfg = (n_coef, knots, spline_order, r)
        return out

    @staticmethod
    def backward(ctx, gout):
        y, w = ctx.saved_tensors
        n_coef, knots, spline_order, r = ctx.cfg
        B, _, H, W = y.shape
        gout = _f32c(gout)
        gw = torch.empty_like(w)
        kn = host_floats(knots)
        want_gx = ctx.needs_input_grad[0]
        gy = torch.empty_like(y) if want_gx else None       # luma: the cell's own use of the pixel value
        check(lib().ksr_kan_cell_bwd(ptr(y), ptr(w), ptr(gout), ptr(gw), ptr(gy), c_int(B), c_int(1), c_int(r * r),
                                     c_int(H), c_int(W), c_int(n_coef), kn, c_int(len(knots)), c_int(spline_order),
                                     c_int(1), c_int(r), c_int64(3 * H * r * W * r), stream()), "ksr_kan_cell_bwd")
        gx = None
        if want_gx:
            gx = torch.empty(B, 3, H, W, device=y.device, dtype=torch.float32)
            check(lib().ksr_bilinear_up_bwd(ptr(gout), ptr(gx), c_int(B), c_int(2), c_int(H), c_int(W), c_int(r),
                                            c_int64(3 * H * W), c_int(1), c_int64(3 * H * r * W * r), c_int(1), stream()),
                  "ksr_bilinear_up_bwd")
            gx[:, :1].copy_(gy)
        return gx, gw, None, None, None, None


def kan_head(x: torch.Tensor, w: torch.Tensor, grid_size: int, spline_order: int, grid_range: Sequence[float],
             r: int) -> torch.Tensor:
    """x [B,3,H,W], w [B*H*W, r*r*(G+k+2)] -> [B,3,rH,rW]."""
    knots = make_knots(grid_size, spline_order, grid_range)
    return _KanHead.apply(x, w, grid_size + spline_order, knots, spline_order, r)


class _KanLinHead(torch.autograd.Function):
    """_KanHead with the parameter tensor never materialised: w = OutConv(y4) (the UNet's last 1x1 conv, 4 -> P) is
    computed inside the KAN-cell kernels.  y4 [B*H*W, 4] is the UNet's last activation; the conv's weight and bias are read
    from `net.reproj`, and their gradients are left in the UNet's kernel-layout gradient buffer (net._KG["reproj"],
    rows: weight^T | bias), from where the UNet's own backward hands them on with all its other parameter gradients."""

    @staticmethod
    def forward(ctx, x, y4, net, n_coef, knots, spline_order, r):
        x, y4 = _f32c(x), _f32c(y4)
        B, _, H, W = x.shape
        P = net.out_channels
        lw, lb = net.reproj.conv.weight.detach().view(P, 4), net.reproj.conv.bias.detach()
        out = torch.empty(B, 3, H * r, W * r, device=x.device, dtype=torch.float32)
        bilinear_up_into(x, out, r, 1, 2, 1)
        y = x[:, :1].contiguous()
        kn = host_floats(knots)
        check(lib().ksr_kan_lin_fwd(ptr(y), ptr(y4), ptr(lw), ptr(lb), ptr(out), c_int(B), c_int(r * r), c_int(H), c_int(W),
                                    c_int(n_coef), kn, c_int(len(knots)), c_int(spline_order), c_int(r),
                                    c_int64(3 * H * r * W * r), stream()), "ksr_kan_lin_fwd")
        ctx.save_for_backward(y, y4)
        ctx.net, ctx.cfg = net, (n_coef, knots, spline_order, r)
        return out

    @staticmethod
    def backward(ctx, gout):
        y, y4 = ctx.saved_tensors
        n_coef, knots, spline_order, r = ctx.cfg
        net = ctx.net
        B, _, H, W = y.shape
        if ctx.needs_input_grad[0]:
            raise _lib.KsrError("gradient with respect to the input image is not implemented")
        gout = _f32c(gout)
        P = net.out_channels
        lw, lb = net.reproj.conv.weight.detach().view(P, 4), net.reproj.conv.bias.detach()
        net._ensure_emit_plan(B)
        dy4 = torch.empty_like(y4)
        slab = torch.empty(int(lib().ksr_kan_lin_slab_floats(c_int(B), c_int(H), c_int(W), c_int(r * r), c_int(n_coef))),
                           device=y.device, dtype=torch.float32)
        kn = host_floats(knots)
        check(lib().ksr_kan_lin_bwd(ptr(y), ptr(y4), ptr(lw), ptr(lb), ptr(gout), ptr(dy4), ptr(slab), ptr(net._KG["reproj"]),
                                    c_int(B), c_int(r * r), c_int(H), c_int(W), c_int(n_coef), kn, c_int(len(knots)),
                                    c_int(spline_order), c_int(r), c_int64(3 * H * r * W * r), stream()), "ksr_kan_lin_bwd")
        return None, dy4, None, None, None, None, None


def kan_lin_head(x: torch.Tensor, y4: torch.Tensor, net, grid_size: int, spline_order: int, grid_range: Sequence[float],
                 r: int) -> torch.Tensor:
    """x [B,3,H,W], y4 [B*H*W, 4] (the input of net.reproj) -> [B,3,rH,rW]."""
    knots = make_knots(grid_size, spline_order, grid_range)
    return _KanLinHead.apply(x, y4, net, grid_size + spline_order, knots, spline_order, r)


def bilinear_up_into(x: torch.Tensor, out: torch.Tensor, r: int, in_channel0: int, n_channels: int,
                     out_channel0: int) -> None:
    """Upsample planes [in_channel0, in_channel0+n) of x [B,C,H,W] into planes out_channel0.. of
    out [B,C',rH,rW] (no gradient: the chroma planes are inputs)."""
    x = _f32c(x)
    B, C, H, W = x.shape
    assert out.is_contiguous() and out.shape[0] == B and out.shape[2] == H * r and out.shape[3] == W * r
    check(lib().ksr_bilinear_up(ptr(x), ptr(out), c_int(B), c_int(n_channels), c_int(H), c_int(W), c_int(r),
                                c_int64(C * H * W), c_int(in_channel0), c_int64(out.shape[1] * H * r * W * r),
                                c_int(out_channel0), stream()),
          "ksr_bilinear_up")
