"""The constructible blocks of the reference's SeemoRe backbone (conv_sep_kan/seemore.py) on the HIP path, forward only.

The reference's top class `SeemoRe` (seemore.py:13-77) raises in its constructor (`RME.__init__` passes `mlp_ratio=` to a
`GatedFFN` whose constructor has no such argument, :176 vs :355-358; and the default `lr_space: int = 1` reaches `MoEBlock`'s
string switch, :219-226), so there is no whole-model behaviour to reproduce: the drop-in surface is the blocks that can be
built.  Each class below has the reference's constructor arguments, parameter names (state_dicts interchange) and NCHW
forward signature; the computation is a sequence of libksr kernels over NHWC fp32 tensors (ksr_sm_*), PyTorch owns the
buffers and does the NCHW <-> NHWC relayout at the module boundary (`forward_nhwc` skips it).  GPU float32 only; there
is no fallback.
"""
from __future__ import annotations

import ctypes
from ctypes import c_float, c_int, c_int64, c_void_p
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from ._lib import KsrError, check, lib, ptr, stream

C = 64          # channel count the kernels are written for (the reference's default embedding_dim)


def _need(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda or t.dtype is not torch.float32:
        raise KsrError(f"{what}: the HIP path needs float32 GPU tensors (there is no other path)")


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1).contiguous()


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2).contiguous()


def _dense_w(conv: nn.Conv2d, out_perm: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """torch [co][ci][kh][kw] -> kernel layout [kh*kw][ci][co] (+ bias), optionally with the output channels permuted."""
    w, b = conv.weight, conv.bias
    if out_perm is not None:
        w, b = w[out_perm], (None if b is None else b[out_perm])
    co, ci, kh, kw = w.shape
    return w.permute(2, 3, 1, 0).reshape(kh * kw, ci, co).contiguous(), (None if b is None else b.contiguous())


def _dw_w(conv: nn.Conv2d) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """torch depthwise [c][1][kh][kw] -> kernel layout [kh*kw][c]."""
    c, _, kh, kw = conv.weight.shape
    return conv.weight.reshape(c, kh * kw).t().contiguous(), (None if conv.bias is None else conv.bias.contiguous())


def conv_nhwc(x: torch.Tensor, cin: int, x_c0: int, w: torch.Tensor, b: Optional[torch.Tensor], cout: int, kh: int, kw: int,
              stride: int = 1, pad: Tuple[int, int] = (0, 0), depthwise: bool = False, gelu: bool = False,
              mul: Optional[Tuple[torch.Tensor, int]] = None, res: Optional[Tuple[torch.Tensor, int]] = None) -> torch.Tensor:
    """One ksr_sm_conv_fwd launch.  x [B,H,W,ld] read from channel x_c0; mul / res = (tensor, first channel)."""
    _need(x, "seemore conv")
    B, H, W, ld = x.shape
    Ho, Wo = (H + 2 * pad[0] - kh) // stride + 1, (W + 2 * pad[1] - kw) // stride + 1
    out = torch.empty(B, Ho, Wo, cout, device=x.device, dtype=torch.float32)
    mt, mc = mul if mul is not None else (None, 0)
    rt, rc = res if res is not None else (None, 0)
    check(lib().ksr_sm_conv_fwd(ptr(x), c_int(ld), c_int(x_c0), ptr(mt), c_int(0 if mt is None else mt.shape[-1]), c_int(mc),
                                ptr(w), ptr(b), ptr(rt), c_int(0 if rt is None else rt.shape[-1]), c_int(rc), ptr(out),
                                c_int(cout), c_int(0), c_int(B), c_int(H), c_int(W), c_int(cin), c_int(cout), c_int(kh), c_int(kw),
                                c_int(stride), c_int(pad[0]), c_int(pad[1]), c_int(int(depthwise)), c_int(int(gelu)), stream()),
          "ksr_sm_conv_fwd")
    return out


class _Block(nn.Module):
    """NCHW in / NCHW out like the reference module; forward_nhwc is the computation."""

    def forward(self, x: torch.Tensor, *more: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return to_nchw(self.forward_nhwc(to_nhwc(x), *(to_nhwc(t) for t in more)))


class LayerNorm(_Block):                                   # seemore.py:386-410 (channels_first form only)
    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_first"):
        super().__init__()
        if data_format != "channels_first":
            raise KsrError("the HIP path has the channels_first LayerNorm of seemore.py (the only form its blocks use)")
        if normalized_shape != C:
            raise KsrError(f"LayerNorm kernels are written for {C} channels")
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps

    def forward_nhwc(self, x):
        _need(x, "seemore LayerNorm")
        out = torch.empty_like(x)
        check(lib().ksr_sm_layernorm_fwd(ptr(x), ptr(self.weight), ptr(self.bias), ptr(out), c_int64(x.numel() // C),
                                         c_float(self.eps), stream()), "ksr_sm_layernorm_fwd")
        return out


class StripedConv2d(_Block):                               # seemore.py:325-341
    def __init__(self, in_ch: int, kernel_size: int, depthwise: bool = False):
        super().__init__()
        if not depthwise:
            raise KsrError("StripedConv2d: the blocks of seemore.py only use the depthwise form")
        self.in_ch, self.kernel_size, self.padding = in_ch, kernel_size, kernel_size // 2
        self.conv = nn.Sequential(
            nn.Conv2d(in_ch, in_ch, kernel_size=(1, kernel_size), padding=(0, self.padding), groups=in_ch),
            nn.Conv2d(in_ch, in_ch, kernel_size=(kernel_size, 1), padding=(self.padding, 0), groups=in_ch))

    def forward_nhwc(self, x, x_c0: int = 0, gelu: bool = False):
        ks, pd = self.kernel_size, self.padding
        w0, b0 = _dw_w(self.conv[0])
        w1, b1 = _dw_w(self.conv[1])
        h = conv_nhwc(x, self.in_ch, x_c0, w0, b0, self.in_ch, 1, ks, pad=(0, pd), depthwise=True)
        return conv_nhwc(h, self.in_ch, 0, w1, b1, self.in_ch, ks, 1, pad=(pd, 0), depthwise=True, gelu=gelu)


class GatedFFN(_Block):                                    # seemore.py:354-382
    def __init__(self, in_ch, kernel_size, act_layer):
        super().__init__()
        if not isinstance(act_layer, nn.GELU) or getattr(act_layer, "approximate", "none") != "none":
            raise KsrError("GatedFFN: the kernels implement the exact GELU the reference's blocks pass (nn.GELU())")
        mlp_ch = in_ch * 2
        self.fn_1 = nn.Sequential(nn.Conv2d(in_ch, mlp_ch, kernel_size=1, padding=0), act_layer)
        self.fn_2 = nn.Sequential(nn.Conv2d(in_ch, in_ch, kernel_size=1, padding=0), act_layer)
        self.gate = nn.Conv2d(mlp_ch // 2, mlp_ch // 2, kernel_size=kernel_size, padding=kernel_size // 2, groups=mlp_ch // 2)
        self.in_ch, self.kernel_size = in_ch, kernel_size

    def forward_nhwc(self, x, res: Optional[torch.Tensor] = None):
        c, ks = self.in_ch, self.kernel_size
        w1, b1 = _dense_w(self.fn_1[0])
        t = conv_nhwc(x, c, 0, w1, b1, 2 * c, 1, 1, gelu=True)                     # [.., 2C]: value | gate
        wg, bg = _dw_w(self.gate)
        g = conv_nhwc(t, c, c, wg, bg, c, ks, ks, pad=(ks // 2, ks // 2), depthwise=True)
        w2, b2 = _dense_w(self.fn_2[0])
        return conv_nhwc(t, c, 0, w2, b2, c, 1, 1, gelu=True, mul=(g, 0), res=None if res is None else (res, 0))


class StripedConvFormer(_Block):                           # seemore.py:134-155
    def __init__(self, in_ch: int, kernel_size: int):
        super().__init__()
        self.in_ch, self.kernel_size, self.padding = in_ch, kernel_size, kernel_size // 2
        self.proj = nn.Conv2d(in_ch, in_ch, kernel_size=1, padding=0)
        self.to_qv = nn.Sequential(nn.Conv2d(in_ch, in_ch * 2, kernel_size=1, padding=0), nn.GELU())
        self.attn = StripedConv2d(in_ch, kernel_size=kernel_size, depthwise=True)

    def forward_nhwc(self, x, res: Optional[torch.Tensor] = None):
        c = self.in_ch
        wq, bq = _dense_w(self.to_qv[0])
        t = conv_nhwc(x, c, 0, wq, bq, 2 * c, 1, 1, gelu=True)                     # q | v
        q = self.attn.forward_nhwc(t, 0)
        wp, bp = _dense_w(self.proj)
        return conv_nhwc(q, c, 0, wp, bp, c, 1, 1, mul=(t, c), res=None if res is None else (res, 0))


class SME(_Block):                                         # seemore.py:114-129
    def __init__(self, in_ch: int, kernel_size: int = 11):
        super().__init__()
        self.norm_1 = LayerNorm(in_ch, data_format="channels_first")
        self.block = StripedConvFormer(in_ch=in_ch, kernel_size=kernel_size)
        self.norm_2 = LayerNorm(in_ch, data_format="channels_first")
        self.ffn = GatedFFN(in_ch, kernel_size=3, act_layer=nn.GELU())

    def forward_nhwc(self, x):
        x = self.block.forward_nhwc(self.norm_1.forward_nhwc(x), res=x)           # the residual rides on the last conv
        return self.ffn.forward_nhwc(self.norm_2.forward_nhwc(x), res=x)


class Expert(_Block):                                      # seemore.py:289-302
    def __init__(self, in_ch: int, low_dim: int):
        super().__init__()
        self.conv_1 = nn.Conv2d(in_ch, low_dim, kernel_size=1, padding=0)
        self.conv_2 = nn.Conv2d(in_ch, low_dim, kernel_size=1, padding=0)
        self.conv_3 = nn.Conv2d(low_dim, in_ch, kernel_size=1, padding=0)
        self.low_dim = low_dim

    def params6(self) -> List[torch.Tensor]:
        return [self.conv_1.weight, self.conv_1.bias, self.conv_2.weight, self.conv_2.bias, self.conv_3.weight, self.conv_3.bias]

    def forward_nhwc(self, x, k):
        """conv_3(conv_2(k) * conv_1(x)): the mixture kernel with this one expert at weight 1, minus its `x +`."""
        B, H, W, _ = x.shape
        ones = torch.ones(B, 1, device=x.device, dtype=torch.float32)
        return _experts(x, 0, k, 0, ones, [self]) - x


class Router(nn.Module):                                   # seemore.py:305-318
    def __init__(self, in_ch: int, num_experts: int):
        super().__init__()
        # same parameter name as the reference's nn.Sequential(AdaptiveAvgPool2d, Rearrange, Linear): body.2.weight
        self.body = nn.Sequential(nn.Identity(), nn.Identity(), nn.Linear(in_ch, num_experts, bias=False))
        self.num_experts = num_experts

    def route(self, x_nhwc, x_c0: int, topk: int):
        """(weights [B,E] with all but the top-k probabilities zeroed, logits [B,E])."""
        _need(x_nhwc, "seemore Router")
        B, H, W, ld = x_nhwc.shape
        E = self.num_experts
        wts = torch.empty(B, E, device=x_nhwc.device, dtype=torch.float32)
        logits = torch.empty(B, E, device=x_nhwc.device, dtype=torch.float32)
        check(lib().ksr_sm_route_fwd(ptr(x_nhwc), c_int(ld), c_int(x_c0), ptr(self.body[2].weight), c_int(E), c_int(topk), c_int(B),
                                     c_int64(H * W), ptr(wts), ptr(logits), stream()), "ksr_sm_route_fwd")
        return wts, logits

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return self.route(to_nhwc(x), 0, 1)[1]


def _experts(x, x_c0, k, k_c0, weights, experts: List[Expert]) -> torch.Tensor:
    _need(x, "seemore experts")
    B, H, W, xld = x.shape
    n = len(experts)
    arr = (c_void_p * (6 * n))()
    for e, ex in enumerate(experts):
        for q, t in enumerate(ex.params6()):
            if not t.is_contiguous():
                raise KsrError("expert parameters must be contiguous")
            arr[6 * e + q] = ptr(t).value
    low = (c_int * n)(*[ex.low_dim for ex in experts])
    out = torch.empty(B, H, W, C, device=x.device, dtype=torch.float32)
    check(lib().ksr_sm_experts_fwd(ptr(x), c_int(xld), c_int(x_c0), ptr(k), c_int(k.shape[-1]), c_int(k_c0), ptr(weights), arr, low,
                                   c_int(n), ptr(out), c_int(B), c_int64(H * W), stream()), "ksr_sm_experts_fwd")
    return out


class MoELayer(_Block):                                    # seemore.py:261-285
    def __init__(self, experts: List[nn.Module], gate: nn.Module, num_expert: int = 1):
        super().__init__()
        assert len(experts) > 0
        self.experts = nn.ModuleList(experts)
        self.gate = gate
        self.num_expert = num_expert

    def forward_nhwc(self, inputs, k, x_c0: int = 0, k_c0: int = 0):
        """inputs + sum over each image's top-k experts of weight * expert(inputs, k).  The reference's train branch runs
        every expert with the others' weights zeroed and its eval branch (one image) only the selected ones: one sum."""
        weights, _ = self.gate.route(inputs, x_c0, self.num_expert)
        return _experts(inputs, x_c0, k, k_c0, weights, list(self.experts))


class MoEBlock(_Block):                                    # seemore.py:188-258
    def __init__(self, in_ch: int, num_experts: int, topk: int, use_shuffle: bool = False, lr_space: str = "linear",
                 recursive: int = 2):
        super().__init__()
        if in_ch != C:
            raise KsrError(f"MoEBlock kernels are written for {C} channels")
        self.use_shuffle, self.recursive, self.in_ch = use_shuffle, recursive, in_ch
        self.conv_1 = nn.Sequential(nn.Conv2d(in_ch, in_ch, kernel_size=3, padding=1), nn.GELU(),
                                    nn.Conv2d(in_ch, 2 * in_ch, kernel_size=1, padding=0))
        self.agg_conv = nn.Sequential(nn.Conv2d(in_ch, in_ch, kernel_size=4, stride=4, groups=in_ch), nn.GELU())
        self.conv = nn.Sequential(nn.Conv2d(in_ch, in_ch, kernel_size=3, stride=1, padding=1, groups=in_ch),
                                  nn.Conv2d(in_ch, in_ch, kernel_size=1, padding=0))
        self.conv_2 = nn.Sequential(StripedConv2d(in_ch, kernel_size=3, depthwise=True), nn.GELU())
        if lr_space == "linear":
            grow_func = lambda i: i + 2                     # noqa: E731
        elif lr_space == "exp":
            grow_func = lambda i: 2 ** (i + 1)              # noqa: E731
        elif lr_space == "double":
            grow_func = lambda i: 2 * i + 2                 # noqa: E731
        else:
            raise NotImplementedError(f"lr_space {lr_space} not implemented")
        self.moe_layer = MoELayer(experts=[Expert(in_ch=in_ch, low_dim=grow_func(i)) for i in range(num_experts)],
                                  gate=Router(in_ch=in_ch, num_experts=num_experts), num_expert=topk)
        self.proj = nn.Conv2d(in_ch, in_ch, kernel_size=1, padding=0)

    def forward_nhwc(self, x):
        c = self.in_ch
        B, H, W, _ = x.shape
        w, b = _dense_w(self.conv_1[0])
        t = conv_nhwc(x, c, 0, w, b, c, 3, 3, pad=(1, 1), gelu=True)
        perm = None
        if self.use_shuffle:
            # channel_shuffle(groups=2) of the 2C outputs (seemore.py:345-351) = a permutation of conv_1[2]'s output channels
            j = torch.arange(2 * c, device=x.device)
            perm = (j % 2) * c + j // 2
        w, b = _dense_w(self.conv_1[2], perm)
        t = conv_nhwc(t, c, 0, w, b, 2 * c, 1, 1)                                   # x half | k half, read in place below
        xs = self.conv_2[0].forward_nhwc(t, 0, gelu=True)
        wa, ba = _dw_w(self.agg_conv[0])
        a, a_c0 = t, c
        for _ in range(self.recursive):
            a = conv_nhwc(a, c, a_c0, wa, ba, c, 4, 4, stride=4, depthwise=True, gelu=True)
            a_c0 = 0
        w, b = _dw_w(self.conv[0])
        a = conv_nhwc(a, c, a_c0, w, b, c, 3, 3, pad=(1, 1), depthwise=True)
        w, b = _dense_w(self.conv[1])
        a = conv_nhwc(a, c, 0, w, b, c, 1, 1)
        k = torch.empty(B, H, W, c, device=x.device, dtype=torch.float32)
        check(lib().ksr_sm_resize_add_fwd(ptr(a), c_int(a.shape[1]), c_int(a.shape[2]), ptr(t), c_int(2 * c), c_int(c), ptr(k),
                                          c_int(B), c_int(H), c_int(W), stream()), "ksr_sm_resize_add_fwd")
        y = self.moe_layer.forward_nhwc(xs, k)
        w, b = _dense_w(self.proj)
        return conv_nhwc(y, c, 0, w, b, c, 1, 1)
