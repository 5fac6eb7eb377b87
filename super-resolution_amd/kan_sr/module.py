"""Drop-in `ConvSepKan` nn.Module (constructor / forward signature and state_dict keys of
conv_sep_kan/conv_sep_kan.py:8-42) whose forward and backward run on libksr's HIP kernels.

The sub-modules below are PARAMETER CONTAINERS: they exist so that `state_dict()` has the reference's
52 keys, so that `.modules()` yields nn.Conv2d / nn.Linear instances for the training pipeline's
isinstance-based re-initialisation (src/ml/pipelines/default.py:41-55), and so that default
initialisation consumes the RNG in the reference's order.  They have no forward of their own.
"""
from __future__ import annotations

import copy
from typing import Dict, List

import torch
import torch.nn as nn

from . import engine
from ._lib import KsrError

MID = 64


class _Container(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise KsrError("parameter container: the computation runs in ConvSepKan.forward (HIP kernels)")


class FFN(_Container):                      # conv_sep_kan/unet_blocks.py:212-234
    def __init__(self, in_features, out_features):
        super().__init__()
        hidden = in_features
        self.pointwise1 = nn.Conv2d(in_features, hidden, kernel_size=1)
        self.depthwise = nn.Conv2d(hidden, hidden, kernel_size=3, padding=1, groups=hidden)
        self.pointwise2 = nn.Conv2d(hidden, out_features, kernel_size=1)


class GatedFFN(_Container):                 # conv_sep_kan/unet_blocks.py:244-265
    def __init__(self, channels):
        super().__init__()
        self.fn_1 = nn.Sequential(nn.Conv2d(channels, 2 * channels, kernel_size=1), nn.ReLU())
        self.fn_2 = nn.Conv2d(channels, channels, kernel_size=1)
        self.gate = nn.Conv2d(channels, channels, kernel_size=3, padding=1, groups=channels)


class TransformerEncoderLayer(_Container):  # conv_sep_kan/gazetr.py:35-51
    def __init__(self, channels, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(channels, nhead, dropout=dropout, batch_first=True)
        self.linear1 = nn.Linear(channels, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, channels)
        self.norm1 = nn.LayerNorm(channels)
        self.norm2 = nn.LayerNorm(channels)
        self.p_drop = dropout


class TransformerEncoder(_Container):       # conv_sep_kan/gazetr.py:16-22
    def __init__(self, layer, num_layers, norm):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])
        self.norm = norm


class GazeTR(_Container):                   # conv_sep_kan/gazetr.py:69-98
    def __init__(self, in_channels, out_channels, attention=False):
        super().__init__()
        self.ffn1 = FFN(in_channels, MID)
        self.ffn2 = GatedFFN(MID)
        self.ffn3 = FFN(MID, MID)
        self.ffn4 = GatedFFN(MID)
        layer = TransformerEncoderLayer(MID, 2, 32, 0.1)
        self.encoder = TransformerEncoder(layer, 2, nn.LayerNorm(MID))
        if attention:
            from .chan_attention import ConvSepKanAttention
            self.feed = ConvSepKanAttention(MID, MID, out_channels)
            # optional name -> tensor map (names relative to this module): the trunk's backward on the autograd tape then
            # writes each parameter gradient straight into its tensor and reports none to autograd (TrainStep: views of
            # its flat gradient buffer, so no accumulation launch per parameter)
            self.grad_sink = None
        else:
            self.feed = nn.Conv2d(MID, out_channels, kernel_size=1)


class ConvSepKanEncoder(_Container):        # conv_sep_kan/conv_sep_kan_encoder.py:12-17
    def __init__(self, in_channels, out_channels, attention=False, backbone="gazetr"):
        super().__init__()
        if backbone == "unet":              # the backbone conv_sep_kan_encoder.py:7 imports beside GazeTR
            from .unet import UNet
            self.encoder = UNet(in_channels, out_channels)
        else:
            self.encoder = GazeTR(in_channels, out_channels, attention)


class ConvSepKanCell(_Container):           # conv_sep_kan/conv_sep_kan_cells.py:106-134 (no parameters)
    def __init__(self, in_channels, out_channels, grid_size, spline_order, residual_std, grid_range):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.grid_size, self.spline_order, self.grid_range = grid_size, spline_order, list(grid_range)
        self.size = out_channels * in_channels * (grid_size + spline_order + 2)


class ConvSepKanLayer(_Container):          # conv_sep_kan/conv_sep_kan_layer.py:7-15
    def __init__(self, in_channels, out_channels, grid_size, spline_order, residual_std, grid_range, attention=False,
                 backbone="gazetr"):
        super().__init__()
        self.kan_cell = ConvSepKanCell(in_channels, out_channels, grid_size, spline_order, residual_std, grid_range)
        self.encoder = ConvSepKanEncoder(in_channels, self.kan_cell.size, attention, backbone)


# cfg-4 wiring: evaluate the attention head's per-image 1x1 map inside the fused feed+KAN kernel (False: materialise the
# [N, P] parameter tensor and run the stand-alone KAN cell, the first version of this path)
FUSED_ATTENTION_HEAD = True


class ConvSepKan(nn.Module):
    """Same constructor arguments as the reference; `kernel_sizes` and `residual_std` are accepted and
    unused there too.  forward(x[B,3,H,W]) -> [B,3,rH,rW] (Y', Cb, Cr).

    attention=True (keyword only, not in the reference's constructor) is this build's wiring of the
    reference's stand-alone ConvSepKanAttention (conv_sep_kan/conv_sep_kan_attention.py) into the model: the
    `feed` 1x1 conv that maps the 64 encoder features to the per-pixel KAN parameters is replaced by
    ConvSepKanAttention(64, 64, P) (state_dict keys `...encoder.encoder.feed.{q,k,v,a}_{proj,dw}.*`).

    backbone="unet" (keyword only) selects the reference's UNet (conv_sep_kan/unet.py) as the hyper-encoder in place
    of GazeTR, i.e. `self.encoder = UNet(in_channels, out_channels)` at conv_sep_kan_encoder.py:17; state_dict keys
    `layers.N.encoder.encoder.{proj,down1..4,up1..4,att1..4,reproj}.*`.  Forward and backward (fp32 only)."""

    def __init__(self, in_channels: list, out_channels: list, kernel_sizes: list, grid_size, spline_order,
                 residual_std, grid_range, upscale, *, attention: bool = False, backbone: str = "gazetr"):
        super().__init__()
        if backbone not in ("gazetr", "unet"):
            raise KsrError(f"backbone must be 'gazetr' or 'unet', got {backbone!r}")
        if backbone == "unet" and attention:
            raise KsrError("attention=True replaces GazeTR's feed conv: it does not combine with backbone='unet'")
        self.upscale = upscale
        self.attention = attention
        self.backbone = backbone
        self.grid_size, self.spline_order, self.grid_range = grid_size, spline_order, list(grid_range)
        # "fp32": exact fp32 MFMA (parity with the reference to fp32 round-off); "bf16": bf16 MFMA operands with
        # fp32 accumulation (separately stated tolerance).  Not part of the reference's constructor.
        self.precision = "fp32"
        self.layers = nn.ModuleList()
        self.shuffle = nn.PixelShuffle(upscale)          # kept for module-tree parity; no parameters
        for in_ch, out_ch in zip(in_channels, out_channels):
            self.layers.append(ConvSepKanLayer(in_ch, out_ch, grid_size, spline_order, residual_std, grid_range,
                                               attention, backbone))
        if len(self.layers) == 0:
            raise KsrError("ConvSepKan needs at least one layer")
        # Every layer is fed the Y plane and only the last layer's output survives the reference's
        # loop (conv_sep_kan.py:36-37): the earlier layers are dead weight and are not evaluated.
        last = self.layers[-1].kan_cell
        if last.in_channels != 1:
            raise KsrError("the layers read x[:, :1]: in_channels must be 1")
        if last.out_channels != upscale * upscale:
            raise KsrError("concat([y, cb, cr]) needs out_channels[-1] == upscale**2")

    def _params(self) -> Dict[str, torch.Tensor]:
        li = len(self.layers) - 1
        pre = f"layers.{li}."
        return {pre + n: t for n, t in self.layers[li].named_parameters()}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        if self.backbone == "unet":
            from .ops import kan_head, kan_lin_head
            engine.set_precision(self.precision)
            net = self.layers[-1].encoder.encoder
            if net.out_channels <= 512 and not (torch.is_grad_enabled() and x.requires_grad):
                # the [N, P] parameter tensor is never materialised: the KAN-cell kernels apply the UNet's last 1x1 conv
                # (they return no gradient for the image: a pass that asks for one takes the materialised form below)
                y4 = net.forward_nhwc(x[:, :1], fuse_head=True)
                return kan_lin_head(x, y4, net, self.grid_size, self.spline_order, self.grid_range, self.upscale)
            w = net.forward_nhwc(x[:, :1])
            return kan_head(x, w, self.grid_size, self.spline_order, self.grid_range, self.upscale)
        g = engine.Geometry(x, self.upscale, self.grid_size, self.spline_order, self.grid_range)
        p = self._params()
        pre = f"layers.{len(self.layers) - 1}.encoder.encoder."
        engine.set_precision(self.precision)
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(t.requires_grad for t in p.values()))
        p_drop = self.layers[-1].encoder.encoder.encoder.layers[0].p_drop if self.training else 0.0
        seed = engine.new_seed() if p_drop > 0 else 0
        if self.attention:
            return self._forward_attention(x, p, pre, g, p_drop, seed, needs_grad)
        if not needs_grad:
            with torch.no_grad():
                return engine.forward_infer(x, p, pre, g, p_drop, seed)
        from .autograd import conv_sep_kan_train
        return conv_sep_kan_train(x, p, pre, g, p_drop, seed, self.precision)

    def _forward_attention(self, x, p, pre, g, p_drop, seed, needs_grad):
        """Encoder trunk -> channel attention head -> KAN + pixel shuffle + chroma."""
        from .ops import kan_head
        head = self.layers[-1].encoder.encoder.feed
        # bf16 storage: the encoder trunk runs in that mode, and so does the head when its last step runs inside the fused
        # feed+KAN kernel; the head with a materialised parameter tensor keeps fp32 tensors (bf16 operands), so there the
        # 64-channel features are widened once at the boundary (and their gradient narrowed by autograd's cast)
        head_bf16s = self.precision == "bf16s" and FUSED_ATTENTION_HEAD
        head.precision = self.precision if (self.precision != "bf16s" or head_bf16s) else "bf16"
        with torch.set_grad_enabled(needs_grad):
            if needs_grad:
                from .autograd import trunk_train
                sink = self.layers[-1].encoder.encoder.grad_sink
                feat = trunk_train(x, p, pre, g, p_drop, seed, self.precision,
                                   None if sink is None else {pre + n: t for n, t in sink.items()})
            else:
                feat = engine.encoder_forward(x, p, pre, g, None, p_drop, seed)
            if feat.dtype is not torch.float32 and not head_bf16s:
                feat = feat.float()
            if FUSED_ATTENTION_HEAD:
                # the head's last 1x1 map (one matrix per image) runs inside the fused feed+KAN kernel: no [N, P] tensor
                return head.forward_kan(feat, (g.B, g.H, g.W), x, g.n_coef, g.knots, g.k, g.r)
            w = head.forward_nhwc(feat, (g.B, g.H, g.W))
            return kan_head(x, w, self.grid_size, self.spline_order, self.grid_range, self.upscale)
