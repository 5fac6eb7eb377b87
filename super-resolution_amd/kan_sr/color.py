"""RGB <-> YCbCr on the GPU (reference: src/ml/utils/colors/ycbcr.py): same function names, argument checks
and constants as the reference's helpers, computed by libksr kernels, plus the fused RGB-in / RGB-out inference
entry `predict_rgb`."""
from __future__ import annotations

from ctypes import c_int, c_int64

import torch

from . import engine
from ._lib import KsrError, check, host_floats, lib, ptr, stream

# ycbcr.py:6-10
O = torch.tensor([16., 128., 128.]) / 255.
T = torch.tensor([[0.256788235294118, 0.504129411764706, 0.097905882352941],
                  [-0.148223529411765, -0.290992156862745, 0.439215686274510],
                  [0.439215686274510, -0.367788235294118, -0.071427450980392]]).T


def _mats():
    return (host_floats(T.contiguous().flatten().tolist()), host_floats(T.inverse().contiguous().flatten().tolist()),
            host_floats(O.tolist()))


def _check(image):
    if not isinstance(image, torch.Tensor):
        raise TypeError(f"Input type is not a Tensor. Got {type(image)}")
    if len(image.shape) < 3 or image.shape[-3] != 3:
        raise ValueError(f"Input size must have a shape of (*, 3, H, W). Got {image.shape}")
    if not image.is_cuda:
        raise KsrError("colour conversion runs on the GPU (the HIP path is the only path)")


def rgb_to_ycbcr(image: torch.Tensor) -> torch.Tensor:
    """RGB in [0,1], shape (*, 3, H, W) -> YCbCr (SDTV range), ycbcr.py:13-41."""
    _check(image)
    x = image.to(torch.float32).contiguous()
    H, W = x.shape[-2:]
    B = x.numel() // (3 * H * W)
    out = torch.empty_like(x)
    m, _, o = _mats()
    check(lib().ksr_rgb_to_ycbcr(ptr(x), ptr(out), c_int(B), c_int(H), c_int(W), m, o, stream()), "ksr_rgb_to_ycbcr")
    return out


def ycbcr_to_rgb(image: torch.Tensor) -> torch.Tensor:
    """YCbCr, shape (*, 3, H, W) -> RGB, ycbcr.py:44-73 (the result stays on the GPU)."""
    _check(image)
    x = image.to(torch.float32).contiguous()
    H, W = x.shape[-2:]
    B = x.numel() // (3 * H * W)
    out = torch.empty_like(x)
    _, mi, o = _mats()
    check(lib().ksr_ycbcr_to_rgb(ptr(x), ptr(out), c_int(B), c_int(H), c_int(W), mi, o, stream()), "ksr_ycbcr_to_rgb")
    return out


class RgbToYcbcr(torch.nn.Module):
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return rgb_to_ycbcr(image)


class YcbcrToRgb(torch.nn.Module):
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return ycbcr_to_rgb(image)


@torch.no_grad()
def predict_rgb(model, rgb: torch.Tensor) -> torch.Tensor:
    """RGB frame(s) [B,3,H,W] in [0,1] -> super-resolved RGB [B,3,rH,rW]: rgb_to_ycbcr, the model's luma path, and
    one fused kernel for bilinear chroma upsampling + ycbcr_to_rgb (the chroma planes of the YCbCr output are
    never written).  Same result as ycbcr_to_rgb(model(rgb_to_ycbcr(rgb))) in eval mode."""
    if getattr(model, "attention", False):
        raise KsrError("predict_rgb is implemented for the plain feed head")
    x = rgb_to_ycbcr(rgb)
    g = engine.Geometry(x, model.upscale, model.grid_size, model.spline_order, model.grid_range)
    p = model._params()
    pre = f"layers.{len(model.layers) - 1}.encoder.encoder."
    engine.set_precision("fp32" if model.precision == "bf16s" else model.precision)   # inference kernels: fp32 storage
    feat = engine.encoder_forward(x, p, pre, g, None, 0.0, 0)
    plane = g.H * g.r * g.W * g.r
    y_hr = torch.empty(g.B, 1, g.H * g.r, g.W * g.r, device=x.device, dtype=torch.float32)
    engine.feed_kan_forward(feat, x, p, pre, g, y_hr)
    out = torch.empty(g.B, 3, g.H * g.r, g.W * g.r, device=x.device, dtype=torch.float32)
    _, mi, o = _mats()
    check(lib().ksr_bilinear_up_to_rgb(ptr(x), ptr(y_hr), c_int64(plane), ptr(out), c_int(g.B), c_int(g.H), c_int(g.W),
                                       c_int(g.r), mi, o, stream()), "ksr_bilinear_up_to_rgb")
    return out
