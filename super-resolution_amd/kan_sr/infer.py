"""Tile-with-halo inference: a frame that is split over several launches (memory) or several GPUs (one tile each).

The receptive field of the hyper-encoder is 4 low-resolution pixels in every direction: four stacked depthwise
3x3 convolutions (ffn1.depthwise, ffn2.gate, ffn3.depthwise, ffn4.gate; conv_sep_kan/unet_blocks.py:236-241,267-275);
the transformer (over the batch axis), `feed`, the KAN cell and the pixel shuffle are per pixel
(conv_sep_kan/gazetr.py:110-117, conv_sep_kan/conv_sep_kan_cells.py:136-156).  A tile evaluated together with a 4-pixel
halo of real image data therefore reproduces the whole-frame result on its interior exactly; halo pixels that would
lie outside the frame are simply absent, so the tile sees the frame's own zero padding there.  The chroma planes'
bilinear interpolation (conv_sep_kan/conv_sep_kan.py:30-34) has a support of one pixel.

All images of a batch are tiled alike: the attention mixes the images of the batch at one pixel position, so a tile
must hold every image of the batch (a batch must never be split between tiles)."""
from __future__ import annotations

from typing import Iterator, Optional, Sequence, Tuple

import torch

from ._lib import KsrError

HALO = 4


def tile_grid(H: int, W: int, tile: Tuple[int, int], halo: int = HALO) -> Iterator[Tuple[int, int, int, int, int, int, int, int]]:
    """Yields (y0, y1, x0, x1, ya, yb, xa, xb): interior rows/columns [y0,y1) x [x0,x1) of a tile and the rows/columns
    [ya,yb) x [xa,xb) that are read for it (interior + halo, clipped to the frame)."""
    th, tw = tile
    if th <= 0 or tw <= 0:
        raise KsrError(f"tile size must be positive, got {tile}")
    for y0 in range(0, H, th):
        y1 = min(H, y0 + th)
        for x0 in range(0, W, tw):
            x1 = min(W, x0 + tw)
            yield y0, y1, x0, x1, max(0, y0 - halo), min(H, y1 + halo), max(0, x0 - halo), min(W, x1 + halo)


def predict_tiled(model, x: torch.Tensor, tile: Tuple[int, int] = (270, 480), out: Optional[torch.Tensor] = None,
                  tiles: Optional[Sequence[int]] = None) -> torch.Tensor:
    """model(x) evaluated tile by tile: x [B,3,H,W] -> [B,3,rH,rW].  Each tile's interior is written into `out` (allocated
    here unless given).  `tiles` restricts the call to those tile indices (row-major order of tile_grid): several GPUs
    can share one frame, each evaluating its own tiles into its own copy of `out`.  The model must be in eval mode
    (dropout would draw different masks per tile) and use the GazeTR backbone (the UNet's receptive field is the
    frame)."""
    if model.training:
        raise KsrError("predict_tiled needs model.eval(): train-mode dropout masks depend on the tile")
    if getattr(model, "backbone", "gazetr") != "gazetr" or getattr(model, "attention", False):
        raise KsrError("predict_tiled: only the GazeTR hyper-encoder with the plain feed head has a 4-pixel receptive field")
    if x.dim() != 4 or x.shape[1] != 3:
        raise KsrError(f"expected input [B,3,H,W], got {tuple(x.shape)}")
    B, _, H, W = x.shape
    r = model.upscale
    if out is None:
        out = torch.empty(B, 3, r * H, r * W, device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, 3, r * H, r * W) or out.device != x.device or out.dtype != torch.float32:
        raise KsrError(f"out must be a float32 [B,3,{r * H},{r * W}] tensor on the input's device")
    want = None if tiles is None else set(int(i) for i in tiles)
    with torch.no_grad():
        for i, (y0, y1, x0, x1, ya, yb, xa, xb) in enumerate(tile_grid(H, W, tile)):
            if want is not None and i not in want:
                continue
            yt = model(x[:, :, ya:yb, xa:xb].contiguous())
            out[:, :, r * y0: r * y1, r * x0: r * x1] = yt[:, :, r * (y0 - ya): r * (y1 - ya), r * (x0 - xa): r * (x1 - xa)]
    return out
