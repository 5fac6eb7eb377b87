"""Colour pre/post-processing kernels against the reference goldens and the oracle."""
import pytest
import torch

from conftest import load_golden, t
from oracle import convsepkan_oracle as orc

pytestmark = pytest.mark.gpu


def test_golden():
    from kan_sr import color
    g = load_golden("color")
    assert (color.ycbcr_to_rgb(t(g["ycbcr"]).cuda()).cpu() - t(g["rgb_back"])).abs().max() <= 1e-6
    assert (color.ycbcr_to_rgb(t(g["ycc2"]).cuda()).cpu() - t(g["rgb2"])).abs().max() <= 1e-6      # (3, H, W) input
    assert (color.rgb_to_ycbcr(t(g["rgb"]).cuda()).cpu() - t(g["ycbcr"])).abs().max() <= 1e-6
    with pytest.raises(ValueError):
        color.rgb_to_ycbcr(torch.zeros(2, 4, 5, 5, device="cuda"))
    with pytest.raises(TypeError):
        color.ycbcr_to_rgb([1, 2, 3])


@pytest.mark.parametrize("B,H,W,r", [(1, 37, 45, 4), (2, 16, 24, 2)])
def test_predict_rgb_equals_unfused_pipeline(B, H, W, r):
    """RGB in -> RGB out with the fused chroma-upsampling + colour stage against ycbcr_to_rgb(model(rgb_to_ycbcr(x)))
    built from the oracle."""
    from kan_sr import color
    from kan_sr.module import ConvSepKan
    torch.manual_seed(B + H)
    m = ConvSepKan([1], [r * r], [3], 5, 3, 0.1, [-1, 1], r).eval()
    params = {n: v.detach().clone() for n, v in m.state_dict().items()}
    rgb = torch.rand(B, 3, H, W)
    ref = orc.ycbcr_to_rgb(orc.conv_sep_kan_forward(orc.rgb_to_ycbcr(rgb), params, out_channels=[r * r], grid_size=5,
                                                    spline_order=3, grid_range=[-1, 1], upscale=r))
    out = color.predict_rgb(m.cuda(), rgb.cuda())
    err = (out.cpu() - ref).abs()
    assert bool((err <= 2e-5 + 1e-4 * ref.abs()).all()), float(err.max())
