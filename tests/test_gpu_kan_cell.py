"""GPU parity: stand-alone KAN cell + pixel shuffle + chroma bilinear vs the CPU oracle and goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2, t
from oracle import convsepkan_oracle as orc

pytestmark = pytest.mark.gpu
FWD_ATOL, FWD_RTOL, GRAD_REL = 1e-5, 1e-4, 1e-4


def assert_fwd(a, ref):
    a, ref = a.cpu(), ref.cpu()
    assert a.shape == ref.shape
    err = (a - ref).abs()
    assert bool((err <= FWD_ATOL + FWD_RTOL * ref.abs()).all()), f"max err {float(err.max()):.3e}"


@pytest.mark.parametrize("r,G,k,I", [(2, 5, 3, 1), (4, 5, 3, 1), (4, 16, 3, 1), (2, 5, 3, 3)])
@pytest.mark.parametrize("nhwc", [False, True])
def test_kan_cell_golden(r, G, k, I, nhwc):
    from kan_sr import ops
    g = load_golden("kan_cell")
    tag = f"r{r}_G{G}_k{k}_I{I}_"
    O = r * r
    x = t(g[tag + "x"], device="cuda").requires_grad_()
    w = t(g[tag + "w"], device="cuda")
    if nhwc:
        w = w.permute(0, 2, 3, 1).reshape(-1, w.shape[1]).contiguous()
    w.requires_grad_()
    y = ops.kan_cell(x, w, O, G, k, [-1, 1], shuffle_r=r, w_nhwc=nhwc)
    assert_fwd(y.detach(), t(g[tag + "y"]))
    y.backward(t(g[tag + "gy"], device="cuda"))
    gw = w.grad
    if nhwc:
        B, _, H, W = x.shape
        gw = gw.reshape(B, H, W, -1).permute(0, 3, 1, 2)
    assert rel_l2(gw.cpu(), t(g[tag + "gw"])) <= GRAD_REL
    assert rel_l2(x.grad.cpu(), t(g[tag + "gx"])) <= GRAD_REL


@pytest.mark.parametrize("O,I,G,k", [(4, 1, 5, 3), (16, 1, 5, 3), (7, 5, 5, 3), (16, 1, 16, 3)])
def test_kan_layer_golden(O, I, G, k):
    """The reference KANLayer vectors (N samples) viewed as an N x 1 image."""
    from kan_sr import ops
    g = load_golden("kan_layer")
    tag = f"O{O}_I{I}_G{G}_k{k}_"
    N = g[tag + "x"].shape[0]
    x = t(g[tag + "x"], device="cuda").t().reshape(1, I, N, 1).contiguous().requires_grad_()
    w = torch.cat([t(g[tag + "coef"]).reshape(N, -1), t(g[tag + "uw"]).reshape(N, -1),
                   t(g[tag + "rw"]).reshape(N, -1)], dim=1).cuda().requires_grad_()
    y = ops.kan_cell(x, w, O, G, k, [-1, 1], shuffle_r=0, w_nhwc=True)      # [1,O,N,1]
    assert_fwd(y.detach()[0, :, :, 0].t(), t(g[tag + "y"])[0])
    y.backward(t(g[tag + "gy"], device="cuda")[0].t().reshape(1, O, N, 1).contiguous())
    K = G + k
    gw = w.grad.cpu()
    assert rel_l2(gw[:, :O * I * K], t(g[tag + "gcoef"]).reshape(N, -1)) <= GRAD_REL
    assert rel_l2(gw[:, O * I * K:O * I * (K + 1)], t(g[tag + "guw"]).reshape(N, -1)) <= GRAD_REL
    assert rel_l2(gw[:, O * I * (K + 1):], t(g[tag + "grw"]).reshape(N, -1)) <= GRAD_REL
    assert rel_l2(x.grad.cpu()[0, :, :, 0].t(), t(g[tag + "gx"])) <= GRAD_REL


@pytest.mark.parametrize("G,k", [(5, 3), (16, 3), (11, 3), (5, 1), (8, 2)])
def test_basis_bit_exact_through_cell(G, k):
    """With coef = one-hot(g), uw = 1, rw = 0 the cell returns B_g(x): compare bitwise with the
    golden basis values (knot-exact, knot +- 1 ulp, out-of-range and random x)."""
    from kan_sr import ops
    g = load_golden("bspline")
    xs = t(g[f"G{G}_k{k}_x"])
    ref = g[f"G{G}_k{k}_basis"]
    N, K = xs.numel(), G + k
    O = K
    w = torch.zeros(N, O * (K + 2))
    w[:, :O * K] = torch.eye(K).reshape(1, -1)
    w[:, O * K:O * K + O] = 1.0
    y = ops.kan_cell(xs.reshape(1, 1, N, 1).cuda(), w.cuda(), O, G, k, [-1, 1], 0, True)
    got = y[0, :, :, 0].t().cpu().numpy()
    assert np.array_equal(got, ref)


def test_kan_cell_random_vs_oracle_large():
    from kan_sr import ops
    torch.manual_seed(3)
    B, H, W, r, G, k = 3, 37, 53, 4, 5, 3
    O = r * r
    P = O * (G + k + 2)
    x = (torch.rand(B, 1, H, W) * 2.6 - 1.3)
    w = torch.randn(B, P, H, W)
    ref = orc.pixel_shuffle(orc.kan_cell(x, w, O, G, k, [-1, 1]), r)
    y = ops.kan_cell(x.cuda(), w.cuda(), O, G, k, [-1, 1], shuffle_r=r)
    assert_fwd(y, ref)


@pytest.mark.parametrize("r,H,W", [(4, 16, 16), (2, 9, 11), (4, 5, 7), (3, 6, 5), (2, 10, 70), (4, 7, 70)])
def test_bilinear(r, H, W):
    from kan_sr import ops
    torch.manual_seed(1)
    x = torch.rand(2, 3, H, W)
    ref = torch.nn.functional.interpolate(x[:, 1:], scale_factor=r, mode="bilinear")
    assert_fwd(orc.bilinear_up(x[:, 1:], r), ref)
    out = torch.zeros(2, 3, H * r, W * r, device="cuda")
    ops.bilinear_up_into(x.cuda(), out, r, 1, 2, 1)
    assert_fwd(out[:, 1:], ref)
    assert float(out[:, 0].abs().max()) == 0.0
