"""Pins the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Runs without a GPU.

Tolerances (fp32 vs reference fp32): forward max|d| <= 1e-5 + 1e-4*|ref|;
gradients rel-L2 <= 1e-4 per tensor.  The basis functions are bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2, t
from oracle import convsepkan_oracle as orc

FWD_ATOL, FWD_RTOL, GRAD_REL = 1e-5, 1e-4, 1e-4


def assert_fwd(a, ref):
    assert a.shape == ref.shape
    err = (a - ref).abs()
    bound = FWD_ATOL + FWD_RTOL * ref.abs()
    assert bool((err <= bound).all()), f"max err {float(err.max()):.3e}"


@pytest.mark.parametrize("G,k", [(5, 3), (16, 3), (11, 3), (5, 1), (8, 2)])
def test_knots_and_basis_bit_exact(G, k):
    g = load_golden("bspline")
    knots = orc.knot_vector(G, k, [-1.0, 1.0])
    assert np.array_equal(knots.numpy(), g[f"G{G}_k{k}_knots"])
    x = t(g[f"G{G}_k{k}_x"])
    b = orc.bspline_basis(x[:, None], knots, k)[:, 0]
    assert np.array_equal(b.numpy(), g[f"G{G}_k{k}_basis"])


def test_basis_known_values():
    """Spot values measured on the reference: x=-1 -> [1/6,2/3,1/6,0..]; x=1.5 sums to
    0.682 (partial support past the range); |x|>=2.2 -> all zero (G=5,k=3)."""
    knots = orc.knot_vector(5, 3, [-1.0, 1.0])
    b = orc.bspline_basis(torch.tensor([[-1.0], [1.5], [2.2], [-2.3]]), knots, 3)[:, 0]
    assert torch.allclose(b[0, :3], torch.tensor([1 / 6, 2 / 3, 1 / 6]), atol=1e-6)
    assert float(b[0, 3:].abs().max()) == 0.0
    assert abs(float(b[1].sum()) - 0.682) < 1e-3
    assert float(b[2:].abs().max()) == 0.0


def test_basis_reference_main_parameters():
    g = load_golden("bspline")
    knots = orc.knot_vector(11, 3, [-1.0, 1.0])
    b = orc.bspline_basis(t(g["main_x"]), knots, 3)          # [2, I, K]
    ref = t(g["main_basis"])                                 # [2, O, I, K]
    for o in range(ref.shape[1]):
        assert torch.equal(b, ref[:, o])


@pytest.mark.parametrize("O,I,G,k", [(4, 1, 5, 3), (16, 1, 5, 3), (7, 5, 5, 3), (16, 1, 16, 3)])
def test_kan_layer_fwd_grads(O, I, G, k):
    g = load_golden("kan_layer")
    tag = f"O{O}_I{I}_G{G}_k{k}_"
    x, coef, uw, rw = (t(g[tag + n]).requires_grad_() for n in ("x", "coef", "uw", "rw"))
    y = orc.kan_eval(x, coef, uw, rw, orc.knot_vector(G, k, [-1, 1]), k)
    # the reference's coef[None] broadcast leaves a leading singleton: [1, N, O]
    assert_fwd(y.detach(), t(g[tag + "y"])[0])
    (y * t(g[tag + "gy"])[0]).sum().backward()
    for n, v in (("gx", x), ("gcoef", coef), ("guw", uw), ("grw", rw)):
        assert rel_l2(v.grad, t(g[tag + n])) <= GRAD_REL, n


@pytest.mark.parametrize("r,G,k,I", [(2, 5, 3, 1), (4, 5, 3, 1), (4, 16, 3, 1), (2, 5, 3, 3)])
def test_kan_cell_shuffle(r, G, k, I):
    g = load_golden("kan_cell")
    tag = f"r{r}_G{G}_k{k}_I{I}_"
    O = r * r
    assert orc.kan_param_count(I, O, G, k) == int(g[tag + "size"])
    K = G + k
    assert list(g[tag + "offsets"]) == [0, O * I * K, O * I * (K + 1), O * I * (K + 2)]
    x = t(g[tag + "x"]).requires_grad_()
    w = t(g[tag + "w"]).requires_grad_()
    y = orc.pixel_shuffle(orc.kan_cell(x, w, O, G, k, [-1, 1]), r)
    assert_fwd(y.detach(), t(g[tag + "y"]))
    (y * t(g[tag + "gy"])).sum().backward()
    assert rel_l2(x.grad, t(g[tag + "gx"])) <= GRAD_REL
    assert rel_l2(w.grad, t(g[tag + "gw"])) <= GRAD_REL


@pytest.mark.parametrize("name,kind", [("ffn_1_64", "ffn"), ("ffn_64_64", "ffn"), ("gated", "gated")])
def test_ffn_blocks(name, kind):
    g = load_golden("ffn")
    p = {k[len(name) + 3:]: t(v).requires_grad_() for k, v in g.items() if k.startswith(name + ".p.")}
    x = t(g[name + ".x"]).requires_grad_()
    y = orc.ffn(x, p, "") if kind == "ffn" else orc.gated_ffn(x, p, "")
    assert_fwd(y.detach(), t(g[name + ".y"]))
    (y * t(g[name + ".gy"])).sum().backward()
    assert rel_l2(x.grad, t(g[name + ".gx"])) <= GRAD_REL
    for n, v in p.items():
        assert rel_l2(v.grad, t(g[f"{name}.g.{n}"])) <= GRAD_REL, n


@pytest.mark.parametrize("B", [1, 2, 4])
def test_gazetr_eval(B):
    """Pins the batch-axis attention, including its B=1 collapse."""
    g = load_golden("gazetr")
    p = {k[2:]: t(v).requires_grad_() for k, v in g.items() if k.startswith("p.")}
    x = t(g[f"B{B}.x"]).requires_grad_()
    y = orc.gazetr(x, p, "")
    assert_fwd(y.detach(), t(g[f"B{B}.y"]))
    (y * t(g[f"B{B}.gy"])).sum().backward()
    assert rel_l2(x.grad, t(g[f"B{B}.gx"])) <= GRAD_REL
    for n, v in p.items():
        assert rel_l2(v.grad, t(g[f"B{B}.g.{n}"])) <= GRAD_REL, n


@pytest.mark.parametrize("r,G,k", [(2, 5, 3), (4, 5, 3), (4, 16, 3)])
def test_full_model(r, G, k):
    g = load_golden(f"model_r{r}_G{G}_k{k}")
    p = {n[2:]: t(v).requires_grad_() for n, v in g.items() if n.startswith("p.")}
    assert {n: tuple(v.shape) for n, v in p.items()} == orc.param_shapes(r * r, G, k)
    x = t(g["x"])
    y = orc.conv_sep_kan_forward(x, p, out_channels=[r * r], grid_size=G, spline_order=k,
                                 grid_range=[-1, 1], upscale=r)
    assert_fwd(y.detach(), t(g["y"]))
    (y * t(g["gy"])).sum().backward()
    for n, v in p.items():
        ref, ref64 = t(g["g." + n]), t(g["g64." + n])
        assert rel_l2(v.grad, ref) <= GRAD_REL, n
        # error against the fp64 graph is no worse than twice the reference's own
        e_me = float((v.grad.double() - ref64).norm())
        e_ref = float((ref.double() - ref64).norm())
        assert e_me <= 2 * e_ref + 1e-6 * float(ref64.norm()) + 1e-9, n


@pytest.mark.parametrize("r,G,k", [(2, 5, 3), (4, 5, 3), (4, 16, 3)])
def test_full_model_input_gradient(r, G, k):
    """Gradient with respect to the input image (the reference's autograd with x.requires_grad: chroma through the
    bilinear upsampling, luma through the KAN evaluation and through the encoder)."""
    g, gi = load_golden(f"model_r{r}_G{G}_k{k}"), load_golden("model_input_grad")
    p = {n[2:]: t(v) for n, v in g.items() if n.startswith("p.")}
    x = t(g["x"]).requires_grad_()
    y = orc.conv_sep_kan_forward(x, p, out_channels=[r * r], grid_size=G, spline_order=k, grid_range=[-1, 1], upscale=r)
    (y * t(g["gy"])).sum().backward()
    ref, ref64 = t(gi[f"r{r}_G{G}.gx"]), t(gi[f"r{r}_G{G}.gx64"])
    for c in range(3):
        assert rel_l2(x.grad[:, c], ref[:, c]) <= GRAD_REL, c
    e_me, e_ref = float((x.grad.double() - ref64).norm()), float((ref.double() - ref64).norm())
    assert e_me <= 2 * e_ref + 1e-6 * float(ref64.norm()) + 1e-9


def test_full_model_fp64_matches_reference_fp64():
    g = load_golden("model_r4_G5_k3")
    p = {n[2:]: t(v).double() for n, v in g.items() if n.startswith("p.")}
    y = orc.conv_sep_kan_forward(t(g["x"]).double(), p, out_channels=[16], grid_size=5,
                                 spline_order=3, grid_range=[-1, 1], upscale=4)
    assert float((y - t(g["y64"])).abs().max()) < 1e-9


def test_bilinear_matches_chroma_planes():
    g = load_golden("model_r4_G5_k3")
    x = t(g["x"])
    up = orc.bilinear_up(x[:, 1:], 4)
    assert_fwd(up, t(g["y"])[:, 1:])


def test_cfg1_shape_smoke():
    g = load_golden("cfg1_smoke")
    p = {n[2:]: t(v) for n, v in g.items() if n.startswith("p.")}
    with torch.no_grad():
        y = orc.conv_sep_kan_forward(t(g["x"]), p, out_channels=[4], grid_size=5, spline_order=3,
                                     grid_range=[-1, 1], upscale=2)
    assert y.shape == (4, 3, 128, 128)
    assert_fwd(y[:, :, :4, :4], t(g["y_corner"]))
    assert_fwd(y[1, 0, 77], t(g["y_row"]))
    assert abs(float(y.double().sum()) - float(g["y_sum"])) < 1e-2


@pytest.mark.parametrize("tag", ["o64", "o160", "o336"])
def test_chan_attention(tag):
    """ConvSepKanAttention (conv_sep_kan/conv_sep_kan_attention.py:48-71) forward and gradients."""
    g = load_golden("chan_attention")
    p = {k[len(tag) + 3:]: t(v).requires_grad_() for k, v in g.items() if k.startswith(tag + ".p.")}
    x = t(g[tag + ".x"]).requires_grad_()
    y = orc.chan_attention(x, p, "")
    assert_fwd(y.detach(), t(g[tag + ".y"]))
    (y * t(g[tag + ".gy"])).sum().backward()
    assert rel_l2(x.grad, t(g[tag + ".gx"])) <= GRAD_REL
    for n, v in p.items():
        if n.startswith("a_dw."):
            assert v.grad is None and f"{tag}.g.{n}" not in g      # defined, never used by the forward
        else:
            assert rel_l2(v.grad, t(g[f"{tag}.g.{n}"])) <= GRAD_REL, n


def test_color_conversion():
    """ycbcr_to_rgb against the reference function's outputs; rgb_to_ycbcr against the reference's formula evaluated
    with its own T and O (the function itself raises on multi-pixel images), known values and the round trip."""
    g = load_golden("color")
    assert torch.equal(t(g["T"]), orc.YCBCR_T) and torch.equal(t(g["O"]), orc.YCBCR_O)
    assert (orc.ycbcr_to_rgb(t(g["ycbcr"])) - t(g["rgb_back"])).abs().max() <= 1e-6
    assert (orc.ycbcr_to_rgb(t(g["ycc2"])) - t(g["rgb2"])).abs().max() <= 1e-6
    ycc = orc.rgb_to_ycbcr(t(g["rgb"]))
    assert (ycc - t(g["ycbcr"])).abs().max() <= 1e-6
    assert (ycc[0, :, 0, 0] - torch.tensor([16., 128., 128.]) / 255).abs().max() <= 1e-6      # black
    assert (ycc[0, :, 0, 1] - torch.tensor([235., 128., 128.]) / 255).abs().max() <= 2e-6     # white
    assert (t(g["rgb_back"]) - t(g["rgb"])).abs().max() <= 1e-5


@pytest.mark.parametrize("tag,heads", [("att8", 2), ("att32", 4)])
def test_unet_anchor_attention(tag, heads):
    """SelfAttention (conv_sep_kan/unet_blocks.py:91-178) forward and gradients."""
    g = load_golden("unet")
    p = {k[len(tag) + 3:]: t(v).requires_grad_() for k, v in g.items() if k.startswith(tag + ".p.")}
    x = t(g[tag + ".x"]).requires_grad_()
    y = orc.anchor_attention(x, p, "", heads)
    assert_fwd(y.detach(), t(g[tag + ".y"]))
    (y * t(g[tag + ".gy"])).sum().backward()
    assert rel_l2(x.grad, t(g[tag + ".gx"])) <= GRAD_REL
    for n, v in p.items():
        assert rel_l2(v.grad, t(g[f"{tag}.g.{n}"])) <= GRAD_REL, n


@pytest.mark.parametrize("tag", ["u40", "u160", "odd"])
def test_unet(tag):
    """UNet (conv_sep_kan/unet.py:7-42): eval forward, train forward with the running-statistics update and, for the
    small case, the parameter gradients of the train-mode pass.  "odd" is a 40x52 image: two poolings floor and two Up
    blocks zero-pad (unet_blocks.py:68-73)."""
    g = load_golden("unet")
    p = {k[len(tag) + 4:]: t(v) for k, v in g.items() if k.startswith(tag + ".sd.")}
    p = {k: (v.requires_grad_() if v.dtype == torch.float32 and "running" not in k else v) for k, v in p.items()}
    x = t(g[tag + ".x"])
    with torch.no_grad():
        assert_fwd(orc.unet(x, p, "", False), t(g[tag + ".y_eval"]))
    stats = {}
    y = orc.unet(x, p, "", True, stats)
    assert_fwd(y.detach(), t(g[tag + ".y_train"]))
    for k, v in stats.items():
        ref = t(g[f"{tag}.after.{k}"])
        if "num_batches" in k:
            assert int(v) == int(ref)
        else:
            assert rel_l2(v.detach(), ref) <= 1e-6, k
    if tag + ".gy" in g:
        (y * t(g[tag + ".gy"])).sum().backward()
        for n, v in p.items():
            if v.requires_grad:
                assert rel_l2(v.grad, t(g[f"{tag}.g.{n}"])) <= 5e-4, n


def test_attention_head_gradients_are_ill_conditioned_for_bf16_features():
    """Why the cfg-4 wiring (channel-attention head, G = 16) has its own bf16 gradient bound: in the ORACLE itself, rounding
    nothing but the head's input features (the trunk's final-LayerNorm output) to bf16 in the forward moves the parameter
    gradients outside the bf16 contract tolerance of the plain model (cosine >= 0.999, rel-L2 <= 5e-2 for tensors of >= 64
    elements).  The head's normalised Gram product + softmax amplifies a 2^-9 relative perturbation of its input about
    twenty-fold in the gradients, so no bf16 evaluation of the trunk can meet that bound; the HIP path's bf16 modes are held
    to cosine >= 0.99 / rel-L2 <= 0.15 there (tests/test_gpu_chan_attention.py)."""
    class RoundFwd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.bfloat16().float()

        @staticmethod
        def backward(ctx, g):
            return g

    torch.manual_seed(5)
    r, G, B, H, W = 4, 16, 16, 32, 32
    from kan_sr.module import ConvSepKan          # parameter container only (default initialisation); no kernel runs
    m = ConvSepKan([1], [r * r], [3], G, 3, 0.1, [-1, 1], r, attention=True).eval()
    with torch.no_grad():
        for n, prm in m.named_parameters():
            if "norm" in n:
                prm.add_(0.1 * torch.randn_like(prm))
    params = {n: v.detach().clone() for n, v in m.state_dict().items()}
    x = torch.rand(B, 3, H, W)
    gy = torch.randn(B, 3, r * H, r * W, generator=torch.Generator().manual_seed(5))
    grads = []
    orig = orc.chan_attention
    for rounded in (False, True):
        orc.chan_attention = (lambda xx, p, prefix, **kw: orig(RoundFwd.apply(xx), p, prefix, **kw)) if rounded else orig
        try:
            p = {n: v.clone().requires_grad_() for n, v in params.items()}
            y = orc.conv_sep_kan_forward(x, p, out_channels=[r * r], grid_size=G, spline_order=3, grid_range=[-1, 1], upscale=r,
                                         attention=True)
            (y * gy).sum().backward()
        finally:
            orc.chan_attention = orig
        grads.append({n: v.grad for n, v in p.items() if v.grad is not None})
    outside = []
    for n, ref in grads[0].items():
        if ref.numel() < 64:
            continue
        a, b = grads[1][n].double().flatten(), ref.double().flatten()
        cos = float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30))
        if cos < 0.999 or rel_l2(a, b) > 5e-2:
            outside.append((n, cos))
    assert len(outside) >= 5, outside          # measured: 20 of the tensors, worst cosine 0.9977 (layer 1 linear1.weight)
    assert all(c >= 0.99 for _, c in outside), outside


def _sm_params(g, tag):
    pre = tag + ".p."
    return {k[len(pre):]: t(v) for k, v in g.items() if k.startswith(pre)}


def test_seemore_blocks_oracle_vs_reference(golden):
    """The constructible blocks of conv_sep_kan/seemore.py (:114-155, :188-410), restated in the oracle, against outputs of
    the reference's own modules (tests/golden/seemore.npz, make_golden.py::gold_seemore): forward |d| <= 1e-5 + 1e-4 |ref|."""
    g = golden("seemore")
    x, k, xb = t(g["x"]), t(g["k"]), t(g["xb"])

    def close(got, tag):
        ref = t(g[tag + ".y"])
        assert got.shape == ref.shape, (tag, got.shape, ref.shape)
        err = (got - ref).abs()
        assert bool((err <= FWD_ATOL + FWD_RTOL * ref.abs()).all()), (tag, float(err.max()))

    for ks in (3, 11):
        close(orc.sm_striped_conv(x, _sm_params(g, f"striped{ks}"), ""), f"striped{ks}")
    close(orc.sm_layer_norm_cf(x, _sm_params(g, "ln"), ""), "ln")
    close(orc.sm_gated_ffn(x, _sm_params(g, "gffn"), ""), "gffn")
    close(orc.sm_striped_conv_former(x, _sm_params(g, "scf"), ""), "scf")
    close(orc.sm_sme(x, _sm_params(g, "sme"), ""), "sme")
    for low in (2, 7, 64):
        close(orc.sm_expert(x, k, _sm_params(g, f"expert{low}"), ""), f"expert{low}")
    close(orc.sm_router(x, _sm_params(g, "router"), ""), "router")
    pl = _sm_params(g, "moelayer_train")
    close(orc.sm_moe_layer(x, k, pl, "", topk=2), "moelayer_train")
    close(orc.sm_moe_layer(x[:1], k[:1], pl, "", topk=2), "moelayer_eval")
    for space in ("linear", "exp", "double"):
        pb = _sm_params(g, f"moeblock_{space}_train")
        close(orc.sm_moe_block(xb, pb, "", topk=2, recursive=2), f"moeblock_{space}_train")
        close(orc.sm_moe_block(xb[1:], pb, "", topk=2, recursive=2), f"moeblock_{space}_eval")
    close(orc.sm_moe_block(xb[:, :, :, :16].contiguous(), _sm_params(g, "moeblock_shuffle_train"), "", topk=1, recursive=1,
                           use_shuffle=True), "moeblock_shuffle_train")
