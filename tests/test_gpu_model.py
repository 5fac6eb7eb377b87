"""GPU parity of the encoder kernels and the whole ConvSepKan forward against the CPU oracle and the
reference-generated golden vectors.  fp32 tolerance: max|d| <= 1e-5 + 1e-4*|ref|."""
import ctypes
from ctypes import c_float, c_int, c_int64, c_uint64

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2, t
from oracle import convsepkan_oracle as orc

pytestmark = pytest.mark.gpu
FWD_ATOL, FWD_RTOL, GRAD_REL = 1e-5, 1e-4, 1e-4


def assert_fwd(a, ref, atol=FWD_ATOL, rtol=FWD_RTOL):
    a, ref = a.detach().cpu(), ref.detach().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    err = (a - ref).abs()
    assert bool((err <= atol + rtol * ref.abs()).all()), f"max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e})"


def nhwc(x):   # [B,C,H,W] -> [B*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def nchw(x, B, H, W):
    return x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


def cuda_params(p):
    return {k: v.detach().cuda().contiguous() for k, v in p.items()}


@pytest.mark.parametrize("B,H,W", [(2, 9, 11), (1, 37, 70), (3, 4, 32)])
def test_encoder_trunk_stages(B, H, W):
    """stem / gate / pointwise / dwrelu kernels one by one against the oracle's conv blocks."""
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(5)
    p = orc.random_params(16, 5, 3, seed=7)
    pre = "layers.0.encoder.encoder."
    x = torch.rand(B, 3, H, W)
    pc = cuda_params(p)
    xc = x.cuda()
    g = engine.Geometry(xc, 4, 5, 3, [-1, 1])
    saves = {}
    feat = engine.encoder_forward(xc, pc, pre, g, saves)
    # oracle stage by stage
    y1 = orc.ffn(x[:, :1], p, pre + "ffn1.")
    t2 = torch.relu(torch.nn.functional.conv2d(y1, p[pre + "ffn2.fn_1.0.weight"], p[pre + "ffn2.fn_1.0.bias"]))
    assert_fwd(nchw(saves["t2"], B, H, W), t2)
    x3 = torch.sigmoid(orc.gated_ffn(y1, p, pre + "ffn2."))
    assert_fwd(nchw(saves["x3"], B, H, W), x3)
    x4 = orc.ffn(x3, p, pre + "ffn3.")
    assert_fwd(nchw(saves["x4"], B, H, W), x4)
    x6 = torch.sigmoid(orc.gated_ffn(x4, p, pre + "ffn4."))
    assert_fwd(nchw(saves["x6"], B, H, W), x6)
    ref = orc.gazetr_features(x[:, :1], p, pre)
    assert_fwd(nchw(feat, B, H, W), ref, atol=2e-5)


@pytest.mark.parametrize("S", [1, 2, 3, 8, 32, 64, 100])
def test_batch_tx_layer_vs_oracle(S):
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(S)
    HW = 50 if S < 64 else 7
    p = orc.random_params(16, 5, 3, seed=3, scale=2.0)
    pre = "layers.0.encoder.encoder.encoder.layers.0."
    s = torch.randn(HW, S, 64)
    ref = orc.batch_tx_layer(s, p, pre)
    pc = cuda_params(p)
    sin = s.permute(1, 0, 2).contiguous().cuda()          # [S, HW, 64]
    out = torch.empty_like(sin)
    plist = [pc[pre + n] for n in engine.TX_PARAM_NAMES] + [None, None]
    check(lib().ksr_batch_tx_layer_fwd(ptr(sin), ptr(out), engine._ptr_array(plist), None, c_int(S), c_int(HW),
                                       c_float(1e-5), c_float(0.0), c_uint64(0), None, stream()), "tx")
    assert_fwd(out.permute(1, 0, 2), ref, atol=2e-5)


@pytest.mark.parametrize("HW", [50, 256, 1000, 4133])
def test_single_image_layer_split_fp32_vs_exact_and_float64(HW):
    """The single-image transformer layer with its fp32 products split over bf16 matrix instructions (ksr_set_f32_split, the
    default) against the fp32 matrix instruction and against the oracle in float64: both forms within the fp32 tolerance of
    the oracle, the split form no further from float64 than 1.5 x the exact form (+ 1e-7), ragged tile counts of the
    eight-wave workgroups included."""
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(HW)
    p = orc.random_params(16, 5, 3, seed=4, scale=2.0)
    pre = "layers.0.encoder.encoder.encoder.layers.1."
    s = torch.randn(HW, 1, 64)
    ref32 = orc.batch_tx_layer(s, p, pre)
    ref64 = orc.batch_tx_layer(s.double(), {k: v.double() for k, v in p.items()}, pre)
    pc = cuda_params(p)
    sin = s.permute(1, 0, 2).contiguous().cuda()
    plist = [pc[pre + n] for n in engine.TX_PARAM_NAMES] + [None, None]
    outs = {}
    try:
        for split in (False, True):
            engine.set_f32_split(split)
            assert int(lib().ksr_get_f32_split()) == int(split)
            out = torch.full_like(sin, float("nan"))
            check(lib().ksr_batch_tx_layer_fwd(ptr(sin), ptr(out), engine._ptr_array(plist), None, c_int(1), c_int(HW),
                                               c_float(1e-5), c_float(0.0), c_uint64(0), None, stream()), "tx")
            outs[split] = out.permute(1, 0, 2).cpu()
            assert_fwd(outs[split], ref32, atol=2e-5)
    finally:
        engine.set_f32_split(None)
    e_exact = float((outs[False].double() - ref64).abs().max())
    e_split = float((outs[True].double() - ref64).abs().max())
    assert e_split <= 1.5 * e_exact + 1e-7, (e_split, e_exact)
    assert not torch.equal(outs[False], outs[True]) or HW < 64      # two different instruction sequences


@pytest.mark.parametrize("B", [1, 2, 4])
def test_gazetr_golden_features_through_feed(B):
    """GazeTR golden (P=40, i.e. upscale 2): encoder + fused feed/KAN output equals the oracle's KAN
    applied to the reference's own `gaze` tensor."""
    from kan_sr import engine
    g = load_golden("gazetr")
    pre = "layers.0.encoder.encoder."
    p = {pre + k[2:]: t(v) for k, v in g.items() if k.startswith("p.")}
    x1 = t(g[f"B{B}.x"])
    x = torch.cat([x1, torch.zeros(B, 2, 8, 8)], 1).contiguous()
    geo = engine.Geometry(x.cuda(), 2, 5, 3, [-1, 1])
    out = engine.forward_infer(x.cuda(), cuda_params(p), pre, geo)
    ref = orc.pixel_shuffle(orc.kan_cell(x1, t(g[f"B{B}.y"]), 4, 5, 3, [-1, 1]), 2)
    assert_fwd(out[:, :1], ref, atol=2e-5)


@pytest.mark.parametrize("r,G,k", [(2, 5, 3), (4, 5, 3), (4, 16, 3)])
def test_full_model_forward_golden(r, G, k):
    from kan_sr.module import ConvSepKan
    g = load_golden(f"model_r{r}_G{G}_k{k}")
    m = ConvSepKan([1], [r * r], [3], G, k, 0.1, [-1, 1], r)
    sd = {n[2:]: t(v) for n, v in g.items() if n.startswith("p.")}
    assert set(sd) == set(m.state_dict().keys())
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        y = m(t(g["x"]).cuda())
    assert_fwd(y, t(g["y"]), atol=2e-5)


@pytest.mark.parametrize("r,G,k", [(2, 5, 3), (4, 5, 3), (4, 16, 3)])
def test_full_model_input_gradient_golden(r, G, k):
    """x.requires_grad: the gradient with respect to the input image against the reference's autograd (chroma planes
    through the bilinear upsampling's adjoint; luma through the KAN evaluation's own use of the pixel value, with the
    order-0 indicator constant, and through the encoder's first convolution), parameter gradients unchanged."""
    g, m = _load_model_golden(r, G, k)
    gi = load_golden("model_input_grad")
    x = t(g["x"]).cuda().requires_grad_()
    y = m(x)
    assert_fwd(y, t(g["y"]), atol=2e-5)
    (y * t(g["gy"]).cuda()).sum().backward()
    ref, ref64 = t(gi[f"r{r}_G{G}.gx"]), t(gi[f"r{r}_G{G}.gx64"])
    assert x.grad is not None and x.grad.shape == x.shape
    for c in range(3):
        assert rel_l2(x.grad[:, c].cpu(), ref[:, c]) <= 1e-4, c
    e_me, e_ref = float((x.grad.cpu().double() - ref64).norm()), float((ref.double() - ref64).norm())
    assert e_me <= 2 * e_ref + 1e-6 * float(ref64.norm())
    for n, prm in m.named_parameters():
        assert rel_l2(prm.grad.cpu(), t(g["g." + n])) <= 1e-4, n
    # only the image requires a gradient
    m.zero_grad()
    for prm in m.parameters():
        prm.requires_grad_(False)
    x2 = t(g["x"]).cuda().requires_grad_()
    (m(x2) * t(g["gy"]).cuda()).sum().backward()
    assert torch.equal(x2.grad, x.grad)


def test_input_gradient_odd_sizes_vs_oracle():
    """Image sizes that are not multiples of the tile sizes, B = 3 (the transformer mixes the images), upscale 2."""
    from kan_sr.module import ConvSepKan
    torch.manual_seed(5)
    m = ConvSepKan([1], [4], [3], 5, 3, 0.1, [-1, 1], 2).eval()
    p = {n: v.detach().clone() for n, v in m.state_dict().items()}
    x = (torch.rand(3, 3, 9, 13) * 1.4 - 0.2).requires_grad_()
    y = orc.conv_sep_kan_forward(x, p, out_channels=[4], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=2)
    gy = torch.randn(y.shape)
    (y * gy).sum().backward()
    xg = x.detach().cuda().requires_grad_()
    (m.cuda()(xg) * gy.cuda()).sum().backward()
    for c in range(3):
        assert rel_l2(xg.grad[:, c].cpu(), x.grad[:, c]) <= 1e-4, c


def test_input_gradient_bf16_storage_vs_fp32_mode():
    """The image gradient in the bf16 modes (build-side modes; the luma path runs through every bf16 kernel of the
    backward) against the exact mode at 8 x 32 x 32: cosine >= 0.999, rel-L2 <= 5e-2; the chroma planes, which touch no
    bf16 kernel, bit-equal."""
    from kan_sr.module import ConvSepKan
    torch.manual_seed(21)
    m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4).cuda().eval()
    x0 = torch.rand(8, 3, 32, 32, device="cuda")
    gy = torch.randn(8, 3, 128, 128, device="cuda")
    grads = {}
    for prec in ("fp32", "bf16", "bf16s"):
        m.precision = prec
        x = x0.clone().requires_grad_()
        (m(x) * gy).sum().backward()
        grads[prec] = x.grad.clone()
    m.precision = "fp32"
    for prec in ("bf16", "bf16s"):
        assert torch.equal(grads[prec][:, 1:], grads["fp32"][:, 1:])
        a, b = grads[prec][:, 0].flatten().double(), grads["fp32"][:, 0].flatten().double()
        assert float(torch.dot(a, b) / (a.norm() * b.norm())) >= 0.999, prec
        assert rel_l2(grads[prec][:, 0].cpu(), grads["fp32"][:, 0].cpu()) <= 5e-2, prec


def test_cfg1_shape_golden():
    from kan_sr.module import ConvSepKan
    g = load_golden("cfg1_smoke")
    m = ConvSepKan([1], [4], [3], 5, 3, 0.1, [-1, 1], 2)
    m.load_state_dict({n[2:]: t(v) for n, v in g.items() if n.startswith("p.")})
    m = m.cuda().eval()
    with torch.no_grad():
        y = m(t(g["x"]).cuda()).cpu()
    assert y.shape == (4, 3, 128, 128)
    assert_fwd(y[:, :, :4, :4], t(g["y_corner"]), atol=2e-5)
    assert_fwd(y[1, 0, 77], t(g["y_row"]), atol=2e-5)
    assert abs(float(y.double().sum()) - float(g["y_sum"])) < 5e-2


def test_default_init_matches_reference_rng_order():
    """Constructing under the same torch seed reproduces the reference's initial conv / linear weights."""
    from kan_sr.module import ConvSepKan
    g = load_golden("model_r4_G5_k3")
    torch.manual_seed(600 + 10 * 4 + 5)
    m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4)
    for n, v in m.state_dict().items():
        if "norm" in n:
            continue
        assert torch.equal(v, t(g["p." + n])), n


def _load_model_golden(r, G, k):
    from kan_sr.module import ConvSepKan
    g = load_golden(f"model_r{r}_G{G}_k{k}")
    m = ConvSepKan([1], [r * r], [3], G, k, 0.1, [-1, 1], r)
    m.load_state_dict({n[2:]: t(v) for n, v in g.items() if n.startswith("p.")})
    return g, m.cuda().eval()


@pytest.mark.parametrize("r,G,k", [(2, 5, 3), (4, 5, 3), (4, 16, 3)])
def test_full_model_gradients_golden(r, G, k):
    """All 52 parameter gradients of the eval-mode graph against the reference's autograd (fp32 goldens),
    and against the reference's fp64 run: err(build, fp64) <= 2*err(ref fp32, fp64) + 1e-6."""
    g, m = _load_model_golden(r, G, k)
    y = m(t(g["x"]).cuda())
    assert_fwd(y, t(g["y"]), atol=2e-5)
    y.backward(t(g["gy"]).cuda())
    worst = 0.0
    for n, prm in m.named_parameters():
        ref, ref64 = t(g["g." + n]), t(g["g64." + n])
        got = prm.grad.cpu()
        rel = rel_l2(got, ref)
        worst = max(worst, rel)
        assert rel <= GRAD_REL, f"{n}: rel-L2 {rel:.3e}"
        e_me = float((got.double() - ref64).norm())
        e_ref = float((ref.double() - ref64).norm())
        assert e_me <= 2 * e_ref + 1e-6 * float(ref64.norm()) + 1e-9, f"{n}: {e_me:.3e} vs ref {e_ref:.3e}"


@pytest.mark.parametrize("B", [1, 2, 4])
def test_gazetr_golden_gradients(B):
    """GazeTR golden (P=40): parameter gradients for cotangent gy on the `gaze` tensor, obtained here by
    pushing gy through the oracle's KAN cell ... not needed: compare via the full chain instead."""
    from kan_sr.module import ConvSepKan
    g = load_golden("gazetr")
    pre = "layers.0.encoder.encoder."
    m = ConvSepKan([1], [4], [3], 5, 3, 0.1, [-1, 1], 2)
    m.load_state_dict({pre + k[2:]: t(v) for k, v in g.items() if k.startswith("p.")})
    x1 = t(g[f"B{B}.x"])
    x = torch.cat([x1, torch.zeros(B, 2, 8, 8)], 1).contiguous()
    # oracle chain on CPU with the same parameters (the oracle is pinned on this golden separately)
    p = {n: v.detach().clone().requires_grad_() for n, v in m.state_dict().items()}
    yr = orc.conv_sep_kan_forward(x, p, out_channels=[4], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=2)
    gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(B))
    (yr * gy).sum().backward()
    m = m.cuda().eval()
    y = m(x.cuda())
    assert_fwd(y, yr, atol=2e-5)
    y.backward(gy.cuda())
    for n, prm in m.named_parameters():
        rel = rel_l2(prm.grad.cpu(), p[n].grad)
        assert rel <= GRAD_REL, f"{n}: rel-L2 {rel:.3e}"


def test_gradients_odd_sizes_vs_oracle():
    """Ragged tiles: H, W not multiples of the 4x32 tile, batch not a power of two."""
    from kan_sr.module import ConvSepKan
    torch.manual_seed(11)
    B, H, W, r = 3, 13, 45, 4
    p0 = orc.random_params(16, 5, 3, seed=21)
    m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], r)
    m.load_state_dict(p0)
    x = torch.rand(B, 3, H, W)
    p = {n: v.clone().requires_grad_() for n, v in p0.items()}
    yr = orc.conv_sep_kan_forward(x, p, out_channels=[16], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=r)
    gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(5))
    (yr * gy).sum().backward()
    m = m.cuda().eval()
    y = m(x.cuda())
    assert_fwd(y, yr, atol=2e-5)
    y.backward(gy.cuda())
    for n, prm in m.named_parameters():
        rel = rel_l2(prm.grad.cpu(), p[n].grad)
        assert rel <= GRAD_REL, f"{n}: rel-L2 {rel:.3e}"


@pytest.mark.parametrize("r,G,k", [(2, 3, 3), (2, 8, 3), (2, 10, 3), (2, 16, 3), (4, 3, 3), (4, 8, 3), (4, 10, 3),
                                   (2, 5, 1), (2, 5, 2), (4, 5, 1), (4, 5, 2)])
def test_other_grid_sizes_vs_oracle(r, G, k):
    """The (grid size, spline order) pairs without a full-model reference fixture: forward and all gradients against
    the oracle, whose basis / KAN code is pinned by the bspline fixtures (orders 1-3) and the G = 5 / G = 16 model
    fixtures; one TrainStep in bf16 storage mode must also run."""
    from kan_sr.module import ConvSepKan
    torch.manual_seed(100 + 10 * r + G + k)
    B, H, W, O = 2, 10, 37, r * r
    p0 = orc.random_params(O, G, k, seed=31 + G)
    m = ConvSepKan([1], [O], [3], G, k, 0.1, [-1, 1], r)
    m.load_state_dict(p0)
    x = torch.rand(B, 3, H, W) * 1.6 - 0.3            # reaches the outer knot intervals as well
    p = {n: v.clone().requires_grad_() for n, v in p0.items()}
    yr = orc.conv_sep_kan_forward(x, p, out_channels=[O], grid_size=G, spline_order=k, grid_range=[-1, 1], upscale=r)
    gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(5))
    (yr * gy).sum().backward()
    m = m.cuda().eval()
    y = m(x.cuda())
    assert_fwd(y, yr, atol=2e-5)
    y.backward(gy.cuda())
    for n, prm in m.named_parameters():
        rel = rel_l2(prm.grad.cpu(), p[n].grad)
        # the 1- and 9-element stem tensors are cancelling sums over all pixels: fp32 summation order shows at 1e-4
        assert rel <= (GRAD_REL if prm.numel() >= 64 else 10 * GRAD_REL), f"{n}: rel-L2 {rel:.3e}"
    from kan_sr.train import TrainStep
    m.train()
    m.precision = "bf16s"
    ts = TrainStep(m, lr=1e-3, loss="l1+ssim")
    loss = float(ts.step(x.cuda().clamp(0, 1), torch.rand(B, 3, r * H, r * W).cuda()))
    assert loss == loss and 0.0 < loss < 2.0


def test_backward_is_bitwise_reproducible():
    g, m = _load_model_golden(4, 5, 3)
    x, gy = t(g["x"]).cuda(), t(g["gy"]).cuda()
    runs = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        m(x).backward(gy)
        runs.append({n: prm.grad.clone() for n, prm in m.named_parameters()})
    for n in runs[0]:
        assert torch.equal(runs[0][n], runs[1][n]), n


@pytest.mark.parametrize("S", [17, 32, 33, 64, 100, 128])
@pytest.mark.parametrize("HW", [5, 37])
def test_mfma_attention_split_layer_vs_oracle(S, HW):
    """Large local batch: QKV GEMM + MFMA attention + post kernel against the oracle layer."""
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(S * 100 + HW)
    p = orc.random_params(16, 5, 3, seed=3, scale=2.0)
    pre = "layers.0.encoder.encoder.encoder.layers.0."
    s = torch.randn(HW, S, 64)
    ref = orc.batch_tx_layer(s, p, pre)
    pc = cuda_params(p)
    sin = s.permute(1, 0, 2).contiguous().cuda()
    N = S * HW
    plist = [pc[pre + n] for n in engine.TX_PARAM_NAMES] + [None, None]
    qkv = torch.empty(N, 192, device="cuda")
    ctx = torch.empty(N, 64, device="cuda")
    out = torch.empty_like(sin)
    L = lib()
    check(L.ksr_pointwise_fwd(ptr(sin), ptr(plist[0]), ptr(plist[1]), ptr(qkv), c_int(192), c_int(0), c_int(S), c_int(HW),
                              c_int(1), stream()), "qkv")
    check(L.ksr_batch_attn_fwd(ptr(qkv), ptr(ctx), c_int(S), c_int(HW), c_float(0.0), c_uint64(0), stream()), "attn")
    check(L.ksr_batch_tx_layer_post_fwd(ptr(sin), ptr(ctx), ptr(out), engine._ptr_array(plist), None, c_int64(N),
                                        c_float(1e-5), c_float(0.0), c_uint64(0), None, stream()), "post")
    assert_fwd(out.permute(1, 0, 2), ref, atol=2e-5)
    # the fused kernel gives the same answer
    out2 = torch.empty_like(sin)
    check(L.ksr_batch_tx_layer_fwd(ptr(sin), ptr(out2), engine._ptr_array(plist), None, c_int(S), c_int(HW),
                                   c_float(1e-5), c_float(0.0), c_uint64(0), None, stream()), "tx")
    assert_fwd(out2, out, atol=2e-5)


@pytest.mark.parametrize("S,HW", [(3, 41), (8, 20), (20, 9), (32, 6), (64, 5), (80, 4), (100, 3)])
@pytest.mark.parametrize("p_drop", [0.0, 0.3])
def test_mfma_attention_matches_valu_attention(S, HW, p_drop):
    """Forward and backward MFMA attention kernels against the VALU kernels (same dropout masks), and the
    VALU backward against autograd of the oracle's attention when dropout is off."""
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(S + HW)
    N = S * HW
    qkv = torch.randn(N, 192, device="cuda")
    dctx = torch.randn(N, 64, device="cuda")
    L = lib()
    ctx = torch.empty(N, 64, device="cuda")
    check(L.ksr_batch_attn_fwd(ptr(qkv), ptr(ctx), c_int(S), c_int(HW), c_float(p_drop), c_uint64(99), stream()), "fwd")
    d1 = torch.empty(N, 192, device="cuda")
    d2 = torch.empty(N, 192, device="cuda")
    check(L.ksr_batch_attn_bwd(ptr(qkv), ptr(ctx), ptr(dctx), ptr(d1), c_int(S), c_int(HW), c_float(p_drop), c_uint64(99),
                               stream()), "bwd mfma")
    check(L.ksr_batch_tx_layer_bwd_attn(ptr(qkv), ptr(ctx), ptr(dctx), ptr(d2), c_int(S), c_int(HW), c_float(p_drop),
                                        c_uint64(99), stream()), "bwd valu")
    assert rel_l2(d1.cpu(), d2.cpu()) <= 2e-5
    if p_drop == 0.0:
        x = qkv.cpu().reshape(S, HW, 3, 2, 32).permute(2, 1, 3, 0, 4).clone().requires_grad_()   # [3, HW, head, S, 32]
        q, k, v = x[0], x[1], x[2]
        att = torch.softmax(q @ k.transpose(-1, -2) / 32 ** 0.5, -1)
        o = (att @ v).permute(2, 0, 1, 3).reshape(N, 64)
        assert_fwd(ctx, o.detach(), atol=2e-5)
        o.backward(dctx.cpu())
        ref = x.grad.permute(3, 1, 0, 2, 4).reshape(N, 192)
        assert rel_l2(d1.cpu(), ref) <= 1e-5


@pytest.mark.parametrize("precision", ["bf16", "bf16s"])
@pytest.mark.parametrize("r,G,k", [(2, 5, 3), (4, 5, 3), (4, 16, 3)])
def test_bf16_mfma_mode_golden(r, G, k, precision):
    """bf16 MFMA operands / fp32 accumulate, with fp32 ("bf16") or bf16 ("bf16s") storage of the activations
    between kernels (build-side modes the reference does not have).  Stated tolerance:
    forward rel-L2 <= 2e-2 on the Y plane; every parameter gradient of >= 64 elements cosine >= 0.995 and
    rel-L2 <= 0.1 on this 512-pixel input, where the rounding noise of the per-pixel products averages over few pixels
    (measured worst case 0.9983 / 0.058; the bf16 tolerance proper - cosine >= 0.999, rel-L2 <= 5e-2 - is held against
    the oracle at 16 384 pixels by test_gpu_benched_path.py::test_bf16_modes_vs_oracle_at_contract_tolerance); the four stem tensors of 1 or 9 elements (sums of ~1e5 cancelling terms, |g| ~ 1e-5)
    |d| <= 0.3 |ref| + 1e-5 sqrt(numel).  With bf16 storage, on this 512-pixel input (rounding noise averages over
    few pixels): cosine >= 0.98 and rel-L2 <= 0.2 (0.5 for the stem tensors); test_bf16_storage_large_batch pins
    the mode at a realistic pixel count."""
    g, m = _load_model_golden(r, G, k)
    m.precision = precision
    y = m(t(g["x"]).cuda())
    assert rel_l2(y[:, 0].detach().cpu(), t(g["y"])[:, 0]) <= 2e-2
    assert_fwd(y[:, 1:], t(g["y"])[:, 1:])           # chroma does not go through a GEMM
    y.backward(t(g["gy"]).cuda())
    for n, prm in m.named_parameters():
        got, ref = prm.grad.cpu().flatten().double(), t(g["g." + n]).flatten().double()
        cos = float((got * ref).sum() / (got.norm() * ref.norm()).clamp_min(1e-30))
        big = got.numel() >= 64
        if precision == "bf16s":
            cos_min, rel_max = (0.98, 0.2) if big else (0.98, 0.5)
        else:
            cos_min, rel_max = (0.995, 0.1) if big else (0.99, 0.3)
        if not big:
            # 1 or 9 elements, each the sum of 512 x 64 cancelling products of magnitude ~1e-5 that carry bf16 rounding
            # noise (2^-9 relative): the noise of the sum is ~1e-5 in absolute terms whatever the sum's own size
            # (|g| ~ 1e-5 here), so the bound has an absolute part
            err = float((got - ref).norm())
            assert err <= rel_max * float(ref.norm()) + 1e-5 * got.numel() ** 0.5, f"{n}: |d| {err:.3e} vs |ref| {float(ref.norm()):.3e}"
            continue
        assert cos >= cos_min, f"{n}: cos {cos:.5f}"
        assert rel_l2(got, ref) <= rel_max, f"{n}: rel-L2 {rel_l2(got, ref):.3e}"
    # and fp32 mode is restored by the next fp32 module call
    m.precision = "fp32"
    with torch.no_grad():
        assert_fwd(m.eval()(t(g["x"]).cuda()), t(g["y"]), atol=2e-5)


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
@pytest.mark.parametrize("S,HW,p_drop", [(3, 50, 0.0), (8, 40, 0.1), (32, 9, 0.0), (64, 130, 0.1), (100, 7, 0.1), (128, 5, 0.0),
                                         (64, 1500, 0.1), (32, 1300, 0.1), (12, 2101, 0.1), (48, 600, 0.1), (64, 1030, 0.0),
                                         (17, 1027, 0.0), (8, 4100, 0.1), (5, 9001, 0.1), (3, 8200, 0.0)])
def test_bf16_attention_matches_fp32_attention(S, HW, p_drop, storage):
    """bf16-operand attention kernels against the fp32 ones on the same inputs and dropout masks.  Stated
    tolerance (bf16 operands, fp32 softmax and accumulation): rel-L2 <= 2e-2 forward and backward.  The second
    row of cases gives every persistent workgroup of the bf16-storage kernels several passes (both staging buffers,
    the deferred stores) at each slot width (4 ... 64), with and without absent slots (S below its padded width,
    a pixel count that does not fill the last pass)."""
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(7 * S + HW)
    N = S * HW
    qkv = torch.randn(N, 192, device="cuda")
    dctx = torch.randn(N, 64, device="cuda")
    L = lib()
    res = {}
    lowp = "bf16s" if storage == "bf16" else "bf16"
    if storage == "bf16":       # the fp32 run sees the same (bf16-representable) inputs
        qkv, dctx = qkv.bfloat16().float(), dctx.bfloat16().float()
    try:
        for prec in ("fp32", lowp):
            engine.set_precision(prec)
            dt = engine.act_dtype()
            ctx = torch.full((N, 64), float("nan"), device="cuda", dtype=dt)
            dqkv = torch.full((N, 192), float("nan"), device="cuda", dtype=dt)
            q_in, d_in = qkv.to(dt), dctx.to(dt)
            check(L.ksr_batch_attn_fwd(ptr(q_in), ptr(ctx), c_int(S), c_int(HW), c_float(p_drop), c_uint64(5), stream()), "fwd")
            check(L.ksr_batch_attn_bwd(ptr(q_in), ptr(ctx), ptr(d_in), ptr(dqkv), c_int(S), c_int(HW), c_float(p_drop),
                                       c_uint64(5), stream()), "bwd")
            res["fp32" if prec == "fp32" else "bf16"] = (ctx.float().cpu(), dqkv.float().cpu())
    finally:
        engine.set_precision("fp32")
    assert bool(torch.isfinite(res["bf16"][0]).all()) and bool(torch.isfinite(res["bf16"][1]).all())
    assert rel_l2(res["bf16"][0], res["fp32"][0]) <= 2e-2
    for c0, name in ((0, "dq"), (64, "dk"), (128, "dv")):
        assert rel_l2(res["bf16"][1][:, c0:c0 + 64], res["fp32"][1][:, c0:c0 + 64]) <= 2e-2, name


def test_bf16_storage_large_batch():
    """bf16 activation storage against the exact fp32 mode of the same kernels at 32 768 pixels (8 x 64 x 64), eval
    mode.  Stated tolerance: forward rel-L2 <= 1e-2; gradients of tensors with >= 64 elements cosine >= 0.999 and
    rel-L2 <= 5e-2."""
    from kan_sr.module import ConvSepKan
    from kan_sr.init import init_like_training_setup
    torch.manual_seed(3)
    m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4).eval()
    init_like_training_setup(m, 0)
    m = m.cuda()
    x = torch.rand(8, 3, 64, 64, device="cuda")
    gy = torch.randn(8, 3, 256, 256, device="cuda")
    res = {}
    for prec in ("fp32", "bf16s"):
        m.precision = prec
        m.zero_grad(set_to_none=True)
        y = m(x)
        y.backward(gy)
        res[prec] = (y.detach().cpu(), {n: p.grad.detach().cpu().double().flatten() for n, p in m.named_parameters()})
    m.precision = "fp32"
    assert rel_l2(res["bf16s"][0][:, 0], res["fp32"][0][:, 0]) <= 1e-2
    worst = []
    for n, ref in res["fp32"][1].items():
        got = res["bf16s"][1][n]
        if got.numel() < 64:
            continue
        cos = float((got * ref).sum() / (got.norm() * ref.norm()).clamp_min(1e-30))
        worst.append((cos, rel_l2(got, ref), n))
    bad = [w for w in worst if w[0] < 0.999 or w[1] > 5e-2]
    assert not bad, bad


@pytest.mark.parametrize("B,H,W", [(2, 9, 11), (1, 37, 70), (3, 4, 32), (1, 64, 64)])
def test_fused_stem_gate_kernel_equals_the_three_separate_kernels(B, H, W):
    """ksr_stem_gate_fwd (ffn1 + ffn2 + ffn3.pointwise1 from the Y plane, relu(fn_1(ffn1)) folded to relu(A v + B))
    against ksr_stem_fwd -> ksr_gate_pw_fwd -> ksr_pointwise_fwd: same tensors to fp32 round-off of the folding."""
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(B * 100 + H)
    p = cuda_params(orc.random_params(16, 5, 3, seed=7))
    pre = "layers.0.encoder.encoder."
    x = torch.rand(B, 3, H, W).cuda()
    g = engine.Geometry(x, 4, 5, 3, [-1, 1])
    engine.set_precision("fp32")            # direct library calls: the mode is whatever this thread used last
    saves = {}
    engine.encoder_forward(x, p, pre, g, saves)
    L = lib()
    N = B * H * W
    dims = (c_int(B), c_int(H), c_int(W))
    t2 = torch.empty(N, 128, device="cuda")
    v = torch.empty(N, device="cuda")
    check(L.ksr_stem_fwd(ptr(x), c_int64(3 * H * W), ptr(p[pre + "ffn1.pointwise1.weight"]), ptr(p[pre + "ffn1.pointwise1.bias"]),
                         ptr(p[pre + "ffn1.depthwise.weight"]), ptr(p[pre + "ffn1.depthwise.bias"]),
                         ptr(p[pre + "ffn1.pointwise2.weight"]), ptr(p[pre + "ffn1.pointwise2.bias"]),
                         ptr(p[pre + "ffn2.fn_1.0.weight"]), ptr(p[pre + "ffn2.fn_1.0.bias"]), ptr(t2), ptr(v), *dims, stream()), "stem")
    x3, g2 = torch.empty(N, 64, device="cuda"), torch.empty(N, 64, device="cuda")
    check(L.ksr_gate_pw_fwd(ptr(t2), ptr(p[pre + "ffn2.gate.weight"]), ptr(p[pre + "ffn2.gate.bias"]), ptr(p[pre + "ffn2.fn_2.weight"]),
                            ptr(p[pre + "ffn2.fn_2.bias"]), ptr(x3), ptr(g2), None, *dims, stream()), "gate")
    t3 = torch.empty(N, 64, device="cuda")
    check(L.ksr_pointwise_fwd(ptr(x3), ptr(p[pre + "ffn3.pointwise1.weight"]), ptr(p[pre + "ffn3.pointwise1.bias"]), ptr(t3),
                              c_int(64), c_int(0), *dims, stream()), "pw")
    assert torch.equal(saves["v"], v)
    for name, ref in (("t2", t2), ("g2", g2), ("x3", x3), ("t3", t3)):
        assert_fwd(saves[name], ref, atol=1e-5)


@pytest.mark.parametrize("B,H,W", [(2, 9, 11), (3, 40, 32), (16, 64, 64)])
def test_fused_weight_gradient_equals_separate_wgrad(B, H, W):
    """bf16 storage mode: ksr_gate_pw_bwd_wg / ksr_pw2_bwd_wg (weight gradient accumulated inside the data-gradient
    kernel) against ksr_gate_pw_bwd / ksr_pw2_bwd + ksr_wgrad on the same bf16 tensors: identical data gradients, weight
    and bias gradients equal up to fp32 summation order, and both close to an fp64 evaluation."""
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(B + H)
    L = lib()
    N = B * H * W
    dims = (c_int(B), c_int(H), c_int(W))
    bf = lambda *s: torch.randn(*s, device="cuda").bfloat16()
    dz, gate, t, a4 = bf(N, 64), bf(N, 64), bf(N, 128).abs() * (torch.rand(N, 128, device="cuda") > 0.3).bfloat16(), bf(N, 64).relu()
    w = torch.randn(64, 64, device="cuda") * 0.1
    engine.set_precision("bf16s")
    try:
        slab = torch.empty(int(L.ksr_fused_wgrad_slab_floats()), device="cuda")
        ws = torch.empty(int(L.ksr_wgrad_slab_floats(c_int(64), c_int(64))), device="cuda")
        # ---- gated block tail
        dt0, dg0 = torch.zeros(N, 128, device="cuda").bfloat16(), torch.empty(N, 64, device="cuda").bfloat16()
        dt1, dg1 = torch.zeros(N, 128, device="cuda").bfloat16(), torch.empty(N, 64, device="cuda").bfloat16()
        dW0, db0, dW1, db1 = (torch.empty(s, device="cuda") for s in ((64, 64), (64,), (64, 64), (64,)))
        check(L.ksr_gate_pw_bwd(ptr(dz), ptr(w), ptr(gate), ptr(t), ptr(dt0), ptr(dg0), *dims, stream()), "gate_pw_bwd")
        check(L.ksr_wgrad(ptr(dz), c_int(64), c_int(0), c_int(64), ptr(t), c_int(128), c_int(0), c_int(64), ptr(gate), c_int(64),
                          c_int(0), c_int64(N), ptr(ws), ptr(dW0), ptr(db0), c_int(0), c_int(0), stream()), "wgrad")
        check(L.ksr_gate_pw_bwd_wg(ptr(dz), ptr(w), ptr(gate), ptr(t), ptr(dt1), ptr(dg1), ptr(slab), ptr(dW1), ptr(db1), *dims,
                                   stream()), "gate_pw_bwd_wg")
        assert torch.equal(dt0, dt1) and torch.equal(dg0, dg1)
        ref = dz.double().T @ (t[:, :64].double() * gate.double()).bfloat16().double()
        assert rel_l2(dW1.cpu(), ref.cpu()) <= 2e-3 and rel_l2(dW0.cpu(), ref.cpu()) <= 2e-3
        assert rel_l2(dW1.cpu(), dW0.cpu()) <= 1e-5
        assert rel_l2(db1.cpu(), dz.double().sum(0).cpu()) <= 1e-5 and rel_l2(db1.cpu(), db0.cpu()) <= 1e-5
        # ---- FFN pointwise2
        ds0, ds1 = torch.empty(N, 64, device="cuda").bfloat16(), torch.empty(N, 64, device="cuda").bfloat16()
        check(L.ksr_pw2_bwd(ptr(dz), ptr(w), ptr(a4), ptr(ds0), *dims, stream()), "pw2_bwd")
        check(L.ksr_wgrad(ptr(dz), c_int(64), c_int(0), c_int(64), ptr(a4), c_int(64), c_int(0), c_int(64), None, c_int(0), c_int(0),
                          c_int64(N), ptr(ws), ptr(dW0), ptr(db0), c_int(0), c_int(0), stream()), "wgrad")
        check(L.ksr_pw2_bwd_wg(ptr(dz), ptr(w), ptr(a4), ptr(ds1), ptr(slab), ptr(dW1), ptr(db1), *dims, stream()), "pw2_bwd_wg")
        assert torch.equal(ds0, ds1)
        assert rel_l2(dW1.cpu(), (dz.double().T @ a4.double()).cpu()) <= 1e-5
        assert rel_l2(dW1.cpu(), dW0.cpu()) <= 1e-5 and rel_l2(db1.cpu(), db0.cpu()) <= 1e-5
        # deterministic
        dW2, db2 = torch.empty_like(dW1), torch.empty_like(db1)
        check(L.ksr_pw2_bwd_wg(ptr(dz), ptr(w), ptr(a4), ptr(ds1), ptr(slab), ptr(dW2), ptr(db2), *dims, stream()), "pw2_bwd_wg")
        assert torch.equal(dW1, dW2) and torch.equal(db1, db2)
        # ---- FFN pointwise1 (stencil prologue; the product's blocks split over the four waves)
        dww = torch.randn(64, 9, device="cuda") * 0.3
        y3 = torch.rand(N, 64, device="cuda").bfloat16()
        dt0 = torch.empty(N, 64, device="cuda").bfloat16()
        dp0, dp1 = torch.empty(N, 64, device="cuda").bfloat16(), torch.full((N, 64), float("nan"), device="cuda").bfloat16()
        check(L.ksr_pw1_bwd(ptr(dz), ptr(dww), ptr(w), ptr(y3), ptr(dt0), ptr(dp0), *dims, stream()), "pw1_bwd")
        check(L.ksr_wgrad(ptr(dt0), c_int(64), c_int(0), c_int(64), ptr(y3), c_int(64), c_int(0), c_int(64), None, c_int(0), c_int(0),
                          c_int64(N), ptr(ws), ptr(dW0), ptr(db0), c_int(0), c_int(0), stream()), "wgrad")
        check(L.ksr_pw1_bwd_wg(ptr(dz), ptr(dww), ptr(w), ptr(y3), ptr(dp1), ptr(slab), ptr(dW1), ptr(db1), *dims, stream()), "pw1_bwd_wg")
        assert torch.equal(dp0, dp1)
        # both products see the conv-transposed gradient rounded to bf16 (as stored / as written to the LDS tile)
        assert rel_l2(dW1.cpu(), (dt0.double().T @ y3.double()).cpu()) <= 1e-5
        assert rel_l2(dW1.cpu(), dW0.cpu()) <= 1e-5
        assert rel_l2(db1.cpu(), dt0.double().sum(0).cpu()) <= 1e-5 and rel_l2(db1.cpu(), db0.cpu()) <= 1e-5
        dW2, db2 = torch.empty_like(dW1), torch.empty_like(db1)
        check(L.ksr_pw1_bwd_wg(ptr(dz), ptr(dww), ptr(w), ptr(y3), ptr(dp1), ptr(slab), ptr(dW2), ptr(db2), *dims, stream()), "pw1_bwd_wg")
        assert torch.equal(dW1, dW2) and torch.equal(db1, db2)
        engine.set_precision("fp32")
        rc = L.ksr_pw2_bwd_wg(ptr(dz), ptr(w), ptr(a4), ptr(ds1), ptr(slab), ptr(dW2), ptr(db2), *dims, stream())
        assert rc == -2
    finally:
        engine.set_precision("fp32")


@pytest.mark.parametrize("N", [32, 1000, 128 * 128 * 16 + 5])
@pytest.mark.parametrize("sig", [0, 1])
def test_fused_in_proj_backward_equals_separate_kernels(N, sig):
    """bf16 storage mode: ksr_in_proj_bwd_wg (data gradient + weight / bias gradient of the attention in-projection in one
    pass over dqkv) against ksr_in_proj_bwd + ksr_wgrad on the same bf16 tensors, and against an fp64 evaluation."""
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(N + sig)
    L = lib()
    bf = lambda *s: torch.randn(*s, device="cuda").bfloat16()
    dqkv, resid = bf(N, 192), bf(N, 64)
    xin = torch.rand(N, 64, device="cuda").bfloat16()
    w = torch.randn(192, 64, device="cuda") * 0.1
    engine.set_precision("bf16s")
    try:
        ds0, ds1 = torch.empty(N, 64, device="cuda").bfloat16(), torch.full((N, 64), float("nan"), device="cuda").bfloat16()
        dW0, db0, dW1, db1 = (torch.empty(s, device="cuda") for s in ((192, 64), (192,), (192, 64), (192,)))
        ws = torch.empty(int(L.ksr_wgrad_slab_floats(c_int(192), c_int(64))), device="cuda")
        slab = torch.empty(int(L.ksr_in_proj_bwd_wg_slab_floats()), device="cuda")
        check(L.ksr_in_proj_bwd(ptr(dqkv), ptr(w), ptr(resid), ptr(xin) if sig else None, ptr(ds0), c_int(1), c_int(1), c_int(N),
                                stream()), "in_proj_bwd")
        check(L.ksr_wgrad(ptr(dqkv), c_int(192), c_int(0), c_int(192), ptr(xin), c_int(64), c_int(0), c_int(64), None, c_int(0),
                          c_int(0), c_int64(N), ptr(ws), ptr(dW0), ptr(db0), c_int(0), c_int(0), stream()), "wgrad")
        check(L.ksr_in_proj_bwd_wg(ptr(dqkv), ptr(w), ptr(resid), ptr(xin), c_int(sig), ptr(ds1), ptr(slab), ptr(dW1), ptr(db1),
                                   c_int64(N), stream()), "in_proj_bwd_wg")
        ref = resid.double() + dqkv.double() @ w.bfloat16().double()
        if sig:
            ref = ref * xin.double() * (1 - xin.double())
        assert rel_l2(ds1.float().cpu(), ref.cpu()) <= 4e-3          # bf16 output rounding
        assert rel_l2(ds1.float().cpu(), ds0.float().cpu()) <= 3e-3
        assert rel_l2(dW1.cpu(), (dqkv.double().T @ xin.double()).cpu()) <= 1e-5
        assert rel_l2(dW1.cpu(), dW0.cpu()) <= 1e-5
        assert rel_l2(db1.cpu(), dqkv.double().sum(0).cpu()) <= 1e-5 and rel_l2(db1.cpu(), db0.cpu()) <= 1e-5
        dW2, db2, ds2 = torch.empty_like(dW1), torch.empty_like(db1), torch.empty_like(ds1)
        check(L.ksr_in_proj_bwd_wg(ptr(dqkv), ptr(w), ptr(resid), ptr(xin), c_int(sig), ptr(ds2), ptr(slab), ptr(dW2), ptr(db2),
                                   c_int64(N), stream()), "in_proj_bwd_wg")
        assert torch.equal(dW1, dW2) and torch.equal(db1, db2) and torch.equal(ds1.view(torch.int16), ds2.view(torch.int16))
    finally:
        engine.set_precision("fp32")


@pytest.mark.parametrize("B,H,W", [(2, 9, 11), (1, 37, 70), (3, 8, 64)])
def test_backward_kernels_that_rebuild_t_from_v_equal_the_tensor_reading_ones(B, H, W):
    """The `_v` backward entry points of the first gated block rebuild t = relu(A v + B) [N,128] from the per-pixel scalar v
    instead of reading the tensor.  In the exact fp32 mode, with t materialised from the same (v, A, B), they must agree
    with ksr_gate_pw_bwd / ksr_dwconv_wgrad / ksr_fn1_bwd (stem form) to fp32 round-off."""
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(B * 10 + H)
    L = lib()
    N = B * H * W
    dims = (c_int(B), c_int(H), c_int(W))
    dev = "cuda"
    v = torch.rand(N, device=dev) * (torch.rand(N, device=dev) > 0.2)
    ab = torch.randn(256, device=dev) * 0.5
    t = torch.relu(torch.addcmul(ab[None, 128:], v[:, None], ab[None, :128])).contiguous()      # fma-free reference: A v + B
    t = torch.relu(v[:, None] * ab[None, :128] + ab[None, 128:]).contiguous()
    dz, gate = torch.randn(N, 64, device=dev), torch.randn(N, 64, device=dev)
    w2, gw, w1, p2w = (torch.randn(*s, device=dev) * 0.1 for s in ((64, 64), (64, 9), (128, 64), (64,)))
    engine.set_precision("fp32")
    # gate_pw backward
    dt0, dg0 = torch.zeros(N, 128, device=dev), torch.empty(N, 64, device=dev)
    dt1, dg1 = torch.zeros(N, 128, device=dev), torch.empty(N, 64, device=dev)
    check(L.ksr_gate_pw_bwd(ptr(dz), ptr(w2), ptr(gate), ptr(t), ptr(dt0), ptr(dg0), *dims, stream()), "gate_pw_bwd")
    check(L.ksr_gate_pw_bwd_v(ptr(dz), ptr(w2), ptr(gate), ptr(v), ptr(ab), ptr(dt1), ptr(dg1), None, None, None, *dims, stream()),
          "gate_pw_bwd_v")
    assert rel_l2(dt1[:, :64].cpu(), dt0[:, :64].cpu()) <= 1e-6 and rel_l2(dg1.cpu(), dg0.cpu()) <= 1e-6
    # depthwise weight gradient of the gate conv
    slab = torch.empty(512 * 640, device=dev)
    dW0, db0, dW1, db1 = (torch.empty(s, device=dev) for s in ((64, 9), (64,), (64, 9), (64,)))
    check(L.ksr_dwconv_wgrad(ptr(t), c_int(128), c_int(64), ptr(dg0), c_int(64), c_int(0), ptr(slab), ptr(dW0), ptr(db0), *dims,
                             stream()), "dwconv_wgrad")
    check(L.ksr_dwconv_wgrad_v(ptr(v), ptr(ab), ptr(dg0), c_int(64), c_int(0), ptr(slab), ptr(dW1), ptr(db1), *dims, stream()),
          "dwconv_wgrad_v")
    assert rel_l2(dW1.cpu(), dW0.cpu()) <= 1e-5 and rel_l2(db1.cpu(), db0.cpu()) <= 1e-6
    # fn_1 backward, stem form with column sums
    sums = torch.empty(3 * 512 * 128, device=dev)
    outs = []
    for use_v in (False, True):
        dt = dt0.clone()
        dpre, rc, r2 = torch.empty(N, device=dev), torch.empty(2, 128, device=dev), torch.empty(2, 64, device=dev)
        if use_v:
            check(L.ksr_fn1_bwd_stem_v(ptr(dt), ptr(dg0), ptr(gw), ptr(v), ptr(ab), ptr(w1), ptr(p2w), ptr(dpre), ptr(sums), ptr(rc),
                                       ptr(r2), c_int(64), *dims, stream()), "fn1_bwd_stem_v")
        else:
            check(L.ksr_fn1_bwd(ptr(dt), ptr(dg0), ptr(gw), ptr(t), ptr(w1), None, c_int(1), ptr(v), ptr(p2w), ptr(dpre), ptr(sums),
                                ptr(rc), ptr(r2), c_int(64), *dims, stream()), "fn1_bwd")
        outs.append((dpre.cpu(), rc.cpu(), r2.cpu()))
    for a, b in zip(outs[0], outs[1]):
        assert rel_l2(b, a) <= 1e-5


@pytest.mark.parametrize("B,H,W", [(2, 9, 11), (1, 37, 70), (3, 8, 64), (2, 1, 4), (1, 5, 1), (4, 128, 128)])
def test_stem_scalar_parameter_gradients_vs_autograd(B, H, W):
    """ksr_stem_scalar_wgrad (ffn1.pointwise1 1 -> 1 and ffn1.depthwise 3x3 on one channel, given dpre) against fp64
    autograd of pre = bd + conv3x3(w1 Y + b1); widths that are and are not multiples of 4 (two kernel instances)."""
    import torch.nn.functional as F
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(7 * B + H)
    dev = "cuda"
    x = torch.rand(B, 3, H, W, device=dev)
    dpre = torch.randn(B * H * W, device=dev)
    w1, b1, wd = torch.randn(1, device=dev), torch.randn(1, device=dev), torch.randn(9, device=dev)
    slab, out12 = torch.empty(256 * 12, device=dev), torch.empty(12, device=dev)
    check(lib().ksr_stem_scalar_wgrad(ptr(x), c_int64(3 * H * W), ptr(dpre), ptr(w1), ptr(b1), ptr(wd), ptr(slab), ptr(out12),
                                      c_int(B), c_int(H), c_int(W), stream()), "ksr_stem_scalar_wgrad")
    w1d, b1d, wdd, bdd = (t.double().clone().requires_grad_(True) for t in (w1, b1, wd, torch.zeros(1, device=dev)))
    u = x[:, :1].double() * w1d + b1d
    pre = F.conv2d(u, wdd.view(1, 1, 3, 3), bdd, padding=1)
    pre.backward(dpre.double().view(B, 1, H, W))
    ref = torch.cat([w1d.grad, b1d.grad, wdd.grad, bdd.grad])
    assert (out12.double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-5


@pytest.mark.parametrize("B,H,W", [(2, 9, 11), (1, 37, 70), (3, 8, 64), (5, 64, 96)])
def test_stem_fn1_backward_bf16_storage_vs_fp64(B, H, W):
    """ksr_fn1_bwd_stem_v in bf16 storage (no matrix product: dpre and the sums of dx follow from dt alone; column sums on
    the matrix cores from bf16 tiles) against an fp64 evaluation on the same bf16 inputs.  dpre is exact fp32 arithmetic;
    the sums see the gate half of dt rounded to bf16."""
    import torch.nn.functional as F
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(B * 100 + H)
    L = lib()
    N = B * H * W
    dev = "cuda"
    v = torch.rand(N, device=dev) * (torch.rand(N, device=dev) > 0.2)
    ab = torch.randn(256, device=dev) * 0.5
    dt = torch.zeros(N, 128, device=dev, dtype=torch.bfloat16)
    dt[:, :64] = torch.randn(N, 64, device=dev)
    dg = torch.randn(N, 64, device=dev).bfloat16()
    gw, w1, p2w = (torch.randn(*s, device=dev) * 0.1 for s in ((64, 9), (128, 64), (64,)))
    sums = torch.empty(3 * 512 * 128, device=dev)
    engine.set_precision("bf16s")
    try:
        res = []
        for _ in range(2):
            dpre, rc, r2 = torch.full((N,), 7.0, device=dev), torch.empty(2, 128, device=dev), torch.empty(2, 64, device=dev)
            check(L.ksr_fn1_bwd_stem_v(ptr(dt), ptr(dg), ptr(gw), ptr(v), ptr(ab), ptr(w1), ptr(p2w), ptr(dpre), ptr(sums), ptr(rc),
                                       ptr(r2), c_int(64), c_int(B), c_int(H), c_int(W), stream()), "fn1_bwd_stem_v")
            res.append((dpre, rc, r2))
    finally:
        engine.set_precision("fp32")
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)                                     # fixed summation order
    dpre, rc, r2 = res[0]
    vd = v.double()
    dgn = dg.double().view(B, H, W, 64).permute(0, 3, 1, 2)
    dtg = F.conv_transpose2d(dgn, gw.double().view(64, 1, 3, 3), padding=1, groups=64).permute(0, 2, 3, 1).reshape(N, 64)
    tg = vd[:, None] * ab[None, 64:128].double() + ab[None, 192:].double()
    # the kernel forms t's gate half with one fp32 fma; pixels where that lands within round-off of 0 are left out of dpre
    tg32 = torch.addcmul(ab[None, 192:], v[:, None], ab[None, 64:128])
    dtf = torch.cat([dt[:, :64].double(), dtg * (tg > 0)], 1)
    dx = dtf @ w1.double()
    ref_dpre = (dx @ p2w.double()) * (vd > 0)
    sure = ((tg32.abs() > 1e-6).all(1)).cpu()
    assert sure.float().mean() > 0.99
    assert rel_l2(dpre.cpu()[sure], ref_dpre.cpu()[sure]) <= 2e-5
    ref_rc = torch.stack([(dtf * vd[:, None]).sum(0), dtf.sum(0)])
    ref_r2 = torch.stack([(dx * vd[:, None]).sum(0), dx.sum(0)])
    assert rel_l2(rc.cpu(), ref_rc.cpu()) <= 2e-3
    assert rel_l2(r2.cpu(), ref_r2.cpu()) <= 2e-3


@pytest.mark.parametrize("S,HW,p_drop,save", [(64, 1100, 0.1, True), (64, 130, 0.0, False), (32, 1301, 0.1, True), (12, 2101, 0.1, True),
                                              (48, 64, 0.0, True), (5, 40, 0.1, True), (8, 4100, 0.1, True), (2, 40, 0.1, True), (100, 9, 0.0, True)])
def test_fused_in_projection_attention_equals_the_two_kernels(S, HW, p_drop, save):
    """ksr_batch_qkv_attn_fwd (bf16 storage: in_proj inside the attention kernel, q / k / v formed in LDS) against
    ksr_pointwise_fwd + ksr_batch_attn_fwd on the same input, weights and dropout stream: the same bf16-rounded q, k, v
    up to the summation order of the fp32 accumulation (a few values land on the other side of a rounding boundary),
    the same context.  S = 2 and S = 100 take the entry point's two-kernel route and must agree exactly."""
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(S + HW)
    N = S * HW
    L = lib()
    w = torch.randn(192, 64, device="cuda") * 0.2
    bias = torch.randn(192, device="cuda") * 0.1
    try:
        engine.set_precision("bf16s")
        x = torch.randn(N, 64, device="cuda").bfloat16()
        qkv_a = torch.full((N, 192), float("nan"), device="cuda").bfloat16()
        ctx_a = torch.full((N, 64), float("nan"), device="cuda").bfloat16()
        check(L.ksr_pointwise_fwd(ptr(x), ptr(w), ptr(bias), ptr(qkv_a), c_int(192), c_int(0), c_int(S), c_int(1), c_int(HW),
                                  stream()), "pointwise")
        check(L.ksr_batch_attn_fwd(ptr(qkv_a), ptr(ctx_a), c_int(S), c_int(HW), c_float(p_drop), c_uint64(9), stream()), "attn")
        fused_route = 3 <= S <= 64
        qkv_b = torch.full((N, 192), float("nan"), device="cuda").bfloat16() if (save or not fused_route) else None
        ctx_b = torch.full((N, 64), float("nan"), device="cuda").bfloat16()
        check(L.ksr_batch_qkv_attn_fwd(ptr(x), ptr(w), ptr(bias), ptr(qkv_b), ptr(ctx_b), c_int(S), c_int(HW), c_float(p_drop),
                                       c_uint64(9), stream()), "fused")
    finally:
        engine.set_precision("fp32")
    assert bool(torch.isfinite(ctx_b.float()).all())
    if not fused_route:
        assert torch.equal(ctx_a, ctx_b) and torch.equal(qkv_a, qkv_b)
        return
    if qkv_b is not None:
        d = (qkv_a.float() - qkv_b.float()).abs()
        assert float(d.max()) <= 2 ** -7 * float(qkv_a.float().abs().max())          # at most one bf16 step
        assert float((d > 0).float().mean()) <= 0.02
        assert rel_l2(qkv_b.float().cpu(), qkv_a.float().cpu()) <= 1e-3
    assert rel_l2(ctx_b.float().cpu(), ctx_a.float().cpu()) <= 4e-3


@pytest.mark.parametrize("J,K,ld_dz,c0_dz,ld_a,c0_a", [(64, 64, 64, 0, 64, 0), (32, 64, 32, 0, 64, 0), (64, 32, 64, 0, 32, 0),
                                                      (128, 64, 128, 0, 64, 0), (64, 64, 128, 64, 192, 128), (96, 64, 96, 0, 64, 0)])
@pytest.mark.parametrize("N", [64 * 1024, 70001, 130, 37])
def test_weight_gradient_kernels_bf16_storage_vs_fp64(J, K, ld_dz, c0_dz, ld_a, c0_a, N):
    """ksr_wgrad in bf16 storage (the specialised kernel for the training step's shapes and, for J = 96, the general
    one) against an fp64 product of the same bf16 tensors: dW = dz^T a, db = column sums of dz; channel slices of wider
    tensors, pixel counts that leave ragged chunks and idle workgroups.  The products of bf16 values are exact in fp32;
    only the summation order differs: rel-L2 <= 1e-5."""
    from kan_sr import engine
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(J + K + N)
    L = lib()
    dz = torch.randn(N, ld_dz, device="cuda").bfloat16()
    a = torch.randn(N, ld_a, device="cuda").bfloat16()
    dW = torch.full((J, K), float("nan"), device="cuda")
    db = torch.full((J,), float("nan"), device="cuda")
    slab = torch.empty(int(L.ksr_wgrad_slab_floats(c_int(J), c_int(K))), device="cuda")
    try:
        engine.set_precision("bf16s")
        for _ in range(2):          # twice: the second call must not depend on what the first left in the slab
            check(L.ksr_wgrad(ptr(dz), c_int(ld_dz), c_int(c0_dz), c_int(J), ptr(a), c_int(ld_a), c_int(c0_a), c_int(K), None,
                              c_int(0), c_int(0), c_int64(N), ptr(slab), ptr(dW), ptr(db), c_int(0), c_int(0), stream()), "wgrad")
    finally:
        engine.set_precision("fp32")
    dzs, as_ = dz[:, c0_dz:c0_dz + J].double(), a[:, c0_a:c0_a + K].double()
    assert rel_l2(dW.double().cpu(), (dzs.t() @ as_).cpu()) <= 1e-5
    ref_b = dzs.sum(0).cpu()
    assert float((db.double().cpu() - ref_b).norm()) <= 1e-5 * float(ref_b.norm()) + 1e-4
