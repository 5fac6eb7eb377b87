"""ConvSepKanAttention on the HIP path against the reference goldens and the CPU oracle."""
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


def _module(cout, params):
    from kan_sr.chan_attention import ConvSepKanAttention
    m = ConvSepKanAttention(64, 64, cout)
    m.load_state_dict({k: v.detach().clone() for k, v in params.items()})
    return m.cuda()


@pytest.mark.parametrize("tag,cout", [("o64", 64), ("o160", 160), ("o336", 336)])
def test_golden_forward_and_gradients(tag, cout):
    g = load_golden("chan_attention")
    params = {k[len(tag) + 3:]: t(v) for k, v in g.items() if k.startswith(tag + ".p.")}
    m = _module(cout, params)
    x = t(g[tag + ".x"]).cuda().requires_grad_()
    y = m(x)
    assert y.is_contiguous()
    assert_fwd(y, t(g[tag + ".y"]))
    y.backward(t(g[tag + ".gy"]).cuda())
    assert rel_l2(x.grad.cpu(), t(g[tag + ".gx"])) <= GRAD_REL
    for n, p in m.named_parameters():
        if n.startswith("a_dw."):
            assert p.grad is None
        else:
            assert rel_l2(p.grad.cpu(), t(g[f"{tag}.g.{n}"])) <= GRAD_REL, n


def _oracle_run(params, x, gy, q=None, k=None, v=None):
    p = {n: w.clone().requires_grad_() for n, w in params.items()}
    xs = [None if a is None else a.clone().requires_grad_() for a in (x, q, k, v)]
    y = orc.chan_attention(xs[0], p, "", *xs[1:])
    (y * gy).sum().backward()
    return y.detach(), p, xs


@pytest.mark.parametrize("B,H,W,cout", [(1, 5, 7, 40), (3, 37, 45, 160), (2, 64, 96, 336), (5, 3, 130, 64)])
def test_vs_oracle_ragged_sizes(B, H, W, cout):
    """Sizes that are not multiples of the 4x32 conv tile, the 128-pixel GEMM tile or the 64-pixel Gram chunk."""
    torch.manual_seed(B * 1000 + H)
    from kan_sr.chan_attention import ConvSepKanAttention
    ref = ConvSepKanAttention(64, 64, cout)
    params = {n: w.detach().clone() for n, w in ref.state_dict().items()}
    x, gy = torch.randn(B, 64, H, W), torch.randn(B, cout, H, W)
    y_ref, p_ref, xs = _oracle_run(params, x, gy)
    m = _module(cout, params)
    xg = x.cuda().requires_grad_()
    y = m(xg)
    assert_fwd(y, y_ref)
    y.backward(gy.cuda())
    assert rel_l2(xg.grad.cpu(), xs[0].grad) <= GRAD_REL
    for n, p in m.named_parameters():
        if not n.startswith("a_dw."):
            assert rel_l2(p.grad.cpu(), p_ref[n].grad) <= GRAD_REL, n


def test_precomputed_branches():
    """forward(x, q, k, v) with caller-supplied branches (conv_sep_kan_attention.py:52-57)."""
    torch.manual_seed(5)
    from kan_sr.chan_attention import ConvSepKanAttention
    ref = ConvSepKanAttention(64, 64, 160)
    params = {n: w.detach().clone() for n, w in ref.state_dict().items()}
    B, H, W = 2, 9, 12
    x, gy = torch.randn(B, 64, H, W), torch.randn(B, 160, H, W)
    k, v = torch.randn(B, 64, H, W), torch.randn(B, 64, H, W)
    y_ref, p_ref, xs = _oracle_run(params, x, gy, None, k, v)
    m = _module(160, params)
    xg, kg, vg = (a.cuda().requires_grad_() for a in (x, k, v))
    y = m(xg, k=kg, v=vg)
    assert_fwd(y, y_ref)
    y.backward(gy.cuda())
    assert rel_l2(xg.grad.cpu(), xs[0].grad) <= GRAD_REL
    assert rel_l2(kg.grad.cpu(), xs[2].grad) <= GRAD_REL
    assert rel_l2(vg.grad.cpu(), xs[3].grad) <= GRAD_REL
    assert rel_l2(m.q_proj.weight.grad.cpu(), p_ref["q_proj.weight"].grad) <= GRAD_REL
    # the k and v branches are computed but unused: their parameters get zero gradient
    assert float(m.k_proj.weight.grad.abs().max()) == 0.0


def test_gram_batched_properties():
    """Batched Gram product at a size the oracle does not cover: equals torch.einsum in fp64 and is
    independent of how the pixels are split into workgroups (bitwise reproducible)."""
    from ctypes import c_int, c_int64
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(1)
    B, HW, J = 6, 40000, 336
    dz = torch.randn(B * HW, J, device="cuda")
    a = torch.randn(B * HW, 192, device="cuda")
    L = lib()
    outs = []
    for _ in range(2):
        slab = torch.empty(int(L.ksr_gram_slab_floats(c_int(J), c_int(64), c_int(B), c_int64(HW))), device="cuda")
        out, tail = torch.empty(B, J, 64, device="cuda"), torch.empty(B, J, device="cuda")
        check(L.ksr_gram_batched(ptr(dz), c_int(J), c_int(0), c_int(J), ptr(a), c_int(192), c_int(128), c_int(64), c_int64(HW),
                                 c_int(B), ptr(slab), ptr(out), ptr(tail), c_int(0), stream()), "gram")
        outs.append((out.clone(), tail.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = torch.einsum("bnj,bnk->bjk", dz.view(B, HW, J).double(), a.view(B, HW, 192)[:, :, 128:].double())
    assert rel_l2(outs[0][0].double().cpu(), ref.cpu()) <= 1e-5
    assert rel_l2(outs[0][1].double().cpu(), dz.view(B, HW, J).double().sum(1).cpu()) <= 1e-5


def test_bf16_mode():
    """bf16 MFMA operands in the 1x1 convs (build-side mode).  Stated tolerance: forward rel-L2 <= 2e-2,
    gradients cosine >= 0.999 and rel-L2 <= 5e-2."""
    g = load_golden("chan_attention")
    tag, cout = "o160", 160
    params = {k[len(tag) + 3:]: t(v) for k, v in g.items() if k.startswith(tag + ".p.")}
    m = _module(cout, params)
    m.precision = "bf16"
    x = t(g[tag + ".x"]).cuda().requires_grad_()
    y = m(x)
    assert rel_l2(y.detach().cpu(), t(g[tag + ".y"])) <= 2e-2
    y.backward(t(g[tag + ".gy"]).cuda())
    for n, p in list(m.named_parameters()) + [("x", x)]:
        if n.startswith("a_dw."):
            continue
        ref = t(g[f"{tag}.gx"]) if n == "x" else t(g[f"{tag}.g.{n}"])
        got = p.grad.cpu().flatten().double()
        ref = ref.flatten().double()
        cos = float((got * ref).sum() / (got.norm() * ref.norm()))
        assert cos >= 0.999 and rel_l2(got, ref) <= 5e-2, (n, cos, rel_l2(got, ref))
    from kan_sr import engine
    engine.set_precision("fp32")


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("r,G,B,H,W", [(4, 16, 2, 12, 20), (4, 5, 3, 16, 16), (2, 5, 1, 9, 33), (4, 16, 5, 16, 24)])
def test_model_with_attention_head(r, G, B, H, W, fused, monkeypatch):
    """ConvSepKan(..., attention=True): encoder trunk -> ConvSepKanAttention(64, 64, P) -> KAN -> shuffle, against
    the oracle built from the same reference-pinned pieces (the wiring itself is not in the reference).  fused: the
    head's per-image 1x1 map inside the fused feed+KAN kernel (no [N, P] tensor); otherwise the tensor is materialised
    and the stand-alone KAN cell evaluates it."""
    from kan_sr import module as kmod
    from kan_sr.module import ConvSepKan
    monkeypatch.setattr(kmod, "FUSED_ATTENTION_HEAD", fused)
    torch.manual_seed(40 + G)
    m = ConvSepKan([1], [r * r], [3], G, 3, 0.1, [-1, 1], r, attention=True).eval()
    with torch.no_grad():
        for n, prm in m.named_parameters():          # non-degenerate LayerNorm gains / biases
            if "norm" in n:
                prm.add_(0.1 * torch.randn_like(prm))
    params = {n: v.detach().clone() for n, v in m.state_dict().items()}
    x, gy = torch.rand(B, 3, H, W), torch.randn(B, 3, r * H, r * W)
    p_ref = {n: v.clone().requires_grad_() for n, v in params.items()}
    # both wirings also return the gradient with respect to the input image (chroma through the upsampling, luma through
    # the KAN evaluation and through the trunk)
    x_ref = x.clone().requires_grad_()
    y_ref = orc.conv_sep_kan_forward(x_ref, p_ref, out_channels=[r * r], grid_size=G, spline_order=3, grid_range=[-1, 1],
                                     upscale=r, attention=True)
    (y_ref * gy).sum().backward()
    y_ref = y_ref.detach()
    m = m.cuda()
    with torch.no_grad():
        assert_fwd(m(x.cuda()), y_ref, atol=2e-5)
    xg = x.cuda().requires_grad_()
    y = m(xg)
    assert_fwd(y, y_ref, atol=2e-5)
    y.backward(gy.cuda())
    assert xg.grad.shape == x.shape
    for c in range(3):
        assert rel_l2(xg.grad[:, c].cpu(), x_ref.grad[:, c]) <= 2e-4, (c, rel_l2(xg.grad[:, c].cpu(), x_ref.grad[:, c]))
    for n, prm in m.named_parameters():
        if ".a_dw." in n:
            assert prm.grad is None
            continue
        ref = p_ref[n].grad
        assert rel_l2(prm.grad.cpu(), ref) <= 2e-4 or float((prm.grad.cpu() - ref).norm()) <= 1e-7, (n, rel_l2(prm.grad.cpu(), ref))


@pytest.mark.parametrize("loss,precision", [("l1", "fp32"), ("l1+ssim", "fp32"), ("l1+ssim", "bf16"), ("l1+ssim", "bf16s")])
def test_attention_head_train_step_vs_oracle_adam(loss, precision):
    """TrainStep on ConvSepKan(..., attention=True) (BASELINE configs[3]: G = 16, channel-attention head; tape backward
    into the flat buffer, one exchange bucket, fused Adam): loss value, gradients of the first step and the parameters
    after three steps against oracle + autograd + torch.optim.Adam on the CPU.  Dropout is off (eval mode), as in the
    other oracle comparisons of the step.  bf16 MFMA operands: loss within 2e-2, parameters after three steps within
    the swing of three Adam steps."""
    from kan_sr.module import ConvSepKan
    from kan_sr.train import TrainStep
    from kan_sr import engine
    torch.manual_seed(77)
    r, G = 4, 16
    m = ConvSepKan([1], [r * r], [3], G, 3, 0.1, [-1, 1], r, attention=True).eval()
    with torch.no_grad():
        for n, prm in m.named_parameters():
            if "norm" in n:
                prm.add_(0.1 * torch.randn_like(prm))
    m.precision = precision
    p0 = {n: v.detach().clone() for n, v in m.state_dict().items()}
    pr = {n: v.clone().requires_grad_() for n, v in p0.items()}
    opt = torch.optim.Adam(list(pr.values()), lr=1e-3)
    ts = TrainStep(m.cuda(), lr=1e-3, loss=loss)
    assert ts.tape_backward and ts.flat_p.numel() == sum(q.numel() for q in m.parameters())
    exact = precision == "fp32"
    for it in range(3):
        x, tgt = torch.rand(3, 3, 12, 16), torch.rand(3, 3, 12 * r, 16 * r)
        opt.zero_grad()
        y = orc.conv_sep_kan_forward(x, pr, out_channels=[r * r], grid_size=G, spline_order=3, grid_range=[-1, 1],
                                     upscale=r, attention=True)
        loss_ref = orc.training_loss(y, tgt) if loss == "l1+ssim" else (y[:, :1] - tgt[:, :1]).abs().mean()
        loss_ref.backward()
        got = ts.step(x.cuda(), tgt.cuda())
        lr_ = float(loss_ref.detach())
        assert abs(float(got) - lr_) <= ((1e-5 + 1e-4 * abs(lr_)) if exact else 2e-2 * abs(lr_))
        if it == 0 and exact:
            for n, v in pr.items():
                if ".a_dw." in n:
                    assert float(ts.arena[n].abs().max()) == 0.0
                    continue
                have = ts.arena[n].cpu()
                assert rel_l2(have, v.grad) <= 2e-4 or float((have - v.grad).norm()) <= 1e-7, (n, rel_l2(have, v.grad))
        opt.step()
    for n, prm in m.named_parameters():
        want, have = pr[n].detach() - p0[n], prm.detach().cpu() - p0[n]
        assert float((have - want).abs().max()) <= (2.1e-3 if exact else 6.1e-3), n
        if exact:
            assert rel_l2(prm.detach().cpu(), pr[n].detach()) <= 5e-3, n
    engine.set_precision("fp32")
    ts.close()


def test_attention_head_model_bf16_storage_vs_oracle():
    """ConvSepKan(..., attention=True) with precision "bf16s": the encoder trunk AND the head's fused kernel chain store
    their [N, C] tensors as bf16 (ksr_chan_dw_*, ksr_chan_apply, ksr_feed_kan_*_img, the Gram products and weight gradients
    read them; the per-image 64 x 64 algebra stays fp32).  16 x 32 x 32 pixels, G = 16, eval mode, against the fp32 oracle:
    forward rel-L2 <= 2e-2; gradients of tensors with >= 64 elements cosine >= 0.99 and rel-L2 <= 0.15.
    This is a SEPARATE bound from the plain model's bf16 contract (cosine >= 0.999, rel-L2 <= 5e-2), which this wiring cannot
    meet in any bf16 evaluation: rounding only the head's input features to bf16 inside the oracle already moves 20 of the
    gradients outside it (tests/test_oracle_golden.py::test_attention_head_gradients_are_ill_conditioned_for_bf16_features).
    Measured with tools/grad_parity.py --attention --G 16 --init default --seed 5 (another output gradient than this test's):
    nearly every tensor at cosine 0.998 / rel-L2 0.06 in both bf16 modes; furthest: ffn3.depthwise.bias 0.962 / 0.27,
    ffn3.pointwise1.weight 0.988 / 0.15, ffn2.fn_2.weight 0.991 / 0.14, feed.q_proj.weight 0.996 / 0.09."""
    from kan_sr.module import ConvSepKan
    from kan_sr import engine
    torch.manual_seed(5)
    r, G, B, H, W = 4, 16, 16, 32, 32
    m = ConvSepKan([1], [r * r], [3], G, 3, 0.1, [-1, 1], r, attention=True).eval()
    with torch.no_grad():
        for n, prm in m.named_parameters():
            if "norm" in n:
                prm.add_(0.1 * torch.randn_like(prm))
    params = {n: v.detach().clone() for n, v in m.state_dict().items()}
    x, gy = torch.rand(B, 3, H, W), torch.randn(B, 3, r * H, r * W)
    p_ref = {n: v.clone().requires_grad_() for n, v in params.items()}
    y_ref = orc.conv_sep_kan_forward(x, p_ref, out_channels=[r * r], grid_size=G, spline_order=3, grid_range=[-1, 1],
                                     upscale=r, attention=True)
    (y_ref * gy).sum().backward()
    m = m.cuda()
    m.precision = "bf16s"
    try:
        with torch.no_grad():
            y0 = m(x.cuda())
        assert rel_l2(y0[:, 0].cpu(), y_ref[:, 0].detach()) <= 2e-2
        y = m(x.cuda())
        y.backward(gy.cuda())
    finally:
        engine.set_precision("fp32")
    assert rel_l2(y[:, 0].detach().cpu(), y_ref[:, 0].detach()) <= 2e-2
    for n, prm in m.named_parameters():
        if ".a_dw." in n:
            continue
        ref, got = p_ref[n].grad.flatten(), prm.grad.cpu().flatten()
        assert bool(torch.isfinite(got).all()), n
        if ref.numel() >= 64:
            cos = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
            assert cos >= 0.99, (n, cos)
            assert rel_l2(got, ref) <= 0.15, (n, rel_l2(got, ref))


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (3, 9, 21), (1, 5, 70)])
def test_fused_dw_bwd_and_projection_gradient_is_bit_equal_to_the_two_kernels(B, H, W):
    """bf16 storage: ksr_chan_dw_bwd_dx (transposed depthwise conv of q | k | v with the projections' data gradient in its
    epilogue) against ksr_chan_dw_bwd followed by ksr_pointwise_bwd on the same inputs: both outputs bit-equal (the same
    products in the same order), on sizes with ragged tiles; and the pair itself is held to the oracle by the head's
    gradient tests above."""
    from ctypes import c_int
    from kan_sr import engine
    from kan_sr._lib import check, lib, ptr, stream
    torch.manual_seed(B * 100 + W)
    N = B * H * W
    dqkv = torch.randn(N, 192, device="cuda").to(torch.bfloat16)
    dw_w = torch.randn(192, 9, device="cuda") * 0.3
    w = torch.randn(192, 64, device="cuda") * 0.1
    L = lib()
    engine.set_precision("bf16s")
    try:
        dpq_a = torch.empty(N, 192, device="cuda", dtype=torch.bfloat16)
        dx_a = torch.empty(N, 64, device="cuda", dtype=torch.bfloat16)
        check(L.ksr_chan_dw_bwd(ptr(dqkv), ptr(dw_w), ptr(dpq_a), c_int(192), c_int(3), c_int(B), c_int(H), c_int(W), stream()),
              "ksr_chan_dw_bwd")
        check(L.ksr_pointwise_bwd(ptr(dpq_a), c_int(192), ptr(w), ptr(dx_a), c_int(B), c_int(H), c_int(W), stream()),
              "ksr_pointwise_bwd")
        dpq_b = torch.full((N, 192), float("nan"), device="cuda", dtype=torch.bfloat16)
        dx_b = torch.full((N, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
        check(L.ksr_chan_dw_bwd_dx(ptr(dqkv), ptr(dw_w), ptr(w), ptr(dpq_b), ptr(dx_b), c_int(B), c_int(H), c_int(W), stream()),
              "ksr_chan_dw_bwd_dx")
        torch.cuda.synchronize()
    finally:
        engine.set_precision("fp32")
    assert torch.equal(dpq_a, dpq_b)
    assert torch.equal(dx_a, dx_b)
    # and against a plain evaluation of the projection gradient from the stored (bf16) depthwise gradient
    ref = dpq_a.float() @ w.to(torch.bfloat16).float()
    assert rel_l2(dx_b.float().cpu(), ref.cpu()) <= 1e-2
    # the fp32 mode refuses
    rc = L.ksr_chan_dw_bwd_dx(ptr(dqkv), ptr(dw_w), ptr(w), ptr(dpq_b), ptr(dx_b), c_int(B), c_int(H), c_int(W), stream())
    assert rc != 0
