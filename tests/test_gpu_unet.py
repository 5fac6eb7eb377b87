"""UNet hyper-encoder (forward and backward) on the HIP path against the reference goldens and the CPU oracle."""
import pytest
import torch

from conftest import load_golden, rel_l2, t
from oracle import convsepkan_oracle as orc

pytestmark = pytest.mark.gpu

# fp32 against the reference's fp32: the network is 18 convolutions deep with BatchNorm rescaling in between, so the
# bound is stated on the whole tensor (relative L2) plus a loose element-wise one
REL_L2, ATOL, RTOL = 2e-5, 2e-4, 1e-3


def assert_close(a, ref, what=""):
    a, ref = a.detach().cpu(), ref.detach().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    assert rel_l2(a, ref) <= REL_L2, f"{what}: rel L2 {rel_l2(a, ref):.3e}"
    err = (a - ref).abs()
    assert bool((err <= ATOL + RTOL * ref.abs()).all()), f"{what}: max err {float(err.max()):.3e}"


def _unet(P, sd=None):
    from kan_sr.unet import UNet
    m = UNet(1, P)
    if sd is not None:
        m.load_state_dict({k: v.clone() for k, v in sd.items()})
    return m.cuda()


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("tag,name", [("att8", "att2"), ("att32", "att4")])
def test_anchor_attention_golden(tag, name):
    """SelfAttention alone (dim 8 / 2 heads and dim 32 / 4 heads), including the factor `width` on v."""
    g = load_golden("unet")
    m = _unet(40)
    att = getattr(m, name)
    att.load_state_dict({k[len(tag) + 3:]: t(v) for k, v in g.items() if k.startswith(tag + ".p.")})
    att.cuda()
    x = t(g[tag + ".x"]).cuda()
    B, C, H, W = x.shape
    with torch.no_grad():
        m._ensure_prepared()
        y = m._attention(name, _nhwc(x), B, H, W)
    assert_close(y, _nhwc(t(g[tag + ".y"])), tag)


@pytest.mark.parametrize("tag,P", [("u40", 40), ("u160", 160), ("odd", 40)])
def test_unet_golden_eval_and_train(tag, P):
    """"odd" is the reference's output on a 40x52 image (not a multiple of 16: poolings floor, Up blocks zero-pad)."""
    g = load_golden("unet")
    sd = {k[len(tag) + 4:]: t(v) for k, v in g.items() if k.startswith(tag + ".sd.")}
    m = _unet(P, sd)
    x = t(g[tag + ".x"]).cuda()
    with torch.no_grad():
        m.eval()
        assert_close(m(x), t(g[tag + ".y_eval"]), "eval")
        for k, v in m.state_dict().items():                       # eval leaves the running statistics alone
            if "running" in k or "num_batches" in k:
                assert torch.equal(v.cpu().reshape(-1), sd[k].reshape(-1)), k
        m.train()
        assert_close(m(x), t(g[tag + ".y_train"]), "train")
    after = m.state_dict()
    for k, v in g.items():
        if k.startswith(tag + ".after."):
            n = k[len(tag) + 7:]
            if "num_batches" in n:
                assert int(after[n]) == int(v), n
            else:
                assert rel_l2(after[n].cpu(), t(v)) <= 1e-5, n
    # eval again: the train pass moved the running statistics, so the cached eval-mode scale / shift must not be reused
    p_after = {k: v.detach().cpu().clone() for k, v in after.items()}
    with torch.no_grad():
        m.eval()
        assert_close(m(x), orc.unet(x.cpu(), p_after, "", False), "eval after the train pass")
        assert_close(m(x), orc.unet(x.cpu(), p_after, "", False), "eval, cached affine form")


@pytest.mark.parametrize("B,H,W,P,train", [(1, 16, 16, 40, False), (3, 48, 80, 160, True), (2, 64, 32, 336, True),
                                           (5, 16, 112, 64, False), (2, 256, 256, 160, True), (2, 17, 31, 40, True),
                                           (1, 135, 240, 160, False), (3, 23, 16, 64, True)])
def test_unet_vs_oracle(B, H, W, P, train):
    """Other sizes: a single image with a 1x1 bottleneck, non-square sizes, sizes with partial 256-pixel workgroups,
    attention inputs longer than one Gram slab (256x256), every supported head size, sizes that are not multiples of
    16 (odd at every level: 17x31; a 1080p frame's 1/8 tile: 135x240)."""
    torch.manual_seed(B * 100 + H)
    m = _unet(P)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.2)
                mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
            if hasattr(mod, "temperature_a"):
                mod.temperature_a.normal_(1, 0.5); mod.temperature_v.normal_(1, 0.5)
    p = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = torch.rand(B, 1, H, W)
    stats = {}
    ref = orc.unet(x, p, "", train, stats)
    m.train(train)
    with torch.no_grad():
        y = m(x.cuda())
    assert_close(y, ref, "forward")
    if train:
        after = m.state_dict()
        for k, v in stats.items():
            if "num_batches" in k:
                assert int(after[k]) == int(v)
            else:
                assert rel_l2(after[k].cpu(), v) <= 1e-5, k


def test_model_with_unet_backbone_vs_oracle():
    """ConvSepKan(backbone='unet'): UNet -> KAN cell -> pixel shuffle, chroma bilinear; state_dict keys of the reference's
    UNet under layers.0.encoder.encoder."""
    from kan_sr.module import ConvSepKan
    torch.manual_seed(3)
    m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4, backbone="unet").cuda().eval()
    g = load_golden("unet")
    ref_keys = [k[len("u160.sd."):] for k in g if k.startswith("u160.sd.")]
    assert [k[len("layers.0.encoder.encoder."):] for k in m.state_dict()] == ref_keys
    p = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = torch.rand(2, 3, 32, 48)
    ref = orc.conv_sep_kan_forward(x, p, out_channels=[16], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=4,
                                   backbone="unet")
    with torch.no_grad():
        y = m(x.cuda())
    assert y.shape == (2, 3, 128, 192)
    assert_close(y, ref, "model")


GRAD_REL = 1e-3      # per parameter tensor, relative L2 (fp32 sums over up to 10^5 pixels through BatchNorm's cancellations)


def _check_grads(named_grads, ref_grads, rel=GRAD_REL):
    worst = ("", 0.0)
    for n, gr in named_grads.items():
        assert gr is not None, n
        ref = ref_grads[n]
        if float(ref.norm()) < 1e-6 * max(1.0, ref.numel() ** 0.5):      # structurally ~zero gradients: absolute bound
            assert float((gr.cpu() - ref).abs().max()) <= 1e-5, n
            continue
        e = rel_l2(gr.cpu(), ref)
        worst = max(worst, (n, e), key=lambda t: t[1])
    assert worst[1] <= rel, f"worst gradient: {worst[0]} rel L2 {worst[1]:.3e}"


@pytest.mark.parametrize("tag,name", [("att8", "att2"), ("att32", "att4")])
def test_anchor_attention_golden_gradients(tag, name):
    g = load_golden("unet")
    m = _unet(40)
    att = getattr(m, name)
    att.load_state_dict({k[len(tag) + 3:]: t(v) for k, v in g.items() if k.startswith(tag + ".p.")})
    att.cuda()
    x = t(g[tag + ".x"]).cuda()
    B, C, H, W = x.shape
    with torch.no_grad():
        tape = {}
        m._ensure_prepared()
        m._ensure_emit_plan(B)
        m._attention(name, _nhwc(x), B, H, W, tape=tape)
        m._tape = tape
        g0, g1 = m._attention_bwd(name, _nhwc(t(g[tag + ".gy"]).cuda()))
        grads = m._collect_grads()           # every parameter's slot; only this block's were computed
    assert rel_l2((g0 + g1).cpu(), _nhwc(t(g[tag + ".gx"]))) <= GRAD_REL
    _check_grads({k[len(name) + 1:]: v for k, v in grads.items() if k.startswith(name + ".")},
                 {k[len(tag) + 3:]: t(v) for k, v in g.items() if k.startswith(tag + ".g.")})


def test_unet_golden_train_gradients():
    """Train-mode pass of the reference (batch statistics differentiated through) on the 2x32x32 case."""
    g = load_golden("unet")
    tag = "u40"
    sd = {k[len(tag) + 4:]: t(v) for k, v in g.items() if k.startswith(tag + ".sd.")}
    m = _unet(40, sd).train()
    y = m(t(g[tag + ".x"]).cuda())
    assert_close(y, t(g[tag + ".y_train"]), "train")
    y.backward(t(g[tag + ".gy"]).cuda())
    _check_grads({n: p.grad for n, p in m.named_parameters()},
                 {k[len(tag) + 3:]: t(v) for k, v in g.items() if k.startswith(tag + ".g.")})


@pytest.mark.parametrize("B,H,W,P,train", [(1, 16, 32, 40, True), (3, 48, 80, 160, True), (2, 64, 32, 336, False),
                                           (2, 256, 256, 160, True), (2, 40, 52, 40, True), (3, 37, 67, 64, True),
                                           (2, 135, 120, 160, True), (1, 45, 35, 40, False)])
def test_unet_gradients_vs_oracle(B, H, W, P, train):
    """Against the oracle evaluated in fp64: at 256x256 the fp32 CPU evaluation itself is off by up to 1e-2 on some
    BatchNorm gradients (sums over 10^5 pixels), so fp32-vs-fp32 would measure the checker, not the kernels."""
    torch.manual_seed(B * 10 + H)
    m = _unet(P)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.2)
                mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
            if hasattr(mod, "temperature_a"):
                mod.temperature_a.normal_(1, 0.5); mod.temperature_v.normal_(1, 0.5)
    p = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu().clone()) for k, v in m.state_dict().items()}
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    x, gy = torch.rand(B, 1, H, W), torch.randn(B, P, H, W)
    ref = orc.unet(x.double(), p, "", train)
    (ref * gy.double()).sum().backward()
    m.train(train)
    y = m(x.cuda())
    assert_close(y, ref.float(), "forward")
    y.backward(gy.cuda())
    # BatchNorm over fewer than 16 values at the bottleneck (1x2 pixels here) amplifies fp32 rounding: looser bound there
    rel = 1e-4 if B * (H // 16) * (W // 16) >= 16 else 1e-3
    _check_grads({n: q.grad for n, q in m.named_parameters()}, {n: p[n].grad for n, _ in m.named_parameters()}, rel=rel)


def test_model_with_unet_backbone_trains():
    """ConvSepKan(backbone='unet') in train mode: loss gradients of every parameter against the oracle's autograd."""
    from kan_sr.module import ConvSepKan
    torch.manual_seed(5)
    m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4, backbone="unet").cuda().train()
    p = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for k, v in p.items():
        if v.dtype == torch.float32 and "running" not in k:
            v.requires_grad_()
    x, gy = torch.rand(2, 3, 32, 48), torch.randn(2, 3, 128, 192)
    ref = orc.conv_sep_kan_forward(x, p, out_channels=[16], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=4,
                                   backbone="unet", train=True)
    (ref * gy).sum().backward()
    y = m(x.cuda())
    assert_close(y, ref, "model")
    y.backward(gy.cuda())
    _check_grads({n: q.grad for n, q in m.named_parameters()}, {n: p[n].grad for n, _ in m.named_parameters()})


@pytest.mark.parametrize("train", [False, True])
def test_model_with_unet_backbone_input_gradient(train):
    """Gradient with respect to the input image through ConvSepKan(backbone='unet') (chroma through the upsampling, luma
    through the KAN evaluation and through the UNet's first conv) against the oracle's autograd."""
    from kan_sr.module import ConvSepKan
    torch.manual_seed(6)
    m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4, backbone="unet").cuda().train(train)
    p = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x, gy = torch.rand(2, 3, 32, 48), torch.randn(2, 3, 128, 192)
    x_ref = x.clone().requires_grad_()
    ref = orc.conv_sep_kan_forward(x_ref, p, out_channels=[16], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=4,
                                   backbone="unet", train=train)
    (ref * gy).sum().backward()
    xg = x.cuda().requires_grad_()
    y = m(xg)
    assert_close(y, ref.detach(), "model")
    y.backward(gy.cuda())
    for c in range(3):
        err = float((xg.grad[:, c].cpu() - x_ref.grad[:, c]).norm() / x_ref.grad[:, c].norm())
        assert err <= 5e-4, (c, err)


@pytest.mark.parametrize("loss", ["l1", "l1+ssim"])
def test_unet_train_step_vs_oracle_adam(loss):
    """TrainStep on the UNet backbone (tape backward into the flat buffer, fused Adam): loss value, gradients of the
    first step and the parameters after three steps against oracle + autograd + torch.optim.Adam on the CPU."""
    from kan_sr.module import ConvSepKan
    from kan_sr.train import TrainStep
    torch.manual_seed(21)
    m = ConvSepKan([1], [4], [3], 5, 3, 0.1, [-1, 1], 2, backbone="unet").train()
    p0 = {n: v.detach().clone() for n, v in m.state_dict().items()}
    pr = {n: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in n else v.clone()) for n, v in p0.items()}
    opt = torch.optim.Adam([v for v in pr.values() if v.requires_grad], lr=1e-3)
    ts = TrainStep(m.cuda(), lr=1e-3, loss=loss)
    assert ts.tape_backward and ts.flat_p.numel() == sum(q.numel() for q in m.parameters())
    for it in range(3):
        x, tgt = torch.rand(2, 3, 32, 48), torch.rand(2, 3, 64, 96)
        opt.zero_grad()
        stats = {}
        y = orc.conv_sep_kan_forward(x, pr, out_channels=[4], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=2,
                                     backbone="unet", train=True, stats=stats)
        loss_ref = orc.training_loss(y, tgt) if loss == "l1+ssim" else (y[:, :1] - tgt[:, :1]).abs().mean()
        loss_ref.backward()
        got = ts.step(x.cuda(), tgt.cuda())
        assert abs(float(got) - float(loss_ref.detach())) <= 1e-5 + 1e-4 * abs(float(loss_ref.detach()))
        if it == 0:
            _check_grads({n: ts.arena[n] for n, v in pr.items() if v.requires_grad}, {n: v.grad for n, v in pr.items() if v.requires_grad})
        opt.step()
        with torch.no_grad():
            for k, v in stats.items():
                pr[k] = v.detach().clone()
    for n, prm in m.named_parameters():
        want, have = pr[n].detach() - p0[n], prm.detach().cpu() - p0[n]
        assert float((have - want).abs().max()) <= 2.1e-3, n          # at most ~2/3 of a 3-step swing of lr per step
        assert rel_l2(prm.detach().cpu(), pr[n].detach()) <= 5e-3, n
    for n, v in m.state_dict().items():
        if "running" in n:
            assert rel_l2(v.cpu(), pr[n]) <= 1e-4, n


@pytest.mark.parametrize("B,H,W,r,G,k", [(2, 17, 23, 4, 5, 3), (1, 16, 40, 4, 16, 3), (3, 9, 11, 2, 5, 2), (1, 300, 200, 4, 5, 3),
                                         (2, 384, 320, 4, 5, 3), (1, 33, 35, 4, 8, 3), (2, 20, 20, 4, 5, 1)])
def test_fused_outconv_kan_head_equals_unfused(B, H, W, r, G, k):
    """ksr_kan_lin_fwd / _bwd (the UNet's last 1x1 conv applied inside the KAN-cell kernels) against the same head with
    the parameter tensor materialised (torch 1x1 conv + ksr_kan_cell, itself pinned on the reference goldens): pixel
    counts that are not multiples of the 32-pixel tile, more tiles than persistent workgroups (300x200), more than 256
    parameters per pixel (G = 16), another spline order."""
    from kan_sr.ops import kan_head, kan_lin_head
    torch.manual_seed(B + H)
    P = r * r * (G + k + 2)
    net = _unet(P)
    with torch.no_grad():
        net.reproj.conv.weight.normal_(0, 0.5)
        net.reproj.conv.bias.normal_(0, 0.5)
    x = torch.rand(B, 3, H, W, device="cuda") * 2.4 - 1.2                 # some pixels outside the grid range
    y4 = torch.randn(B * H * W, 4, device="cuda", requires_grad=True)
    gy = torch.randn(B, 3, r * H, r * W, device="cuda")
    lw = net.reproj.conv.weight.detach().view(P, 4).clone().requires_grad_()
    lb = net.reproj.conv.bias.detach().clone().requires_grad_()
    ref = kan_head(x, y4 @ lw.t() + lb, G, k, [-1, 1], r)
    ref.backward(gy)
    want_dy, y4.grad = y4.grad, None
    out = kan_lin_head(x, y4, net, G, k, [-1, 1], r)
    assert float((out - ref).detach().abs().max()) <= 1e-5 + 1e-5 * float(ref.detach().abs().max())
    out.backward(gy)
    assert rel_l2(y4.grad.cpu(), want_dy.cpu()) <= 1e-5
    got = net._KG["reproj"].cpu()                                          # [5][P]: weight^T | bias
    assert rel_l2(got[:4].t(), lw.grad.cpu()) <= 1e-4
    assert rel_l2(got[4], lb.grad.cpu()) <= 1e-4


def test_unet_captured_step_equals_launched_step():
    """TrainStep(capture=True): the replayed HIP graph (steps 3..) gives bit-for-bit the parameters, running statistics
    and losses of the step launched kernel by kernel, with fresh input tensors every step and across a change of the
    input shape (which records a new graph)."""
    from kan_sr.module import ConvSepKan
    from kan_sr.train import TrainStep
    runs = []
    for capture in (False, True):
        torch.manual_seed(33)
        m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4, backbone="unet").cuda().train()
        ts = TrainStep(m, lr=1e-3, loss="l1+ssim", capture=capture)
        gen = torch.Generator().manual_seed(7)
        losses = []
        for it in range(9):
            h, w = (32, 48) if it < 5 else (40, 36)                       # the second shape pads inside the Up blocks
            x, tgt = torch.rand(2, 3, h, w, generator=gen).cuda(), torch.rand(2, 3, 4 * h, 4 * w, generator=gen).cuda()
            losses.append(float(ts.step(x, tgt)))
        if capture:
            assert ts._graph is not None
        runs.append((losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, ts.out.cpu().clone()))
    (l0, s0, o0), (l1, s1, o1) = runs
    assert l0 == l1
    assert torch.equal(o0, o1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_unet_capture_after_an_interleaved_forward_keeps_following_the_weights():
    """A forward of the model between the last warm-up step and the recording step (a validation or preview call)
    refreshes the kernel-layout weight copies early; the recording must still contain the launches that rebuild them,
    or every replay would convolve with the weights of the recording moment."""
    from kan_sr.module import ConvSepKan
    from kan_sr.train import TrainStep
    runs = []
    for capture in (False, True):
        torch.manual_seed(34)
        m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4, backbone="unet").cuda().train()
        ts = TrainStep(m, lr=1e-2, loss="l1", capture=capture)
        gen = torch.Generator().manual_seed(8)
        losses = []
        for it in range(7):
            x, tgt = torch.rand(2, 3, 32, 32, generator=gen).cuda(), torch.rand(2, 3, 128, 128, generator=gen).cuda()
            if it in (2, 4):                       # between warm-up step 2 and the recording step; and between replays
                with torch.no_grad():
                    m(x)
                m.eval()
                with torch.no_grad():
                    m(x)
                m.train()
            losses.append(float(ts.step(x, tgt)))
        if capture:
            assert ts._graph is not None
        runs.append((losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}))
    (l0, s0), (l1, s1) = runs
    assert l0 == l1, (l0, l1)
    for k in s0:
        if "num_batches_tracked" in k or "running" in k:
            continue                               # the interleaved train-mode forwards move these in both runs alike
        assert torch.equal(s0[k], s1[k]), k
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_unet_training_reduces_the_loss():
    from kan_sr.module import ConvSepKan
    from kan_sr.init import init_like_training_setup
    from kan_sr.train import TrainStep
    torch.manual_seed(2)
    m = init_like_training_setup(ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4, backbone="unet")).cuda().train()
    ts = TrainStep(m, lr=2e-3, loss="l1+ssim")
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 128), torch.linspace(0, 1, 128), indexing="ij")
    hr = torch.stack([0.5 + 0.4 * torch.sin(6 * xx + 3 * yy), 0.5 + 0.1 * xx, 0.5 - 0.1 * yy])[None].repeat(4, 1, 1, 1).cuda()
    lr = torch.nn.functional.avg_pool2d(hr, 4)
    losses = [float(ts.step(lr, hr)) for _ in range(60)]
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])


def test_unet_refuses_bad_sizes():
    from kan_sr._lib import KsrError
    m = _unet(40)
    with torch.no_grad(), pytest.raises(KsrError, match="at least 16x16"):
        m(torch.rand(1, 1, 24, 15, device="cuda"))
