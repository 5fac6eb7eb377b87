"""GPU tests of the native training step (loss, in-place gradients, fused Adam) and of train-mode dropout."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2, t
from oracle import convsepkan_oracle as orc

pytestmark = pytest.mark.gpu


def _model(r=4, G=5, seed=3):
    from kan_sr.module import ConvSepKan
    p0 = orc.random_params(r * r, G, 3, seed=seed)
    m = ConvSepKan([1], [r * r], [3], G, 3, 0.1, [-1, 1], r)
    m.load_state_dict(p0)
    return m, p0


def test_train_step_matches_torch_adam_on_oracle():
    """Three steps of TrainStep (eval-mode graph: dropout off) against oracle forward + L1(Y) + autograd +
    torch.optim.Adam on the CPU."""
    from kan_sr.train import TrainStep
    torch.manual_seed(0)
    r, B, H, W = 4, 3, 12, 40
    m, p0 = _model(r)
    m = m.cuda().eval()
    ts = TrainStep(m, lr=1e-3)
    pr = {n: v.clone().requires_grad_() for n, v in p0.items()}
    opt = torch.optim.Adam(list(pr.values()), lr=1e-3)
    for it in range(3):
        x = torch.rand(B, 3, H, W)
        tgt = torch.rand(B, 3, r * H, r * W)
        opt.zero_grad()
        y = orc.conv_sep_kan_forward(x, pr, out_channels=[r * r], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=r)
        loss_ref = (y[:, :1] - tgt[:, :1]).abs().mean()
        loss_ref.backward()
        loss = ts.step(x.cuda(), tgt.cuda())
        assert abs(float(loss) - float(loss_ref.detach())) <= 1e-5 + 1e-4 * abs(float(loss_ref.detach()))
        if it == 0:
            for n, v in pr.items():
                got = ts.arena[n].cpu()
                assert rel_l2(got, v.grad) <= 2e-4, f"grad {n}: {rel_l2(got, v.grad):.3e}"
        opt.step()
    # Adam normalises each gradient element (sign-like for the first steps): near-zero gradient elements
    # make the update ill-conditioned, so compare on the update direction in aggregate
    for n, v in pr.items():
        got = dict(m.named_parameters())[n].detach().cpu()
        d_ref, d_got = v.detach() - p0[n], got - p0[n]
        assert float((d_got - d_ref).abs().max()) <= 2.1e-3, n          # at most ~2/3 of a 3-step swing
        assert rel_l2(got, v.detach()) <= 5e-3, f"param {n}: {rel_l2(got, v.detach()):.3e}"


def test_parameters_live_in_flat_buffer_and_module_forward_sees_updates():
    from kan_sr.train import TrainStep
    m, _ = _model(2)
    m = m.cuda().eval()
    ts = TrainStep(m)
    x = torch.rand(2, 3, 8, 32).cuda()
    with torch.no_grad():
        y0 = m(x).clone()
    before = ts.flat_p.clone()
    ts.step(x, torch.rand(2, 3, 16, 64).cuda())
    assert not torch.equal(before, ts.flat_p)
    for n, prm in m.named_parameters():
        assert prm.data_ptr() >= ts.flat_p.data_ptr() and prm.data_ptr() < ts.flat_p.data_ptr() + 4 * ts.flat_p.numel()
    with torch.no_grad():
        y1 = m(x)
    assert not torch.equal(y0[:, 0], y1[:, 0])
    assert torch.equal(y0[:, 1:], y1[:, 1:])       # chroma has no parameters


def test_train_mode_dropout_statistics_match_reference():
    """Dropout is live in train mode; bit parity needs the reference's RNG stream, so parity is statistical:
    mean / variance of the output over 64 seeds against the reference's own 64-seed statistics."""
    from kan_sr.module import ConvSepKan
    g = load_golden("dropout_stats")
    m = ConvSepKan([1], [4], [3], 5, 3, 0.1, [-1, 1], 2)
    m.load_state_dict({n[2:]: t(v) for n, v in g.items() if n.startswith("p.")})
    m = m.cuda()
    x = t(g["x"]).cuda()
    m.eval()
    with torch.no_grad():
        ye = m(x).cpu()
    err = (ye - t(g["y_eval"])).abs()
    assert bool((err <= 2e-5 + 1e-4 * t(g["y_eval"]).abs()).all())
    m.train()
    ys = []
    torch.manual_seed(123)
    with torch.no_grad():
        for _ in range(64):
            ys.append(m(x).cpu())
    ys = torch.stack(ys)
    assert float(ys[:, :, 0].var(0).mean()) > 0            # dropout really is on
    mean, var = ys.mean(0)[:, 0], ys.var(0)[:, 0]
    rmean, rvar = t(g["y_train_mean"])[:, 0], t(g["y_train_var"])[:, 0]
    # difference of two 64-sample means: sigma = sqrt((var + rvar)/64)
    z = (mean - rmean).abs() / ((var + rvar) / 64).sqrt().clamp_min(1e-6)
    assert float((z < 4.0).float().mean()) > 0.99, float((z < 4.0).float().mean())
    ratio = float(var.mean() / rvar.mean())
    assert 0.75 < ratio < 1.33, ratio


def test_dropout_backward_uses_same_masks():
    """With dropout on, a finite-difference directional derivative through the SAME masks (same seed)
    matches the analytic gradient: the backward regenerates the forward's masks."""
    from kan_sr import engine
    from kan_sr.train import TrainStep
    m, p0 = _model(2, seed=9)
    m = m.cuda().train()
    ts = TrainStep(m, lr=0.0)
    x = torch.rand(2, 3, 8, 32).cuda()
    tgt = torch.rand(2, 3, 16, 64).cuda()
    seeds = iter([77, 77, 77])
    orig = engine.new_seed
    engine.new_seed = lambda: next(seeds)
    try:
        l0 = float(ts.step(x, tgt))
        gflat = ts.flat_g.clone()
        d = torch.randn_like(gflat)
        d = d / d.norm()
        eps = 2e-3
        ts.flat_p.add_(eps * d)
        lp = float(ts.step(x, tgt))
        ts.flat_p.add_(-2 * eps * d)
        lm = float(ts.step(x, tgt))
    finally:
        engine.new_seed = orig
    fd = (lp - lm) / (2 * eps)
    an = float((gflat * d).sum())
    assert abs(fd - an) <= 0.08 * max(abs(an), abs(fd)) + 2e-4, (fd, an)


def test_rccl_path_single_rank_matches_no_comm_path():
    """The RCCL wrapper + two-bucket side-stream schedule with a 1-rank communicator must give exactly the
    parameters of the communicator-free step (sum over one rank, scale 1)."""
    import torch.distributed as dist
    from kan_sr.train import RcclComm, TrainStep
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(4)
        x = torch.rand(2, 3, 8, 32).cuda()
        tgt = torch.rand(2, 3, 32, 128).cuda()
        ma, _ = _model(4, seed=5)
        mb, _ = _model(4, seed=5)
        ta = TrainStep(ma.cuda().eval())
        comm = RcclComm(0, 1)
        tb = TrainStep(mb.cuda().eval(), world_size=1, rank=0, comm=comm)
        assert tb.side is not None
        for _ in range(3):
            la, lb = ta.step(x, tgt), tb.step(x, tgt)
        torch.cuda.synchronize()
        assert float(la) == float(lb)
        assert torch.equal(ta.flat_p, tb.flat_p)
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B,H,W", [(2, 40, 52), (1, 33, 97), (3, 12, 11), (1, 270, 400), (2, 135, 203), (1, 133, 193)])
def test_ssim_kernels_vs_oracle(B, H, W):
    """SSIM value and its luma gradient against autograd of the oracle's restatement (parity with torchmetrics itself
    is unpinned: the library is not available).  Values outside [0,1] exercise the clamp."""
    from ctypes import c_float, c_int, c_int64
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(H)
    pred = (torch.rand(B, 3, H, W) * 1.2 - 0.1).requires_grad_()
    tgt = torch.rand(B, 3, H, W) * 1.1 - 0.05
    val = orc.ssim(pred, tgt)
    val.backward()
    L = lib()
    p, t_ = pred.detach().cuda(), tgt.cuda()
    slab = torch.empty(int(L.ksr_ssim_slab_floats(c_int(3), c_int(B), c_int(H), c_int(W))), device="cuda")
    maps = torch.empty(3 * B * (H - 10) * (W - 10), device="cuda")
    out = torch.zeros(1, device="cuda")
    plane = H * W
    check(L.ksr_ssim_fwd(ptr(p), c_int64(3 * plane), ptr(t_), c_int64(3 * plane), c_int(3), c_int(B), c_int(H), c_int(W),
                         ptr(maps), ptr(slab), ptr(out), stream()), "ssim_fwd")
    assert abs(float(out) - float(val.detach())) <= 2e-6
    g = torch.zeros(B, plane, device="cuda")
    check(L.ksr_ssim_y_grad(ptr(maps), ptr(p), c_int64(3 * plane), ptr(t_), c_int64(3 * plane), ptr(g), c_int64(plane), c_int(B),
                            c_int(H), c_int(W), c_float(1.0 / (B * 3 * (H - 10) * (W - 10))), stream()), "ssim_grad")
    ref = pred.grad[:, 0].reshape(B, plane)
    assert rel_l2(g.cpu(), ref) <= 1e-4
    # both loss terms in one pass: L1(Y) + w (1 - SSIM), gradient written (the buffer's old contents must not matter)
    wgt = 0.15
    pred2 = pred.detach().clone().requires_grad_()
    (torch.nn.functional.l1_loss(pred2[:, :1], tgt[:, :1]) + wgt * (1.0 - orc.ssim(pred2, tgt))).backward()
    g2 = torch.full((B, plane), 3.0, device="cuda")
    l1, total = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    slab2 = torch.empty(int(L.ksr_ssim_l1_slab_floats(c_int(B), c_int(H), c_int(W))), device="cuda")
    check(L.ksr_ssim_y_grad_l1(ptr(maps), ptr(p), c_int64(3 * plane), ptr(t_), c_int64(3 * plane), ptr(g2), c_int64(plane), c_int(B),
                               c_int(H), c_int(W), c_float(-wgt / (B * 3 * (H - 10) * (W - 10))), c_float(1.0), ptr(slab2), ptr(l1),
                               ptr(out), c_float(wgt), ptr(total), stream()), "ssim_y_grad_l1")
    assert abs(float(l1) - float(torch.nn.functional.l1_loss(pred[:, :1], tgt[:, :1]))) <= 1e-6
    assert torch.equal(total, l1 + (1.0 - out) * wgt)            # the sum two one-element tensors would give
    assert rel_l2(g2.cpu(), pred2.grad[:, 0].reshape(B, plane)) <= 1e-5


def test_ssim_forward_full_size_is_the_mean_over_half_batches():
    """The SSIM forward at the size of the default training bench (64 x 3 planes of 512 x 512: one task per strip and plane,
    walking the whole height) against the same kernel on the two halves of the batch, which it splits into two bands
    per strip: the value is the mean of the halves' values and the derivative maps are the halves' maps."""
    from ctypes import c_int, c_int64
    from kan_sr._lib import lib, ptr, stream, check
    torch.manual_seed(7)
    B, H, W = 64, 512, 512
    p = (torch.rand(B, 3, H, W, device="cuda") * 1.2 - 0.1)
    t_ = p + 0.1 * torch.randn(B, 3, H, W, device="cuda")
    L = lib()
    plane, win = H * W, (H - 10) * (W - 10)

    def run(pp, tt, b):
        slab = torch.empty(int(L.ksr_ssim_slab_floats(c_int(3), c_int(b), c_int(H), c_int(W))), device="cuda")
        maps = torch.empty(3, b, win, device="cuda")
        out = torch.zeros(1, device="cuda")
        check(L.ksr_ssim_fwd(ptr(pp), c_int64(3 * plane), ptr(tt), c_int64(3 * plane), c_int(3), c_int(b), c_int(H), c_int(W),
                             ptr(maps), ptr(slab), ptr(out), stream()), "ssim_fwd")
        return float(out), maps

    v, maps = run(p, t_, B)
    v0, m0 = run(p[:32].contiguous(), t_[:32].contiguous(), 32)
    v1, m1 = run(p[32:].contiguous(), t_[32:].contiguous(), 32)
    assert 0.0 < v < 1.0 and abs(v - 0.5 * (v0 + v1)) <= 1e-6
    assert torch.equal(maps[:, :32], m0) and torch.equal(maps[:, 32:], m1)


def test_train_step_with_ssim_loss_vs_oracle():
    """TrainStep(loss='l1+ssim') = the reference's training_step loss: loss value and one Adam step against autograd of
    the oracle (eval mode, so no dropout)."""
    from kan_sr.module import ConvSepKan
    from kan_sr.train import TrainStep
    torch.manual_seed(11)
    m = ConvSepKan([1], [4], [3], 5, 3, 0.1, [-1, 1], 2).eval()
    params = {n: v.detach().clone() for n, v in m.state_dict().items()}
    x, tgt = torch.rand(2, 3, 16, 20), torch.rand(2, 3, 32, 40)
    p_ref = {n: v.clone().requires_grad_() for n, v in params.items()}
    y = orc.conv_sep_kan_forward(x, p_ref, out_channels=[4], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=2)
    loss = orc.training_loss(y, tgt)
    loss.backward()
    opt = torch.optim.Adam(list(p_ref.values()), lr=1e-3)
    before = {n: v.detach().clone() for n, v in p_ref.items()}
    opt.step()
    ts = TrainStep(m.cuda(), loss="l1+ssim")
    got = ts.step(x.cuda(), tgt.cuda())
    assert abs(float(got) - float(loss)) <= 1e-5
    for n, prm in m.named_parameters():
        want = p_ref[n].detach() - before[n]
        have = prm.detach().cpu() - before[n]
        # Adam's first step moves every element by lr * g / (|g| + 1e-8): compare the update where the gradient is well
        # above eps (there the step is ~lr * sign(g)); where |g| is within a few hundred eps (the q / k rows of the
        # in-projection: sums that cancel to ~1e-8) the step follows the last bits of g, so only its bound is checked
        clear = p_ref[n].grad.abs() > 1e-6
        assert (have - want)[clear].abs().max() <= 2e-4 * 1e-3 + 0.02 * want.abs().max() if bool(clear.any()) else True, n
        assert (have - want).abs().max() <= 1.001e-3, n


def test_training_trajectory_bf16_storage_tracks_fp32():
    """25 optimisation steps (reference loss, eval-mode graph so that both runs see identical noise-free steps) on a
    fixed batch: the loss goes down, and the bf16-storage run stays within 3 % of the exact fp32 run."""
    from kan_sr.module import ConvSepKan
    from kan_sr.init import init_like_training_setup
    from kan_sr.train import TrainStep
    torch.manual_seed(21)
    x = torch.rand(4, 3, 32, 32, device="cuda")
    tgt = torch.nn.functional.interpolate(x, scale_factor=2, mode="bicubic").clamp(0, 1)
    hist = {}
    for prec in ("fp32", "bf16s"):
        torch.manual_seed(22)                # same default-initialised in_proj / LayerNorm parameters in both runs
        m = ConvSepKan([1], [4], [3], 5, 3, 0.1, [-1, 1], 2).eval()
        init_like_training_setup(m, 0)
        m = m.cuda()
        m.precision = prec
        ts = TrainStep(m, lr=2e-4, loss="l1+ssim")
        hist[prec] = [float(ts.step(x, tgt)) for _ in range(25)]
    assert hist["fp32"][-1] < 0.9 * hist["fp32"][0], hist["fp32"]
    assert hist["bf16s"][-1] < 0.9 * hist["bf16s"][0], hist["bf16s"]
    for a, b in zip(hist["fp32"], hist["bf16s"]):
        assert abs(a - b) <= 0.03 * abs(a), (hist["fp32"], hist["bf16s"])


def test_checkpoint_resume_reproduces_the_next_step(tmp_path):
    """Three steps, save, load into a fresh model + TrainStep, one more step on both with the same dropout stream:
    parameters and loss agree bit for bit (every kernel on the step is deterministic)."""
    from kan_sr.checkpoint import load_checkpoint, save_checkpoint
    from kan_sr.train import TrainStep
    r, B, H, W = 2, 4, 16, 24
    torch.manual_seed(11)
    m, _ = _model(r)
    m = m.cuda().train()
    ts = TrainStep(m, lr=1e-3, loss="l1+ssim")
    xs = [torch.rand(B, 3, H, W).cuda() for _ in range(4)]
    tg = [torch.rand(B, 3, r * H, r * W).cuda() for _ in range(4)]
    for it in range(3):
        ts.step(xs[it], tg[it])
    path = str(tmp_path / "last.ckpt")
    save_checkpoint(m, path, epoch=3, global_step=3, train_step=ts)
    m2, _ = _model(r, seed=99)
    m2 = m2.cuda().train()
    ts2 = TrainStep(m2, lr=5.0, loss="l1+ssim")
    assert load_checkpoint(m2, path, train_step=ts2) == {"epoch": 3, "global_step": 3}
    torch.manual_seed(77)
    l1 = float(ts.step(xs[3], tg[3]))
    torch.manual_seed(77)
    l2 = float(ts2.step(xs[3], tg[3]))
    assert l1 == l2 and ts2.steps == 4
    assert torch.equal(ts.flat_p, ts2.flat_p) and torch.equal(ts.exp_avg, ts2.exp_avg)
    ts.close(); ts2.close()


def test_training_learns_smooth_images_in_bf16_storage_mode():
    """200 steps of the reference loss on smooth synthetic images (bicubic-enlarged noise, LR = bicubic x1/4): the loss
    must fall by more than 5x and stay finite in the bf16 storage mode (dropout live)."""
    import torch.nn.functional as F
    from kan_sr.init import init_like_training_setup
    from kan_sr.module import ConvSepKan
    from kan_sr.train import TrainStep
    torch.manual_seed(0)
    m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4)
    init_like_training_setup(m, 0)
    m = m.cuda().train()
    m.precision = "bf16s"
    ts = TrainStep(m, lr=1e-3, loss="l1+ssim")
    g = torch.Generator().manual_seed(1)
    first = last = None
    for it in range(200):
        hr = F.interpolate(torch.rand(8, 3, 16, 16, generator=g), scale_factor=8, mode="bicubic", align_corners=False).clamp(0, 1)
        lr_img = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
        loss = ts.step(lr_img.cuda(), hr.cuda())
        if it == 0:
            first = float(loss)
    last = float(loss)
    assert last == last and last < first / 5, (first, last)
    assert bool(torch.isfinite(ts.flat_p).all())
    ts.close()


@pytest.mark.parametrize("B", [8, 12])
def test_gazetr_captured_step_equals_launched_step(B):
    """TrainStep(capture=True) on the GazeTR encoder: the replayed HIP graph (forward + loss + backward + Adam; steps 3..)
    gives bit for bit the losses, outputs and parameters of the same step launched kernel by kernel, with dropout
    live (masks from the device-resident step counter), fresh input tensors every step, a change of the input shape and
    a return to the first shape (both graphs stay recorded).  B = 8: fused transformer kernel; B = 12: QKV GEMM + MFMA
    attention + post kernel."""
    from kan_sr.train import TrainStep
    runs = []
    for capture in (False, True):
        torch.manual_seed(41)
        m, _ = _model(4, seed=6)
        m = m.cuda().train()
        m.precision = "bf16s"
        ts = TrainStep(m, lr=1e-3, loss="l1+ssim", capture=capture, device_state=True)
        gen = torch.Generator().manual_seed(7)
        losses, outs = [], []
        for it in range(12):
            h, w = (16, 24) if (it < 5 or it >= 9) else (20, 12)
            x, tgt = torch.rand(B, 3, h, w, generator=gen).cuda(), torch.rand(B, 3, 4 * h, 4 * w, generator=gen).cuda()
            losses.append(float(ts.step(x, tgt)))
            outs.append(ts.out[:, 0].clone())
        if capture:
            assert len([e for e in ts._graphs.values() if e["graph"] is not None]) == 2
        assert ts.steps == 12 and int(ts._ctr) == 12
        runs.append((losses, outs, ts.flat_p.clone(), ts.exp_avg_sq.clone()))
        ts.close()
    (l0, o0, p0, v0), (l1, o1, p1, v1) = runs
    assert l0 == l1, (l0, l1)
    assert len(set(l0)) == len(l0)                     # the steps differ (new masks, new data)
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    assert torch.equal(p0, p1) and torch.equal(v0, v1)


def test_device_resident_adam_matches_host_argument_adam():
    """ksr_adam_step_dev (step count and lr read from device memory, bias corrections evaluated on the device) against
    ksr_adam_step (by-value arguments) over 6 steps in eval mode, with a learning-rate change in between."""
    from kan_sr.train import TrainStep
    res = []
    for dev_state in (False, True):
        torch.manual_seed(5)
        m, _ = _model(2, seed=8)
        ts = TrainStep(m.cuda().eval(), lr=1e-3, device_state=dev_state)
        gen = torch.Generator().manual_seed(3)
        for it in range(6):
            if it == 3:
                ts.set_epoch(250)
            ts.step(torch.rand(2, 3, 8, 32, generator=gen).cuda(), torch.rand(2, 3, 16, 64, generator=gen).cuda())
        res.append((ts.flat_p.clone(), ts.exp_avg.clone()))
    assert torch.equal(res[0][1], res[1][1])
    assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-7


def test_dropout_masks_change_with_the_device_step_counter():
    """With device_state the dropout stream is seed_base mixed with the step counter: two steps on identical data and
    parameters (lr 0) draw different masks; resetting the counter reproduces the first step's loss bit for bit."""
    from kan_sr.train import TrainStep
    torch.manual_seed(9)
    m, _ = _model(2, seed=4)
    ts = TrainStep(m.cuda().train(), lr=0.0, device_state=True)
    x, tgt = torch.rand(4, 3, 8, 32).cuda(), torch.rand(4, 3, 16, 64).cuda()
    a = float(ts.step(x, tgt))
    b = float(ts.step(x, tgt))
    ts.steps = 0
    c = float(ts.step(x, tgt))
    assert a != b and a == c


@pytest.mark.parametrize("precision,side_wgrad", [("fp32", False), ("bf16s", False), ("bf16s", True)])
def test_gradient_exchange_schedule_order(precision, side_wgrad, monkeypatch):
    """The two-bucket exchange of the GazeTR step, with a recording stand-in for the communicator and the library
    calls logged in launch order: bucket 1 (feed + transformer layers + final norm) is issued after the last launch
    that writes into it (layer 0's in-projection weight gradient) and before the first launch of the conv-trunk
    backward (ffn4's gate_pw backward); bucket 2 after the last backward launch; Adam after both.  The buckets are the
    two halves [0, tail_end) and [tail_end, total) of the flat gradient buffer.  Run in the exact mode and in bf16 storage
    (the benchmarked mode, whose backward uses other entry points: weight gradients accumulated inside the data-gradient
    kernels), there also with the parameter-gradient reductions on the second stream."""
    from kan_sr import _lib, autograd
    from kan_sr.train import TrainStep
    log = []

    class Recorder:
        world_size = 2                                   # the step then scales the summed gradients by 1/2

        def allreduce_sum(self, t, hip_stream):
            log.append(("allreduce", t.data_ptr(), t.numel(), int(hip_stream.value or 0)))

        def close(self):
            pass

    monkeypatch.setattr(autograd, "SIDE_STREAM_FOR_WGRAD", side_wgrad)
    monkeypatch.setattr(autograd, "SIDE_STREAM_MIN_TOKENS", 0)
    monkeypatch.setattr(autograd, "DEFER_REDUCTIONS", "1")
    m, _ = _model(4, seed=2)
    m = m.cuda().eval()
    m.precision = precision
    ts = TrainStep(m, world_size=2, rank=0, comm=Recorder(), loss="l1+ssim")
    proxy = _lib.lib()
    orig_getattr = type(proxy).__getattr__

    def logging_getattr(self, name):
        fn = orig_getattr(self, name)
        if not name.startswith("ksr_") or name in ("ksr_last_error",):
            return fn

        def wrapped(*a):
            log.append((name, torch.cuda.current_stream().cuda_stream))
            return fn(*a)
        return wrapped

    type(proxy).__getattr__ = logging_getattr
    try:
        x, tgt = torch.rand(4, 3, 16, 32).cuda(), torch.rand(4, 3, 64, 128).cuda()
        ts.step(x, tgt)
        torch.cuda.synchronize()
    finally:
        type(proxy).__getattr__ = orig_getattr
        from kan_sr import engine
        engine.set_precision("fp32")
    names = [e[0] for e in log]
    ar = [i for i, n in enumerate(names) if n == "allreduce"]
    assert len(ar) == 2
    b1, b2 = log[ar[0]], log[ar[1]]
    assert b1[1] == ts.flat_g.data_ptr() and b1[2] == ts.tail_end
    assert b2[1] == ts.flat_g.data_ptr() + 4 * ts.tail_end and b2[2] == ts.flat_g.numel() - ts.tail_end
    main = torch.cuda.current_stream().cuda_stream
    assert b1[3] == b2[3] == ts.side.cuda_stream != main
    # launches that write bucket-1 gradients: the feed + KAN backward and its weight gradient, every transformer-layer
    # backward kernel, the in-projection backward (bf16 storage: data and weight gradient in one kernel)
    in_proj = "ksr_in_proj_bwd_wg" if precision == "bf16s" else "ksr_in_proj_bwd"
    in_proj_bwd = [i for i, n in enumerate(names) if n == in_proj]
    assert len(in_proj_bwd) == 2
    tail_only = ("ksr_feed_kan_bwd", "ksr_feed_kan_bwd_lnf", "ksr_batch_tx_layer_bwd_token", "ksr_batch_tx_layer_bwd_mlp_wg", "ksr_batch_tx_layer_bwd_ln1_wg",
                 "ksr_batch_tx_layer_bwd_out_wg", "ksr_batch_attn_bwd", "ksr_batch_tx_layer_bwd_attn", "ksr_wgrad_affine_input", in_proj)
    tail_launches = [i for i, n in enumerate(names) if n in tail_only]
    if precision == "bf16s":
        # every transformer weight gradient is accumulated by a kernel that holds its operands: no separate token kernel;
        # the final norm's backward runs inside the feed + KAN backward
        assert names.count("ksr_feed_kan_bwd_lnf") == 1 and "ksr_feed_kan_bwd" not in names
        assert names.count("ksr_batch_tx_layer_bwd_mlp_wg") == 2 and names.count("ksr_batch_tx_layer_bwd_ln1_wg") == 2
        assert names.count("ksr_batch_tx_layer_bwd_out_wg") == 2 and "ksr_in_proj_bwd" not in names
        assert "ksr_batch_tx_layer_bwd_token" not in names
    else:
        assert names.count("ksr_batch_tx_layer_bwd_token") == 2
    last_tail_wgrad = max(i for i, n in enumerate(names) if n == "ksr_wgrad" and i < in_proj_bwd[1])
    trunk_first = "ksr_gate_pw_bwd_wg" if precision == "bf16s" else "ksr_gate_pw_bwd"
    first_trunk = min(i for i, n in enumerate(names) if n == trunk_first)
    assert last_tail_wgrad < ar[0] and max(tail_launches) < ar[0]     # after every launch that writes a bucket-1 gradient
    assert ar[0] < first_trunk, (ar[0], first_trunk)
    backward_launches = [i for i, n in enumerate(names) if n.endswith(("_bwd", "_wgrad", "_wg", "outer2", "_bwd_stem_v",
                                                                       "_wgrad_v", "_bwd_v"))]
    assert ar[1] > max(backward_launches)
    # the slab reductions are recorded and run in one launch per bucket: each exchange follows the flush that completes
    # its gradients (and the affine map of linear1's gradient, which continues from a reduced gradient)
    flushes = [i for i, n in enumerate(names) if n == "ksr_reduce_flush"]
    assert len(flushes) == 2, flushes
    assert max(i for i in tail_launches if names[i] != "ksr_wgrad_affine_input") < flushes[0] < ar[0]
    assert all(flushes[0] < i < ar[0] for i, n in enumerate(names) if n == "ksr_wgrad_affine_input")
    assert max(backward_launches) < flushes[1] < ar[1]
    used_side = {e[1] for e in log if e[0] != "allreduce"} - {main}
    assert bool(used_side) == side_wgrad, "parameter-gradient reductions on the second stream: not as requested"
    adam = names.index("ksr_adam_step")
    assert adam > ar[1] and adam == len(names) - 1
    assert log[adam][1] == main
    # the 1/world scaling is applied by Adam: a second, exchange-free step on the same data with doubled gradients
    # is not needed here; the gloo test in test_host_logic.py covers the arithmetic
    ts.close()


def test_ddp_comm_hook_through_rccl_single_rank():
    """torch DistributedDataParallel (what Lightning's DDP strategy wraps) with ksr_allreduce_hook: every gradient bucket
    goes through libksr's RCCL communicator; with one rank the result must equal the plain module's gradients bit for
    bit, and an optimizer step on the wrapped module trains the same parameters."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from kan_sr.ddp import KsrCommState, ksr_allreduce_hook
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29547", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    state = None
    try:
        torch.manual_seed(3)
        x = torch.rand(2, 3, 8, 32).cuda()
        gy = torch.randn(2, 3, 32, 128).cuda()
        ma, _ = _model(4, seed=5)
        mb, _ = _model(4, seed=5)
        ma, mb = ma.cuda().eval(), mb.cuda().eval()
        ma(x).backward(gy)
        ddp = DDP(mb, device_ids=[0])
        state = KsrCommState(0, 1)
        ddp.register_comm_hook(state, ksr_allreduce_hook)
        ddp(x).backward(gy)
        torch.cuda.synchronize()
        assert state.buckets_reduced >= 1
        for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
            assert torch.equal(pa.grad, pb.grad), n
    finally:
        if state is not None:
            state.close()
        dist.destroy_process_group()


def test_lightning_strategy_reports_missing_package():
    from kan_sr._lib import KsrError
    from kan_sr.ddp import lightning_strategy
    try:
        import lightning  # noqa: F401
    except ImportError:
        with pytest.raises(KsrError, match="lightning"):
            lightning_strategy()


def test_deferred_slab_reductions_are_bit_equal(monkeypatch):
    """The slab reductions of a backward pass recorded and run in one launch per bucket, or in groups of three per stream
    (ksr_reduce_defer / ksr_reduce_flush), give the bits of the immediate form: whole step in bf16 storage and in the exact mode, flat gradient buffers compared."""
    from kan_sr import autograd, engine
    from kan_sr.train import TrainStep
    for precision in ("bf16s", "fp32"):
        flats = []
        for mode in ("0", "1", "3", "0"):
            monkeypatch.setattr(autograd, "DEFER_REDUCTIONS", mode)
            m, _ = _model(4, seed=6)
            m = m.cuda().eval()
            m.precision = precision
            ts = TrainStep(m, lr=0.0, loss="l1+ssim")
            torch.manual_seed(11)
            x, tgt = torch.rand(5, 3, 24, 40).cuda(), torch.rand(5, 3, 96, 160).cuda()
            loss = ts.step(x, tgt)
            torch.cuda.synchronize()
            flats.append((float(loss), ts.flat_g.clone()))
            ts.close()
            assert int(_lib_pending()) == 0
        engine.set_precision("fp32")
        assert flats[0][0] == flats[1][0] == flats[2][0] == flats[3][0]
        assert all(torch.equal(flats[0][1], f[1]) for f in flats[1:]), precision
        assert float(flats[0][1].abs().max()) > 0


def _lib_pending():
    from kan_sr._lib import lib
    return lib().ksr_reduce_pending()
