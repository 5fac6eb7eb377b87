"""CPU tests: ABI surface, host-side layout logic, and a 2-process gloo rehearsal of the data-parallel step."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

from conftest import ROOT
from oracle import convsepkan_oracle as orc


def test_library_exports_every_declared_symbol():
    """The shared library loads (no GPU needed) and exports each function include/ksr.h declares."""
    from kan_sr import _lib
    hdr = open(os.path.join(ROOT, "include", "ksr.h")).read()
    declared = set(re.findall(r"\b(ksr_[a-z0-9_]+)\s*\(", hdr)) - {"ksr_stream_t"}
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(cdll, name), name
    assert cdll.ksr_version() >= 100


def test_error_channel_without_gpu():
    from kan_sr import _lib
    L = _lib.lib()
    rc = L.ksr_wgrad(None, 64, 0, 48, None, 64, 0, 64, None, 0, 0, ctypes.c_int64(10), None, None, None, 0, 0, None)
    assert rc == -1
    assert b"null pointer" in L.ksr_last_error()
    assert L.ksr_feed_kan_supported(4, 8, 3) == 1 and L.ksr_feed_kan_supported(3, 8, 3) == 0
    assert L.ksr_feed_kan_lane_len(4, 8) == 80 and L.ksr_feed_kan_lane_len(2, 8) == 32 and L.ksr_feed_kan_lane_len(4, 19) == 176


def test_module_tree_and_state_dict_keys():
    from kan_sr.module import ConvSepKan
    m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes == orc.param_shapes(16, 5, 3)
    assert sum(p.numel() for p in m.parameters()) == 88044
    import torch.nn as nn
    kinds = [type(x) for x in m.modules()]
    assert kinds.count(nn.Conv2d) == 13 and nn.MultiheadAttention in kinds and kinds.count(nn.LayerNorm) == 5
    with pytest.raises(Exception):
        m(torch.rand(1, 3, 8, 8))          # CPU tensor: the HIP path is the only path


def test_drop_in_import_name():
    import conv_sep_kan
    from kan_sr.module import ConvSepKan
    assert conv_sep_kan.ConvSepKan is ConvSepKan


def test_flat_layout_adjacency_and_buckets():
    from kan_sr.train import FLAT_ORDER, flat_layout, _TAIL
    shapes = {n[len("layers.0.encoder.encoder."):]: torch.Size(s) for n, s in orc.param_shapes(16, 5, 3).items()}
    table, tail_end, total = flat_layout(shapes)
    assert total == 88044 and sorted(FLAT_ORDER) == sorted(shapes)
    assert tail_end == sum(shapes[n].numel() for n in _TAIL)
    offs = sorted(table.values())
    assert all(offs[i][0] + offs[i][1] == offs[i + 1][0] for i in range(len(offs) - 1))
    for w in FLAT_ORDER:
        if w.endswith("norm1.weight") or w.endswith("norm2.weight") or w.endswith("norm.weight") or w.endswith("pointwise2.weight"):
            b = w[:-6] + "bias"
            assert table[b][0] == table[w][0] + table[w][1]
    names12 = ["ffn1.pointwise1.weight", "ffn1.pointwise1.bias", "ffn1.depthwise.weight", "ffn1.depthwise.bias"]
    o0 = table[names12[0]][0]
    assert [table[n][0] - o0 for n in names12] == [0, 1, 2, 11]


def test_shard_batch_partitions():
    from kan_sr.train import shard_batch
    for gb in (64, 7, 3):
        for ws in (1, 2, 4, 8):
            spans = [shard_batch(gb, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_knots_host_side_bit_exact():
    from kan_sr.ops import make_knots
    for G, k in [(5, 3), (16, 3), (11, 3), (8, 2)]:
        assert make_knots(G, k, [-1, 1]) == [float(v) for v in orc.knot_vector(G, k, [-1, 1])]
    assert abs(make_knots(5, 3, [-1, 1])[6] - 0.20000005) < 1e-8 and make_knots(5, 3, [-1, 1])[6] != 0.2


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["KSR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["KSR_ROOT"], "super-resolution_amd"))
from oracle import convsepkan_oracle as orc
from kan_sr.train import FLAT_ORDER, flat_layout, shard_batch, GlooComm
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
pre = "layers.0.encoder.encoder."
p0 = orc.random_params(4, 5, 3, seed=1)
x = torch.rand(6, 3, 8, 8); tgt = torch.rand(6, 3, 16, 16)
kw = dict(out_channels=[4], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=2)
def shard_grads(lo, hi):
    p = {n: v.clone().requires_grad_() for n, v in p0.items()}
    y = orc.conv_sep_kan_forward(x[lo:hi], p, **kw)
    (y[:, :1] - tgt[lo:hi, :1]).abs().mean().backward()
    return p
table, tail_end, total = flat_layout({n[len(pre):]: v.shape for n, v in p0.items()})
lo, hi = shard_batch(6, world, rank)
p = shard_grads(lo, hi)
flat = torch.zeros(total)
for n in FLAT_ORDER:
    off, k = table[n]; flat[off:off + k] = p[pre + n].grad.flatten()
comm = GlooComm(world)
comm.allreduce_sum(flat[:tail_end]); comm.allreduce_sum(flat[tail_end:])      # the two buckets
flat /= world
# serial reference: mean of the per-shard gradients (attention is over the LOCAL batch, as under DDP)
ref = torch.zeros(total)
for r in range(world):
    a, b = shard_batch(6, world, r)
    q = shard_grads(a, b)
    for n in FLAT_ORDER:
        off, k = table[n]; ref[off:off + k] += q[pre + n].grad.flatten() / world
assert torch.allclose(flat, ref, rtol=1e-5, atol=1e-7), float((flat - ref).abs().max())
# every rank holds the same averaged gradient
g = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(g, flat)
assert all(torch.equal(g[0], gi) for gi in g)
dist.destroy_process_group()
print("OK", rank)
'''


def test_two_rank_gloo_gradient_exchange(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, KSR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2",
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o[-2000:]


DDP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["KSR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["KSR_ROOT"], "super-resolution_amd"))
from torch.nn.parallel import DistributedDataParallel as DDP
from kan_sr.ddp import KsrCommState, ksr_allreduce_hook
from kan_sr.train import GlooComm
from kan_sr._lib import KsrError
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
def net():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Conv2d(1, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1))
torch.manual_seed(100 + rank)                       # every rank its own shard
x, tgt = torch.rand(3, 1, 8, 8), torch.rand(3, 4, 8, 8)
def grads(model):
    model.zero_grad(set_to_none=True)
    (model(x) - tgt).abs().mean().backward()
    return [p.grad.clone() for p in model.parameters()]
plain = net()
local = grads(plain)                                # this rank's own gradients
state = KsrCommState(rank, world, comm=GlooComm(world))
hooked = DDP(net(), bucket_cap_mb=0.0005)           # tiny buckets: several hook calls per backward
hooked.register_comm_hook(state, ksr_allreduce_hook)
got = grads(hooked)
assert state.buckets_reduced >= 1
default = grads(DDP(net()))                         # torch's own all-reduce hook: the mean over ranks
for a, b in zip(got, default):
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-8), float((a - b).abs().max())
# and the mean of the per-rank gradients, gathered by hand
for g_loc, g_got in zip(local, got):
    parts = [torch.zeros_like(g_loc) for _ in range(world)]
    dist.all_gather(parts, g_loc)
    assert torch.allclose(sum(parts) / world, g_got, rtol=1e-6, atol=1e-8)
    assert not torch.equal(parts[0], parts[1])      # the shards really differ
# a second backward through the same hook (buckets rebuilt after the first iteration)
got2 = grads(hooked)
for a, b in zip(got, got2):
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)
assert state.buckets_reduced >= 2, state.buckets_reduced
# a state whose world size disagrees with the process group is refused
try:
    KsrCommState(0, 1, comm=GlooComm(1)); raise SystemExit("world-size mismatch accepted")
except KsrError:
    pass
dist.destroy_process_group()
print("OK", rank)
'''


def test_two_rank_ddp_comm_hook_over_gloo(tmp_path):
    """kan_sr.ddp.ksr_allreduce_hook under real torch DistributedDataParallel with TWO ranks (CPU, gloo, a CPU stand-in for
    the RCCL communicator): the divide-by-world branch and the completed-future contract give the gradients of DDP's own
    all-reduce (= the mean of the per-rank gradients), over two iterations (DDP rebuilds its buckets after the first)."""
    script = tmp_path / "ddp_worker.py"
    script.write_text(DDP_WORKER)
    env = dict(os.environ, KSR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2",
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o[-2000:]


def test_cosine_warm_restarts_schedule_matches_torch():
    """The host-side lr schedule against torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(T_0=500, eta_min=1e-5)
    (src/ml/pipelines/default.py:66-68)."""
    import torch
    from kan_sr.train import cosine_warm_restarts_lr
    from oracle import convsepkan_oracle as orc
    prm = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([prm], lr=1e-3)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=500, T_mult=1, eta_min=1e-5)
    for epoch in range(1203):
        want = opt.param_groups[0]["lr"]
        assert abs(cosine_warm_restarts_lr(1e-3, epoch) - want) <= 1e-12, epoch
        assert abs(orc.cosine_warm_restarts_lr(1e-3, epoch) - want) <= 1e-12, epoch
        opt.step()
        sch.step()


def test_checkpoint_roundtrip_in_reference_layout(tmp_path):
    """save_checkpoint writes what the reference's tools read back (src/cli/transfer.py:57-58: torch.load with
    weights_only=True, ['state_dict'], 'model.' prefix) and an Adam state torch.optim.Adam accepts."""
    from kan_sr.checkpoint import PREFIX, adam_state_dict, load_checkpoint, save_checkpoint
    from kan_sr.module import ConvSepKan
    from kan_sr.train import TrainStep
    torch.manual_seed(5)
    m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4)
    ts = TrainStep(m, lr=3e-4)
    ts.steps = 7
    ts.exp_avg.normal_()
    ts.exp_avg_sq.uniform_(0.0, 1.0)
    path = str(tmp_path / "last.ckpt")
    save_checkpoint(m, path, epoch=12, global_step=340, train_step=ts)
    raw = torch.load(path, weights_only=True)
    assert set(raw["state_dict"]) == {PREFIX + k for k in orc.param_shapes(16, 5, 3)}
    assert raw["epoch"] == 12 and raw["global_step"] == 340
    sched = raw["lr_schedulers"][0]
    assert sched["T_0"] == 500 and sched["last_epoch"] == 12 and abs(sched["_last_lr"][0] - orc.cosine_warm_restarts_lr(3e-4, 12)) < 1e-12
    # torch's own optimiser takes the state, parameter by parameter
    opt = torch.optim.Adam(m.parameters(), lr=1.0)
    opt.load_state_dict(raw["optimizer_states"][0])
    assert opt.param_groups[0]["lr"] == 3e-4
    named = dict(m.named_parameters())
    off = 0
    from kan_sr.train import FLAT_ORDER
    for n in FLAT_ORDER:
        prm = named[ts.pre + n]
        st = opt.state[prm]
        assert float(st["step"]) == 7.0
        assert torch.equal(st["exp_avg"].reshape(-1), ts.exp_avg[off:off + prm.numel()])
        assert torch.equal(st["exp_avg_sq"].reshape(-1), ts.exp_avg_sq[off:off + prm.numel()])
        off += prm.numel()
    # and back into a fresh model / TrainStep
    m2 = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4)
    ts2 = TrainStep(m2, lr=1.0)
    info = load_checkpoint(m2, path, train_step=ts2)
    assert info == {"epoch": 12, "global_step": 340}
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert ts2.steps == 7 and ts2.lr == 3e-4 and torch.equal(ts2.exp_avg, ts.exp_avg) and torch.equal(ts2.exp_avg_sq, ts.exp_avg_sq)
    assert m2.layers[-1].encoder.encoder.feed.weight.data_ptr() >= ts2.flat_p.data_ptr()      # still bound to the flat buffer
    assert adam_state_dict(TrainStep(ConvSepKan([1], [4], [3], 5, 3, 0.1, [-1, 1], 2)))["state"] == {}


def _run_gather_plan_on_host(plan):
    """Executes the rows of a kan_sr.unet._GatherPlan built over CPU tensors exactly as ksr_gather_batch defines them
    (include/ksr.h), reading and writing the tensors' memory through their addresses."""
    import ctypes
    import itertools
    for sp, s2p, dp, shape, ss, s2s, ds, nred, rs, r2s, count in plan.rows:
        f = ctypes.POINTER(ctypes.c_float)
        src, src2, dst = ctypes.cast(sp, f), (ctypes.cast(s2p, f) if s2p else None), ctypes.cast(dp, f)
        n = 0
        for idx in itertools.product(*[range(k) for k in shape]):
            so = sum(i * s for i, s in zip(idx, ss))
            s2o = sum(i * s for i, s in zip(idx, s2s))
            acc = 0.0
            for r in range(nred):
                acc += src[so + r * rs] * (src2[s2o + r * r2s] if src2 is not None else 1.0)
            dst[sum(i * s for i, s in zip(idx, ds))] = acc
            n += 1
        assert n == count


def test_unet_gather_tables_on_host():
    """The two tables of strided gathers the UNet backbone launches per step, interpreted on the CPU: the weight table
    must reproduce the kernels' layouts (torch permutes / flips / the composed attention projection), the gradient
    table the chain rule back to the parameters' shapes."""
    import torch.nn.functional as F
    from kan_sr.unet import UNet
    torch.manual_seed(0)
    m = UNet(1, 40)
    with torch.no_grad():
        for p in m.parameters():
            p.normal_()
    plan, W = m._build_prep_plan()
    _run_gather_plan_on_host(plan)
    dc = m.down2.maxpool_conv[1].double_conv
    w1, w2 = dc[0].weight.detach(), dc[3].weight.detach()
    key = "down2.maxpool_conv.1"
    assert torch.equal(W[key + ".0"], w1.permute(2, 3, 1, 0).reshape(9, 8, 16))
    assert torch.equal(W[key + ".3.T"], w2.permute(2, 3, 1, 0).reshape(9, 16, 16).flip(0).transpose(1, 2))
    assert torch.equal(W[key + ".0.T0"], w1.permute(2, 3, 1, 0).reshape(9, 8, 16).flip(0).transpose(1, 2))
    wu = m.up2.conv.double_conv[0].weight.detach()                              # [16][32][3][3]: skip 16 | upsampled 16
    assert torch.equal(W["up2.conv.0.T1"], wu.permute(2, 3, 1, 0).reshape(9, 32, 16).flip(0).transpose(1, 2)[:, :, 16:])
    assert torch.equal(W["up3.up"], m.up3.up.weight.detach().permute(2, 3, 0, 1).reshape(4, 16, 8))
    # the composed projection z = (q | k | a) equals the three reference convs (unet_blocks.py:100-127, without padding)
    att, c = m.att2, 8
    x = torch.randn(1, c, 3, 3)
    z = torch.einsum("tiz,it->z", W["att2.wz"], x[0].reshape(c, 9)) + W["att2.bz"]
    q = F.conv2d(x, att.q_proj.weight, groups=c).reshape(-1)
    k = F.conv2d(x, att.k_proj.weight).reshape(-1)
    a = att.a_proj[1](F.conv2d(x, att.a_proj[0].weight, groups=c)).reshape(-1)
    assert torch.allclose(z[:2 * c + c // 2], torch.cat([q, k, a]).detach(), atol=1e-5)
    assert torch.equal(W["att2.wzT"], W["att2.wz"].transpose(1, 2))
    # gradient table: fill the kernel-layout buffers, compare with autograd through the same compositions
    B = 2
    eplan, KG, gdst, own = m._build_emit_plan(B)
    for v in KG.values():
        v.normal_()
    _run_gather_plan_on_host(eplan)
    g = KG["att2.wz"]                                                            # [9][c+1][zc]
    wz = torch.zeros(9, c, g.shape[2])
    ps = [att.q_proj.weight, att.k_proj.weight, att.a_proj[0].weight, att.a_proj[1].weight]
    for p in ps:
        p.grad = None
    wz[:, :, :c] = torch.diag_embed(att.q_proj.weight.reshape(c, 9).t())
    wz[:, :, c:2 * c] = att.k_proj.weight.permute(2, 3, 1, 0).reshape(9, c, c)
    wz[:, :, 2 * c:2 * c + c // 2] = att.a_proj[0].weight.reshape(c, 9).t()[:, :, None] * att.a_proj[1].weight.reshape(c // 2, c).t()[None]
    (wz * g[:, :c]).sum().backward()
    for name, p in zip(["q_proj.weight", "k_proj.weight", "a_proj.0.weight", "a_proj.1.weight"], ps):
        assert torch.allclose(gdst["att2." + name], p.grad, atol=1e-5, rtol=1e-5), name
    assert torch.equal(gdst["att2.a_proj.1.bias"], g[0, c, 2 * c:2 * c + c // 2])
    assert torch.allclose(gdst["att2.v_proj.weight"].reshape(-1), KG["att2.dpar"][:, :c * c].sum(0), atol=1e-6)
    assert torch.allclose(gdst["up3.up.bias"], KG["up3.up"][:, 16].sum(0), atol=1e-6)
    assert torch.equal(gdst["up3.up.weight"], KG["up3.up"][:, :16].reshape(2, 2, 16, 8).permute(2, 3, 0, 1))
    assert torch.equal(gdst["reproj.conv.weight"].reshape(40, 4), KG["reproj"][:4].t())
    assert torch.equal(gdst[key + ".double_conv.0.weight"], KG[key + ".0"].reshape(3, 3, 8, 16).permute(3, 2, 0, 1))


@pytest.mark.skipif(not os.path.exists("/root/reference/conv_sep_kan/seemore.py"), reason="reference tree not present")
def test_reference_seemore_cannot_be_constructed():
    """The reference's SeemoRe class (conv_sep_kan/seemore.py:13-77) raises on construction with its defaults and with
    every valid `lr_space`: there is no reference behaviour for that backbone to reproduce."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("seemore_ref", "/root/reference/conv_sep_kan/seemore.py")
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode, old = True, sys.dont_write_bytecode
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = old
    with pytest.raises(NotImplementedError, match="lr_space"):
        mod.SeemoRe()                                              # default lr_space = 1 (seemore.py:24 vs :219-226)
    for lr_space in ("linear", "exp", "double"):
        with pytest.raises(TypeError, match="mlp_ratio"):
            mod.SeemoRe(lr_space=lr_space)                         # RME passes mlp_ratio to GatedFFN (:176 vs :355-358)


def test_attention_head_parameter_buffer_layout():
    """The q | k | v branches' twelve parameters in one buffer (ConvSepKanAttention._qkv with a gradient sink): the four views
    the kernels take equal the per-kind concatenations of the plain path, and piece i of the buffer is parameter i of
    _qkv_params() — the order in which the backward's multi-tensor copy hands the gradients to the sink."""
    from kan_sr import chan_attention as ca
    torch.manual_seed(0)
    head = ca.ConvSepKanAttention(64, 64, 40)
    params = head._qkv_params()
    assert len(params) == 12
    buf = torch.cat([t.detach().reshape(-1) for t in params])
    assert buf.numel() == ca._QKV_CAT[-1] and all(o % 4 == 0 for o in ca._QKV_CAT)
    w, b, dw_w, dw_b = ca._qkv_views(buf)
    assert torch.equal(w, torch.cat([getattr(head, n + "_proj").weight.detach().reshape(64, 64) for n in "qkv"]))
    assert torch.equal(b, torch.cat([getattr(head, n + "_proj").bias.detach() for n in "qkv"]))
    assert torch.equal(dw_w, torch.cat([getattr(head, n + "_dw").weight.detach().reshape(64, 9) for n in "qkv"]))
    assert torch.equal(dw_b, torch.cat([getattr(head, n + "_dw").bias.detach() for n in "qkv"]))
    named = dict(head.named_parameters())
    names = [n + m + "." + kind for m, kind in (("_proj", "weight"), ("_proj", "bias"), ("_dw", "weight"), ("_dw", "bias")) for n in "qkv"]
    for piece, prm, name in zip(ca._qkv_pieces(buf), params, names):
        assert prm is named[name]
        assert torch.equal(piece, prm.detach().reshape(-1)), name
