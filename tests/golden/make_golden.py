"""Generates the golden fixtures in this directory by IMPORTING THE REFERENCE.

Run once, in the build container (the reference checkout is not available on the
GPU machine):   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Only data is written (inputs, weights, expected outputs / gradients as .npz).
The one shim is `torch.Any = typing.Any`: the reference annotates three return
types with torch.Any, which current torch no longer exposes.
"""
import os
import sys
import typing

import numpy as np
import torch

REF = os.environ.get("KSR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
torch.Any = typing.Any  # noqa

from conv_sep_kan.bspline import compute_bspline                      # noqa: E402
from conv_sep_kan.kan_layer import KANLayer, generate_control_points   # noqa: E402
from conv_sep_kan.conv_sep_kan_cells import ConvSepKanCell             # noqa: E402
from conv_sep_kan.unet_blocks import FFN, GatedFFN                     # noqa: E402
from conv_sep_kan.gazetr import GazeTR                                 # noqa: E402
from conv_sep_kan import ConvSepKan                                    # noqa: E402
from conv_sep_kan.conv_sep_kan_attention import ConvSepKanAttention    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items() if v.ndim} if len(out) < 12 else len(out))


def perturb_norms(m, gen):
    """Default LayerNorm params (1, 0) hide scale/shift bugs: randomise them."""
    for mod in m.modules():
        if isinstance(mod, torch.nn.LayerNorm):
            with torch.no_grad():
                mod.weight.copy_(1 + 0.1 * torch.randn(mod.weight.shape, generator=gen))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=gen))


# ---------------------------------------------------------------- (1) basis functions
def gold_bspline():
    cases = [(5, 3), (16, 3), (11, 3), (5, 1), (8, 2)]
    gen = torch.Generator().manual_seed(11)
    out = {}
    for G, k in cases:
        grid = generate_control_points(-1.0, 1.0, 1, 1, k, G)          # [1,1,M]
        t = grid[0, 0]
        ulp_up = torch.nextafter(t, torch.full_like(t, 10.0))
        ulp_dn = torch.nextafter(t, torch.full_like(t, -10.0))
        special = torch.tensor([-1.0, 1.0, 0.0, 0.2, 0.6, 1.25, 1.5, 2.2, -2.2, 2.3, -2.3])
        xs = torch.cat([t, ulp_up, ulp_dn, special,
                        torch.rand(256, generator=gen),
                        torch.rand(256, generator=gen) * 5.0 - 2.5])
        b = compute_bspline(xs[:, None], grid, k)                      # [N,1,1,K]
        out[f"G{G}_k{k}_knots"] = t
        out[f"G{G}_k{k}_x"] = xs
        out[f"G{G}_k{k}_basis"] = b[:, 0, 0, :]
    # the reference's own (broken) __main__ parameter set, bspline.py:45-52
    G, k, I, O = 11, 3, 5, 7
    grid = generate_control_points(-1.0, 1.0, I, O, k, G)
    x = torch.ones(2, I) / 0.8
    out["main_x"] = x
    out["main_basis"] = compute_bspline(x, grid, k)                    # [2,O,I,K]
    save("bspline", **out)


# ---------------------------------------------------------------- (2) KAN layer + grads
def gold_kan_layer():
    gen = torch.Generator().manual_seed(22)
    out = {}
    for (O, I, G, k) in [(4, 1, 5, 3), (16, 1, 5, 3), (7, 5, 5, 3), (16, 1, 16, 3)]:
        N, K = 64, G + k
        layer = KANLayer(I, O, G, k, 0.1, [-1, 1])
        x = (torch.rand(N, I, generator=gen) * 2.6 - 1.3).requires_grad_()
        coef = torch.randn(N, O, I, K, generator=gen).requires_grad_()
        uw = torch.randn(N, O, I, generator=gen).requires_grad_()
        rw = torch.randn(N, O, I, generator=gen).requires_grad_()
        y = layer(x, coef, uw, rw)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        tag = f"O{O}_I{I}_G{G}_k{k}_"
        out.update({tag + "x": x, tag + "coef": coef, tag + "uw": uw, tag + "rw": rw,
                    tag + "y": y, tag + "gy": gy, tag + "gx": x.grad, tag + "gcoef": coef.grad,
                    tag + "guw": uw.grad, tag + "grw": rw.grad})
    save("kan_layer", **out)


# ---------------------------------------------------------------- (3) cell + shuffle
def gold_cell():
    gen = torch.Generator().manual_seed(33)
    out = {}
    for (r, G, k, I) in [(2, 5, 3, 1), (4, 5, 3, 1), (4, 16, 3, 1), (2, 5, 3, 3)]:
        O = r * r
        cell = ConvSepKanCell(I, O, G, k, 0.1, [-1, 1])
        B, H, W = 2, 8, 8
        x = (torch.rand(B, I, H, W, generator=gen) * 2.4 - 1.2).requires_grad_()
        w = torch.randn(B, int(cell.size), H, W, generator=gen).requires_grad_()
        y = torch.nn.PixelShuffle(r)(cell(x, w))
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        tag = f"r{r}_G{G}_k{k}_I{I}_"
        out.update({tag + "x": x, tag + "w": w, tag + "y": y, tag + "gy": gy,
                    tag + "gx": x.grad, tag + "gw": w.grad, tag + "size": int(cell.size),
                    tag + "offsets": np.asarray(cell._kan_params_indices)})
    save("kan_cell", **out)


# ---------------------------------------------------------------- (4) FFN / GatedFFN
def gold_ffn():
    gen = torch.Generator().manual_seed(44)
    torch.manual_seed(44)
    out = {}
    for name, mod, cin in [("ffn_1_64", FFN(1, 64), 1), ("ffn_64_64", FFN(64, 64), 64),
                           ("gated", GatedFFN(64, 64, 3, torch.nn.ReLU()), 64)]:
        x = torch.randn(2, cin, 9, 11, generator=gen).requires_grad_()
        y = mod(x)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        out.update({name + ".x": x, name + ".y": y, name + ".gy": gy, name + ".gx": x.grad})
        for n, p in mod.named_parameters():
            out[f"{name}.p.{n}"] = p
            out[f"{name}.g.{n}"] = p.grad
    save("ffn", **out)


# ---------------------------------------------------------------- (5) GazeTR eval
def gold_gazetr():
    torch.manual_seed(55)
    gen = torch.Generator().manual_seed(55)
    P = 40
    m = GazeTR(1, P).eval()
    perturb_norms(m, gen)
    out = {f"p.{n}": p for n, p in m.named_parameters()}
    for B in (1, 2, 4):
        m.zero_grad()
        x = torch.rand(B, 1, 8, 8, generator=gen).requires_grad_()
        y = m(x)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        out.update({f"B{B}.x": x, f"B{B}.y": y, f"B{B}.gy": gy, f"B{B}.gx": x.grad})
        for n, p in m.named_parameters():
            out[f"B{B}.g.{n}"] = p.grad.clone()
    save("gazetr", **out)


# ---------------------------------------------------------------- (6)+(8) whole model
def gold_model():
    for (r, G, k) in [(2, 5, 3), (4, 5, 3), (4, 16, 3)]:
        torch.manual_seed(600 + 10 * r + G)
        gen = torch.Generator().manual_seed(600 + 10 * r + G)
        m = ConvSepKan([1], [r * r], [3], G, k, 0.1, [-1, 1], r).eval()
        perturb_norms(m, gen)
        B, H, W = 2, 16, 16
        x = torch.rand(B, 3, H, W, generator=gen)
        y = m(x)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        out = {"x": x, "y": y, "gy": gy}
        for n, p in m.named_parameters():
            out["p." + n] = p
            out["g." + n] = p.grad
        # fp64 run of the same graph to apportion fp32 rounding error
        m64 = ConvSepKan([1], [r * r], [3], G, k, 0.1, [-1, 1], r).double().eval()
        m64.load_state_dict({n: v.double() for n, v in m.state_dict().items()})
        # the knot tensor is a plain attribute, not a buffer: convert it by hand
        act = m64.layers[0].kan_cell.kan_layer.activation_fn
        act.grid = generate_control_points(-1.0, 1.0, 1, r * r, k, G).double()
        y64 = m64(x.double())
        (y64 * gy.double()).sum().backward()
        out["y64"] = y64
        for n, p in m64.named_parameters():
            out["g64." + n] = p.grad
        save(f"model_r{r}_G{G}_k{k}", **out)


# ---------------------------------------------------------------- (9) cfg-1 shaped smoke
def gold_cfg1():
    torch.manual_seed(9)
    gen = torch.Generator().manual_seed(9)
    m = ConvSepKan([1], [4], [3], 5, 3, 0.1, [-1, 1], 2).eval()
    perturb_norms(m, gen)
    x = torch.rand(4, 3, 64, 64, generator=gen)
    y = m(x)
    out = {"x_seed": 9, "y_sum": y.double().sum(), "y_abs_sum": y.double().abs().sum(),
           "y_corner": y[:, :, :4, :4], "y_row": y[1, 0, 77, :]}
    for n, p in m.named_parameters():
        out["p." + n] = p
    out["x"] = x
    save("cfg1_smoke", **out)


# ---------------------------------------------------------------- train-mode statistics
def gold_dropout_stats():
    """Train-mode (dropout p=0.1 live) output mean/var over 64 RNG seeds."""
    torch.manual_seed(77)
    gen = torch.Generator().manual_seed(77)
    m = ConvSepKan([1], [4], [3], 5, 3, 0.1, [-1, 1], 2)
    perturb_norms(m, gen)
    x = torch.rand(4, 3, 8, 8, generator=gen)
    m.eval()
    y_eval = m(x)
    m.train()
    ys = []
    for s in range(64):
        torch.manual_seed(1000 + s)
        ys.append(m(x).detach())
    ys = torch.stack(ys)
    out = {"x": x, "y_eval": y_eval, "y_train_mean": ys.mean(0), "y_train_var": ys.var(0)}
    for n, p in m.named_parameters():
        out["p." + n] = p
    save("dropout_stats", **out)


# ---------------------------------------------------------------- (7) channel attention
def gold_chan_attention():
    out = {}
    for tag, cout, (B, H, W) in [("o64", 64, (2, 8, 8)), ("o160", 160, (2, 8, 8)), ("o336", 336, (3, 9, 11))]:
        torch.manual_seed(70 + cout)
        gen = torch.Generator().manual_seed(70 + cout)
        m = ConvSepKanAttention(64, 64, cout)
        x = torch.randn(B, 64, H, W, generator=gen).requires_grad_()
        y = m(x)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        out.update({f"{tag}.x": x, f"{tag}.y": y, f"{tag}.gy": gy, f"{tag}.gx": x.grad})
        for n, p in m.named_parameters():
            out[f"{tag}.p.{n}"] = p
            if p.grad is not None:
                out[f"{tag}.g.{n}"] = p.grad
    save("chan_attention", **out)


# ---------------------------------------------------------------- colour conversion
def gold_color():
    """src/ml/utils/colors/ycbcr.py, loaded by file path (the package __init__ pulls in modules that need
    libraries which are not installed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_ycbcr", os.path.join(REF, "src/ml/utils/colors/ycbcr.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gen = torch.Generator().manual_seed(123)
    rgb = torch.rand(2, 3, 5, 7, generator=gen)
    rgb[0, :, 0, 0] = torch.tensor([0., 0., 0.])
    rgb[0, :, 0, 1] = torch.tensor([1., 1., 1.])
    # The reference's rgb_to_ycbcr raises on every image with more than one pixel (a .view() on a permuted tensor,
    # ycbcr.py:38), so only its stated formula `rgb @ T + O` with the module's own T and O is evaluated here;
    # ycbcr_to_rgb (which uses reshape) runs as is.
    try:
        mod.rgb_to_ycbcr(rgb)
        raise SystemExit("reference rgb_to_ycbcr works now: regenerate this fixture from it")
    except RuntimeError:
        pass
    ycc = (rgb.permute(0, 2, 3, 1) @ mod.T + mod.O).permute(0, 3, 1, 2).contiguous()
    back = mod.ycbcr_to_rgb(ycc)
    ycc2 = torch.rand(3, 4, 6, generator=gen)
    save("color", rgb=rgb, ycbcr=ycc, rgb_back=back, ycc2=ycc2, rgb2=mod.ycbcr_to_rgb(ycc2), T=mod.T, O=mod.O)


# ---------------------------------------------------------------- UNet hyper-encoder
def gold_unet():
    """conv_sep_kan/unet.py UNet(1, P) and its SelfAttention block: eval-mode forward (running statistics),
    train-mode forward (batch statistics + the running-statistics update) and the gradients of the train-mode pass."""
    from conv_sep_kan.unet import UNet
    from conv_sep_kan.unet_blocks import SelfAttention
    out = {}
    gen = torch.Generator().manual_seed(140)
    torch.manual_seed(140)
    for tag, dim, heads, (B, H, W) in [("att8", 8, 2, (2, 12, 20)), ("att32", 32, 4, (3, 4, 6))]:
        att = SelfAttention(dim=dim, num_heads=heads)
        with torch.no_grad():
            att.temperature_a.copy_(1 + 0.5 * torch.randn(att.temperature_a.shape, generator=gen))
            att.temperature_v.copy_(1 + 0.5 * torch.randn(att.temperature_v.shape, generator=gen))
        x = torch.randn(B, dim, H, W, generator=gen).requires_grad_()
        y = att(x)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        out.update({f"{tag}.x": x, f"{tag}.y": y, f"{tag}.gy": gy, f"{tag}.gx": x.grad})
        for n, p in att.named_parameters():
            out[f"{tag}.p.{n}"] = p
            out[f"{tag}.g.{n}"] = p.grad
    # "odd": 40 / 8 = 5 and 52 / 4 = 13 are odd, so two poolings floor and two Up blocks pad
    for tag, P, (B, H, W) in [("u40", 40, (2, 32, 32)), ("u160", 160, (2, 16, 32)), ("odd", 40, (2, 40, 52))]:
        m = UNet(1, P)
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.weight.copy_(1 + 0.2 * torch.randn(mod.weight.shape, generator=gen))
                    mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=gen))
                    mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=gen))
                    mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=gen))
                if isinstance(mod, SelfAttention):
                    mod.temperature_a.copy_(1 + 0.5 * torch.randn(mod.temperature_a.shape, generator=gen))
                    mod.temperature_v.copy_(1 + 0.5 * torch.randn(mod.temperature_v.shape, generator=gen))
        for n, v in m.state_dict().items():
            out[f"{tag}.sd.{n}"] = v.clone()
        x = torch.rand(B, 1, H, W, generator=gen)
        out[f"{tag}.x"] = x
        m.eval()
        with torch.no_grad():
            out[f"{tag}.y_eval"] = m(x)
        m.train()
        y = m(x)
        out[f"{tag}.y_train"] = y
        if P == 40 and tag != "odd":         # gradients for the small case only (fixture size)
            gy = torch.randn(y.shape, generator=gen)
            (y * gy).sum().backward()
            out[f"{tag}.gy"] = gy
            for n, p in m.named_parameters():
                out[f"{tag}.g.{n}"] = p.grad
        for n, v in m.state_dict().items():
            if "running" in n or "num_batches" in n:
                out[f"{tag}.after.{n}"] = v.clone()
    save("unet", **out)


# ---------------------------------------------------------------- whole model: gradient with respect to the input image
def gold_model_input_grad():
    """The model fixtures' setup (same seeds, hence the same weights, x and gy as model_r*_G*_k*.npz) with x requiring a
    gradient: what autograd returns for the image (all three planes)."""
    out = {}
    for (r, G, k) in [(2, 5, 3), (4, 5, 3), (4, 16, 3)]:
        torch.manual_seed(600 + 10 * r + G)
        gen = torch.Generator().manual_seed(600 + 10 * r + G)
        m = ConvSepKan([1], [r * r], [3], G, k, 0.1, [-1, 1], r).eval()
        perturb_norms(m, gen)
        x = torch.rand(2, 3, 16, 16, generator=gen).requires_grad_()
        y = m(x)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        ref = np.load(os.path.join(HERE, f"model_r{r}_G{G}_k{k}.npz"))
        assert np.array_equal(ref["x"], x.detach().numpy()) and np.array_equal(ref["gy"], gy.numpy())
        out[f"r{r}_G{G}.gx"] = x.grad
        m64 = ConvSepKan([1], [r * r], [3], G, k, 0.1, [-1, 1], r).double().eval()
        m64.load_state_dict({n: v.double() for n, v in m.state_dict().items()})
        m64.layers[0].kan_cell.kan_layer.activation_fn.grid = generate_control_points(-1.0, 1.0, 1, r * r, k, G).double()
        x64 = x.detach().double().requires_grad_()
        (m64(x64) * gy.double()).sum().backward()
        out[f"r{r}_G{G}.gx64"] = x64.grad
    save("model_input_grad", **out)


def gold_seemore():
    """Blocks of conv_sep_kan/seemore.py that can be constructed (the top class SeemoRe and RME / ResGroup raise in their
    constructors): StripedConv2d, LayerNorm(channels_first), GatedFFN(GELU), StripedConvFormer, SME, Expert, Router,
    MoELayer (train mode, batch 2; eval mode, batch 1: the reference's eval branch squeezes the batch axis) and MoEBlock
    for the three lr_space rules (train mode, batch 2, 32 x 20: the second aggregation step floors) + eval mode (second image) +
    use_shuffle (32 x 16, one aggregation step, 4 experts, top 1).
    Every case: the module's state_dict, the input(s) and the forward output; no gradients (forward kernels only)."""
    from conv_sep_kan import seemore as sm
    gen = torch.Generator().manual_seed(777)
    out = {}

    def randomise(m):
        with torch.no_grad():
            for p_ in m.parameters():
                p_.copy_(p_ + 0.05 * torch.randn(p_.shape, generator=gen))

    def put(tag, m, y):
        for n, v in m.state_dict().items():
            out[f"{tag}.p.{n}"] = v
        out[f"{tag}.y"] = y

    C = 64
    x = torch.randn(2, C, 16, 20, generator=gen)
    k = torch.randn(2, C, 16, 20, generator=gen)
    xb = torch.randn(2, C, 32, 20, generator=gen)       # inputs are stored once: x, k (small blocks), xb (MoEBlock)
    out["x"], out["k"], out["xb"] = x, k, xb
    for ks in (3, 11):
        m = sm.StripedConv2d(C, kernel_size=ks, depthwise=True).eval(); randomise(m)
        put(f"striped{ks}", m, m(x))
    m = sm.LayerNorm(C, data_format="channels_first").eval(); randomise(m)
    put("ln", m, m(x))
    m = sm.GatedFFN(C, kernel_size=3, act_layer=torch.nn.GELU()).eval(); randomise(m)
    put("gffn", m, m(x))
    m = sm.StripedConvFormer(C, kernel_size=11).eval(); randomise(m)
    put("scf", m, m(x))
    m = sm.SME(C, kernel_size=11).eval(); randomise(m)
    put("sme", m, m(x))
    for low in (2, 7, 64):
        m = sm.Expert(C, low).eval(); randomise(m)
        put(f"expert{low}", m, m(x, k))
    m = sm.Router(C, 6).eval(); randomise(m)
    put("router", m, m(x))
    experts = [sm.Expert(C, i + 2) for i in range(6)]
    m = sm.MoELayer(experts=experts, gate=sm.Router(C, 6), num_expert=2); randomise(m)
    m.train()
    put("moelayer_train", m, m(x, k))
    m.eval()
    out["moelayer_eval.y"] = m(x[:1], k[:1])       # same parameters, first image
    for space in ("linear", "exp", "double"):
        m = sm.MoEBlock(C, num_experts=6, topk=2, use_shuffle=False, lr_space=space, recursive=2); randomise(m)
        m.train()
        put(f"moeblock_{space}_train", m, m(xb))
        m.eval()
        out[f"moeblock_{space}_eval.y"] = m(xb[1:])
    m = sm.MoEBlock(C, num_experts=4, topk=1, use_shuffle=True, lr_space="exp", recursive=1); randomise(m)
    m.train()
    put("moeblock_shuffle_train", m, m(xb[:, :, :, :16].contiguous()))
    save("seemore", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            globals()["gold_" + name]()
        sys.exit(0)
    gold_bspline()
    gold_kan_layer()
    gold_cell()
    gold_ffn()
    gold_gazetr()
    gold_model()
    gold_cfg1()
    gold_dropout_stats()
    gold_chan_attention()
    gold_color()
    gold_unet()
    gold_model_input_grad()
    gold_seemore()
