"""Diagnostic: per-pixel forward error of the bf16-storage mode against the oracle and the first LayerNorm's input variance."""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "super-resolution_amd")):
    sys.path.insert(0, p)
from kan_sr import engine
from kan_sr.module import ConvSepKan
from kan_sr.init import init_like_training_setup
from oracle import convsepkan_oracle as orc

seed = int(sys.argv[1]); B, H, W = 64, 32, 32; r = 4
torch.manual_seed(seed)
m = ConvSepKan([1], [16], [3], 5, 3, 0.1, [-1, 1], 4).eval()
init_like_training_setup(m, 0)
p = {n: v.detach().clone() for n, v in m.state_dict().items()}
x = torch.rand(B, 3, H, W)
with torch.no_grad():
    yr = orc.conv_sep_kan_forward(x, p, out_channels=[16], grid_size=5, spline_order=3, grid_range=[-1, 1], upscale=4)
    pre = "layers.0.encoder.encoder."
    h = orc.ffn(x[:, :1], p, pre + "ffn1.")
    h = torch.sigmoid(orc.gated_ffn(h, p, pre + "ffn2."))
    h = orc.ffn(h, p, pre + "ffn3.")
    h = torch.sigmoid(orc.gated_ffn(h, p, pre + "ffn4."))
    s = h.permute(2, 3, 0, 1).reshape(H * W, B, 64)
    q = pre + "encoder.layers.0."
    qkv = s @ p[q + "self_attn.in_proj_weight"].t() + p[q + "self_attn.in_proj_bias"]
    qq, kk, vv = qkv.split(64, dim=-1)
    f = lambda t: t.reshape(H * W, B, 2, 32).transpose(1, 2)
    att = torch.softmax(f(qq) @ f(kk).transpose(-1, -2) / 32 ** 0.5, dim=-1)
    ctx = (att @ f(vv)).transpose(1, 2).reshape(H * W, B, 64)
    a = ctx @ p[q + "self_attn.out_proj.weight"].t() + p[q + "self_attn.out_proj.bias"]
    var = (s + a).var(dim=-1, unbiased=False)          # [HW, B]
    print("LN1 input: mean %.4f  std over channels: median %.3e min %.3e" % (float((s + a).mean()), float(var.sqrt().median()), float(var.sqrt().min())))
    m = m.cuda(); m.precision = "bf16s"
    y = m(x.cuda()).cpu(); engine.set_precision("fp32")
    err = (y[:, 0] - yr[:, 0]).reshape(B, H, r, W, r).permute(0, 1, 3, 2, 4).reshape(B, H, W, 16).pow(2).mean(-1).sqrt()   # per LR pixel
    sig = yr[:, 0].pow(2).mean().sqrt()
    e = err.flatten() / sig
    print("per-pixel rel err: median %.3e  p99 %.3e  max %.3e; share of squared error in top 1%% pixels: %.3f" % (
        float(e.median()), float(e.quantile(0.99)), float(e.max()), float(e.topk(len(e) // 100).values.pow(2).sum() / e.pow(2).sum())))
    sd = var.sqrt().t().reshape(B, H, W).flatten()
    print("corr(log err, log std) = %.3f" % float(torch.corrcoef(torch.stack([e.log(), sd.log()]))[0, 1]))
    per_img = err.pow(2).mean((1, 2)).sqrt() / sig
    print("per-image err: min %.3e median %.3e max %.3e" % (float(per_img.min()), float(per_img.median()), float(per_img.max())))
