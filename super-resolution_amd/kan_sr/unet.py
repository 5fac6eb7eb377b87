"""UNet hyper-encoder on the HIP path (reference: conv_sep_kan/unet.py:7-42, blocks conv_sep_kan/unet_blocks.py:9-178).

The reference imports this network next to GazeTR in conv_sep_kan_encoder.py:7 as the alternative backbone that maps
the luma plane [B,1,H,W] to the per-pixel KAN parameters [B,P,H,W].  The classes below are parameter containers with
the reference's attribute names, so `state_dict()` carries the same keys (`proj.double_conv.0.weight`,
`down1.maxpool_conv.1.double_conv.4.running_var`, `att3.a_proj.1.bias`, `up2.up.weight`, `reproj.conv.bias`, ...);
the forward is a sequence of libksr kernels over NHWC activations.  Only the forward exists: asking for gradients
raises KsrError.  H and W must be multiples of 16 (the reference's F.pad branch for other sizes, unet_blocks.py:68-73,
is not built).
"""
from __future__ import annotations

from ctypes import c_float, c_int, c_int64

import torch
import torch.nn as nn

from ._lib import KsrError, check, lib, ptr, stream


def _new(dev, *shape):
    return torch.empty(*shape, device=dev, dtype=torch.float32)


class DoubleConv(nn.Module):                      # unet_blocks.py:9-28
    def __init__(self, cin, cout):
        super().__init__()
        self.double_conv = nn.Sequential(
            nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
            nn.Conv2d(cout, cout, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class Down(nn.Module):                            # unet_blocks.py:34-40
    def __init__(self, cin, cout):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(cin, cout))


class Up(nn.Module):                              # unet_blocks.py:46-63 (bilinear=False)
    def __init__(self, cin, cout):
        super().__init__()
        self.up = nn.ConvTranspose2d(cin, cin // 2, kernel_size=2, stride=2)
        self.conv = DoubleConv(cin, cout)


class OutConv(nn.Module):                         # unet_blocks.py:81-85
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=1)


class SelfAttention(nn.Module):                   # unet_blocks.py:91-129
    def __init__(self, dim, num_heads, bias=False):
        super().__init__()
        self.num_heads, self.dim = num_heads, dim
        self.temperature_a = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.temperature_v = nn.Parameter(torch.ones(num_heads, 1, 1))
        s2 = dict(kernel_size=3, padding=1, stride=2, padding_mode="reflect", bias=bias)
        self.q_proj = nn.Conv2d(dim, dim, groups=dim, **s2)
        self.k_proj = nn.Conv2d(dim, dim, **s2)
        self.v_proj = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)
        self.a_proj = nn.Sequential(nn.Conv2d(dim, dim, groups=dim, **s2), nn.Conv2d(dim, dim // 2, kernel_size=1))
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)
        if bias:
            raise KsrError("SelfAttention(bias=True) has no kernels (the reference's UNet never sets it)")


def _taps(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d weight [cout,cin,kh,kw] -> the kernels' layout [kh*kw][cin][cout]."""
    co, ci, kh, kw = w.shape
    return w.detach().permute(2, 3, 1, 0).reshape(kh * kw, ci, co).contiguous()


def _dw_as_dense(w: torch.Tensor) -> torch.Tensor:
    """Depthwise weight [c,1,3,3] -> dense [9][c][c] with the taps on the diagonal."""
    c = w.shape[0]
    return torch.diag_embed(w.detach().reshape(c, 9).t())          # [9][ci][co], non-zero for ci == co


class UNet(nn.Module):
    """UNet(in_channels, out_channels) as in the reference; in_channels must be 1 and out_channels a multiple of 4."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.proj = DoubleConv(in_channels, 4)
        self.down1, self.down2, self.down3, self.down4 = Down(4, 8), Down(8, 16), Down(16, 32), Down(32, 64)
        self.up1, self.up2, self.up3, self.up4 = Up(64, 32), Up(32, 16), Up(16, 8), Up(8, 4)
        self.att1, self.att2 = SelfAttention(dim=4, num_heads=2), SelfAttention(dim=8, num_heads=2)
        self.att3, self.att4 = SelfAttention(dim=16, num_heads=4), SelfAttention(dim=32, num_heads=4)
        self.reproj = OutConv(4, out_channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        self._prepared: dict = {}

    # ------------------------------------------------------------------ weight layouts (cached per parameter version)
    def _prep(self, key, params, make):
        ver = tuple((t.data_ptr(), t._version) for t in params)
        hit = self._prepared.get(key)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                hit = (ver, make())
            self._prepared[key] = hit
        return hit[1]

    def _att_weights(self, name):
        att = getattr(self, name)
        c = att.dim
        zc = (2 * c + c // 2 + 3) // 4 * 4

        def make():
            wz = torch.zeros(9, c, zc, device=att.q_proj.weight.device)
            wz[:, :, 0:c] = _dw_as_dense(att.q_proj.weight)
            wz[:, :, c:2 * c] = _taps(att.k_proj.weight)
            a1 = att.a_proj[1].weight.detach().reshape(c // 2, c)                       # [co][ci]
            wz[:, :, 2 * c:2 * c + c // 2] = att.a_proj[0].weight.detach().reshape(c, 9).t()[:, :, None] * a1.t()[None]
            bz = torch.zeros(zc, device=wz.device)
            bz[2 * c:2 * c + c // 2] = att.a_proj[1].bias.detach()
            return (wz.contiguous(), bz, att.v_proj.weight.detach().reshape(c, c).contiguous(),
                    att.project_out.weight.detach().reshape(c, c).contiguous(),
                    att.temperature_a.detach().reshape(-1).contiguous(), att.temperature_v.detach().reshape(-1).contiguous())

        ps = [att.q_proj.weight, att.k_proj.weight, att.a_proj[0].weight, att.a_proj[1].weight, att.a_proj[1].bias,
              att.v_proj.weight, att.project_out.weight, att.temperature_a, att.temperature_v]
        return (zc,) + self._prep(name, ps, make)

    # ------------------------------------------------------------------ kernels
    @staticmethod
    def _conv(src0, c0, weight, cout, B, H, W, *, src1=None, c1=0, affine=None, bias=None, stats=False, ks=3, stride=1,
              reflect=False, w_image_stride=0):
        L = lib()
        pad = ks // 2
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        out = _new(src0.device, B, Ho, Wo, cout)
        nparts = int(L.ksr_unet_conv_partials(c_int(B), c_int(Ho), c_int(Wo)))
        partials = _new(src0.device, nparts, 2, cout) if stats else None
        sc, sh = affine if affine is not None else (None, None)
        check(L.ksr_unet_conv_fwd(ptr(src0), c_int(c0), ptr(src1), c_int(c1), ptr(sc), ptr(sh), ptr(weight),
                                  c_int64(w_image_stride), ptr(bias), ptr(out), c_int(cout), ptr(partials), c_int(ks),
                                  c_int(stride), c_int(int(reflect)), c_int(B), c_int(H), c_int(W), stream()),
              "ksr_unet_conv_fwd")
        return out, partials

    def _bn(self, bn: nn.BatchNorm2d, partials, count, train):
        C = bn.num_features
        dev = bn.weight.device
        scale, shift = _new(dev, C), _new(dev, C)
        if bn.running_mean is None:
            raise KsrError("BatchNorm2d without running statistics has no kernel")
        momentum = 0.1 if bn.momentum is None else bn.momentum
        check(lib().ksr_unet_bn_finalize(ptr(partials), c_int(0 if partials is None else partials.shape[0]), c_int64(count),
                                         ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(bn.running_mean),
                                         ptr(bn.running_var), c_float(momentum), c_float(bn.eps), c_int(int(train)),
                                         ptr(scale), ptr(shift), c_int(C), stream()), "ksr_unet_bn_finalize")
        return scale, shift

    def _double_conv(self, key, dc: DoubleConv, src0, c0, B, H, W, train, src1=None, c1=0, pool=False):
        seq = dc.double_conv
        cout = seq[0].out_channels
        w1 = self._prep(key + ".0", [seq[0].weight], lambda: _taps(seq[0].weight))
        w2 = self._prep(key + ".3", [seq[3].weight], lambda: _taps(seq[3].weight))
        raw1, part1 = self._conv(src0, c0, w1, cout, B, H, W, src1=src1, c1=c1, stats=train)
        aff1 = self._bn(seq[1], part1, B * H * W, train)
        raw2, part2 = self._conv(raw1, cout, w2, cout, B, H, W, affine=aff1, stats=train)
        sc2, sh2 = self._bn(seq[4], part2, B * H * W, train)
        out = _new(src0.device, B, H, W, cout)
        pooled = _new(src0.device, B, H // 2, W // 2, cout) if pool else None
        check(lib().ksr_unet_bn_relu_fwd(ptr(raw2), ptr(sc2), ptr(sh2), ptr(out), ptr(pooled), c_int(B), c_int(H), c_int(W),
                                         c_int(cout), stream()), "ksr_unet_bn_relu_fwd")
        return out, pooled

    def _attention(self, name, x, B, H, W):
        L = lib()
        att = getattr(self, name)
        c = att.dim
        zc, wz, bz, wv, wp, ta, tv = self._att_weights(name)
        z, _ = self._conv(x, c, wz, zc, B, H, W, bias=bz, ks=3, stride=2, reflect=True)
        n = z.shape[1] * z.shape[2]
        slab = _new(x.device, int(L.ksr_unet_att_slab_floats(c_int(c), c_int(B), c_int64(n))))
        T = _new(x.device, B, c, c)
        # v is multiplied by the image width in the reference (unet_blocks.py:132,137-138: `w` is rebound to the width)
        check(L.ksr_unet_att_mix(ptr(z), c_int(zc), c_int(c), c_int(att.num_heads), c_int64(n), c_int(B), ptr(ta), ptr(tv),
                                 ptr(wv), ptr(wp), c_float(float(W)), ptr(slab), ptr(T), stream()), "ksr_unet_att_mix")
        out, _ = self._conv(x, c, T, c, B, H, W, ks=1, w_image_stride=c * c)
        return out

    def _up(self, i, y, cy, skip, B, h, w, train):
        up: Up = getattr(self, f"up{i}")
        cu = cy // 2
        wt = self._prep(f"up{i}.up", [up.up.weight],
                        lambda: up.up.weight.detach().permute(2, 3, 0, 1).reshape(4, cy, cu).contiguous())
        u = _new(y.device, B, 2 * h, 2 * w, cu)
        check(lib().ksr_unet_convt_fwd(ptr(y), ptr(wt), ptr(up.up.bias.detach()), ptr(u), c_int(cy), c_int(cu), c_int(B),
                                       c_int(h), c_int(w), stream()), "ksr_unet_convt_fwd")
        out, _ = self._double_conv(f"up{i}.conv", up.conv, skip, cu, B, 2 * h, 2 * w, train, src1=u, c1=cu)
        return out

    # ------------------------------------------------------------------ forward
    def forward_nhwc(self, y_plane: torch.Tensor) -> torch.Tensor:
        """y_plane [B,1,H,W] (contiguous, fp32, GPU) -> per-pixel parameters [B*H*W, P] (NHWC)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise KsrError("the UNet hyper-encoder has forward kernels only: call it under torch.no_grad() "
                           "(training uses the GazeTR encoder)")
        if self.in_channels != 1 or self.out_channels % 4:
            raise KsrError("UNet kernels need in_channels = 1 and out_channels a multiple of 4")
        if not y_plane.is_cuda or y_plane.dtype != torch.float32:
            raise KsrError("UNet needs fp32 GPU tensors (the HIP path is the only path)")
        B, _, H, W = y_plane.shape
        if H % 16 or W % 16:
            raise KsrError(f"UNet needs H and W to be multiples of 16, got {H}x{W}")
        train = self.training
        x = y_plane.contiguous()                  # [B,1,H,W] is [B,H,W,1]
        feats, pooled = [], x
        blocks = [("proj", self.proj)] + [(f"down{i}.maxpool_conv.1", getattr(self, f"down{i}").maxpool_conv[1]) for i in range(1, 5)]
        cin, h, w = 1, H, W
        for lvl, (key, dc) in enumerate(blocks):
            out, pooled = self._double_conv(key, dc, pooled, cin, B, h, w, train, pool=lvl < 4)
            cin = dc.double_conv[0].out_channels
            feats.append((out, cin, h, w))
            h, w = h // 2, w // 2
        y, cy, h, w = feats[4]
        for i in range(1, 5):
            xs, cs, hs, ws = feats[4 - i]
            skip = self._attention(f"att{5 - i}", xs, B, hs, ws)
            y = self._up(i, y, cy, skip, B, h, w, train)
            cy, h, w = cs, hs, ws
        wr = self._prep("reproj", [self.reproj.conv.weight], lambda: _taps(self.reproj.conv.weight))
        out, _ = self._conv(y, 4, wr, self.out_channels, B, H, W, bias=self.reproj.conv.bias.detach(), ks=1)
        if train:
            torch._foreach_add_([m.num_batches_tracked for m in self.modules() if isinstance(m, nn.BatchNorm2d)], 1)
        return out.view(B * H * W, self.out_channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[B,1,H,W] -> [B,P,H,W], the reference's signature (the relayout to NCHW is a torch copy)."""
        B, _, H, W = x.shape
        w = self.forward_nhwc(x)
        return w.view(B, H, W, self.out_channels).permute(0, 3, 1, 2).contiguous()
