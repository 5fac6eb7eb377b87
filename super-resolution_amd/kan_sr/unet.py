"""UNet hyper-encoder on the HIP path (reference: conv_sep_kan/unet.py:7-42, blocks conv_sep_kan/unet_blocks.py:9-178).

The reference imports this network next to GazeTR in conv_sep_kan_encoder.py:7 as the alternative backbone that maps
the luma plane [B,1,H,W] to the per-pixel KAN parameters [B,P,H,W].  The classes below are parameter containers with
the reference's attribute names, so `state_dict()` carries the same keys (`proj.double_conv.0.weight`,
`down1.maxpool_conv.1.double_conv.4.running_var`, `att3.a_proj.1.bias`, `up2.up.weight`, `reproj.conv.bias`, ...);
forward and backward are sequences of libksr kernels over NHWC activations (the weight re-layouts and the chain rule
through them are torch ops on the tiny weight tensors).  Any H, W >= 16: for sizes that are not multiples of 16 the
upsampled tensors are zero-padded to their skip's size as the reference does (unet_blocks.py:68-73).
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
        # name -> preallocated gradient tensor: when set (TrainStep does), the backward writes every parameter gradient
        # there and hands autograd no gradients, instead of returning fresh tensors for autograd to accumulate
        self.grad_sink = None
        self._counters = None          # the 18 num_batches_tracked buffers, listed once
        self._named = None             # (names, parameters), listed once: the module tree is fixed

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() replace the buffer tensors: forget everything that was listed or derived from them
        self._counters, self._named = None, None
        self._prepared.clear()
        return super()._apply(fn, *args, **kwargs)

    # ------------------------------------------------------------------ weight layouts (cached per parameter version)
    def invalidate_prepared(self):
        """Drops the cached kernel-layout copies of the weights.  Needed after the parameters were changed behind
        torch's back (a kernel writing through their storage, as TrainStep's Adam does): torch-side updates
        (optimizers, load_state_dict, copy_) bump the version counters the cache is keyed on and need no call."""
        self._prepared.clear()

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
        pad = 1 if ks == 3 else 0
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        out = _new(src0.device, B, Ho, Wo, cout)
        nparts = int(L.ksr_unet_conv_partials(c_int(B), c_int(Ho), c_int(Wo), c_int(c0 + c1), c_int(cout)))
        partials = _new(src0.device, nparts, 2, cout) if stats else None
        sc, sh = affine if affine is not None else (None, None)
        check(L.ksr_unet_conv_fwd(ptr(src0), c_int(c0), ptr(src1), c_int(c1), ptr(sc), ptr(sh), ptr(weight),
                                  c_int64(w_image_stride), ptr(bias), ptr(out), c_int(cout), ptr(partials), c_int(ks),
                                  c_int(stride), c_int(int(reflect)), c_int(B), c_int(H), c_int(W), stream()),
              "ksr_unet_conv_fwd")
        return out, partials

    @staticmethod
    def _wgrad(g, cg, B, Hg, Wg, *, a0=None, ca0=0, a1=None, ca1=0, affine=None, Ha=0, Wa=0, mode=0, ks=3, stride=1,
               reflect=False, per_image=False, ones=False):
        """Weight gradient in the kernels' layout [taps][ca][cg] (leading [B] with per_image); mode 2: column sums [cg].
        ones: [taps][ca+1][cg], the extra row of each tap being the sum of g over that tap's pixels."""
        L = lib()
        ca = ca0 + ca1 + int(ones)
        taps = {0: ks * ks, 1: 4, 2: 1}[mode]
        E = taps * (1 if mode == 2 else ca) * cg
        nslab = int(L.ksr_unet_wgrad_slab_floats(c_int(mode), c_int(ks), c_int(ca), c_int(cg), c_int(B), c_int(Ha), c_int(Wa),
                                                 c_int(Hg), c_int(Wg)))
        slab = _new(g.device, max(1, nslab))
        out = _new(g.device, *((B, E) if per_image else (E,)))
        sc, sh = affine if affine is not None else (None, None)
        check(L.ksr_unet_wgrad(ptr(a0), c_int(ca0), ptr(a1), c_int(ca1), ptr(sc), ptr(sh), ptr(g), c_int(cg), ptr(slab),
                               ptr(out), c_int(int(per_image)), c_int(int(ones)), c_int(mode), c_int(ks), c_int(stride), c_int(int(reflect)),
                               c_int(B), c_int(Ha), c_int(Wa), c_int(Hg), c_int(Wg), stream()), "ksr_unet_wgrad")
        return out

    def _bn(self, bn: nn.BatchNorm2d, partials, count, train):
        C = bn.num_features
        dev = bn.weight.device
        scale, shift, saved = _new(dev, C), _new(dev, C), _new(dev, 2, C)
        if bn.running_mean is None:
            raise KsrError("BatchNorm2d without running statistics has no kernel")
        momentum = 0.1 if bn.momentum is None else bn.momentum
        check(lib().ksr_unet_bn_finalize(ptr(partials), c_int(0 if partials is None else partials.shape[0]), c_int64(count),
                                         ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(bn.running_mean),
                                         ptr(bn.running_var), c_float(momentum), c_float(bn.eps), c_int(int(train)),
                                         ptr(scale), ptr(shift), ptr(saved), c_int(C), stream()), "ksr_unet_bn_finalize")
        return scale, shift, saved

    def _emit(self, grads, name, value):
        """Hands one parameter gradient over: into the sink when there is one, else into the dict for autograd."""
        if self.grad_sink is not None:
            dst = self.grad_sink[name]
            if dst.data_ptr() != value.data_ptr():
                dst.copy_(value)
        else:
            grads[name] = value

    def _bn_bwd(self, bn: nn.BatchNorm2d, dout, raw, aff, train, npix, names):
        """-> gradient of raw; dgamma, dbeta go to names = (weight name, bias name)."""
        L = lib()
        C = bn.num_features
        dev = raw.device
        nparts = int(L.ksr_unet_bn_bwd_partials(c_int64(npix)))
        partials, coef, dx = _new(dev, nparts, 2, C), _new(dev, 3, C), torch.empty_like(raw)
        sink = self.grad_sink
        dgamma = sink[names[0]] if sink is not None else _new(dev, C)      # the kernel writes straight into the sink
        dbeta = sink[names[1]] if sink is not None else _new(dev, C)
        check(L.ksr_unet_bn_relu_bwd(ptr(dout), ptr(raw), ptr(aff[0]), ptr(aff[1]), ptr(bn.weight.detach()), ptr(aff[2]),
                                     c_int(int(train)), ptr(partials), ptr(coef), ptr(dgamma), ptr(dbeta), ptr(dx),
                                     c_int64(npix), c_int(C), stream()), "ksr_unet_bn_relu_bwd")
        return dx, dgamma, dbeta

    def _double_conv(self, key, dc: DoubleConv, src0, c0, B, H, W, train, src1=None, c1=0, pool=False, tape=None):
        seq = dc.double_conv
        cout = seq[0].out_channels
        w1 = self._prep(key + ".0", [seq[0].weight], lambda: _taps(seq[0].weight))
        w2 = self._prep(key + ".3", [seq[3].weight], lambda: _taps(seq[3].weight))
        raw1, part1 = self._conv(src0, c0, w1, cout, B, H, W, src1=src1, c1=c1, stats=train)
        aff1 = self._bn(seq[1], part1, B * H * W, train)
        raw2, part2 = self._conv(raw1, cout, w2, cout, B, H, W, affine=aff1[:2], stats=train)
        aff2 = self._bn(seq[4], part2, B * H * W, train)
        out = _new(src0.device, B, H, W, cout)
        pooled = _new(src0.device, B, H // 2, W // 2, cout) if pool else None
        check(lib().ksr_unet_bn_relu_fwd(ptr(raw2), ptr(aff2[0]), ptr(aff2[1]), ptr(out), ptr(pooled), c_int(B), c_int(H),
                                         c_int(W), c_int(cout), stream()), "ksr_unet_bn_relu_fwd")
        if tape is not None:
            tape[key] = dict(src0=src0, c0=c0, src1=src1, c1=c1, raw1=raw1, aff1=aff1, raw2=raw2, aff2=aff2, out=out,
                             w1=w1, w2=w2, dims=(B, H, W), cout=cout, train=train)
        return out, pooled

    def _double_conv_bwd(self, key, dc: DoubleConv, dout, grads, need_input_grad=True):
        """dout: gradient of the block's activated output.  Fills grads[...] and returns the gradients of src0, src1."""
        t = self._tape[key]
        seq = dc.double_conv
        B, H, W = t["dims"]
        c, c0, c1, train = t["cout"], t["c0"], t["c1"], t["train"]
        npix = B * H * W
        n1, n2 = (key + ".double_conv.1.weight", key + ".double_conv.1.bias"), (key + ".double_conv.4.weight", key + ".double_conv.4.bias")
        draw2, dg2, db2 = self._bn_bwd(seq[4], dout, t["raw2"], t["aff2"], train, npix, n2)
        dw2 = self._wgrad(draw2, c, B, H, W, a0=t["raw1"], ca0=c, affine=t["aff1"][:2], Ha=H, Wa=W)
        dmid, _ = self._conv(draw2, c, t["w2"].flip(0).transpose(1, 2).contiguous(), c, B, H, W)
        draw1, dg1, db1 = self._bn_bwd(seq[1], dmid, t["raw1"], t["aff1"], train, npix, n1)
        dw1 = self._wgrad(draw1, c, B, H, W, a0=t["src0"], ca0=c0, a1=t["src1"], ca1=c1, Ha=H, Wa=W)
        self._emit(grads, key + ".double_conv.0.weight", dw1.view(9, c0 + c1, c).permute(2, 1, 0).unflatten(2, (3, 3)))
        self._emit(grads, n1[0], dg1), self._emit(grads, n1[1], db1)
        self._emit(grads, key + ".double_conv.3.weight", dw2.view(9, c, c).permute(2, 1, 0).unflatten(2, (3, 3)))
        self._emit(grads, n2[0], dg2), self._emit(grads, n2[1], db2)
        if not need_input_grad:
            return None, None
        wf = t["w1"].flip(0).transpose(1, 2)                      # [9][c][c0+c1]
        d0, _ = self._conv(draw1, c, wf[:, :, :c0].contiguous(), c0, B, H, W)
        d1 = None
        if c1:
            d1, _ = self._conv(draw1, c, wf[:, :, c0:].contiguous(), c1, B, H, W)
        return d0, d1

    def _attention(self, name, x, B, H, W, tape=None):
        L = lib()
        att = getattr(self, name)
        c = att.dim
        zc, wz, bz, wv, wp, ta, tv = self._att_weights(name)
        z, _ = self._conv(x, c, wz, zc, B, H, W, bias=bz, ks=3, stride=2, reflect=True)
        Ho, Wo = z.shape[1], z.shape[2]
        gram = self._wgrad(z, zc, B, Ho, Wo, a0=z, ca0=zc, Ha=Ho, Wa=Wo, ks=1, per_image=True)      # z^T z per image
        T = _new(x.device, B, c, c)
        # v is multiplied by the image width in the reference (unet_blocks.py:132,137-138: `w` is rebound to the width)
        check(L.ksr_unet_att_mix(ptr(gram), c_int(zc), c_int(c), c_int(att.num_heads), c_int(B), ptr(ta), ptr(tv), ptr(wv),
                                 ptr(wp), c_float(float(W)), ptr(T), stream()), "ksr_unet_att_mix")
        out, _ = self._conv(x, c, T, c, B, H, W, ks=1, w_image_stride=c * c)
        if tape is not None:
            tape[name] = dict(x=x, z=z, T=T, g=gram, dims=(B, H, W))
        return out

    def _attention_bwd(self, name, dout, grads):
        """dout: gradient of the attention output [B,H,W,c] -> the two gradients of its input (direct path, z path)."""
        L = lib()
        t = self._tape[name]
        att = getattr(self, name)
        c, heads, ca = att.dim, att.num_heads, att.dim // 2
        B, H, W = t["dims"]
        zc, wz, bz, wv, wp, ta, tv = self._att_weights(name)
        x, z = t["x"], t["z"]
        Ho, Wo = z.shape[1], z.shape[2]
        dT = self._wgrad(dout, c, B, H, W, a0=x, ca0=c, Ha=H, Wa=W, ks=1, per_image=True)            # [B][c*c], [in][out]
        dx_direct, _ = self._conv(dout, c, t["T"].transpose(1, 2).contiguous(), c, B, H, W, ks=1, w_image_stride=c * c)
        D, dpar = _new(x.device, B, zc, zc), _new(x.device, B, 2 * c * c + 2 * heads)
        check(L.ksr_unet_att_mix_bwd(ptr(t["g"]), ptr(dT), c_int(c), c_int(heads), c_int(zc), c_int(B), ptr(ta), ptr(tv),
                                     ptr(wv), ptr(wp), c_float(float(W)), ptr(D), ptr(dpar), stream()), "ksr_unet_att_mix_bwd")
        dz, _ = self._conv(z, zc, D, zc, B, Ho, Wo, ks=1, w_image_stride=zc * zc)
        dwzb = self._wgrad(dz, zc, B, Ho, Wo, a0=x, ca0=c, Ha=H, Wa=W, ks=3, stride=2, reflect=True, ones=True).view(9, c + 1, zc)
        dwz, dbz = dwzb[:, :c], dwzb[0, c]                                           # any tap's row of ones sums all of dz
        dx_z = _new(x.device, B, H, W, c)
        check(L.ksr_unet_s2_dgrad(ptr(dz), c_int(zc), ptr(wz.transpose(1, 2).contiguous()), ptr(dx_z), c_int(c), c_int(B),
                                  c_int(H), c_int(W), stream()), "ksr_unet_s2_dgrad")
        # chain rule through the composition of the projection weights (tiny tensors)
        dsum = dpar.sum(0)
        adw = att.a_proj[0].weight.detach().reshape(c, 9)                           # [ci][tap]
        a1 = att.a_proj[1].weight.detach().reshape(ca, c)                            # [co][ci]
        dwa = dwz[:, :, 2 * c:2 * c + ca]                                            # [tap][ci][co]
        self._emit(grads, name + ".q_proj.weight", torch.diagonal(dwz[:, :, 0:c], dim1=1, dim2=2).t().reshape(c, 1, 3, 3))
        self._emit(grads, name + ".k_proj.weight", dwz[:, :, c:2 * c].permute(2, 1, 0).unflatten(2, (3, 3)))
        self._emit(grads, name + ".a_proj.0.weight", (dwa * a1.t()[None]).sum(2).t().reshape(c, 1, 3, 3))
        self._emit(grads, name + ".a_proj.1.weight", (dwa * adw.t()[:, :, None]).sum(0).t().reshape(ca, c, 1, 1))
        self._emit(grads, name + ".a_proj.1.bias", dbz[2 * c:2 * c + ca])
        self._emit(grads, name + ".v_proj.weight", dsum[:c * c].reshape(c, c, 1, 1))
        self._emit(grads, name + ".project_out.weight", dsum[c * c:2 * c * c].reshape(c, c, 1, 1))
        self._emit(grads, name + ".temperature_a", dsum[2 * c * c:2 * c * c + heads].reshape(heads, 1, 1))
        self._emit(grads, name + ".temperature_v", dsum[2 * c * c + heads:].reshape(heads, 1, 1))
        return dx_direct, dx_z

    def _up(self, i, y, cy, skip, B, h, w, train, tape=None):
        up: Up = getattr(self, f"up{i}")
        cu = cy // 2
        wt = self._prep(f"up{i}.up", [up.up.weight],
                        lambda: up.up.weight.detach().permute(2, 3, 0, 1).reshape(4, cy, cu).contiguous())
        u = _new(y.device, B, 2 * h, 2 * w, cu)
        check(lib().ksr_unet_convt_fwd(ptr(y), ptr(wt), ptr(up.up.bias.detach()), ptr(u), c_int(cy), c_int(cu), c_int(B),
                                       c_int(h), c_int(w), stream()), "ksr_unet_convt_fwd")
        hs, ws = skip.shape[1], skip.shape[2]
        oy, ox = (hs - 2 * h) // 2, (ws - 2 * w) // 2
        if (hs, ws) != (2 * h, 2 * w):
            # sizes that are not multiples of 16: MaxPool2d floored on the way down, so the upsampled tensor is zero-padded
            # to the skip's size (F.pad of Up.forward, unet_blocks.py:68-73); a torch copy, taken on this path only
            up_small, u = u, torch.zeros(B, hs, ws, cu, device=y.device, dtype=torch.float32)
            u[:, oy:oy + 2 * h, ox:ox + 2 * w] = up_small
        out, _ = self._double_conv(f"up{i}.conv", up.conv, skip, cu, B, hs, ws, train, src1=u, c1=cu, tape=tape)
        if tape is not None:
            tape[f"up{i}.up"] = dict(y=y, cy=cy, wt=wt, dims=(B, h, w), pad=(oy, ox, hs, ws))
        return out

    def _up_bwd(self, i, dout, grads):
        """dout: gradient of the Up block's output -> (gradient of the skip input, gradient of the low-resolution input)."""
        up: Up = getattr(self, f"up{i}")
        dskip, du = self._double_conv_bwd(f"up{i}.conv", up.conv, dout, grads)
        t = self._tape[f"up{i}.up"]
        B, h, w = t["dims"]
        cy, cu = t["cy"], t["cy"] // 2
        oy, ox, hs, ws = t["pad"]
        if (hs, ws) != (2 * h, 2 * w):
            du = du[:, oy:oy + 2 * h, ox:ox + 2 * w].contiguous()       # backward of the zero padding
        dwt = self._wgrad(du, cu, B, 2 * h, 2 * w, a0=t["y"], ca0=cy, Ha=h, Wa=w, mode=1, ones=True).view(4, cy + 1, cu)
        self._emit(grads, f"up{i}.up.weight", dwt[:, :cy].unflatten(0, (2, 2)).permute(2, 3, 0, 1))
        self._emit(grads, f"up{i}.up.bias", dwt[:, cy].sum(0))           # the four output positions of every input pixel
        dy, _ = self._conv(du, cu, t["wt"].transpose(1, 2).contiguous(), cy, B, 2 * h, 2 * w, ks=2, stride=2)
        return dskip, dy

    # ------------------------------------------------------------------ forward / backward over the whole network
    def _blocks(self):
        return [("proj", self.proj)] + [(f"down{i}.maxpool_conv.1", getattr(self, f"down{i}").maxpool_conv[1]) for i in range(1, 5)]

    def _run(self, x, train, tape):
        B, _, H, W = x.shape
        feats, pooled = [], x                     # [B,1,H,W] is [B,H,W,1]
        cin, h, w = 1, H, W
        for lvl, (key, dc) in enumerate(self._blocks()):
            out, pooled = self._double_conv(key, dc, pooled, cin, B, h, w, train, pool=lvl < 4, tape=tape)
            cin = dc.double_conv[0].out_channels
            feats.append((out, cin, h, w))
            h, w = h // 2, w // 2
        y, cy, h, w = feats[4]
        for i in range(1, 5):
            xs, cs, hs, ws = feats[4 - i]
            skip = self._attention(f"att{5 - i}", xs, B, hs, ws, tape=tape)
            y = self._up(i, y, cy, skip, B, h, w, train, tape=tape)
            cy, h, w = cs, hs, ws
        wr = self._prep("reproj", [self.reproj.conv.weight], lambda: _taps(self.reproj.conv.weight))
        out, _ = self._conv(y, 4, wr, self.out_channels, B, H, W, bias=self.reproj.conv.bias.detach(), ks=1)
        if tape is not None:
            tape["reproj"] = dict(y=y, dims=(B, H, W))
        if train:
            if self._counters is None:
                self._counters = [m.num_batches_tracked for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
            torch._foreach_add_(self._counters, 1)
        return out.view(B * H * W, self.out_channels)

    def _backward(self, tape, dw):
        """dw [B*H*W, P]: gradient of the output -> {parameter name: gradient}."""
        self._tape = tape
        grads = {}
        B, H, W = tape["reproj"]["dims"]
        P = self.out_channels
        y = tape["reproj"]["y"]
        dwb = self._wgrad(dw, P, B, H, W, a0=y, ca0=4, Ha=H, Wa=W, ks=1, ones=True).view(5, P)       # weight rows | bias row
        self._emit(grads, "reproj.conv.weight", dwb[:4].t()[:, :, None, None])
        self._emit(grads, "reproj.conv.bias", dwb[4])
        dy, _ = self._conv(dw, P, self.reproj.conv.weight.detach().reshape(P, 4), 4, B, H, W, ks=1)
        att_grads = {}
        for i in range(4, 0, -1):
            dskip, dy = self._up_bwd(i, dy, grads)
            att_grads[4 - i] = self._attention_bwd(f"att{5 - i}", dskip, grads)       # encoder level 4-i
        blocks = self._blocks()
        dpooled = None
        for lvl in range(4, -1, -1):
            key, dc = blocks[lvl]
            if lvl == 4:
                dout = dy
            else:
                g0, g1 = att_grads[lvl]
                t = tape[key]
                Bq, h, w = t["dims"]
                dout = torch.empty_like(g0)
                check(lib().ksr_unet_grad_merge(ptr(g0), ptr(g1), ptr(dpooled), ptr(t["out"]), ptr(dout), c_int(Bq), c_int(h),
                                                c_int(w), c_int(t["cout"]), stream()), "ksr_unet_grad_merge")
            dpooled, _ = self._double_conv_bwd(key, dc, dout, grads, need_input_grad=lvl > 0)
        self._tape = None
        return grads

    def forward_nhwc(self, y_plane: torch.Tensor) -> torch.Tensor:
        """y_plane [B,1,H,W] (fp32, GPU) -> per-pixel parameters [B*H*W, P] (NHWC); differentiable with respect to the
        parameters (not the image)."""
        if self.in_channels != 1 or self.out_channels % 4:
            raise KsrError("UNet kernels need in_channels = 1 and out_channels a multiple of 4")
        if not y_plane.is_cuda or y_plane.dtype != torch.float32:
            raise KsrError("UNet needs fp32 GPU tensors (the HIP path is the only path)")
        B, _, H, W = y_plane.shape
        if H < 16 or W < 16:
            raise KsrError(f"UNet needs at least 16x16 pixels (four 2x2 poolings), got {H}x{W}")
        x = y_plane.contiguous()
        if self._named is None:
            self._named = tuple(zip(*self.named_parameters()))
        names, params = self._named
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _UNetFn.apply(x, self, names, *params)
        with torch.no_grad():
            return self._run(x, self.training, None)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[B,1,H,W] -> [B,P,H,W], the reference's signature (the relayout to NCHW is a torch copy)."""
        B, _, H, W = x.shape
        w = self.forward_nhwc(x)
        return w.view(B, H, W, self.out_channels).permute(0, 3, 1, 2).contiguous()


class _UNetFn(torch.autograd.Function):
    """The whole network as one tape node: the parameters are inputs only so that autograd hands their gradients back;
    the kernels read them from the module."""

    @staticmethod
    def forward(ctx, x, net, names, *params):
        if x.requires_grad:
            raise KsrError("gradient with respect to the input image is not implemented")
        ctx.net, ctx.names, ctx.tape = net, names, {}
        return net._run(x, net.training, ctx.tape)

    @staticmethod
    def backward(ctx, dw):
        grads = ctx.net._backward(ctx.tape, dw.contiguous())
        ctx.tape = None
        if ctx.net.grad_sink is not None:
            return (None,) * (3 + len(ctx.names))
        return (None, None, None) + tuple(grads[n].contiguous() for n in ctx.names)
