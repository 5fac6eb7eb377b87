"""UNet hyper-encoder on the HIP path (reference: conv_sep_kan/unet.py:7-42, blocks conv_sep_kan/unet_blocks.py:9-178).

The reference imports this network next to GazeTR in conv_sep_kan_encoder.py:7 as the alternative backbone that maps
the luma plane [B,1,H,W] to the per-pixel KAN parameters [B,P,H,W].  The classes below are parameter containers with
the reference's attribute names, so `state_dict()` carries the same keys (`proj.double_conv.0.weight`,
`down1.maxpool_conv.1.double_conv.4.running_var`, `att3.a_proj.1.bias`, `up2.up.weight`, `reproj.conv.bias`, ...);
forward and backward are sequences of libksr kernels over NHWC activations; the re-layouts of the (tiny) weight tensors
into the kernels' layouts, and of the kernels' gradient layouts back into the parameters' shapes with the chain rule
through the composed attention projections, are two tables of strided gathers, one launch each (_GatherPlan).  Any H, W >= 16: for sizes that are not multiples of 16 the
upsampled tensors are zero-padded to their skip's size as the reference does (unet_blocks.py:68-73).
"""
from __future__ import annotations

from ctypes import c_float, c_int, c_int64, c_void_p

import numpy as np
import torch
import torch.nn as nn

from ._lib import KsrError, check, lib, ptr, stream


# One launch for the reductions that finish the plain weight gradients of a backward pass (see UNet._backward)
DEFER_WGRAD_REDUCTIONS = True
_DEFERRED_SLABS = None          # while a backward pass records its reductions: the slabs they will read


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


# KsrGatherDesc / KsrGatherBlock of include/ksr.h
_GATHER_DESC = np.dtype([("src", "<u8"), ("src2", "<u8"), ("dst", "<u8"), ("shape", "<i4", 4), ("sstride", "<i4", 4),
                         ("s2stride", "<i4", 4), ("dstride", "<i4", 4), ("nred", "<i4"), ("rstride", "<i4"),
                         ("r2stride", "<i4"), ("count", "<i4")])
assert _GATHER_DESC.itemsize == 104


class _GatherPlan:
    """A table of small strided re-layouts executed by one ksr_gather_batch launch.  Entries are stated with torch
    views (torch does the stride arithmetic: permute, diagonal, expand, slices): dst[i] = sum_r src[i, r] * src2[i, r]
    over the views' common index space, r running over `red_dim` of src / src2 when given.  The table holds raw
    addresses, so every tensor an entry names is kept alive here and must not be reallocated."""

    def __init__(self, dev):
        self.dev, self.rows, self.keep, self.descs, self.blocks = dev, [], [], None, None

    def add(self, dst, src, src2=None, red_dim=None, flip=()):
        def strides(v):
            shape, st = list(v.shape), list(v.stride())
            n, rs = (shape.pop(red_dim), st.pop(red_dim)) if red_dim is not None else (1, 0)
            return shape, st, n, rs

        shape, ss, nred, rs = strides(src)
        sp = src.data_ptr()
        for d in flip:                                   # flipped source axis: start at its last element, walk backwards
            sp += 4 * (shape[d] - 1) * ss[d]
            ss[d] = -ss[d]
        if list(dst.shape) != shape or len(shape) > 4:
            raise KsrError(f"gather entry: dst {tuple(dst.shape)} vs src index space {tuple(shape)}")
        s2p, s2s, r2s = 0, [0] * len(shape), 0
        if src2 is not None:
            shape2, s2s, n2, r2s = strides(src2)
            if shape2 != shape or n2 != nred:
                raise KsrError("gather entry: src2 does not match src")
            s2p = src2.data_ptr()
        pad = [0] * (4 - len(shape))
        count = 1
        for n in shape:
            count *= n
        self.rows.append((sp, s2p, dst.data_ptr(), [1] * len(pad) + shape, pad + ss, pad + list(s2s), pad + list(dst.stride()),
                          nred, rs, r2s, count))
        self.keep += [dst, src, src2]

    def finalize(self):
        descs = np.zeros(len(self.rows), dtype=_GATHER_DESC)
        blocks = []
        for i, row in enumerate(self.rows):
            descs[i] = row
            blocks += [(i, first) for first in range(0, row[-1], 256)]
        self.nblocks = len(blocks)
        self.descs = torch.from_numpy(descs.view(np.uint8).copy()).to(self.dev)
        self.blocks = torch.tensor(blocks, dtype=torch.int32).to(self.dev)
        return self

    def run(self):
        check(lib().ksr_gather_batch(c_void_p(self.descs.data_ptr()), c_void_p(self.blocks.data_ptr()), c_int(self.nblocks),
                                     stream()), "ksr_gather_batch")


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
        self._W: dict = {}             # key -> weight in a kernel's layout (static buffers of the current _GatherPlan)
        self._prep_plans: dict = {}    # parameter addresses -> (plan, buffers); never dropped: HIP graphs may replay them
        self._prep_ver = None
        self._ptrs, self._sink_key = None, None
        self._bn_eval: dict = {}       # BatchNorm2d -> (versions, (scale, shift, saved)) of the eval-mode affine form
        self._emit_plans: dict = {}    # (batch, addresses) -> (plan, kernel-layout gradient buffers, own gradient buffer)
        # name -> preallocated gradient tensor: when set (TrainStep does), the backward writes every parameter gradient
        # there and hands autograd no gradients, instead of returning fresh tensors for autograd to accumulate
        self.grad_sink = None
        self._counters = None          # the 18 num_batches_tracked buffers, listed once
        self._named = None             # (names, parameters), listed once: the module tree is fixed

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() replace the buffer tensors: forget everything that was listed or derived from them
        self._counters, self._named = None, None
        self._W, self._prep_ver, self._ptrs, self._sink_key = {}, None, None, None
        self._bn_eval.clear()
        self._prep_plans.clear(), self._emit_plans.clear()
        return super()._apply(fn, *args, **kwargs)

    # ------------------------------------------------------------------ weight layouts (one gather launch per parameter version)
    def invalidate_prepared(self):
        """Marks the kernel-layout copies of the weights stale.  Needed after the parameters were changed behind
        torch's back (a kernel writing through their storage, as TrainStep's Adam does): torch-side updates
        (optimizers, load_state_dict, copy_) bump the version counters the copies are keyed on and need no call."""
        self._prep_ver = None
        self._bn_eval.clear()

    def _double_convs(self):
        """(key, module, channels of the first and second input) of the nine DoubleConv blocks."""
        out = [(key, dc, dc.double_conv[0].in_channels, 0) for key, dc in self._blocks()]
        for i in range(1, 5):
            dc = getattr(self, f"up{i}").conv
            cu = dc.double_conv[0].in_channels // 2
            out.append((f"up{i}.conv", dc, cu, cu))
        return out

    def _named_params(self):
        if self._named is None:
            self._named = tuple(zip(*self.named_parameters()))
        return self._named

    def _param_ptrs(self):
        """Addresses of the parameters, read once per forward pass (the backward reuses them)."""
        if self._ptrs is None:
            self._ptrs = tuple(p.data_ptr() for p in self._named_params()[1])
        return self._ptrs

    def _ensure_prepared(self):
        names, params = self._named_params()
        ptrs = self._param_ptrs()
        hit = self._prep_plans.get(ptrs)
        if hit is None:
            with torch.no_grad():
                hit = self._prep_plans[ptrs] = self._build_prep_plan()
            self._prep_ver = None
        ver = (ptrs, tuple(p._version for p in params))
        if self._prep_ver != ver:
            hit[0].run()
            self._prep_ver = ver
        self._W = hit[1]

    def _build_prep_plan(self):
        dev = self.reproj.conv.weight.device
        plan, W = _GatherPlan(dev), {}

        def buf(key, *shape, zero=False):
            W[key] = (torch.zeros if zero else torch.empty)(*shape, device=dev, dtype=torch.float32)
            return W[key]

        for key, dc, c0, c1 in self._double_convs():
            w1, w2 = dc.double_conv[0].weight.detach(), dc.double_conv[3].weight.detach()
            c = w1.shape[0]
            # forward: [tap][cin][cout]; data gradients: taps flipped, [tap][cout][cin] (split by input for the first conv)
            plan.add(buf(key + ".0", 9, c0 + c1, c).view(3, 3, c0 + c1, c), w1.permute(2, 3, 1, 0))
            plan.add(buf(key + ".3", 9, c, c).view(3, 3, c, c), w2.permute(2, 3, 1, 0))
            plan.add(buf(key + ".3.T", 9, c, c).view(3, 3, c, c), w2.permute(2, 3, 0, 1), flip=(0, 1))
            if key != "proj":
                plan.add(buf(key + ".0.T0", 9, c, c0).view(3, 3, c, c0), w1[:, :c0].permute(2, 3, 0, 1), flip=(0, 1))
            if c1:
                plan.add(buf(key + ".0.T1", 9, c, c1).view(3, 3, c, c1), w1[:, c0:].permute(2, 3, 0, 1), flip=(0, 1))
        for i in range(1, 5):
            w = getattr(self, f"up{i}").up.weight.detach()                              # [cy][cu][2][2]
            cy, cu = w.shape[:2]
            plan.add(buf(f"up{i}.up", 4, cy, cu).view(2, 2, cy, cu), w.permute(2, 3, 0, 1))
            plan.add(buf(f"up{i}.up.T", 4, cu, cy).view(2, 2, cu, cy), w.permute(2, 3, 1, 0))
        wr = self.reproj.conv.weight.detach()
        plan.add(buf("reproj", 1, 4, self.out_channels).view(1, 1, 4, self.out_channels), wr.permute(2, 3, 1, 0))
        for name in ("att1", "att2", "att3", "att4"):
            att = getattr(self, name)
            c, ca = att.dim, att.dim // 2
            zc = (2 * c + ca + 3) // 4 * 4
            # z = (q | k | a): the depthwise q projection embedded on the diagonal of a dense weight, the dense k
            # projection, the a branch's depthwise conv composed with its 1x1 conv; [tap][ci][z] and, for the data
            # gradient, [tap][z][ci].  Channels beyond 2.5 c stay zero.
            wz, wzT, bz = buf(name + ".wz", 9, c, zc, zero=True), buf(name + ".wzT", 9, zc, c, zero=True), buf(name + ".bz", zc, zero=True)
            q = att.q_proj.weight.detach().reshape(c, 9).t()                           # [tap][ci]
            k = att.k_proj.weight.detach()
            a0 = att.a_proj[0].weight.detach().reshape(c, 9).t()                       # [tap][ci]
            a1 = att.a_proj[1].weight.detach().reshape(ca, c)                          # [co][ci]
            plan.add(wz[:, :, 0:c].diagonal(dim1=1, dim2=2), q)
            plan.add(wzT[:, 0:c, :].diagonal(dim1=1, dim2=2), q)
            plan.add(wz.view(3, 3, c, zc)[:, :, :, c:2 * c], k.permute(2, 3, 1, 0))
            plan.add(wzT.view(3, 3, zc, c)[:, :, c:2 * c, :], k.permute(2, 3, 0, 1))
            plan.add(wz[:, :, 2 * c:2 * c + ca], a0[:, :, None].expand(9, c, ca), a1.t()[None].expand(9, c, ca))
            plan.add(wzT[:, 2 * c:2 * c + ca, :], a0[:, None, :].expand(9, ca, c), a1[None].expand(9, ca, c))
            plan.add(bz[2 * c:2 * c + ca], att.a_proj[1].bias.detach())
        return plan.finalize(), W

    def _att_weights(self, name):
        att = getattr(self, name)
        c = att.dim
        return ((2 * c + c // 2 + 3) // 4 * 4, self._W[name + ".wz"], self._W[name + ".bz"],
                att.v_proj.weight.detach().view(c, c), att.project_out.weight.detach().view(c, c),
                att.temperature_a.detach().view(-1), att.temperature_v.detach().view(-1))

    # ------------------------------------------------------------------ parameter gradients (one gather launch per backward)
    def _ensure_emit_plan(self, B):
        """Static buffers for the gradients in the kernels' layouts (self._KG), the destinations of the parameter
        gradients (self._gdst: the sink's views, or views of an own flat buffer) and the table that converts one into
        the other."""
        names, params = self._named_params()
        sink = self.grad_sink
        if sink is not None and (self._sink_key is None or self._sink_key[0] is not sink):
            self._sink_key = (sink, tuple(sink[n].data_ptr() for n in names))      # the sink's tensors are fixed views
        key = (B, self._param_ptrs(), None if sink is None else self._sink_key[1])
        hit = self._emit_plans.get(key)
        if hit is None:
            with torch.no_grad():
                hit = self._emit_plans[key] = self._build_emit_plan(B)
        self._emit, self._KG, self._gdst, self._own_flat = hit

    def _build_emit_plan(self, B):
        dev = self.reproj.conv.weight.device
        names, params = self._named_params()
        plan, KG = _GatherPlan(dev), {}
        own_flat = None
        if self.grad_sink is not None:
            gdst = {n: self.grad_sink[n] for n in names}
        else:
            own_flat = torch.zeros(sum(p.numel() for p in params), device=dev, dtype=torch.float32)
            gdst, off = {}, 0
            for n, p in zip(names, params):
                gdst[n] = own_flat[off:off + p.numel()].view(p.shape)
                off += p.numel()

        def buf(key, *shape):
            KG[key] = torch.empty(*shape, device=dev, dtype=torch.float32)
            return KG[key]

        for key, dc, c0, c1 in self._double_convs():
            c = dc.double_conv[0].out_channels
            plan.add(gdst[key + ".double_conv.0.weight"].permute(2, 3, 1, 0), buf(key + ".0", 9, c0 + c1, c).view(3, 3, c0 + c1, c))
            plan.add(gdst[key + ".double_conv.3.weight"].permute(2, 3, 1, 0), buf(key + ".3", 9, c, c).view(3, 3, c, c))
        for i in range(1, 5):
            cy, cu = getattr(self, f"up{i}").up.weight.shape[:2]
            g = buf(f"up{i}.up", 4, cy + 1, cu)           # per output position: weight rows | sum of the gradient
            plan.add(gdst[f"up{i}.up.weight"].permute(2, 3, 0, 1), g.view(2, 2, cy + 1, cu)[:, :, :cy])
            plan.add(gdst[f"up{i}.up.bias"], g[:, cy], red_dim=0)
        P = self.out_channels
        g = buf("reproj", 5, P)                            # weight rows | bias row
        plan.add(gdst["reproj.conv.weight"].view(P, 4).t(), g[:4])
        plan.add(gdst["reproj.conv.bias"], g[4])
        for name in ("att1", "att2", "att3", "att4"):
            att = getattr(self, name)
            c, ca, heads = att.dim, att.dim // 2, att.num_heads
            zc = (2 * c + ca + 3) // 4 * 4
            g = buf(name + ".wz", 9, c + 1, zc)            # gradient of wz [tap][ci][z] | per tap the sum of dz (bias row)
            dpar = buf(name + ".dpar", B, 2 * c * c + 2 * heads)
            a0 = att.a_proj[0].weight.detach().reshape(c, 9)                           # [ci][tap]
            a1 = att.a_proj[1].weight.detach().reshape(ca, c)                          # [co][ci]
            ga = g[:, :c, 2 * c:2 * c + ca]                                            # [tap][ci][co]
            plan.add(gdst[name + ".q_proj.weight"].view(c, 9).t(), g[:, :c, 0:c].diagonal(dim1=1, dim2=2))
            plan.add(gdst[name + ".k_proj.weight"].permute(2, 3, 1, 0), g.view(3, 3, c + 1, zc)[:, :, :c, c:2 * c])
            # chain rule through wz[t, ci, 2c + co] = a0[ci, t] * a1[co, ci]
            plan.add(gdst[name + ".a_proj.0.weight"].view(c, 9).t(), ga, a1.t()[None].expand(9, c, ca), red_dim=2)
            plan.add(gdst[name + ".a_proj.1.weight"].view(ca, c), ga.permute(2, 1, 0), a0[None].expand(ca, c, 9), red_dim=2)
            plan.add(gdst[name + ".a_proj.1.bias"], g[0, c, 2 * c:2 * c + ca])          # any tap's bias row sums all of dz
            cc = c * c
            for pname, lo, n in (("v_proj.weight", 0, cc), ("project_out.weight", cc, cc), ("temperature_a", 2 * cc, heads),
                                 ("temperature_v", 2 * cc + heads, heads)):
                plan.add(gdst[f"{name}.{pname}"].view(n), dpar[:, lo:lo + n], red_dim=0)   # sum over the images
        return plan.finalize(), KG, gdst, own_flat

    def _collect_grads(self):
        """Runs the gradient table -> {name: gradient} (the sink's tensors, or views of a fresh copy of the own buffer)."""
        self._emit.run()
        if self._own_flat is None:
            return self._gdst
        flat, out, off = self._own_flat.clone(), {}, 0
        for n, p in zip(*self._named_params()):
            out[n] = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        return out

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
               reflect=False, per_image=False, ones=False, out=None):
        """Weight gradient in the kernels' layout [taps][ca][cg] (leading [B] with per_image); mode 2: column sums [cg].
        ones: [taps][ca+1][cg], the extra row of each tap being the sum of g over that tap's pixels."""
        L = lib()
        ca = ca0 + ca1 + int(ones)
        taps = {0: ks * ks, 1: 4, 2: 1}[mode]
        E = taps * (1 if mode == 2 else ca) * cg
        nslab = int(L.ksr_unet_wgrad_slab_floats(c_int(mode), c_int(ks), c_int(ca), c_int(cg), c_int(B), c_int(Ha), c_int(Wa),
                                                 c_int(Hg), c_int(Wg)))
        slab = _new(g.device, max(1, nslab))
        if _DEFERRED_SLABS is not None and not per_image:
            _DEFERRED_SLABS.append(slab)          # its reduction is recorded, not run: alive until the flush
        if out is None:
            out = _new(g.device, *((B, E) if per_image else (E,)))
        elif out.numel() != (B * E if per_image else E):
            raise KsrError(f"weight-gradient buffer of {out.numel()} floats for {E} entries")
        sc, sh = affine if affine is not None else (None, None)
        check(L.ksr_unet_wgrad(ptr(a0), c_int(ca0), ptr(a1), c_int(ca1), ptr(sc), ptr(sh), ptr(g), c_int(cg), ptr(slab),
                               ptr(out), c_int(int(per_image)), c_int(int(ones)), c_int(mode), c_int(ks), c_int(stride), c_int(int(reflect)),
                               c_int(B), c_int(Ha), c_int(Wa), c_int(Hg), c_int(Wg), stream()), "ksr_unet_wgrad")
        return out

    def _bn(self, bn: nn.BatchNorm2d, partials, count, train):
        C = bn.num_features
        dev = bn.weight.device
        if bn.running_mean is None:
            raise KsrError("BatchNorm2d without running statistics has no kernel")
        ver = None
        if not train:
            # eval: scale and shift depend on the parameters and running statistics only; kept until one of them changes
            # (torch-side changes bump the version counters, kernels that write them call invalidate_prepared / run in
            # train mode, which clears the cache)
            ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version)
            hit = self._bn_eval.get(bn)
            if hit is not None and hit[0] == ver:
                return hit[1]
        else:
            self._bn_eval.clear()
        scale, shift, saved = _new(dev, C), _new(dev, C), _new(dev, 2, C)
        momentum = 0.1 if bn.momentum is None else bn.momentum
        check(lib().ksr_unet_bn_finalize(ptr(partials), c_int(0 if partials is None else partials.shape[0]), c_int64(count),
                                         ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(bn.running_mean),
                                         ptr(bn.running_var), c_float(momentum), c_float(bn.eps), c_int(int(train)),
                                         ptr(scale), ptr(shift), ptr(saved), c_int(C), stream()), "ksr_unet_bn_finalize")
        if ver is not None:
            self._bn_eval[bn] = (ver, (scale, shift, saved))
        return scale, shift, saved

    def _bn_bwd(self, bn: nn.BatchNorm2d, dout, raw, aff, train, npix, names):
        """-> gradient of raw; dgamma, dbeta go to names = (weight name, bias name)."""
        L = lib()
        C = bn.num_features
        dev = raw.device
        nparts = int(L.ksr_unet_bn_bwd_partials(c_int64(npix)))
        partials, coef, dx = _new(dev, nparts, 2, C), _new(dev, 3, C), torch.empty_like(raw)
        dgamma, dbeta = self._gdst[names[0]], self._gdst[names[1]]          # the kernel writes the parameter gradients in place
        check(L.ksr_unet_bn_relu_bwd(ptr(dout), ptr(raw), ptr(aff[0]), ptr(aff[1]), ptr(bn.weight.detach()), ptr(aff[2]),
                                     c_int(int(train)), ptr(partials), ptr(coef), ptr(dgamma), ptr(dbeta), ptr(dx),
                                     c_int64(npix), c_int(C), stream()), "ksr_unet_bn_relu_bwd")
        return dx

    def _double_conv(self, key, dc: DoubleConv, src0, c0, B, H, W, train, src1=None, c1=0, pool=False, tape=None):
        seq = dc.double_conv
        cout = seq[0].out_channels
        w1, w2 = self._W[key + ".0"], self._W[key + ".3"]
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
                             dims=(B, H, W), cout=cout, train=train)
        return out, pooled

    def _double_conv_bwd(self, key, dc: DoubleConv, dout, need_input_grad=True):
        """dout: gradient of the block's activated output -> the gradients of src0, src1 (parameter gradients go to
        self._KG / self._gdst)."""
        t = self._tape[key]
        seq = dc.double_conv
        B, H, W = t["dims"]
        c, c0, c1, train = t["cout"], t["c0"], t["c1"], t["train"]
        npix = B * H * W
        n1, n2 = (key + ".double_conv.1.weight", key + ".double_conv.1.bias"), (key + ".double_conv.4.weight", key + ".double_conv.4.bias")
        draw2 = self._bn_bwd(seq[4], dout, t["raw2"], t["aff2"], train, npix, n2)
        self._wgrad(draw2, c, B, H, W, a0=t["raw1"], ca0=c, affine=t["aff1"][:2], Ha=H, Wa=W, out=self._KG[key + ".3"])
        dmid, _ = self._conv(draw2, c, self._W[key + ".3.T"], c, B, H, W)
        draw1 = self._bn_bwd(seq[1], dmid, t["raw1"], t["aff1"], train, npix, n1)
        self._wgrad(draw1, c, B, H, W, a0=t["src0"], ca0=c0, a1=t["src1"], ca1=c1, Ha=H, Wa=W, out=self._KG[key + ".0"])
        if not need_input_grad:
            return None, None
        if c0 % 4:
            # the first block's single input channel (asked for only when the image itself requires a gradient): the conv
            # kernels write channels in fours, so the transposed weights are padded with zero output channels
            w1 = seq[0].weight.detach()                                   # [c][c0][3][3] -> taps flipped, [tap][c][c0 .. 4]
            wp = torch.zeros(3, 3, c, 4, device=w1.device, dtype=torch.float32)
            wp[..., :c0] = w1.permute(2, 3, 0, 1).flip(0, 1)
            d4, _ = self._conv(draw1, c, wp, 4, B, H, W)
            return d4.view(B, H, W, 4)[..., :c0].contiguous(), None
        d0, _ = self._conv(draw1, c, self._W[key + ".0.T0"], c0, B, H, W)
        d1 = None
        if c1:
            d1, _ = self._conv(draw1, c, self._W[key + ".0.T1"], c1, B, H, W)
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

    def _attention_bwd(self, name, dout):
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
        D, dpar = _new(x.device, B, zc, zc), self._KG[name + ".dpar"]
        check(L.ksr_unet_att_mix_bwd(ptr(t["g"]), ptr(dT), c_int(c), c_int(heads), c_int(zc), c_int(B), ptr(ta), ptr(tv),
                                     ptr(wv), ptr(wp), c_float(float(W)), ptr(D), ptr(dpar), stream()), "ksr_unet_att_mix_bwd")
        dz, _ = self._conv(z, zc, D, zc, B, Ho, Wo, ks=1, w_image_stride=zc * zc)
        # gradient of the composed weight wz (+ its bias row); the chain rule down to q/k/a_proj is part of the gather table
        self._wgrad(dz, zc, B, Ho, Wo, a0=x, ca0=c, Ha=H, Wa=W, ks=3, stride=2, reflect=True, ones=True, out=self._KG[name + ".wz"])
        dx_z = _new(x.device, B, H, W, c)
        check(L.ksr_unet_s2_dgrad(ptr(dz), c_int(zc), ptr(self._W[name + ".wzT"]), ptr(dx_z), c_int(c), c_int(B),
                                  c_int(H), c_int(W), stream()), "ksr_unet_s2_dgrad")
        return dx_direct, dx_z

    def _up(self, i, y, cy, skip, B, h, w, train, tape=None):
        up: Up = getattr(self, f"up{i}")
        cu = cy // 2
        wt = self._W[f"up{i}.up"]
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
            tape[f"up{i}.up"] = dict(y=y, cy=cy, dims=(B, h, w), pad=(oy, ox, hs, ws))
        return out

    def _up_bwd(self, i, dout):
        """dout: gradient of the Up block's output -> (gradient of the skip input, gradient of the low-resolution input)."""
        up: Up = getattr(self, f"up{i}")
        dskip, du = self._double_conv_bwd(f"up{i}.conv", up.conv, dout)
        t = self._tape[f"up{i}.up"]
        B, h, w = t["dims"]
        cy, cu = t["cy"], t["cy"] // 2
        oy, ox, hs, ws = t["pad"]
        if (hs, ws) != (2 * h, 2 * w):
            du = du[:, oy:oy + 2 * h, ox:ox + 2 * w].contiguous()       # backward of the zero padding
        self._wgrad(du, cu, B, 2 * h, 2 * w, a0=t["y"], ca0=cy, Ha=h, Wa=w, mode=1, ones=True, out=self._KG[f"up{i}.up"])
        dy, _ = self._conv(du, cu, self._W[f"up{i}.up.T"], cy, B, 2 * h, 2 * w, ks=2, stride=2)
        return dskip, dy

    # ------------------------------------------------------------------ forward / backward over the whole network
    def _blocks(self):
        return [("proj", self.proj)] + [(f"down{i}.maxpool_conv.1", getattr(self, f"down{i}").maxpool_conv[1]) for i in range(1, 5)]

    def _run(self, x, train, tape, fuse_head=False):
        B, _, H, W = x.shape
        self._ptrs = None
        self._ensure_prepared()
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
        if fuse_head:         # the caller's KAN-cell kernels apply reproj themselves (ops.kan_lin_head)
            out = y
        else:
            out, _ = self._conv(y, 4, self._W["reproj"], self.out_channels, B, H, W, bias=self.reproj.conv.bias.detach(), ks=1)
        if tape is not None:
            tape["reproj"] = dict(y=y, dims=(B, H, W), fused=fuse_head)
        if train:
            if self._counters is None:
                self._counters = [m.num_batches_tracked for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
            torch._foreach_add_(self._counters, 1)
        return out.view(B * H * W, -1)

    def _backward(self, tape, dw, want_input_grad=False):
        """dw [B*H*W, P]: gradient of the output (with the head fused: [B*H*W, 4], gradient of reproj's input, reproj's
        own parameter gradients being in self._KG["reproj"] already) -> {parameter name: gradient}; with want_input_grad
        the gradient of the input plane [B,H,W,1] is left in self._input_grad."""
        global _DEFERRED_SLABS
        self._tape = tape
        B, H, W = tape["reproj"]["dims"]
        self._ensure_prepared()
        self._ensure_emit_plan(B)
        # The 27 plain weight gradients end in a small fixed-order reduction each, whose results only the parameter-gradient
        # table below reads: they are recorded and run in ONE launch before it (ksr_reduce_defer / ksr_reduce_flush).
        defer = DEFER_WGRAD_REDUCTIONS
        if defer:
            check(lib().ksr_reduce_defer(c_int(1)), "ksr_reduce_defer")
            _DEFERRED_SLABS = []
        try:
            return self._backward_body(tape, dw, want_input_grad, B, H, W, defer)
        except BaseException:
            if defer:
                lib().ksr_reduce_defer(c_int(-1))
            raise
        finally:
            _DEFERRED_SLABS = None

    def _backward_body(self, tape, dw, want_input_grad, B, H, W, defer):
        P = self.out_channels
        y = tape["reproj"]["y"]
        if tape["reproj"]["fused"]:
            dy = dw
        else:
            self._wgrad(dw, P, B, H, W, a0=y, ca0=4, Ha=H, Wa=W, ks=1, ones=True, out=self._KG["reproj"])
            dy, _ = self._conv(dw, P, self.reproj.conv.weight.detach().view(P, 4), 4, B, H, W, ks=1)
        att_grads = {}
        for i in range(4, 0, -1):
            dskip, dy = self._up_bwd(i, dy)
            att_grads[4 - i] = self._attention_bwd(f"att{5 - i}", dskip)       # encoder level 4-i
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
            dpooled, _ = self._double_conv_bwd(key, dc, dout, need_input_grad=lvl > 0 or want_input_grad)
        self._input_grad = dpooled if want_input_grad else None
        self._tape = None
        if defer:
            check(lib().ksr_reduce_flush(stream()), "ksr_reduce_flush")
            check(lib().ksr_reduce_defer(c_int(0)), "ksr_reduce_defer")
        return self._collect_grads()

    def forward_nhwc(self, y_plane: torch.Tensor, fuse_head: bool = False) -> torch.Tensor:
        """y_plane [B,1,H,W] (fp32, GPU) -> per-pixel parameters [B*H*W, P] (NHWC); differentiable with respect to the
        parameters (not the image).  fuse_head: stop before reproj and return its input [B*H*W, 4] for
        ops.kan_lin_head, which applies reproj inside the KAN-cell kernels."""
        if self.in_channels != 1 or self.out_channels % 4:
            raise KsrError("UNet kernels need in_channels = 1 and out_channels a multiple of 4")
        if not y_plane.is_cuda or y_plane.dtype != torch.float32:
            raise KsrError("UNet needs fp32 GPU tensors (the HIP path is the only path)")
        B, _, H, W = y_plane.shape
        if H < 16 or W < 16:
            raise KsrError(f"UNet needs at least 16x16 pixels (four 2x2 poolings), got {H}x{W}")
        x = y_plane.contiguous()
        names, params = self._named_params()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _UNetFn.apply(x, self, names, fuse_head, *params)
        with torch.no_grad():
            return self._run(x, self.training, None, fuse_head)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[B,1,H,W] -> [B,P,H,W], the reference's signature (the relayout to NCHW is a torch copy)."""
        B, _, H, W = x.shape
        w = self.forward_nhwc(x)
        return w.view(B, H, W, self.out_channels).permute(0, 3, 1, 2).contiguous()


class _UNetFn(torch.autograd.Function):
    """The whole network as one tape node: the parameters are inputs only so that autograd hands their gradients back;
    the kernels read them from the module."""

    @staticmethod
    def forward(ctx, x, net, names, fuse_head, *params):
        ctx.net, ctx.names, ctx.tape = net, names, {}
        return net._run(x, net.training, ctx.tape, fuse_head)

    @staticmethod
    def backward(ctx, dw):
        want_gx = ctx.needs_input_grad[0]
        grads = ctx.net._backward(ctx.tape, dw.contiguous(), want_input_grad=want_gx)
        ctx.tape = None
        gx = None
        if want_gx:         # [B,H,W,1] is [B,1,H,W]
            gx, ctx.net._input_grad = ctx.net._input_grad.view(-1, 1, *ctx.net._input_grad.shape[1:3]), None
        if ctx.net.grad_sink is not None:
            return (gx,) + (None,) * (3 + len(ctx.names))
        return (gx, None, None, None) + tuple(grads[n].contiguous() for n in ctx.names)
