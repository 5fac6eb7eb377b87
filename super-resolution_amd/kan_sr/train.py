"""Data-parallel training step for ConvSepKan on MI355X.

One process per GPU.  Parameters, gradients and Adam state of the live layer are single flat fp32
buffers; the kernels write every gradient in place.  The flat order follows the order in which the
backward FINISHES gradients (feed, transformer layers, final norm | conv trunk), so the first bucket
can be all-reduced on a side stream (RCCL over xGMI) while the conv-trunk backward is still running.
The loss of the step is L1 on the Y plane (loss="l1") or the reference's training_step loss
L1(Y) + 0.15 (1 - SSIM) (loss="l1+ssim", src/ml/pipelines/default.py:79-81; SSIM as torchmetrics defines it,
restated - that library is not available here, so this term's parity is unpinned).

The attention of the hyper-encoder mixes the images of the LOCAL batch (as under the reference's DDP),
so no activation crosses GPUs: the only exchange is the gradient all-reduce.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict
from ctypes import c_float, c_int, c_int64, c_void_p
from typing import Callable, Dict, List, Optional

import torch

from . import engine
from . import _lib
from ._lib import KsrError, check, lib, ptr, stream
from .autograd import PARAM_ORDER, backward_all
from .engine import TX_PARAM_NAMES

# gradient-completion order of the parameter groups (see module docstring)
_TAIL = (["feed.weight", "feed.bias"]
         + [f"encoder.layers.1.{n}" for n in TX_PARAM_NAMES] + ["encoder.norm.weight", "encoder.norm.bias"]
         + [f"encoder.layers.0.{n}" for n in TX_PARAM_NAMES])
_TRUNK = [n for n in PARAM_ORDER if n not in _TAIL]   # ffn1..ffn4 in module order (adjacency preserved)
FLAT_ORDER = _TAIL + _TRUNK


def flat_layout(shapes: Dict[str, torch.Size]):
    """name -> (offset, numel) in FLAT_ORDER, plus the element count of the first bucket and the total."""
    off, table = 0, {}
    for n in FLAT_ORDER:
        k = shapes[n].numel()
        table[n] = (off, k)
        off += k
        if n == _TAIL[-1]:
            tail_end = off
    return table, tail_end, off


def cosine_warm_restarts_lr(base_lr: float, epoch: int, T_0: int = 500, eta_min: float = 1e-5) -> float:
    """Learning rate of CosineAnnealingWarmRestarts(T_0=500, T_mult=1, eta_min=1e-5) after `epoch` scheduler steps
    (src/ml/pipelines/default.py:66-68; Lightning steps it once per epoch)."""
    import math
    t_cur = epoch % T_0
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / T_0)) / 2


def shard_batch(global_batch: int, world_size: int, rank: int):
    """Contiguous shard [lo, hi) of a global batch for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RcclComm:
    """Gradient exchange through libksr's RCCL wrapper: the process's only RCCL communicator.  The 128-byte unique id
    travels over the torch.distributed process group the launcher set up; a CPU (gloo) group is enough and keeps a
    second RCCL communicator out of the process."""

    def __init__(self, rank: int, world_size: int):
        import torch.distributed as dist
        L = lib()
        buf = (ctypes.c_ubyte * 128)()
        if rank == 0:
            check(L.ksr_comm_unique_id(buf), "ksr_comm_unique_id")
        on_gpu = dist.get_backend() == "nccl"
        idt = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device="cuda" if on_gpu else "cpu")
        dist.broadcast(idt, src=0)
        raw = bytes(idt.cpu().tolist())
        check(L.ksr_comm_init(c_int(rank), c_int(world_size), ctypes.c_char_p(raw)), "ksr_comm_init")
        self.world_size = world_size

    def allreduce_sum(self, t: torch.Tensor, hip_stream: c_void_p) -> None:
        check(lib().ksr_allreduce_sum(ptr(t), c_int64(t.numel()), hip_stream), "ksr_allreduce_sum")

    def close(self):
        lib().ksr_comm_destroy()


class GlooComm:
    """Same interface over a CPU torch.distributed group: used by the multi-process CPU tests that
    rehearse the N > 1 host logic (sharding, flat layout, sum-then-scale).  Not used on GPUs."""

    def __init__(self, world_size: int):
        self.world_size = world_size

    def allreduce_sum(self, t: torch.Tensor, hip_stream=None) -> None:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def close(self):
        pass


class TrainStep:
    MAX_GRAPHS = 4          # recorded input shapes kept at once (least recently used is dropped)

    def __init__(self, model, world_size: int = 1, rank: int = 0, lr: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, comm=None, loss: str = "l1", capture: bool = False,
                 device_state: Optional[bool] = None):
        """capture: after two ordinary steps at a given input shape, forward + loss + backward (and, when nothing has
        to be exchanged between GPUs first, Adam) are recorded once into a HIP graph and replayed from then on, so
        that the 130 (GazeTR) to 470 (UNet) launches of a step cost the host one call: small per-GPU batches are bound
        by the host's launch rate otherwise.  Up to MAX_GRAPHS input shapes stay recorded.  Inputs are copied into the
        graph's own buffers unless the caller passes the same tensors again; `loss` / `out` of a replayed step alias
        graph-owned memory that the next replay of that shape overwrites.  While _lib.PROFILE collects per-call events
        the step runs unrecorded.

        device_state (default: capture, for the GazeTR encoder): the step counter, the learning rate and the dropout
        stream word live in device memory (a recorded graph replays its by-value arguments): Adam runs as
        ksr_adam_step_dev, the dropout masks of step t come from a seed drawn once at construction mixed with the step
        counter.  The UNet encoder (no dropout) keeps Adam outside its graph with by-value arguments."""
        if loss not in ("l1", "l1+ssim"):
            raise KsrError(f"loss must be 'l1' or 'l1+ssim', got {loss!r}")
        self.loss_kind = loss
        self.capture = capture
        self._graphs: "OrderedDict" = OrderedDict()
        self.base_lr = lr
        self.model = model
        self.world_size, self.rank = world_size, rank
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        li = len(model.layers) - 1
        self.pre = f"layers.{li}.encoder.encoder."
        layer = model.layers[li].encoder.encoder
        named = dict(layer.named_parameters())
        dev = next(model.parameters()).device
        # The GazeTR encoder with the plain `feed` head has a hand-scheduled backward that writes into the flat gradient
        # buffer in completion order.  Any other model (UNet backbone, channel-attention head) runs its backward through
        # the autograd tape (every node a HIP kernel sequence); its parameters are laid out in module order and their
        # .grad tensors are views of the flat buffer, exchanged in one bucket.
        self.tape_backward = getattr(model, "backbone", "gazetr") != "gazetr" or bool(getattr(model, "attention", False))
        for n, prm in named.items():
            if prm.dtype is not torch.float32:
                raise KsrError(f"TrainStep needs float32 parameters, {n} is {prm.dtype}")
        self.device_state = (capture and not self.tape_backward) if device_state is None else bool(device_state)
        if capture and not self.tape_backward and not self.device_state:
            raise KsrError("capture=True with the GazeTR encoder needs device_state: a recorded graph would replay one "
                           "dropout seed and one Adam step count for ever")
        self.order = [n for n, _ in layer.named_parameters()] if self.tape_backward else FLAT_ORDER
        if self.tape_backward:
            table, off = {}, 0
            for n in self.order:
                table[n] = (off, named[n].numel())
                off += named[n].numel()
            self.tail_end = total = off
        else:
            table, self.tail_end, total = flat_layout({n: named[n].shape for n in FLAT_ORDER})
        self.flat_p = torch.empty(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        self.p: Dict[str, torch.Tensor] = {}
        self.arena: Dict[str, torch.Tensor] = {}
        for n in self.order:
            off, k = table[n]
            prm = named[n]
            view = self.flat_p[off:off + k].view(prm.shape)
            view.copy_(prm.data)
            prm.data = view                       # the module's parameters now live in the flat buffer
            self.p[self.pre + n] = prm.data
            self.arena[self.pre + n] = self.flat_g[off:off + k].view(prm.shape)
            if self.tape_backward:
                prm.grad = self.arena[self.pre + n]   # .grad is a view of the flat buffer
        if self.tape_backward and hasattr(layer, "grad_sink"):
            # the backbone's backward writes each parameter gradient straight into its view (no accumulation pass)
            # (the GazeTR trunk under the channel-attention head: the head has a sink of its own, for the fused head + KAN
            # path; a_dw, which the reference's forward never uses, keeps the zeros the flat buffer starts with)
            layer.grad_sink = {n: self.arena[self.pre + n] for n in self.order
                               if not (getattr(model, "attention", False) and n.startswith("feed."))}
            if getattr(model, "attention", False):
                from .module import FUSED_ATTENTION_HEAD
                if FUSED_ATTENTION_HEAD:
                    layer.feed.grad_sink = {n[len("feed."):]: self.arena[self.pre + n] for n in self.order if n.startswith("feed.")}
        self._steps = 0
        if self.device_state:
            self._ctr = torch.zeros(1, device=dev, dtype=torch.int64)        # step counter = dropout stream word
            self._lr_dev = torch.full((1,), float(lr), device=dev, dtype=torch.float32)
            self._lr_on_dev = float(lr)
            self._seed_base = engine.new_seed()
        self.comm = comm
        if world_size > 1 and comm is None:
            self.comm = RcclComm(rank, world_size)
        self.side = torch.cuda.Stream(device=dev) if self.comm is not None else None
        self.slab = torch.empty(256, device=dev, dtype=torch.float32)
        self._l1 = torch.zeros(1, device=dev, dtype=torch.float32)
        self._total = torch.zeros(1, device=dev, dtype=torch.float32)
        self.loss = self._l1
        self.last_seed = 0

    @property
    def steps(self) -> int:
        return self._steps

    @steps.setter
    def steps(self, value: int) -> None:
        """Optimisation steps taken so far (Adam's bias correction; restored by a checkpoint)."""
        self._steps = int(value)
        if self.device_state:
            self._ctr.fill_(self._steps)

    @property
    def _graph(self):
        """The most recently used recorded graph (None before the first recording)."""
        for ent in reversed(self._graphs.values()):
            if ent["graph"] is not None:
                return ent["graph"]
        return None

    def step(self, x: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
        """One optimisation step on this rank's shard.  Returns the (device) loss tensor of the shard."""
        m = self.model
        engine.set_precision(m.precision)
        x = x.contiguous()
        r = m.upscale
        if x.dim() != 4 or x.shape[1] != 3 or x.dtype is not torch.float32 or not x.is_cuda:
            raise KsrError(f"TrainStep.step: x must be a float32 GPU tensor [B,3,H,W], got {x.dtype} {tuple(x.shape)}")
        want = (x.shape[0], r * x.shape[2], r * x.shape[3])
        if (tgt.dim() != 4 or tgt.dtype is not torch.float32 or tgt.device != x.device or tgt.shape[1] < 1
                or (tgt.shape[0], tgt.shape[2], tgt.shape[3]) != want
                or (self.loss_kind == "l1+ssim" and tgt.shape[1] < 3)):
            raise KsrError(f"TrainStep.step: tgt must be a float32 tensor [{want[0]}, >=1 (3 with the SSIM term), {want[1]}, "
                           f"{want[2]}] on the input's device, got {tgt.dtype} {tuple(tgt.shape)}")
        tgt = tgt.contiguous()
        if self.tape_backward and (self.capture or self.device_state) and self._tape_dropout_live():
            # The tape path draws its dropout seed on the host inside ConvSepKan.forward and passes it by value: a recorded
            # graph would replay one set of masks for ever; and the device-resident seed word is a per-thread setting that
            # autograd's backward thread does not see, so forward and backward masks would differ.
            raise KsrError("TrainStep(capture=True / device_state=True) with live dropout is only supported for the GazeTR "
                           "encoder with the plain feed head; put the model in eval() mode or construct the step without "
                           "capture / device_state")
        if self.device_state and self._lr_on_dev != float(self.lr):
            self._lr_dev.fill_(float(self.lr))
            self._lr_on_dev = float(self.lr)
        if self.capture and _lib.PROFILE is None:
            exchanged = self._replay(x, tgt)
        else:
            exchanged = self._body(x, tgt, overlap_comm=True)
        if self.side is not None and not exchanged:
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)
            self.comm.allreduce_sum(self.flat_g, c_void_p(self.side.cuda_stream))
            main.wait_stream(self.side)
        self._steps += 1
        if not self._adam_in_body():
            self._adam()
        if self.tape_backward:
            encoder = m.layers[-1].encoder.encoder
            if hasattr(encoder, "invalidate_prepared"):
                encoder.invalidate_prepared()         # the weights changed under the cached kernel layouts
        return self.loss

    def _tape_dropout_live(self) -> bool:
        """Whether a forward of the model on the autograd tape would draw dropout masks (GazeTR trunk in train mode)."""
        m = self.model
        if not m.training or getattr(m, "backbone", "gazetr") != "gazetr":
            return False
        return m.layers[-1].encoder.encoder.encoder.layers[0].p_drop > 0

    def _adam_in_body(self) -> bool:
        return self.device_state and self.comm is None

    def _adam(self) -> None:
        L = lib()
        if self.device_state:
            check(L.ksr_adam_step_dev(ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                      c_int64(self.flat_p.numel()), ptr(self._lr_dev), c_float(self.betas[0]),
                                      c_float(self.betas[1]), c_float(self.eps), c_float(self.wd),
                                      c_void_p(self._ctr.data_ptr()), c_float(1.0 / self.world_size), stream()),
                  "ksr_adam_step_dev")
            return
        check(L.ksr_adam_step(ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                              c_int64(self.flat_p.numel()), c_float(self.lr), c_float(self.betas[0]),
                              c_float(self.betas[1]), c_float(self.eps), c_float(self.wd), c_int(self._steps),
                              c_float(1.0 / self.world_size), stream()), "ksr_adam_step")

    def _body(self, x: torch.Tensor, tgt: torch.Tensor, overlap_comm: bool) -> bool:
        """Forward + loss + backward of one step (+ Adam when its inputs are device-resident and nothing has to be
        exchanged first): the part of a step that a HIP graph can hold.  Returns whether the gradients were already
        exchanged (the two overlapped buckets of the GazeTR path)."""
        L = lib()
        if self.device_state:
            self._ctr.add_(1)
            check(L.ksr_set_dropout_seed_source(c_void_p(self._ctr.data_ptr())), "ksr_set_dropout_seed_source")
        try:
            if self.tape_backward:
                self._forward_backward_on_tape(x, tgt)
                exchanged = False
            else:
                exchanged = self._forward_backward_native(x, tgt, overlap_comm)
        finally:
            if self.device_state:
                L.ksr_set_dropout_seed_source(c_void_p(0))
        if self._adam_in_body():
            self._adam()
        return exchanged

    def _forward_backward_native(self, x: torch.Tensor, tgt: torch.Tensor, overlap_comm: bool) -> bool:
        """GazeTR encoder: hand-scheduled forward, loss and backward; with a communicator (and overlap_comm) the two
        gradient buckets are all-reduced on the side stream while the backward continues."""
        L = lib()
        m = self.model
        g = engine.Geometry(x, m.upscale, m.grid_size, m.spline_order, m.grid_range)
        p_drop = m.layers[-1].encoder.encoder.encoder.layers[0].p_drop if m.training else 0.0
        if self.device_state:
            seed = self._seed_base if p_drop > 0 else 0
        else:
            seed = engine.new_seed() if p_drop > 0 else 0
        self.last_seed = seed
        out = torch.empty(g.B, 3, g.H * g.r, g.W * g.r, device=x.device, dtype=torch.float32)
        engine.bilinear_up_into(x, out, g.r, 1, 2, 1)
        sv: dict = {}
        feat = engine.encoder_forward(x, self.p, self.pre, g, sv, p_drop, seed)
        save_us = torch.empty(g.N, g.r * g.r, 2, device=x.device, dtype=torch.float32)
        engine.feed_kan_forward(feat, x, self.p, self.pre, g, out, save_us)
        # loss + dL/dY'
        plane = g.H * g.r * g.W * g.r
        gy = torch.empty(g.B, plane, device=x.device, dtype=torch.float32)
        self._loss_and_luma_gradient(out, tgt, gy, plane, (g.B, g.H * g.r, g.W * g.r), plane)
        main = torch.cuda.current_stream()
        overlap = overlap_comm and self.side is not None

        def tail_done():
            self.side.wait_stream(main)
            self.comm.allreduce_sum(self.flat_g[:self.tail_end], c_void_p(self.side.cuda_stream))

        backward_all(gy, plane, x, self.p, self.pre, g, sv, save_us, p_drop, seed, arena=self.arena,
                     on_tail_done=tail_done if overlap else None, allow_side_stream=overlap_comm)
        if overlap:
            self.side.wait_stream(main)
            self.comm.allreduce_sum(self.flat_g[self.tail_end:], c_void_p(self.side.cuda_stream))
            main.wait_stream(self.side)
        self.out = out
        return overlap

    def _replay(self, x: torch.Tensor, tgt: torch.Tensor) -> bool:
        """This step's _body through the HIP graph of the current input shape (recorded on the third step at that
        shape; the two before it run unrecorded and size every lazily allocated workspace)."""
        key = (tuple(x.shape), tuple(tgt.shape), self.model.training, self.model.precision)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = dict(graph=None, warm=0, gx=torch.empty_like(x),
                                           gt=torch.empty_like(tgt, memory_format=torch.contiguous_format))
            while len(self._graphs) > self.MAX_GRAPHS:
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        if ent["graph"] is None and ent["warm"] < 2:
            ent["warm"] += 1
            return self._body(x, tgt, overlap_comm=True)
        if x.data_ptr() != ent["gx"].data_ptr():
            ent["gx"].copy_(x)
        if tgt.data_ptr() != ent["gt"].data_ptr():
            ent["gt"].copy_(tgt)
        if ent["graph"] is None:
            # A forward of the model between the last Adam step and this recording (validation, a preview) would have
            # refreshed the kernel-layout weight copies and the eval-mode BatchNorm cache already: the recording would
            # then lack the launches that rebuild them, and every replay would run on the weights of this moment.
            encoder = self.model.layers[-1].encoder.encoder
            if hasattr(encoder, "invalidate_prepared"):
                encoder.invalidate_prepared()
            graph = torch.cuda.CUDAGraph()
            # thread_local: only this thread's calls are policed while recording.  Other threads of the process (the
            # process group's watchdog polling its events when world_size > 1) would abort a recording in "global" mode.
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                self._body(ent["gx"], ent["gt"], overlap_comm=False)
            ent.update(graph=graph, loss=self.loss, out=self.out, ssim=getattr(self, "ssim", None))
            # recording launches nothing: the replay below is this step
        ent["graph"].replay()
        self.loss, self.out, self.ssim = ent["loss"], ent["out"], ent["ssim"]
        return False

    def _forward_backward_on_tape(self, x: torch.Tensor, tgt: torch.Tensor) -> None:
        L = lib()
        m = self.model
        with torch.enable_grad():
            out = m(x)
        B, _, Hh, Wh = out.shape
        plane = Hh * Wh
        if getattr(self, "_gfull", None) is None or self._gfull.shape != out.shape:
            self._gfull = torch.empty_like(out)
        gfull = self._gfull
        outd = out.detach()
        self._loss_and_luma_gradient(outd, tgt, gfull, plane, (B, Hh, Wh), 3 * plane)
        enc = m.layers[-1].encoder.encoder
        if getattr(enc, "grad_sink", None) is None or (getattr(m, "attention", False) and enc.feed.grad_sink is None):
            self.flat_g.zero_()                   # autograd accumulates into the views
        out.backward(gfull)
        self.out = outd

    def _loss_and_luma_gradient(self, out, tgt, gy, plane, dims, gy_stride):
        """self.loss = L1(Y) [+ 0.15 (1 - SSIM(out, tgt))] (src/ml/pipelines/default.py:79-81); gy [B, plane] (rows gy_stride
        elements apart) = its gradient with respect to the luma plane of `out`."""
        L = lib()
        B, Hh, Wh = dims
        if self.loss_kind != "l1+ssim":
            check(L.ksr_l1_y_loss(ptr(out), c_int64(3 * plane), ptr(tgt), c_int64(tgt.shape[1] * plane), ptr(gy),
                                  c_int64(gy_stride), c_int(B), c_int64(plane), c_float(1.0), ptr(self.slab), ptr(self._l1),
                                  stream()), "ksr_l1_y_loss")
            self.loss = self._l1
            return
        C = 3
        nslab = max(int(L.ksr_ssim_slab_floats(c_int(C), c_int(B), c_int(Hh), c_int(Wh))),
                    int(L.ksr_ssim_l1_slab_floats(c_int(B), c_int(Hh), c_int(Wh))))
        if getattr(self, "_ssim_ws", None) is None or self._ssim_ws[0].numel() < nslab or self._ssim_ws[1].numel() < 3 * B * (Hh - 10) * (Wh - 10):
            self._ssim_ws = (torch.empty(nslab, device=out.device), torch.empty(3 * B * (Hh - 10) * (Wh - 10), device=out.device),
                             torch.zeros(1, device=out.device))
        slab, maps, val = self._ssim_ws
        check(L.ksr_ssim_fwd(ptr(out), c_int64(3 * plane), ptr(tgt), c_int64(tgt.shape[1] * plane), c_int(C), c_int(B),
                             c_int(Hh), c_int(Wh), ptr(maps), ptr(slab), ptr(val), stream()), "ksr_ssim_fwd")
        scale = -0.15 / (B * C * (Hh - 10) * (Wh - 10))
        # both terms' gradient in one pass over the luma planes (the SSIM forward's finish kernel has consumed slab)
        check(L.ksr_ssim_y_grad_l1(ptr(maps), ptr(out), c_int64(3 * plane), ptr(tgt), c_int64(tgt.shape[1] * plane), ptr(gy),
                                   c_int64(gy_stride), c_int(B), c_int(Hh), c_int(Wh), c_float(scale), c_float(1.0), ptr(slab),
                                   ptr(self._l1), ptr(val), c_float(0.15), ptr(self._total), stream()), "ksr_ssim_y_grad_l1")
        self.ssim = val
        self.loss = self._total             # = l1 + 0.15 (1 - ssim), formed by the kernel that finishes the L1 sum

    def set_epoch(self, epoch: int, T_0: int = 500, eta_min: float = 1e-5) -> float:
        """Applies the reference's CosineAnnealingWarmRestarts schedule for `epoch` and returns the new lr."""
        self.lr = cosine_warm_restarts_lr(self.base_lr, epoch, T_0, eta_min)
        return self.lr

    def close(self):
        if self.comm is not None:
            self.comm.close()
