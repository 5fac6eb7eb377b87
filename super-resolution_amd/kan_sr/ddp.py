"""Gradient exchange of a reference-style training run (torch DistributedDataParallel, which is what Lightning's DDP
strategy wraps: src/cli/train.py:58-62 passes `devices` to L.Trainer and lets it pick DDP) through libksr's RCCL
communicator instead of the process group's own all-reduce.

    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])
    ddp.register_comm_hook(KsrCommState(rank, world_size), ksr_allreduce_hook)

Under Lightning the same pair goes into the strategy's constructor (`lightning_strategy()` below); everything else of
the reference's pipeline (DefaultPipeline.training_step, configure_optimizers, ModelCheckpoint) stays as it is, and
`kan_sr.checkpoint` reads and writes its `last.ckpt` files.  The native step `kan_sr.train.TrainStep` remains the
faster path (flat buffers, two overlapped buckets, fused Adam); this module is for users who keep the reference's
trainer.

Communicators.  libksr holds ONE RCCL communicator per process (ksr_comm_init / ksr_comm_destroy), and KsrCommState
creates it: do not run a kan_sr.train.TrainStep with world_size > 1 (which owns and, in close(), destroys the same
communicator) in a process that also uses this hook.  The hook always reduces over ALL ranks of that communicator, so the
DDP instance must span the default process group (KsrCommState checks the world size against torch.distributed's; a DDP
built on a sub-group is refused by the size check or would be averaged over the wrong ranks - do not do that).  torch DDP
itself only needs its process group for the parameter broadcast at construction and for its bucket bookkeeping once the
hook carries the gradients: initialise torch.distributed with the "gloo" backend (as bench.py does) and no second RCCL
communicator exists in the process; with the "nccl" backend the two communicators coexist (tested with one rank only: no
multi-GPU machine was available to this build) and each takes its own share of device memory and channels.
The hook returns a COMPLETED future: the collective and the division are enqueued on the stream DDP calls the hook on
(its current stream), and DDP's copy-back of the bucket into the parameters' .grad runs on that same stream afterwards."""
from ctypes import c_void_p
from typing import Optional

import torch
import torch.distributed

from ._lib import KsrError
from .train import RcclComm


class KsrCommState:
    """State object of the hook: owns the RCCL communicator (created over the already initialised torch.distributed
    group, which only carries the 128-byte unique id) unless one is passed in."""

    def __init__(self, rank: int, world_size: int, comm=None):
        if world_size < 1 or not (0 <= rank < world_size):
            raise KsrError(f"bad rank / world size: {rank} / {world_size}")
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() != world_size:
            raise KsrError(f"KsrCommState(world_size={world_size}) in a process group of {dist.get_world_size()} ranks: the "
                           "hook reduces over every rank of libksr's communicator, which must be DDP's default group")
        self.rank, self.world_size = rank, world_size
        self.comm = comm if comm is not None else RcclComm(rank, world_size)
        self.buckets_reduced = 0

    def close(self) -> None:
        self.comm.close()


def ksr_allreduce_hook(state: KsrCommState, bucket: torch.distributed.GradBucket) -> torch.futures.Future[torch.Tensor]:
    """torch DDP communication hook: sums the bucket's flat fp32 gradient buffer over the ranks (RCCL over xGMI through
    libksr) and divides by the world size - the mean DDP's default hook produces.  The collective is enqueued on the
    stream DDP calls the hook on, so the completed future it returns is ordered before DDP's copy-back."""
    buf = bucket.buffer()
    rccl = isinstance(state.comm, RcclComm)
    if buf.dtype is not torch.float32 or (rccl and not buf.is_cuda):
        raise KsrError(f"ksr_allreduce_hook needs float32 GPU gradient buckets, got {buf.dtype} on {buf.device}")
    # (a communicator that is not libksr's RCCL wrapper - the CPU stand-in of the multi-process tests - takes CPU buckets)
    state.comm.allreduce_sum(buf, c_void_p(torch.cuda.current_stream().cuda_stream) if buf.is_cuda else None)
    if state.world_size > 1:
        buf.div_(state.world_size)
    state.buckets_reduced += 1
    fut: torch.futures.Future = torch.futures.Future()
    fut.set_result(buf)
    return fut


def lightning_strategy(rank: Optional[int] = None, world_size: Optional[int] = None, **kwargs):
    """`lightning.pytorch.strategies.DDPStrategy` whose gradient buckets go through ksr_allreduce_hook:
    L.Trainer(strategy=lightning_strategy(), devices=n, ...) in place of the reference's L.Trainer(devices=n, ...)
    (src/cli/train.py:58-62).  The communicator is created when the first bucket arrives (the process group exists
    by then).  Needs the `lightning` package, which the reference's trainer needs as well."""
    try:
        from lightning.pytorch.strategies import DDPStrategy
    except ImportError as e:       # pragma: no cover - lightning is not part of this build's environment
        raise KsrError("lightning_strategy() needs the `lightning` package (the reference's trainer uses it too)") from e

    class _LazyState:
        def __init__(self):
            self.inner: Optional[KsrCommState] = None

    lazy = _LazyState()

    def hook(state: _LazyState, bucket: torch.distributed.GradBucket) -> torch.futures.Future[torch.Tensor]:
        if state.inner is None:
            import torch.distributed as dist
            state.inner = KsrCommState(dist.get_rank() if rank is None else rank,
                                       dist.get_world_size() if world_size is None else world_size)
        return ksr_allreduce_hook(state.inner, bucket)

    # find_unused_parameters: with more than one layer the reference's loop leaves the earlier layers without gradients
    kwargs.setdefault("find_unused_parameters", True)
    return DDPStrategy(ddp_comm_hook=hook, ddp_comm_state=lazy, **kwargs)
