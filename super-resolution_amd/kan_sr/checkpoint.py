"""Reading and writing the reference's checkpoints.

The reference trains under Lightning, whose ModelCheckpoint(save_last=True) writes `last.ckpt`
(src/cli/train.py:64-69,81): a torch-saved dict whose 'state_dict' holds the pipeline's tensors under the
attribute path of the model, i.e. with the prefix 'model.' (src/cli/transfer.py:57-58 strips it the same
way), next to 'epoch', 'global_step', 'optimizer_states' (one torch.optim.Adam state_dict,
src/ml/pipelines/default.py:61) and 'lr_schedulers' (CosineAnnealingWarmRestarts, default.py:62-69).

`load_checkpoint` accepts such a file (or one written by `save_checkpoint`) and fills a ConvSepKan module and,
when given, a TrainStep's Adam moments and step count.  `save_checkpoint` writes the same keys, so the
reference's predict / transfer commands can read the weights back; Lightning's own resume additionally needs
its 'loops' / 'callbacks' entries, which only Lightning can produce, and those are not written.

Files are read with torch.load(weights_only=True): nothing in the file is executed.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from ._lib import KsrError
from .train import TrainStep, cosine_warm_restarts_lr

PREFIX = "model."


def _strip(key: str) -> str:
    return key[len(PREFIX):] if key.startswith(PREFIX) else key


def adam_state_dict(ts: TrainStep) -> dict:
    """The TrainStep's optimiser state in torch.optim.Adam.state_dict() form (parameters numbered in
    model.parameters() order; parameters outside the trained layer have no entry, as in torch)."""
    flat_index = {ts.pre + n: i for i, n in enumerate(ts.order)}
    offsets, off = {}, 0
    named = dict(ts.model.named_parameters())
    for n in ts.order:
        k = named[ts.pre + n].numel()
        offsets[ts.pre + n] = (off, k)
        off += k
    state = {}
    names = [n for n, _ in ts.model.named_parameters()]
    for i, n in enumerate(names):
        prm = named[n]
        if n not in flat_index or ts.steps == 0:
            continue                      # torch keeps no state for a parameter that never had a gradient
        o, k = offsets[n]
        state[i] = {"step": torch.tensor(float(ts.steps)), "exp_avg": ts.exp_avg[o:o + k].view(prm.shape).clone(),
                    "exp_avg_sq": ts.exp_avg_sq[o:o + k].view(prm.shape).clone()}
    group = {"lr": ts.lr, "betas": tuple(ts.betas), "eps": ts.eps, "weight_decay": ts.wd, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "initial_lr": ts.base_lr, "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def load_adam_state(ts: TrainStep, opt_state: dict) -> None:
    """Inverse of adam_state_dict: copies the moments of the trained layer into the flat buffers."""
    names = [n for n, _ in ts.model.named_parameters()]
    named = dict(ts.model.named_parameters())
    off = 0
    steps = 0
    for n in ts.order:
        full = ts.pre + n
        k = named[full].numel()
        st = opt_state["state"].get(names.index(full))
        if st is not None:
            if st["exp_avg"].numel() != k or st["exp_avg_sq"].numel() != k:
                raise KsrError(f"optimizer state of {full}: {st['exp_avg'].numel()} elements in the checkpoint, {k} in the "
                               "model (a checkpoint of a different layer configuration?)")
            ts.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
            ts.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            steps = max(steps, int(float(st["step"])))
        else:
            ts.exp_avg[off:off + k].zero_()
            ts.exp_avg_sq[off:off + k].zero_()
        off += k
    ts.steps = steps
    groups = opt_state.get("param_groups") or [{}]
    ts.lr = float(groups[0].get("lr", ts.lr))
    ts.base_lr = float(groups[0].get("initial_lr", ts.base_lr))
    # the remaining hyper-parameters travel with the checkpoint as well (torch.optim.Adam's load_state_dict semantics)
    if "betas" in groups[0]:
        ts.betas = tuple(float(b) for b in groups[0]["betas"])
    ts.eps = float(groups[0].get("eps", ts.eps))
    ts.wd = float(groups[0].get("weight_decay", ts.wd))
    if groups[0].get("amsgrad") or groups[0].get("maximize"):
        raise KsrError("the checkpoint's optimizer uses amsgrad / maximize, which the fused Adam step does not implement")


def save_checkpoint(model: torch.nn.Module, path: str, epoch: int = 0, global_step: int = 0,
                    train_step: Optional[TrainStep] = None, T_0: int = 500, eta_min: float = 1e-5) -> None:
    """Writes a `last.ckpt`-shaped file: 'state_dict' with the 'model.' prefix, 'epoch', 'global_step' and, with a
    TrainStep, 'optimizer_states' / 'lr_schedulers' in torch's own state_dict layouts."""
    sd = {PREFIX + k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ckpt = {"epoch": int(epoch), "global_step": int(global_step), "state_dict": sd}
    if train_step is not None:
        opt = adam_state_dict(train_step)
        for st in opt["state"].values():
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].cpu(), st["exp_avg_sq"].cpu()
        ckpt["optimizer_states"] = [opt]
        lr_now = cosine_warm_restarts_lr(train_step.base_lr, epoch, T_0, eta_min)
        ckpt["lr_schedulers"] = [{"T_0": T_0, "T_i": T_0, "T_mult": 1, "eta_min": eta_min, "T_cur": epoch % T_0,
                                  "base_lrs": [train_step.base_lr], "last_epoch": int(epoch), "_step_count": int(epoch) + 1,
                                  "_last_lr": [lr_now]}]
    torch.save(ckpt, path)


def load_checkpoint(model: torch.nn.Module, path: str, train_step: Optional[TrainStep] = None,
                    strict: bool = True) -> Dict[str, int]:
    """Loads a reference / Lightning checkpoint (or a bare state_dict file) into `model`; returns
    {'epoch', 'global_step'}.  With `train_step`, also restores Adam's moments and step count."""
    ckpt = torch.load(path, weights_only=True, map_location="cpu")
    if not isinstance(ckpt, dict):
        raise KsrError(f"{path}: not a checkpoint dictionary")
    sd = ckpt.get("state_dict", ckpt)
    sd = {_strip(k): v for k, v in sd.items() if isinstance(v, torch.Tensor)}
    model.load_state_dict(sd, strict=strict)     # copies in place: parameters bound to a flat buffer stay bound
    if train_step is not None and ckpt.get("optimizer_states"):
        load_adam_state(train_step, ckpt["optimizer_states"][0])
    return {"epoch": int(ckpt.get("epoch", 0)), "global_step": int(ckpt.get("global_step", 0))}
