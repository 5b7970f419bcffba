"""Checkpoints of the caption path: reference-compatible weight files, full training state for an exact resume,
and the reference's early-stopping rule.

* `save_weights` / `load_weights` read and write what the reference does (`torch.save(model.state_dict(), path)`,
  train.py:286-289, utils.py:53-60; `load_state_dict(..., strict=False)`, train.py:214-216, eval.py:149-151): the
  same keys, fp32 tensors, so a `.pth` written by either side loads into the other.
* `save_training_state` / `load_training_state` add what the reference never stores and a real resume needs: the
  optimizer moments and step, the LR scheduler, the epoch, the early-stopping counters and the device-side dropout
  seed.  Resuming from such a file continues BIT-identically (tests/test_model_gpu.py).
* `EarlyStopping` restates utils.py:8-59 (score = -val_loss; a score below best + delta counts towards patience).
"""
import os
from typing import Optional

import numpy as np
import torch

FORMAT = "vct_amd.training_state.v1"


def save_weights(model, path: str):
    """reference train.py:288 / utils.py:58: the bare state_dict."""
    _atomic_save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)


def load_weights(model, path: str, map_location="cpu", strict: bool = False):
    """reference train.py:214-216: load_state_dict(torch.load(path), strict=False); returns the load report."""
    # weights_only: a .pth from an untrusted source must not be able to run pickled code (tensors and containers only)
    return model.load_state_dict(torch.load(path, map_location=map_location, weights_only=True), strict=strict)


def _atomic_save(obj, path: str):
    tmp = f"{path}.tmp.{os.getpid()}"
    torch.save(obj, tmp)
    os.replace(tmp, path)        # a crash mid-write never leaves a truncated checkpoint behind


def save_training_state(path: str, model, optimizer=None, scheduler=None, epoch: int = 0, early_stopping=None,
                        extra: Optional[dict] = None, write: bool = True):
    """Data-parallel runs: EVERY rank calls this (a sharded optimizer all-gathers its Adam moments first -- optimizer.gather_state(),
    a collective: each rank only keeps the moments of the shards it owns current), and passes write=(rank == 0) so that one rank
    builds the state dictionary and writes the file."""
    seed = getattr(model, "_seed", None)
    if optimizer is not None and hasattr(optimizer, "gather_state"):
        optimizer.gather_state()                         # collective when sharded; every rank
    if not write:
        return
    opt_state = _to_cpu(optimizer.state_dict()) if optimizer is not None else None
    state = {
        "format": FORMAT,
        "model": {k: v.detach().cpu() for k, v in model.state_dict().items()},
        "optimizer": _plain(opt_state),
        "scheduler": _plain(scheduler.state_dict()) if scheduler is not None else None,
        "epoch": int(epoch),
        "early_stopping": _plain(early_stopping.state_dict()) if early_stopping is not None else None,
        "dropout_seed": None if seed is None else int(seed.item()),
        "torch_rng": torch.get_rng_state(),
        "extra": _plain(extra or {}),
    }
    _atomic_save(state, path)


def load_training_state(path: str, model, optimizer=None, scheduler=None, early_stopping=None) -> dict:
    """Restores everything save_training_state stored; returns {'epoch': next epoch to run, 'extra': ...}."""
    # tensors, numbers, strings and containers only (optimizer / scheduler state_dicts are plain data): no pickled code
    state = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(state, dict) or state.get("format") != FORMAT:
        raise ValueError(f"{path} is not a {FORMAT} file (a bare state_dict loads with load_weights)")
    missing, unexpected = model.load_state_dict(state["model"], strict=False)
    if missing:
        raise ValueError(f"training state lacks parameters {list(missing)[:4]}...: a resume must restore every tensor")
    if optimizer is not None:
        if state["optimizer"] is None:
            raise ValueError("training state holds no optimizer state")
        optimizer.load_state_dict(state["optimizer"])
    if scheduler is not None and state["scheduler"] is not None:
        scheduler.load_state_dict(state["scheduler"])
    if early_stopping is not None and state["early_stopping"] is not None:
        early_stopping.load_state_dict(state["early_stopping"])
    if state["dropout_seed"] is not None and getattr(model, "_seed", None) is not None:
        model._seed.fill_(state["dropout_seed"])
    torch.set_rng_state(state["torch_rng"])
    return {"epoch": state["epoch"] + 1, "extra": state["extra"]}


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def _plain(obj):
    """numpy scalars and 0-d tensors -> Python numbers (recursively): load_training_state reads with weights_only=True, which
    refuses numpy's pickled scalar types -- e.g. a ReduceLROnPlateau.best or an early-stopping score that came from np.mean of
    the validation losses would save fine and then fail to load."""
    if isinstance(obj, np.generic):
        return obj.item()
    if torch.is_tensor(obj) and obj.dim() == 0 and obj.numel() == 1 and not obj.is_complex():
        return obj.item()
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v) for v in obj)
    return obj


class EarlyStopping:
    """reference utils.py:8-59.  `__call__(val_loss, model, do_save)`: lower is better; the model is saved (weights
    only, like the reference) whenever the monitored value improves and do_save is True."""

    def __init__(self, patience=7, verbose=False, delta=0, path="checkpoint.pt", trace_func=print):
        self.patience, self.verbose, self.delta, self.path, self.trace_func = patience, verbose, delta, path, trace_func
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = np.inf

    def __call__(self, val_loss, model, do_save):
        score = -val_loss            # utils.py:38: the value is negated once and used negated from there on
        if self.best_score is None:
            self.best_score = score
            self.save_checkpoint(score, model, do_save)
        elif score < self.best_score + self.delta:
            self.counter += 1
            self.trace_func(f"EarlyStopping counter: {self.counter} out of {self.patience}")
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            self.best_score = score
            self.save_checkpoint(score, model, do_save)
            self.counter = 0

    def save_checkpoint(self, val_loss, model, do_save):
        if self.verbose:
            self.trace_func(f"Validation loss decreased ({self.val_loss_min:.6f} --> {val_loss:.6f}).  Saving model ...")
        if do_save is True:
            save_weights(model, self.path)
        self.val_loss_min = val_loss

    def state_dict(self):
        return {"counter": int(self.counter), "best_score": None if self.best_score is None else float(self.best_score),
                "early_stop": bool(self.early_stop), "val_loss_min": float(self.val_loss_min)}

    def load_state_dict(self, sd):
        self.counter, self.best_score, self.early_stop = sd["counter"], sd["best_score"], sd["early_stop"]
        self.val_loss_min = sd["val_loss_min"]
