"""TEDM proper: the shared-weights pixel classifier over timesteps, its training loop, ensembling and metrics
(BASELINE configs[4]).

Reference pieces mirrored here (host logic only -- the arithmetic of the frozen UNet and of the head's first layer is
HIP, `tedm_amd/dataset_dm.py` + `tedm_amd/head.py`):

* `shared_weights_classifier`  -- trainers/train_datasetDM.py:30-42 (and app.py:66-75): `Rearrange('b (step act) h w ->
  (b step) act h w')` then ONE Conv1x1(960,128)-ReLU-BN-Conv1x1(128,32)-ReLU-BN-Conv1x1(32,1) applied to every timestep.
* `replicate_targets`          -- trainers/train_baseline.py:30,121: `repeat(y, 'b c h w -> (b step) c h w')`.
* `bce_loss`                   -- trainers/train_baseline.py:44-45.
* `ensemble` / `predict`       -- app.py:62-79, auxiliary/postprocessing/testing_shared_weights.py:113-133:
  `sigmoid(logits)` regrouped '(b step) 1 h w -> step b 1 h w', mean over the timesteps, `> .5`.
* `dice` / `precision` / `recall` -- trainers/train_baseline.py:146-161 (0/0 -> nan, averaged with nanmean as :140-142).
* `train_head` / `validate_head`  -- trainers/train_baseline.py:17-144 for the DatasetDM model (Adam on
  `model.classifier.parameters()` only, trainers/train_datasetDM.py:46).

Deliberately NOT reproduced (SURVEY "do not reproduce"): the reference replicates the targets only when
`config.experiment == 'datasetDM'` while train.py:45-48 names the experiment 'TEDM', so `--experiment TEDM` dies with
"Target size must be the same as input size"; here replication follows the MODEL (it has a shared-weights head or not).
"""
from __future__ import annotations

from argparse import Namespace
from typing import Callable, Dict, Iterable, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class StepsToBatch(nn.Module):
    """einops `Rearrange('b (step act) h w -> (b step) act h w', step=n_steps)` without the einops dependency
    (parameter-free, so the state-dict keys of the Sequential behind it are the reference's: '1.weight', '3.*', ...)."""

    def __init__(self, n_steps: int):
        super().__init__()
        self.n_steps = int(n_steps)

    def forward(self, x: Tensor) -> Tensor:
        b, ct, h, w = x.shape
        return x.reshape(b * self.n_steps, ct // self.n_steps, h, w)

    def extra_repr(self) -> str:
        return f"'b (step act) h w -> (b step) act h w', step={self.n_steps}"


def is_steps_to_batch(mod: nn.Module) -> bool:
    """True for `StepsToBatch` and for the einops layer the reference puts there (trainers/train_datasetDM.py:34)."""
    if isinstance(mod, StepsToBatch):
        return True
    pat = getattr(mod, 'pattern', None)
    return type(mod).__name__ == 'Rearrange' and isinstance(pat, str) and \
        ''.join(pat.split()) == 'b(stepact)hw->(bstep)acthw'


def shared_weights_classifier(n_steps: int, feature_channels: int = 960, out_channels: int = 1) -> nn.Sequential:
    """trainers/train_datasetDM.py:33-42.  NB the reference writes the last layer as `nn.Conv2d(32, 1, config.out_channels)`,
    i.e. ONE output channel and a kernel of size out_channels (= 1, train.py:26): kept."""
    return nn.Sequential(StepsToBatch(n_steps), nn.Conv2d(feature_channels, 128, 1), nn.ReLU(), nn.BatchNorm2d(128),
                         nn.Conv2d(128, 32, 1), nn.ReLU(), nn.BatchNorm2d(32), nn.Conv2d(32, 1, out_channels))


def n_steps_of(model) -> int:
    return len(model.steps)


def has_shared_head(model) -> bool:
    c = getattr(model, 'classifier', None)
    return isinstance(c, nn.Sequential) and len(c) > 0 and is_steps_to_batch(c[0])


def replicate_targets(y: Tensor, n_steps: int) -> Tensor:
    """repeat(y, 'b c h w -> (b step) c h w', step=n_steps) (trainers/train_baseline.py:30)."""
    return y.repeat_interleave(n_steps, dim=0)


def bce_loss(pred: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
    """trainers/train_baseline.py:44-45: (loss, expanded_loss[b, c]) of BCE-with-logits."""
    e = F.binary_cross_entropy_with_logits(pred, y, reduction='none').mean(dim=(2, 3))
    return e.mean(), e


def ensemble(logits: Tensor, n_steps: int) -> Tuple[Tensor, Tensor]:
    """(B*n_steps, C, H, W) logits in '(b step)' order -> (probability (B, C, H, W) = mean over the timesteps of the
    sigmoids, mask = probability > .5)   (testing_shared_weights.py:113-133; app.py:79 for B = 1)."""
    p = torch.sigmoid(logits)
    p = p.reshape(-1, n_steps, *p.shape[1:]).mean(dim=1)
    return p, p > .5


def per_step_probabilities(logits: Tensor, n_steps: int) -> Tensor:
    """'(b step) c h w -> step b c h w' sigmoids: the per-timestep predictions testing_shared_weights.py:113-124 scores."""
    p = torch.sigmoid(logits)
    return p.reshape(-1, n_steps, *p.shape[1:]).transpose(0, 1)


def _sum_hw(x: Tensor) -> Tensor:
    return x.sum(dim=(2, 3))


def dice(x_hat: Tensor, x: Tensor) -> Tensor:
    """trainers/train_baseline.py:146-149: per (b, c) 2|A.B| / (|A| + |B|); x_hat bool, x bool or {0,1} float."""
    return 2 * _sum_hw(torch.logical_and(x_hat, x)) / (_sum_hw(x_hat) + _sum_hw(x))


def precision(x_hat: Tensor, x: Tensor) -> Tensor:
    """trainers/train_baseline.py:151-155."""
    tp = _sum_hw(torch.logical_and(x, x_hat))
    fp = _sum_hw(torch.logical_and(torch.logical_not(x.bool()), x_hat))
    return tp / (tp + fp)


def recall(x_hat: Tensor, x: Tensor) -> Tensor:
    """trainers/train_baseline.py:157-161."""
    tp = _sum_hw(torch.logical_and(x, x_hat))
    fn = _sum_hw(torch.logical_and(x, torch.logical_not(x_hat)))
    return tp / (tp + fn)


@torch.no_grad()
def predict(model, x: Tensor, noise: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """app.py:62-79 `predict_TEDM` for a batch: (probability, mask) with the mean over timesteps when the model carries the
    shared-weights head, plain `sigmoid > .5` (app.py:51-60 `predict_LEDM`) otherwise.  Runs the model in eval mode."""
    was = model.training
    model.eval()
    try:
        logits = model(x, noise=noise) if noise is not None else model(x)
    finally:
        model.train(was)
        if was and hasattr(model, 'diffusion_model'):
            model.diffusion_model.eval()
    if has_shared_head(model):
        return ensemble(logits, n_steps_of(model))
    p = torch.sigmoid(logits)
    return p, p > .5


@torch.no_grad()
def validate_head(config: Namespace, model, val_dl: Iterable, dist=None) -> Dict[str, float]:
    """trainers/train_baseline.py:99-144 for a single-channel head.  Reported like the reference: BCE over every (replicated)
    prediction, Dice / precision / recall of the thresholded per-timestep predictions; additionally 'val/dice_ensemble' = the
    Dice of the mean-over-steps mask (what app.py / testing_shared_weights.py report for TEDM)."""
    shared = has_shared_head(model)
    T = n_steps_of(model) if shared else 1
    was = model.training
    model.eval()
    m = {'val/loss': [], 'val/dice': [], 'val/precision': [], 'val/recall': [], 'val/dice_ensemble': []}
    for i, (x, y) in enumerate(val_dl):
        x, y = x.to(config.device), y.to(config.device)
        pred = model(x)
        y_rep = replicate_targets(y, T) if shared else y
        y_hat = torch.sigmoid(pred) > .5
        m['val/dice'].append(dice(y_hat, y_rep))
        m['val/precision'].append(precision(y_hat, y_rep))
        m['val/recall'].append(recall(y_hat, y_rep))
        m['val/loss'].append(F.binary_cross_entropy_with_logits(pred, y_rep, reduction='none').flatten())
        m['val/dice_ensemble'].append(dice(ensemble(pred, T)[1], y))
        if i + 1 == getattr(config, 'max_val_steps', -1) or getattr(config, 'debug', False):
            break
    model.train(was)
    if hasattr(model, 'diffusion_model'):
        model.diffusion_model.eval()
    # Means over ALL samples of all ranks, NaN samples ignored as the reference's nanmean over one device's samples does
    # (trainers/train_baseline.py:146-161): per metric the ranks exchange (sum of the finite values, their count).  A mean of
    # per-rank nanmeans would turn NaN as soon as one rank's shard has no positives.
    sums = {}
    for k in ('val/loss', 'val/dice', 'val/precision', 'val/recall', 'val/dice_ensemble'):
        v = torch.cat(m[k]).double().flatten().cpu()
        fin = torch.isfinite(v) if k != 'val/loss' else torch.ones_like(v, dtype=torch.bool)
        sums[k] = (float(v[fin].sum()), float(fin.sum()))
    if dist is not None:
        from . import dp
        dev = torch.device(config.device) if 'cuda' in str(config.device) else None
        sums = {k: (dp.sum_over_ranks(dist, a, dev), dp.sum_over_ranks(dist, c, dev)) for k, (a, c) in sums.items()}
    return {k: (a / c if c > 0 else float('nan')) for k, (a, c) in sums.items()}


def allreduce_head_grads_(dist, params) -> None:
    """Data-parallel TEDM: the backbone is frozen, only the head's ~127 k gradients cross ranks -- flattened into
    ONE buffer, one sum-all-reduce, divided by the world size.  BatchNorm statistics stay per rank (each rank normalises
    over its own (B*T) pixel batch, the reference's single-device semantics per replica)."""
    if dist is None:
        return
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for p in ps:
        p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
        off += p.numel()


def train_head(config: Namespace, model, optimizer, train_dl: Iterable, val_dl: Optional[Iterable] = None, logger=None,
               step: int = 0, dist=None, on_step: Optional[Callable[[int, float], None]] = None):
    """trainers/train_baseline.py:17-96 for `DatasetDM` (LEDM / LEDMe / TEDM): zero_grad -> model(x) -> BCE -> backward -> Adam
    on the classifier, logging every `log_freq`, validation every `val_freq` with best-checkpoint saving (the reference's dict
    format, trainers/train_base_diffusion.py:164-170) and the `early_stop` rule (:91-93).  The diffusion backbone stays in eval
    mode and receives no gradient.  Returns the model."""
    import os
    from .trainer import save
    shared = has_shared_head(model)
    T = n_steps_of(model) if shared else 1
    rank = dist.get_rank() if dist is not None else 0
    best_val, losses, losses_t = float('inf'), [], []
    model.train()
    if hasattr(model, 'diffusion_model'):
        model.diffusion_model.eval()
    head = list(model.classifier.parameters())
    while True:
        for x, y in train_dl:
            step += 1
            if shared:
                y = replicate_targets(y, T)
            x, y = x.to(config.device), y.to(config.device)
            optimizer.zero_grad()
            pred = model(x)
            loss, expanded = bce_loss(pred, y)
            loss.backward()
            allreduce_head_grads_(dist, head)
            optimizer.step()
            lv = float(loss.detach())
            losses.append(lv)
            if shared:                                              # per-timestep loss (train_baseline.py:56-58)
                losses_t.append(expanded.detach().reshape(-1, T, expanded.shape[1]).mean(dim=(0, 2)).cpu())
            if on_step is not None:
                on_step(step, lv)
            if step % config.log_freq == 0 or getattr(config, 'debug', False):
                avg = sum(losses) / len(losses)
                if rank == 0:
                    print(f'Step {step} - Train loss: {avg:.4f}')
                    if logger is not None:
                        logger.log({'train/loss': avg}, step=step)
                        if shared:
                            per_t = torch.stack(losses_t).mean(0)
                            for i, s in enumerate(model.steps):
                                logger.log({f'train_loss/step_{s}': float(per_t[i])}, step=step)
            if val_dl is not None and (step % config.val_freq == 0 or getattr(config, 'debug', False)):
                res = validate_head(config, model, val_dl, dist)
                if rank == 0 and logger is not None:
                    logger.log(res, step=step)
                if res['val/loss'] < best_val and not getattr(config, 'debug', False):
                    best_val = res['val/loss']
                    if rank == 0:
                        print(f'Step {step} - New best validation loss: {best_val:.4f}, saving model in {config.log_dir}')
                        os.makedirs(str(config.log_dir), exist_ok=True)
                        save(model, optimizer, config, os.path.join(str(config.log_dir), 'best_model.pt'), step)
                elif res['val/loss'] > best_val * 1.5 and getattr(config, 'early_stop', False):
                    print(f'Step {step} - Validation loss increased by more than 50%')
                    return model
            if step >= config.max_steps or getattr(config, 'debug', False):
                return model
