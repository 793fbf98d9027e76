"""Training-step caller for the diffusion model.

Mirrors trainers/train_CXR14.py:16-114 (the loop `train.py` actually dispatches; train_base_diffusion.py:12-90 is
the same loop over (x, y) pairs): `zero_grad -> train_step -> backward -> Adam step`, logging every
`log_freq`, validation every `val_freq`, best-validation checkpoint in the reference's dict format
`{'model_state_dict', 'optimizer_state_dict', 'config', 'step'}` and `--resume_path` loading.

Differences, all deliberate (SURVEY "do not reproduce" list):
* the optimizer step is never routed around a GradScaler (train_CXR14.py:31-32): the HIP path computes in fp32
  storage / bf16x3 products, there is no loss scaling;
* validation returns the loss only (the reference also samples pictures for TensorBoard through
  `sample_plot_image(..., normalized=...)`, which raises TypeError as shipped);
* under `torch.distributed` the image batch each rank receives is ITS shard; gradients are combined by a bucketed
  all-reduce of the flat gradient buffer launched while the backward runs (tedm_amd.dp), only rank 0 writes checkpoints.

With a `tedm_amd.DiffusionModel` the step runs without autograd (`train_step_native`: q-sample, recorded UNet
forward, L1/p2 loss, backward as HIP kernels).  Any other module exposing `train_step(x)` takes the generic
`loss.backward()` route, which is what the CPU tests of this host logic use.
"""
from __future__ import annotations

import os
from argparse import Namespace
from typing import Callable, Dict, Iterable, Optional, Tuple

import torch

from . import dp


def _images(batch):
    """train_CXR14 loaders yield x; train_base_diffusion loaders yield (x, y)."""
    return batch[0] if isinstance(batch, (tuple, list)) else batch


def save(model, optimizer, config, path, step) -> None:
    """train_CXR14.py:96-102."""
    torch.save({'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict(),
                'config': config, 'step': step}, path)


def load(new_config, path, model_cls=None) -> Tuple[torch.nn.Module, torch.optim.Optimizer, int]:
    """train_CXR14.py:105-114: rebuild the model from the checkpoint's own config (a pickled Namespace: loaded through
    `checkpoint.load_checkpoint`, weights_only=True with Namespace / PosixPath allow-listed), restore model + Adam state,
    return (model, optimizer, step)."""
    if model_cls is None:
        from .diffusion import DiffusionModel as model_cls
    from .checkpoint import load_checkpoint
    ckpt = load_checkpoint(path, map_location=torch.device(new_config.device),
                           allow_unsafe=bool(getattr(new_config, 'allow_unsafe_checkpoint', False)))
    old = ckpt['config']
    for k, v in vars(new_config).items():                        # compare_configs (trainers/utils.py): report only
        if k in vars(old) and vars(old)[k] != v and k not in ('resume_path', 'log_dir', 'device'):
            print(f'config mismatch on {k}: checkpoint {vars(old)[k]!r} vs new {v!r}')
    model = model_cls(old).to(new_config.device)
    model.load_state_dict(ckpt['model_state_dict'])
    opt = torch.optim.Adam(model.parameters(), lr=new_config.lr)
    opt.load_state_dict(ckpt['optimizer_state_dict'])
    return model, opt, ckpt['step']


@torch.no_grad()
def validate(config, model, val_loader, dist=None) -> Dict[str, float]:
    """train_CXR14.py:63-93 without the picture sampling: mean `train_step` loss over at most max_val_steps
    batches (random t, as the reference does for large validation sets), averaged over ranks."""
    model.eval()
    losses = []
    for i, batch in enumerate(val_loader):
        x = _images(batch).to(config.device)
        losses.append(float(model.train_step(x)))
        if i + 1 == getattr(config, 'max_val_steps', -1) or getattr(config, 'debug', False):
            break
    model.train()
    avg = sum(losses) / max(1, len(losses))
    if dist is not None:
        avg = dp.sum_over_ranks(dist, avg, torch.device(config.device) if 'cuda' in str(config.device) else None) / dist.get_world_size()
    return {'val/loss': avg}


def train(config: Namespace, model, optimizer, train_loader: Iterable, val_loader: Optional[Iterable] = None,
          logger=None, step: int = 0, dist=None, on_step: Optional[Callable[[int, float], None]] = None):
    """train_CXR14.py:16-60.  Returns the model after `config.max_steps` steps."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    native = hasattr(model, 'train_step_native') and 'cuda' in str(config.device)
    best_val, train_losses = float('inf'), []
    dpt = dp.DataParallelTrainer(model, optimizer, dist, n_buckets=int(getattr(config, 'grad_buckets', 4))) if native else None
    model.train()
    while True:
        for batch in train_loader:
            step += 1
            x = _images(batch).to(config.device)
            if native:      # HIP forward + loss + backward into model.model.grad_flat, bucketed all-reduce under the backward, Adam
                lv = float(dpt.step(x).detach())                    # already the global-batch loss
            else:
                optimizer.zero_grad()
                loss = model.train_step(x)
                loss.backward()
                if dist is not None:
                    for p in model.parameters():
                        if p.grad is not None:
                            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                            p.grad.div_(world)
                optimizer.step()
                lv = float(loss.detach())
                if dist is not None:
                    lv = dp.sum_over_ranks(dist, lv, x.device if x.is_cuda else None) / world
            train_losses.append(lv)
            if on_step is not None:
                on_step(step, lv)
            if step % config.log_freq == 0 or getattr(config, 'debug', False):
                avg = sum(train_losses) / len(train_losses)
                if rank == 0:
                    print(f'Step {step} - Train loss: {avg:.4f}')
                    if logger is not None:
                        logger.log({'train/loss': avg}, step=step)
            if val_loader is not None and (step % config.val_freq == 0 or getattr(config, 'debug', False)):
                res = validate(config, model, val_loader, dist)
                if rank == 0 and logger is not None:
                    logger.log(res, step=step)
                if res['val/loss'] < best_val and not getattr(config, 'debug', False):
                    best_val = res['val/loss']
                    if rank == 0:
                        print(f'Step {step} - New best validation loss: {best_val:.4f}, saving model in {config.log_dir}')
                        os.makedirs(str(config.log_dir), exist_ok=True)
                        save(model, optimizer, config, os.path.join(str(config.log_dir), 'best_model.pt'), step)
            if step >= config.max_steps or getattr(config, 'debug', False):
                return model
