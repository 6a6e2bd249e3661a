"""Gradient validity check + clipping of the training loop (second half).

Restates Trainer._check_and_clip_gradients (train_network.py:368-390): return False (caller skips the optimizer step,
:336-340) if any gradient holds a NaN/Inf, else clip the global L2 norm to `max_norm` like
torch.nn.utils.clip_grad_norm_(parameters, max_norm=1.0).  The reference evaluates `isnan(...).any() or isinf(...).any()`
per parameter -- two device->host synchronisations for each of the model's few hundred tensors; here the multi-tensor
norms (PyTorch's own fused foreach kernels) feed ONE scalar whose finiteness is read back once:
    a gradient contains a non-finite value  <=>  the float64 sum of its squares is non-finite
(float64 accumulation cannot overflow on finite fp32 inputs).
"""
from __future__ import annotations

from typing import Iterable

import torch


def check_and_clip_gradients(parameters: Iterable[torch.nn.Parameter], max_norm: float = 1.0) -> bool:
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return True
    norms = torch._foreach_norm(grads, 2.0)                                   # one multi-tensor launch
    total = torch.linalg.vector_norm(torch.stack([n.double() for n in norms]), 2.0)
    if not bool(torch.isfinite(total)):                                       # the single host sync
        return False
    coef = (max_norm / (total + 1e-6)).clamp(max=1.0).to(grads[0].dtype)     # clip_grad_norm_'s coefficient
    torch._foreach_mul_(grads, coef)
    return True
