"""Gradient validity check + clipping of the training loop (second half).

Restates Trainer._check_and_clip_gradients (train_network.py:368-390): return False (caller skips the optimizer step,
:336-340) if any gradient holds a NaN/Inf, else clip the global L2 norm to `max_norm` like
torch.nn.utils.clip_grad_norm_(parameters, max_norm=1.0).  The reference evaluates `isnan(...).any() or isinf(...).any()`
per parameter -- two device->host synchronisations for each of the model's few hundred tensors; here the multi-tensor
norms (PyTorch's own fused foreach kernels) feed ONE flag that is read back once:
    a gradient contains a non-finite value  <=>  its max-abs (inf-norm) is non-finite
(a maximum cannot overflow, unlike an fp32 sum of squares: a finite gradient whose tensor L2 norm exceeds ~1.8e19 is still a
finite gradient, which the reference's isnan/isinf scan passes and clip_grad_norm_ rescales).  The clip coefficient uses the
L2 norms combined in float64, each tensor's norm taken on values pre-scaled by that max-abs so that it cannot overflow either.
"""
from __future__ import annotations

import math
from typing import Iterable

import torch


def check_and_clip_gradients(parameters: Iterable[torch.nn.Parameter], max_norm: float = 1.0) -> bool:
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return True
    amax = torch.stack(torch._foreach_norm(grads, float("inf"))).max()       # one multi-tensor launch; NaN / Inf propagate
    norms = torch.stack(torch._foreach_norm(grads, 2.0)).double()            # (overflows to inf only beyond ~1.8e19 per tensor)
    total = torch.linalg.vector_norm(norms, 2.0)
    state = torch.stack([amax.double(), total])
    amax_h, total_h = state.tolist()                                         # the single host sync
    if not math.isfinite(amax_h):
        return False
    if not math.isfinite(total_h):
        # finite gradients whose fp32 sum of squares overflowed: redo the norm on values scaled into range (rare, second sync)
        scaled = torch._foreach_div(grads, amax)
        total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(scaled, 2.0)).double(), 2.0) * amax.double()
    coef = (max_norm / (total + 1e-6)).clamp(max=1.0).to(grads[0].dtype)     # clip_grad_norm_'s coefficient
    torch._foreach_mul_(grads, coef)
    return True
