"""Gradient validity check + clipping of the training loop (second half).

Restates Trainer._check_and_clip_gradients (train_network.py:368-390): return False (caller skips the optimizer step,
:336-340) if any gradient holds a NaN/Inf, else clip the global L2 norm to `max_norm` like
torch.nn.utils.clip_grad_norm_(parameters, max_norm=1.0).  The reference evaluates `isnan(...).any() or isinf(...).any()`
per parameter -- two device->host synchronisations for each of the model's few hundred tensors.

On a HIP device this is libunipre3d_gradclip.so (include/unipre3d_gradclip.h, csrc/u3d_gradclip.hip): ONE multi-tensor launch
over a device table of gradient pointers (per-chunk f64 sum of squares, max-abs, non-finite flag), a one-workgroup fixed-order
finalize, ONE 32-byte host read, and -- only when the norm exceeds max_norm -- one multi-tensor scale launch.
`check_and_clip_deferred` is the same without any host read: the flag and the coefficient stay on the device as the
`found_inf` / `grad_scale` scalars torch's fused AdamW consumes.  No fallback on a HIP device: a missing library raises.
Gradients the pointer-table kernels cannot take (not fp32, not contiguous -- channels_last --, or on another device) are
accepted like the reference accepts them: their statistics are computed with torch ops ON THE DEVICE and folded into the same
state block as one more partial record, and they are scaled by the same device-side coefficient.

CPU tensors (the world-size-2 gloo tests of the DP harness run the training step on the host) take the multi-tensor torch
formulation below; it is host logic, not the product path.
"""
from __future__ import annotations

import ctypes
import math
import os
import struct
from typing import Iterable

import torch

from . import _lib

LIB_PATH = os.path.join(_lib.LIB_DIR, "libunipre3d_gradclip.so")   # (U3D_LIB_DIRNAME: experiment builds, see _lib.py)
EXPORTS = ("u3d_gradclip_stats", "u3d_gradclip_finalize", "u3d_gradclip_scale")
GC_CHUNK = 65536
_gc = None


def load() -> ctypes.CDLL:
    global _gc
    if _gc is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing (no fallback on a HIP device): run `make -C unipre3d_amd/csrc`")
        lib = ctypes.CDLL(LIB_PATH)
        vp, i32 = ctypes.c_void_p, ctypes.c_int32
        lib.u3d_gradclip_stats.argtypes = [vp, vp, vp, i32, i32, vp, vp]
        lib.u3d_gradclip_finalize.argtypes = [vp, i32, ctypes.c_float, vp, vp]
        lib.u3d_gradclip_scale.argtypes = [vp, vp, vp, i32, i32, vp, vp]
        for n in EXPORTS:
            getattr(lib, n).restype = ctypes.c_int
        _gc = lib
    return _gc


class _Table:
    """Device table of one gradient set: pointers, element counts, chunk prefix (+ the per-chunk partials and the state block).
    Rebuilt -- one small H2D copy -- only when a gradient moved or changed size; parameters that keep their .grad storage across
    steps (zero_grad(set_to_none=False), DDP's bucket views) never re-upload."""
    __slots__ = ("key", "n", "n_chunks", "host", "buf", "ptrs", "numel", "first", "partials", "state")

    def __init__(self, key, dev):
        self.key = key
        self.n = len(key)
        first, acc = [], 0
        for _, ne in key:
            first.append(acc)
            acc += (ne + GC_CHUNK - 1) // GC_CHUNK
        first.append(acc)
        self.n_chunks = acc
        n = self.n
        host = torch.empty(2 * n + (n + 2) // 2, dtype=torch.int64, pin_memory=True)     # ptrs | numel | first (int32 pairs)
        if n:
            host[:n] = torch.tensor([p for p, _ in key], dtype=torch.int64)
            host[n:2 * n] = torch.tensor([ne for _, ne in key], dtype=torch.int64)
        host[2 * n:].view(torch.int32)[:n + 1] = torch.tensor(first, dtype=torch.int32)
        self.host = host                                       # (pinned, kept alive: the upload is asynchronous on the current stream)
        self.buf = host.to(dev, non_blocking=True)
        self.ptrs, self.numel, self.first = self.buf[:n], self.buf[n:2 * n], self.buf[2 * n:]
        self.partials = torch.empty((self.n_chunks + 1) * 2, dtype=torch.float64, device=dev)     # 16 B per chunk + one spare record
        self.state = torch.zeros(4, dtype=torch.float64, device=dev)                              # 32 B, see the header


_tables = {}


def _table_for(grads, dev) -> _Table:
    key = tuple((g.data_ptr(), g.numel()) for g in grads)
    slot = (dev.index, len(key))
    t = _tables.get(slot)
    if t is None or t.key != key:
        if len(_tables) >= 16:
            _tables.clear()
        t = _tables[slot] = _Table(key, dev)
    return t


def _device_grads(parameters):
    grads = [p.grad for p in parameters if p.grad is not None]
    grads = [g for g in grads if g.numel() > 0]
    return grads


def _split(grads):
    """(device of the set, the gradients the multi-tensor kernels take, the rest).  The kernels read contiguous fp32 tensors on
    ONE device through a pointer table; anything else -- a channels_last conv weight gradient, a bf16 / fp16 parameter, a shard on
    another device -- is handled like the reference handles every gradient (train_network.py:368-390 accepts any dtype / layout):
    with torch ops, folded into the same state block as one extra partial record."""
    dev = next((g.device for g in grads if g.device.type == "cuda"), grads[0].device)
    fast, rest = [], []
    for g in grads:
        (fast if (g.dtype == torch.float32 and g.is_contiguous() and g.device == dev) else rest).append(g)
    return dev, fast, rest


def _rest_record(rest, dev, partials, slot):
    """Statistics of the gradients the kernels cannot take, written on the device (no host read) into partial record `slot`
    (double sum of squares | float max finite |g| | uint32 non-finite flag, include/unipre3d_gradclip.h)."""
    sumsq = torch.zeros((), dtype=torch.float64, device=dev)
    amax = torch.zeros((), dtype=torch.float32, device=dev)
    bad = torch.zeros((), dtype=torch.bool, device=dev)
    for g in rest:
        gd = g.detach()
        fin = torch.isfinite(gd)
        sumsq = sumsq + gd.double().pow(2).sum().to(dev)
        amax = torch.maximum(amax, torch.where(fin, gd.abs(), torch.zeros((), dtype=gd.dtype, device=gd.device)).max().float().to(dev))
        bad = bad | (~fin.all()).to(dev)
    partials[2 * slot] = sumsq
    rec = partials[2 * slot + 1: 2 * slot + 2]
    rec.view(torch.float32)[0] = amax
    rec.view(torch.int32)[1] = bad.to(torch.int32)


def _scale_rest(rest, t):
    """g *= coef for the gradients outside the pointer table; coef is 1 on the device when nothing is to be clipped or a
    non-finite value was found (so the values stay untouched, like the kernel's early exit)."""
    coef = t.state.view(torch.float32)[4]
    for g in rest:
        g.mul_(coef.to(device=g.device))


def _launch_stats(grads, max_norm):
    from .rasterizer import _stream_ptr
    dev, fast, rest = _split(grads)
    if dev.type != "cuda":
        raise RuntimeError("unipre3d_amd.gradcheck: the HIP path needs gradients on a HIP device")
    lib, t, s = load(), _table_for(fast, dev), _stream_ptr(dev)
    rc = 0
    if fast:
        rc = lib.u3d_gradclip_stats(t.ptrs.data_ptr(), t.numel.data_ptr(), t.first.data_ptr(), t.n, t.n_chunks, t.partials.data_ptr(), s)
    if rest:
        _rest_record(rest, dev, t.partials, t.n_chunks)          # (the table keeps one spare record for them)
    if rc == 0:
        rc = lib.u3d_gradclip_finalize(t.partials.data_ptr(), t.n_chunks + (1 if rest else 0), float(max_norm), t.state.data_ptr(), s)
    if rc != 0:
        raise RuntimeError(f"u3d_gradclip_stats / _finalize failed with code {rc}")
    return lib, t, s, rest


def _launch_scale(lib, t, s, rest=()):
    rc = 0
    if t.n:
        rc = lib.u3d_gradclip_scale(t.ptrs.data_ptr(), t.numel.data_ptr(), t.first.data_ptr(), t.n, t.n_chunks, t.state.data_ptr(), s)
    if rc != 0:
        raise RuntimeError(f"u3d_gradclip_scale failed with code {rc}")
    if rest:
        _scale_rest(rest, t)


def gradient_state(parameters: Iterable[torch.nn.Parameter], max_norm: float = 1.0) -> dict:
    """Diagnostics: the state block of one stats pass, read back (one host sync): total_norm, amax, coef, found_inf."""
    grads = _device_grads(parameters)
    if not grads:
        return {"total_norm": 0.0, "amax": 0.0, "coef": 1.0, "grad_scale": 1.0, "found_inf": False}
    _, t, _, _ = _launch_stats(grads, max_norm)
    total, amax, coef, gscale, found, _ = struct.unpack("ddffff", t.state.cpu().numpy().tobytes())
    return {"total_norm": total, "amax": amax, "coef": coef, "grad_scale": gscale, "found_inf": found != 0.0}


def check_and_clip_gradients(parameters: Iterable[torch.nn.Parameter], max_norm: float = 1.0) -> bool:
    """train_network.py:368-390.  False: some gradient holds a NaN / Inf (gradients left untouched; the caller skips the step)."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return True
    if not any(g.device.type == "cuda" for g in grads):
        return _check_and_clip_host(grads, max_norm)
    grads = [g for g in grads if g.numel() > 0]
    if not grads:
        return True
    lib, t, s, rest = _launch_stats(grads, max_norm)
    total, amax, coef, gscale, found, _ = struct.unpack("ddffff", t.state.cpu().numpy().tobytes())     # the single host sync
    if found != 0.0:
        return False
    if coef != 1.0:
        _launch_scale(lib, t, s, rest)
    return True


def check_and_clip_deferred(parameters: Iterable[torch.nn.Parameter], optimizer: torch.optim.Optimizer, max_norm: float = 1.0) -> None:
    """The same decision WITHOUT a host read, for torch's fused optimizers (AdamW(fused=True)): the optimizer's device-side
    `found_inf` makes it skip the step exactly when the reference would (`if not valid: skip`, train_network.py:336-340), and the
    gradients are scaled by the multi-tensor pass (which leaves at once when nothing is to be clipped).  Call instead of
    `if check_and_clip_gradients(...): optimizer.step()`, then `optimizer.step()` unconditionally.  `optimizer.found_inf` stays a view
    of this gradient set's state block: call this before EVERY step of that optimizer (a step without it would reuse the previous
    decision), or reset `optimizer.found_inf = None` when switching back to the host-read form."""
    grads = _device_grads(parameters)
    if not grads:
        return
    if not any(g.device.type == "cuda" for g in grads):
        raise RuntimeError("check_and_clip_deferred needs gradients on a HIP device (and a fused optimizer)")
    lib, t, s, rest = _launch_stats(grads, max_norm)
    _launch_scale(lib, t, s, rest)                             # reads coef / found_inf on the device
    f32 = t.state.view(torch.float32)
    optimizer.grad_scale = None                                # (already applied by the scale pass)
    optimizer.found_inf = f32[6]                               # (0-dim view into the state block)


def _check_and_clip_host(grads, max_norm: float) -> bool:
    """CPU tensors only (gloo tests of the DP harness).  a gradient contains a non-finite value <=> its max-abs is non-finite;
    the clip coefficient uses the L2 norms combined in float64, redone on pre-scaled values if an fp32 sum of squares overflowed."""
    amax = torch.stack(torch._foreach_norm(grads, float("inf"))).max()
    norms = torch.stack(torch._foreach_norm(grads, 2.0)).double()
    total = torch.linalg.vector_norm(norms, 2.0)
    amax_h, total_h = torch.stack([amax.double(), total]).tolist()
    if not math.isfinite(amax_h):
        return False
    if not math.isfinite(total_h):
        scaled = torch._foreach_div(grads, amax)
        total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(scaled, 2.0)).double(), 2.0) * amax.double()
    coef = (max_norm / (total + 1e-6)).clamp(max=1.0).to(grads[0].dtype)
    torch._foreach_mul_(grads, coef)
    return True
