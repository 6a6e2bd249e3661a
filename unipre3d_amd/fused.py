"""Fused render-loss step (SURVEY.md N2 + N3): from the Gaussian head's Linear output to the scalar loss in one
launch sequence of the HIP library -- `u3d_render_loss_step` (training: forward AND backward in one sequence, behind the C++
autograd function of csrc/u3d_torch.cpp) or `u3d_render_loss_forward/_backward` (images kept / differentiable; ctypes, below).

Replaces, per training step: the activation/reshape chain of model/gaussian_predictor.py:279-328, the SH concat and
operator call of gaussian_renderer/__init__.py:78-97, the per-object/per-view loop and torch.stack of
train_network.py:418-446 and the elementwise loss of utils/loss_utils.py:17-45 -- about 60 PyTorch kernel launches
and ~1 GB of intermediate traffic per step at BASELINE config C2.
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import _lib
from .rasterizer import _C, _Plan, _f32c, _stream_ptr


class _RenderLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head_out, center, viewmatrix, projmatrix, campos, gt, bg, H, W, tanfov, mode, offset_scale, sh_degree,
                loss_kind, non_bg_rate, bg_rate, scale_modifier, flags, isotropic, item_offsets, max_P):
        lib = _lib.load()
        dev = head_out.device
        if dev.type != "cuda":
            raise RuntimeError("the MI355X rasterizer needs tensors on a HIP device; there is no CPU fallback")
        B, P, C, total_P = _batch_shape(head_out, item_offsets, max_P)
        NV = viewmatrix.shape[0]
        if B == 0 or NV % B != 0:
            raise ValueError(f"{NV} cameras for {B} Gaussian sets: every set needs the same number of views")
        V = NV // B
        K = (sh_degree + 1) ** 2
        if C != 11 + 3 * K:
            raise ValueError(f"head output has {C} channels, expected {11 + 3 * K} for SH degree {sh_degree}")
        plan = _Plan(B, V, P, H, W, tanfov, tanfov, scale_modifier, sh_degree, K, flags, total_P=total_P)
        desc = plan.with_offsets(item_offsets)
        hd = _lib.HeadDesc(mode, C, offset_scale, int(bool(isotropic)))
        ld = _lib.LossDesc(_lib.LOSS_KINDS[loss_kind], non_bg_rate, bg_rate)
        color = torch.empty((NV, 3, H, W), dtype=torch.float32, device=dev)
        radii = torch.zeros((V * total_P,) if total_P else (NV, P), dtype=torch.int32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
        geom, binning, image, fused = u8(plan.sizes.geom_bytes), u8(plan.sizes.binning_bytes), u8(plan.sizes.image_bytes), \
            u8(plan.sizes.fused_bytes)
        p = _lib.ptr
        rc = lib.u3d_render_loss_forward(ctypes.byref(desc), ctypes.byref(hd), ctypes.byref(ld), p(bg), p(head_out), p(center),
                                         p(viewmatrix), p(projmatrix), p(campos), p(gt), p(color), p(radii), p(loss), p(geom),
                                         p(binning), p(image), p(fused), _stream_ptr(dev))
        _lib.check(rc, "u3d_render_loss_forward")
        ctx.plan, ctx.hd, ctx.ld, ctx.desc, ctx.item_offsets = plan, hd, ld, desc, item_offsets
        ctx.save_for_backward(head_out, center, viewmatrix, projmatrix, campos, gt, bg, color, radii, geom, binning, image, fused)
        ctx.mark_non_differentiable(radii)     # `color` stays differentiable: an image-space term (LPIPS) may hang off it
        ctx.set_materialize_grads(False)
        return loss, color, radii

    @staticmethod
    def backward(ctx, grad_loss, grad_color, _gr):
        lib = _lib.load()
        head_out, center, viewmatrix, projmatrix, campos, gt, bg, color, radii, geom, binning, image, fused = ctx.saved_tensors
        dev = head_out.device
        if grad_loss is None and grad_color is None:
            return (torch.zeros_like(head_out),) + (None,) * 20
        d_head = torch.zeros_like(head_out) if ctx.item_offsets is not None else torch.empty_like(head_out)
        scratch = torch.empty(ctx.plan.sizes.backward_bytes, dtype=torch.uint8, device=dev)
        dloss = _f32c(grad_loss, dev).reshape(1) if grad_loss is not None else torch.zeros(1, dtype=torch.float32, device=dev)
        extra = None
        if grad_color is not None:             # dL/d(rendered) of a further image-space term of the objective (train_network.py:284-300)
            if grad_color.shape != color.shape:
                raise RuntimeError(f"gradient of the rendered images has shape {tuple(grad_color.shape)}, expected {tuple(color.shape)}")
            extra = _f32c(grad_color, dev)
        p = _lib.ptr
        rc = lib.u3d_render_loss_backward(ctypes.byref(ctx.desc), ctypes.byref(ctx.hd), ctypes.byref(ctx.ld), p(bg), p(head_out),
                                          p(center), p(viewmatrix), p(projmatrix), p(campos), p(gt), p(radii), p(color), p(dloss), p(extra),
                                          p(geom), p(binning), p(image), p(fused), p(scratch), p(d_head), _stream_ptr(dev))
        _lib.check(rc, "u3d_render_loss_backward")
        return (d_head,) + (None,) * 20


def _batch_shape(head_out, item_offsets, max_P):
    """(sets, Gaussians per set | largest set, channels, total Gaussians of a ragged batch | 0)."""
    if item_offsets is None:
        if head_out.dim() != 3:
            raise ValueError("head_out must be (B, P, C); pass item_offsets for a packed ragged batch (sum P_i, C)")
        B, P, C = head_out.shape
        return B, P, C, 0
    if head_out.dim() != 2:
        raise ValueError("ragged batch: head_out must be packed (sum P_i, C)")
    if item_offsets.dtype != torch.int32 or item_offsets.device != head_out.device or not item_offsets.is_contiguous():
        raise ValueError("item_offsets must be a contiguous int32 tensor on the Gaussians' device")
    total, C = head_out.shape
    if not max_P or max_P > total:
        raise ValueError("ragged batch: max_P (the largest set) is required")
    return item_offsets.numel() - 1, int(max_P), C, total


def backward_unit(loss: torch.Tensor) -> None:
    """`loss.backward()` for the loss of `render_loss_fused(single_pass=True)` when it is the quantity being minimised
    (dL/dloss = 1): seeds autograd with a cached read-only ones tensor, which the fused step recognises by its storage, so
    neither the `ones_like` fill nor the `d_head * 1` multiply is launched (two ~5 us kernels per step).  Any other use of the
    loss (scaled, summed with other terms) goes through `loss.backward()` as usual."""
    torch.autograd.backward(loss, grad_tensors=(_C().unit_tensor(loss.device.index if loss.device.index is not None else torch.cuda.current_device()),))


def render_loss_fused(head_out: torch.Tensor, center: torch.Tensor, world_view: torch.Tensor, full_proj: torch.Tensor,
                      camera_center: torch.Tensor, gt: torch.Tensor, bg: torch.Tensor, fov_deg: float, H: int, W: int,
                      level: str = "object", offset_scale: float = 1.0, max_sh_degree: int = 1, loss_kind: str = "focal_l2",
                      non_bg_color_loss_rate: float = 4.0, bg_color_loss_rate: float = 1.0, input_images: int = 0,
                      scaling_modifier: float = 1.0, antialiasing: bool = True, debug: bool = False,
                      single_pass: bool = True, return_images: bool = True, differentiable_images: bool = False,
                      isotropic: bool = False, item_offsets: torch.Tensor = None, max_P: int = 0, sizes=None):
    """head_out (B,P,C) point-major raw head output (C = 23 at SH degree 1), center (B,P,3), cameras (B,Vtot,...),
    gt (B,Vtot,3,H,W).  Returns (loss scalar, rendered (B*V',3,H,W), radii (B*V',P)).
    single_pass (default): when a gradient is wanted, forward and backward run as ONE launch sequence
    (u3d_render_loss_step) and autograd's backward only scales the stored gradient; the rendered images it returns are
    DETACHED (requires_grad False); return_images=False additionally skips writing them (the training loop only needs the loss).
    differentiable_images=True: the objective has a further image-space term -- the reference's
    `l12 + lambda_lpips * lpips(rendered, gt)` after `start_lpips_after` iterations (train_network.py:284-300).  The call then
    takes the two-pass route, the returned images carry gradient, and `(loss + lam * g(rendered)).backward()` adds
    dL/d(rendered) to the in-kernel loss seed (u3d_render_loss_backward's dL_dcolor_extra): still one launch sequence each way.
    isotropic: cfg.model.isotropic (the first scaling channel serves all three axes, model/gaussian_predictor.py:308-310).
    Ragged batches (the scene-level branch's per-item lists, model/gaussian_predictor.py:331-364): head_out (sum P_i, C) and center
    (sum P_i, 3) PACKED in set order and `sizes` = the sets' sizes on the host (rasterizer.pack_ragged returns them; the prefix sums
    and the largest set are derived here, so they cannot be inconsistent); radii then come back packed (V' * sum P_i,) -- one launch
    sequence for sets of different sizes.  (Expert form: item_offsets int32 (B+1,) on the device + max_P, trusted by the kernels and
    checked on the device only with debug=True.)"""
    dev = head_out.device
    if sizes is not None:
        if item_offsets is not None:
            raise ValueError("pass either sizes or item_offsets / max_P, not both")
        if head_out.dim() != 2 or sum(int(n) for n in sizes) != head_out.shape[0]:
            raise ValueError(f"ragged batch: head_out must be packed (sum P_i, C) with sum P_i = {sum(sizes)}")
        from .rasterizer import ragged_layout
        item_offsets, max_P = ragged_layout(sizes, dev)
    B = head_out.shape[0] if item_offsets is None else item_offsets.numel() - 1
    wv, fp, cc = world_view[:, input_images:], full_proj[:, input_images:], camera_center[:, input_images:]
    NV = B * wv.shape[1]
    t = math.tan(fov_deg * math.pi / 360)
    flags = (_lib.FLAG_ANTIALIASING if antialiasing else 0) | (_lib.FLAG_DEBUG if debug else 0)
    f = lambda x: _f32c(x, dev)
    if differentiable_images and not return_images:
        raise ValueError("differentiable_images=True needs return_images=True")
    K = (int(max_sh_degree) + 1) ** 2
    if head_out.shape[-1] != 11 + 3 * K:
        raise ValueError(f"head output has {head_out.shape[-1]} channels, expected {11 + 3 * K} for SH degree {max_sh_degree}")
    _batch_shape(head_out, item_offsets, max_P)          # (shape / dtype errors of a ragged batch as ValueError)
    if B == 0 or NV % B != 0:
        raise ValueError(f"{NV} cameras for {B} Gaussian sets: every set needs the same number of views")
    if single_pass and not differentiable_images and head_out.requires_grad and torch.is_grad_enabled():
        # training form: ONE launch sequence computes the loss and d loss / d head_out (u3d_render_loss_step) behind a C++ autograd
        # function (csrc/u3d_torch.cpp) whose backward only scales the stored gradient; it keeps the backward scratch per stream and
        # shape and tells the library that the accumulators are still zero (U3D_FLAG_ACC_CLEAN)
        return _C().render_loss_step(head_out, center, wv, fp, cc, gt[:, input_images:].reshape(NV, 3, H, W), bg, int(H), int(W), float(t),
                                     1 if level == "object" else 2, float(offset_scale), int(max_sh_degree), _lib.LOSS_KINDS[loss_kind],
                                     float(non_bg_color_loss_rate), float(bg_color_loss_rate), float(scaling_modifier), flags,
                                     bool(return_images), bool(isotropic), item_offsets, int(max_P))
    loss, img, radii = _RenderLossFn.apply(f(head_out), f(center), f(wv).reshape(NV, 16), f(fp).reshape(NV, 16), f(cc).reshape(NV, 3),
                                           f(gt[:, input_images:]).reshape(NV, 3, H, W), f(bg).reshape(3), int(H), int(W), float(t),
                                           1 if level == "object" else 2, float(offset_scale), int(max_sh_degree), loss_kind,
                                           float(non_bg_color_loss_rate), float(bg_color_loss_rate), float(scaling_modifier), flags,
                                           bool(isotropic), item_offsets, int(max_P))
    return loss, (img if differentiable_images else img.detach()), radii
