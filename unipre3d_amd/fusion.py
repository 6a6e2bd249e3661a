"""Object-level 2D->3D feature fusion: `FeatureFusion` with the reference's interface
(fusion/feat_fusion.py:5-145), the projection / z-buffer / gather chain running in libunipre3d_fusion.so
(include/unipre3d_fusion.h) without the reference's two host synchronisations."""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn

from . import _lib

LIB_PATH = os.path.join(_lib.LIB_DIR, "libunipre3d_fusion.so")   # (U3D_LIB_DIRNAME: experiment builds, see _lib.py)
EXPORTS = ("u3d_zbuffer_fusion_forward", "u3d_zbuffer_fusion_backward", "u3d_fusion_abi_version", "u3d_zbuffer_fusion_zbuf_bytes")
ABI_VERSION = 2
_fu = None


def load() -> ctypes.CDLL:
    global _fu
    if _fu is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing (no fallback): run `make -C unipre3d_amd/csrc`")
        lib = ctypes.CDLL(LIB_PATH)
        vp, i, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        lib.u3d_zbuffer_fusion_forward.argtypes = [i, i, i, i, i, f, f, f, f, vp, vp, vp, vp, vp, vp]
        lib.u3d_zbuffer_fusion_backward.argtypes = [i, i, i, i, i, vp, vp, vp, vp, vp]
        lib.u3d_fusion_abi_version.argtypes = []
        lib.u3d_zbuffer_fusion_zbuf_bytes.argtypes = [i, i, i]
        for n in EXPORTS:
            getattr(lib, n).restype = ctypes.c_int
        lib.u3d_zbuffer_fusion_zbuf_bytes.restype = ctypes.c_size_t
        if lib.u3d_fusion_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH}: ABI {lib.u3d_fusion_abi_version()}, this module binds ABI {ABI_VERSION}: rebuild (`make -C unipre3d_amd/csrc`)")
        _fu = lib
    return _fu


class _ZBufferGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, camera_points, image_features, fx, fy, cx, cy):
        if camera_points.device.type != "cuda":
            raise RuntimeError("unipre3d_amd.fusion needs tensors on a HIP device; there is no CPU fallback")
        if image_features.device != camera_points.device:
            raise RuntimeError(f"unipre3d_amd.fusion: tensors on different devices ({camera_points.device}, {image_features.device})")
        from .rasterizer import _stream_ptr
        B, N, _ = camera_points.shape
        _, C, H, W = image_features.shape
        cp, feat = camera_points.contiguous().float(), image_features.contiguous().float()
        mapped = torch.empty(B, N, C, dtype=torch.float32, device=cp.device)
        sel = torch.empty(B, N, dtype=torch.int32, device=cp.device)
        # winner table: (depth bits << 32 | first winner) per pixel + the "empty pixel" bitmap behind it
        zbuf = torch.empty((load().u3d_zbuffer_fusion_zbuf_bytes(B, H, W) + 7) // 8, dtype=torch.int64, device=cp.device)
        rc = load().u3d_zbuffer_fusion_forward(B, N, C, H, W, fx, fy, cx, cy, _lib.ptr(cp), _lib.ptr(feat), _lib.ptr(mapped),
                                               _lib.ptr(sel), _lib.ptr(zbuf), _stream_ptr(cp.device))
        if rc != 0:
            raise RuntimeError(f"u3d_zbuffer_fusion_forward failed with code {rc}")
        ctx.save_for_backward(sel, zbuf)
        ctx.shape = (B, N, C, H, W)
        ctx.mark_non_differentiable(sel)
        return mapped, sel

    @staticmethod
    def backward(ctx, grad_mapped, _gs):
        sel, zbuf = ctx.saved_tensors
        B, N, C, H, W = ctx.shape
        # gather form: the kernel writes every element of the (B,C,H,W) gradient exactly once (no zero-fill + scatter)
        grad_feat = torch.empty(B, C, H, W, dtype=torch.float32, device=sel.device)
        g = grad_mapped.contiguous().float()
        from .rasterizer import _stream_ptr
        rc = load().u3d_zbuffer_fusion_backward(B, N, C, H, W, _lib.ptr(g), _lib.ptr(sel), _lib.ptr(zbuf), _lib.ptr(grad_feat), _stream_ptr(sel.device))
        if rc != 0:
            raise RuntimeError(f"u3d_zbuffer_fusion_backward failed with code {rc}")
        return None, grad_feat, None, None, None, None


class FeatureFusion:
    """Same constructor / call signature as the reference's class (fusion/feat_fusion.py:14-22, 58-145)."""

    def __init__(self, fusion_mlp: nn.Module):
        self.fusion_mlp = fusion_mlp

    @staticmethod
    def camera_points(center: torch.Tensor, c2w_matrix: torch.Tensor) -> torch.Tensor:
        """World -> camera exactly as fusion/feat_fusion.py:37-45 (kept in PyTorch: bit-identical pixel rounding)."""
        ones = torch.ones([*center.shape[:2], 1], device=center.device, dtype=center.dtype)
        w2c = torch.linalg.inv(c2w_matrix.permute(0, 2, 1))
        return torch.matmul(w2c, torch.cat([center, ones], dim=2).transpose(1, 2)).transpose(1, 2)

    def mapped_features(self, center, image_features, c2w_projection_matrix, intrinsic):
        """(B,N,C) image features of the points that win their pixel's z-test, zeros elsewhere (:86-131)."""
        if c2w_projection_matrix.dim() == 4:
            c2w_projection_matrix = c2w_projection_matrix[:, 0]
        cp = self.camera_points(center, c2w_projection_matrix)
        fx, fy = float(intrinsic[0][0]), float(intrinsic[1][1])
        cx, cy = float(intrinsic[0][2]), float(intrinsic[1][2])
        mapped, _ = _ZBufferGather.apply(cp, image_features, fx, fy, cx, cy)
        return mapped

    def __call__(self, x, center, image_features, c2w_projection_matrix, intrinsic):
        B, N = center.shape[:2]
        C = image_features.shape[1]
        mapped = self.mapped_features(center, image_features, c2w_projection_matrix, intrinsic)
        if x.shape[1] > N:   # transformer CLS token (:134-140)
            x_patch = torch.cat([x[:, 1:], mapped], dim=-1)
            cls = torch.cat([x[:, 0:1], torch.zeros((B, 1, C), device=center.device)], dim=-1)
            x = torch.cat([cls, x_patch], dim=1)
        else:
            x = torch.cat([x, mapped], dim=-1)
        return self.fusion_mlp(x)
