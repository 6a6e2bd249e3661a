"""Point sampling / grouping operators with the reference's Python names and signatures:
openpoints/models/layers/subsample.py:77-148 (`furthest_point_sample`, `gather_operation`, `fps`) and
openpoints/models/layers/group.py:76-203 (`grouping_operation`, `ball_query`), the `QueryAndGroup` module
(:208-260), and openpoints/models/layers/upsampling.py:11-101 (`three_nn`, `three_interpolate`, `three_interpolation`).  Backed by libunipre3d_pointops.so (include/unipre3d_pointops.h); no CPU fallback."""
from __future__ import annotations

import ctypes
import os
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib

LIB_PATH = os.path.join(_lib.LIB_DIR, "libunipre3d_pointops.so")   # (U3D_LIB_DIRNAME: experiment builds, see _lib.py)
EXPORTS = ("u3d_furthest_point_sampling", "u3d_ball_query", "u3d_group_points", "u3d_group_points_grad",
           "u3d_gather_points", "u3d_gather_points_grad", "u3d_three_nn", "u3d_three_interpolate", "u3d_three_interpolate_grad",
           "u3d_pointops_set_contraction", "u3d_pointops_get_contraction")
CONTRACTIONS = {"fma_llvm": 0, "fma_chain": 1, "none": 2}    # include/unipre3d_pointops.h: U3D_PO_*


def set_contraction(mode: str) -> None:
    """How the reference's three-term sums (squared distances, interpolation) are contracted -- what nvcc's -fmad=true would have
    chosen decides FPS ties.  'fma_llvm' (default: LLVM's combiner order, NVVM is LLVM), 'fma_chain' (rounds 1-3 of this build),
    'none' (-fmad=false).  Process-wide, applies to the launches that follow."""
    rc = load().u3d_pointops_set_contraction(CONTRACTIONS[mode])
    if rc != 0:
        raise ValueError(f"unknown contraction mode {mode!r}")
_po = None


def load() -> ctypes.CDLL:
    global _po
    if _po is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing (no fallback): run `make -C unipre3d_amd/csrc`")
        lib = ctypes.CDLL(LIB_PATH)
        vp, i, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        lib.u3d_furthest_point_sampling.argtypes = [i, i, i, vp, vp, vp, vp]
        lib.u3d_ball_query.argtypes = [i, i, i, f, i, vp, vp, vp, vp]
        lib.u3d_group_points.argtypes = [i, i, i, i, i, vp, vp, vp, vp]
        lib.u3d_group_points_grad.argtypes = [i, i, i, i, i, vp, vp, vp, vp]
        lib.u3d_gather_points.argtypes = [i, i, i, i, vp, vp, vp, vp]
        lib.u3d_gather_points_grad.argtypes = [i, i, i, i, vp, vp, vp, vp]
        lib.u3d_three_nn.argtypes = [i, i, i, vp, vp, vp, vp, vp]
        lib.u3d_three_interpolate.argtypes = [i, i, i, i, vp, vp, vp, vp, vp]
        lib.u3d_three_interpolate_grad.argtypes = [i, i, i, i, vp, vp, vp, vp, vp]
        lib.u3d_pointops_set_contraction.argtypes = [i]
        lib.u3d_pointops_get_contraction.argtypes = []
        for n in EXPORTS:
            getattr(lib, n).restype = ctypes.c_int
        _po = lib
    return _po


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


def _stream(dev=None):
    """torch's current stream on the CURRENT device; tensors living on another device are refused (rasterizer._stream_ptr)."""
    from .rasterizer import _stream_ptr
    return _stream_ptr(dev)


def _need_gpu(*tensors):
    dev = tensors[0].device
    for t in tensors:
        if t.device.type != "cuda":
            raise RuntimeError("unipre3d_amd.pointops needs tensors on a HIP device; there is no CPU fallback")
        if t.device != dev:
            raise RuntimeError(f"unipre3d_amd.pointops: tensors on different devices ({dev} and {t.device})")
    return dev


def _f32(t, what):
    """The kernels read raw fp32: anything else (fp16/bf16 under autocast, float64) is cast, never reinterpreted."""
    if not t.is_floating_point():
        raise TypeError(f"{what} must be a floating-point tensor, got {t.dtype}")
    return t if t.dtype == torch.float32 else t.float()


def _i32(idx, what):
    """Index tensors are int32 in the C-ABI (what furthest_point_sample / ball_query return); int64 ones (.long(), torch.topk)
    are converted, anything else is refused."""
    if idx.dtype == torch.int32:
        return idx
    if idx.dtype == torch.int64:
        return idx.to(torch.int32)
    raise TypeError(f"{what} must be an int32 (or int64) index tensor, got {idx.dtype}")


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        """xyz (B,N,3) -> (B,npoint) int32 indices, starting from index 0 (subsample.py:77-100)."""
        assert xyz.is_contiguous()
        dev = _need_gpu(xyz)
        xyz = _f32(xyz, "xyz")
        B, N, _ = xyz.size()
        out = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
        temp = torch.empty(B, N, dtype=torch.float32, device=xyz.device) if N > 8192 else None
        _check(load().u3d_furthest_point_sampling(B, N, npoint, _lib.ptr(xyz), _lib.ptr(temp), _lib.ptr(out), _stream(dev)), "fps")
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint) -> (B,C,npoint) (subsample.py:110-131)."""
        assert features.is_contiguous() and idx.is_contiguous()
        dev = _need_gpu(features, idx)
        features, idx = _f32(features, "features"), _i32(idx, "idx")
        B, npoint = idx.size()
        _, C, N = features.size()
        out = torch.empty(B, C, npoint, dtype=torch.float32, device=features.device)
        _check(load().u3d_gather_points(B, C, N, npoint, _lib.ptr(features), _lib.ptr(idx), _lib.ptr(out), _stream(dev)), "gather")
        ctx.for_backwards = (idx, C, N)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)
        g = _f32(grad_out, "grad_out").contiguous()
        _check(load().u3d_gather_points_grad(B, C, N, npoint, _lib.ptr(g), _lib.ptr(idx), _lib.ptr(grad), _stream(_need_gpu(g, idx))),
               "gather grad")
        return grad, None


gather_operation = GatherOperation.apply


def fps(data: torch.Tensor, number: int) -> torch.Tensor:
    """data (B,N,C) -> the `number` furthest-point-sampled rows (subsample.py:151-160)."""
    idx = furthest_point_sample(data[:, :, :3].contiguous(), number)
    return torch.gather(data, 1, idx.unsqueeze(-1).long().expand(-1, -1, data.shape[-1]))


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample) (group.py:76-99)."""
        assert features.is_contiguous() and idx.is_contiguous()
        dev = _need_gpu(features, idx)
        features, idx = _f32(features, "features"), _i32(idx, "idx")
        B, npoint, nsample = idx.size()
        _, C, N = features.size()
        out = torch.empty(B, C, npoint, nsample, dtype=torch.float32, device=features.device)
        _check(load().u3d_group_points(B, C, N, npoint, nsample, _lib.ptr(features), _lib.ptr(idx), _lib.ptr(out), _stream(dev)),
               "group")
        ctx.for_backwards = (idx, N)
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor) -> Tuple[torch.Tensor, None]:
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)
        g = _f32(grad_out, "grad_out").contiguous()
        _check(load().u3d_group_points_grad(B, C, N, npoint, nsample, _lib.ptr(g), _lib.ptr(idx), _lib.ptr(grad),
                                            _stream(_need_gpu(g, idx))), "group grad")
        return grad, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        """xyz (B,N,3) support, new_xyz (B,npoint,3) centres -> (B,npoint,nsample) int32 (group.py:175-196)."""
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        dev = _need_gpu(xyz, new_xyz)
        xyz, new_xyz = _f32(xyz, "xyz"), _f32(new_xyz, "new_xyz")
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.empty(B, npoint, nsample, dtype=torch.int32, device=xyz.device)
        _check(load().u3d_ball_query(B, N, npoint, float(radius), nsample, _lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(idx), _stream(dev)),
               "ball query")
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """unknown (B,N,3), known (B,M,3) -> (dist (B,N,3) L2 distances to the three nearest known points, idx (B,N,3) int32)
        (upsampling.py:11-35; the kernel returns squared distances, the wrapper takes the root like the reference)."""
        assert unknown.is_contiguous() and known.is_contiguous()
        dev = _need_gpu(unknown, known)
        unknown, known = _f32(unknown, "unknown"), _f32(known, "known")
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
        idx = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
        _check(load().u3d_three_nn(B, N, m, _lib.ptr(unknown), _lib.ptr(known), _lib.ptr(dist2), _lib.ptr(idx), _stream(dev)), "three_nn")
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """features (B,C,M), idx (B,n,3), weight (B,n,3) -> (B,C,n) (upsampling.py:43-67; inputs are cast to fp32 like the
        reference's custom_fwd(cast_inputs=torch.float32))."""
        assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        dev = _need_gpu(features, idx, weight)
        features, idx, weight = _f32(features, "features"), _i32(idx, "idx"), _f32(weight, "weight")
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        out = torch.empty(B, c, n, dtype=torch.float32, device=dev)
        _check(load().u3d_three_interpolate(B, c, m, n, _lib.ptr(features), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(out), _stream(dev)),
               "three_interpolate")
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.zeros(B, c, m, dtype=torch.float32, device=grad_out.device)
        g = _f32(grad_out, "grad_out").contiguous()
        _check(load().u3d_three_interpolate_grad(B, c, n, m, _lib.ptr(g), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(grad_features),
                                                 _stream(_need_gpu(g, idx, weight))), "three_interpolate grad")
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


def three_interpolation(unknown_xyz, known_xyz, know_feat):
    """Inverse-distance interpolation of `know_feat` (B,C,M) at `unknown_xyz` (B,N,3) (upsampling.py:92-101)."""
    dist, idx = three_nn(unknown_xyz, known_xyz)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=2, keepdim=True)
    weight = dist_recip / norm
    return three_interpolate(know_feat, idx, weight)


class QueryAndGroup(nn.Module):
    """group.py:208-260 (relative_xyz / normalize_dp / return_only_idx options)."""

    def __init__(self, radius: float, nsample: int, relative_xyz=True, normalize_dp=False, return_only_idx=False, **kwargs):
        super().__init__()
        self.radius, self.nsample = radius, nsample
        self.relative_xyz, self.normalize_dp, self.return_only_idx = relative_xyz, normalize_dp, return_only_idx

    def forward(self, query_xyz: torch.Tensor, support_xyz: torch.Tensor, features: torch.Tensor = None):
        idx = ball_query(self.radius, self.nsample, support_xyz, query_xyz)
        if self.return_only_idx:
            return idx
        grouped_xyz = grouping_operation(support_xyz.transpose(1, 2).contiguous(), idx)
        if self.relative_xyz:
            grouped_xyz = grouped_xyz - query_xyz.transpose(1, 2).unsqueeze(-1)
            if self.normalize_dp:
                grouped_xyz = grouped_xyz / self.radius
        grouped_features = grouping_operation(features, idx) if features is not None else None
        return grouped_xyz, grouped_features
