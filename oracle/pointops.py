"""ctypes front-end of oracle/pointops_oracle.c (TEST INFRASTRUCTURE ONLY; see the C file's header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libpointops_oracle.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH) or os.path.getmtime(_PATH) < os.path.getmtime(os.path.join(_HERE, "pointops_oracle.c")):
            subprocess.run(["make", "-C", _HERE, "-s"], check=True, stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_PATH)
        _lib.po_opt_n_threads.restype = ctypes.c_int
    return _lib


CONTRACTIONS = {"fma_llvm": 0, "fma_chain": 1, "none": 2}


def set_contraction(mode):
    """'fma_llvm' (default) | 'fma_chain' | 'none' -- see include/unipre3d_pointops.h."""
    _load().po_set_contraction(CONTRACTIONS[mode] if isinstance(mode, str) else int(mode))


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def opt_n_threads(n):
    return int(_load().po_opt_n_threads(int(n)))


def furthest_point_sampling(xyz, m):
    xyz = _f(xyz); B, N, _ = xyz.shape
    temp = np.full((B, N), 1e10, np.float32)           # openpoints/models/layers/subsample.py:93
    idx = np.zeros((B, m), np.int32)
    _load().po_furthest_point_sampling(B, N, int(m), _p(xyz), _p(temp), _p(idx))
    return idx


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f(xyz), _f(new_xyz); B, N, _ = xyz.shape; M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), np.int32)           # zero_() at openpoints/models/layers/group.py:192
    _load().po_ball_query(B, N, M, ctypes.c_float(radius), int(nsample), _p(new_xyz), _p(xyz), _p(idx))
    return idx


def group_points(points, idx):
    points, idx = _f(points), _i(idx); B, C, N = points.shape; _, M, K = idx.shape
    out = np.zeros((B, C, M, K), np.float32)
    _load().po_group_points(B, C, N, M, K, _p(points), _p(idx), _p(out))
    return out


def group_points_grad(grad_out, idx, N):
    grad_out, idx = _f(grad_out), _i(idx); B, C, M, K = grad_out.shape
    g = np.zeros((B, C, N), np.float32)
    _load().po_group_points_grad(B, C, int(N), M, K, _p(grad_out), _p(idx), _p(g))
    return g


def gather_points(points, idx):
    points, idx = _f(points), _i(idx); B, C, N = points.shape; M = idx.shape[1]
    out = np.zeros((B, C, M), np.float32)
    _load().po_gather_points(B, C, N, M, _p(points), _p(idx), _p(out))
    return out


def gather_points_grad(grad_out, idx, N):
    grad_out, idx = _f(grad_out), _i(idx); B, C, M = grad_out.shape
    g = np.zeros((B, C, N), np.float32)
    _load().po_gather_points_grad(B, C, int(N), M, _p(grad_out), _p(idx), _p(g))
    return g


def three_nn(unknown, known):
    """-> (dist2 (B,N,3) float32 SQUARED distances as the kernel writes them, idx (B,N,3) int32)."""
    unknown, known = _f(unknown), _f(known); B, N, _ = unknown.shape; M = known.shape[1]
    d2 = np.zeros((B, N, 3), np.float32); idx = np.zeros((B, N, 3), np.int32)
    _load().po_three_nn(B, N, M, _p(unknown), _p(known), _p(d2), _p(idx))
    return d2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f(points), _i(idx), _f(weight); B, C, M = points.shape; N = idx.shape[1]
    out = np.zeros((B, C, N), np.float32)
    _load().po_three_interpolate(B, C, M, N, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, M):
    grad_out, idx, weight = _f(grad_out), _i(idx), _f(weight); B, C, N = grad_out.shape
    g = np.zeros((B, C, int(M)), np.float32)
    _load().po_three_interpolate_grad(B, C, N, int(M), _p(grad_out), _p(idx), _p(weight), _p(g))
    return g
