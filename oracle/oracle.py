"""ctypes front-end of the CPU oracle (oracle/raster_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  Nothing under unipre3d_amd/ imports this module.

PARITY UNPINNED (see the header of raster_oracle.c and SURVEY.md section 0): the reference
holds neither the rasterizer's source nor golden vectors for it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

FLAG_PREFILTERED = 1
FLAG_ANTIALIASING = 2
FLAG_DEBUG = 4
FLAG_EXACT_AA_GRAD = 8


def build(force: bool = False) -> None:
    """Compile the two oracle libraries with gcc (seconds)."""
    src = os.path.join(_HERE, "raster_oracle.c")
    outs = [os.path.join(_BUILD, "liboracle_f32.so"), os.path.join(_BUILD, "liboracle_f64.so")]
    fresh = all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(src) for o in outs)
    if fresh and not force:
        return
    subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                   stdout=subprocess.DEVNULL)


_libs = {}


def _lib(dtype):
    dtype = np.dtype(dtype)
    if dtype not in _libs:
        name = "liboracle_f32.so" if dtype == np.float32 else "liboracle_f64.so"
        path = os.path.join(_BUILD, name)
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        pre = "orf_" if dtype == np.float32 else "ord_"
        real = ctypes.c_float if dtype == np.float32 else ctypes.c_double
        vp = ctypes.c_void_p
        fwd = getattr(lib, pre + "forward")
        fwd.restype = vp
        fwd.argtypes = [ctypes.c_int] * 3 + [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, real, vp, vp,
                                             vp, vp, vp, real, real, ctypes.c_int, vp, vp, vp]
        fwd_d = getattr(lib, pre + "forward_discrete")
        fwd_d.restype = vp
        fwd_d.argtypes = list(fwd.argtypes) + [vp, vp, vp]
        getattr(lib, pre + "state_discrete").argtypes = [vp, vp, vp]
        getattr(lib, pre + "state_discrete").restype = None
        bwd = getattr(lib, pre + "backward")
        bwd.restype = None
        bwd.argtypes = [vp] * 11
        getattr(lib, pre + "free").argtypes = [vp]
        getattr(lib, pre + "free").restype = None
        getattr(lib, pre + "num_rendered").argtypes = [vp]
        getattr(lib, pre + "num_rendered").restype = ctypes.c_int64
        for nm in ("depths", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched", "final_T",
                   "n_contrib", "point_list", "ranges"):
            f = getattr(lib, pre + "state_" + nm)
            f.argtypes = [vp]
            f.restype = vp
        getattr(lib, pre + "mark_visible").argtypes = [ctypes.c_int, vp, vp, vp, vp]
        getattr(lib, pre + "mark_visible").restype = None
        getattr(lib, pre + "num_threads").restype = ctypes.c_int
        getattr(lib, pre + "set_num_threads").argtypes = [ctypes.c_int]
        getattr(lib, pre + "set_num_threads").restype = None
        _libs[dtype] = (lib, pre, real)
    return _libs[dtype]


def num_threads() -> int:
    lib, pre, _ = _lib(np.float32)
    return int(getattr(lib, pre + "num_threads")())


def set_num_threads(n: int) -> None:
    """OpenMP threads of the CALLING thread's next oracle calls (both precisions)."""
    for dt in (np.float32, np.float64):
        lib, pre, _ = _lib(dt)
        getattr(lib, pre + "set_num_threads")(int(n))


def _arr(x, dtype, shape=None):
    if x is None:
        return None
    a = np.ascontiguousarray(np.asarray(x, dtype=dtype))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


@dataclass
class OracleRender:
    color: np.ndarray          # (3,H,W)
    invdepth: np.ndarray       # (1,H,W)
    radii: np.ndarray          # (P,) int32
    num_rendered: int
    dtype: np.dtype
    P: int
    M: int
    H: int
    W: int
    _state: Optional[int] = field(default=None, repr=False)

    def _get(self, name, n, dt):
        lib, pre, _ = _lib(self.dtype)
        p = getattr(lib, pre + "state_" + name)(ctypes.c_void_p(self._state))
        if n == 0:
            return np.zeros((0,), dtype=dt)
        buf = (ctypes.c_char * (n * np.dtype(dt).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dt).copy()

    # per-stage state for stage-level parity tests
    @property
    def depths(self): return self._get("depths", self.P, self.dtype)
    @property
    def means2D(self): return self._get("means2D", self.P * 2, self.dtype).reshape(self.P, 2)
    @property
    def cov3D(self): return self._get("cov3D", self.P * 6, self.dtype).reshape(self.P, 6)
    @property
    def conic_opacity(self): return self._get("conic_opacity", self.P * 4, self.dtype).reshape(self.P, 4)
    @property
    def rgb(self): return self._get("rgb", self.P * 3, self.dtype).reshape(self.P, 3)
    @property
    def tiles_touched(self): return self._get("tiles_touched", self.P, np.uint32)
    @property
    def final_T(self): return self._get("final_T", self.H * self.W, self.dtype).reshape(self.H, self.W)
    @property
    def n_contrib(self): return self._get("n_contrib", self.H * self.W, np.uint32).reshape(self.H, self.W)
    @property
    def point_list(self): return self._get("point_list", self.num_rendered, np.uint32)
    @property
    def ranges(self):
        nt = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        return self._get("ranges", nt * 2, np.uint32).reshape(nt, 2)

    def close(self):
        if self._state:
            lib, pre, _ = _lib(self.dtype)
            getattr(lib, pre + "free")(ctypes.c_void_p(self._state))
            self._state = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def forward(means3D, opacities, viewmatrix, projmatrix, campos, bg, image_height, image_width, tanfovx, tanfovy,
            shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0,
            scale_modifier=1.0, antialiasing=True, prefiltered=False, exact_aa_grad=False,
            dtype=np.float32, discrete_from: Optional["OracleRender"] = None) -> OracleRender:
    """CPU restatement of `GaussianRasterizer.forward` (call site gaussian_renderer/__init__.py:89-97).

    discrete_from: an OracleRender of the SAME inputs (normally the fp32 one) whose discrete per-Gaussian decisions -- survival of the
    culls, integer radius, tile rectangle, and the fp32 depth that is the sort key (ties -> index) -- this evaluation takes over while
    computing every continuous quantity in its own dtype.  fp64 + discrete_from=r32 is the parity rule's arbiter: the reference
    operator is fp32, so those integers ARE the reference's semantics (raster_oracle.c: forward_impl)."""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    dtype = np.dtype(dtype)
    lib, pre, real = _lib(dtype)
    means3D = _arr(means3D, dtype)
    P = means3D.shape[0] if means3D.ndim == 2 else 0
    means3D = means3D.reshape(P, 3)
    M = 0
    if shs is not None:
        shs = _arr(shs, dtype)
        M = shs.shape[1] if P > 0 else (shs.shape[1] if shs.ndim == 3 else 0)
        shs = shs.reshape(P, M, 3)
        assert M >= (sh_degree + 1) ** 2
    colors_precomp = _arr(colors_precomp, dtype, (P, 3))
    opacities = _arr(opacities, dtype, (P,))
    scales = _arr(scales, dtype, (P, 3))
    rotations = _arr(rotations, dtype, (P, 4))
    cov3D_precomp = _arr(cov3D_precomp, dtype, (P, 6))
    view = _arr(viewmatrix, dtype, (16,))
    proj = _arr(projmatrix, dtype, (16,))
    cam = _arr(campos, dtype, (3,))
    bgc = _arr(bg, dtype, (3,))
    H, W = int(image_height), int(image_width)
    color = np.zeros((3, H, W), dtype=dtype)
    invd = np.zeros((1, H, W), dtype=dtype)
    radii = np.zeros((max(P, 1),), dtype=np.int32)
    flags = (FLAG_ANTIALIASING if antialiasing else 0) | (FLAG_PREFILTERED if prefiltered else 0) | \
            (FLAG_EXACT_AA_GRAD if exact_aa_grad else 0)
    common = (P, int(sh_degree), M, _ptr(bgc), W, H, _ptr(means3D), _ptr(shs), _ptr(colors_precomp), _ptr(opacities), _ptr(scales),
              real(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(view), _ptr(proj), _ptr(cam), real(tanfovx), real(tanfovy),
              flags, _ptr(color), _ptr(invd), _ptr(radii))
    own = None
    if isinstance(discrete_from, str):
        # "fp32": run the fp32 restatement on the same inputs (rounded to fp32) for the discrete decisions
        assert discrete_from == "fp32", discrete_from
        f32 = lambda x: None if x is None else np.asarray(x, dtype=np.float32)
        own = discrete_from = forward(f32(means3D), f32(opacities), f32(view), f32(proj), f32(cam), f32(bgc), H, W, tanfovx, tanfovy,
                                      shs=f32(shs), colors_precomp=f32(colors_precomp), scales=f32(scales), rotations=f32(rotations),
                                      cov3D_precomp=f32(cov3D_precomp), sh_degree=sh_degree, scale_modifier=scale_modifier,
                                      antialiasing=antialiasing, prefiltered=prefiltered, exact_aa_grad=exact_aa_grad, dtype=np.float32)
    if discrete_from is None:
        st = getattr(lib, pre + "forward")(*common)
    else:
        d = discrete_from
        assert d.P == P and d.H == H and d.W == W and d._state, "discrete_from: an open OracleRender of the same inputs"
        dl, dpre, _ = _lib(d.dtype)
        rect = np.zeros((max(P, 1), 4), np.int32)
        depth = np.zeros((max(P, 1),), np.float32)
        getattr(dl, dpre + "state_discrete")(ctypes.c_void_p(d._state), _ptr(rect), _ptr(depth))
        drad = np.ascontiguousarray(np.concatenate([d.radii, np.zeros(max(P, 1) - P, np.int32)]).astype(np.int32))
        st = getattr(lib, pre + "forward_discrete")(*common, _ptr(drad), _ptr(rect), _ptr(depth))
        if own is not None:
            own.close()
    nr = int(getattr(lib, pre + "num_rendered")(ctypes.c_void_p(st)))
    return OracleRender(color=color, invdepth=invd, radii=radii[:P], num_rendered=nr, dtype=dtype, P=P, M=M,
                        H=H, W=W, _state=st)


def backward(r: OracleRender, dL_dcolor, dL_dinvdepth=None) -> dict:
    """CPU restatement of the operator's backward."""
    lib, pre, _ = _lib(r.dtype)
    dt = r.dtype
    P, M = r.P, r.M
    dpix = _arr(dL_dcolor, dt, (3, r.H, r.W))
    dinv = _arr(dL_dinvdepth, dt, (r.H, r.W)) if dL_dinvdepth is not None else None
    n = max(P, 1)
    out = {
        "means3D": np.zeros((n, 3), dt), "means2D": np.zeros((n, 3), dt), "shs": np.zeros((n, max(M, 1), 3), dt),
        "colors_precomp": np.zeros((n, 3), dt), "opacities": np.zeros((n,), dt), "scales": np.zeros((n, 3), dt),
        "rotations": np.zeros((n, 4), dt), "cov3D_precomp": np.zeros((n, 6), dt),
    }
    getattr(lib, pre + "backward")(ctypes.c_void_p(r._state), _ptr(dpix), _ptr(dinv), _ptr(out["means3D"]),
                                   _ptr(out["means2D"]), _ptr(out["shs"]), _ptr(out["colors_precomp"]),
                                   _ptr(out["opacities"]), _ptr(out["scales"]), _ptr(out["rotations"]),
                                   _ptr(out["cov3D_precomp"]))
    out = {k: v[:P] for k, v in out.items()}
    out["shs"] = out["shs"][:, :M]
    out["opacities"] = out["opacities"].reshape(P, 1)
    return out


def mark_visible(means3D, viewmatrix, projmatrix, dtype=np.float32) -> np.ndarray:
    lib, pre, _ = _lib(dtype)
    m = _arr(means3D, dtype)
    P = m.shape[0]
    out = np.zeros((max(P, 1),), np.uint8)
    getattr(lib, pre + "mark_visible")(P, _ptr(m), _ptr(_arr(viewmatrix, dtype, (16,))),
                                       _ptr(_arr(projmatrix, dtype, (16,))), _ptr(out))
    return out[:P].astype(bool)
