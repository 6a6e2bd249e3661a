"""Python surface of the operator: `GaussianRasterizationSettings` + `GaussianRasterizer`.

Mirrors what gaussian_renderer/__init__.py:8,45-61,89-97 needs from `diff_gaussian_rasterization`
(SURVEY.md section 8b): a 13-field NamedTuple constructible by keyword, and an nn.Module called by
keyword that returns the 3-tuple (color (3,H,W), radii (P,) int32, invdepth (1,H,W)) and is
differentiable w.r.t. means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
cov3D_precomp.  The arithmetic runs in libunipre3d_rasterizer.so (hand-written gfx950 kernels) on
torch's CURRENT stream, with no device->host synchronisation; the autograd function around it is C++
(csrc/u3d_torch.cpp, loaded by `_C()`), because the reference calls the operator once per object and view.

`rasterize_gaussians_batched` is the same operator for B sets x V cameras in one launch sequence
(replaces the Python loop of train_network.py:418-446).
"""
from __future__ import annotations

import ctypes
import os
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool = False


def _f32c(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _Plan:
    """Descriptor + scratch sizes for one call shape (cached).  Ragged batches (total_P > 0) share the sizes of their
    (n_items, views, largest set, total) shape; the device pointer to the prefix sums is set per call (`with_offsets`)."""
    _cache = {}

    def __new__(cls, *key, total_P=0):
        ck = key + (int(total_P),)
        hit = cls._cache.get(ck)
        if hit is not None:
            return hit
        self = super().__new__(cls)
        self.desc = _lib.RasterDesc(*key)
        self.desc.total_P = int(total_P)
        probe = _lib.RasterDesc(*key)
        probe.total_P = int(total_P)
        probe.item_offsets = 8 if total_P else None          # (the size query only looks at whether it is set)
        self.sizes = _lib.ScratchSizes()
        _lib.check(_lib.load().u3d_scratch_query(ctypes.byref(probe), ctypes.byref(self.sizes)), "u3d_scratch_query")
        if len(cls._cache) < 256:
            cls._cache[ck] = self
        return self

    def __init__(self, n_items, vpi, P, H, W, tanfovx, tanfovy, scale_modifier, sh_degree, M, flags, total_P=0):
        pass

    def with_offsets(self, item_offsets: Optional[torch.Tensor]) -> "_lib.RasterDesc":
        """The descriptor of this call: a copy carrying the device pointer of `item_offsets` (int32, n_items + 1)."""
        if not self.desc.total_P:
            return self.desc
        d = _lib.RasterDesc.from_buffer_copy(self.desc)
        d.item_offsets = item_offsets.data_ptr()
        return d


def pack_ragged(tensors, device=None):
    """Per-item list of (P_i, ...) tensors -> (packed (sum P_i, ...), item_offsets int32 (n+1,) on the device, sizes list).
    The reference's scene-level branch hands its Gaussians over as such lists (model/gaussian_predictor.py:331-364)."""
    sizes = [int(t.shape[0]) for t in tensors]
    off = [0]
    for n in sizes:
        off.append(off[-1] + n)
    dev = device if device is not None else tensors[0].device
    return torch.cat(list(tensors), dim=0), torch.tensor(off, dtype=torch.int32, device=dev), sizes


_layout_cache = {}


def ragged_layout(sizes, device):
    """Host-side sizes of the sets -> (item_offsets int32 (n+1,) on `device`, max_P).  This is the checked way to describe a
    ragged batch: the prefix sums are built here from the sizes (monotone, first 0, last sum, largest set = max_P by
    construction), so the kernels' trust in `item_offsets` is never misplaced; the entry points take `sizes=` and call this.
    Recent layouts are kept (a training loop cycles through few of them) so that a repeated layout costs no H2D copy."""
    sizes = tuple(int(n) for n in sizes)
    if not sizes or min(sizes) < 0 or max(sizes) <= 0:
        raise ValueError(f"ragged batch: set sizes must be non-negative with at least one non-empty set, got {sizes}")
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:          # a bare "cuda" means the CURRENT device: key on where the table lives
        device = torch.device("cuda", torch.cuda.current_device())
    key = (sizes, device.type, device.index)
    hit = _layout_cache.get(key)
    if hit is None:
        off = [0]
        for n in sizes:
            off.append(off[-1] + n)
        if off[-1] >= 1 << 31:
            raise ValueError("ragged batch: more than 2^31 Gaussians")
        if len(_layout_cache) >= 64:
            _layout_cache.clear()
        hit = _layout_cache[key] = (torch.tensor(off, dtype=torch.int32, device=device), max(sizes))
    return hit


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(dev=None):
    """torch's CURRENT stream on the current device as a hipStream_t (the raw-handle query costs ~0.3 us, the Stream object
    route ~10 us per call).  The kernels are launched on the calling thread's current HIP device, so tensors on another
    device are refused instead of being launched against the wrong queue."""
    cur = torch.cuda.current_device()
    if dev is not None and dev.index is not None and dev.index != cur:
        raise RuntimeError(f"tensors live on cuda:{dev.index} but the current device is cuda:{cur}; "
                           f"call under torch.cuda.device({dev.index}) (one process per GPU sets it once)")
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(cur))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_BINDING_PATH = os.path.join(_lib.LIB_DIR, "_u3d_torch.so")
_binding = None


def _C():
    """The torch binding of the operator (csrc/u3d_torch.cpp): a C++ autograd function over the C-ABI -- what the third-party
    package's `_C` module is to its Python wrapper.  The reference calls the operator once per object and view
    (train_network.py:418-446), so the per-call host cost decides that route: ~25 us per forward+backward pair here against
    ~200 us for an autograd.Function written in Python over ctypes.  No fallback: a missing binding raises."""
    global _binding
    if _binding is None:
        _lib.load()          # the C-ABI library the binding links against (and `import torch` before either)
        if not os.path.exists(_BINDING_PATH):
            raise RuntimeError(f"{_BINDING_PATH} is missing: the MI355X rasterizer has no fallback path. Build it with "
                               "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C unipre3d_amd/csrc`.")
        import importlib.machinery
        import importlib.util
        spec = importlib.util.spec_from_loader("_u3d_torch", importlib.machinery.ExtensionFileLoader("_u3d_torch", _BINDING_PATH))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.abi_version() != _lib.ABI_VERSION:
            raise RuntimeError("_u3d_torch.so was built against another ABI version of libunipre3d_rasterizer.so; rebuild")
        _binding = mod
    return _binding


def _flags(settings: GaussianRasterizationSettings, exact_aa_grad: bool = False) -> int:
    return ((_lib.FLAG_PREFILTERED if settings.prefiltered else 0) | (_lib.FLAG_ANTIALIASING if settings.antialiasing else 0)
            | (_lib.FLAG_DEBUG if settings.debug else 0) | (_lib.FLAG_EXACT_AA_GRAD if exact_aa_grad else 0))


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings: GaussianRasterizationSettings, exact_aa_grad: bool = False):
    """One view (the reference's operator).  Shapes: means3D (P,3), means2D (P,3), sh (P,M,3) | colors (P,3),
    opacities (P,1), scales (P,3), rotations (P,4) | cov3D (P,6).  Empty tensors count as absent (upstream's convention)."""
    s = raster_settings
    return _C().rasterize_view(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, s.viewmatrix,
                               s.projmatrix, s.campos, s.bg, int(s.image_height), int(s.image_width), float(s.tanfovx), float(s.tanfovy),
                               float(s.scale_modifier), int(s.sh_degree), _flags(s, exact_aa_grad))


def rasterize_gaussians_batched(means3D, opacities, viewmatrix, projmatrix, campos, bg, image_height, image_width, tanfovx,
                                tanfovy, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                                sh_degree=0, scale_modifier=1.0, antialiasing=True, debug=False, exact_aa_grad=False,
                                means2D=None, item_offsets=None, max_P=0, sizes=None):
    """B sets x V cameras in ONE launch sequence.
    means3D (B,P,3), opacities (B,P,1), shs (B,P,M,3) | colors_precomp (B,P,3), scales (B,P,3) + rotations (B,P,4) |
    cov3D_precomp (B,P,6); viewmatrix/projmatrix (B,V,4,4), campos (B,V,3), bg (3,).
    means2D (B*V,P,3), optional: the per-view screen-space gradient sink (`viewspace_points` of
    gaussian_renderer/__init__.py:29); its .grad receives dL/dmean2D.
    Returns color (B,V,3,H,W), radii (B,V,P) int32, invdepth (B,V,1,H,W).
    Ragged batch (`sizes` = the sets' sizes on the host, as `pack_ragged` returns them): every per-Gaussian tensor is PACKED
    (sum P_i, ...) in set order; radii come back packed (V * sum P_i,), the pairs of set i and view v starting at
    V * item_offsets[i] + v * P_i (`split_ragged_radii`), means2D is (V * sum P_i, 3) in the same layout.  (Expert form:
    item_offsets int32 (B+1,) already on the device + max_P = largest set; the kernels TRUST those -- checked on the device only
    with debug=True -- whereas `sizes` cannot be inconsistent.)"""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("the MI355X rasterizer needs tensors on a HIP device; there is no CPU fallback "
                           "(the CPU restatement lives in oracle/ and is test infrastructure only)")
    if sizes is not None:
        if item_offsets is not None:
            raise ValueError("pass either sizes or item_offsets / max_P, not both")
        if sum(int(n) for n in sizes) != means3D.shape[0]:
            raise ValueError(f"ragged batch: sizes sum to {sum(sizes)} but {means3D.shape[0]} Gaussians are packed")
        item_offsets, max_P = ragged_layout(sizes, dev)
    ragged = item_offsets is not None
    B = item_offsets.numel() - 1 if ragged else means3D.shape[0]
    P = int(max_P) if ragged else means3D.shape[1]
    V = viewmatrix.shape[1]
    flags = (_lib.FLAG_ANTIALIASING if antialiasing else 0) | (_lib.FLAG_DEBUG if debug else 0) | \
        (_lib.FLAG_EXACT_AA_GRAD if exact_aa_grad else 0)
    color, radii, invdepth = _C().rasterize_batched(
        means3D, means2D, shs, colors_precomp, opacities.reshape(-1, 1) if ragged else opacities.reshape(B, P, 1), scales, rotations,
        cov3D_precomp, viewmatrix.reshape(B * V, 16), projmatrix.reshape(B * V, 16), campos.reshape(B * V, 3), bg.reshape(3), B, V,
        int(image_height), int(image_width), float(tanfovx), float(tanfovy), float(scale_modifier), int(sh_degree), flags,
        item_offsets, P if ragged else 0)
    return (color.reshape(B, V, 3, image_height, image_width), radii if ragged else radii.reshape(B, V, P),
            invdepth.reshape(B, V, 1, image_height, image_width))


def split_ragged_radii(radii: torch.Tensor, sizes, V: int):
    """Packed (V * sum P_i,) per-(view, Gaussian) values of a ragged call -> list of per-set (V, P_i, ...) views."""
    out, o = [], 0
    for n in sizes:
        out.append(radii[V * o: V * (o + n)].reshape(V, n, *radii.shape[1:]))
        o += n
    return out


# bench.py's control of the per-view route only: a callable taking the operator's differentiable tensor inputs and returning the
# operator's 3-tuple WITHOUT launching anything (`per_view_dropin.noop_operator_ms`: what the loop around the operator costs by
# itself).  None in every product path.
_operator_override = None


def set_operator_override(fn):
    """Install (or, with None / the returned previous value, remove) the bench's NO-OP operator control; returns the previous one."""
    global _operator_override
    prev, _operator_override = _operator_override, fn
    return prev


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        s = self.raster_settings
        with torch.no_grad():
            pos = _f32c(positions, positions.device)
            P = pos.shape[0]
            present = torch.zeros(P, dtype=torch.uint8, device=pos.device)
            rc = _lib.load().u3d_mark_visible(P, _lib.ptr(pos), _lib.ptr(_f32c(s.viewmatrix, pos.device)),
                                              _lib.ptr(_f32c(s.projmatrix, pos.device)), _lib.ptr(present), _stream_ptr(pos.device))
            _lib.check(rc, "u3d_mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if _operator_override is not None:
            return _operator_override(*[t for t in (means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
                                        if t is not None])
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)
