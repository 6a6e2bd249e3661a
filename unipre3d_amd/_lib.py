"""ctypes binding of libunipre3d_rasterizer.so (the C-ABI declared in include/unipre3d_rasterizer.h).

The product path has NO CPU or PyTorch fallback: if the HIP library is missing or cannot be loaded,
`load()` raises.  `import torch` must precede the dlopen so that the library's libamdhip64.so.7
dependency resolves to the HIP runtime PyTorch-ROCm already loaded (one runtime per process, so torch's
streams and device pointers are valid inside the library).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch  # noqa: F401  (must be imported before the library is opened, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (experiments only: U3D_LIB_DIRNAME selects a kernel-variant build directory made with `make LIBDIR=../lib_x EXTRA=...`)
LIB_DIR = os.path.join(_HERE, os.environ.get("U3D_LIB_DIRNAME", "lib"))
LIB_PATH = os.path.join(LIB_DIR, "libunipre3d_rasterizer.so")
CSRC = os.path.join(_HERE, "csrc")

ABI_VERSION = 5   # include/unipre3d_rasterizer.h: U3D_ABI_VERSION
FLAG_PREFILTERED, FLAG_ANTIALIASING, FLAG_DEBUG, FLAG_EXACT_AA_GRAD, FLAG_STATS, FLAG_ACC_CLEAN, FLAG_SPARSE_BWD = 1, 2, 4, 8, 16, 32, 64

EXPORTS = ("u3d_abi_version", "u3d_error_string", "u3d_scratch_query", "u3d_rasterize_forward",
           "u3d_rasterize_backward", "u3d_mark_visible", "u3d_profile_begin", "u3d_profile_end",
           "u3d_render_loss_forward", "u3d_render_loss_backward", "u3d_render_loss_step",
           "u3d_render_loss_step_forward", "u3d_render_loss_step_backward", "u3d_render_view_forward", "u3d_render_view_backward")
PROFILE_KINDS = ("preprocess_fwd", "depth_sort", "render_fwd", "render_bwd", "preprocess_bwd", "render_fb")


class RasterDesc(ctypes.Structure):
    _fields_ = [("n_items", ctypes.c_int32), ("views_per_item", ctypes.c_int32), ("P", ctypes.c_int32),
                ("image_height", ctypes.c_int32), ("image_width", ctypes.c_int32), ("tanfovx", ctypes.c_float),
                ("tanfovy", ctypes.c_float), ("scale_modifier", ctypes.c_float), ("sh_degree", ctypes.c_int32),
                ("sh_coeffs", ctypes.c_int32), ("flags", ctypes.c_int32),
                # ragged batches: sum of the sets' sizes (0 = every set has P) and the DEVICE pointer to the n_items + 1 prefix sums
                ("total_P", ctypes.c_int32), ("item_offsets", ctypes.c_void_p)]


class ScratchSizes(ctypes.Structure):
    _fields_ = [("geom_bytes", ctypes.c_size_t), ("binning_bytes", ctypes.c_size_t), ("image_bytes", ctypes.c_size_t),
                ("backward_bytes", ctypes.c_size_t), ("num_rendered_offset", ctypes.c_size_t),
                ("fused_bytes", ctypes.c_size_t)]


class HeadDesc(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int32), ("channels", ctypes.c_int32), ("offset_scale", ctypes.c_float),
                ("isotropic", ctypes.c_int32)]


class LossDesc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("non_bg_color_loss_rate", ctypes.c_float),
                ("bg_color_loss_rate", ctypes.c_float)]


LOSS_KINDS = {"l2": 1, "focal_l2": 2, "l1": 3}


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc build of libunipre3d_rasterizer.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X rasterizer has no fallback path. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C unipre3d_amd/csrc`.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    lib.u3d_abi_version.restype = ctypes.c_int
    lib.u3d_error_string.restype = ctypes.c_char_p
    lib.u3d_error_string.argtypes = [ctypes.c_int]
    lib.u3d_scratch_query.restype = ctypes.c_int
    lib.u3d_scratch_query.argtypes = [ctypes.POINTER(RasterDesc), ctypes.POINTER(ScratchSizes)]
    lib.u3d_rasterize_forward.restype = ctypes.c_int
    lib.u3d_rasterize_forward.argtypes = [ctypes.POINTER(RasterDesc)] + [vp] * 18
    lib.u3d_rasterize_backward.restype = ctypes.c_int
    lib.u3d_rasterize_backward.argtypes = [ctypes.POINTER(RasterDesc)] + [vp] * 27
    lib.u3d_mark_visible.restype = ctypes.c_int
    lib.u3d_mark_visible.argtypes = [i32, vp, vp, vp, vp, vp]
    lib.u3d_render_loss_forward.restype = ctypes.c_int
    lib.u3d_render_loss_forward.argtypes = [ctypes.POINTER(RasterDesc), ctypes.POINTER(HeadDesc), ctypes.POINTER(LossDesc)] + [vp] * 15
    lib.u3d_render_loss_backward.restype = ctypes.c_int
    lib.u3d_render_loss_backward.argtypes = [ctypes.POINTER(RasterDesc), ctypes.POINTER(HeadDesc), ctypes.POINTER(LossDesc)] + [vp] * 18
    lib.u3d_render_loss_step.restype = ctypes.c_int
    lib.u3d_render_loss_step.argtypes = [ctypes.POINTER(RasterDesc), ctypes.POINTER(HeadDesc), ctypes.POINTER(LossDesc)] + [vp] * 16
    lib.u3d_render_loss_step_forward.restype = ctypes.c_int
    lib.u3d_render_loss_step_forward.argtypes = [ctypes.POINTER(RasterDesc), ctypes.POINTER(HeadDesc), ctypes.POINTER(LossDesc)] + [vp] * 16
    lib.u3d_render_loss_step_backward.restype = ctypes.c_int
    lib.u3d_render_loss_step_backward.argtypes = [ctypes.POINTER(RasterDesc), ctypes.POINTER(HeadDesc)] + [vp] * 13
    lib.u3d_render_view_forward.restype = ctypes.c_int
    lib.u3d_render_view_forward.argtypes = [ctypes.POINTER(RasterDesc)] + [vp] * 17
    lib.u3d_render_view_backward.restype = ctypes.c_int
    lib.u3d_render_view_backward.argtypes = [ctypes.POINTER(RasterDesc)] + [vp] * 24
    lib.u3d_profile_begin.restype = ctypes.c_int
    lib.u3d_profile_begin.argtypes = [i32]
    lib.u3d_profile_end.restype = ctypes.c_int
    lib.u3d_profile_end.argtypes = [vp, vp]
    if lib.u3d_abi_version() != ABI_VERSION:
        raise RuntimeError("libunipre3d_rasterizer.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed: {load().u3d_error_string(code).decode()} (code {code})")


def ptr(t) -> ctypes.c_void_p:
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def profile_begin(max_records: int = 65536, kinds=None, stride: int = 1) -> None:
    """kinds: iterable of PROFILE_KINDS names to record (None = all); stride: sample every stride-th launch of a kind (1..15).
    Each recorded scope costs ~4-5 us on the stream."""
    arg = max_records
    if kinds is not None or stride != 1:
        mask = 0
        for k in (kinds if kinds is not None else PROFILE_KINDS):
            mask |= 1 << PROFILE_KINDS.index(k)
        arg = -((max(1, min(int(stride), 15)) << 26) | (mask << 20) | min(max_records, 0xfffff))
    check(load().u3d_profile_begin(arg), "u3d_profile_begin")


def profile_end() -> dict:
    """{kind: (total_ms, launches)} for the kernels enqueued since profile_begin()."""
    ms = (ctypes.c_float * len(PROFILE_KINDS))()
    cnt = (ctypes.c_int32 * len(PROFILE_KINDS))()
    check(load().u3d_profile_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p)), "u3d_profile_end")
    return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(PROFILE_KINDS)}
