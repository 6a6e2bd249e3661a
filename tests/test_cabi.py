"""C-ABI library: loads without a GPU, exports every symbol include/unipre3d_rasterizer.h declares, host-only
entry points (scratch query, argument checks) behave.  No compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from unipre3d_amd import _lib


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "unipre3d_rasterizer.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(u3d_[a-z_0-9]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert set(names) == set(_lib.EXPORTS)
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.u3d_abi_version() == _lib.ABI_VERSION == 5
    assert lib.u3d_error_string(0) == b"ok" and b"invalid" in lib.u3d_error_string(1)


def test_header_and_ctypes_struct_agree():
    hdr = open(os.path.join(ROOT, "include", "unipre3d_rasterizer.h")).read()
    body = re.search(r"typedef struct u3d_raster_desc \{(.*?)\} u3d_raster_desc;", hdr, re.S).group(1)
    fields = re.findall(r"^\s*(?:int32_t|float|const int32_t\*)\s+(\w+);", body, re.M)
    assert fields == [f[0] for f in _lib.RasterDesc._fields_]
    assert ctypes.sizeof(_lib.RasterDesc) == 56 and _lib.RasterDesc.item_offsets.offset == 48      # 12 x 4 bytes, then the pointer
    body = re.search(r"typedef struct u3d_head_desc \{(.*?)\} u3d_head_desc;", hdr, re.S).group(1)
    assert re.findall(r"^\s*(?:int32_t|float)\s+(\w+);", body, re.M) == [f[0] for f in _lib.HeadDesc._fields_]
    body = re.search(r"typedef struct u3d_scratch_sizes \{(.*?)\} u3d_scratch_sizes;", hdr, re.S).group(1)
    fields = re.findall(r"^\s*size_t\s+(\w+);", body, re.M)
    assert fields == [f[0] for f in _lib.ScratchSizes._fields_]


def test_scratch_query_is_host_arithmetic(lib):
    d = _lib.RasterDesc(32, 4, 128, 256, 256, 0.457, 0.457, 1.0, 1, 4, _lib.FLAG_ANTIALIASING)
    s = _lib.ScratchSizes()
    assert lib.u3d_scratch_query(ctypes.byref(d), ctypes.byref(s)) == 0
    NG, NP = 32 * 4 * 128, 32 * 4 * 256 * 256
    assert s.geom_bytes >= NG * (4 + 8 + 16 + 16 + 8 + 4)
    assert s.image_bytes >= NP * 8 and s.backward_bytes >= NG * 80 and s.binning_bytes >= NG * 12
    assert s.num_rendered_offset % 256 == 0 and s.num_rendered_offset < s.geom_bytes
    # radix temporaries only beyond the LDS-sortable size
    d2 = _lib.RasterDesc(1, 8, 200000, 480, 640, 0.55, 0.55, 1.0, 1, 4, 2)
    s2 = _lib.ScratchSizes()
    assert lib.u3d_scratch_query(ctypes.byref(d2), ctypes.byref(s2)) == 0
    assert s2.binning_bytes >= 8 * 200000 * (12 + 16)
    # empty call is legal
    d0 = _lib.RasterDesc(1, 1, 0, 16, 16, 0.5, 0.5, 1.0, 0, 0, 0)
    assert lib.u3d_scratch_query(ctypes.byref(d0), ctypes.byref(s)) == 0


def test_argument_errors_without_touching_the_device(lib):
    s = _lib.ScratchSizes()
    bad = _lib.RasterDesc(1, 1, 8, 0, 16, 0.5, 0.5, 1.0, 1, 4, 0)
    assert lib.u3d_scratch_query(ctypes.byref(bad), ctypes.byref(s)) == 1
    bad = _lib.RasterDesc(1, 1, 8, 16, 16, 0.5, 0.5, 1.0, 4, 25, 0)   # SH degree 4 not supported
    assert lib.u3d_scratch_query(ctypes.byref(bad), ctypes.byref(s)) == 2
    ok = _lib.RasterDesc(1, 1, 8, 16, 16, 0.5, 0.5, 1.0, 1, 4, 0)
    null = ctypes.c_void_p(0)
    # NULL mandatory pointers are rejected before any launch
    assert lib.u3d_rasterize_forward(ctypes.byref(ok), *([null] * 18)) == 1
    assert lib.u3d_rasterize_backward(ctypes.byref(ok), *([null] * 27)) == 1
    assert lib.u3d_mark_visible(4, null, null, null, null, null) == 1
    assert lib.u3d_mark_visible(0, null, null, null, null, null) == 0


def test_product_path_refuses_cpu_tensors():
    import torch
    from unipre3d_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    st = GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3),
                                       False, False, True)
    r = GaussianRasterizer(st)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.ones(4, 1), colors_precomp=torch.ones(4, 3),
          scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.ones(4, 1), scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair"):
        r(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.ones(4, 1), colors_precomp=torch.ones(4, 3))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "unipre3d_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
    src = open(os.path.join(ROOT, "diff_gaussian_rasterization", "__init__.py")).read()
    assert "oracle" not in src


def _xcd_chunk_in_view(j, view, T):
    """Host restatement of `u3d_xcd_chunk_in_view` (unipre3d_amd/csrc/u3d_common.h): the tile kernels run a (T, views) grid; the
    workgroup with x index j of view `view` has linear id view*T + j and sits on XCD (view*T + j) % 8; within a view the blocks of
    residue class x are numbered in order and mapped to the x-th chunk of the view's tile range."""
    below = lambda n, c: (n >> 3) * c + min(n & 7, c)
    r = (view * T) & 7
    m = r + j
    x = m & 7
    k = ((m + 7 - x) >> 3) - ((r + 7 - x) >> 3)
    return (below(r + T, x) - below(r, x)) + k


def test_xcd_view_chunk_mapping_is_a_bijection():
    for T, nv in ((256, 128), (1200, 16), (64, 8), (1, 5), (2, 3), (7, 9), (8, 4), (9, 4), (13, 3), (15, 8), (63, 3), (70, 3), (1000, 2)):
        seen = set()
        for v in range(nv):
            per_xcd = [[] for _ in range(8)]
            for j in range(T):
                t = _xcd_chunk_in_view(j, v, T)
                assert 0 <= t < T, (T, v, j, t)
                seen.add(v * T + t)
                per_xcd[(v * T + j) % 8].append(t)
            sizes = [len(c) for c in per_xcd]
            assert max(sizes) - min(sizes) <= 1                 # each XCD renders ~1/8 of every view
            for c in per_xcd:                                   # ... as one contiguous chunk of the view's tiles, in order
                assert c == list(range(c[0], c[0] + len(c))) if c else True
        assert len(seen) == T * nv                              # every tile of every view exactly once (any T: per-tile scratch is indexed by it)


def test_tile_row_magic_multiplier_is_exact():
    """Host restatement of `tile_grid` (unipre3d_amd/csrc/u3d_render.hip): the tile kernels find a tile's row as
    (tile * magic) >> 32 with magic = ceil(2^32 / tiles_x), used only when T * tiles_x < 2^32 (else they divide)."""
    import random
    rnd = random.Random(3)
    shapes = [(16, 16), (40, 30), (2, 1), (3, 7), (1024, 1024), (511, 513), (65535, 1), (4096, 255)]
    for tiles_x, tiles_y in shapes:
        T = tiles_x * tiles_y
        if tiles_x == 1 or T * tiles_x >= 1 << 32:
            continue                                            # magic = 0: the kernel divides
        magic = ((1 << 32) + tiles_x - 1) // tiles_x
        assert magic < 1 << 32
        probes = {0, 1, tiles_x - 1, tiles_x, T - 1, T - tiles_x, T // 2} | {rnd.randrange(T) for _ in range(2000)}
        probes |= {y * tiles_x + dx for y in (0, tiles_y - 1, tiles_y // 2) for dx in (-1, 0, 1)}
        for tile in probes:
            if 0 <= tile < T:
                assert (tile * magic) >> 32 == tile // tiles_x, (tiles_x, tiles_y, tile)


def test_scratch_query_rejects_unsupported_shapes():
    """Host-side validation (no GPU work): more than 65535 Gaussian sets per call is reported as unsupported, not launched."""
    import ctypes
    from unipre3d_amd import _lib
    lib = _lib.load()
    d = _lib.RasterDesc(n_items=65536, views_per_item=1, P=4, image_height=16, image_width=16, tanfovx=0.5, tanfovy=0.5,
                        scale_modifier=1.0, sh_degree=1, sh_coeffs=4, flags=0)
    sizes = _lib.ScratchSizes()
    assert lib.u3d_scratch_query(ctypes.byref(d), ctypes.byref(sizes)) == 2      # U3D_ERR_UNSUPPORTED
    d.n_items = 65535
    assert lib.u3d_scratch_query(ctypes.byref(d), ctypes.byref(sizes)) == 0


def test_scratch_query_rejects_index_overflow(lib):
    """Views x tiles, sets x Gaussians and (view, Gaussian) pairs are 32-bit indices inside the launchers: shapes whose products
    leave that range come back as U3D_ERR_UNSUPPORTED instead of undefined behaviour."""
    sizes = _lib.ScratchSizes()
    ok = _lib.RasterDesc(1000, 60, 128, 4096, 4096, 0.5, 0.5, 1.0, 1, 4, 2)          # 60 000 views x 65 536 tiles > 2^31
    assert lib.u3d_scratch_query(ctypes.byref(ok), ctypes.byref(sizes)) == 2
    big_p = _lib.RasterDesc(60000, 1, 40000, 64, 64, 0.5, 0.5, 1.0, 1, 4, 2)         # 2.4e9 Gaussians
    assert lib.u3d_scratch_query(ctypes.byref(big_p), ctypes.byref(sizes)) == 2
    pairs = _lib.RasterDesc(2000, 30, 100000, 64, 64, 0.5, 0.5, 1.0, 1, 4, 2)        # 6e9 (view, Gaussian) pairs
    assert lib.u3d_scratch_query(ctypes.byref(pairs), ctypes.byref(sizes)) == 2
    fine = _lib.RasterDesc(32, 4, 128, 256, 256, 0.5, 0.5, 1.0, 1, 4, 2)
    assert lib.u3d_scratch_query(ctypes.byref(fine), ctypes.byref(sizes)) == 0


def test_ragged_layout_from_sizes_cannot_be_inconsistent():
    """`sizes=` is the checked way to describe a ragged batch (advisor, round 2): prefix sums and the largest set are derived on
    the host, malformed sizes are refused before anything reaches the kernels."""
    import torch
    from unipre3d_amd import fused
    from unipre3d_amd.rasterizer import ragged_layout, rasterize_gaussians_batched
    off, mp = ragged_layout([4, 0, 9, 2], torch.device("cpu"))
    assert off.tolist() == [0, 4, 4, 13, 15] and off.dtype == torch.int32 and mp == 9
    assert ragged_layout((4, 0, 9, 2), torch.device("cpu"))[0] is off          # cached: a repeated layout costs no copy
    for bad in ([], [0, 0], [3, -1]):
        with pytest.raises(ValueError):
            ragged_layout(bad, torch.device("cpu"))
    h = torch.zeros(15, 23)
    with pytest.raises(ValueError):          # sizes do not add up to the packed rows
        fused.render_loss_fused(h, torch.zeros(15, 3), torch.zeros(4, 1, 4, 4), torch.zeros(4, 1, 4, 4), torch.zeros(4, 1, 3),
                                torch.zeros(4, 1, 3, 8, 8), torch.zeros(3), 50.0, 8, 8, level="scene", sizes=[4, 0, 9, 3])
    with pytest.raises(ValueError):          # both forms at once
        fused.render_loss_fused(h, torch.zeros(15, 3), torch.zeros(4, 1, 4, 4), torch.zeros(4, 1, 4, 4), torch.zeros(4, 1, 3),
                                torch.zeros(4, 1, 3, 8, 8), torch.zeros(3), 50.0, 8, 8, level="scene", sizes=[4, 0, 9, 2], item_offsets=off, max_P=9)


def test_ragged_host_helpers():
    """pack_ragged / split_ragged_radii / the fused route's shape checks (host logic only)."""
    import torch
    from unipre3d_amd import fused
    from unipre3d_amd.rasterizer import pack_ragged, split_ragged_radii
    parts = [torch.arange(6.).reshape(2, 3), torch.zeros(0, 3), torch.ones(3, 3)]
    packed, off, sizes = pack_ragged(parts)
    assert packed.shape == (5, 3) and off.tolist() == [0, 2, 2, 5] and off.dtype == torch.int32 and sizes == [2, 0, 3]
    V = 2
    radii = torch.arange(V * 5)
    per = split_ragged_radii(radii, sizes, V)
    assert [tuple(p.shape) for p in per] == [(2, 2), (2, 0), (2, 3)]
    assert per[0].tolist() == [[0, 1], [2, 3]] and per[2].tolist() == [[4, 5, 6], [7, 8, 9]]      # set i, view v at V*off_i + v*P_i
    h3, h2 = torch.zeros(2, 7, 23), torch.zeros(9, 23)
    assert fused._batch_shape(h3, None, 0) == (2, 7, 23, 0)
    assert fused._batch_shape(h2, torch.tensor([0, 4, 9], dtype=torch.int32), 5) == (2, 5, 23, 9)
    for bad in ((h2, None, 0), (h3, torch.tensor([0, 1], dtype=torch.int32), 1), (h2, torch.tensor([0, 9]), 9),
                (h2, torch.tensor([0, 9], dtype=torch.int32), 0), (h2, torch.tensor([0, 9], dtype=torch.int32), 10)):
        with pytest.raises(ValueError):
            fused._batch_shape(*bad)
