"""CPU checks of the point-operator oracle against first-principles definitions, and of the C-ABI library's
exports.  No reference golden vectors exist for these operators; the oracle follows the in-tree CUDA source."""
import ctypes
import os
import re

import numpy as np

from conftest import ROOT
from oracle import pointops as po


def _cloud(B, N, seed=0, grid=False):
    rng = np.random.RandomState(seed)
    if grid:  # lattice points: plenty of exactly equal distances -> exercises the tie rules
        side = int(round(N ** (1 / 3))) + 1
        g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.25
        return np.stack([g[rng.permutation(len(g))[:N]] for _ in range(B)])
    return rng.randn(B, N, 3).astype(np.float32)


def test_fps_greedy_definition_without_ties():
    xyz = _cloud(2, 300, 1)
    idx = po.furthest_point_sampling(xyz, 40)
    for b in range(2):
        d = np.full(300, 1e10, np.float32)
        cur = 0
        assert idx[b, 0] == 0
        for j in range(1, 40):
            diff = xyz[b] - xyz[b, cur]
            d = np.minimum(d, (diff * diff).sum(1).astype(np.float32))
            nxt = int(np.argmax(d))
            assert abs(d[idx[b, j]] - d[nxt]) <= 1e-6 * max(1.0, d[nxt])     # a maximiser (fp32 rounding aside)
            cur = int(idx[b, j])
    assert len(set(idx[0].tolist())) == 40


def test_fps_tie_order_is_the_block_reductions():
    """With exact ties the selected index depends on the reference's block size: the winner is the candidate with the
    smallest (bit-reversed (k mod bs), k div bs) -- NOT the smallest k."""
    # 8 points, all at distance 1 from point 0 except duplicates: bs = opt_n_threads(8) = 8
    assert po.opt_n_threads(8) == 8 and po.opt_n_threads(1000) == 512 and po.opt_n_threads(5000) == 1024
    xyz = np.zeros((1, 8, 3), np.float32)
    xyz[0, 3] = [1, 0, 0]; xyz[0, 4] = [0, 1, 0]; xyz[0, 6] = [0, 0, 1]      # three equidistant candidates 3, 4, 6
    idx = po.furthest_point_sampling(xyz, 2)
    # lowest differing bit decides, the candidate with a 0 there wins: 3=011, 4=100, 6=110 -> 4 beats 3 (bit0), 4 beats 6 (bit1)
    assert idx[0].tolist() == [0, 4]
    # more points than threads: same slot (k mod bs), the earlier k wins
    xyz = np.zeros((1, 24, 3), np.float32)                                    # bs = 16
    xyz[0, 18] = [1, 0, 0]; xyz[0, 2] = [0, 1, 0]
    assert po.furthest_point_sampling(xyz, 2)[0].tolist() == [0, 2]


def test_ball_query_definition():
    xyz = _cloud(2, 200, 3) * 0.3
    new = xyz[:, ::10].copy()
    idx = po.ball_query(0.25, 8, xyz, new)
    for b in range(2):
        for p in range(new.shape[1]):
            d2 = ((new[b, p] - xyz[b]) ** 2).sum(1)
            hits = np.nonzero(d2 < 0.25 ** 2 - 1e-6)[0]
            exp = list(hits[:8]) + [hits[0]] * max(0, 8 - len(hits))
            got = idx[b, p].tolist()
            assert got[: min(len(hits), 8)] == exp[: min(len(hits), 8)] or abs(len(hits) - np.sum(d2 < 0.25 ** 2 + 1e-6)) > 0
            assert got[0] == p * 10 or d2[got[0]] < 0.25 ** 2                # the query point itself is the first hit
    far = np.full((1, 3, 3), 100.0, np.float32)
    assert not po.ball_query(0.1, 4, xyz[:1], far).any()                     # no neighbour -> zeros


def test_group_gather_and_grads():
    rng = np.random.RandomState(5)
    pts = rng.randn(2, 6, 50).astype(np.float32)
    idx = rng.randint(0, 50, (2, 7, 4)).astype(np.int32)
    out = po.group_points(pts, idx)
    assert np.array_equal(out, np.stack([pts[b][:, idx[b]] for b in range(2)]))
    go = rng.randn(*out.shape).astype(np.float32)
    g = po.group_points_grad(go, idx, 50)
    ref = np.zeros_like(pts)
    for b in range(2):
        np.add.at(ref[b], (slice(None), idx[b].reshape(-1)), go[b].reshape(6, -1))
    assert np.allclose(g, ref, atol=1e-5)
    gi = idx[:, :, 0].copy()
    assert np.array_equal(po.gather_points(pts, gi), np.stack([pts[b][:, gi[b]] for b in range(2)]))
    gg = po.gather_points_grad(go[:, :, :, 0].copy(), gi, 50)
    ref = np.zeros_like(pts)
    for b in range(2):
        np.add.at(ref[b], (slice(None), gi[b]), go[b, :, :, 0])
    assert np.allclose(gg, ref, atol=1e-5)


def test_pointops_library_exports_every_declared_symbol():
    from unipre3d_amd import pointops
    hdr = open(os.path.join(ROOT, "include", "unipre3d_pointops.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(u3d_[a-z_0-9]+)\s*\(", hdr)))
    assert set(names) == set(pointops.EXPORTS)
    lib = pointops.load()
    for n in names:
        assert getattr(lib, n) is not None
    null = ctypes.c_void_p(0)
    assert lib.u3d_furthest_point_sampling(1, 8, 2, null, null, null, null) == 1        # NULL pointers rejected
    assert lib.u3d_ball_query(0, 8, 2, ctypes.c_float(0.1), 4, null, null, null, null) == 0   # empty batch is legal


def test_three_nn_is_the_three_smallest_distance_index_pairs():
    """interpolate_gpu.cu:16-59: strict `<` insertions in index order = the three smallest (distance, index) pairs in
    lexicographic order (ties keep the lower index); fewer than three known points leave 1e40 -> +inf and index 0."""
    for grid, seed in ((False, 1), (True, 2)):
        unk, kn = _cloud(2, 300, seed=seed, grid=grid), _cloud(2, 77, seed=seed + 10, grid=grid)
        d2, idx = po.three_nn(unk, kn)
        for b in range(2):
            dx = unk[b][:, None, :] - kn[b][None, :, :]
            # the oracle's default contraction order (LLVM's): fma(dz, dz, fma(dx, dx, dy * dy)) -- float64 products rounded per step
            f32 = np.float32
            t = (dx[..., 1] * dx[..., 1]).astype(f32)
            t = (dx[..., 0].astype(np.float64) * dx[..., 0] + t).astype(f32)
            d = (dx[..., 2].astype(np.float64) * dx[..., 2] + t).astype(f32)
            order = np.lexsort((np.arange(77)[None, :].repeat(300, 0), d), axis=1)[:, :3]
            assert np.array_equal(idx[b], order.astype(np.int32))
            assert np.array_equal(d2[b], np.take_along_axis(d, order, 1))
        if grid:
            assert (d2[..., 0] == d2[..., 1]).any()       # the lattice really produces ties
    d2, idx = po.three_nn(_cloud(1, 5, seed=3), _cloud(1, 2, seed=4))
    assert np.isinf(d2[..., 2]).all() and (idx[..., 2] == 0).all() and np.isfinite(d2[..., :2]).all()
    d2, idx = po.three_nn(_cloud(1, 5, seed=3), np.zeros((1, 0, 3), np.float32))
    assert np.isinf(d2).all() and not idx.any()


def test_three_interpolate_definition_and_grad():
    rng = np.random.RandomState(5)
    feats = rng.randn(2, 6, 40).astype(np.float32)
    idx = rng.randint(0, 40, (2, 90, 3)).astype(np.int32)
    w = rng.rand(2, 90, 3).astype(np.float32)
    out = po.three_interpolate(feats, idx, w)
    ref = np.einsum("bnk,bcnk->bcn", w.astype(np.float64), np.stack([feats[b][:, idx[b]] for b in range(2)]).astype(np.float64))
    assert out.shape == (2, 6, 90) and np.allclose(out, ref, rtol=1e-6, atol=1e-6)
    go = rng.randn(2, 6, 90).astype(np.float32)
    g = po.three_interpolate_grad(go, idx, w, 40)
    gref = np.zeros((2, 6, 40))
    for b in range(2):
        for k in range(3):
            np.add.at(gref[b], (slice(None), idx[b, :, k]), go[b].astype(np.float64) * w[b, :, k][None, :])
    assert np.allclose(g, gref, rtol=1e-5, atol=1e-5)


def test_contraction_modes_and_where_they_differ(capsys):
    """How nvcc's -fmad=true contracts `dx*dx + dy*dy + dz*dz` cannot be read off the reference source, so the oracle (and the HIP
    kernels, tests/test_gpu_pointops.py) carry all three candidates.  This test REPORTS on which clouds the index outputs of FPS /
    ball query / 3-NN differ between them, and pins the structure of the answer: the three forms are different roundings of one sum
    (squared distances agree to 2 ulp), random clouds without exact ties give the same FPS picks in every mode (the arg-max gaps are
    far larger than an ulp), and a cloud built to sit on a rounding boundary does flip."""
    modes = ("fma_llvm", "fma_chain", "none")
    report = []
    try:
        for name, xyz, m in (("random 1024", _cloud(2, 1024, 5), 128), ("random 5000", _cloud(1, 5000, 6), 256),
                             ("lattice 1000", _cloud(2, 1000, 7, grid=True), 100), ("lattice 4096 x 0.1", _cloud(1, 4096, 8, grid=True) * np.float32(0.4), 256)):
            fps, bq, nn, d2 = {}, {}, {}, {}
            new = xyz[:, ::7].copy()
            for mode in modes:
                po.set_contraction(mode)
                fps[mode] = po.furthest_point_sampling(xyz, m)
                bq[mode] = po.ball_query(0.6, 16, xyz, new)
                d2[mode], nn[mode] = po.three_nn(new + np.float32(0.013), xyz)
            for a, b in (("fma_llvm", "fma_chain"), ("fma_llvm", "none"), ("fma_chain", "none")):
                report.append((name, a, b, int((fps[a] != fps[b]).sum()), fps[a].size, int((bq[a] != bq[b]).sum()), bq[a].size,
                               int((nn[a] != nn[b]).sum()), nn[a].size))
                ulp = np.abs(d2[a].view(np.int32).astype(np.int64) - d2[b].view(np.int32).astype(np.int64))
                assert ulp[nn[a] == nn[b]].max() <= 2
            if name.startswith("random"):
                assert np.array_equal(fps["fma_llvm"], fps["fma_chain"]) and np.array_equal(fps["fma_llvm"], fps["none"])
        # a constructed flip: two candidates whose squared distances from point 0 differ only in how the sum is rounded
        rng = np.random.RandomState(11)
        flips = 0
        for _ in range(4000):
            v = rng.randn(3).astype(np.float32)
            picks = set()
            for mode in modes:
                po.set_contraction(mode)
                dd = po.three_nn(np.zeros((1, 1, 3), np.float32), np.stack([v, v[[1, 0, 2]], v[[2, 1, 0]]])[None])[0]
                picks.add(dd.tobytes())
            flips += len(picks) > 1
        assert flips > 0          # permuting the coordinates changes the rounded sum in some mode: the modes are not the same function
    finally:
        po.set_contraction("fma_llvm")
    with capsys.disabled():
        print("\n[pointops contraction] index entries that differ between modes (FPS | ball query | 3-NN):")
        for name, a, b, f, fn, q, qn, t, tn in report:
            print(f"[pointops contraction]   {name:20s} {a:9s} vs {b:9s}: {f}/{fn} | {q}/{qn} | {t}/{tn}")
