"""HIP point operators against the oracle: indices bit-exact, gathered values bit-exact, scatter-add
gradients to fp32 summation order."""
import numpy as np
import pytest
import torch

from oracle import pointops as po
from test_pointops_oracle import _cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["fma_llvm", "fma_chain", "none"])
def contraction(request):
    """Every test of this file runs in each contraction mode of the three-term sums (include/unipre3d_pointops.h): the HIP kernels and
    the oracle are switched together and must agree bit for bit in all of them -- whichever nvcc would have chosen."""
    from unipre3d_amd import pointops
    po.set_contraction(request.param)
    pointops.set_contraction(request.param)
    yield request.param
    po.set_contraction("fma_llvm")
    pointops.set_contraction("fma_llvm")


@pytest.mark.parametrize("B,N,M,grid", [(4, 1024, 128, False), (2, 1000, 100, True), (3, 2048, 512, False), (2, 5000, 64, True),
                                        (1, 9000, 32, False), (2, 17, 17, False), (1, 1, 1, False)])
def test_fps_bit_exact(B, N, M, grid):
    from unipre3d_amd import pointops
    xyz = _cloud(B, N, seed=N, grid=grid)
    got = pointops.furthest_point_sample(torch.from_numpy(xyz).cuda(), M).cpu().numpy()
    assert got.dtype == np.int32 and np.array_equal(got, po.furthest_point_sampling(xyz, M))


@pytest.mark.parametrize("B,N,M,r,K", [(4, 1024, 128, 0.1, 32), (2, 1000, 77, 0.3, 16), (1, 70, 5, 5.0, 64), (2, 300, 10, 1e-4, 8)])
def test_ball_query_bit_exact(B, N, M, r, K):
    from unipre3d_amd import pointops
    xyz = (_cloud(B, N, seed=7) * 0.3).astype(np.float32)
    new = xyz[:, :M].copy() + (0.0 if r > 1e-3 else 10.0)          # last case: no neighbour at all -> zeros
    got = pointops.ball_query(r, K, torch.from_numpy(xyz).cuda(), torch.from_numpy(new).cuda()).cpu().numpy()
    assert np.array_equal(got, po.ball_query(r, K, xyz, new))


def test_group_gather_forward_backward():
    from unipre3d_amd import pointops
    rng = np.random.RandomState(2)
    pts = rng.randn(3, 16, 400).astype(np.float32)
    idx = rng.randint(0, 400, (3, 50, 8)).astype(np.int32)
    f = torch.from_numpy(pts).cuda().requires_grad_(True)
    out = pointops.grouping_operation(f, torch.from_numpy(idx).cuda())
    assert np.array_equal(out.detach().cpu().numpy(), po.group_points(pts, idx))
    go = rng.randn(*out.shape).astype(np.float32)
    out.backward(torch.from_numpy(go).cuda())
    assert np.allclose(f.grad.cpu().numpy(), po.group_points_grad(go, idx, 400), rtol=1e-5, atol=1e-5)
    f.grad = None
    gi = np.ascontiguousarray(idx[:, :, 0])
    out = pointops.gather_operation(f, torch.from_numpy(gi).cuda())
    assert np.array_equal(out.detach().cpu().numpy(), po.gather_points(pts, gi))
    out.backward(torch.from_numpy(np.ascontiguousarray(go[:, :, :, 0])).cuda())
    assert np.allclose(f.grad.cpu().numpy(), po.gather_points_grad(go[:, :, :, 0], gi, 400), rtol=1e-5, atol=1e-5)


def test_transformer_tokenizer_pipeline():
    """The grouping front end of the reference's transformer backbone (openpoints/models/backbone/transformer.py:290-327):
    FPS(128) -> ball query(r=0.1, k=32) -> group, on a ShapeNet-sized cloud."""
    from unipre3d_amd import pointops
    xyz = (_cloud(8, 1024, seed=3) * 0.2).astype(np.float32)
    x = torch.from_numpy(xyz).cuda()
    centers = pointops.fps(x, 128)
    cidx = po.furthest_point_sampling(xyz, 128)
    assert np.array_equal(centers.cpu().numpy(), np.take_along_axis(xyz, cidx[..., None].astype(np.int64), 1))
    gx, gf = pointops.QueryAndGroup(0.1, 32)(centers.contiguous(), x, x.transpose(1, 2).contiguous())
    assert gx.shape == (8, 3, 128, 32) and gf.shape == (8, 3, 128, 32)
    idx = po.ball_query(0.1, 32, xyz, centers.cpu().numpy())
    assert np.array_equal(gf.cpu().numpy(), po.group_points(xyz.transpose(0, 2, 1).copy(), idx))


@pytest.mark.parametrize("B,N,M,grid", [(4, 1024, 128, False), (2, 1000, 300, True), (1, 5000, 4500, False), (2, 130, 2, False),
                                        (1, 7, 1, False), (3, 257, 2049, True)])
def test_three_nn_bit_exact(B, N, M, grid):
    """interpolate_gpu.cu:16-59: squared distances and indices bit-exact against the oracle, incl. lattice clouds full of ties,
    fewer than three known points (1e40 -> +inf, index 0) and a known cloud spanning several LDS tiles."""
    from unipre3d_amd import pointops
    unk, kn = _cloud(B, N, seed=N, grid=grid), _cloud(B, M, seed=M + 1, grid=grid)
    dist, idx = pointops.three_nn(torch.from_numpy(unk).cuda(), torch.from_numpy(kn).cuda())
    d2_ref, idx_ref = po.three_nn(unk, kn)
    assert idx.dtype == torch.int32 and np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.array_equal(dist.cpu().numpy(), np.sqrt(d2_ref))            # the wrapper returns L2 distances (upsampling.py:35)


def test_three_interpolate_forward_backward():
    """Values bit-exact (same left-to-right contraction), scatter-add gradient to fp32 summation order; and the reference's
    three_interpolation composite (upsampling.py:92-101) against the same chain through the oracle."""
    from unipre3d_amd import pointops
    rng = np.random.RandomState(11)
    feats = rng.randn(3, 37, 200).astype(np.float32)
    idx = rng.randint(0, 200, (3, 1000, 3)).astype(np.int32)
    w = rng.rand(3, 1000, 3).astype(np.float32)
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    out = pointops.three_interpolate(f, torch.from_numpy(idx).cuda(), torch.from_numpy(w).cuda())
    assert np.array_equal(out.detach().cpu().numpy(), po.three_interpolate(feats, idx, w))
    go = rng.randn(*out.shape).astype(np.float32)
    out.backward(torch.from_numpy(go).cuda())
    gref = po.three_interpolate_grad(go, idx, w, 200)
    assert np.abs(f.grad.cpu().numpy() - gref).max() <= 1e-5 * np.abs(gref).max()
    # composite
    unk, kn = _cloud(2, 500, seed=1), _cloud(2, 64, seed=2)
    kf = rng.randn(2, 16, 64).astype(np.float32)
    got = pointops.three_interpolation(torch.from_numpy(unk).cuda(), torch.from_numpy(kn).cuda(), torch.from_numpy(kf).cuda()).cpu().numpy()
    d2, ix = po.three_nn(unk, kn)
    rec = 1.0 / (torch.sqrt(torch.from_numpy(d2)) + 1e-8)
    wt = (rec / rec.sum(dim=2, keepdim=True)).numpy()
    ref = po.three_interpolate(kf, ix, wt)
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


def test_gradient_kernels_beyond_the_lds_rows():
    """The scatter-add gradients accumulate channel rows in LDS (round 5); rows that do not fit 64 KB (n resp. m > 16384) take the
    global-atomic form, and a shape whose channel count is not a multiple of the rows per workgroup exercises the tail."""
    from unipre3d_amd import pointops
    rng = np.random.RandomState(5)
    for (C, N, M, K) in ((3, 20000, 40, 8), (37, 700, 33, 5), (20, 16384, 16, 4)):
        pts = rng.randn(2, C, N).astype(np.float32)
        idx = rng.randint(0, N, (2, M, K)).astype(np.int32)
        idx[:, :, K // 2:] = idx[:, :, :1]                                   # ball-query style padding: many duplicates
        f = torch.from_numpy(pts).cuda().requires_grad_(True)
        out = pointops.grouping_operation(f, torch.from_numpy(idx).cuda())
        assert np.array_equal(out.detach().cpu().numpy(), po.group_points(pts, idx))
        go = rng.randn(*out.shape).astype(np.float32)
        out.backward(torch.from_numpy(go).cuda())
        gref = po.group_points_grad(go, idx, N)
        assert np.abs(f.grad.cpu().numpy() - gref).max() <= 1e-5 * max(np.abs(gref).max(), 1.0)
    # ((40, 2048, 512): most entries on ONE destination -- the compare-and-swap accumulate's worst case; (3, 2600, 9): a tiny known cloud)
    for (C, n, m) in ((5, 300, 17000), (19, 1000, 50), (40, 2048, 512), (3, 2600, 9)):
        feats = rng.randn(2, C, m).astype(np.float32)
        idx = rng.randint(0, m, (2, n, 3)).astype(np.int32)
        if m == 512:
            idx[:, ::2, 0] = 7; idx[:, 1::3, 2] = 7
        w = rng.rand(2, n, 3).astype(np.float32)
        f = torch.from_numpy(feats).cuda().requires_grad_(True)
        out = pointops.three_interpolate(f, torch.from_numpy(idx).cuda(), torch.from_numpy(w).cuda())
        go = rng.randn(*out.shape).astype(np.float32)
        out.backward(torch.from_numpy(go).cuda())
        gref = po.three_interpolate_grad(go, idx, w, m)
        assert np.abs(f.grad.cpu().numpy() - gref).max() <= 1e-5 * max(np.abs(gref).max(), 1.0)


@pytest.mark.parametrize("N", [64, 100, 128, 130, 256, 300, 511, 512])
def test_fps_one_wave_per_cloud_ties_and_sizes(N):
    """Round 6: clouds of up to 512 points run as ONE wave (2 / 4 / 8 points per lane).  Its common path takes the holder of the maximum from a vector
    count; exact ties -- every point duplicated, so that two points sit at EVERY distance, and a lattice -- must fall back to the reference's tie order.
    Bit-identical selections against the oracle for every size class and its edges."""
    from unipre3d_amd import pointops
    rng = np.random.RandomState(N)
    half = rng.randn(2, (N + 1) // 2, 3).astype(np.float32)
    dup = np.concatenate([half, half], axis=1)[:, :N]                      # point k and k + ceil(N/2) coincide
    lattice = _cloud(2, N, seed=N + 7, grid=True)
    plain = _cloud(2, N, seed=N + 9)
    for xyz, M in ((dup, min(N, 48)), (lattice, min(N, 64)), (plain, N)):
        got = pointops.furthest_point_sample(torch.from_numpy(np.ascontiguousarray(xyz)).cuda(), M).cpu().numpy()
        assert np.array_equal(got, po.furthest_point_sampling(np.ascontiguousarray(xyz), M)), (N, M)


def test_fps_degenerate_clouds():
    """All points identical (every distance 0: the tie key alone decides, padding slots must never win), more samples than points,
    and sizes straddling every register-slot configuration of the kernel."""
    from unipre3d_amd import pointops
    same = np.zeros((2, 300, 3), np.float32) + np.float32(0.25)
    got = pointops.furthest_point_sample(torch.from_numpy(same).cuda(), 40).cpu().numpy()
    assert np.array_equal(got, po.furthest_point_sampling(same, 40))
    for N, M in ((255, 255), (256, 64), (257, 64), (513, 100), (1025, 100), (2049, 70), (4097, 50), (8192, 40), (8193, 20), (64, 64), (3, 3)):
        xyz = _cloud(2, N, seed=N + 1, grid=(N % 2 == 1))
        got = pointops.furthest_point_sample(torch.from_numpy(xyz).cuda(), M).cpu().numpy()
        assert np.array_equal(got, po.furthest_point_sampling(xyz, M)), (N, M)
