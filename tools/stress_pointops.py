"""Randomised soak of the point operators against the CPU oracle: random shapes around every launch-shape boundary of the
HIP kernels (one-wave FPS up to 512 points, register-slot classes, LDS-row limits of the gradients, channel-block tails), in all three
contraction modes.  Indices and gathered values bit for bit; scatter-add gradients to fp32 summation ORDER: an element that sums n terms may differ
from the oracle's sequential sum by 4 x 2^-24 x sqrt(n) x (sum of the terms' magnitudes) -- the bound is evaluated per element with the oracle itself
(counts and magnitude sums through the same scatter), so single copies must be exact and crowded destinations get what their n allows.
TEST INFRASTRUCTURE (imports oracle/).   usage: python tools/stress_pointops.py [seed] [cases]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from oracle import pointops as po  # noqa: E402
from unipre3d_amd import pointops  # noqa: E402

EDGES = (1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 4096, 4097, 8192, 8193, 9000)


def cloud(rng, B, N, kind):
    if kind == 0:
        return rng.randn(B, N, 3).astype(np.float32)
    if kind == 1:                                                     # lattice: exact distance ties
        return rng.randint(-4, 5, (B, N, 3)).astype(np.float32) * np.float32(0.25)
    if kind == 2:                                                     # every point twice
        h = rng.randn(B, (N + 1) // 2, 3).astype(np.float32)
        return np.ascontiguousarray(np.concatenate([h, h], 1)[:, :N])
    return (rng.rand(B, N, 3).astype(np.float32) * np.float32(1e-3)) + np.float32(5.0)   # tight cluster far from the origin


def order_err(got, ref, mag, cnt):
    """max over elements of |got - ref| / (4 x 2^-24 x sqrt(n) x sum |terms|); elements without terms must be exactly 0, single terms exact copies."""
    got, ref, mag, cnt = (np.asarray(a, np.float64) for a in (got, ref, mag, cnt))
    bound = 4.0 * 2.0 ** -24 * np.sqrt(np.maximum(cnt, 1.0)) * mag
    d = np.abs(got - ref)
    if np.any((d > 0) & (cnt <= 1.0)):
        return float("inf")
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(d > 0, d / bound, 0.0)
    return float(r.max()) if r.size else 0.0


def main(seed=31, cases=120):
    rng = np.random.RandomState(seed)
    dev = torch.device("cuda:0")
    t0 = time.time()
    bad = []
    worst = 0.0
    n_cmp = 0
    for case in range(cases):
        mode = ("fma_llvm", "fma_chain", "none")[case % 3]
        po.set_contraction(mode); pointops.set_contraction(mode)
        B = int(rng.randint(1, 5))
        N = int(EDGES[rng.randint(len(EDGES))]) if rng.rand() < 0.7 else int(rng.randint(1, 6000))
        M = int(min(N, max(1, rng.randint(1, 700) if rng.rand() < 0.8 else N)))
        kind = int(rng.randint(4))
        xyz = cloud(rng, B, N, kind)
        tag = f"case {case} mode {mode} B {B} N {N} M {M} cloud {kind}"
        x = torch.from_numpy(xyz).to(dev)
        # furthest point sampling
        got = pointops.furthest_point_sample(x, M).cpu().numpy(); ref = po.furthest_point_sampling(xyz, M); n_cmp += 1
        if not np.array_equal(got, ref): bad.append(tag + " fps")
        new = np.ascontiguousarray(np.take_along_axis(xyz, ref[..., None].astype(np.int64), 1))
        # ball query + grouping (+ gradient)
        K = int(rng.choice([1, 3, 8, 16, 32, 64])); r = float(rng.choice([1e-4, 0.05, 0.3, 1.0, 50.0]))
        idx = pointops.ball_query(r, K, x, torch.from_numpy(new).to(dev)); iref = po.ball_query(r, K, xyz, new); n_cmp += 1
        if not np.array_equal(idx.cpu().numpy(), iref): bad.append(tag + f" ball_query r {r} K {K}")
        C = int(rng.choice([1, 3, 4, 5, 16, 37, 64, 130]))
        pts = rng.randn(B, C, N).astype(np.float32)
        f = torch.from_numpy(pts).to(dev).requires_grad_(True)
        out = pointops.grouping_operation(f, torch.from_numpy(iref).to(dev)); n_cmp += 1
        if not np.array_equal(out.detach().cpu().numpy(), po.group_points(pts, iref)): bad.append(tag + f" group C {C} K {K}")
        go = rng.randn(*out.shape).astype(np.float32)
        out.backward(torch.from_numpy(go).to(dev))
        gref = po.group_points_grad(go, iref, N); n_cmp += 1
        e = order_err(f.grad.cpu().numpy(), gref, po.group_points_grad(np.abs(go), iref, N), po.group_points_grad(np.ones_like(go[:, :1]), iref, N)); worst = max(worst, e)
        if e > 1.0: bad.append(tag + f" group_grad C {C} K {K} err {e:.2f} x the order bound")
        # three_nn + three_interpolate (+ gradient): unknown = the cloud, known = the sampled points
        if M >= 1:
            d, i3 = pointops.three_nn(x, torch.from_numpy(new).to(dev)); d2r, i3r = po.three_nn(xyz, new); n_cmp += 2
            if not np.array_equal(i3.cpu().numpy(), i3r): bad.append(tag + " three_nn idx")
            if not np.array_equal(d.cpu().numpy(), np.sqrt(d2r)): bad.append(tag + " three_nn dist")
            kf = rng.randn(B, C, M).astype(np.float32); w = rng.rand(B, N, 3).astype(np.float32)
            g = torch.from_numpy(kf).to(dev).requires_grad_(True)
            o3 = pointops.three_interpolate(g, torch.from_numpy(i3r).to(dev), torch.from_numpy(w).to(dev)); n_cmp += 1
            if not np.array_equal(o3.detach().cpu().numpy(), po.three_interpolate(kf, i3r, w)): bad.append(tag + f" three_interpolate C {C}")
            go3 = rng.randn(*o3.shape).astype(np.float32)
            o3.backward(torch.from_numpy(go3).to(dev))
            g3 = po.three_interpolate_grad(go3, i3r, w, M); n_cmp += 1
            e = order_err(g.grad.cpu().numpy(), g3, po.three_interpolate_grad(np.abs(go3), i3r, w, M), po.three_interpolate_grad(np.ones_like(go3[:, :1]), i3r, np.ones_like(w), M)); worst = max(worst, e)
            if e > 1.0: bad.append(tag + f" three_interpolate_grad C {C} err {e:.2f} x the order bound")
    # FeatureFusion z-buffer gather: random image shapes (H*W not a multiple of 4 too), crowded pixels, exact depth ties, points
    # behind the camera / outside / at NaN; forward bit for bit, gather-form backward against the numpy scatter-add
    from oracle import fusion_oracle as fo
    from unipre3d_amd import fusion
    fcases = max(10, cases // 3)
    for case in range(fcases):
        B = int(rng.randint(1, 5)); N = int(rng.choice([1, 7, 64, 128, 300, 1000])); C = int(rng.choice([1, 3, 8, 37, 96]))
        H = int(rng.choice([1, 7, 9, 16, 31, 64, 128])); W = int(rng.choice([1, 5, 16, 30, 64, 128]))
        spread = float(rng.choice([0.02, 0.3, 1.5]))                  # 0.02: nearly every point on a handful of pixels
        cp = np.concatenate([rng.randn(B, N, 2) * spread, 1.0 + rng.rand(B, N, 1), np.ones((B, N, 1))], -1).astype(np.float32)
        if N >= 7:
            cp[:, 1] = cp[:, 0]; cp[:, 5] = cp[:, 0]                  # exact ties (all winners)
            cp[:, 2, 2] = -0.5; cp[:, 3, 2] = 0.0; cp[:, 4, 0] = np.nan; cp[:, 6, :2] = 1e30
        f = float(max(1, min(H, W)))
        feat = rng.randn(B, C, H, W).astype(np.float32)
        tag = f"fusion case {case} B {B} N {N} C {C} H {H} W {W} spread {spread}"
        ft = torch.from_numpy(feat).to(dev).requires_grad_(True)
        mapped, sel = fusion._ZBufferGather.apply(torch.from_numpy(cp).to(dev), ft, f, f, H / 2.0, W / 2.0)
        mref, sref = fo.mapped_features(cp, feat, f, f, H / 2.0, W / 2.0); n_cmp += 2
        if not np.array_equal(sel.cpu().numpy(), sref): bad.append(tag + " sel")
        if not np.array_equal(mapped.detach().cpu().numpy(), mref): bad.append(tag + " mapped")
        w = rng.randn(B, N, C).astype(np.float32)
        junk = torch.full((B, C, H, W), float("nan"), device=dev); del junk
        (mapped * torch.from_numpy(w).to(dev)).sum().backward()
        gref = fo.mapped_grad(w, sref, B, C, H, W); got = ft.grad.cpu().numpy(); n_cmp += 1
        e = order_err(got, gref, fo.mapped_grad(np.abs(w), sref, B, C, H, W), fo.mapped_grad(np.ones_like(w[:, :, :1]), sref, B, 1, H, W)) if np.isfinite(got).all() else float("inf")
        worst = max(worst, e if np.isfinite(e) else 99.0)
        if not e <= 1.0: bad.append(tag + f" grad err {e:.2f} x the order bound")
    po.set_contraction("fma_llvm"); pointops.set_contraction("fma_llvm")
    print(f"point operators + fusion vs oracle: cases {cases} (seed {seed}), comparisons {n_cmp}, bad {len(bad)}, worst gradient difference {worst:.2f} x the summation-order bound ({time.time() - t0:.1f} s)")
    for b in bad[:20]:
        print("  BAD", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 31, int(sys.argv[2]) if len(sys.argv) > 2 else 120))
