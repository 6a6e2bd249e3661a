"""Where do the fp32 and fp64 restatements part at a given shape?  Per view of one object: image distance, Gaussians whose visibility / radius differ,
and how close those sit to the z_view = 0.2 cull plane.  usage: python tools/diag_fp32_gap.py C4_fused 1"""
import os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from unipre3d_amd import synthetic
from conftest import rel_l2
import arbiter
from oracle import oracle as oracle_mod
oracle_mod.build()
name, bi = sys.argv[1], int(sys.argv[2])
cfg = synthetic.CONFIGS[name]
B, P, V, H, W, level = cfg["B"], cfg["P"], cfg["V"], cfg["H"], cfg["W"], cfg["level"]
b = synthetic.make_batch(B, P, V, H, W, level=level, seed=42)
g = synthetic.gaussians_from_batch(b)
for v in (list(map(int, sys.argv[3:])) or range(V)):
    r32, r64 = arbiter.oracle_view(oracle_mod, g, b, bi, v, H, W, np.float32), arbiter.oracle_view(oracle_mod, g, b, bi, v, H, W, np.float64)
    e = rel_l2(r32.color, r64.color)
    dif = np.flatnonzero(r32.radii != r64.radii)
    vis = np.flatnonzero((r32.radii > 0) != (r64.radii > 0))
    xyz = g["xyz"][bi].double().numpy()
    zv = xyz @ b.world_view[bi, v].double().numpy()[:3, 2] + b.world_view[bi, v].double().numpy()[3, 2]
    worst_px = np.abs(r32.color - r64.color).max()
    print(f"view {v}: image |f32-f64| {e:.2e} (max pixel diff {worst_px:.3f}); radii differ for {dif.size}, visibility for {vis.size}: ids {vis[:6].tolist()} "
          f"z_view(fp64) {[float('%.9f' % zv[i]) for i in vis[:6]]} scales {[np.round(g['scaling'][bi][i].numpy(), 2).tolist() for i in vis[:3]]}", flush=True)
    if e > 1e-4:
        # which Gaussian's projected state parts most between the two precisions, among those in front (smallest depths first)?
        c32, c64 = r32.conic_opacity.astype(np.float64), r64.conic_opacity
        visb = (r64.radii > 0) & (r32.radii > 0)
        rel = np.abs(c32 - c64).max(axis=1) / np.maximum(np.abs(c64).max(axis=1), 1e-300)
        rel[~visb] = 0
        order = np.argsort(-rel)[:5]
        d64 = r64.depths
        rank = np.argsort(np.argsort(np.where(visb, d64, np.inf)))
        # front of the depth order in both precisions: the first position where they part, and how close the two depths are
        o32 = np.lexsort((np.arange(P), np.where(r32.radii > 0, r32.depths.astype(np.float64), np.inf)))
        o64 = np.lexsort((np.arange(P), np.where(r64.radii > 0, r64.depths, np.inf)))
        first = int(np.flatnonzero(o32 != o64)[0]) if np.any(o32 != o64) else -1
        if first >= 0:
            a, c = o32[first], o64[first]
            print(f"    depth order parts at sorted position {first}: fp32 puts Gaussian {a} (z32 {r32.depths[a]:.9f}, z64 {r64.depths[a]:.12f}) before {c} "
                  f"(z32 {r32.depths[c]:.9f}, z64 {r64.depths[c]:.12f}); order differs at {int((o32[:200] != o64[:200]).sum())} of the first 200 positions; "
                  f"radii {r64.radii[a]}, {r64.radii[c]}; opacity x aa {r64.conic_opacity[a][3]:.3f}, {r64.conic_opacity[c][3]:.3f}", flush=True)
        for i in order[:2]:
            print(f"    Gaussian {i}: conic/opacity rel diff {rel[i]:.2e}, depth {d64[i]:.6f} (depth rank {rank[i]}), radius {r64.radii[i]}, scales {np.round(g['scaling'][bi][i].numpy(), 3).tolist()}, "
                  f"conic f32 {c32[i].tolist()} f64 {c64[i].tolist()}", flush=True)
    r32.close(); r64.close()
