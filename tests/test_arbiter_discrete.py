"""The arbiter's discrete semantics: the reference operator sorts on the fp32 depth's bits with index ties and
takes cull / radius / tile-rectangle decisions in fp32.  An fp64 evaluation that re-decides them is an arbiter of a different discrete
problem as soon as two fp32 depths tie; `oracle.forward(dtype=float64, discrete_from=r32)` keeps all continuous arithmetic in fp64
under the fp32 evaluation's integers.  CPU only."""
import numpy as np
import torch

from conftest import rel_l2
from unipre3d_amd import cameras


def _tie_scene(oracle_mod, H=48, W=64):
    """Two big, opaque, differently coloured Gaussians whose view depths are EQUAL in fp32 and ordered against their indices in fp64,
    in front of a few hundred ordinary ones."""
    rng = np.random.default_rng(5)
    c2w = np.eye(4); ang = 0.37
    c2w[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]); c2w[:3, 3] = [0.3, -0.1, -2.0]
    wv, fp, cam = cameras.assemble_camera(c2w[:3, :3].copy(), -c2w[:3, :3].T @ c2w[:3, 3], cameras.projection_matrix(0.2, 10.0, 1.0, 1.0))
    # candidates in a slab so thin that fp32 depths collide
    n = 4000
    cand = np.c_[rng.uniform(-0.2, 0.2, n), rng.uniform(-0.2, 0.2, n), rng.uniform(0.0, 3e-6, n)].astype(np.float32)
    V32, V64 = wv.numpy().astype(np.float32), wv.numpy().astype(np.float64)
    d32 = ((V32[0, 2] * cand[:, 0] + V32[1, 2] * cand[:, 1]) + V32[2, 2] * cand[:, 2]) + V32[3, 2]       # the restatement's operation order
    d64 = cand.astype(np.float64) @ V64[:3, 2] + V64[3, 2]
    order = np.argsort(d32, kind="stable")
    pair = None
    for a, b in zip(order[:-1], order[1:]):
        if d32[a] == d32[b] and abs(d64[a] - d64[b]) > 1e-9:
            lo, hi = (a, b) if d64[a] < d64[b] else (b, a)            # lo is nearer in fp64
            pair = (hi, lo)                                            # ... and gets the LARGER index: index order contradicts fp64 order
            break
    assert pair is not None, "no fp32 depth tie among the candidates"
    P = 300
    means = np.r_[cand[list(pair)], np.c_[rng.uniform(-1, 1, P - 2), rng.uniform(-1, 1, P - 2), rng.uniform(0.5, 2.0, P - 2)].astype(np.float32)]
    scales = np.r_[np.full((2, 3), 0.6, np.float32), np.exp(rng.normal(-2.5, 0.4, (P - 2, 3))).astype(np.float32)]
    rots = rng.normal(size=(P, 4)).astype(np.float32); rots /= np.linalg.norm(rots, axis=1, keepdims=True)
    opac = np.r_[np.array([0.6, 0.6], np.float32), rng.uniform(0.1, 0.9, P - 2).astype(np.float32)]
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32); cols[0] = [1, 0, 0]; cols[1] = [0, 0, 1]
    kw = dict(means3D=means, opacities=opac, viewmatrix=wv.numpy(), projmatrix=fp.numpy(), campos=cam.numpy(), bg=np.zeros(3, np.float32),
              image_height=H, image_width=W, tanfovx=float(np.tan(0.5)), tanfovy=float(np.tan(0.5)), colors_precomp=cols, scales=scales,
              rotations=rots)
    return kw


def test_fp64_under_fp32_discrete_decisions_resolves_a_depth_tie_like_the_reference(oracle_mod):
    kw = _tie_scene(oracle_mod)
    r32 = oracle_mod.forward(dtype=np.float32, **kw)
    assert r32.depths[0] == r32.depths[1] and r32.radii[0] > 0 and r32.radii[1] > 0                 # the tie is real in the restatement
    r64 = oracle_mod.forward(dtype=np.float64, **kw)                                               # re-decides in fp64: the other order
    assert r64.depths[0] > r64.depths[1]
    r64d = oracle_mod.forward(dtype=np.float64, discrete_from=r32, **kw)
    r64s = oracle_mod.forward(dtype=np.float64, discrete_from="fp32", **kw)
    assert rel_l2(r32.color, r64.color) > 1e-2                                                     # a different discrete problem
    assert rel_l2(r32.color, r64d.color) < 1e-5                                                    # the same one, resolved in fp64
    assert np.array_equal(r64d.color, r64s.color)
    assert np.array_equal(r64d.radii, r32.radii) and r64d.num_rendered == r32.num_rendered
    assert np.array_equal(r64d.point_list, r32.point_list) and np.array_equal(r64d.ranges, r32.ranges)
    d = np.random.default_rng(1).normal(size=r32.color.shape)
    g32, g64, g64d = oracle_mod.backward(r32, d.astype(np.float32)), oracle_mod.backward(r64, d), oracle_mod.backward(r64d, d)
    for k in ("means3D", "opacities", "scales", "rotations", "colors_precomp"):
        assert rel_l2(g32[k], g64d[k]) < 1e-4, (k, rel_l2(g32[k], g64d[k]))
    assert max(rel_l2(g32[k], g64[k]) for k in ("opacities", "colors_precomp")) > 1e-2
    for r in (r32, r64, r64d, r64s):
        r.close()


def test_discrete_mode_is_the_identity_where_fp32_and_fp64_decide_alike(oracle_mod):
    from unipre3d_amd import synthetic
    sc = synthetic.single_view_scene(P=300, H=48, W=64, seed=3, level="object", sh_degree=1)
    kw = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in sc.items()}
    r32 = oracle_mod.forward(dtype=np.float32, **kw)
    r64, r64d = oracle_mod.forward(dtype=np.float64, **kw), oracle_mod.forward(dtype=np.float64, discrete_from=r32, **kw)
    if np.array_equal(r64.radii, r32.radii) and np.array_equal(r64.point_list, r32.point_list):
        assert np.array_equal(r64.color, r64d.color)
    assert rel_l2(r64d.color, r64.color) < 1e-6
