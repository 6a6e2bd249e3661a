"""Ragged batches -- the reference's scene-level branch returns per-item LISTS of (M_i, .) tensors
(model/gaussian_predictor.py:331-364): sets of different sizes in ONE launch sequence (u3d_raster_desc.total_P / item_offsets)
must equal one call per set.  Also cfg.model.isotropic (:308-310) through the fused route."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
KEYS = ("xyz", "opacity", "scaling", "rotation")


def _sets(sizes, V, H, W, level, seed):
    from unipre3d_amd import synthetic
    dev = torch.device("cuda:0")
    out = []
    for k, n in enumerate(sizes):
        b = synthetic.make_batch(1, max(n, 1), V, H, W, level=level, seed=seed + k).to(dev)     # (cameras / targets; empty sets: see the test)
        out.append(b)
    return out


@pytest.mark.parametrize("sizes,level", [((100, 37, 256), "object"), ((300, 2048, 1000), "scene"), ((700, 5000, 1, 4200), "scene"),
                                         ((64, 0, 200), "scene"), ((0, 5000), "scene")])
def test_ragged_operator_equals_one_call_per_set(sizes, level):
    """Covers the three sort routes (largest set <= 256: fused into the projection kernel; <= 4096: one-workgroup LDS sort;
    beyond: bucketed), sets shorter than a workgroup's span, a one-Gaussian set and EMPTY sets (background-only views)."""
    from unipre3d_amd import head, synthetic
    from unipre3d_amd.rasterizer import pack_ragged, rasterize_gaussians_batched, split_ragged_radii
    V, H, W = 2, 64, 80
    bs = _sets(sizes, V, H, W, level, 40)
    t = math.tan(bs[0].fov_deg * math.pi / 360)
    def gaussians(b, n):
        if n == 0:      # (the reference's scene branch cannot produce an empty set; the operator must still accept one)
            sh = {"xyz": (0, 3), "opacity": (0, 1), "scaling": (0, 3), "rotation": (0, 4), "features_dc": (0, 1, 3), "features_rest": (0, 3, 3)}
            return {k: torch.zeros(v, device="cuda", requires_grad=True) for k, v in sh.items()}
        return {k: v[0].detach().clone().requires_grad_(True) for k, v in synthetic.gaussians_from_batch(b).items()}
    gs = [gaussians(b, n) for b, n in zip(bs, sizes)]
    cot = torch.randn(len(sizes), V, 3, H, W, generator=torch.Generator().manual_seed(2)).cuda()
    # one call per set
    ref = []
    for i, (b, g) in enumerate(zip(bs, gs)):
        shs = head.concat_sh(g["features_dc"], g["features_rest"])
        shs.retain_grad()
        m2d = torch.zeros(V, sizes[i], 3, device="cuda", requires_grad=True)
        col, radii, inv = rasterize_gaussians_batched(g["xyz"][None], g["opacity"][None], b.world_view, b.full_proj, b.camera_center, b.bg, H, W, t, t,
                                                      shs=shs[None], scales=g["scaling"][None], rotations=g["rotation"][None], sh_degree=1, means2D=m2d)
        (col * cot[i]).sum().backward()
        grad = lambda t: t.grad.clone() if t.grad is not None else torch.zeros_like(t)
        ref.append((col.detach(), radii, inv.detach(), {k: grad(g[k]) for k in KEYS}, grad(shs), grad(m2d)))
        for k in KEYS:
            g[k].grad = None
    # the same sets packed into ONE launch sequence
    packed = {k: pack_ragged([g[k] for g in gs]) for k in KEYS}
    off, szs = packed["xyz"][1], packed["xyz"][2]
    shs_p = torch.cat([head.concat_sh(g["features_dc"], g["features_rest"]) for g in gs])
    shs_p.retain_grad()
    m2d_p = torch.zeros(V * sum(sizes), 3, device="cuda", requires_grad=True)
    wv, fp, cc = (torch.cat([getattr(b, n) for b in bs]) for n in ("world_view", "full_proj", "camera_center"))
    col, radii, inv = rasterize_gaussians_batched(packed["xyz"][0], packed["opacity"][0], wv, fp, cc, bs[0].bg, H, W, t, t, shs=shs_p,
                                                  scales=packed["scaling"][0], rotations=packed["rotation"][0], sh_degree=1, means2D=m2d_p,
                                                  item_offsets=off, max_P=max(sizes), debug=True)
    (col * cot).sum().backward()
    torch.cuda.synchronize()
    r_list, m_list = split_ragged_radii(radii, szs, V), split_ragged_radii(m2d_p.grad, szs, V)
    o = 0
    for i, n in enumerate(sizes):
        assert torch.equal(col[i], ref[i][0][0]) and torch.equal(inv[i], ref[i][2][0]), i      # same forward arithmetic
        assert torch.equal(r_list[i], ref[i][1][0]), i
        for k in KEYS:       # (the cross-slice f64 atomics of the gradient reduction are order-insensitive, not bit-identical)
            a, b_ = gs[i][k].grad, ref[i][3][k]
            if n == 0:
                assert a is None or a.numel() == 0
                continue
            assert rel_l2(a.cpu().numpy(), b_.cpu().numpy()) < 1e-6 if b_.abs().sum() > 0 else not a.any(), (i, k)
        if n:
            assert rel_l2(shs_p.grad[o:o + n].cpu().numpy(), ref[i][4].cpu().numpy()) < 1e-6
            assert rel_l2(m_list[i].cpu().numpy(), ref[i][5].cpu().numpy()) < 1e-6
        o += n


@pytest.mark.parametrize("sizes,level,kind", [((128, 60, 200), "object", "focal_l2"), ((900, 5000, 2500), "scene", "l2")])
@pytest.mark.parametrize("single_pass", [True, False])
def test_ragged_fused_step_equals_one_call_per_set(sizes, level, kind, single_pass):
    from unipre3d_amd import fused
    from unipre3d_amd.rasterizer import pack_ragged
    V, H, W = 3, 48, 64
    bs = _sets(sizes, V, H, W, level, 60)
    B = len(sizes)
    ref_loss, ref_grad, ref_img = [], [], []
    for b in bs:
        h = b.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
        loss, img, _ = fused.render_loss_fused(h, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, H, W, level=level,
                                               offset_scale=b.offset_scale, loss_kind=kind, single_pass=single_pass)
        loss.backward()
        ref_loss.append(loss.item()); ref_grad.append(h.grad[0].clone()); ref_img.append(img)
    hp, off, szs = pack_ragged([b.raw[0].t().contiguous() for b in bs])
    hp = hp.detach().requires_grad_(True)
    cp = torch.cat([b.center[0] for b in bs])
    wv, fp, cc, gt = (torch.cat([getattr(b, n) for b in bs]) for n in ("world_view", "full_proj", "camera_center", "gt"))
    for rnd in range(2):                                     # the second round runs with U3D_FLAG_ACC_CLEAN on the kept workspace
        hp.grad = None
        loss, img, radii = fused.render_loss_fused(hp, cp, wv, fp, cc, gt, bs[0].bg, bs[0].fov_deg, H, W, level=level, offset_scale=bs[0].offset_scale,
                                                   loss_kind=kind, single_pass=single_pass, item_offsets=off, max_P=max(sizes), debug=True)
        loss.backward()
        torch.cuda.synchronize()
        assert abs(loss.item() - sum(ref_loss) / B) <= 1e-6 * abs(loss.item())
        assert radii.shape == (V * sum(sizes),)
        o = 0
        for i, n in enumerate(sizes):
            assert torch.equal(img[i * V:(i + 1) * V], ref_img[i]), (rnd, i)
            # the packed loss averages over B times as many pixels: every seed is scaled by 1/B in-kernel, i.e. every term of the
            # gradient sum is re-rounded (scene-level sums cancel ~1000x, so that alone moves the fp32 result by a few 1e-6)
            a, b_ = hp.grad[o:o + n] * B, ref_grad[i]
            assert rel_l2(a.cpu().numpy(), b_.cpu().numpy()) < 2e-5, (rnd, i, rel_l2(a.cpu().numpy(), b_.cpu().numpy()))
            o += n


def test_ragged_argument_errors():
    from unipre3d_amd import _lib, fused
    import ctypes
    lib = _lib.load()
    sizes = _lib.ScratchSizes()
    d = _lib.RasterDesc(2, 1, 100, 32, 32, 0.5, 0.5, 1.0, 1, 4, 2)
    d.total_P = 150                                          # ragged without prefix sums
    assert lib.u3d_scratch_query(ctypes.byref(d), ctypes.byref(sizes)) == 1
    d.item_offsets = 8
    assert lib.u3d_scratch_query(ctypes.byref(d), ctypes.byref(sizes)) == 0
    d.total_P = 250                                          # more than n_items * largest set
    assert lib.u3d_scratch_query(ctypes.byref(d), ctypes.byref(sizes)) == 1
    d.total_P = 50                                           # less than the largest set
    assert lib.u3d_scratch_query(ctypes.byref(d), ctypes.byref(sizes)) == 1
    h = torch.zeros(10, 23, device="cuda", requires_grad=True)
    with pytest.raises(ValueError):
        fused._batch_shape(h, torch.zeros(3, dtype=torch.int64, device="cuda"), 5)
    with pytest.raises(ValueError):
        fused._batch_shape(h, torch.zeros(3, dtype=torch.int32, device="cuda"), 0)


def test_malformed_prefix_sums_are_refused_in_debug_mode_and_harmless_otherwise():
    """The kernels trust `item_offsets`.  With debug=True the table is checked on the device before anything is launched
    (U3D_ERR_INVALID_ARGUMENT -> RuntimeError); without it a set that claims more than max_P Gaussians is truncated to max_P
    (u3d_set_span) and its unprojected pairs read as culled (radii start from zeros) -- finite results, no out-of-bounds access."""
    from unipre3d_amd import fused, synthetic
    dev = torch.device("cuda:0")
    V, H, W = 2, 32, 32
    bs = [synthetic.make_batch(1, n, V, H, W, level="scene", seed=70 + n).to(dev) for n in (300, 500)]
    cat = lambda k: torch.cat([getattr(b, k) for b in bs])
    hp = torch.cat([b.raw[0].t() for b in bs]).contiguous()
    args = (torch.cat([b.center[0] for b in bs]), cat("world_view"), cat("full_proj"), cat("camera_center"), cat("gt"), bs[0].bg, bs[0].fov_deg, H, W)
    kw = dict(level="scene", offset_scale=bs[0].offset_scale, loss_kind="l2")
    good = torch.tensor([0, 300, 800], dtype=torch.int32, device=dev)
    for off, max_P in ((torch.tensor([0, 300, 800], dtype=torch.int32, device=dev), 400),      # largest set understated
                       (torch.tensor([0, 500, 300], dtype=torch.int32, device=dev), 500),      # not monotone
                       (torch.tensor([5, 300, 800], dtype=torch.int32, device=dev), 500),      # does not start at 0
                       (torch.tensor([0, 300, 790], dtype=torch.int32, device=dev), 500)):     # does not end at the total
        for sp in (True, False):
            with pytest.raises(RuntimeError, match="invalid argument"):
                fused.render_loss_fused(hp.clone().requires_grad_(True), *args, single_pass=sp, item_offsets=off, max_P=max_P, debug=True, **kw)
    l_ok, _, _ = fused.render_loss_fused(hp.clone().requires_grad_(True), *args, item_offsets=good, max_P=500, debug=True, **kw)
    # understated largest set without debug: the second set is rendered from its first 400 Gaussians, the rest read as culled
    h = hp.clone().requires_grad_(True)
    l, img, radii = fused.render_loss_fused(h, *args, item_offsets=good, max_P=400, **kw)
    l.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(l) and torch.isfinite(img).all() and torch.isfinite(h.grad).all()
    # (the truncated set's pairs are laid out with P_i = 400: [V * 300, V * 300 + V * 400) written, the tail untouched zeros)
    assert radii[V * 300: V * 700].any() and not radii[V * 700:].any()
    assert h.grad[300:700].any() and not h.grad[700:].any()


@pytest.mark.parametrize("single_pass", [True, False])
def test_isotropic_head_variant(single_pass):
    """cfg.model.isotropic (model/gaussian_predictor.py:308-310): scaling[:, :1].expand(-1, 3, -1) -- the first scaling channel
    serves all three axes and receives the three gradients, the other two channels receive none."""
    from unipre3d_amd import fused, head, losses, renderer, synthetic
    dev = torch.device("cuda:0")
    b = synthetic.make_batch(2, 128, 3, 64, 64, seed=15).to(dev)
    h = b.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    loss, img, _ = fused.render_loss_fused(h, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, 64, 64,
                                           loss_kind="focal_l2", single_pass=single_pass, isotropic=True, debug=True)
    loss.backward()
    raw = b.raw.clone().requires_grad_(True)
    g = head.process_object_output(raw, b.center, 1.0, 1, isotropic=True)
    out = renderer.render_views(g, b.world_view, b.full_proj, b.camera_center, b.bg, b.fov_deg, 64, 64)
    lu = losses.render_loss(out, b.gt.reshape(-1, 3, 64, 64), "focal_l2")
    lu.backward()
    assert rel_l2(img.cpu().numpy(), out.detach().cpu().numpy()) < 1e-5 and abs(loss.item() - lu.item()) < 1e-6
    assert rel_l2(h.grad.cpu().numpy(), raw.grad.permute(0, 2, 1).cpu().numpy()) < 1e-4
    assert not h.grad[..., 5:7].any() and h.grad[..., 4].abs().sum() > 0
    # and it is a different render from the anisotropic one
    _, img_a, _ = fused.render_loss_fused(h.detach(), b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, 64, 64)
    assert rel_l2(img.cpu().numpy(), img_a.cpu().numpy()) > 1e-2


def test_fuzz_ragged_fused_step():
    """40 seeded draws: 1-4 sets of sizes from {1, 2, 40, 64, 65, 200, 256, 257, 900, 3000, 4096, 4097, 6000} (all three sort routes mixed
    inside one batch), 1-3 views, images 8..100 px, both levels, two losses, isotropic on / off, single-pass and two-pass: the packed
    fused step equals one fused call per set (image bit for bit; loss; gradient to the re-rounding of the 1/B seed)."""
    from unipre3d_amd import fused, synthetic
    from unipre3d_amd.rasterizer import pack_ragged
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(20260929)
    pool = [1, 2, 40, 64, 65, 200, 256, 257, 900, 3000, 4096, 4097, 6000]
    for it in range(40):
        B, V = int(rng.integers(1, 5)), int(rng.integers(1, 4))
        H, W = int(rng.integers(8, 100)), int(rng.integers(8, 100))
        sizes = [int(rng.choice(pool)) for _ in range(B)]
        level = ("object", "scene")[int(rng.integers(0, 2))]
        kind = ("focal_l2", "l2")[int(rng.integers(0, 2))]
        iso = bool(rng.integers(0, 2)) and level == "object"
        sp = bool(rng.integers(0, 2))
        bs = [synthetic.make_batch(1, n, V, H, W, level=level, seed=int(rng.integers(0, 1 << 30))).to(dev) for n in sizes]
        tag = (it, sizes, V, H, W, level, kind, iso, sp)
        ref_l, ref_g, ref_i = [], [], []
        for b in bs:
            h = b.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
            l, img, _ = fused.render_loss_fused(h, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, H, W, level=level,
                                                offset_scale=b.offset_scale, loss_kind=kind, single_pass=sp, isotropic=iso)
            l.backward()
            ref_l.append(l.item()); ref_g.append(h.grad[0].clone()); ref_i.append(img)
        hp, off, _ = pack_ragged([b.raw[0].t().contiguous() for b in bs])
        hp = hp.detach().requires_grad_(True)
        cat = lambda n: torch.cat([getattr(b, n) for b in bs])
        l, img, radii = fused.render_loss_fused(hp, torch.cat([b.center[0] for b in bs]), cat("world_view"), cat("full_proj"), cat("camera_center"),
                                                cat("gt"), bs[0].bg, bs[0].fov_deg, H, W, level=level, offset_scale=bs[0].offset_scale, loss_kind=kind,
                                                single_pass=sp, isotropic=iso, item_offsets=off, max_P=max(sizes))
        l.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(hp.grad).all() and abs(l.item() - sum(ref_l) / B) <= 2e-6 * max(abs(l.item()), 1e-12), tag
        o = 0
        for i, n in enumerate(sizes):
            assert torch.equal(img[i * V:(i + 1) * V], ref_i[i]), tag
            a, r = hp.grad[o:o + n] * B, ref_g[i]
            if r.abs().sum() > 0:
                assert rel_l2(a.cpu().numpy(), r.cpu().numpy()) < 1e-4, tag
            else:
                assert not a.any(), tag
            o += n


def test_render_views_takes_the_scene_branch_lists():
    """GaussianSplatPredictor's scene branch returns per-item LISTS of (M_i, .) tensors (model/gaussian_predictor.py:331-364); the
    reference's training loop cannot render those (`v.shape` on a list, train_network.py:423), its eval loop indexes them item by item
    (eval.py:97-99).  renderer.render_views packs them and renders every scene of the rank in one launch sequence: images equal the
    item-by-item renders bit for bit, and the gradient reaches the raw head output of every scene."""
    from unipre3d_amd import head, renderer, synthetic
    dev = torch.device("cuda:0")
    V, H, W = 3, 48, 80
    sizes = [700, 5000, 129]
    bs = [synthetic.make_batch(1, n, V, H, W, level="scene", seed=90 + i).to(dev) for i, n in enumerate(sizes)]
    raw = torch.cat([b.raw[0].t() for b in bs]).contiguous().requires_grad_(True)              # (sum M_i, 23) per-voxel head output
    center = torch.cat([b.center[0] for b in bs])
    idx = torch.cat([torch.full((n, 1), i, dtype=torch.long, device=dev) for i, n in enumerate(sizes)])
    lists = head.process_scene_output(raw, center, idx, bs[0].offset_scale, 1)
    assert isinstance(lists["xyz"], list) and [x.shape[0] for x in lists["xyz"]] == sizes
    cat = lambda k: torch.cat([getattr(b, k) for b in bs])
    out = renderer.render_views(lists, cat("world_view"), cat("full_proj"), cat("camera_center"), bs[0].bg, bs[0].fov_deg, H, W)
    assert out.shape == (len(sizes) * V, 3, H, W)
    for i, b in enumerate(bs):
        one = {k: v[i][None] for k, v in lists.items()}
        ref = renderer.render_views(one, b.world_view, b.full_proj, b.camera_center, b.bg, b.fov_deg, H, W)
        assert torch.equal(out[i * V:(i + 1) * V], ref), i
    gt = cat("gt").reshape(-1, 3, H, W)
    ((out - gt) ** 2).mean().backward()
    o = 0
    for n in sizes:
        assert torch.isfinite(raw.grad[o:o + n]).all() and raw.grad[o:o + n].abs().sum() > 0
        o += n
