"""Further parity cases of the batched / ragged operator against the oracle (fp32 restatement + fp64 arbiter, tests/arbiter.py):
settings the reference's shipped configs do not use but the operator's interface carries -- SH degrees 2 and 3, scale_modifier != 1,
antialiasing off, precomputed colours and 3D covariances -- at batch shapes (several sets x several views in one launch sequence),
through all three sort routes and as a ragged batch."""
import math

import numpy as np
import pytest
import torch

from arbiter import assert_parity, assert_radii
from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _sets(sizes, V, H, W, level, deg, seed):
    """Per set: operator-layout tensors (means3D, opacities, scales, rotations, shs (P, M, 3)) + cameras."""
    from unipre3d_amd import synthetic
    out = []
    for k, n in enumerate(sizes):
        b = synthetic.make_batch(1, n, V, H, W, level=level, seed=seed + k)
        g = synthetic.gaussians_from_batch(b)
        gen = torch.Generator().manual_seed(seed + 100 + k)
        M = (deg + 1) ** 2
        shs = torch.randn(n, M, 3, generator=gen) * (0.5 if deg else 1.0)
        shs[:, 0] = g["features_dc"][0, :, 0]
        out.append(dict(b=b, means3D=g["xyz"][0], opacities=g["opacity"][0], scales=g["scaling"][0], rotations=g["rotation"][0], shs=shs))
    return out


def _oracle(oracle_mod, s, v, H, W, t, deg, dtype, **kw):
    b = s["b"]
    n = lambda x: np.ascontiguousarray(x.detach().cpu().numpy().astype(dtype))
    args = dict(shs=n(s["shs"]), scales=n(s["scales"]), rotations=n(s["rotations"]))
    args.update({k: (n(x) if torch.is_tensor(x) else x) for k, x in kw.pop("override", {}).items()})
    for k in kw.pop("drop", ()):
        args.pop(k)
    return oracle_mod.forward(n(s["means3D"]), n(s["opacities"]), n(b.world_view[0, v]), n(b.full_proj[0, v]), n(b.camera_center[0, v]), n(b.bg), H, W,
                              t, t, sh_degree=deg, dtype=dtype, discrete_from=("fp32" if np.dtype(dtype) == np.float64 else None), **args, **kw)


@pytest.mark.parametrize("sizes,level,deg,mod,aa,ragged", [
    ((200, 200, 200), "object", 3, 1.0, True, False),        # fused 256-key sort, SH degree 3
    ((1500, 1500), "scene", 2, 0.7, True, False),            # one-workgroup LDS sort, scale_modifier != 1
    ((5000, 5000), "scene", 1, 1.3, False, False),           # bucketed sort, antialiasing off
    ((700, 90, 4500), "scene", 3, 0.8, True, True),          # ragged, all of the above at once
])
def test_batched_operator_settings_vs_oracle(oracle_mod, sizes, level, deg, mod, aa, ragged):
    from unipre3d_amd.rasterizer import pack_ragged, rasterize_gaussians_batched, split_ragged_radii
    V, H, W = 2, 72, 88
    dev = torch.device("cuda:0")
    sets = _sets(sizes, V, H, W, level, deg, 300)
    t = math.tan(sets[0]["b"].fov_deg * math.pi / 360)
    keys = ("means3D", "opacities", "scales", "rotations", "shs")
    leaf = [{k: s[k].to(dev).requires_grad_(True) for k in keys} for s in sets]
    cam = lambda n: torch.cat([getattr(s["b"], n) for s in sets]).to(dev)
    if ragged:
        packed = {k: torch.cat([l[k] for l in leaf]) for k in keys}
        _, off, szs = pack_ragged([l["means3D"] for l in leaf])
        extra = dict(item_offsets=off, max_P=max(sizes))
    else:
        packed = {k: torch.stack([l[k] for l in leaf]) for k in keys}
        extra = {}
    color, radii, invd = rasterize_gaussians_batched(packed["means3D"], packed["opacities"], cam("world_view"), cam("full_proj"), cam("camera_center"),
                                                     sets[0]["b"].bg.to(dev), H, W, t, t, shs=packed["shs"], scales=packed["scales"],
                                                     rotations=packed["rotations"], sh_degree=deg, scale_modifier=mod, antialiasing=aa, debug=True, **extra)
    gen = torch.Generator().manual_seed(9)
    # one view per set carries a cotangent (colour AND inverse depth), so that each set's gradient is that view's
    pick = [int(torch.randint(0, V, (1,), generator=gen)) for _ in sizes]
    dcol = [torch.randn(3, H, W, generator=gen) for _ in sizes]
    dinv = [torch.randn(1, H, W, generator=gen) for _ in sizes]
    cot_c, cot_i = torch.zeros_like(color), torch.zeros_like(invd)
    for i, v in enumerate(pick):
        cot_c[i, v], cot_i[i, v] = dcol[i].to(dev), dinv[i].to(dev)
    ((color * cot_c).sum() + (invd * cot_i).sum()).backward()
    torch.cuda.synchronize()
    r_list = split_ragged_radii(radii, sizes, V) if ragged else [radii[i] for i in range(len(sizes))]
    for i, (s, v) in enumerate(zip(sets, pick)):
        r32 = _oracle(oracle_mod, s, v, H, W, t, deg, np.float32, scale_modifier=mod, antialiasing=aa)
        r64 = _oracle(oracle_mod, s, v, H, W, t, deg, np.float64, scale_modifier=mod, antialiasing=aa)
        assert_radii(r_list[i][v].cpu().numpy(), r32.radii, r64.radii, f"radii set {i}")
        assert_parity(color[i, v].detach().cpu().numpy(), r32.color, r64.color, f"image set {i}")
        assert_parity(invd[i, v].detach().cpu().numpy(), r32.invdepth, r64.invdepth, f"inverse depth set {i}")
        g32 = oracle_mod.backward(r32, dcol[i].numpy(), dinv[i].numpy())
        g64 = oracle_mod.backward(r64, dcol[i].numpy().astype(np.float64), dinv[i].numpy().astype(np.float64))
        for k in keys:
            assert_parity(leaf[i][k].grad.cpu().numpy(), g32[k], g64[k], f"d{k} set {i} (deg {deg}, modifier {mod}, aa {aa})")


@pytest.mark.parametrize("ragged", [False, True])
def test_batched_precomputed_colours_and_covariances_vs_oracle(oracle_mod, ragged):
    from oracle import torch_ref
    from unipre3d_amd.rasterizer import pack_ragged, rasterize_gaussians_batched
    V, H, W = 3, 64, 80
    sizes = (300, 120, 2600) if ragged else (300, 300)
    dev = torch.device("cuda:0")
    sets = _sets(sizes, V, H, W, "scene", 0, 500)
    t = math.tan(sets[0]["b"].fov_deg * math.pi / 360)
    for k, s in enumerate(sets):
        Rm = torch_ref.quat_to_rotmat_raw(s["rotations"].double())
        Mx = Rm * s["scales"].double()[:, None, :]
        S = (Mx @ Mx.transpose(1, 2)).float()
        s["cov"] = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1).contiguous()
        s["colors"] = torch.rand(s["means3D"].shape[0], 3, generator=torch.Generator().manual_seed(k))
    keys = ("means3D", "opacities", "cov", "colors")
    leaf = [{k: s[k].to(dev).requires_grad_(True) for k in keys} for s in sets]
    cam = lambda n: torch.cat([getattr(s["b"], n) for s in sets]).to(dev)
    join = torch.cat if ragged else torch.stack
    packed = {k: join([l[k] for l in leaf]) for k in keys}
    extra = dict(item_offsets=pack_ragged([l["means3D"] for l in leaf])[1], max_P=max(sizes)) if ragged else {}
    color, radii, invd = rasterize_gaussians_batched(packed["means3D"], packed["opacities"], cam("world_view"), cam("full_proj"), cam("camera_center"),
                                                     sets[0]["b"].bg.to(dev), H, W, t, t, colors_precomp=packed["colors"], cov3D_precomp=packed["cov"],
                                                     debug=True, **extra)
    gen = torch.Generator().manual_seed(4)
    dcol = [torch.randn(3, H, W, generator=gen) for _ in sizes]
    cot = torch.zeros_like(color)
    for i in range(len(sizes)):
        cot[i, i % V] = dcol[i].to(dev)
    (color * cot).sum().backward()
    for i, s in enumerate(sets):
        v = i % V
        ov = dict(override=dict(colors_precomp=s["colors"], cov3D_precomp=s["cov"]), drop=("shs", "scales", "rotations"))
        r32 = _oracle(oracle_mod, s, v, H, W, t, 0, np.float32, **dict(ov)), _oracle(oracle_mod, s, v, H, W, t, 0, np.float64, **dict(ov))
        r32, r64 = r32
        assert_parity(color[i, v].detach().cpu().numpy(), r32.color, r64.color, f"image set {i}")
        g32, g64 = oracle_mod.backward(r32, dcol[i].numpy()), oracle_mod.backward(r64, dcol[i].numpy().astype(np.float64))
        for k, ok in (("means3D", "means3D"), ("opacities", "opacities"), ("cov", "cov3D_precomp"), ("colors", "colors_precomp")):
            assert_parity(leaf[i][k].grad.cpu().numpy(), g32[ok], g64[ok], f"d{ok} set {i}")


def _step_via_cabi(bd, H, W, level, flags, prefill):
    """u3d_render_loss_step (one call: forward half + backward half) through ctypes; d_head_out starts as `prefill`."""
    import ctypes
    import math
    from unipre3d_amd import _lib
    from unipre3d_amd.rasterizer import _Plan
    dev = bd.raw.device
    B, C, P = bd.raw.shape
    V = bd.world_view.shape[1]
    NV = B * V
    t = math.tan(bd.fov_deg * math.pi / 360)
    plan = _Plan(B, V, P, H, W, t, t, 1.0, 1, 4, flags)
    hd = _lib.HeadDesc(1 if level == "object" else 2, C, bd.offset_scale, 0)
    ld = _lib.LossDesc(_lib.LOSS_KINDS["l2"], 4.0, 1.0)
    head_out = bd.raw.permute(0, 2, 1).contiguous()
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
    geom, binning, fused_s, bwd = u8(plan.sizes.geom_bytes), u8(plan.sizes.binning_bytes), u8(plan.sizes.fused_bytes), torch.zeros(plan.sizes.backward_bytes, dtype=torch.uint8, device=dev)
    radii = torch.zeros(NV, P, dtype=torch.int32, device=dev)
    loss = torch.zeros((), device=dev)
    d_head = torch.full_like(head_out, prefill)
    p, c = _lib.ptr, lambda x: x.contiguous()
    rc = _lib.load().u3d_render_loss_step(ctypes.byref(plan.desc), ctypes.byref(hd), ctypes.byref(ld), p(bd.bg), p(head_out), p(c(bd.center)),
                                          p(c(bd.world_view).reshape(NV, 16)), p(c(bd.full_proj).reshape(NV, 16)), p(c(bd.camera_center).reshape(NV, 3)),
                                          p(c(bd.gt).reshape(NV, 3, H, W)), p(None), p(radii), p(loss), p(d_head), p(geom), p(binning), p(fused_s), p(bwd),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "u3d_render_loss_step")
    torch.cuda.synchronize()
    return loss.item(), d_head, geom, plan


@pytest.mark.parametrize("P,level", [(6000, "scene"), (3000, "scene"), (6000, "object")])
def test_sparse_backward_flag_equals_the_dense_chain_rule(P, level):
    """U3D_FLAG_SPARSE_BWD (ABI 5): the forward half zero-fills d_head_out beside its gradient reduction and lists the Gaussians that
    received a gradient; the backward half runs the chain rule over that list only.  Same loss and the same d loss / d head_out as the
    dense chain rule (which visits every Gaussian) -- from a buffer pre-filled with garbage, so every row must have been written by
    one of the two -- at a size where the flag is honoured (scene-level head, P > 4096) and where it is ignored (small P; object head)."""
    from unipre3d_amd import _lib, synthetic
    H, W = 96, 128
    bd = synthetic.make_batch(2, P, 3, H, W, level=level, seed=23).to(torch.device("cuda:0"))
    l0, g0, _, _ = _step_via_cabi(bd, H, W, level, _lib.FLAG_ANTIALIASING, 7.0)
    l1, g1, geom, plan = _step_via_cabi(bd, H, W, level, _lib.FLAG_ANTIALIASING | _lib.FLAG_SPARSE_BWD, -3.0)
    assert l0 == l1
    assert torch.equal(g0 == 0, g1 == 0)                                   # the same rows are exact zeros
    assert rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) < 1e-6               # (cross-slice f64 atomics: order-insensitive, not bit-identical)
    touched_rows = int((g0.abs().sum(dim=-1) > 0).sum().item())
    assert 0 < touched_rows <= 2 * P
    if level == "scene" and P > 4096:
        # the list the backward half walked: every touched Gaussian exactly once
        al = lambda n: ((n + 255) // 256) * 256
        NG, NV, tot = 3 * 2 * P, 6, 2 * P
        o = al(4 * NG) + al(8 * NG) + al(16 * NG) + al(16 * NG) + al(8 * NG) + al(4 * NG) + al(4 * NV) + al(4 * ((tot + 31) // 32 + 1))
        lst = geom[o:o + 8 * tot].view(torch.int32).reshape(tot, 2)
        cnt = int(geom[o + al(8 * tot):][:4].view(torch.int32).item())
        ids = (lst[:cnt, 0].long() * P + lst[:cnt, 1].long()).cpu().numpy()
        assert cnt == len(set(ids.tolist())) and cnt >= touched_rows
        rows = np.flatnonzero((g0.abs().sum(dim=-1) > 0).reshape(-1).cpu().numpy())
        assert set(rows.tolist()) <= set(ids.tolist())


@pytest.mark.parametrize("fwd_sparse,bwd_sparse", [(False, True), (True, False)])
def test_sparse_flag_in_one_half_only_still_gives_the_dense_gradient(fwd_sparse, bwd_sparse):
    """U3D_FLAG_SPARSE_BWD set in ONE half of the step.  Forward without / backward with used to return U3D_OK and an
    uninitialised gradient (no list was built); now the forward half leaves the count at U3D_TOUCHED_NOT_LISTED and the backward half walks
    its grid densely.  Forward with / backward without was always fine (the dense form writes every row).  Both from a garbage-filled buffer."""
    import ctypes
    import math
    from unipre3d_amd import _lib, synthetic
    from unipre3d_amd.rasterizer import _Plan
    dev = torch.device("cuda:0")
    H, W, P, level = 96, 128, 5000, "scene"
    bd = synthetic.make_batch(2, P, 3, H, W, level=level, seed=29).to(dev)
    l0, g0, _, _ = _step_via_cabi(bd, H, W, level, _lib.FLAG_ANTIALIASING, 7.0)
    B, C, _ = bd.raw.shape
    V = bd.world_view.shape[1]
    NV = B * V
    t = math.tan(bd.fov_deg * math.pi / 360)
    fl = lambda sp: _lib.FLAG_ANTIALIASING | (_lib.FLAG_SPARSE_BWD if sp else 0)
    pf, pb = _Plan(B, V, P, H, W, t, t, 1.0, 1, 4, fl(fwd_sparse)), _Plan(B, V, P, H, W, t, t, 1.0, 1, 4, fl(bwd_sparse))
    hd = _lib.HeadDesc(2, C, bd.offset_scale, 0)
    ld = _lib.LossDesc(_lib.LOSS_KINDS["l2"], 4.0, 1.0)
    head_out = bd.raw.permute(0, 2, 1).contiguous()
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
    geom, binning, fused_s = u8(pf.sizes.geom_bytes), u8(pf.sizes.binning_bytes), u8(pf.sizes.fused_bytes)
    bwd = torch.zeros(pf.sizes.backward_bytes, dtype=torch.uint8, device=dev)
    radii = torch.zeros(NV, P, dtype=torch.int32, device=dev)
    loss = torch.zeros((), device=dev)
    d_head = torch.full_like(head_out, -5.0)
    p, c = _lib.ptr, lambda x: x.contiguous()
    cams = (p(c(bd.world_view).reshape(NV, 16)), p(c(bd.full_proj).reshape(NV, 16)), p(c(bd.camera_center).reshape(NV, 3)))
    strm = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = _lib.load()
    _lib.check(lib.u3d_render_loss_step_forward(ctypes.byref(pf.desc), ctypes.byref(hd), ctypes.byref(ld), p(bd.bg), p(head_out), p(c(bd.center)), *cams,
                                                p(c(bd.gt).reshape(NV, 3, H, W)), p(None), p(radii), p(loss), p(geom), p(binning), p(fused_s), p(bwd),
                                                p(d_head), strm), "u3d_render_loss_step_forward")
    _lib.check(lib.u3d_render_loss_step_backward(ctypes.byref(pb.desc), ctypes.byref(hd), p(head_out), p(c(bd.center)), *cams, p(radii), p(None),
                                                 p(geom), p(binning), p(fused_s), p(bwd), p(d_head), strm), "u3d_render_loss_step_backward")
    torch.cuda.synchronize()
    assert loss.item() == l0
    assert torch.equal(g0 == 0, d_head == 0)
    assert rel_l2(d_head.cpu().numpy(), g0.cpu().numpy()) < 1e-6
    # a misaligned / missing gradient buffer with the flag is refused BEFORE anything is enqueued: the accumulators stay zero
    if fwd_sparse:
        bwd.zero_()
        rc = lib.u3d_render_loss_step_forward(ctypes.byref(pf.desc), ctypes.byref(hd), ctypes.byref(ld), p(bd.bg), p(head_out), p(c(bd.center)), *cams,
                                              p(c(bd.gt).reshape(NV, 3, H, W)), p(None), p(radii), p(loss), p(geom), p(binning), p(fused_s), p(bwd),
                                              p(None), strm)
        torch.cuda.synchronize()
        assert rc == 1 and not bool(bwd.any())


def test_fused_step_under_no_grad_and_inference_mode_then_training():
    """(i) an eval pass under torch.inference_mode() BEFORE training must not poison the cached viewspace sink of
    render_predicted; (ii) the scene-level fused step under no_grad allocates no gradient buffer and a later training step is unaffected."""
    import types
    from unipre3d_amd import fused, renderer, synthetic
    dev = torch.device("cuda:0")
    H, W, P = 64, 64, 777
    bd = synthetic.make_batch(1, P, 1, H, W, level="object", seed=3).to(dev)
    g = synthetic.gaussians_from_batch(bd)
    cfg = types.SimpleNamespace(data=types.SimpleNamespace(fov=bd.fov_deg, training_resolution=H), model=types.SimpleNamespace(max_sh_degree=1))
    pc = {k: v[0].contiguous() for k, v in g.items()}
    with torch.inference_mode():
        e = renderer.render_predicted(pc, bd.world_view[0, 0], bd.full_proj[0, 0], bd.camera_center[0, 0], bd.bg, cfg)["render"]
    pc_t = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    out = renderer.render_predicted(pc_t, bd.world_view[0, 0], bd.full_proj[0, 0], bd.camera_center[0, 0], bd.bg, cfg)
    out["render"].sum().backward()
    assert torch.equal(out["render"].detach(), e) and pc_t["xyz"].grad is not None and out["viewspace_points"].grad is not None
    bs = synthetic.make_batch(1, 5000, 2, H, W, level="scene", seed=4).to(dev)
    h = bs.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    args = (bs.center, bs.world_view, bs.full_proj, bs.camera_center, bs.gt, bs.bg, bs.fov_deg, H, W)
    kw = dict(level="scene", offset_scale=bs.offset_scale, loss_kind="l2", return_images=False)
    with torch.no_grad():
        l_eval, _, _ = fused.render_loss_fused(h, *args, **kw)
    l_train, _, _ = fused.render_loss_fused(h, *args, **kw)
    l_train.backward()
    assert l_eval.item() == l_train.item() and h.grad is not None and bool(torch.isfinite(h.grad).all()) and float(h.grad.abs().sum()) > 0


def test_render_view_entry_points_equal_the_operator_entry_points(oracle_mod):
    """u3d_render_view_forward / _backward (ABI 5: features_dc / features_rest through two pointers, visibility from the projection
    kernel, no inverse-depth plane) against u3d_rasterize_forward / _backward on the concatenated SH tensor, through ctypes, for a
    batched call (2 sets x 3 views) -- bit-identical image and radii, gradients equal (deterministic fixed-order route)."""
    import ctypes
    import math
    from unipre3d_amd import _lib, head, synthetic
    from unipre3d_amd.rasterizer import _Plan
    dev = torch.device("cuda:0")
    B, P, V, H, W = 2, 200, 3, 64, 80
    bd = synthetic.make_batch(B, P, V, H, W, level="object", seed=31).to(dev)
    g = synthetic.gaussians_from_batch(bd)
    NV = B * V
    t = math.tan(bd.fov_deg * math.pi / 360)
    plan = _Plan(B, V, P, H, W, t, t, 1.0, 1, 4, _lib.FLAG_ANTIALIASING)
    p, c = _lib.ptr, lambda x: x.contiguous()
    shs = c(head.concat_sh(g["features_dc"], g["features_rest"]))
    dc, rest = c(g["features_dc"]), c(g["features_rest"])
    cams = (p(c(bd.world_view).reshape(NV, 16)), p(c(bd.full_proj).reshape(NV, 16)), p(c(bd.camera_center).reshape(NV, 3)))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dcol = torch.randn(NV, 3, H, W, generator=torch.Generator().manual_seed(4)).to(dev)
    u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)
    lib = _lib.load()
    xyz, op, sc, rot = c(g["xyz"]), c(g["opacity"]), c(g["scaling"]), c(g["rotation"])

    def grads():
        return {k: torch.full(s, 5.0, device=dev) for k, s in (("xyz", (B, P, 3)), ("m2d", (NV, P, 3)), ("shs", (B, P, 4, 3)), ("dc", (B, P, 1, 3)),
                                                               ("rest", (B, P, 3, 3)), ("op", (B, P, 1)), ("sc", (B, P, 3)), ("rot", (B, P, 4)))}
    # operator entry points
    ca, ia, ra = torch.empty(NV, 3, H, W, device=dev), torch.empty(NV, 1, H, W, device=dev), torch.zeros(NV, P, dtype=torch.int32, device=dev)
    sa = [u8(plan.sizes.geom_bytes), u8(plan.sizes.binning_bytes), u8(plan.sizes.image_bytes), u8(plan.sizes.backward_bytes)]
    _lib.check(lib.u3d_rasterize_forward(ctypes.byref(plan.desc), p(bd.bg), p(xyz), p(shs), p(None), p(op), p(sc), p(rot), p(None), *cams, p(ca), p(ia),
                                         p(ra), p(sa[0]), p(sa[1]), p(sa[2]), stream), "fwd")
    ga = grads()
    _lib.check(lib.u3d_rasterize_backward(ctypes.byref(plan.desc), p(bd.bg), p(xyz), p(shs), p(None), p(op), p(sc), p(rot), p(None), *cams, p(ra), p(dcol),
                                          p(None), p(sa[0]), p(sa[1]), p(sa[2]), p(sa[3]), p(ga["xyz"]), p(ga["m2d"]), p(ga["shs"]), p(None), p(ga["op"]),
                                          p(ga["sc"]), p(ga["rot"]), p(None), stream), "bwd")
    # the wrapper-body entry points
    cb, rb, vb = torch.empty(NV, 3, H, W, device=dev), torch.zeros(NV, P, dtype=torch.int32, device=dev), torch.full((NV, P), 9, dtype=torch.uint8, device=dev)
    sb = [u8(plan.sizes.geom_bytes), u8(plan.sizes.binning_bytes), u8(plan.sizes.image_bytes), u8(plan.sizes.backward_bytes)]
    _lib.check(lib.u3d_render_view_forward(ctypes.byref(plan.desc), p(bd.bg), p(xyz), p(dc), p(rest), p(op), p(sc), p(rot), *cams, p(cb), p(rb), p(vb),
                                           p(sb[0]), p(sb[1]), p(sb[2]), stream), "view fwd")
    gb = grads()
    _lib.check(lib.u3d_render_view_backward(ctypes.byref(plan.desc), p(bd.bg), p(xyz), p(dc), p(rest), p(op), p(sc), p(rot), *cams, p(rb), p(dcol),
                                            p(sb[0]), p(sb[1]), p(sb[2]), p(sb[3]), p(gb["xyz"]), p(gb["m2d"]), p(gb["dc"]), p(gb["rest"]), p(gb["op"]),
                                            p(gb["sc"]), p(gb["rot"]), stream), "view bwd")
    torch.cuda.synchronize()
    assert torch.equal(ca, cb) and torch.equal(ra, rb) and torch.equal(vb.bool(), ra > 0)
    for k in ("xyz", "m2d", "op", "sc", "rot"):
        assert torch.equal(ga[k], gb[k]), k
    assert torch.equal(ga["shs"][:, :, :1], gb["dc"]) and torch.equal(ga["shs"][:, :, 1:], gb["rest"])
    # degree 0: no features_rest at all
    plan0 = _Plan(B, V, P, H, W, t, t, 1.0, 0, 1, _lib.FLAG_ANTIALIASING)
    c0, r0 = torch.empty(NV, 3, H, W, device=dev), torch.zeros(NV, P, dtype=torch.int32, device=dev)
    _lib.check(lib.u3d_render_view_forward(ctypes.byref(plan0.desc), p(bd.bg), p(xyz), p(dc), p(None), p(op), p(sc), p(rot), *cams, p(c0), p(r0), p(None),
                                           p(sb[0]), p(sb[1]), p(sb[2]), stream), "view fwd deg 0")
    c1, i1, r1 = torch.empty_like(c0), torch.empty(NV, 1, H, W, device=dev), torch.zeros_like(r0)
    _lib.check(lib.u3d_rasterize_forward(ctypes.byref(plan0.desc), p(bd.bg), p(xyz), p(dc), p(None), p(op), p(sc), p(rot), p(None), *cams, p(c1), p(i1),
                                         p(r1), p(sa[0]), p(sa[1]), p(sa[2]), stream), "fwd deg 0")
    torch.cuda.synchronize()
    assert torch.equal(c0, c1) and torch.equal(r0, r1)
    assert lib.u3d_render_view_forward(ctypes.byref(plan.desc), p(bd.bg), p(xyz), p(dc), p(None), p(op), p(sc), p(rot), *cams, p(cb), p(rb), p(vb),
                                       p(sb[0]), p(sb[1]), p(sb[2]), stream) == 1          # M = 4 without features_rest: invalid argument


def test_general_loop_variant_on_a_plain_scene_agrees_with_the_plain_variant(tmp_path):
    """since alpha_run the tile kernels' two loop variants are not bit-identical (PLAIN: alpha from the forward-differenced
    recurrence; general: exp2(pw) per pixel).  A build that sends EVERY tile through the general variant (-DU3D_FORCE_GENERAL) must agree with
    the product build on a scene whose tiles all qualify for PLAIN: images to 1e-5, gradients to 1e-4 (a 1/255 threshold decided differently
    by a forward and its backward would desynchronise stop_pos from the recomputed alphas and show as an O(1) gradient error), radii equal --
    in the single-pass kernel, the two-pass kernels (which hand the variant over through U3D_TILE_PLAIN_BIT) and the operator path."""
    import os
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc, var = os.path.join(root, "unipre3d_amd", "csrc"), os.path.join(root, "unipre3d_amd", "lib_general")
    lib = os.path.join(var, "libunipre3d_rasterizer.so")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        res = subprocess.run(["make", "-C", csrc, "-j4", "LIBDIR=../lib_general", "EXTRA=-DU3D_FORCE_GENERAL", "../lib_general/libunipre3d_rasterizer.so"],
                             capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
    shutil.copy2(os.path.join(root, "unipre3d_amd", "lib", "_u3d_torch.so"), os.path.join(var, "_u3d_torch.so"))   # (rpath $ORIGIN: binds the variant)
    outs = {}
    for name, dirname in (("plain", "lib"), ("general", "lib_general")):
        path = str(tmp_path / f"{name}.npz")
        res = subprocess.run([sys.executable, os.path.join(root, "tests", "variant_dump.py"), path], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, U3D_LIB_DIRNAME=dirname))
        assert res.returncode == 0, res.stderr[-2000:]
        outs[name] = np.load(path)
    a, g = outs["plain"], outs["general"]
    for route in ("single", "two"):
        assert np.array_equal(a[route + "_radii"], g[route + "_radii"])
        assert rel_l2(g[route + "_img"], a[route + "_img"]) < 1e-5 and abs(g[route + "_loss"] - a[route + "_loss"]) < 1e-5 * abs(a[route + "_loss"])
        assert rel_l2(g[route + "_grad"], a[route + "_grad"]) < 1e-4, (route, rel_l2(g[route + "_grad"], a[route + "_grad"]))
    assert np.array_equal(a["single_img"], a["two_img"]) and np.array_equal(g["single_img"], g["two_img"])     # within a build the routes share the forward
    assert rel_l2(g["op_img"], a["op_img"]) < 1e-5
    for k in ("xyz", "opacity", "scaling", "rotation"):
        assert rel_l2(g["op_d" + k], a["op_d" + k]) < 1e-4, (k, rel_l2(g["op_d" + k], a["op_d" + k]))
    assert not np.array_equal(a["single_img"], g["single_img"]), "the forced build took the same variant: U3D_FORCE_GENERAL did not reach the kernel"
