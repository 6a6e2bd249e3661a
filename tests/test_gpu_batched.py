"""Batched operator, drop-in module name, render-loss step and full-size properties (BASELINE config C2)."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _batch(B, P, V, H, W, level="object", seed=42, compact=False):
    from unipre3d_amd import synthetic
    b = synthetic.make_batch(B, P, V, H, W, level=level, seed=seed, compact=compact)
    return b, b.to(torch.device("cuda:0"))


def test_batched_equals_per_view_operator_and_oracle(oracle_mod):
    """C1 geometry: B=2 objects x V=4 views, 128 Gaussians, 128x128."""
    from unipre3d_amd import head, renderer, synthetic
    import diff_gaussian_rasterization as dgr           # the reference's import name resolves to our module
    b, bd = _batch(2, 128, 4, 128, 128)
    bd.raw.requires_grad_(True)
    g = synthetic.gaussians_from_batch(bd)
    out = renderer.render_views(g, bd.world_view, bd.full_proj, bd.camera_center, bd.bg, bd.fov_deg, 128, 128)
    assert out.shape == (8, 3, 128, 128)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).cuda()
    (out * w).sum().backward()
    g_batched = bd.raw.grad.clone()
    # the per-view route, exactly as the reference loops (train_network.py:418-446)
    bd.raw.grad = None
    g2 = synthetic.gaussians_from_batch(bd)
    t = math.tan(bd.fov_deg * math.pi / 360)
    imgs = []
    for bi in range(2):
        for v in range(4):
            st = dgr.GaussianRasterizationSettings(128, 128, t, t, bd.bg, 1.0, bd.world_view[bi, v], bd.full_proj[bi, v], 1,
                                                   bd.camera_center[bi, v], False, False, True)
            shs = head.concat_sh(g2["features_dc"][bi], g2["features_rest"][bi])
            img, radii, _ = dgr.GaussianRasterizer(st)(means3D=g2["xyz"][bi], means2D=torch.zeros_like(g2["xyz"][bi]),
                                                       opacities=g2["opacity"][bi], shs=shs, scales=g2["scaling"][bi],
                                                       rotations=g2["rotation"][bi])
            imgs.append(img)
    per_view = torch.stack(imgs)
    (per_view * w).sum().backward()
    assert rel_l2(out.detach().cpu().numpy(), per_view.detach().cpu().numpy()) < 1e-6
    assert rel_l2(g_batched.cpu().numpy(), bd.raw.grad.cpu().numpy()) < 1e-4
    # and against the oracle, view by view
    gc = {k: v.detach().cpu() for k, v in g2.items()}
    for bi in range(2):
        shs = head.concat_sh(gc["features_dc"][bi], gc["features_rest"][bi]).numpy()
        for v in range(4):
            r = oracle_mod.forward(gc["xyz"][bi].numpy(), gc["opacity"][bi].numpy(), b.world_view[bi, v].numpy(),
                                   b.full_proj[bi, v].numpy(), b.camera_center[bi, v].numpy(), b.bg.numpy(), 128, 128, t, t,
                                   shs=shs, scales=gc["scaling"][bi].numpy(), rotations=gc["rotation"][bi].numpy(), sh_degree=1)
            assert rel_l2(out[bi * 4 + v].detach().cpu().numpy(), r.color) < TOL


def test_reference_style_render_predicted_runs_on_device():
    from unipre3d_amd import renderer, synthetic

    class NS:
        def __init__(self, **k):
            self.__dict__.update(k)
    b, bd = _batch(1, 128, 2, 128, 128)
    g = synthetic.gaussians_from_batch(bd)
    pc = {k: v[0] for k, v in g.items()}
    cfg = NS(data=NS(fov=bd.fov_deg, training_resolution=128), model=NS(max_sh_degree=1))
    out = renderer.render_predicted(pc, bd.world_view[0, 0], bd.full_proj[0, 0], bd.camera_center[0, 0], bd.bg, cfg)
    assert out["render"].shape == (3, 128, 128) and out["radii"].shape == (128,) and out["radii"].dtype == torch.int32
    assert out["visibility_filter"].dtype == torch.bool and torch.isfinite(out["render"]).all()
    batched = renderer.render_views(g, bd.world_view, bd.full_proj, bd.camera_center, bd.bg, bd.fov_deg, 128, 128)
    assert rel_l2(out["render"].detach().cpu().numpy(), batched[0].detach().cpu().numpy()) < 1e-6


def test_scene_level_batched_vs_oracle(oracle_mod):
    """Scene-level geometry (white background, 4:3 image with one fov, P above the LDS-sort limit)."""
    from unipre3d_amd import head, renderer, synthetic
    b, bd = _batch(1, 6000, 3, 120, 160, level="scene", seed=7)
    g = synthetic.gaussians_from_batch(bd)
    out = renderer.render_views(g, bd.world_view, bd.full_proj, bd.camera_center, bd.bg, bd.fov_deg, 120, 160)
    t = math.tan(bd.fov_deg * math.pi / 360)
    gc = {k: v.detach().cpu() for k, v in g.items()}
    shs = head.concat_sh(gc["features_dc"][0], gc["features_rest"][0]).numpy()
    for v in range(3):
        r = oracle_mod.forward(gc["xyz"][0].numpy(), gc["opacity"][0].numpy(), b.world_view[0, v].numpy(),
                               b.full_proj[0, v].numpy(), b.camera_center[0, v].numpy(), b.bg.numpy(), 120, 160, t, t, shs=shs,
                               scales=gc["scaling"][0].numpy(), rotations=gc["rotation"][0].numpy(), sh_degree=1)
        assert rel_l2(out[v].detach().cpu().numpy(), r.color) < TOL


def test_full_size_C2_properties(oracle_mod):
    """BASELINE config C2 (B=32, V=4, P=128, 256x256): size-independent properties + spot checks vs the oracle."""
    from unipre3d_amd import head, losses, renderer, synthetic
    b, bd = _batch(32, 128, 4, 256, 256)
    bd.raw.requires_grad_(True)
    g = synthetic.gaussians_from_batch(bd)
    out = renderer.render_views(g, bd.world_view, bd.full_proj, bd.camera_center, bd.bg, bd.fov_deg, 256, 256)
    assert out.shape == (128, 3, 256, 256) and torch.isfinite(out).all()
    # black background + colours clamped >= 0 -> image >= 0 ; transmittance bound: each channel <= max colour
    assert out.min().item() >= 0.0
    loss = losses.render_loss(out, bd.gt.reshape(128, 3, 256, 256), "focal_l2")
    loss.backward()
    assert torch.isfinite(bd.raw.grad).all() and bd.raw.grad.abs().sum().item() > 0
    # determinism of the forward (no atomics there): two runs are bit-identical
    out2 = renderer.render_views(synthetic.gaussians_from_batch(bd), bd.world_view, bd.full_proj, bd.camera_center, bd.bg,
                                 bd.fov_deg, 256, 256)
    assert torch.equal(out, out2)
    # linearity of the backward in the cotangent: grad(2w) == 2 grad(w)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).cuda()
    g1, = torch.autograd.grad((out2 * w).sum(), bd.raw, retain_graph=True)
    g2, = torch.autograd.grad((out2 * (2 * w)).sum(), bd.raw)
    assert rel_l2(g2.cpu().numpy(), 2 * g1.cpu().numpy()) < 1e-5
    # spot-check 3 of the 128 views against the oracle at full resolution
    t = math.tan(bd.fov_deg * math.pi / 360)
    gc = {k: v.detach().cpu() for k, v in g.items()}
    for (bi, v) in ((0, 0), (13, 2), (31, 3)):
        shs = head.concat_sh(gc["features_dc"][bi], gc["features_rest"][bi]).numpy()
        r = oracle_mod.forward(gc["xyz"][bi].numpy(), gc["opacity"][bi].numpy(), b.world_view[bi, v].numpy(),
                               b.full_proj[bi, v].numpy(), b.camera_center[bi, v].numpy(), b.bg.numpy(), 256, 256, t, t,
                               shs=shs, scales=gc["scaling"][bi].numpy(), rotations=gc["rotation"][bi].numpy(), sh_degree=1)
        assert rel_l2(out[bi * 4 + v].detach().cpu().numpy(), r.color) < TOL


@pytest.mark.parametrize("level,loss_kind,P,V,H,W", [("object", "focal_l2", 128, 4, 128, 128), ("object", "l2", 300, 2, 64, 96),
                                                      ("scene", "l2", 500, 3, 120, 160), ("object", "l1", 64, 2, 48, 48)])
@pytest.mark.parametrize("single_pass", [False, True])
@pytest.mark.parametrize("opaque", [False, True])
def test_fused_render_loss_equals_unfused_path(oracle_mod, level, loss_kind, P, V, H, W, single_pass, opaque):
    """N2+N3: head-activation + render + loss in the HIP library == torch activations + batched operator + torch loss,
    for the loss value and for the gradient w.r.t. the raw head output.  `opaque`: every 9th Gaussian gets opacity
    sigmoid(6) = 0.9975, so tiles leave the clamp-free loop variant (tile_stage in u3d_render.hip) and the 0.99 clamp is live."""
    from arbiter import assert_parity, head_grad_arbiter_all
    from unipre3d_amd import fused, step
    b, bd = _batch(2, P, V, H, W, level=level, seed=11)
    if opaque:
        bd.raw[:, 3, ::9] = 6.0
        b.raw[:, 3, ::9] = 6.0
    head_out = bd.raw.permute(0, 2, 1).contiguous().requires_grad_(True)      # (B,P,23): what `final` emits
    loss_f, img_f, radii_f = fused.render_loss_fused(head_out, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt,
                                                     bd.bg, bd.fov_deg, H, W, level=level, offset_scale=bd.offset_scale,
                                                     loss_kind=loss_kind, debug=True, single_pass=single_pass)
    loss_f.backward(retain_graph=True)
    g_fused = head_out.grad.clone()
    head_out.grad = None
    (3.0 * loss_f).backward()                                                  # dL/dloss != 1
    g_fused3 = head_out.grad.clone()
    raw = bd.raw.clone().requires_grad_(True)                                  # (B,23,P) view the reference works on
    loss_u, img_u = step.render_loss_forward(raw, bd, H, W, 0, loss_kind)
    loss_u.backward()
    g_unfused = raw.grad.permute(0, 2, 1)
    assert rel_l2(img_f.cpu().numpy(), img_u.detach().cpu().numpy()) < 1e-5
    assert abs(loss_f.item() - loss_u.item()) < 1e-5 * max(1.0, abs(loss_u.item()))
    # every route -- fused, fused with dL/dloss = 3 (the two-pass kernels scale their seeds, so every term is re-rounded), and the
    # operator chain -- is held to the parity bar against the fp64 arbiter of the WHOLE chain (reference activations -> oracle ->
    # reference loss); scene-level gradients cancel ~1000x across tiles, there the fp32 restatement's own gap sets the bar
    a32, l32 = head_grad_arbiter_all(oracle_mod, b, H, W, loss_kind, np.float32)
    a64, l64 = head_grad_arbiter_all(oracle_mod, b, H, W, loss_kind, np.float64)
    if level == "scene":
        # the HIP routes are evaluated under two roundings below (dL/dloss = 1 and 3); so is the fp32 restatement, and the bar uses
        # its worse sample (measured on this draw: HIP 3e-5 ... 2.1e-4 over four scales, restatement 4e-5 ... 1.9e-4 run to run)
        from arbiter import worst_fp32_sample
        a32 = worst_fp32_sample([a32, head_grad_arbiter_all(oracle_mod, b, H, W, loss_kind, np.float32, loss_scale=3.0)[0]], a64)
    assert abs(loss_f.item() - l64) <= 1e-5 * max(1.0, abs(l64))
    if loss_kind != "l1":    # (L1's gradient is discontinuous where a pixel equals its target: value / image bar only)
        tr = lambda x: x.permute(0, 2, 1).cpu().numpy()
        assert_parity(tr(g_fused), a32, a64, "fused d(head_out)")
        assert_parity(tr(g_fused3) / 3.0, a32, a64, "fused d(head_out), dL/dloss = 3")
        assert_parity(tr(g_unfused), a32, a64, "operator-chain d(head_out)")
    else:
        assert rel_l2(g_fused.cpu().numpy(), g_unfused.cpu().numpy()) < TOL
        assert rel_l2(g_fused3.cpu().numpy(), 3.0 * g_fused.cpu().numpy()) < TOL
    # every channel group carries gradient (xyz, opacity, scaling, rotation, dc, rest)
    for lo, hi in ((0, 3), (3, 4), (4, 7), (7, 11), (11, 14), (14, 23)):
        assert g_fused[..., lo:hi].abs().sum().item() > 0


def test_backward_is_run_to_run_deterministic():
    """Dense regime (every splat reaches every tile, contributions within the first 64 sorted positions): the per-tile
    partial buffer + fixed-order f64 reduce make the gradient BIT-identical across runs (the original's fp32 atomics are
    not).  Sparse regime: f64 atomics -> identical to ~1e-7."""
    from unipre3d_amd import fused
    for level, P, tol in (("object", 128, 0.0), ("scene", 700, 1e-6)):
        b, bd = _batch(2, P, 3, 96, 128, level=level, seed=4)
        grads = []
        for _ in range(4):
            h = bd.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
            loss, _, _ = fused.render_loss_fused(h, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg,
                                                 96, 128, level=level, offset_scale=bd.offset_scale, loss_kind="l2")
            loss.backward()
            grads.append((loss.detach().clone(), h.grad.clone()))
        for l, g in grads[1:]:
            assert torch.equal(l, grads[0][0])
            if tol == 0.0:
                assert torch.equal(g, grads[0][1])
            else:
                assert rel_l2(g.cpu().numpy(), grads[0][1].cpu().numpy()) < tol


def test_fused_sh_degree0_and_uneven_shapes():
    """C = 14 channels (max_sh_degree 0), P not a multiple of 64/256, image not a multiple of 16, 1 view per object."""
    from unipre3d_amd import fused, head, losses, renderer, synthetic
    b, bd = _batch(3, 77, 1, 50, 70, seed=21)
    h14 = bd.raw.permute(0, 2, 1)[..., :14].contiguous().requires_grad_(True)
    loss, img, radii = fused.render_loss_fused(h14, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg,
                                               50, 70, max_sh_degree=0, loss_kind="focal_l2", debug=True)
    loss.backward()
    raw = bd.raw[:, :14].clone().requires_grad_(True)
    g = head.process_object_output(raw, bd.center, 1.0, max_sh_degree=0)
    out = renderer.render_views(g, bd.world_view, bd.full_proj, bd.camera_center, bd.bg, bd.fov_deg, 50, 70, max_sh_degree=0)
    lu = losses.render_loss(out, bd.gt.reshape(3, 3, 50, 70), "focal_l2")
    lu.backward()
    assert img.shape == (3, 3, 50, 70) and radii.shape == (3, 77)
    assert rel_l2(img.cpu().numpy(), out.detach().cpu().numpy()) < 1e-5 and abs(loss.item() - lu.item()) < 1e-6
    assert rel_l2(h14.grad.cpu().numpy(), raw.grad.permute(0, 2, 1).cpu().numpy()) < TOL
    with pytest.raises(ValueError):
        fused.render_loss_fused(h14, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg, 50, 70,
                                max_sh_degree=1)


def test_end_to_end_standin_step_runs_and_trains():
    """Every native operator in its real position: FPS/ball-query/group -> tokenizer -> blocks -> 2D->3D fusion -> final ->
    fused render-loss; a few AdamW steps must reduce the loss and keep all gradients finite."""
    from unipre3d_amd import cameras, fused
    from unipre3d_amd.gradcheck import check_and_clip_gradients
    from unipre3d_amd.standin import PointTransformerStandIn, object_intrinsics
    b, bd = _batch(4, 128, 2, 64, 64, seed=9)
    dev = bd.raw.device
    g = torch.Generator().manual_seed(1)
    pts = (torch.randn(4, 1024, 3, generator=g) * 0.2).to(dev)
    img = torch.randn(4, 128, 32, 32, generator=g).to(dev)
    c2w = torch.linalg.inv(bd.world_view[:, 0]).contiguous()
    torch.manual_seed(0)
    net = PointTransformerStandIn(depth=2).to(dev)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        head_out, center = net(pts, img, c2w, object_intrinsics(cameras.OBJECT_FOV_DEG, 32))
        assert head_out.shape == (4, 128, 23) and center.shape == (4, 128, 3)
        loss, _, _ = fused.render_loss_fused(head_out, center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg,
                                             64, 64, return_images=False)
        loss.backward()
        assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
        assert net.image_conv[1].weight.grad.abs().sum() > 0          # gradient reaches the image branch through the HIP gather
        assert check_and_clip_gradients(net.parameters(), 1.0)
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]


def test_backward_unit_equals_loss_backward():
    """fused.backward_unit(loss) (cached unit dL/dloss, no scaling multiply) gives bit-identical gradients to loss.backward(),
    for the single-pass step and -- falling back to the ordinary path -- for the two-pass one; a scaled loss still scales."""
    from unipre3d_amd import fused
    b, bd = _batch(2, 128, 2, 64, 64, seed=9)
    grads = []
    for mode in ("backward", "unit", "unit_two_pass", "scaled"):
        h = bd.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
        loss, _, _ = fused.render_loss_fused(h, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg, 64, 64,
                                             offset_scale=bd.offset_scale, loss_kind="focal_l2", single_pass=(mode != "unit_two_pass"))
        if mode == "backward":
            loss.backward()
        elif mode == "scaled":
            (2.0 * loss).backward()
        else:
            fused.backward_unit(loss)
        grads.append(h.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert rel_l2(grads[2].cpu().numpy(), grads[0].cpu().numpy()) < 1e-6
    assert rel_l2(grads[3].cpu().numpy(), 2.0 * grads[0].cpu().numpy()) < 1e-6


def test_more_than_65535_views_in_one_call():
    """The tile kernels index views through gridDim.y and, beyond 65535 views, gridDim.z slabs: 70 000 views of a tiny image (four
    distinct cameras repeated) must reproduce the 4-view render in every slab, and the gradient of sum(images) must be 17 500 times
    the 4-view gradient."""
    import math
    from unipre3d_amd import head, synthetic
    from unipre3d_amd.rasterizer import rasterize_gaussians_batched
    dev = torch.device("cuda:0")
    V = 70000
    b = synthetic.make_batch(1, 6, 4, 16, 16, seed=3).to(dev)
    g = synthetic.gaussians_from_batch(b)
    rep = lambda x: x.repeat(1, V // 4, *([1] * (x.dim() - 2))).contiguous()
    t = math.tan(b.fov_deg * math.pi / 360)
    kw = dict(shs=head.concat_sh(g["features_dc"], g["features_rest"]), scales=g["scaling"], rotations=g["rotation"], sh_degree=1)
    xyz = g["xyz"].clone().requires_grad_(True)
    col, _, _ = rasterize_gaussians_batched(xyz, g["opacity"], rep(b.world_view), rep(b.full_proj), rep(b.camera_center), b.bg, 16, 16, t, t, **kw)
    col.sum().backward()
    xyz4 = g["xyz"].clone().requires_grad_(True)
    ref, _, _ = rasterize_gaussians_batched(xyz4, g["opacity"], b.world_view, b.full_proj, b.camera_center, b.bg, 16, 16, t, t, **kw)
    ref.sum().backward()
    for i in (0, 3, 65532, 65535, 65536, 65539, V - 4, V - 1):
        assert torch.equal(col[0, i], ref[0, i % 4]), i
    # (each Gaussian's gradient is an fp32 sum over its 70 000 views: rounding noise ~1e-4)
    assert rel_l2(xyz.grad.cpu().numpy(), (V // 4) * xyz4.grad.cpu().numpy()) < 1e-3


@pytest.mark.parametrize("level,P,V,H,W,faint", [("object", 128, 4, 64, 64, False), ("scene", 700, 3, 48, 80, True), ("scene", 5000, 2, 64, 64, False)])
def test_single_pass_step_reuses_its_workspace_with_clean_accumulators(level, P, V, H, W, faint):
    """The single-pass step keeps its backward scratch between calls and tells the library that the gradient accumulators are still
    zero (U3D_FLAG_ACC_CLEAN: every step re-zeroes what it touched).  Steps on different inputs through the same workspace must be
    bit-identical to the same steps on a fresh workspace -- including `faint` scenes whose tiles go past the partial-row blocks and
    add into the accumulators with f64 atomics."""
    from unipre3d_amd import fused
    dev = torch.device("cuda:0")

    def step(bd):
        h = bd.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
        loss, _, _ = fused.render_loss_fused(h, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg, H, W,
                                             level=level, offset_scale=bd.offset_scale, loss_kind="l2", single_pass=True, return_images=False)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), h.grad.clone()

    from unipre3d_amd import rasterizer
    C = rasterizer._C()
    batches = []
    for seed in (3, 4, 5):
        _, bd = _batch(2, P, V, H, W, level=level, seed=seed)
        if faint:
            bd.raw[:, 3] -= 4.0                      # opacity ~ 0.02: no early saturation, every tile walks the whole list
        batches.append(bd)
    fresh = []
    for bd in batches:
        C.clear_workspaces()
        fresh.append(step(bd))
        assert C.workspaces()[:3] == (1, 1, 0)                   # the step left its workspace marked clean, lease returned
    C.clear_workspaces()
    for rnd in range(2):
        for bd, (l0, g0) in zip(batches, fresh):
            l1, g1 = step(bd)                                    # from the second step on: U3D_FLAG_ACC_CLEAN
            assert torch.equal(l1, l0) and torch.equal(g1, g0), (rnd, float((g1 - g0).abs().max()))
    assert C.workspaces()[:3] == (1, 1, 0)


@pytest.mark.parametrize("level,loss_kind,P,V,H,W", [("object", "focal_l2", 128, 4, 96, 96), ("scene", "l2", 600, 2, 48, 80)])
def test_fused_route_with_an_extra_image_space_term(oracle_mod, level, loss_kind, P, V, H, W):
    """The reference's objective after `start_lpips_after` iterations is l12 + lambda_lpips * LPIPS(rendered, gt)
    (train_network.py:284-300): the fused two-pass route returns differentiable images and adds dL/d(rendered) of such a term to
    the in-kernel loss seed.  A fixed random perceptual-style term g(x) = mean(w * tanh(x)^2) stands in for the LPIPS network
    (its weights are not available offline); fused(loss + 0.01 g) must equal the operator chain and the fp64 arbiter."""
    from arbiter import assert_parity, head_grad_arbiter_all
    from unipre3d_amd import fused, step
    b, bd = _batch(2, P, V, H, W, level=level, seed=23)
    wgt = torch.rand(2 * V, 3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    g_img = lambda x: (wgt * torch.tanh(x) ** 2).mean()
    h = bd.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    loss, img, _ = fused.render_loss_fused(h, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg, H, W,
                                           level=level, offset_scale=bd.offset_scale, loss_kind=loss_kind, differentiable_images=True)
    assert img.requires_grad
    (loss + 0.01 * g_img(img)).backward()
    raw = bd.raw.clone().requires_grad_(True)
    loss_u, img_u = step.render_loss_forward(raw, bd, H, W, 0, loss_kind)
    (loss_u + 0.01 * g_img(img_u)).backward()
    e = rel_l2(h.grad.cpu().numpy(), raw.grad.permute(0, 2, 1).cpu().numpy())
    # the image term ALONE through the fused route (dL/dloss absent: the in-kernel seed must vanish)
    h2 = bd.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    _, img2, _ = fused.render_loss_fused(h2, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg, H, W,
                                         level=level, offset_scale=bd.offset_scale, loss_kind=loss_kind, differentiable_images=True)
    g_img(img2).backward()
    raw2 = bd.raw.clone().requires_grad_(True)
    _, img_u2 = step.render_loss_forward(raw2, bd, H, W, 0, loss_kind)
    g_img(img_u2).backward()
    e2 = rel_l2(h2.grad.cpu().numpy(), raw2.grad.permute(0, 2, 1).cpu().numpy())
    # bar: 1e-4, or the measured fp32 noise floor of this workload (distance of the fp32 restatement of the loss-only chain
    # from its fp64 arbiter) where that is larger
    a32, _ = head_grad_arbiter_all(oracle_mod, b, H, W, loss_kind, np.float32)
    a64, _ = head_grad_arbiter_all(oracle_mod, b, H, W, loss_kind, np.float64)
    gap = rel_l2(a32, a64)
    assert e <= max(TOL, 2 * gap) and e2 <= max(TOL, 2 * gap), (e, e2, gap)
    # single-pass images stay detached, and asking for gradient through them without the switch is refused loudly
    h3 = bd.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    _, img3, _ = fused.render_loss_fused(h3, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg, H, W,
                                         level=level, offset_scale=bd.offset_scale, loss_kind=loss_kind)
    assert not img3.requires_grad
    with pytest.raises(ValueError):
        fused.render_loss_fused(h3, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg, H, W,
                                level=level, offset_scale=bd.offset_scale, loss_kind=loss_kind, differentiable_images=True,
                                return_images=False)


@pytest.mark.parametrize("level,P,H,W,faint", [("object", 128, 64, 64, False), ("scene", 700, 48, 80, True), ("scene", 5000, 64, 64, False)])
def test_operator_backward_reuses_its_workspace_with_clean_accumulators(level, P, H, W, faint):
    """The torch binding keeps the operator's backward scratch per (device, stream, shape) and tells the library that the gradient
    accumulators are still zero (U3D_FLAG_ACC_CLEAN: every backward hands back what it touched zeroed) -- no memset launch per call on
    the per-view route.  Backward passes on different inputs through the kept workspace must be bit-identical to the same passes on a
    fresh one, including `faint` scenes whose tiles add into the accumulators with f64 atomics beyond the partial-row blocks."""
    from unipre3d_amd import rasterizer
    from scenes import DIFF_KEYS, cotangents, scene
    from test_gpu_parity import _settings
    dev = torch.device("cuda:0")
    C = rasterizer._C()

    def run(seed):
        sc = scene(P, H, W, seed, level, False, 1)
        if faint:
            sc["opacities"] = sc["opacities"] * 0.02
        t = {k: (v.to(dev).requires_grad_(k in DIFF_KEYS) if torch.is_tensor(v) else v) for k, v in sc.items()}
        m2d = torch.zeros_like(t["means3D"], requires_grad=True)
        color, radii, invd = rasterizer.rasterize_gaussians(t["means3D"], m2d, t["shs"], None, t["opacities"], t["scales"], t["rotations"], None,
                                                            _settings(sc, t, debug=False))
        dcol, dinv = cotangents(H, W, seed=seed)
        ((color * dcol.to(dev)).sum() + (invd * dinv.to(dev)).sum()).backward()
        torch.cuda.synchronize()
        return [t[k].grad.clone() for k in DIFF_KEYS] + [m2d.grad.clone()]

    fresh = []
    for seed in (3, 4, 5):
        C.clear_workspaces()
        fresh.append(run(seed))
        assert C.workspaces()[:3] == (1, 1, 0)              # one scratch, left clean
    C.clear_workspaces()
    for rnd in range(2):
        for seed, g0 in zip((3, 4, 5), fresh):
            g1 = run(seed)                                  # from the second pass on: U3D_FLAG_ACC_CLEAN
            if faint or P > 256:
                # f64 atomics beyond the partial-row blocks: order-insensitive at fp32 output precision, not bit-identical
                assert all(rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-6 for a, b in zip(g1, g0) if b.abs().sum() > 0), (rnd, seed)
            else:
                assert all(torch.equal(a, b) for a, b in zip(g1, g0)), (rnd, seed)
    assert C.workspaces()[:3] == (1, 1, 0)


def test_fused_step_backward_half_scales_by_grad_output_and_survives_interleaving():
    """ABI 4: the autograd forward runs projection -> tiles -> reduce, the autograd backward runs the chain rule and multiplies by
    autograd's grad_output INSIDE the projection-backward kernel (no d_head * g launch).  Checks: (a) loss.backward(),
    backward_unit(loss) and (3 * loss).backward() / 3 agree (bit-identical for the first two: g = 1 is an exact multiply);
    (b) two forwards before either backward (the second finds the scratch leased and takes its own) give the gradients of two
    separate steps; (c) a forward whose loss is dropped does not poison the next step; (d) backward(retain_graph=True) followed by a
    second backward works (the forward half is recomputed); (e) one scratch buffer per stream is all that is ever cached."""
    from unipre3d_amd import fused, rasterizer
    C = rasterizer._C()
    _, b1 = _batch(2, 128, 2, 64, 64, level="object", seed=31)
    _, b2 = _batch(2, 128, 2, 64, 64, level="object", seed=32)

    def fwd(bd):
        h = bd.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
        loss, _, _ = fused.render_loss_fused(h, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg, 64, 64,
                                             level="object", offset_scale=bd.offset_scale, loss_kind="focal_l2", return_images=False)
        return h, loss

    C.clear_workspaces()
    h, l = fwd(b1); l.backward(); g_plain = h.grad.clone()
    h, l = fwd(b1); fused.backward_unit(l); g_unit = h.grad.clone()
    h, l = fwd(b1); (3.0 * l).backward(); g_3 = h.grad.clone() / 3.0
    h, l = fwd(b2); l.backward(); g2_ref = h.grad.clone()
    assert torch.equal(g_plain, g_unit)
    assert rel_l2(g_3.cpu().numpy(), g_plain.cpu().numpy()) < 1e-6
    assert C.workspaces()[:3] == (1, 1, 0)
    # (b) interleaved
    ha, la = fwd(b1)
    hb, lb = fwd(b2)
    assert C.workspaces()[2] == 1                       # the cached buffer is leased to the second forward; the first keeps its own
    la.backward(); lb.backward()
    assert torch.equal(ha.grad, g_plain) and torch.equal(hb.grad, g2_ref)
    assert C.workspaces()[:3] == (1, 1, 0)
    # (c) dropped loss
    hd, ld = fwd(b1)
    del hd, ld
    h, l = fwd(b2); l.backward()
    assert torch.equal(h.grad, g2_ref)
    h, l = fwd(b1); l.backward()
    assert torch.equal(h.grad, g_plain)
    # (d) second backward
    h, l = fwd(b1)
    l.backward(retain_graph=True)
    assert torch.equal(h.grad, g_plain)
    h.grad = None
    (2.0 * l).backward()
    assert rel_l2(h.grad.cpu().numpy() / 2.0, g_plain.cpu().numpy()) < 1e-6
    assert C.workspaces()[:3] == (1, 1, 0)
    # (e)
    n, clean, out, nbytes = C.workspaces()
    assert n == 1 and out == 0


def test_ragged_steps_share_one_scratch_and_never_accumulate():
    """Ragged scene-level batches bring a new total almost every step: the binding keeps ONE grow-only backward scratch per
    (device, stream) -- no buffer per distinct total, no promise across shapes -- and results equal a fresh-workspace run."""
    from unipre3d_amd import fused, rasterizer, synthetic
    C = rasterizer._C()
    dev = torch.device("cuda:0")
    V, H, W = 2, 48, 64

    def step(sizes, seed):
        bs = [synthetic.make_batch(1, n, V, H, W, level="scene", seed=seed + i).to(dev) for i, n in enumerate(sizes)]
        hp = torch.cat([b.raw[0].t() for b in bs]).contiguous().requires_grad_(True)
        cat = lambda k: torch.cat([getattr(b, k) for b in bs])
        loss, _, radii = fused.render_loss_fused(hp, torch.cat([b.center[0] for b in bs]), cat("world_view"), cat("full_proj"), cat("camera_center"),
                                                 cat("gt"), bs[0].bg, bs[0].fov_deg, H, W, level="scene", offset_scale=bs[0].offset_scale,
                                                 loss_kind="l2", return_images=False, sizes=sizes)
        loss.backward()
        return loss.detach().clone(), hp.grad.clone()

    shapes = [([300, 500], 1), ([700, 100, 250], 2), ([300, 500], 3), ([64, 65], 4), ([300, 500], 3)]
    fresh = []
    for sizes, seed in shapes:
        C.clear_workspaces()
        fresh.append(step(sizes, seed))
    C.clear_workspaces()
    peak = 0
    for rnd in range(2):
        for (sizes, seed), (l0, g0) in zip(shapes, fresh):
            l1, g1 = step(sizes, seed)
            assert torch.equal(l1, l0) and rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) < 1e-6
            n, _, out, nbytes = C.workspaces()
            assert n == 1 and out == 0
            peak = max(peak, nbytes)
    assert C.workspaces()[3] == peak          # grow-only: the buffer of the largest shape, nothing else


def _clip_reference(grads, max_norm):
    """train_network.py:368-390 on copies: (valid, clipped gradients)."""
    ps = [torch.nn.Parameter(torch.zeros_like(g)) for g in grads]
    for p, g in zip(ps, grads):
        p.grad = g.clone()
    bad = any(bool(torch.isnan(p.grad).any() or torch.isinf(p.grad).any()) for p in ps)
    if bad:
        return False, [p.grad for p in ps]
    torch.nn.utils.clip_grad_norm_(ps, max_norm=max_norm)
    return True, [p.grad for p in ps]


@pytest.mark.parametrize("case", ["large", "small", "nan", "inf", "neg_inf", "overflow", "bucket_views", "many"])
def test_gradclip_kernels_match_reference_semantics(case):
    """N4(b): the multi-tensor HIP pass (csrc/u3d_gradclip.hip) against the reference's own two statements --
    the per-parameter isnan / isinf scan and torch.nn.utils.clip_grad_norm_(max_norm=1.0) (train_network.py:376-389):
    clipping when the norm exceeds 1, identity (bit-exact) below it, skip-step on NaN / +Inf / -Inf with the gradients left
    untouched, finite gradients whose fp32 sum of squares overflows, DDP-style bucket views at odd 4-byte offsets, and a few
    hundred tensors of every size around the 64 K-element chunk and the 16-byte vector boundary."""
    import math
    from unipre3d_amd import gradcheck
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(dev)
    flat = None
    if case == "many":
        sizes = [1, 2, 3, 4, 5, 63, 64, 65, 1023, 4096, 65535, 65536, 65537, 131072, 200001, 384, 0] * 12
        grads = [mk(n, s=0.01) for n in sizes]
    elif case == "bucket_views":
        flat = mk(300000, s=0.05)
        offs, grads = 1, []                       # views at element offsets 1, 779, ...: every 4-byte alignment class
        for n in (777, 65536, 3, 100001, 2):
            grads.append(flat[offs:offs + n]); offs += n + 1
    elif case == "small":
        grads = [mk(384, 128, s=1e-4), mk(128, s=1e-4), mk(23, 128, s=1e-4)]
    else:
        grads = [mk(384, 128), mk(128), mk(70000), mk(23, 128)]
    if case == "nan":
        grads[2][12345] = float("nan")
    if case == "inf":
        grads[0][3, 7] = float("inf")
    if case == "neg_inf":
        grads[3][22, 127] = float("-inf")
    if case == "overflow":
        grads[2].mul_(1e19)                        # finite values, fp32 sum of squares = inf
    before = [x.clone() for x in grads]
    flat_before = flat.clone() if flat is not None else None
    ps = [torch.nn.Parameter(torch.zeros_like(x)) for x in grads]
    for p, x in zip(ps, grads):
        p.grad = x                                 # (bucket views are clipped in place inside the flat buffer)
    ok_ref, ref = _clip_reference(before, 1.0)
    st = gradcheck.gradient_state(ps, 1.0)
    ok = gradcheck.check_and_clip_gradients(ps, 1.0)
    torch.cuda.synchronize()
    if case == "overflow":
        # the reference's fp32 total norm is inf here -> clip_grad_norm_ zeroes everything; the f64 norm stays finite and the
        # gradients are scaled to unit norm (what clip_grad_norm_ does for any finite norm)
        assert ok and math.isfinite(st["total_norm"]) and st["total_norm"] > 1e19
        tot = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in ps))
        assert abs(tot - 1.0) < 1e-5
        return
    assert ok == ok_ref and st["found_inf"] == (not ok_ref)
    if not ok_ref:
        nn = lambda x: x.nan_to_num(7.0, 8.0, 9.0)
        assert all(torch.equal(nn(p.grad), nn(b)) for p, b in zip(ps, before))      # untouched
        return
    tot_ref = math.sqrt(sum(float((b.double() ** 2).sum()) for b in before))
    assert abs(st["total_norm"] - tot_ref) <= 1e-9 * max(tot_ref, 1.0)
    assert st["amax"] == max(float(b.abs().max()) for b in before if b.numel())
    if case == "small":
        assert st["coef"] == 1.0 and all(torch.equal(p.grad, b) for p, b in zip(ps, before))       # below max_norm: bit-identical
    else:
        assert st["coef"] < 1.0
    for p, r in zip(ps, ref):
        if p.numel():
            assert rel_l2(p.grad.cpu().numpy(), r.cpu().numpy()) < 1e-6
    if flat is not None:        # the elements between the views are not ours to touch
        mask = torch.ones_like(flat, dtype=torch.bool)
        offs = 1
        for n in (777, 65536, 3, 100001, 2):
            mask[offs:offs + n] = False; offs += n + 1
        assert torch.equal(flat[mask], flat_before[mask])


def test_gradclip_deferred_drives_fused_adamw_without_a_host_read():
    """check_and_clip_deferred: found_inf stays on the device; AdamW(fused=True) skips the step exactly when the reference's
    `if not valid: skip` would, and otherwise steps on the clipped gradients -- same parameters as the host-read route."""
    from unipre3d_amd import gradcheck
    dev = torch.device("cuda:0")

    def run(deferred, poison):
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 23)).to(dev)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-2, fused=True)
        x = torch.randn(512, 64, generator=torch.Generator().manual_seed(1)).to(dev)
        out = []
        for it in range(4):
            opt.zero_grad(set_to_none=False)
            (m(x) ** 2).sum().backward()                     # norm >> 1: clipping is live
            if poison and it == 1:
                m[0].weight.grad[5, 5] = float("nan")
            if deferred:
                gradcheck.check_and_clip_deferred(m.parameters(), opt, 1.0)
                opt.step()
            elif gradcheck.check_and_clip_gradients(m.parameters(), 1.0):
                opt.step()
            out.append([p.detach().clone() for p in m.parameters()])
        return out

    for poison in (False, True):
        a, b = run(False, poison), run(True, poison)
        for it, (pa, pb) in enumerate(zip(a, b)):
            assert all(torch.isfinite(x).all() for x in pb), (poison, it)
            assert all(rel_l2(x.cpu().numpy(), y.cpu().numpy()) < 1e-6 for x, y in zip(pa, pb)), (poison, it)
        if poison:      # iteration 1 was skipped: parameters equal those after iteration 0
            assert all(torch.equal(x, y) for x, y in zip(b[0], b[1]))


@pytest.mark.parametrize("case", ["mixed_clip", "mixed_small", "mixed_nan_in_bf16", "only_odd"])
def test_gradclip_accepts_any_dtype_and_layout_like_the_reference(case):
    """The reference's check accepts any gradient (train_network.py:368-390); the pointer-table kernels take contiguous fp32 only.
    A channels_last conv-weight gradient, a bf16 and an fp16 parameter ride along: their statistics are folded into the same
    device-side state block and they are scaled by the same coefficient -- against clip_grad_norm_ on the same set."""
    import math
    from unipre3d_amd import gradcheck
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    s = 1e-4 if case == "mixed_small" else 1.0
    mk = lambda *shape: (torch.randn(*shape, generator=g) * s).to(dev)
    grads = [] if case == "only_odd" else [mk(384, 128), mk(70001)]
    grads += [mk(16, 8, 3, 3).contiguous(memory_format=torch.channels_last), mk(256, 64).to(torch.bfloat16), mk(33).to(torch.float16),
              mk(40, 50).t()]
    assert not grads[-1].is_contiguous() and not grads[-4].is_contiguous()
    if case == "mixed_nan_in_bf16":
        grads[-3][3, 3] = float("nan")
    before = [x.clone() for x in grads]
    ps = [torch.nn.Parameter(torch.zeros_like(x)) for x in grads]
    for p, x in zip(ps, grads):
        p.grad = x
    ref_ps = [torch.nn.Parameter(torch.zeros_like(x)) for x in before]
    for p, x in zip(ref_ps, before):
        p.grad = x.clone()
    bad_ref = any(bool(torch.isnan(p.grad).any() or torch.isinf(p.grad).any()) for p in ref_ps)
    if not bad_ref:
        # clip_grad_norm_'s statement -- g *= min(1, max_norm / (total_norm + 1e-6)) -- with the total norm in float64 (torch's own
        # mixed-dtype route takes the bf16 / fp16 tensors' norms in THEIR precision: 1e-4 off on the coefficient)
        tot = math.sqrt(sum(float((b.double() ** 2).sum()) for b in before))
        c = min(1.0, 1.0 / (tot + 1e-6))
        for p in ref_ps:
            p.grad = (p.grad.double() * c).to(p.grad.dtype) if c < 1.0 else p.grad
        torch_ps = [torch.nn.Parameter(torch.zeros_like(x)) for x in before]
        for p, x in zip(torch_ps, before):
            p.grad = x.clone()
        torch.nn.utils.clip_grad_norm_(torch_ps, max_norm=1.0)
        for p, q in zip(ref_ps, torch_ps):          # ... which torch's own result agrees with to its precision
            assert rel_l2(p.grad.float().cpu().numpy(), q.grad.float().cpu().numpy()) < 2e-2
    st = gradcheck.gradient_state(ps, 1.0)
    ok = gradcheck.check_and_clip_gradients(ps, 1.0)
    torch.cuda.synchronize()
    assert ok == (not bad_ref) and st["found_inf"] == bad_ref
    if bad_ref:
        nn = lambda x: x.float().nan_to_num(7.0, 8.0, 9.0)
        assert all(torch.equal(nn(p.grad), nn(b)) for p, b in zip(ps, before))
        return
    tot_ref = math.sqrt(sum(float((b.double() ** 2).sum()) for b in before))
    assert abs(st["total_norm"] - tot_ref) <= 1e-9 * max(tot_ref, 1.0)
    assert (st["coef"] == 1.0) == (case == "mixed_small")
    for p, r in zip(ps, ref_ps):
        assert p.grad.dtype == r.grad.dtype and p.grad.stride() == r.grad.stride()
        tol = 1e-6 if p.grad.dtype == torch.float32 else 2e-2          # (half-precision gradients round after the multiply)
        assert rel_l2(p.grad.float().cpu().numpy(), r.grad.float().cpu().numpy()) < tol
    # the host-read-free route on the same set
    for p, x in zip(ps, before):
        p.grad = x.clone()
    opt = torch.optim.AdamW([p for p in ps if p.dtype == torch.float32 and p.grad.is_contiguous()] or [torch.nn.Parameter(torch.zeros(1, device=dev))],
                            lr=0.0, fused=True)
    gradcheck.check_and_clip_deferred(ps, opt, 1.0)
    torch.cuda.synchronize()
    assert float(opt.found_inf) == 0.0
    for p, r in zip(ps, ref_ps):
        tol = 1e-6 if p.grad.dtype == torch.float32 else 2e-2
        assert rel_l2(p.grad.float().cpu().numpy(), r.grad.float().cpu().numpy()) < tol


def test_train_step_does_not_leave_a_stale_found_inf_on_the_optimizer():
    """The deferred clip hands AdamW(fused=True) a device-side found_inf; train_step takes it off again after the step, so a later
    step that does not clip (clip_grad=None) is not silently skipped, and return_found_inf exposes the decision to the caller."""
    from unipre3d_amd import dp, step as step_mod
    dev = torch.device("cuda:0")
    _, b = _batch(2, 128, 2, 64, 64, level="object", seed=13)
    torch.manual_seed(0)
    model = dp.GaussianHead(32, 64).to(dev)
    feats = torch.randn(2, 128, 32, generator=torch.Generator().manual_seed(2)).to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
    _, f0 = step_mod.train_step(model, feats, b, opt, 64, 64, 0, "focal_l2", fused=True, return_found_inf=True)
    assert float(f0) == 0.0 and getattr(opt, "found_inf", None) is None and getattr(opt, "grad_scale", None) is None
    bad = feats.clone(); bad[0, 0, 0] = float("nan")
    w0 = [p.detach().clone() for p in model.parameters()]
    _, f1 = step_mod.train_step(model, bad, b, opt, 64, 64, 0, "focal_l2", fused=True, return_found_inf=True)
    assert float(f1) == 1.0 and all(torch.equal(x, p.detach()) for x, p in zip(w0, model.parameters()))      # skipped
    assert getattr(opt, "found_inf", None) is None
    step_mod.train_step(model, feats, b, opt, 64, 64, 0, "focal_l2", fused=True, clip_grad=None)               # no clip: must still step
    assert any(not torch.equal(x, p.detach()) for x, p in zip(w0, model.parameters()))


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_render_predicted_one_call_body_equals_the_op_by_op_body(oracle_mod, deg):
    """renderer.render_predicted on a HIP device is ONE binding call (u3d_render_view_*: split SH pointers, cached zero leaf as
    `viewspace_points`, visibility from the projection kernel, no inverse-depth plane).  Against the op-by-op body that mirrors
    gaussian_renderer/__init__.py line by line (pinned by G3), and against the oracle: image, radii, visibility, the gradient of
    every Gaussian parameter and `viewspace_points.grad`."""
    import types
    from unipre3d_amd import renderer
    from scenes import scene, to_numpy
    dev = torch.device("cuda:0")
    H, W, P = 64, 80, 150
    sc = scene(P, H, W, seed=7 + deg, level="object", compact=False, deg=deg)
    cfg = types.SimpleNamespace(data=types.SimpleNamespace(fov=2 * math.degrees(math.atan(sc["tanfovx"])), training_height=H, training_width=W),
                                model=types.SimpleNamespace(max_sh_degree=deg))
    dcol = torch.randn(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
    res = {}
    for fast in (True, False):
        renderer.FAST_PATH = fast
        try:
            pc = {"xyz": sc["means3D"], "opacity": sc["opacities"], "scaling": sc["scales"], "rotation": sc["rotations"],
                  "features_dc": sc["shs"][:, :1].contiguous(), "features_rest": sc["shs"][:, 1:].contiguous()}
            pc = {k: v.to(dev).requires_grad_(True) for k, v in pc.items()}
            out = renderer.render_predicted(pc, sc["viewmatrix"].to(dev), sc["projmatrix"].to(dev), sc["campos"].to(dev), sc["bg"].to(dev), cfg)
            (out["render"] * dcol).sum().backward()
            torch.cuda.synchronize()
            res[fast] = (out, pc)
        finally:
            renderer.FAST_PATH = True
    (a, pa), (b, pb) = res[True], res[False]
    assert a["viewspace_points"].is_leaf and a["viewspace_points"].requires_grad and not bool(a["viewspace_points"].any())
    assert torch.equal(a["radii"], b["radii"]) and torch.equal(a["visibility_filter"], b["visibility_filter"])
    assert a["visibility_filter"].dtype == torch.bool and torch.equal(a["visibility_filter"], a["radii"] > 0)
    assert rel_l2(a["render"].detach().cpu().numpy(), b["render"].detach().cpu().numpy()) < 1e-6
    for k in pa:
        if pa[k].numel():
            assert pa[k].grad.shape == pb[k].grad.shape
            assert rel_l2(pa[k].grad.cpu().numpy(), pb[k].grad.cpu().numpy()) < 1e-5, k
    assert rel_l2(a["viewspace_points"].grad.cpu().numpy(), b["viewspace_points"].grad.cpu().numpy()) < 1e-5
    r = oracle_mod.forward(dtype=np.float32, **to_numpy(sc))
    go = oracle_mod.backward(r, dcol.cpu().numpy())
    assert np.array_equal(a["radii"].cpu().numpy(), r.radii) and rel_l2(a["render"].detach().cpu().numpy(), r.color) < TOL
    g_rest = pa["features_rest"].grad if pa["features_rest"].numel() else torch.zeros_like(pa["features_rest"])
    gshs = torch.cat([pa["features_dc"].grad, g_rest], dim=1).cpu().numpy()
    assert rel_l2(gshs, go["shs"]) < TOL and rel_l2(pa["xyz"].grad.cpu().numpy(), go["means3D"]) < TOL
    assert rel_l2(a["viewspace_points"].grad.cpu().numpy(), go["means2D"]) < TOL
    # two views of the same Gaussians: the cached zero storage is shared, the leaves (and their .grad) are not
    s1 = renderer.render_predicted(pa, sc["viewmatrix"].to(dev), sc["projmatrix"].to(dev), sc["campos"].to(dev), sc["bg"].to(dev), cfg)["viewspace_points"]
    assert s1 is not a["viewspace_points"] and s1.grad is None and a["viewspace_points"].grad is not None


def test_render_predicted_one_call_body_edge_cases():
    """The one-call body under the conditions the reference's callers produce: torch.no_grad() (eval.py:101-109), float64 / non-contiguous
    inputs (converted like the operator does), no Gaussians at all (background image), an override colour or per-view focals (general body)."""
    import types
    from unipre3d_amd import renderer
    from scenes import scene
    dev = torch.device("cuda:0")
    H, W, P = 48, 64, 70
    sc = scene(P, H, W, seed=3, level="object", compact=False, deg=1)
    cfg = types.SimpleNamespace(data=types.SimpleNamespace(fov=2 * math.degrees(math.atan(sc["tanfovx"])), training_height=H, training_width=W),
                                model=types.SimpleNamespace(max_sh_degree=1))
    cams = (sc["viewmatrix"].to(dev), sc["projmatrix"].to(dev), sc["campos"].to(dev), sc["bg"].to(dev))
    pc = {"xyz": sc["means3D"], "opacity": sc["opacities"], "scaling": sc["scales"], "rotation": sc["rotations"],
          "features_dc": sc["shs"][:, :1].contiguous(), "features_rest": sc["shs"][:, 1:].contiguous()}
    pc = {k: v.to(dev) for k, v in pc.items()}
    ref = renderer.render_predicted(pc, *cams, cfg)
    with torch.no_grad():
        out = renderer.render_predicted(pc, *cams, cfg)
    assert torch.equal(out["render"], ref["render"]) and not out["render"].requires_grad
    odd = {k: v.double() for k, v in pc.items()}
    odd["xyz"] = torch.cat([odd["xyz"], odd["xyz"]], dim=1)[:, :3]                      # a non-contiguous view
    assert not odd["xyz"].is_contiguous()
    out = renderer.render_predicted(odd, *cams, cfg)
    assert out["render"].dtype == torch.float32 and torch.equal(out["render"], ref["render"]) and torch.equal(out["radii"], ref["radii"])
    empty = {k: v[:0] for k, v in pc.items()}
    out = renderer.render_predicted(empty, *cams, cfg)
    assert out["radii"].numel() == 0 and out["visibility_filter"].numel() == 0
    assert torch.equal(out["render"], sc["bg"].to(dev)[:, None, None].expand(3, H, W))
    col = torch.rand(P, 3, device=dev)
    a = renderer.render_predicted(pc, *cams, cfg, override_color=col)                  # general body: precomputed colours
    assert a["render"].shape == (3, H, W) and not torch.equal(a["render"], ref["render"])
    sq = types.SimpleNamespace(data=types.SimpleNamespace(fov=cfg.data.fov, training_resolution=H), model=cfg.model)
    f = torch.tensor([H / (2 * sc["tanfovx"])] * 2)
    b = renderer.render_predicted(pc, *cams, sq, focals_pixels=f)                       # general body: per-view focal lengths
    assert b["render"].shape == (3, H, H) and b["viewspace_points"].shape == (P, 3)


class _DropGrad(torch.autograd.Function):
    """y = x; hands NO gradient back (None): the node upstream is then run with an undefined grad_output."""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return None


def test_fused_step_backward_after_an_unused_loss_recomputes():
    """A retained node whose first backward arrived with an UNDEFINED gradient for the loss released its lease on the backward
    scratch; a later backward through the same node must recompute the forward half instead of reading a buffer that another
    forward has reused in the meantime."""
    from unipre3d_amd import fused
    _, b = _batch(2, 128, 2, 64, 64, level="object", seed=17)
    h = b.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    args = (b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, 64, 64)
    ref_loss, _, _ = fused.render_loss_fused(h, *args, level="object", return_images=False)
    ref_loss.backward()
    g_ref = h.grad.clone(); h.grad = None
    loss, _, _ = fused.render_loss_fused(h, *args, level="object", return_images=False)
    _DropGrad.apply(loss).backward(retain_graph=True)                  # the step's node sees an undefined gradient
    assert h.grad is None or not bool(h.grad.any())
    h.grad = None
    h2 = (h.detach() * 0.5).requires_grad_(True)                         # another forward takes over the released scratch ...
    l2, _, _ = fused.render_loss_fused(h2, *args, level="object", return_images=False)
    loss.backward()                                                     # ... before the retained node is backpropagated for real
    l2.backward()
    torch.cuda.synchronize()
    assert rel_l2(h.grad.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6
    assert bool(torch.isfinite(h2.grad).all()) and bool(h2.grad.any())


def test_fused_step_is_capturable_in_a_hip_graph():
    """The fused step (render_loss_fused + loss.backward()) holds no host synchronisation and sizes everything from shapes, so it
    can be captured in a HIP graph (torch.cuda.CUDAGraph) and replayed on new head outputs copied into the static input: loss and
    gradient of the replay are bit-identical to the eager step (object level: fixed-order reductions).  tools/graph_probe.py
    times it: the host-bound C1 step goes from 0.143 to 0.058 ms."""
    from unipre3d_amd import fused
    _, b = _batch(2, 128, 2, 64, 64, level="object", seed=41)
    _, b2 = _batch(2, 128, 2, 64, 64, level="object", seed=42)
    static_h = b.raw.permute(0, 2, 1).contiguous().requires_grad_(True)

    def run(h):
        loss, _, _ = fused.render_loss_fused(h, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, 64, 64,
                                             level="object", offset_scale=b.offset_scale, loss_kind="focal_l2", return_images=False)
        loss.backward()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                      # warm-up on a side stream, as torch's capture recipe asks
        for _ in range(3):
            static_h.grad = None
            run(static_h)
    torch.cuda.current_stream().wait_stream(s)
    static_h.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = run(static_h)
    for src in (b2, b):
        with torch.no_grad():
            static_h.copy_(src.raw.permute(0, 2, 1))
        g.replay()
        torch.cuda.synchronize()
        h = src.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
        l = run(h)
        torch.cuda.synchronize()
        assert torch.equal(static_loss.detach(), l.detach()) and torch.equal(static_h.grad, h.grad)
    del g
    from unipre3d_amd import rasterizer
    rasterizer._C().clear_workspaces()              # (the cached scratch may live in the destroyed graph's private pool)


@pytest.mark.parametrize("level,kind,fname", [("object", "focal_l2", "g10_chain.npz"), ("scene", "l2", "g10_chain_scene.npz")])
@pytest.mark.parametrize("single_pass", [True, False])
def test_g10_chain_known_answers(golden, single_pass, level, kind, fname):
    """The fused HIP step against tests/golden/g10_chain.npz: loss and d loss / d head_out of the whole chain (activations incl. the
    across-point quaternion quirk -> render -> focal-L2) from an independent float64 numpy transcription and central differences over
    all P x 23 head entries (tests/golden/make_g10_chain.py imports neither oracle/ nor the product).  Anti-aliasing off: the published
    backward differentiates the AA factor inexactly on purpose."""
    from unipre3d_amd import fused
    g = golden(fname)
    dev = torch.device("cuda:0")
    T = lambda k: torch.from_numpy(np.asarray(g[k])).to(dev)
    H, W = int(g["H"]), int(g["W"])
    h = T("head_out").contiguous().requires_grad_(True)
    loss, img, _ = fused.render_loss_fused(h, T("center"), T("world_view"), T("full_proj"), T("camera_center"), T("gt"), T("bg"), float(g["fov_deg"]), H, W,
                                           level=level, offset_scale=float(g["offset_scale"]), loss_kind=kind, antialiasing=False,
                                           single_pass=single_pass, debug=True)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    assert rel_l2(img.cpu().numpy(), g["images"]) < 2e-6
    e = rel_l2(h.grad.cpu().numpy(), g["d_head"])
    print(f"g10 chain {level} (single_pass={single_pass}): d head_out rel-L2 {e:.1e}")
    assert e < 2e-5
    for lo, hi in ((0, 3), (3, 4), (4, 7), (7, 11), (11, 14), (14, 23)):     # every channel group on its own
        assert rel_l2(h.grad[..., lo:hi].cpu().numpy(), g["d_head"][..., lo:hi]) < 1e-4, (lo, hi)
