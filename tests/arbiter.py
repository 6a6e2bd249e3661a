"""CPU arbiters for the full-size parity tests (TEST INFRASTRUCTURE: the only users of oracle/ besides smoke() and the
bench's cpu_baseline leg).

Two levels, both in fp32 (the restatement at the kernels' own precision) and fp64 (the arbiter):
  * operator level -- `oracle_view`: one (item, view) of a batch through oracle/raster_oracle.c, image + the gradients of
    the operator's six differentiable inputs for a given dL/dcolor;
  * head level -- `head_grad_arbiter`: d loss / d (raw head output of one item) when only ONE of its views carries loss,
    chained through the reference's activations (unipre3d_amd/head.py, pinned by golden G2) in the same dtype and the
    render loss of utils/loss_utils.py (pinned by G4), with the oracle as the differentiable renderer.

Parity bar -- ONE rule for every comparison that brings both restatements (`assert_parity`; tests/test_gpu_parity.py's old
`near()` is gone).  With e64 / e32 = relative L2 distance of the HIP result from the fp64 arbiter / the fp32 restatement and
gap = the fp32 restatement's own distance from the fp64 arbiter:
    pass  <=>  e64 <= tol                                            (north_star: 1e-4 of the arbiter)
           or  e64 <= k * gap  and  (e32 <= tol and e64 <= RESTATEMENT_CEIL * tol   or  e64 <= GAP_CEIL * tol)
i.e. where fp32 arithmetic itself cannot resolve the quantity to 1e-4 (ill-conditioned draws; a pixel on the other side of a
discrete threshold), the result may sit as far from the arbiter as k x the fp32 restatement does -- provided it either agrees with
that restatement to the tolerance (the reference operator IS fp32: this is north_star's own criterion; with an absolute ceiling of
RESTATEMENT_CEIL x tol, so that a defect the HIP kernel and the fp32 restatement share in an ill-conditioned regime is not waved through at
any distance -- the far-depth scene sits at 58 x) or stays under the tighter absolute cap.  Every pass through the second line is collected in `GAP_PASSES` and listed at the end of the pytest session
(tests/conftest.py), so a regression that starts leaning on it shows in the GPUTEST tail.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from conftest import rel_l2

TOL = 1e-4          # BASELINE.json north_star: 1e-4 relative L2 on images and gradients
GAP_K = 2.0         # allowance over the fp32 restatement's own distance from the fp64 arbiter
GAP_CEIL = 10.0     # ... capped: the relative branch never accepts more than GAP_CEIL x tol
RESTATEMENT_CEIL = 100.0   # ... and agreeing with the fp32 restatement buys no more than this many x tol from the arbiter
GAP_PASSES = []     # (what, e64, gap) of every case that passed only through the gap branch


def parity_errors(x, o32, o64):
    """(distance of x from the fp64 arbiter, distance of x from the fp32 restatement, fp32-vs-fp64 gap)."""
    return rel_l2(x, o64), rel_l2(x, o32), rel_l2(o32, o64)


def _rule(e64, e32, gap, tol, k):
    return e64 <= tol or (e64 <= k * gap and ((e32 <= tol and e64 <= RESTATEMENT_CEIL * tol) or e64 <= GAP_CEIL * tol))


def parity_ok(x, o32, o64, tol=TOL, k=GAP_K):
    e64, e32, gap = parity_errors(x, o32, o64)
    return _rule(e64, e32, gap, tol, k)


def assert_parity(x, o32, o64, what="", tol=TOL, k=GAP_K):
    """x: HIP result; o32 / o64: the CPU restatement in fp32 / fp64."""
    x, o32, o64 = np.asarray(x), np.asarray(o32), np.asarray(o64)
    if not np.any(o32) and not np.any(o64):
        assert not np.any(x), f"{what}: oracle is all-zero, HIP is not"
        return 0.0, 0.0, 0.0
    e64, e32, gap = parity_errors(x.reshape(o64.shape), o32, o64)
    assert _rule(e64, e32, gap, tol, k), \
        (f"{what}: |hip-f64| {e64:.2e}, |hip-f32| {e32:.2e}, fp32 restatement's own gap |f32-f64| {gap:.2e} (bar {tol:.0e}, or {k:g} x gap "
         f"with |hip-f32| <= {tol:.0e} and under {RESTATEMENT_CEIL * tol:.0e}, or under the cap {GAP_CEIL * tol:.0e})")
    if e64 > tol:
        GAP_PASSES.append((what + (" [= fp32 restatement]" if e32 <= tol else " [capped gap]"), e64, gap))
    return e64, e32, gap


def _np(x, dtype):
    return np.ascontiguousarray(x.detach().cpu().numpy().astype(dtype))


def oracle_view(oracle_mod, g, b, bi, v, H, W, dtype=np.float32, sh_degree=1, discrete="fp32"):
    """One (item, view) of batch `b` with Gaussian dict `g` ((B,P,...) torch tensors, any device) through the oracle.  In fp64 this
    is the ARBITER: continuous arithmetic in fp64 under the fp32 evaluation's discrete decisions (cull, radius, tile rectangle, fp32
    depth key with index ties -- the reference operator's own integer semantics; oracle.forward(discrete_from=...)).
    discrete=None: every decision re-taken in fp64 (rounds 1-5)."""
    from unipre3d_amd import head
    t = math.tan(b.fov_deg * math.pi / 360)
    shs = head.concat_sh(g["features_dc"][bi], g["features_rest"][bi])
    return oracle_mod.forward(_np(g["xyz"][bi], dtype), _np(g["opacity"][bi], dtype), _np(b.world_view[bi, v], dtype),
                              _np(b.full_proj[bi, v], dtype), _np(b.camera_center[bi, v], dtype), _np(b.bg, dtype), H, W, t, t,
                              shs=_np(shs, dtype), scales=_np(g["scaling"][bi], dtype), rotations=_np(g["rotation"][bi], dtype),
                              sh_degree=sh_degree, dtype=dtype, discrete_from=(discrete if np.dtype(dtype) == np.float64 else None))


class _OracleRender(torch.autograd.Function):
    """The CPU oracle as a differentiable renderer (one view) for torch autograd on CPU tensors of either dtype."""

    @staticmethod
    def forward(ctx, oracle_mod, means3D, opacities, scales, rotations, shs, view, proj, campos, bg, H, W, t, sh_degree, exact_aa, antialiasing=True,
                disc32=None):
        """disc32 (fp64 evaluations only): the SAME Gaussians as the fp32 chain produces them (fp32 activations of the fp32 head output:
        (means3D, opacities, scales, rotations, shs) float32 tensors).  The fp32 restatement of those decides cull / radius / rectangle /
        depth order for this fp64 evaluation -- the reference operator's integer semantics."""
        dt = np.float32 if means3D.dtype == torch.float32 else np.float64
        n = lambda x: np.ascontiguousarray(x.detach().numpy())
        d32 = None
        if dt == np.float64 and disc32 is not None:
            f = np.float32
            d32 = oracle_mod.forward(n(disc32[0]), n(disc32[1]), n(view).astype(f), n(proj).astype(f), n(campos).astype(f), n(bg).astype(f), H, W, t, t,
                                     shs=n(disc32[4]), scales=n(disc32[2]), rotations=n(disc32[3]), sh_degree=sh_degree, dtype=f,
                                     exact_aa_grad=exact_aa, antialiasing=antialiasing)
        r = oracle_mod.forward(n(means3D), n(opacities), n(view).astype(dt), n(proj).astype(dt), n(campos).astype(dt), n(bg).astype(dt),
                               H, W, t, t, shs=n(shs), scales=n(scales), rotations=n(rotations), sh_degree=sh_degree, dtype=dt,
                               exact_aa_grad=exact_aa, antialiasing=antialiasing, discrete_from=d32)
        if d32 is not None:
            d32.close()
        ctx.r, ctx.oracle_mod, ctx.tdt = r, oracle_mod, means3D.dtype
        return torch.from_numpy(r.color.copy())

    @staticmethod
    def backward(ctx, gcol):
        go = ctx.oracle_mod.backward(ctx.r, np.ascontiguousarray(gcol.numpy()))
        ctx.r.close()
        f = lambda k: torch.from_numpy(np.ascontiguousarray(go[k])).to(ctx.tdt)
        return (None, f("means3D"), f("opacities"), f("scales"), f("rotations"), f("shs")) + (None,) * 11


def _activate(b, raw, center, sh_degree):
    """Gaussians of ONE item from its raw head output (1,C,P) in raw's dtype, through the reference's activations (head.py, G2)."""
    from unipre3d_amd import head
    if b.level == "object":
        g = head.process_object_output(raw, center, b.offset_scale, sh_degree)
        return {k: x[0] for k, x in g.items()}
    C, P = raw.shape[1], raw.shape[2]
    flat = raw.permute(0, 2, 1).reshape(P, C)
    lists = head.process_scene_output(flat, center.reshape(P, 3), torch.zeros(P, 1, dtype=torch.long), b.offset_scale, sh_degree)
    return {k: x[0] for k, x in lists.items()}


def head_grad_arbiter(oracle_mod, b, bi, v, H, W, n_views_total, loss_kind, dtype=np.float64, sh_degree=1,
                      non_bg_rate=4.0, bg_rate=1.0, exact_aa_grad=False, loss_scale=1.0, antialiasing=True, discrete="fp32"):
    """d loss / d raw[bi] ((C, P), the reference's (B, 23, N) layout) where loss = render loss over ALL n_views_total views'
    pixels but only view (bi, v) differs from its target -- i.e. the per-view contribution the fused kernels can be made to
    isolate by setting gt = rendered for every other view.  Returns (gradient (C,P) ndarray, loss value, image (3,H,W))."""
    from unipre3d_amd import head, losses
    tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    raw = b.raw[bi:bi + 1].detach().cpu().to(tdt).clone().requires_grad_(True)
    center = b.center[bi:bi + 1].detach().cpu().to(tdt)
    g = _activate(b, raw, center, sh_degree)
    t = math.tan(b.fov_deg * math.pi / 360)
    shs = head.concat_sh(g["features_dc"], g["features_rest"])
    disc32 = None
    if tdt == torch.float64 and discrete == "fp32":
        # the arbiter: fp64 arithmetic under the discrete decisions (cull, radius, rectangle, fp32 depth order with index ties) of the
        # fp32 chain -- fp32 activations of the fp32 head output through the fp32 restatement, i.e. what the reference's operator sees
        with torch.no_grad():
            g32 = _activate(b, b.raw[bi:bi + 1].detach().cpu().float(), b.center[bi:bi + 1].detach().cpu().float(), sh_degree)
            disc32 = (g32["xyz"], g32["opacity"], g32["scaling"], g32["rotation"], head.concat_sh(g32["features_dc"], g32["features_rest"]))
    c = lambda x: x.detach().cpu()
    img = _OracleRender.apply(oracle_mod, g["xyz"], g["opacity"], g["scaling"], g["rotation"], shs, c(b.world_view[bi, v]),
                              c(b.full_proj[bi, v]), c(b.camera_center[bi, v]), c(b.bg), H, W, t, sh_degree, exact_aa_grad, antialiasing, disc32)
    gt = c(b.gt[bi, v]).to(tdt)
    white = bool(b.bg[0].item() > 0.5) if loss_kind == "focal_l2" else False
    # the loss of this one view, re-normalised to the whole batch's pixel count (the other views contribute exact zeros)
    loss = losses.render_loss(img[None], gt[None], loss_kind, white_background=white, non_bg_color_loss_rate=non_bg_rate,
                              bg_color_loss_rate=bg_rate) / n_views_total
    # loss_scale != 1: backpropagate loss_scale * loss and divide the gradient again -- the same mathematics under a different
    # rounding, i.e. a second sample of what fp32 arithmetic can resolve on this workload
    (loss * loss_scale).backward()
    return raw.grad[0].numpy().copy() / loss_scale, float(loss.item()), img.detach().numpy().copy()


def head_grad_arbiter_all(oracle_mod, b, H, W, loss_kind, dtype=np.float64, sh_degree=1, input_images=0, loss_scale=1.0, antialiasing=True,
                          discrete="fp32"):
    """d loss / d raw for the WHOLE batch ((B, C, P)): the per-view arbiters summed over every item's views (small shapes only:
    one oracle render per view).  Also returns the loss value."""
    B, V = b.raw.shape[0], b.world_view.shape[1] - input_images
    out, loss = [], 0.0
    for bi in range(B):
        acc = None
        for v in range(input_images, input_images + V):
            g, l, _ = head_grad_arbiter(oracle_mod, b, bi, v, H, W, B * V, loss_kind, dtype, sh_degree, loss_scale=loss_scale, antialiasing=antialiasing,
                                        discrete=discrete)
            acc = g if acc is None else acc + g
            loss += l
        out.append(acc)
    return np.stack(out), loss


def assert_radii(rd, r32, r64=None, what=""):
    """radii are the operator's integer output: BIT-EXACT against the fp32 restatement.  `preprocess_fwd` evaluates the
    cov3D -> J W -> cov2D -> +0.3 -> det -> lambda -> ceil(3 sqrt(lambda)) -> getRect chain in the oracle's operation order with FMA
    contraction off and IEEE divide / sqrt (oracle/raster_oracle.c:317-363 is built -ffp-contract=off), so even at radii of
    10^3..10^4 px, where fp32 and fp64 land on different sides of the ceil for one Gaussian in ~2000, HIP and the fp32 oracle agree.
    Returns (#radii differing from the fp64 arbiter for HIP, same for the fp32 restatement) -- equal by construction."""
    rd, r32 = np.asarray(rd), np.asarray(r32)
    bad = np.flatnonzero(rd != r32)
    assert bad.size == 0, f"{what}: {bad.size} radii differ from the fp32 oracle, first at {bad[:5]}: hip {rd[bad[:5]]} oracle {r32[bad[:5]]}"
    if r64 is None:
        return 0, 0
    r64 = np.asarray(r64)
    return int((rd != r64).sum()), int((r32 != r64).sum())


def worst_fp32_sample(samples, o64):
    """Of several fp32 evaluations of the same quantity (different roundings: dL/dloss scale, summation order), the one farthest
    from the fp64 arbiter: the honest measure of what fp32 can resolve on an ill-conditioned workload (scene-level gradients
    cancel ~1000 x across tiles; a single multi-threaded evaluation of the restatement lands anywhere between 4e-5 and 2e-4)."""
    return max(samples, key=lambda a: rel_l2(a, o64))
