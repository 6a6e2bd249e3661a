"""Differentiable float64 PyTorch restatement of the rasterizer (TEST INFRASTRUCTURE ONLY).

Purpose: obtain gradients by *automatic differentiation* of the forward semantics
 so that the hand-derived analytic backward of oracle/raster_oracle.c and of the
HIP kernels (R5/R6) can be checked against true derivatives.  The analytic backward deviates
from the true derivative on purpose in a few places (R6 i-v); `emulate_deviations=True`
reproduces those with detach tricks so that the comparison can be made everywhere:

  (i)   gradient passes straight through alpha = min(0.99, o*G)
  (ii)  the +-1.3 tan(fov) clamp of t.x/t.z, t.y/t.z: the clamped coordinate is a constant
  (iii) colour channels clamped at 0 get no gradient            (true derivative as well)
  (iv)  culling / 1/255 skip / 1e-4 stop are constants          (true derivative a.e.)
  (v)   anti-aliasing scale floor 2.5e-5 gives zero gradient    (true derivative as well)

The anti-aliasing closed form evaluated after the low-pass (DEV(vi) in raster_oracle.c) is NOT
emulated; compare against this module with exact_aa_grad=True.

Conventions pinned against the reference: row-vector matrices (utils/graphics_utils.py:22-30),
quaternion order (r,x,y,z) (utils/general_utils.py:178-193), SH constants (utils/sh_utils.py).
PARITY UNPINNED for the rasterizer arithmetic itself.
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def quat_to_rotmat_raw(q: torch.Tensor) -> torch.Tensor:
    """(P,4) un-normalised (r,x,y,z) -> (P,3,3); utils/general_utils.py:185-193 without :172-176."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def eval_sh_pm3(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh (P,M,3), dirs (P,3) unit -> (P,3).  Same polynomial as utils/sh_utils.py:57-116 (which
    takes (...,C,K) layout; the rasterizer consumes (P,K,3))."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
                   + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def render(means3D, opacities, viewmatrix, projmatrix, campos, bg, H, W, tanfovx, tanfovy, shs=None,
           colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0, scale_modifier=1.0,
           antialiasing=True, emulate_deviations=True):
    """Returns (color (3,H,W), invdepth (1,H,W), radii (P,) int).  All inputs float64 tensors."""
    dt = means3D.dtype
    P = means3D.shape[0]
    view, proj = viewmatrix, projmatrix
    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
    radii = torch.zeros(P, dtype=torch.int32)
    color = torch.zeros(3, H, W, dtype=dt)
    if P == 0:
        return color + bg[:, None, None], torch.zeros(1, H, W, dtype=dt), radii
    p_view = means3D @ view[:3, :3] + view[3, :3]
    hom = torch.cat([means3D, torch.ones(P, 1, dtype=dt)], 1) @ proj
    p_w = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * p_w[:, None]
    vis = p_view[:, 2] > 0.2
    if cov3D_precomp is not None:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                            -1).reshape(P, 3, 3)
    else:
        Rm = quat_to_rotmat_raw(rotations)
        Mx = Rm * (scale_modifier * scales)[:, None, :]
        Sigma = Mx @ Mx.transpose(1, 2)
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    tz = p_view[:, 2]
    tz_safe = torch.where(vis, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    rx, ry = p_view[:, 0] / tz_safe, p_view[:, 1] / tz_safe
    if emulate_deviations:
        tx = torch.where((rx < -limx) | (rx > limx), (rx.clamp(-limx, limx) * tz_safe).detach(), p_view[:, 0])
        ty = torch.where((ry < -limy) | (ry > limy), (ry.clamp(-limy, limy) * tz_safe).detach(), p_view[:, 1])
    else:
        tx, ty = rx.clamp(-limx, limx) * tz_safe, ry.clamp(-limy, limy) * tz_safe
    zeros = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zeros, -fx * tx / (tz_safe * tz_safe),
                     zeros, fy / tz_safe, -fy * ty / (tz_safe * tz_safe)], -1).reshape(P, 2, 3)
    Wm = view[:3, :3].t()
    M2 = J @ Wm
    cov2 = M2 @ Sigma @ M2.transpose(1, 2)
    a0, b, c0 = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    det0 = a0 * c0 - b * b
    a, c = a0 + 0.3, c0 + 0.3
    det1 = a * c - b * b
    ok = vis & (det1 != 0)
    det1s = torch.where(ok, det1, torch.ones_like(det1))
    aa = torch.sqrt(torch.clamp_min(det0 / det1s, 0.000025)) if antialiasing else torch.ones_like(det1)
    A, B, Cc = c / det1s, -b / det1s, a / det1s
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det1, 0.1))
    lam2 = mid - torch.sqrt(torch.clamp_min(mid * mid - det1, 0.1))
    rad = torch.ceil(3.0 * torch.sqrt(torch.maximum(lam, lam2))).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5

    def _ti(v, g):  # C-style (int) truncation then clamp to [0,g]
        return torch.clamp(torch.trunc(v), 0, g).to(torch.int64)

    rminx = _ti((px.detach() - rad) / 16.0, tiles_x)
    rminy = _ti((py.detach() - rad) / 16.0, tiles_y)
    rmaxx = _ti((px.detach() + rad + 15.0) / 16.0, tiles_x)
    rmaxy = _ti((py.detach() + rad + 15.0) / 16.0, tiles_y)
    ok = ok & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
    radii = torch.where(ok, rad.to(torch.int32), torch.zeros_like(radii))
    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh_pm3(sh_degree, shs, d) + 0.5, 0.0)
    op = opacities.reshape(P) * aa
    depth_key = torch.where(ok, tz.detach(), torch.full_like(tz, float("inf")))
    order = torch.sort(depth_key, stable=True).indices
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    tyi, txi = (ys // 16).to(torch.int64), (xs // 16).to(torch.int64)
    T = torch.ones(H, W, dtype=dt)
    Cacc = torch.zeros(3, H, W, dtype=dt)
    Dacc = torch.zeros(H, W, dtype=dt)
    done = torch.zeros(H, W, dtype=torch.bool)
    for g in order.tolist():
        if not bool(ok[g]):
            break
        in_tile = (txi >= rminx[g]) & (txi < rmaxx[g]) & (tyi >= rminy[g]) & (tyi < rmaxy[g])
        dx, dy = px[g] - xs, py[g] - ys
        power = -0.5 * (A[g] * dx * dx + Cc[g] * dy * dy) - B[g] * dx * dy
        G = torch.exp(torch.clamp_max(power, 0.0))
        alpha_raw = op[g] * G
        if emulate_deviations:
            alpha = alpha_raw + (torch.clamp_max(alpha_raw, 0.99) - alpha_raw).detach()
        else:
            alpha = torch.clamp_max(alpha_raw, 0.99)
        valid = in_tile & (~done) & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        stop = valid & (test_T < 0.0001)
        done = done | stop
        contrib = valid & (~stop)
        w = torch.where(contrib, alpha * T, torch.zeros_like(T))
        Cacc = Cacc + rgb[g][:, None, None] * w
        Dacc = Dacc + w / tz[g]
        T = torch.where(contrib, test_T, T)
    color = Cacc + T[None] * bg[:, None, None]
    return color, Dacc[None], radii
