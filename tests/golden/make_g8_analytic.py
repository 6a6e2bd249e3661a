#!/usr/bin/env python3
"""Generates tests/golden/g8_analytic.npz: AUTHOR-INDEPENDENT known answers for the rasterizer arithmetic.

The rasterizer's source is absent from the reference tree, so oracle/ and the HIP kernels would otherwise
share one reading of the published algorithm.  The cases below are configurations in which that algorithm collapses to closed
forms that can be derived on paper; this script evaluates those closed forms in float64 numpy and NEVER imports oracle/ or
unipre3d_amd/ -- only the reference's own camera and SH functions (utils/graphics_utils.py, utils/sh_utils.py), imported from
/root/reference.  Both the CPU oracle (tests/test_oracle_golden.py) and the HIP path (tests/test_gpu_parity.py) must reproduce them.

Geometry shared by all cases: a camera whose optical axis passes through every Gaussian centre (so the view-space mean is
t = (0, 0, z): the Jacobian's third column vanishes, J = diag(f/z, f/z)), an odd square image of S = 33 pixels (the projected
centre is exactly the integer pixel (16, 16): pix = ((ndc + 1) S - 1) / 2 with ndc = 0), f = S / (2 tan(fov/2)).
For a Gaussian with world covariance diag(sx^2, sy^2, sz^2) in a frame aligned with the camera:
    cov2D = diag(a, c),  a = (f sx / z)^2,  c = (f sy / z)^2           (b = 0)
    low-pass:  a' = a + 0.3, c' = c + 0.3;   anti-aliasing factor  aa = sqrt(max(2.5e-5, a c / (a' c')))
    conic = (1/a', 0, 1/c');   radius = ceil(3 sqrt(max(a', c') [+ sqrt(0.1) when a' == c']))   (general: lambda = mid +- sqrt(max(0.1, mid^2 - det)))
    pixel (16 + dx, 16 + dy):  alpha = min(0.99, o * aa * exp(-dx^2 / (2 a') - dy^2 / (2 c'))),  skipped when alpha < 1/255
    image = sum_i rgb_i alpha_i prod_{j<i} (1 - alpha_j)  +  bg prod_i (1 - alpha_i)   (front to back by z)
    rgb = max(0, eval_sh(deg, sh, dir) + 0.5),  dir = (mean - campos) / |mean - campos|
    culled when z <= 0.2.
Cases: A one isotropic Gaussian; B two co-axial Gaussians (occlusion order, transmittance product); C one behind z = 0.2
(culled: background only, radius 0); D SH degree 1 seen along the six axis directions (colour through the reference's own
eval_sh); F a NON-UNIT quaternion (0,0,0,2), which the rasterizer must not normalise: R = diag(-7,-7,1), so the splat is 7 x wider
than a normalising implementation would draw it; E gradients of L = <G, image> for a fixed random cotangent G, by central
differences of the closed form (opacity and SH with anti-aliasing on; scales and depth with anti-aliasing OFF, because the
published backward deliberately differentiates the anti-aliasing factor inexactly, DESIGN.md DEV(vi)).
"""
import math
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

S = 33
FOV_DEG = 49.13434264120263
ZNEAR, ZFAR = 0.5, 100.0


def camera_looking_along(direction):
    """World->view for a camera at the origin whose +z axis is `direction` (unit), through the reference's own functions."""
    import torch
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2
    d = np.asarray(direction, dtype=np.float64)
    d = d / np.linalg.norm(d)
    up = np.array([0.0, 1.0, 0.0]) if abs(d[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
    x = np.cross(up, d); x /= np.linalg.norm(x)
    y = np.cross(d, x)
    R = np.stack([x, y, d], axis=1)               # columns: camera axes in world coordinates (c2w rotation)
    T = np.zeros(3)
    w2v = getWorld2View2(R, T)                    # utils/graphics_utils.py:38-49 (takes the c2w rotation, like the datasets do)
    fov = FOV_DEG * 2 * math.pi / 360
    P = getProjectionMatrix(znear=ZNEAR, zfar=ZFAR, fovX=fov, fovY=fov).transpose(0, 1)     # :64-84
    world_view = torch.tensor(w2v, dtype=torch.float32).transpose(0, 1)
    full_proj = (world_view.unsqueeze(0).bmm(P.unsqueeze(0))).squeeze(0)                     # dataset/shapenet.py:305-320
    cam_center = world_view.inverse()[3, :3]
    return world_view.numpy().astype(np.float32), full_proj.numpy().astype(np.float32), cam_center.numpy().astype(np.float32)


def sh_colour(deg, sh_pm3, direction):
    """rgb = max(0, eval_sh + 0.5) with the REFERENCE's eval_sh (utils/sh_utils.py:57-116; it wants (..., 3, M))."""
    import torch
    from utils.sh_utils import eval_sh
    sh = torch.tensor(sh_pm3, dtype=torch.float64).transpose(-1, -2)[None]      # (1, 3, M)
    d = torch.tensor(direction, dtype=torch.float64)[None]
    d = d / d.norm(dim=-1, keepdim=True)
    return np.maximum(eval_sh(deg, sh, d)[0].numpy() + 0.5, 0.0)


F = S / (2 * math.tan(FOV_DEG * math.pi / 360))


def splat(o, sx, sy, z, antialiasing=True):
    """(alpha map (S,S) float64 with the 1/255 skip and the 0.99 clamp applied, radius) of an axis-aligned on-axis Gaussian."""
    a, c = (F * sx / z) ** 2, (F * sy / z) ** 2
    a2, c2 = a + 0.3, c + 0.3
    aa = math.sqrt(max(0.000025, a * c / (a2 * c2))) if antialiasing else 1.0
    mid, det = 0.5 * (a2 + c2), a2 * c2
    lam = mid + math.sqrt(max(0.1, mid * mid - det))
    radius = int(math.ceil(3 * math.sqrt(lam)))
    d = np.arange(S, dtype=np.float64) - (S - 1) / 2
    dx, dy = d[None, :], d[:, None]
    alpha = np.minimum(0.99, o * aa * np.exp(-dx * dx / (2 * a2) - dy * dy / (2 * c2)))
    alpha = np.where(alpha < 1.0 / 255.0, 0.0, alpha)
    return alpha, radius


def composite(layers, bg):
    """layers: [(rgb (3,), alpha (S,S))] front to back -> image (3,S,S)."""
    T = np.ones((S, S))
    img = np.zeros((3, S, S))
    for rgb, alpha in layers:
        img += rgb[:, None, None] * (alpha * T)[None]
        T = T * (1 - alpha)
    return img + np.asarray(bg, dtype=np.float64)[:, None, None] * T[None]


def main():
    out = {"S": S, "fov_deg": FOV_DEG, "tanfov": math.tan(FOV_DEG * math.pi / 360)}
    bg = np.array([0.1, 0.3, 0.6])
    out["bg"] = bg.astype(np.float32)
    wv, fp, cc = camera_looking_along([0, 0, 1])
    out["view"], out["proj"], out["campos"] = wv, fp, cc
    unit_q = np.array([1, 0, 0, 0], dtype=np.float32)

    # ---- A: one isotropic Gaussian, SH degree 0 ----
    shA = np.array([[0.9, -0.4, 0.3]])
    rgbA = sh_colour(0, shA, [0, 0, 1])
    alphaA, radA = splat(0.7, 0.5, 0.5, 2.0)
    out.update(A_means=np.array([[0, 0, 2.0]], np.float32), A_scales=np.full((1, 3), 0.5, np.float32), A_rots=unit_q[None],
               A_opac=np.array([[0.7]], np.float32), A_shs=shA[:, None, :].astype(np.float32), A_image=composite([(rgbA, alphaA)], bg),
               A_radii=np.array([radA], np.int32), A_alpha_centre=alphaA[16, 16], A_rgb=rgbA)

    # ---- B: two co-axial Gaussians, the SECOND in the arrays is nearer (order must come from depth, not from the index) ----
    shB = np.array([[1.2, 0.2, -0.5], [-0.6, 0.8, 0.4]])
    rgbB = [sh_colour(0, shB[i:i + 1], [0, 0, 1]) for i in range(2)]
    aB0, rB0 = splat(0.6, 0.4, 0.4, 3.0)
    aB1, rB1 = splat(0.8, 0.2, 0.2, 1.5)
    out.update(B_means=np.array([[0, 0, 3.0], [0, 0, 1.5]], np.float32), B_scales=np.array([[0.4] * 3, [0.2] * 3], np.float32),
               B_rots=np.stack([unit_q, unit_q]), B_opac=np.array([[0.6], [0.8]], np.float32), B_shs=shB[:, None, :].astype(np.float32),
               B_image=composite([(rgbB[1], aB1), (rgbB[0], aB0)], bg), B_radii=np.array([rB0, rB1], np.int32))

    # ---- C: behind the near cull (z_view <= 0.2) and behind the camera ----
    out.update(C_means=np.array([[0, 0, 0.15], [0, 0, -1.0], [0, 0, 0.19]], np.float32), C_scales=np.full((3, 3), 0.05, np.float32),
               C_rots=np.stack([unit_q] * 3), C_opac=np.full((3, 1), 0.9, np.float32), C_shs=np.ones((3, 1, 3), np.float32),
               C_image=composite([], bg), C_radii=np.zeros(3, np.int32))

    # ---- D: SH degree 1, the Gaussian seen along the six axis directions (camera at the origin looks along `d`, Gaussian at 2 d) ----
    rng = np.random.RandomState(8)
    shD = rng.randn(4, 3) * 0.6
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    alphaD, radD = splat(0.65, 0.3, 0.3, 2.0)
    D_view, D_proj, D_cam, D_img, D_rgb = [], [], [], [], []
    for d in dirs:
        v, p, c = camera_looking_along(d)
        rgb = sh_colour(1, shD, d)
        D_view.append(v); D_proj.append(p); D_cam.append(c); D_rgb.append(rgb); D_img.append(composite([(rgb, alphaD)], bg))
    out.update(D_dirs=dirs, D_view=np.stack(D_view), D_proj=np.stack(D_proj), D_campos=np.stack(D_cam), D_shs=shD[None].astype(np.float32),
               D_scales=np.full((1, 3), 0.3, np.float32), D_opac=np.array([[0.65]], np.float32), D_rgb=np.stack(D_rgb),
               D_image=np.stack(D_img), D_radius=radD)

    # ---- F: non-unit quaternion (0,0,0,2): R = diag(1-2*4, 1-2*4, 1) = diag(-7,-7,1) -> Sigma = diag(49 sx^2, 49 sy^2, sz^2) ----
    alphaF, radF = splat(0.5, 7 * 0.05, 7 * 0.08, 2.5)
    alphaF_norm, _ = splat(0.5, 0.05, 0.08, 2.5)          # what a normalising implementation would draw
    out.update(F_means=np.array([[0, 0, 2.5]], np.float32), F_scales=np.array([[0.05, 0.08, 0.3]], np.float32),
               F_rots=np.array([[0, 0, 0, 2]], np.float32), F_opac=np.array([[0.5]], np.float32), F_shs=shA[:, None, :].astype(np.float32),
               F_image=composite([(rgbA, alphaF)], bg), F_image_if_normalised=composite([(rgbA, alphaF_norm)], bg), F_radii=np.array([radF], np.int32))

    # ---- E: gradients of L = <G, image> by central differences of the closed form (case A geometry, anisotropic scales) ----
    G = rng.randn(3, S, S)
    out["E_cotangent"] = G.astype(np.float32)
    base = dict(o=0.7, sx=0.5, sy=0.35, z=2.0, sh=shA.copy())

    def L(antialiasing, **kw):
        p = dict(base, **kw)
        rgb = np.maximum(0.28209479177387814 * p["sh"][0] + 0.5, 0.0)       # degree 0: eval_sh = C0 * sh[0]
        alpha, _ = splat(p["o"], p["sx"], p["sy"], p["z"], antialiasing)
        return float((G * composite([(rgb, alpha)], bg)).sum())

    def fd(antialiasing, name, h=1e-6, idx=None):
        def at(v):
            if idx is None:
                return L(antialiasing, **{name: base[name] + v})
            sh = base["sh"].copy(); sh[0, idx] += v
            return L(antialiasing, sh=sh)
        return (at(h) - at(-h)) / (2 * h)

    out.update(E_means=np.array([[0, 0, 2.0]], np.float32), E_scales=np.array([[0.5, 0.35, 0.2]], np.float32), E_rots=unit_q[None],
               E_opac=np.array([[0.7]], np.float32), E_shs=shA[:, None, :].astype(np.float32),
               E_aa_dL_dopacity=fd(True, "o"), E_aa_dL_dsh=np.array([fd(True, "sh", idx=i) for i in range(3)]),
               E_noaa_dL_dopacity=fd(False, "o"), E_noaa_dL_dsh=np.array([fd(False, "sh", idx=i) for i in range(3)]),
               E_noaa_dL_dscales=np.array([fd(False, "sx"), fd(False, "sy"), 0.0]), E_noaa_dL_dz=fd(False, "z"),
               E_aa_L=L(True), E_noaa_L=L(False))
    np.savez_compressed(os.path.join(OUT, "g8_analytic.npz"), **out)
    print("wrote g8_analytic.npz:", {k: (np.asarray(v).shape if hasattr(v, "shape") or isinstance(v, (list, tuple)) else v) for k, v in out.items() if k.endswith(("_L", "radii", "radius", "centre"))})


if __name__ == "__main__":
    main()
