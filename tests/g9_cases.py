"""Shared driver for the general known answers of tests/golden/g9_general.npz (see tests/golden/make_g9_general.py): off-axis
means, rotated anisotropic covariances (unit and non-unit quaternions), the 1.3 tan(fov/2) clamp active, a 4:3 image rendered
with one field of view for both axes (gaussian_renderer/__init__.py:35-37), SH degree 2-3 through the reference's eval_sh,
tile-rectangle culling, and central-difference gradients for means3D / rotations / scales / opacity (anti-aliasing off).
Runs a renderer -- the CPU oracle or the HIP operator -- over every case."""
import numpy as np

from conftest import rel_l2


def run_g9(g, render, tol_img=2e-6, tol_grad=2e-5):
    """render(means, scales, rots, opac, shs, view, proj, campos, bg, H, W, tanfov, sh_degree, antialiasing, dL_dcolor)
         -> (image (3,H,W), radii (P,), grads dict or None).  All arrays numpy float32."""
    H, W, t, bg = int(g["H"]), int(g["W"]), float(g["tanfov"]), g["bg"]
    cam = (g["view"], g["proj"], g["campos"])
    rep = {}
    for deg in (2, 3):
        shs = np.ascontiguousarray(g["P_shs"][:, :(deg + 1) ** 2])
        img, radii, _ = render(g["P_means"], g["P_scales"], g["P_rots"], g["P_opac"], shs, *cam, bg, H, W, t, deg, True, None)
        assert np.array_equal(radii, g["P_radii"]), (radii, g["P_radii"])
        rep[f"P_deg{deg}"] = rel_l2(img, g[f"P_image_deg{deg}"])
    for c, deg in (("Q", 1), ("R", 0)):
        img, radii, _ = render(g[c + "_means"], g[c + "_scales"], g[c + "_rots"], g[c + "_opac"], g[c + "_shs"], *cam, bg, H, W, t, deg, True, None)
        assert np.array_equal(radii, g[c + "_radii"]), (c, radii, g[c + "_radii"])
        rep[c] = rel_l2(img, g[c + "_image"])
    for k, e in rep.items():
        assert e < tol_img, (k, e)
    G = g["G_cotangent"]
    img, _, gr = render(g["G_means"], g["G_scales"], g["G_rots"], g["G_opac"], g["G_shs"], *cam, bg, H, W, t, 1, False, G)
    Lv = float((G.astype(np.float64) * img).sum())
    assert abs(Lv - float(g["G_L"])) < 2e-5 * abs(float(g["G_L"]))
    for name, key in (("means3D", "G_dL_dmeans"), ("rotations", "G_dL_drots"), ("scales", "G_dL_dscales"), ("opacities", "G_dL_dopac")):
        e = rel_l2(np.asarray(gr[name], np.float64).reshape(g[key].shape), g[key])
        rep["G_" + name] = e
        assert e < tol_grad, (name, e)
    return rep
