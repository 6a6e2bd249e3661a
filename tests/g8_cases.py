"""Shared driver for the analytic known answers of tests/golden/g8_analytic.npz (see tests/golden/make_g8_analytic.py): runs a
renderer -- the CPU oracle or the HIP operator -- over every case and checks it against the closed forms."""
import numpy as np

from conftest import rel_l2


def run_g8(g, render, tol_img=2e-6, tol_grad=2e-5):
    """render(means, scales, rots, opac, shs, view, proj, campos, bg, S, tanfov, sh_degree, antialiasing, dL_dcolor)
         -> (image (3,S,S), radii (P,), grads dict or None).  All arrays numpy float32."""
    S, t, bg = int(g["S"]), float(g["tanfov"]), g["bg"]
    cam = (g["view"], g["proj"], g["campos"])
    report = {}
    for c in ("A", "B", "C", "F"):
        img, radii, _ = render(g[c + "_means"], g[c + "_scales"], g[c + "_rots"], g[c + "_opac"], g[c + "_shs"], *cam, bg, S, t, 0, True, None)
        assert np.array_equal(radii, g[c + "_radii"]), (c, radii, g[c + "_radii"])
        report[c] = rel_l2(img, g[c + "_image"])
        assert report[c] < tol_img, (c, report[c])
    # A: the centre pixel is rgb * alpha + bg * (1 - alpha)
    img, _, _ = render(g["A_means"], g["A_scales"], g["A_rots"], g["A_opac"], g["A_shs"], *cam, bg, S, t, 0, True, None)
    al = float(g["A_alpha_centre"])
    assert np.allclose(img[:, 16, 16], g["A_rgb"] * al + bg * (1 - al), rtol=2e-6, atol=1e-7)
    # F: the quaternion must NOT be normalised
    img, _, _ = render(g["F_means"], g["F_scales"], g["F_rots"], g["F_opac"], g["F_shs"], *cam, bg, S, t, 0, True, None)
    assert rel_l2(img, g["F_image_if_normalised"]) > 0.05
    # D: SH degree 1 along the six axis directions, colour from the reference's own eval_sh
    for k, d in enumerate(g["D_dirs"]):
        means = (2.0 * d).astype(np.float32)[None]
        img, radii, _ = render(means, g["D_scales"], np.array([[1, 0, 0, 0]], np.float32), g["D_opac"], g["D_shs"], g["D_view"][k],
                               g["D_proj"][k], g["D_campos"][k], bg, S, t, 1, True, None)
        assert int(radii[0]) == int(g["D_radius"])
        e = rel_l2(img, g["D_image"][k])
        assert e < tol_img, ("D", k, e)
        report[f"D{k}"] = e
    # E: gradients of L = <G, image> against central differences of the closed form
    G = g["E_cotangent"]
    for aa in (True, False):
        img, _, gr = render(g["E_means"], g["E_scales"], g["E_rots"], g["E_opac"], g["E_shs"], *cam, bg, S, t, 0, aa, G)
        pre = "E_aa_" if aa else "E_noaa_"
        assert abs(float((G.astype(np.float64) * img).sum()) - float(g[pre + "L"])) < 2e-5 * abs(float(g[pre + "L"]))
        e_o = abs(float(gr["opacities"].reshape(-1)[0]) - float(g[pre + "dL_dopacity"])) / abs(float(g[pre + "dL_dopacity"]))
        e_sh = rel_l2(gr["shs"].reshape(-1)[:3], g[pre + "dL_dsh"])
        assert e_o < tol_grad and e_sh < tol_grad, (aa, e_o, e_sh)
        report[pre + "opacity"], report[pre + "sh"] = e_o, e_sh
        if not aa:
            e_s = rel_l2(gr["scales"].reshape(-1)[:2], g["E_noaa_dL_dscales"][:2])
            e_z = abs(float(gr["means3D"].reshape(-1)[2]) - float(g["E_noaa_dL_dz"])) / abs(float(g["E_noaa_dL_dz"]))
            assert e_s < tol_grad and e_z < tol_grad, (e_s, e_z)
            assert abs(float(gr["scales"].reshape(-1)[2])) < 1e-6 * np.abs(g["E_noaa_dL_dscales"][:2]).max()   # on-axis: sz does not reach the image
            report["E_noaa_scales"], report["E_noaa_z"] = e_s, e_z
    return report
