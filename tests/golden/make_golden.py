#!/usr/bin/env python3
"""Generates tests/golden/*.npz by IMPORTING the reference's own Python (never copying it).

Run in the authoring container only (needs /root/reference):   python tests/golden/make_golden.py
The .npz files are data: seeded inputs + the outputs the reference's functions produced for them.

The rasterizer arithmetic itself cannot be pinned this way (third-party CUDA extension absent from
the reference tree); what IS pinned here are all the reference-side pieces
that fix the operator's conventions and inputs:
  G1 camera matrices            utils/graphics_utils.py:38-84, dataset/shapenet.py:305-320
  G2 head activations           model/gaussian_predictor.py:174-181,249-254,279-364
  G3 operator boundary capture  gaussian_renderer/__init__.py:13-104 (settings + kwargs)
  G4 render losses + seeds      utils/loss_utils.py:17-45
  G5 SH polynomials             utils/sh_utils.py:57-116
  G6 projection / covariance    utils/graphics_utils.py:22-30, utils/general_utils.py:171-206
  G7 2D->3D feature fusion      fusion/feat_fusion.py:23-145 (projection, z-buffer, gather; output and gradient)
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Cfg(dict):
    __getattr__ = dict.__getitem__

    def __hasattr__(self, k):
        return k in self


def g1_cameras():
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2, getView2World
    rng = np.random.RandomState(1)
    out = {}
    cases = [(49.13434264120263, 0.5, 2.0), (49.13434264120263, 0.5, 2.0), (49.13434264120263, 0.5, 2.0),
             (57.9516132895, 0.2, 10.0)]
    for i, (fov, zn, zf) in enumerate(cases):
        A = rng.randn(3, 3)
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        T = rng.randn(3) * (1.0 if i < 3 else 3.0)
        P = getProjectionMatrix(znear=zn, zfar=zf, fovX=fov * 2 * np.pi / 360, fovY=fov * 2 * np.pi / 360)
        # dataset/shapenet.py:305-320
        wv = torch.tensor(getWorld2View2(Q, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        vw = torch.tensor(getView2World(Q, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        Pm = P.transpose(0, 1)  # dataset/shapenet.py:113-121 stores the transposed projection
        fp = (wv.unsqueeze(0).bmm(Pm.unsqueeze(0))).squeeze(0)
        cc = wv.inverse()[3, :3]
        out.update({f"R{i}": Q, f"T{i}": T, f"fov{i}": fov, f"znear{i}": zn, f"zfar{i}": zf, f"P{i}": P.numpy(),
                    f"world_view{i}": wv.numpy(), f"view_world{i}": vw.numpy(), f"full_proj{i}": fp.numpy(),
                    f"camera_center{i}": cc.numpy()})
    out["n"] = len(cases)
    np.savez(os.path.join(OUT, "g1_cameras.npz"), **out)


def _load_gaussian_predictor():
    _stub("spconv")
    _stub("spconv.pytorch", SparseSequential=object, SubMConv3d=object, SparseConvTensor=object)
    sys.modules["spconv"].pytorch = sys.modules["spconv.pytorch"]
    _stub("timm"); _stub("timm.models"); _stub("timm.models.layers", trunc_normal_=lambda *a, **k: None)
    _stub("model"); _stub("model.image_predictor", ImageFeaturePredictor=object)
    _stub("model.point_predictor", PointFeaturePredictor=object)
    for extra in ("fusion", "fusion.feat_fusion", "fusion.point_fusion"):
        if extra not in sys.modules:
            try:
                importlib.import_module(extra)
            except Exception:
                _stub(extra, FeatureFusion=object, PointFusion=object)
    spec = importlib.util.spec_from_file_location("ref_gaussian_predictor", os.path.join(REF, "model/gaussian_predictor.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def g2_head():
    mod = _load_gaussian_predictor()
    cls = mod.GaussianSplatPredictor
    self = cls.__new__(cls)
    torch.nn.Module.__init__(self)
    out = {}
    g = torch.Generator().manual_seed(42)
    # object level: (B,23,N) + centers
    self.cfg = _Cfg(model=_Cfg(offset_scale=1.0, isotropic=False, max_sh_degree=1), data=_Cfg(category="shapenet"),
                    opt=_Cfg(level="object"))
    cls._init_activations(self)
    raw = torch.randn(2, 23, 128, generator=g)
    center = torch.randn(2, 128, 3, generator=g) * 0.3
    split = [3, 1, 3, 4, 3, 9]
    parts = list(raw.split(split, dim=1))
    d = cls._process_network_output(self, parts, center=center, is_scene_level=False)
    d = cls._make_contiguous(self, d)
    out["obj_raw"], out["obj_center"], out["split"] = raw.numpy(), center.numpy(), np.array(split)
    for k, v in d.items():
        out["obj_" + k] = v.numpy()
    # scene level: (M,23) + indices
    self.cfg = _Cfg(model=_Cfg(offset_scale=0.2, isotropic=False, max_sh_degree=1), data=_Cfg(category="scannet"),
                    opt=_Cfg(level="scene"))
    raw = torch.randn(1000, 23, generator=g)
    center = torch.rand(1000, 3, generator=g) * 5
    idx = torch.cat([torch.zeros(600, 1), torch.ones(400, 1)]).long()
    idx = torch.cat([idx, torch.zeros(1000, 3).long()], 1)
    d = cls._process_network_output(self, list(raw.split(split, dim=1)), center=center, indices=idx,
                                    is_scene_level=True)
    out["scn_raw"], out["scn_center"], out["scn_indices"] = raw.numpy(), center.numpy(), idx.numpy()
    for k, v in d.items():
        for b, t in enumerate(v):
            out[f"scn_{k}_{b}"] = t.numpy()
    np.savez(os.path.join(OUT, "g2_head.npz"), **out)


def g3_boundary():
    rec = {}

    class Settings:  # recording stub for the 13-field settings object
        def __init__(self, **kw):
            rec["settings_keys"] = list(kw.keys())
            rec["settings"] = kw

    class Rasterizer:
        def __init__(self, raster_settings):
            pass

        def __call__(self, **kw):
            rec["kwargs"] = kw
            P = kw["means3D"].shape[0]
            return torch.zeros(3, 4, 4), torch.zeros(P, dtype=torch.int32), None

    _stub("diff_gaussian_rasterization", GaussianRasterizationSettings=Settings, GaussianRasterizer=Rasterizer)
    orig = torch.zeros_like
    torch.zeros_like = lambda *a, **k: orig(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        import gaussian_renderer
        g = torch.Generator().manual_seed(5)
        P = 128
        pc = {"xyz": torch.randn(P, 3, generator=g), "opacity": torch.rand(P, 1, generator=g),
              "scaling": torch.rand(P, 3, generator=g), "rotation": torch.randn(P, 4, generator=g),
              "features_dc": torch.randn(P, 1, 3, generator=g), "features_rest": torch.randn(P, 3, 3, generator=g)}
        cfg = _Cfg(data=_Cfg(fov=49.13434264120263, training_resolution=128), model=_Cfg(max_sh_degree=1))
        _Cfg.__getattr__ = lambda s, k: dict.__getitem__(s, k) if k in s else (_ for _ in ()).throw(AttributeError(k))
        wv, fp, cc = torch.eye(4), torch.eye(4) * 2, torch.tensor([0.0, 0.0, -1.75])
        res = gaussian_renderer.render_predicted(pc, wv, fp, cc, torch.zeros(3), cfg, focals_pixels=None)
        out = {"settings_keys": np.array(rec["settings_keys"]), "result_keys": np.array(sorted(res.keys()))}
        for k, v in rec["settings"].items():
            out["s_" + k] = v.numpy() if torch.is_tensor(v) else np.array(v)
        for k, v in rec["kwargs"].items():
            out["kw_isnone_" + k] = np.array(v is None)
            if v is not None:
                out["kw_" + k] = v.detach().numpy()
        for k, v in pc.items():
            out["pc_" + k] = v.numpy()
        # scene-level config has separate height/width and one fov for both axes
        cfg2 = _Cfg(data=_Cfg(fov=57.9516132895, training_height=120, training_width=160), model=_Cfg(max_sh_degree=1))
        gaussian_renderer.render_predicted(pc, wv, fp, cc, torch.ones(3), cfg2)
        for k in ("image_height", "image_width", "tanfovx", "tanfovy"):
            out["s2_" + k] = np.array(rec["settings"][k])
        np.savez(os.path.join(OUT, "g3_boundary.npz"), **out)
    finally:
        torch.zeros_like = orig


def g4_losses():
    from utils.loss_utils import focal_l2_loss, l2_loss, l1_loss
    g = torch.Generator().manual_seed(9)
    out = {}
    for tag, bgv in (("black", 0.0), ("white", 1.0)):
        x = torch.rand(6, 3, 16, 16, generator=g, requires_grad=True)
        gt = torch.rand(6, 3, 16, 16, generator=g)
        m = torch.rand(6, 1, 16, 16, generator=g) < 0.4
        gt = torch.where(m, torch.full_like(gt, bgv), gt)
        gt[0, 0, 0, 0] = bgv + 5e-7  # inside atol
        bg = torch.tensor([bgv] * 3)
        loss = focal_l2_loss(x, gt, bg, 4, 1)
        (gx,) = torch.autograd.grad(loss, x)
        out.update({f"{tag}_x": x.detach().numpy(), f"{tag}_gt": gt.numpy(), f"{tag}_focal": loss.item(),
                    f"{tag}_focal_grad": gx.numpy(), f"{tag}_l2": l2_loss(x, gt).item(), f"{tag}_l1": l1_loss(x, gt).item()})
    np.savez(os.path.join(OUT, "g4_losses.npz"), **out)


def g5_sh():
    from utils.sh_utils import eval_sh, C0, C1, C2, C3
    g = torch.Generator().manual_seed(11)
    out = {"C0": C0, "C1": C1, "C2": np.array(C2), "C3": np.array(C3)}
    for deg in range(4):
        K = (deg + 1) ** 2
        sh_pk3 = torch.randn(64, K, 3, generator=g, dtype=torch.float64)  # rasterizer layout (P,K,3)
        dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g, dtype=torch.float64), dim=-1)
        res = eval_sh(deg, sh_pk3.transpose(1, 2), dirs)  # reference layout (...,C,K)
        out[f"sh{deg}"], out[f"dirs{deg}"], out[f"rgb{deg}"] = sh_pk3.numpy(), dirs.numpy(), res.numpy()
    np.savez(os.path.join(OUT, "g5_sh.npz"), **out)


def g6_geometry():
    from utils.graphics_utils import geom_transform_points
    import utils.general_utils as gu
    oz = torch.zeros
    torch.zeros = lambda *a, **k: oz(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        g = torch.Generator().manual_seed(13)
        pts = torch.randn(50, 3, generator=g)
        M = torch.randn(4, 4, generator=g)
        M[3, 3] = 3.0
        ndc = geom_transform_points(pts, M)
        q = torch.randn(50, 4, generator=g)
        s = torch.rand(50, 3, generator=g) + 0.1
        R = gu.build_rotation(q)
        L = gu.build_scaling_rotation(s, q)
        cov = L @ L.transpose(1, 2)
        np.savez(os.path.join(OUT, "g6_geometry.npz"), pts=pts.numpy(), M=M.numpy(), ndc=ndc.numpy(), q=q.numpy(),
                 s=s.numpy(), R=R.numpy(), cov=cov.numpy())
    finally:
        torch.zeros = oz


def g7_feature_fusion():
    """fusion/feat_fusion.py (pure torch) run as is on the CPU: output + gradient w.r.t. the image features."""
    spec = importlib.util.spec_from_file_location("ref_feat_fusion", os.path.join(REF, "fusion/feat_fusion.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for tag, (B, N, Cx, C, H, W, cls) in {"sq": (2, 64, 8, 6, 16, 16, True), "rect": (3, 40, 4, 5, 12, 20, False)}.items():
        g = torch.Generator().manual_seed(17 + H)
        center = torch.randn(B, N, 3, generator=g) * 0.4
        center[:, 5] = center[:, 4]                       # exact duplicate -> depth tie at one pixel
        center[:, 9, 2] = -5.0                            # behind the camera
        center[:, 11, 0] = 30.0                           # projects outside the image
        # row-vector view->world matrices like dataset/shapenet.py:311-313 (transpose of a rigid transform)
        A = torch.eye(3).repeat(B, 1, 1) + 0.15 * torch.randn(B, 3, 3, generator=g)   # camera looks roughly along +z
        Q, Rr = torch.linalg.qr(A)
        Q = Q * torch.sign(torch.diagonal(Rr, dim1=1, dim2=2)).unsqueeze(1)
        c2w = torch.eye(4).repeat(B, 1, 1)
        c2w[:, :3, :3] = Q
        c2w[:, :3, 3] = torch.tensor([0.0, 0.0, -2.0]) + 0.1 * torch.randn(B, 3, generator=g)
        c2w_rowvec = c2w.transpose(1, 2).contiguous()
        focal = (H / 2.0) / math.tan(math.radians(49.13434264120263 / 2.0))
        intr = np.zeros((3, 4)); intr[2, 2] = 1; intr[0, 0] = intr[1, 1] = focal; intr[0, 2] = H / 2.0; intr[1, 2] = W / 2.0
        feat = torch.randn(B, C, H, W, generator=g, requires_grad=True)
        x = torch.randn(B, N + (1 if cls else 0), Cx, generator=g)
        ff = mod.FeatureFusion(torch.nn.Identity())
        y = ff(x, center, feat, c2w_rowvec, intr)
        w = torch.randn(y.shape, generator=g)
        (gfeat,) = torch.autograd.grad((y * w).sum(), feat)
        pi_xy, depth = ff.project_points_to_image(center, c2w_rowvec, intr)
        out.update({f"{tag}_x": x.numpy(), f"{tag}_center": center.numpy(), f"{tag}_c2w": c2w_rowvec.numpy(), f"{tag}_intr": intr,
                    f"{tag}_feat": feat.detach().numpy(), f"{tag}_out": y.detach().numpy(), f"{tag}_w": w.numpy(),
                    f"{tag}_gfeat": gfeat.numpy(), f"{tag}_pix": pi_xy.numpy(), f"{tag}_depth": depth.numpy()})
    np.savez(os.path.join(OUT, "g7_feature_fusion.npz"), **out)


if __name__ == "__main__":
    g1_cameras(); g5_sh(); g4_losses(); g6_geometry(); g7_feature_fusion(); g2_head(); g3_boundary()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
