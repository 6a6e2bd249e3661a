"""Camera matrices consumed by the rasterizer.

Restates, in numpy/float64 -> float32 like the reference, the conventions of
  utils/graphics_utils.py:38-49 (getWorld2View2), :51-62 (getView2World), :64-84 (getProjectionMatrix)
and the assembly in dataset/shapenet.py:297-329:
  world_view = W2V^T ; full_proj = world_view @ P^T ; camera_center = inverse(world_view)[3,:3].
All matrices are stored transposed for ROW-vector use: p_view = [p,1] @ world_view.
"""
from __future__ import annotations

import math

import numpy as np
import torch

# reference values: configs/transformer_pretraining.yaml:11, dataset/shapenet.py:36, configs/dataset/shapenet.yaml
OBJECT_FOV_DEG = 49.13434264120263
OBJECT_ZNEAR, OBJECT_ZFAR = 0.5, 2.0  # SURVEY R0 (object level); configs/dataset/shapenet.yaml
OBJECT_CAMERA_DISTANCE = 1.75
SCENE_FOV_DEG = 57.9516132895
SCENE_ZNEAR, SCENE_ZFAR = 0.2, 10.0


def projection_matrix(znear: float, zfar: float, fov_x: float, fov_y: float) -> torch.Tensor:
    """OpenGL-like perspective matrix P (column-vector form); utils/graphics_utils.py:64-84."""
    tan_y, tan_x = math.tan(fov_y / 2), math.tan(fov_x / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def world_to_view(R: np.ndarray, t: np.ndarray, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    """4x4 world->view (column-vector form), float32; utils/graphics_utils.py:38-49."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate)) * scale
    return np.float32(np.linalg.inv(C2W))


def view_to_world(R: np.ndarray, t: np.ndarray, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    """utils/graphics_utils.py:51-62."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate)) * scale
    return np.float32(C2W)


def assemble_camera(R: np.ndarray, t: np.ndarray, proj: torch.Tensor):
    """(world_view_transform, full_proj_transform, camera_center) as dataset/shapenet.py:305-320."""
    world_view = torch.tensor(world_to_view(R, t)).transpose(0, 1)
    full_proj = world_view.unsqueeze(0).bmm(proj.transpose(0, 1).unsqueeze(0)).squeeze(0)
    center = world_view.inverse()[3, :3]
    return world_view, full_proj, center


def look_at_R_t(eye: np.ndarray, target: np.ndarray, up=(0.0, 0.0, 1.0)):
    """Rotation R (camera-to-world, columns = camera axes, +z forward, +y down) and t = -R^T eye,
    i.e. the (R, T) pair the dataset readers hand to getWorld2View2."""
    eye, target, up = np.asarray(eye, float), np.asarray(target, float), np.asarray(up, float)
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    if np.linalg.norm(right) < 1e-8:
        right = np.cross(fwd, np.array([1.0, 0.0, 0.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)
    t = -R.T @ eye
    return R, t


def orbit_cameras(n_views: int, distance: float, fov_deg: float, znear: float, zfar: float, generator=None,
                  target=(0.0, 0.0, 0.0)):
    """n_views poses on a sphere of radius `distance` looking at `target` (SURVEY 8d synthetic cameras).
    Returns stacked (V,4,4), (V,4,4), (V,3) float32 tensors."""
    g = generator
    fov = math.radians(fov_deg)
    proj = projection_matrix(znear, zfar, fov, fov)
    wv, fp, cc = [], [], []
    for _ in range(n_views):
        az = float(torch.rand((), generator=g)) * 2 * math.pi
        el = (float(torch.rand((), generator=g)) - 0.5) * math.pi * 0.6
        eye = np.asarray(target) + distance * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az),
                                                        math.sin(el)])
        R, t = look_at_R_t(eye, np.asarray(target))
        a, b, c = assemble_camera(R, t, proj)
        wv.append(a); fp.append(b); cc.append(c)
    return torch.stack(wv), torch.stack(fp), torch.stack(cc)


def room_cameras(n_views: int, box=(6.0, 5.0, 3.0), fov_deg: float = SCENE_FOV_DEG, znear: float = SCENE_ZNEAR,
                 zfar: float = SCENE_ZFAR, generator=None):
    """Scene-level synthetic poses: eyes inside a box, looking at random points of it."""
    g = generator
    fov = math.radians(fov_deg)
    proj = projection_matrix(znear, zfar, fov, fov)
    box = np.asarray(box)
    wv, fp, cc = [], [], []
    for _ in range(n_views):
        eye = (torch.rand(3, generator=g).numpy() * 0.5 + 0.25) * box
        tgt = torch.rand(3, generator=g).numpy() * box
        R, t = look_at_R_t(eye, tgt)
        a, b, c = assemble_camera(R, t, proj)
        wv.append(a); fp.append(b); cc.append(c)
    return torch.stack(wv), torch.stack(fp), torch.stack(cc)
