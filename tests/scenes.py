"""Shared seeded test scenes (single view) in the operator's layout."""
import numpy as np
import torch

from unipre3d_amd import synthetic

DIFF_KEYS = ("means3D", "opacities", "scales", "rotations", "shs")


def scene(P=48, H=40, W=56, seed=3, level="object", compact=False, deg=1, **kw):
    return synthetic.single_view_scene(P=P, H=H, W=W, seed=seed, level=level, compact=compact, sh_degree=deg, **kw)


def to_numpy(sc):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in sc.items()}


def cotangents(H, W, seed=7, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(3, H, W, generator=g, dtype=dtype), torch.randn(1, H, W, generator=g, dtype=dtype)


# (P, H, W, level, compact, sh degree, seed): sizes the oracle finishes in well under a second
SMALL_CASES = [
    (48, 40, 56, "object", False, 1, 3),
    (128, 128, 128, "object", False, 1, 1),     # BASELINE config C1 geometry (one view)
    (300, 120, 160, "scene", False, 2, 2),      # reference's scene-level 160x120 render, ragged last tile row
    (1000, 100, 100, "object", True, 3, 4),     # compact splats: diverse binning
    (64, 33, 17, "object", False, 0, 8),        # image smaller than 2 tiles wide, SH degree 0
    (257, 64, 64, "scene", True, 1, 9),         # P just above one 256-batch
]
