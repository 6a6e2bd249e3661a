"""The head-level arbiter of tests/arbiter.py (reference activations -> oracle render -> reference loss, autograd-chained) against
AUTOMATIC differentiation of the whole chain through the float64 PyTorch restatement (oracle/torch_ref.py) -- CPU only."""
import math

import numpy as np
import pytest
import torch

from arbiter import head_grad_arbiter
from conftest import rel_l2
from oracle import torch_ref


@pytest.mark.parametrize("level,kind", [("object", "focal_l2"), ("scene", "l2"), ("object", "l1")])
def test_head_grad_arbiter_matches_autograd_of_the_whole_chain(oracle_mod, level, kind):
    from unipre3d_amd import head, losses, synthetic
    B, P, V, H, W = 2, 40, 3, 40, 56
    b = synthetic.make_batch(B, P, V, H, W, level=level, seed=17)
    bi, v = 1, 2
    ga, la, img = head_grad_arbiter(oracle_mod, b, bi, v, H, W, B * V, kind, np.float64, exact_aa_grad=True)
    raw = b.raw[bi:bi + 1].double().clone().requires_grad_(True)
    if level == "object":
        g = {k: x[0] for k, x in head.process_object_output(raw, b.center[bi:bi + 1].double(), b.offset_scale, 1).items()}
    else:
        flat = raw.permute(0, 2, 1).reshape(P, 23)
        g = {k: x[0] for k, x in head.process_scene_output(flat, b.center[bi].double(), torch.zeros(P, 1, dtype=torch.long), b.offset_scale, 1).items()}
    t = math.tan(b.fov_deg * math.pi / 360)
    col, _, _ = torch_ref.render(g["xyz"], g["opacity"], b.world_view[bi, v].double(), b.full_proj[bi, v].double(), b.camera_center[bi, v].double(),
                                 b.bg.double(), H, W, t, t, shs=head.concat_sh(g["features_dc"], g["features_rest"]), scales=g["scaling"],
                                 rotations=g["rotation"], sh_degree=1, emulate_deviations=True)
    white = bool(b.bg[0] > 0.5) if kind == "focal_l2" else False
    loss = losses.render_loss(col[None], b.gt[bi, v].double()[None], kind, white_background=white) / (B * V)
    loss.backward()
    assert rel_l2(img, col.detach().numpy()) < 1e-12 and abs(la - loss.item()) < 1e-12 * abs(loss.item())
    assert rel_l2(ga, raw.grad[0].numpy()) < 1e-9
    # and the fp32 restatement of the same chain sits where fp32 should
    g32, _, _ = head_grad_arbiter(oracle_mod, b, bi, v, H, W, B * V, kind, np.float32, exact_aa_grad=True)
    assert rel_l2(g32, ga) < 1e-3
