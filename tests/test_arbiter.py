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
    # (discrete=None: the torch restatement decides cull / radius / order in fp64 itself, so the comparison is like for like)
    ga, la, img = head_grad_arbiter(oracle_mod, b, bi, v, H, W, B * V, kind, np.float64, exact_aa_grad=True, discrete=None)
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


@pytest.mark.parametrize("level,kind,fname", [("object", "focal_l2", "g10_chain.npz"), ("scene", "l2", "g10_chain_scene.npz")])
def test_g10_chain_known_answers_pin_the_arbiter(golden, oracle_mod, level, kind, fname):
    """tests/golden/make_g10_chain.py: loss and d loss / d head_out of the WHOLE chain (reference activations incl. the across-point
    quaternion quirk -> render -> focal-L2), evaluated by an independent float64 numpy transcription and central differences over all
    P x 23 head entries, without oracle/ or the product.  The fp64 arbiter chain (head.py -> oracle -> losses.py) must hit it --
    the same fixture the fused HIP step is held to in tests/test_gpu_batched.py."""
    import torch
    from arbiter import head_grad_arbiter_all
    from unipre3d_amd import synthetic
    g = golden(fname)
    H, W = int(g["H"]), int(g["W"])
    T = lambda k: torch.from_numpy(np.asarray(g[k]))
    b = synthetic.SyntheticBatch(raw=T("head_out").permute(0, 2, 1).contiguous(), center=T("center"), world_view=T("world_view"),
                                 full_proj=T("full_proj"), camera_center=T("camera_center"), gt=T("gt"), bg=T("bg"), fov_deg=float(g["fov_deg"]),
                                 level=level, offset_scale=float(g["offset_scale"]))
    for dt, tol in ((np.float64, 1e-7), (np.float32, 2e-5)):
        a, l = head_grad_arbiter_all(oracle_mod, b, H, W, kind, dt, antialiasing=False)
        e = rel_l2(a[0].T, g["d_head"][0])
        assert abs(l - float(g["loss"])) <= (1e-12 if dt == np.float64 else 1e-6) * max(1.0, abs(float(g["loss"]))) or abs(l - float(g["loss"])) < 1e-7, (l, float(g["loss"]))
        assert e < tol, (dt, e)
        print("g10 chain", level, dt.__name__, f"{e:.1e}")


def test_the_one_parity_rule():
    """tests/arbiter.py::assert_parity on synthetic vectors: 1e-4 of the arbiter passes; beyond it only under k x the fp32
    restatement's own gap AND (agreement with that restatement to 1e-4, or the absolute cap); the second line is logged."""
    import arbiter
    o64 = np.ones(1000)
    mk = lambda e: o64 * (1.0 + e)              # a vector at relative distance e from the arbiter
    n0 = len(arbiter.GAP_PASSES)
    arbiter.assert_parity(mk(5e-5), mk(2e-5), o64, "within tol")
    assert len(arbiter.GAP_PASSES) == n0
    arbiter.assert_parity(mk(3.6e-3), mk(3.6e-3 + 1e-6), o64, "equals the fp32 restatement, far from fp64")      # e32 <= tol, under 100 x tol
    arbiter.assert_parity(mk(5e-4), mk(3e-4), o64, "capped gap")                                                  # e32 = 2e-4 > tol, under the cap
    assert len(arbiter.GAP_PASSES) == n0 + 2 and "[= fp32 restatement]" in arbiter.GAP_PASSES[n0][0] and "[capped gap]" in arbiter.GAP_PASSES[n0 + 1][0]
    del arbiter.GAP_PASSES[n0:]
    for x, o32 in ((mk(3e-3), mk(2e-3)),        # beyond the cap and not equal to the fp32 restatement
                   (mk(5e-4), mk(2e-4)),        # farther from the arbiter than 2 x the restatement's own gap
                   (mk(2e-4), mk(1e-6)),        # the restatement resolves the quantity: no allowance
                   (mk(2e-2), mk(2e-2 + 1e-6))):  # equal to the fp32 restatement but 200 x tol from the arbiter: beyond the absolute ceiling
        with pytest.raises(AssertionError):
            arbiter.assert_parity(x, o32, o64, "must fail")
    assert len(arbiter.GAP_PASSES) == n0
