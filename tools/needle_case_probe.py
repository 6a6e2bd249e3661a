"""Per-channel look at a soak case that exceeded the parity rule (profiles/r06/soak_stress_seeds2.log, case 18: two Gaussians, needle scales, 12 x 23 image):
d loss / d head_out from HIP (single- and two-pass), the fp32 restatement and the arbiter, channel by channel."""
import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from unipre3d_amd import fused, synthetic
import arbiter
from oracle import oracle as om
om.build()
dev = torch.device("cuda")
np.set_printoptions(precision=3, linewidth=220)
for (B, P, V, H, W, seed, ops) in ((2, 1, 1, 12, 23, 507298146, ((3, 2, 6.0), (4, 3, 3.0), (5, 3, -1.0), (6, 3, -1.0))),):
    b = synthetic.make_batch(B, P, V, H, W, level="object", seed=seed, compact=False)
    for ch, st, val in ops:
        b.raw[:, ch, ::st] = val
    bd = b.to(dev)
    res = []
    for sp in (True, False):
        h = bd.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
        loss, img, _ = fused.render_loss_fused(h, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg, H, W, level="object",
                                               offset_scale=bd.offset_scale, loss_kind="l2", single_pass=sp)
        loss.backward()
        res.append(h.grad.permute(0, 2, 1).cpu().numpy())
    a64, _ = arbiter.head_grad_arbiter_all(om, b, H, W, "l2", np.float64)
    a32s = [arbiter.head_grad_arbiter_all(om, b, H, W, "l2", np.float32, loss_scale=sc)[0] for sc in (1.0, 3.0, 1.0 / 3.0)]
    nrm = np.linalg.norm(a64)
    print("rel: hip1 %.2e hip2 %.2e f32 %s" % (np.linalg.norm(res[0] - a64) / nrm, np.linalg.norm(res[1] - a64) / nrm, ["%.2e" % (np.linalg.norm(a - a64) / nrm) for a in a32s]))
    for bi in range(B):
        print("item", bi)
        print(" f64      ", a64[bi][:, 0])
        print(" hip - f64", res[0][bi][:, 0] - a64[bi][:, 0])
        print(" f32 - f64", a32s[0][bi][:, 0] - a64[bi][:, 0])
    g = synthetic.gaussians_from_batch(b)
    print("scales", g["scaling"][:, 0].numpy(), "opacity", g["opacity"][:, 0].numpy().ravel(), "xyz", g["xyz"][:, 0].numpy())
