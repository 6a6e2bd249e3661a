"""GPU stress: random shapes, fused single-pass vs fused two-pass vs torch+operator chain; finite-ness; repeatability."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from unipre3d_amd import fused, step, synthetic
from conftest import rel_l2
dev = torch.device("cuda")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
worst = {"img": 0.0, "loss": 0.0, "g12": 0.0, "g1u": 0.0}
bad = 0
t0 = time.time()
for it in range(N):
    B = int(rng.integers(1, 4)); P = int(rng.choice([1, 3, 17, 63, 64, 65, 128, 200, 256, 257, 300, 511, 700, 1500, 5000]))
    V = int(rng.integers(1, 4)); H = int(rng.integers(1, 130)); W = int(rng.integers(1, 130))
    level = ("object", "scene")[int(rng.integers(0, 2))]; kind = ("focal_l2", "l2", "l1")[int(rng.integers(0, 3))]
    compact = bool(rng.integers(0, 2)); seed = int(rng.integers(0, 1 << 30))
    b = synthetic.make_batch(B, P, V, H, W, level=level, seed=seed, compact=compact).to(dev)
    variant = int(rng.integers(0, 4))      # 1: some opacities sigmoid(6) (0.99 clamp live), 2: needles (nearly singular conics), 3: both
    if variant & 1:
        b.raw[:, 3, ::int(rng.integers(2, 9))] = 6.0
    if variant & 2:
        st = int(rng.integers(3, 11))
        b.raw[:, 4, ::st] = 3.0; b.raw[:, 5, ::st] = -1.0; b.raw[:, 6, ::st] = -1.0
    res = []
    for sp in (True, False):
        h = b.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
        loss, img, radii = fused.render_loss_fused(h, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, H, W,
                                                   level=level, offset_scale=b.offset_scale, loss_kind=kind, single_pass=sp)
        loss.backward()
        res.append((loss.detach().clone(), img.clone(), h.grad.clone()))
    raw = b.raw.clone().requires_grad_(True)
    loss_u, img_u = step.render_loss_forward(raw, b, H, W, 0, kind)
    loss_u.backward()
    gu = raw.grad.permute(0, 2, 1)
    torch.cuda.synchronize()
    fin = all(torch.isfinite(x).all().item() for r in res for x in r) and torch.isfinite(gu).all().item()
    e_img = rel_l2(res[0][1].cpu().numpy(), img_u.detach().cpu().numpy())
    e_loss = abs(res[0][0].item() - loss_u.item()) / max(abs(loss_u.item()), 1e-12)
    gn = max(gu.abs().max().item(), 1e-30)
    e12 = (res[0][2] - res[1][2]).abs().max().item() / gn
    e1u = rel_l2(res[0][2].cpu().numpy(), gu.cpu().numpy()) if gu.abs().sum().item() > 0 else res[0][2].abs().sum().item()
    worst = {"img": max(worst["img"], e_img), "loss": max(worst["loss"], e_loss), "g12": max(worst["g12"], e12), "g1u": max(worst["g1u"], e1u)}
    # fused vs torch-activation chain: the l1 seed is sign(d) (a 1e-6 image difference flips it where d ~ 0) and needles amplify the
    # 1-ulp differences of the two exp/tanh implementations through det(cov); both stay within 5e-3 (same with any earlier build)
    tol_chain = 1e-2 if (variant & 2) else (5e-3 if kind == "l1" else 1e-3)
    if (not fin) or e_img > 1e-4 or e_loss > 1e-4 or e1u > tol_chain or e12 > 1e-3:
        bad += 1
        print("CASE", it, dict(B=B, P=P, V=V, H=H, W=W, level=level, kind=kind, compact=compact, seed=seed, variant=variant), "finite", fin,
              "img %.2e loss %.2e g(single vs two-pass) %.2e g(single vs chain) %.2e" % (e_img, e_loss, e12, e1u))
print("cases %d bad %d worst %s  (%.1f s)" % (N, bad, {k: "%.2e" % v for k, v in worst.items()}, time.time() - t0))
