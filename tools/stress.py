"""GPU stress: random shapes.  PARITY: every case the oracle finishes quickly (P <= 2000, l2 / focal-l2) goes through the one parity rule
(tests/arbiter.py::assert_parity: image of a sampled view and d loss / d head_out of the WHOLE batch against oracle/raster_oracle.c in fp32
and fp64, chained through the reference's activations and loss).  DETERMINISM / CONSISTENCY (not parity evidence): fused single-pass vs fused
two-pass vs torch + operator chain -- two HIP routes against each other, tolerance stated in the log line."""
import os, sys, time, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from unipre3d_amd import fused, step, synthetic
from conftest import rel_l2
import arbiter
from oracle import oracle as oracle_mod
oracle_mod.build()
dev = torch.device("cuda")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
worst = {"img": 0.0, "loss": 0.0, "g12": 0.0, "g1u": 0.0}
bad = 0
orc = {"cases": 0, "bad": 0, "worst_img_e64": 0.0, "worst_grad_e64": 0.0, "unresolvable": 0, "unresolvable_worst_ratio": 0.0}
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
    # ---- parity against the oracle (the one rule), where the oracle is quick ----
    if P <= 2000 and kind != "l1" and B * V <= 6:
        orc["cases"] += 1
        hb = b.to(torch.device("cpu"))
        try:
            bi, v = int(rng.integers(0, B)), int(rng.integers(0, V))
            g_act = synthetic.gaussians_from_batch(hb)
            r32, r64 = arbiter.oracle_view(oracle_mod, g_act, hb, bi, v, H, W, np.float32), arbiter.oracle_view(oracle_mod, g_act, hb, bi, v, H, W, np.float64)
            ei = arbiter.assert_parity(res[0][1][bi * V + v].cpu().numpy(), r32.color, r64.color, f"case {it} image ({bi},{v})")
            r32.close(); r64.close()
            a64, l64 = arbiter.head_grad_arbiter_all(oracle_mod, hb, H, W, kind, np.float64)
            # the fp32 restatement under three roundings of the same mathematics (dL/dloss scale): the farthest from fp64 is what fp32 resolves
            a32 = arbiter.worst_fp32_sample([arbiter.head_grad_arbiter_all(oracle_mod, hb, H, W, kind, np.float32, loss_scale=sc)[0] for sc in (1.0, 3.0, 1.0 / 3.0)], a64)
            assert abs(res[0][0].item() - l64) <= 1e-4 * max(abs(l64), 1e-12), ("loss", res[0][0].item(), l64)
            hg = res[0][2].permute(0, 2, 1).cpu().numpy()
            e64, e32, gap = arbiter.parity_errors(hg, a32, a64)
            if gap > arbiter.GAP_CEIL * arbiter.TOL:
                # the fp32 restatement ITSELF sits beyond the rule's absolute cap (needle / far-depth draws): no fp32 implementation can be
                # held to it; recorded apart, held to k x the restatement's own distance only
                orc["unresolvable"] += 1
                orc["unresolvable_worst_ratio"] = max(orc["unresolvable_worst_ratio"], e64 / gap)
                # OUTSIDE the parity rule (not counted as a pass of it): held to k x the restatement's own distance AND to the rule's absolute
                # ceiling RESTATEMENT_CEIL x tol, so that a defect both fp32 implementations share cannot hide at any distance
                assert e64 <= arbiter.GAP_K * gap and e64 <= arbiter.RESTATEMENT_CEIL * arbiter.TOL, \
                    (f"case {it} d(head_out) [fp32-unresolvable]: |hip-f64| {e64:.2e} vs {arbiter.GAP_K:g} x gap {gap:.2e} and the ceiling "
                     f"{arbiter.RESTATEMENT_CEIL * arbiter.TOL:.0e}")
                eg = (0.0, e32, gap)
            else:
                eg = arbiter.assert_parity(hg, a32, a64, f"case {it} d(head_out)")
            orc["worst_img_e64"] = max(orc["worst_img_e64"], ei[0]); orc["worst_grad_e64"] = max(orc["worst_grad_e64"], eg[0])
        except AssertionError as ex:
            orc["bad"] += 1
            print("ORACLE CASE", it, dict(B=B, P=P, V=V, H=H, W=W, level=level, kind=kind, compact=compact, seed=seed, variant=variant), str(ex)[:400])
    # fused vs torch-activation chain: the l1 seed is sign(d) (a 1e-6 image difference flips it where d ~ 0) and needles amplify the
    # 1-ulp differences of the two exp/tanh implementations through det(cov); both stay within 5e-3 (same with any earlier build)
    tol_chain = 1e-2 if (variant & 2) else (5e-3 if kind == "l1" else 1e-3)
    if (not fin) or e_img > 1e-4 or e_loss > 1e-4 or e1u > tol_chain or e12 > 1e-3:
        bad += 1
        print("CASE", it, dict(B=B, P=P, V=V, H=H, W=W, level=level, kind=kind, compact=compact, seed=seed, variant=variant), "finite", fin,
              "img %.2e loss %.2e g(single vs two-pass) %.2e g(single vs chain) %.2e" % (e_img, e_loss, e12, e1u))
print("vs oracle (tests/arbiter.py::assert_parity, tol %.0e): cases %d bad %d, worst e64 image %.2e gradient %.2e; OUTSIDE THE RULE (not passes of it): %d gradients "
      "where the fp32 restatement itself is > %.0e from the arbiter (held to %g x its distance under the 100 x tol ceiling: worst ratio %.2f); gap passes %d %s"
      % (arbiter.TOL, orc["cases"], orc["bad"], orc["worst_img_e64"], orc["worst_grad_e64"], orc["unresolvable"], arbiter.GAP_CEIL * arbiter.TOL, arbiter.GAP_K,
         orc["unresolvable_worst_ratio"], len(arbiter.GAP_PASSES),
         [(w, "%.2e" % e, "%.2e" % g) for (w, e, g) in arbiter.GAP_PASSES[:8]]))
print("HIP vs HIP (determinism / route consistency only, NOT parity): cases %d bad %d worst %s; accepted: image / loss 1e-4, single- vs two-pass "
      "gradient 1e-3 of the largest entry, fused vs operator chain 1e-3 (l1: 5e-3, needles: 1e-2)  (%.1f s)"
      % (N, bad, {k: "%.2e" % v for k, v in worst.items()}, time.time() - t0))
