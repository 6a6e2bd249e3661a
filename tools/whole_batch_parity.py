"""Whole-batch parity of the fused step at a BASELINE config's full per-GPU shape: loss and d loss / d head_out of EVERY object and view (what bench.py
times) against the CPU oracle chained through the reference's activations and loss, fp32 and fp64, under the one parity rule
(tests/arbiter.py::assert_parity).  The GPU tests sample 1 - 3 (item, view) pairs per scene-level config because the oracle needs minutes for a
whole batch there; this tool spends them once per round and its log is committed (profiles/r06/whole_batch_parity.log).
usage: python tools/whole_batch_parity.py C4 [C5 ...]      (U3D_WBP_SEED=<n>: another draw of the synthetic batch; default 42)"""
import os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from unipre3d_amd import fused, synthetic
from conftest import rel_l2
import arbiter
from oracle import oracle as oracle_mod
oracle_mod.build()
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["C4"]:
    cfg = synthetic.CONFIGS[name]
    B, P, V, H, W, level = cfg["B"], cfg["P"], cfg["V"], cfg["H"], cfg["W"], cfg["level"]
    kind = "focal_l2" if level == "object" else "l2"
    b = synthetic.make_batch(B, P, V, H, W, level=level, seed=int(os.environ.get("U3D_WBP_SEED", "42")))
    bd = b.to(dev)
    h = bd.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    loss, _, _ = fused.render_loss_fused(h, bd.center, bd.world_view, bd.full_proj, bd.camera_center, bd.gt, bd.bg, bd.fov_deg, H, W, level=level,
                                         offset_scale=bd.offset_scale, loss_kind=kind, return_images=False)
    loss.backward()
    torch.cuda.synchronize()
    t0 = time.time()
    a64, l64 = arbiter.head_grad_arbiter_all(oracle_mod, b, H, W, kind, np.float64)
    # fp32 restatement under two roundings (dL/dloss scale): the farther one from fp64 is what fp32 resolves on this workload
    s32 = [arbiter.head_grad_arbiter_all(oracle_mod, b, H, W, kind, np.float32, loss_scale=sc)[0] for sc in (1.0, 3.0)]
    a32 = arbiter.worst_fp32_sample(s32, a64)
    hg = h.grad.permute(0, 2, 1).cpu().numpy()
    e64, e32, gap = arbiter.parity_errors(hg, a32, a64)
    ok = arbiter.parity_ok(hg, a32, a64)
    per_item = [rel_l2(hg[i], a64[i]) for i in range(B)]
    verdict = "PASS (within 1e-4 of the arbiter)" if e64 <= arbiter.TOL else ("PASS (through the fp32-gap branch)" if ok else "FAIL")
    print(f"{name}: {B} x {V} views, P = {P}, {H}x{W}, loss {kind}: loss hip {loss.item():.8f} arbiter {l64:.8f} (rel {abs(loss.item() - l64) / abs(l64):.1e}); "
          f"d(head_out) whole batch |hip-f64| {e64:.2e} |hip-f32| {e32:.2e} fp32 restatement's own |f32-f64| {gap:.2e} -> parity rule {verdict}; "
          f"per object {[f'{x:.2e}' for x in per_item]}; oracle time {time.time() - t0:.0f} s", flush=True)
