"""Host cost of issuing one fused training step (no synchronisation inside the measured loop), per configuration."""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from unipre3d_amd import synthetic
from unipre3d_amd.fused import backward_unit, render_loss_fused
dev = torch.device("cuda")
for name in sys.argv[1:] or ["C1", "C2", "C3", "C4"]:
    c = synthetic.CONFIGS[name]
    b = synthetic.make_batch(c["B"], c["P"], c["V"], c["H"], c["W"], level=c["level"], seed=1).to(dev)
    h = b.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    kind = "focal_l2" if c["level"] == "object" else "l2"
    def step():
        h.grad = None
        loss, _, _ = render_loss_fused(h, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, c["H"], c["W"], level=c["level"],
                                       offset_scale=b.offset_scale, loss_kind=kind, return_images=False)
        backward_unit(loss)
    for _ in range(20): step()
    torch.cuda.synchronize()
    res = []
    for n in (8, 32, 128):
        t0 = time.perf_counter()
        for _ in range(n): step()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        res.append("%d steps: host %.1f us/step, wall %.1f us/step" % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
    print(name, " | ".join(res))
