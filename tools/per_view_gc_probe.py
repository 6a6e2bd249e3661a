"""Are the slow outlier steps of the per-view route Python garbage collections?  Same loop as tools/per_view_route.py, per-step times
with the collector on / off."""
import gc, os, sys, time, types
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from unipre3d_amd import head, losses, renderer, synthetic
dev = torch.device("cuda")
c = synthetic.CONFIGS["C2"]; B, P, V, H, W = c["B"], c["P"], c["V"], c["H"], c["W"]
b = synthetic.make_batch(B, P, V, H, W, seed=42).to(dev)
cfg = types.SimpleNamespace(data=types.SimpleNamespace(fov=b.fov_deg, training_resolution=H), model=types.SimpleNamespace(max_sh_degree=1))
raw = b.raw.clone().requires_grad_(True)
gt = b.gt.reshape(B * V, 3, H, W)
def step():
    raw.grad = None
    gs = head.process_object_output(raw, b.center, b.offset_scale)
    imgs = []
    for i in range(B):
        pc = {k: v[i].contiguous() for k, v in gs.items()}
        for v in range(V):
            imgs.append(renderer.render_predicted(pc, b.world_view[i, v], b.full_proj[i, v], b.camera_center[i, v], b.bg, cfg)["render"])
    loss = losses.render_loss(torch.stack(imgs), gt, "focal_l2")
    loss.backward()
for mode in ("gc on", "gc off", "gc on"):
    (gc.disable if mode == "gc off" else gc.enable)()
    for _ in range(3): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(mode, "per-step ms:", " ".join("%.1f" % t for t in ts), "| mean %.2f median %.2f" % (sum(ts) / len(ts), sorted(ts)[10]))
