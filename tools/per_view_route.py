"""The unmodified reference loop (one operator call per object and view, train_network.py:418-446) against the batched routes, C2."""
import sys, os, time, types, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from unipre3d_amd import synthetic, renderer, losses, fused
dev = torch.device("cuda")
cfgc = synthetic.CONFIGS["C2"]; B, P, V, H, W = cfgc["B"], cfgc["P"], cfgc["V"], cfgc["H"], cfgc["W"]
b = synthetic.make_batch(B, P, V, H, W, seed=42).to(dev)
cfg = types.SimpleNamespace(data=types.SimpleNamespace(fov=b.fov_deg, training_resolution=H), model=types.SimpleNamespace(max_sh_degree=1))
raw = b.raw.clone().requires_grad_(True)

def per_view_step():
    raw.grad = None
    from unipre3d_amd import head
    gs = head.process_object_output(raw, b.center, b.offset_scale)
    imgs = []
    for i in range(B):
        pc = {k: v[i].contiguous() for k, v in gs.items()}
        for v in range(V):
            imgs.append(renderer.render_predicted(pc, b.world_view[i, v], b.full_proj[i, v], b.camera_center[i, v], b.bg, cfg)["render"])
    rendered = torch.stack(imgs)
    loss = losses.focal_l2_loss(rendered, b.gt.reshape(B * V, 3, H, W), b.bg, 4.0, 1.0)
    loss.backward()
    return loss

for name, fn, n in (("per-view operator loop (reference call pattern)", per_view_step, 5),):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): l = fn()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print("%-50s %.2f ms/step  %.0f views/s  loss %.6f" % (name, ms, B * V / ms * 1e3, float(l)))
