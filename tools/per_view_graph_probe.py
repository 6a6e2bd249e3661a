"""the reference's UNCHANGED per-view double loop (train_network.py:418-446: one render_predicted per object and
view, torch.stack, torch loss, loss.backward()) captured ONCE into a HIP graph (torch.cuda.CUDAGraph, static inputs) and replayed.
Prints eager ms/step, graph-replay ms/step, and whether the replay on fresh inputs equals the eager step bit for bit."""
import os, sys, time, types
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", __file__.rsplit("/tools", 1)[0]))
from unipre3d_amd import head, losses, renderer, synthetic

dev = torch.device("cuda:0")
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
c = synthetic.CONFIGS[cfgname]
B, P, V, H, W = c["B"], c["P"], c["V"], c["H"], c["W"]
b, b2 = synthetic.make_batch(B, P, V, H, W, seed=1).to(dev), synthetic.make_batch(B, P, V, H, W, seed=2).to(dev)
cfg = types.SimpleNamespace(data=types.SimpleNamespace(fov=b.fov_deg, training_resolution=H), model=types.SimpleNamespace(max_sh_degree=1))
static_raw = b.raw.clone().requires_grad_(True)
gt = b.gt.reshape(B * V, 3, H, W)


def loop(raw):
    gs = head.process_object_output(raw, b.center, b.offset_scale)
    imgs = []
    for i in range(B):
        pc = {k: v[i].contiguous() for k, v in gs.items()}
        for v in range(V):
            imgs.append(renderer.render_predicted(pc, b.world_view[i, v], b.full_proj[i, v], b.camera_center[i, v], b.bg, cfg)["render"])
    return losses.render_loss(torch.stack(imgs), gt, "focal_l2")


def eager():
    static_raw.grad = None
    l = loop(static_raw)
    l.backward()
    return l


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        eager()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
g = torch.cuda.CUDAGraph()
static_raw.grad = None
with torch.cuda.graph(g):
    static_loss = loop(static_raw)
    static_loss.backward()
torch.cuda.synchronize()
print(f"{cfgname}: captured {2 * B * V} operator calls in {time.perf_counter() - t0:.2f} s")
with torch.no_grad():
    static_raw.copy_(b2.raw)
g.replay()
torch.cuda.synchronize()
l_graph, g_graph = static_loss.detach().clone(), static_raw.grad.detach().clone()
r2 = b2.raw.clone().requires_grad_(True)
l2 = loop(r2)
l2.backward()
torch.cuda.synchronize()
print("  graph == eager on fresh inputs:", bool(torch.equal(l_graph, l2.detach())), bool(torch.equal(g_graph, r2.grad)))
for label, fn, n in (("eager", eager, 10), ("graph replay", g.replay, 50)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"  {label}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per step")
