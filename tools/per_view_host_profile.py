"""Where the host time of the per-view drop-in route goes (C2: 128 forward + 128 backward operator calls per step)."""
import cProfile, os, pstats, sys, time, types
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from unipre3d_amd import head, losses, renderer, synthetic
from unipre3d_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda")
c = synthetic.CONFIGS["C2"]; B, P, V, H, W = c["B"], c["P"], c["V"], c["H"], c["W"]
b = synthetic.make_batch(B, P, V, H, W, seed=42).to(dev)
cfg = types.SimpleNamespace(data=types.SimpleNamespace(fov=b.fov_deg, training_resolution=H), model=types.SimpleNamespace(max_sh_degree=1))
raw = b.raw.clone().requires_grad_(True)
gt = b.gt.reshape(B * V, 3, H, W)
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0

def step():
    raw.grad = None
    t0 = time.perf_counter(); gs = head.process_object_output(raw, b.center, b.offset_scale); tick("activations", t0)
    imgs = []
    for i in range(B):
        t0 = time.perf_counter(); pc = {k: v[i].contiguous() for k, v in gs.items()}; tick("slice item", t0)
        for v in range(V):
            t0 = time.perf_counter()
            imgs.append(renderer.render_predicted(pc, b.world_view[i, v], b.full_proj[i, v], b.camera_center[i, v], b.bg, cfg)["render"])
            tick("render_predicted", t0)
    t0 = time.perf_counter(); rendered = torch.stack(imgs); loss = losses.render_loss(rendered, gt, "focal_l2"); tick("stack+loss", t0)
    t0 = time.perf_counter(); loss.backward(); tick("backward", t0)
    return loss

for _ in range(3): step()
torch.cuda.synchronize(); T.clear()
N = 5
t0 = time.perf_counter()
for _ in range(N): step()
host = time.perf_counter() - t0
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("ms/step %.2f (host issue %.2f)" % (1e3 * el / N, 1e3 * host / N))
for k, v in T.items(): print("  %-18s %.2f ms/step" % (k, 1e3 * v / N))
# inside render_predicted: the wrapper's own torch ops vs the operator call
pc = {k: v[0].detach().contiguous().requires_grad_(True) for k, v in head.process_object_output(raw, b.center, b.offset_scale).items()}
import math
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    r = 1e6 * (time.perf_counter() - t0) / n; torch.cuda.synchronize(); return r
tan = math.tan(b.fov_deg * math.pi / 360)
st = GaussianRasterizationSettings(H, W, tan, tan, b.bg, 1.0, b.world_view[0, 0], b.full_proj[0, 0], 1, b.camera_center[0, 0], False, False, True)
shs = head.concat_sh(pc["features_dc"], pc["features_rest"]).detach().requires_grad_(True)
ssp = torch.zeros_like(pc["xyz"], requires_grad=True) + 0
rast = GaussianRasterizer(st)
with torch.no_grad():
    print("no-grad operator call            %.1f us" % t(lambda: rast(means3D=pc["xyz"], means2D=ssp, shs=shs, opacities=pc["opacity"], scales=pc["scaling"], rotations=pc["rotation"])))
print("grad-mode operator call (fwd)    %.1f us" % t(lambda: rast(means3D=pc["xyz"], means2D=ssp, shs=shs, opacities=pc["opacity"], scales=pc["scaling"], rotations=pc["rotation"])))
print("zeros_like(...)+0                %.1f us" % t(lambda: torch.zeros_like(pc["xyz"], requires_grad=True) + 0))
print("settings tuple                   %.1f us" % t(lambda: GaussianRasterizationSettings(H, W, tan, tan, b.bg, 1.0, b.world_view[0, 0], b.full_proj[0, 0], 1, b.camera_center[0, 0], False, False, True)))
print("camera indexing x3               %.1f us" % t(lambda: (b.world_view[0, 0], b.full_proj[0, 0], b.camera_center[0, 0])))
print("GaussianRasterizer(...)          %.1f us" % t(lambda: GaussianRasterizer(st)))
print("concat_sh                        %.1f us" % t(lambda: head.concat_sh(pc["features_dc"], pc["features_rest"])))
print("render_predicted (whole, fwd)    %.1f us" % t(lambda: renderer.render_predicted(pc, b.world_view[0, 0], b.full_proj[0, 0], b.camera_center[0, 0], b.bg, cfg)))
def fb():
    out = renderer.render_predicted(pc, b.world_view[0, 0], b.full_proj[0, 0], b.camera_center[0, 0], b.bg, cfg)["render"]
    out.sum().backward()
print("render_predicted + sum().backward %.1f us" % t(fb, 500))
def fb2():
    out, _, _ = rast(means3D=pc["xyz"], means2D=ssp, shs=shs, opacities=pc["opacity"], scales=pc["scaling"], rotations=pc["rotation"])
    out.sum().backward()
print("operator + sum().backward         %.1f us" % t(fb2, 500))
pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
