import sys, time, torch
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from unipre3d_amd import synthetic, dp
from unipre3d_amd.fused import render_loss_fused, backward_unit
dev = torch.device("cuda")
cfg = synthetic.CONFIGS["C2"]; B, P, V, H, W = cfg["B"], cfg["P"], cfg["V"], cfg["H"], cfg["W"]
batch = synthetic.make_batch(B, P, V, H, W, seed=42).to(dev)
torch.manual_seed(42)
model = dp.GaussianHead(384, 128).to(dev)
feats = torch.randn(B, P, 384, generator=torch.Generator().manual_seed(1000)).to(dev)
with torch.no_grad():
    raw0 = model(feats); model.final[2].weight.div_(raw0.std()); model.final[2].bias.zero_()
head_out = model(feats, point_major=True).detach().contiguous().requires_grad_(True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ref = None; bad = 0
acc = torch.zeros(2, dtype=torch.float64, device=dev)
t0 = time.time()
for it in range(N):
    head_out.grad = None
    loss, _, _ = render_loss_fused(head_out, batch.center, batch.world_view, batch.full_proj, batch.camera_center, batch.gt, batch.bg, batch.fov_deg, H, W,
                                   level="object", offset_scale=batch.offset_scale, loss_kind="focal_l2", single_pass=True, return_images=False)
    loss.backward() if it % 2 else backward_unit(loss)     # both entries, alternating: bit-identical by construction (g = 1 is an exact multiply)
    sig = torch.stack([loss.detach().double(), head_out.grad.double().abs().sum()])
    if ref is None: ref = sig.clone()
    bad_t = (sig != ref).any()
    acc += torch.stack([bad_t.double(), (~torch.isfinite(sig)).any().double()])
torch.cuda.synchronize()
print("steps %d  mismatching steps %d  non-finite steps %d  (%.1f s)  loss %.9f  device memory %.1f MB (max %.1f MB)" %
      (N, int(acc[0].item()), int(acc[1].item()), time.time() - t0, ref[0].item(), torch.cuda.memory_allocated() / 1e6, torch.cuda.max_memory_allocated() / 1e6))
