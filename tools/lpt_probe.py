import sys, math, torch, ctypes
sys.path.insert(0, '/root/repo')
from unipre3d_amd import synthetic, head, _lib, dp
from unipre3d_amd.rasterizer import _Plan
from unipre3d_amd.fused import render_loss_fused
dev = torch.device("cuda")
cfg = synthetic.CONFIGS["C2"]; B, P, V, H, W = cfg["B"], cfg["P"], cfg["V"], cfg["H"], cfg["W"]
batch = synthetic.make_batch(B, P, V, H, W, seed=42).to(dev)
torch.manual_seed(42)
model = dp.GaussianHead(384, 128).to(dev)
feats = torch.randn(B, P, 384, generator=torch.Generator().manual_seed(1000)).to(dev)
with torch.no_grad():
    raw0 = model(feats); model.final[2].weight.div_(raw0.std()); model.final[2].bias.zero_()
head_out = model(feats, point_major=True).detach().contiguous()

def run(order, tag):
    ho = head_out[order].contiguous().requires_grad_(True)
    cen, wv, fp, cc, gt = batch.center[order].contiguous(), batch.world_view[order].contiguous(), batch.full_proj[order].contiguous(), batch.camera_center[order].contiguous(), batch.gt[order].contiguous()
    def f():
        ho.grad = None
        loss, _, _ = render_loss_fused(ho, cen, wv, fp, cc, gt, batch.bg, batch.fov_deg, H, W, level="object", offset_scale=batch.offset_scale,
                                       loss_kind="focal_l2", single_pass=True, return_images=False)
        loss.backward()
    for _ in range(50): f()
    torch.cuda.synchronize()
    _lib.profile_begin(4096, ("render_fb",))
    for _ in range(50): f()
    torch.cuda.synchronize()
    ms, n = _lib.profile_end()["render_fb"]
    print("%-22s render_fb %.1f us" % (tag, 1e3 * ms / n))

# per-item cost from the operator forward's tile_last
g = synthetic.gaussians_from_batch(synthetic.SyntheticBatch(**dict(batch.__dict__, raw=head_out.permute(0, 2, 1))))
t = math.tan(batch.fov_deg * math.pi / 360)
plan = _Plan(B, V, P, H, W, t, t, 1.0, 1, 4, _lib.FLAG_ANTIALIASING)
NV = B * V
color = torch.empty((NV, 3, H, W), device=dev); radii = torch.zeros((NV, P), dtype=torch.int32, device=dev)
geom = torch.empty(plan.sizes.geom_bytes, dtype=torch.uint8, device=dev); binning = torch.empty(plan.sizes.binning_bytes, dtype=torch.uint8, device=dev)
image = torch.zeros(plan.sizes.image_bytes, dtype=torch.uint8, device=dev)
p = _lib.ptr; c = lambda x: x.contiguous()
shs = head.concat_sh(g["features_dc"], g["features_rest"])
_lib.load().u3d_rasterize_forward(ctypes.byref(plan.desc), p(batch.bg), p(c(g["xyz"])), p(c(shs)), p(None), p(c(g["opacity"])), p(c(g["scaling"])),
    p(c(g["rotation"])), p(None), p(c(batch.world_view).reshape(NV, 16)), p(c(batch.full_proj).reshape(NV, 16)), p(c(batch.camera_center).reshape(NV, 3)),
    p(color), p(None), p(radii), p(geom), p(binning), p(image), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
npx = NV * H * W; al = lambda n: ((n + 255) // 256) * 256; T = (H // 16) * (W // 16)
tl = image[al(npx * 4) * 2: al(npx * 4) * 2 + NV * T * 4].view(torch.int32).bitwise_and(0x7fffffff).float().reshape(B, V, T)
cost = tl.sum((1, 2))
print("per-item cost min %.0f max %.0f mean %.0f ; per-view min %.1f max %.1f" % (cost.min(), cost.max(), cost.mean(), tl.mean(2).min(), tl.mean(2).max()))
ident = torch.arange(B, device=dev)
run(ident, "original order")
run(torch.argsort(cost, descending=True), "heavy items first")
run(torch.argsort(cost, descending=False), "light items first")
run(ident, "original order (again)")
