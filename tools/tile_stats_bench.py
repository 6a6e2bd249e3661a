"""tile_last statistics on the exact Gaussians the bench renders (head output of the seeded GaussianHead, not batch.raw)."""
import sys, math, torch, ctypes
sys.path.insert(0, '/root/repo')
from unipre3d_amd import synthetic, head, _lib, dp
from unipre3d_amd.rasterizer import _Plan
dev = torch.device("cuda")
for cfgname in sys.argv[1:] or ["C2"]:
    cfg = synthetic.CONFIGS[cfgname]
    B, P, V, H, W, level = cfg["B"], cfg["P"], cfg["V"], cfg["H"], cfg["W"], cfg["level"]
    batch = synthetic.make_batch(B, P, V, H, W, level=level, seed=42).to(dev)
    feat_dim = 384 if level == "object" else 64
    torch.manual_seed(42)
    model = dp.GaussianHead(feat_dim, 128 if level == "object" else 32).to(dev)
    feats = torch.randn(B, P, feat_dim, generator=torch.Generator().manual_seed(1000)).to(dev)
    with torch.no_grad():
        raw0 = model(feats); model.final[2].weight.div_(raw0.std()); model.final[2].bias.zero_()
        head_out = model(feats, point_major=True)
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
    npx = NV * H * W; al = lambda n: ((n + 255) // 256) * 256; T = ((H + 15) // 16) * ((W + 15) // 16)
    tl = image[al(npx * 4) * 2: al(npx * 4) * 2 + NV * T * 4].view(torch.int32).bitwise_and(0x7fffffff).float()   # bit 31: the tile ran the plain loop variant
    plain = (image[al(npx * 4) * 2: al(npx * 4) * 2 + NV * T * 4].view(torch.int32) < 0).float().mean().item()   # bit 31
    lim = image[al(npx * 4):][: npx * 4].view(torch.int32)
    print(cfgname, "tiles on the plain loop variant %.3f ;" % plain, "tile_last mean %.1f median %.1f max %d frac>64 %.3f ; unsaturated pixels %d ; visible/view %.0f" % (
        tl.mean().item(), tl.median().item(), int(tl.max().item()), (tl > 64).float().mean().item(), int((lim == -1).sum()), (radii > 0).float().sum(1).mean().item()))
