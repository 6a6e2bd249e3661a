import sys, math, torch, ctypes
sys.path.insert(0, '/root/repo')
from unipre3d_amd import synthetic, head, _lib
from unipre3d_amd.rasterizer import _Plan
dev = torch.device("cuda")
for cfgname in sys.argv[1:] or ["C2"]:
    cfg = synthetic.CONFIGS[cfgname]
    B, P, V, H, W = cfg["B"], cfg["P"], cfg["V"], cfg["H"], cfg["W"]
    b = synthetic.make_batch(B, P, V, H, W, level=cfg["level"], seed=42).to(dev)
    g = synthetic.gaussians_from_batch(b)
    t = math.tan(b.fov_deg * math.pi / 360)
    plan = _Plan(B, V, P, H, W, t, t, 1.0, 1, 4, _lib.FLAG_ANTIALIASING)
    NV = B * V
    color = torch.empty((NV, 3, H, W), device=dev); radii = torch.zeros((NV, P), dtype=torch.int32, device=dev)
    geom = torch.empty(plan.sizes.geom_bytes, dtype=torch.uint8, device=dev)
    binning = torch.empty(plan.sizes.binning_bytes, dtype=torch.uint8, device=dev)
    image = torch.zeros(plan.sizes.image_bytes, dtype=torch.uint8, device=dev)
    p = _lib.ptr; c = lambda x: x.contiguous()
    shs = head.concat_sh(g["features_dc"], g["features_rest"])
    rc = _lib.load().u3d_rasterize_forward(ctypes.byref(plan.desc), p(b.bg), p(c(g["xyz"])), p(c(shs)), p(None), p(c(g["opacity"])), p(c(g["scaling"])),
        p(c(g["rotation"])), p(None), p(c(b.world_view).reshape(NV, 16)), p(c(b.full_proj).reshape(NV, 16)), p(c(b.camera_center).reshape(NV, 3)),
        p(color), p(None), p(radii), p(geom), p(binning), p(image), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    npx = NV * H * W
    al = lambda n: ((n + 255) // 256) * 256
    T = ((H + 15) // 16) * ((W + 15) // 16)
    off = al(npx * 4) * 2
    tl = image[off: off + NV * T * 4].view(torch.int32).bitwise_and(0x7fffffff).float()   # bit 31: the tile ran the plain loop variant
    lim = image[al(npx * 4):][: npx * 4].view(torch.int32)
    sat = (lim != -1).float().mean().item()
    print(cfgname, "tile_last mean %.1f  median %.1f  frac>64 %.3f  max %d ; pixels saturated %.3f ; visible/view %.1f" % (tl.mean().item(), tl.median().item(), (tl > 64).float().mean().item(), int(tl.max().item()), sat, (radii > 0).float().sum(1).mean().item()))
    if cfgname == "C2":
        m = tl.reshape(NV, H // 16, W // 16).mean(0)
        print("mean tile_last per tile position (rows):")
        for r in m.tolist(): print(" ".join("%4.1f" % x for x in r))
        print("per-view mean: min %.1f max %.1f" % (tl.reshape(NV, -1).mean(1).min().item(), tl.reshape(NV, -1).mean(1).max().item()))
    unsat = (lim == -1).reshape(NV, H, W)
    tiles_unsat = unsat.reshape(NV, H // 16, 16, W // 16, 16).any(dim=4).any(dim=2)
    print(cfgname, "unsaturated pixels %d of %d ; tiles with an unsaturated pixel %d of %d" % (int(unsat.sum()), unsat.numel(), int(tiles_unsat.sum()), tiles_unsat.numel()))
