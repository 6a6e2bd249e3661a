"""How many (tile, Gaussian) pairs whose 3-sigma rectangle covers the tile contribute NOTHING to it (alpha < 1/255 at every pixel)?
That is the head-room of an exact tile-level cull in tile_stage.  Dense torch evaluation on the GPU, one C2 / C3-like view at a time."""
import math, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from unipre3d_amd import synthetic, head
from unipre3d_amd.rasterizer import rasterize_gaussians_batched
dev = torch.device("cuda")
for name in (sys.argv[1:] or ["C2", "C3"]):
    c = synthetic.CONFIGS[name]; B, P, V, H, W, level = 2, c["P"], 2, c["H"], c["W"], c["level"]
    b = synthetic.make_batch(B, P, V, H, W, level=level, seed=42).to(dev)
    g = synthetic.gaussians_from_batch(b)
    t = math.tan(b.fov_deg * math.pi / 360)
    shs = head.concat_sh(g["features_dc"], g["features_rest"])
    tot = hit_pairs = dead_pairs = 0
    for bi in range(B):
        for v in range(V):
            xyz, op, sc, rot = g["xyz"][bi], g["opacity"][bi, :, 0], g["scaling"][bi], g["rotation"][bi]
            Vm, Pm = b.world_view[bi, v], b.full_proj[bi, v]
            ph = torch.cat([xyz, torch.ones(P, 1, device=dev)], 1)
            pv = ph @ Vm; hom = ph @ Pm
            pw = 1.0 / (hom[:, 3] + 1e-7)
            px = ((hom[:, 0] * pw + 1) * W - 1) * 0.5; py = ((hom[:, 1] * pw + 1) * H - 1) * 0.5
            r, x, y, z = rot.unbind(1)
            R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                             2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
            M = R * sc[:, None, :]; Sig = M @ M.transpose(1, 2)
            fx, fy = W / (2 * t), H / (2 * t)
            tz = pv[:, 2]; lim = 1.3 * t
            tx = (pv[:, 0] / tz).clamp(-lim, lim) * tz; ty = (pv[:, 1] / tz).clamp(-lim, lim) * tz
            J = torch.zeros(P, 2, 3, device=dev); J[:, 0, 0] = fx / tz; J[:, 0, 2] = -fx * tx / tz ** 2; J[:, 1, 1] = fy / tz; J[:, 1, 2] = -fy * ty / tz ** 2
            Wm = Vm[:3, :3].T
            T = J @ Wm; cov = T @ Sig @ T.transpose(1, 2)
            a0, b0, c0 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
            det0 = a0 * c0 - b0 * b0; a, cc = a0 + 0.3, c0 + 0.3; det = a * cc - b0 * b0
            aa = (det0 / det).clamp(min=2.5e-5).sqrt()
            A, Bc, C = cc / det, -b0 / det, a / det
            mid = 0.5 * (a + cc); rad = torch.ceil(3 * torch.sqrt(mid + torch.sqrt((mid * mid - det).clamp(min=0.1))))
            vis = (tz > 0.2)
            x0 = ((px - rad) / 16).int().clamp(0, W // 16); x1 = ((px + rad + 15) / 16).int().clamp(0, W // 16)
            y0 = ((py - rad) / 16).int().clamp(0, H // 16); y1 = ((py + rad + 15) / 16).int().clamp(0, H // 16)
            ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
            for i0 in range(0, P, 64):
                sl = slice(i0, min(P, i0 + 64))
                dx = px[sl, None, None] - xs[None]; dy = py[sl, None, None] - ys[None]
                power = -0.5 * (A[sl, None, None] * dx * dx + C[sl, None, None] * dy * dy) - Bc[sl, None, None] * dx * dy
                alpha = (op[sl] * aa[sl])[:, None, None] * torch.exp(power)
                ok = (alpha >= 1 / 255) & (power <= 0)
                tile_any = ok.reshape(-1, H // 16, 16, W // 16, 16).any(4).any(2)          # (n, ty, tx)
                tyi = torch.arange(H // 16, device=dev)[None, :, None]; txi = torch.arange(W // 16, device=dev)[None, None, :]
                cover = (txi >= x0[sl, None, None]) & (txi < x1[sl, None, None]) & (tyi >= y0[sl, None, None]) & (tyi < y1[sl, None, None]) & vis[sl, None, None]
                hit_pairs += int(cover.sum()); dead_pairs += int((cover & ~tile_any).sum())
    print(f"{name}: rectangle-covered (tile, Gaussian) pairs {hit_pairs}, of which contribute nothing to the tile {dead_pairs} = {100.0 * dead_pairs / max(hit_pairs, 1):.1f} %")
