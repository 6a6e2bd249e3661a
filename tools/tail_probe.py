"""How much does the last partial round of waves cost the tile kernel?  (evidence for the decision not to build a half-tile
tail variant.)  The fused step at the C5 shape (P = 200 000, 480 x 640 -> 1200 tiles per view) with 4 .. 10 views: 8192 waves are resident at once
(256 CUs x 32), so 6 views (7200 tiles) fit in one round, 7 views (8400) spill 208 tiles, 8 views (9600, the C5 config) spill 1408.
Prints the HIP-event time of the render_fb scope (tile kernel + its reduce) and the time per 1000 tiles."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", __file__.rsplit("/tools", 1)[0]))
from unipre3d_amd import _lib, synthetic
from unipre3d_amd.fused import render_loss_fused

dev = torch.device("cuda:0")
P, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 200000), 480, 640
for V in (4, 5, 6, 7, 8, 9, 10, 12):
    b = synthetic.make_batch(1, P, V, H, W, level="scene", seed=42).to(dev)
    h = b.raw.permute(0, 2, 1).contiguous().requires_grad_(True)

    def step():
        h.grad = None
        l, _, _ = render_loss_fused(h, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, H, W, level="scene",
                                    offset_scale=b.offset_scale, loss_kind="l2", return_images=False)
        l.backward()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    _lib.profile_begin(2048, ("render_fb",))
    for _ in range(60):
        step()
    torch.cuda.synchronize()
    ms, n = _lib.profile_end()["render_fb"]
    tiles = V * 1200
    print(f"views {V:2d} tiles {tiles:6d} rounds {tiles / 8192:.2f}  render_fb scope {1e3 * ms / n:7.1f} us   per 1000 tiles {1e6 * ms / n / tiles:6.2f} us")
