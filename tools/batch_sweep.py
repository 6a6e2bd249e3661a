import sys, time, torch
sys.path.insert(0, '/root/repo')
from unipre3d_amd import synthetic, _lib
from unipre3d_amd.fused import render_loss_fused
dev = torch.device("cuda")
for B in (8, 16, 32, 64, 128):
    P, V, H, W = 128, 4, 256, 256
    batch = synthetic.make_batch(B, P, V, H, W, seed=42).to(dev)
    torch.manual_seed(0)
    head_out = batch.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    def f():
        head_out.grad = None
        loss, _, _ = render_loss_fused(head_out, batch.center, batch.world_view, batch.full_proj, batch.camera_center, batch.gt, batch.bg, batch.fov_deg, H, W,
                                       level="object", offset_scale=batch.offset_scale, loss_kind="focal_l2", single_pass=True, return_images=False)
        loss.backward()
    for _ in range(30): f()
    torch.cuda.synchronize()
    _lib.profile_begin(4096, ("render_fb",))
    for _ in range(30): f()
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    ms, n = prof["render_fb"]
    print("B=%d views=%d  render_fb %.1f us  per view %.3f us" % (B, B * V, 1e3 * ms / n, 1e3 * ms / n / (B * V)))
