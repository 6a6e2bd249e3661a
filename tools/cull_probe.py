"""What would lane compaction of the culled (view, Gaussian) pairs buy preprocess_fwd at scene level?  The kernel's heavy block (EWA, conic, radius,
SH colour) runs for a wave whenever ANY lane is in front of the camera; with random scenes half the lanes are.  Times the kernel (HIP events) on the
C5 shape with 4 identical views per slice and the Gaussians resampled to be all visible / half visible (interleaved) / all culled."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", __file__.rsplit("/tools", 1)[0]))
from unipre3d_amd import _lib, synthetic
from unipre3d_amd.fused import render_loss_fused

dev = torch.device("cuda:0")
P, V, H, W = 200000, 8, 480, 640
b = synthetic.make_batch(1, P, V, H, W, level="scene", seed=42).to(dev)
for name in ("world_view", "full_proj", "camera_center"):
    t = getattr(b, name)
    t[:] = t[:, 0:1]                                     # the same camera 8 times
g = synthetic.gaussians_from_batch(b)
xyz = g["xyz"][0]
zv = xyz @ b.world_view[0, 0][:3, 2] + b.world_view[0, 0][3, 2]
front, back = torch.nonzero(zv > 0.2)[:, 0], torch.nonzero(zv <= 0.2)[:, 0]
print(f"in front of the camera: {front.numel()} of {P}")


def run(tag, idx):
    raw = b.raw[:, :, idx].contiguous(); cen = b.center[:, idx].contiguous()
    h = raw.permute(0, 2, 1).contiguous().requires_grad_(True)

    def step():
        h.grad = None
        l, _, _ = render_loss_fused(h, cen, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, H, W, level="scene",
                                    offset_scale=b.offset_scale, loss_kind="l2", return_images=False)
        l.backward()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    _lib.profile_begin(4096, ("preprocess_fwd", "depth_sort", "preprocess_bwd", "render_fb"))
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    pr = _lib.profile_end()
    print(f"{tag:28s}", {k: round(1e3 * ms / max(n, 1), 1) for k, (ms, n) in pr.items() if n})


rep = lambda ix: ix[torch.arange(P, device=dev) % ix.numel()]
run("all visible", rep(front))
mix = torch.stack([rep(front)[: P // 2], rep(back)[: P // 2]], 1).reshape(-1)
run("half visible, interleaved", mix)
run("half visible, front first", torch.cat([rep(front)[: P // 2], rep(back)[: P // 2]]))
run("all culled", rep(back))
