import sys, time, torch, ctypes
sys.path.insert(0, '/root/repo')
from unipre3d_amd import synthetic, _lib, fused
from unipre3d_amd.rasterizer import _Plan, _stream_ptr, _f32c
dev = torch.device("cuda")
b1 = synthetic.make_batch(2, 16, 1, 16, 16, seed=1).to(dev)
h1 = torch.randn(2, 16, 23, device=dev).requires_grad_(True)
N = 2000
def T(f, n=N):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
def full():
    h1.grad = None
    loss, _, _ = fused.render_loss_fused(h1, b1.center, b1.world_view, b1.full_proj, b1.camera_center, b1.gt, b1.bg, b1.fov_deg, 16, 16,
                                         level="object", offset_scale=b1.offset_scale, loss_kind="focal_l2", single_pass=True, return_images=False)
    fused.backward_unit(loss)
def fwd_only():
    loss, _, _ = fused.render_loss_fused(h1, b1.center, b1.world_view, b1.full_proj, b1.camera_center, b1.gt, b1.bg, b1.fov_deg, 16, 16,
                                         level="object", offset_scale=b1.offset_scale, loss_kind="focal_l2", single_pass=True, return_images=False)
print("full step (fwd+bwd)      %.1f us" % T(full))
print("forward call only        %.1f us" % T(fwd_only))
# pieces
plan = _Plan(2, 1, 16, 16, 16, 0.45, 0.45, 1.0, 1, 4, 2)
print("_Plan lookup             %.1f us" % T(lambda: _Plan(2, 1, 16, 16, 16, 0.45, 0.45, 1.0, 1, 4, 2)))
print("7x torch.empty           %.1f us" % T(lambda: [torch.empty(1000, dtype=torch.uint8, device=dev) for _ in range(7)]))
print("_stream_ptr              %.1f us" % T(_stream_ptr))
print("6x _f32c + reshape       %.1f us" % T(lambda: [_f32c(x, dev) for x in (h1, b1.center, b1.world_view, b1.full_proj, b1.camera_center, b1.gt)]))
lib = _lib.load()
hd = _lib.HeadDesc(1, 23, 1.0); ld = _lib.LossDesc(2, 4.0, 1.0)
NV = 2; P = 16
color = None; radii = torch.empty((NV, P), dtype=torch.int32, device=dev); loss = torch.empty((), device=dev); d_head = torch.empty_like(h1)
u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
geom, binning, fu, scratch = u8(plan.sizes.geom_bytes), u8(plan.sizes.binning_bytes), u8(plan.sizes.fused_bytes), u8(plan.sizes.backward_bytes)
p = _lib.ptr
wv, fp, cc, gt = b1.world_view.reshape(NV, 16), b1.full_proj.reshape(NV, 16), b1.camera_center.reshape(NV, 3), b1.gt.reshape(NV, 3, 16, 16)
hh = h1.detach()
def ccall():
    lib.u3d_render_loss_step(ctypes.byref(plan.desc), ctypes.byref(hd), ctypes.byref(ld), p(b1.bg), p(hh), p(b1.center), p(wv), p(fp), p(cc), p(gt), p(color), p(radii),
                             p(loss), p(d_head), p(geom), p(binning), p(fu), p(scratch), _stream_ptr())
print("ctypes u3d_render_loss_step (5 launches) %.1f us" % T(ccall))
print("16x _lib.ptr             %.1f us" % T(lambda: [p(x) for x in (b1.bg, hh, b1.center, wv, fp, cc, gt, color, radii, loss, d_head, geom, binning, fu, scratch, hh)]))
