"""Can the fused render-loss step (forward + loss.backward()) be captured in a HIP graph (torch.cuda.CUDAGraph) and replayed?
Compares replayed loss / gradient with the eager step on fresh inputs and times both for a host-bound shape (C1) and C2."""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
from unipre3d_amd import synthetic
from unipre3d_amd.fused import render_loss_fused

dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["C1", "C2"]:
    cfg = synthetic.CONFIGS[name]
    B, P, V, H, W, level = cfg["B"], cfg["P"], cfg["V"], cfg["H"], cfg["W"], cfg["level"]
    b = synthetic.make_batch(B, P, V, H, W, level=level, seed=1).to(dev)
    b2 = synthetic.make_batch(B, P, V, H, W, level=level, seed=2).to(dev)
    kind = "focal_l2" if level == "object" else "l2"
    static_h = b.raw.permute(0, 2, 1).contiguous().requires_grad_(True)

    def step():
        static_h.grad = None
        loss, _, _ = render_loss_fused(static_h, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, H, W, level=level,
                                       offset_scale=b.offset_scale, loss_kind=kind, return_images=False)
        loss.backward()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    static_h.grad = None
    with torch.cuda.graph(g):
        static_loss, _, _ = render_loss_fused(static_h, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, H, W, level=level,
                                              offset_scale=b.offset_scale, loss_kind=kind, return_images=False)
        static_loss.backward()
    # replay on NEW head output (copied into the static input)
    with torch.no_grad():
        static_h.copy_(b2.raw.permute(0, 2, 1))
    g.replay()
    torch.cuda.synchronize()
    l_graph, g_graph = static_loss.detach().clone(), static_h.grad.detach().clone()
    h2 = b2.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    l2, _, _ = render_loss_fused(h2, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, H, W, level=level,
                                 offset_scale=b.offset_scale, loss_kind=kind, return_images=False)
    l2.backward()
    torch.cuda.synchronize()
    print(name, "graph == eager:", bool(torch.equal(l_graph, l2.detach())), bool(torch.equal(g_graph, h2.grad)))
    for label, fn in (("eager", step), ("graph replay", g.replay)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        print(f"  {label}: {1e3 * (time.perf_counter() - t0) / 200:.4f} ms per step")
