"""The unmodified reference loop (one operator call per object and view, train_network.py:418-446) through the drop-in module at
C2, and the same loop with a NO-OP operator (allocates its outputs, launches nothing): what the wrapper around the operator costs
by itself -- render_predicted's zeros_like / cat / radii>0, the per-item slicing, torch.stack, the loss, and autograd's backward
through all of them (~14 small kernels per view).  Usage: python tools/per_view_route.py [steps]"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from unipre3d_amd import head, losses, rasterizer, renderer, synthetic  # noqa: E402

dev = torch.device("cuda")
cfgc = synthetic.CONFIGS["C2"]; B, P, V, H, W = cfgc["B"], cfgc["P"], cfgc["V"], cfgc["H"], cfgc["W"]
b = synthetic.make_batch(B, P, V, H, W, seed=42).to(dev)
cfg = types.SimpleNamespace(data=types.SimpleNamespace(fov=b.fov_deg, training_resolution=H), model=types.SimpleNamespace(max_sh_degree=1))
raw = b.raw.clone().requires_grad_(True)


def per_view_step():
    raw.grad = None
    gs = head.process_object_output(raw, b.center, b.offset_scale)
    imgs = []
    for i in range(B):
        pc = {k: v[i].contiguous() for k, v in gs.items()}
        for v in range(V):
            imgs.append(renderer.render_predicted(pc, b.world_view[i, v], b.full_proj[i, v], b.camera_center[i, v], b.bg, cfg)["render"])
    rendered = torch.stack(imgs)
    loss = losses.focal_l2_loss(rendered, b.gt.reshape(B * V, 3, H, W), b.bg, 4.0, 1.0)
    loss.backward()
    return loss


class _NullOp(torch.autograd.Function):
    """Same inputs / outputs as the operator, no kernel: outputs and gradients are uninitialised allocations."""

    @staticmethod
    def forward(ctx, *tensors):
        ctx.shapes = [t.shape for t in tensors]
        ctx.dev = tensors[0].device
        return torch.empty(3, H, W, device=ctx.dev), torch.empty(P, dtype=torch.int32, device=ctx.dev)

    @staticmethod
    def backward(ctx, g, _):
        return tuple(torch.empty(s, device=ctx.dev) for s in ctx.shapes)


def timeit(name, fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    ms = sum(ts) / n
    print("%-58s mean %.2f  min %.2f  median %.2f ms/step  %.0f views/s  %.0f us per forward+backward pair (min)"
          % (name, ms, ts[0], ts[n // 2], B * V / ts[0] * 1e3, 1e3 * ts[0] / (B * V)))


n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
timeit("per-view operator loop (reference call pattern)", per_view_step, n)
prev = rasterizer.set_operator_override(lambda *tensors: _NullOp.apply(*tensors) + (None,))
timeit("same loop, NO-OP operator (the wrapper's own cost)", per_view_step, n)
rasterizer.set_operator_override(prev)
renderer.FAST_PATH = False
timeit("per-view loop, op-by-op wrapper body (round 3)", per_view_step, n)
renderer.FAST_PATH = True
