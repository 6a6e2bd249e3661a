"""Helper of tests/test_gpu_more_parity.py::test_general_loop_variant_...: runs the fused step (single- and two-pass) and the batched operator
on a fixed PLAIN-qualifying batch with whatever library build U3D_LIB_DIRNAME selects and writes the results to an .npz."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unipre3d_amd import fused, head, synthetic  # noqa: E402
from unipre3d_amd.rasterizer import rasterize_gaussians_batched  # noqa: E402

out_path = sys.argv[1]
dev = torch.device("cuda:0")
B, P, V, H, W = 3, 128, 2, 96, 80
b = synthetic.make_batch(B, P, V, H, W, level="object", seed=23).to(dev)
res = {}
for name, sp in (("single", True), ("two", False)):
    h = b.raw.permute(0, 2, 1).contiguous().requires_grad_(True)
    loss, img, radii = fused.render_loss_fused(h, b.center, b.world_view, b.full_proj, b.camera_center, b.gt, b.bg, b.fov_deg, H, W, level="object",
                                               offset_scale=b.offset_scale, loss_kind="focal_l2", single_pass=sp)
    loss.backward()
    res[name + "_loss"], res[name + "_img"], res[name + "_grad"], res[name + "_radii"] = loss.item(), img.detach().cpu().numpy(), h.grad.cpu().numpy(), radii.cpu().numpy()
g = {k: v.detach().clone().requires_grad_(True) for k, v in synthetic.gaussians_from_batch(b).items()}
shs = head.concat_sh(g["features_dc"], g["features_rest"])
t = math.tan(b.fov_deg * math.pi / 360)
color, radii, _ = rasterize_gaussians_batched(g["xyz"], g["opacity"], b.world_view, b.full_proj, b.camera_center, b.bg, H, W, t, t, shs=shs,
                                              scales=g["scaling"], rotations=g["rotation"], sh_degree=1)
cot = torch.randn(color.shape, generator=torch.Generator().manual_seed(4)).to(dev)
(color * cot).sum().backward()
torch.cuda.synchronize()
res["op_img"] = color.detach().cpu().numpy()
for k in ("xyz", "opacity", "scaling", "rotation"):
    res["op_d" + k] = g[k].grad.cpu().numpy()
np.savez(out_path, **res)
