"""Worker functions for the multi-process CPU tests of the DP runtime (spawned by tests/test_dp.py)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from unipre3d_amd import dp, head, step, synthetic  # noqa: E402

B_GLOBAL, P, V, H, W, FEAT = 4, 6, 2, 16, 16, 12


def cpu_render_views(g, world_view, full_proj, camera_center, bg, fov_deg, H, W, input_images=0, max_sh_degree=1,
                     scaling_modifier=1.0):
    """Differentiable stand-in with render_views' signature for CPU-only tests: the float64 autograd restatement
    (test infrastructure) rendered per (object, view)."""
    from oracle import torch_ref
    t = math.tan(fov_deg * math.pi / 360)
    out = []
    for b in range(g["xyz"].shape[0]):
        shs = head.concat_sh(g["features_dc"][b], g["features_rest"][b]).double()
        for v in range(input_images, world_view.shape[1]):
            col, _, _ = torch_ref.render(g["xyz"][b].double(), g["opacity"][b].double(), world_view[b, v].double(),
                                         full_proj[b, v].double(), camera_center[b, v].double(), bg.double(), H, W, t, t, shs=shs,
                                         scales=g["scaling"][b].double(), rotations=g["rotation"][b].double(), sh_degree=1)
            out.append(col.float())
    return torch.stack(out)


def make_inputs():
    batch = synthetic.make_batch(B_GLOBAL, P, V, H, W, level="object", seed=5)
    feats = torch.randn(B_GLOBAL, P, FEAT, generator=torch.Generator().manual_seed(6))
    torch.manual_seed(7)
    model = dp.GaussianHead(FEAT, 16)
    return batch, feats, model


def slice_batch(batch, r):
    return synthetic.SyntheticBatch(**{k: (v[r.start:r.stop] if torch.is_tensor(v) and v.dim() > 1 else v)
                                       for k, v in batch.__dict__.items()})


def ddp_worker(out_dir):
    rank, world = dp.get_rank(), dp.get_world_size()
    assert world == 2 and dp.get_local_rank() == rank
    batch, feats, model = make_inputs()
    r = dp.shard_range(B_GLOBAL)
    assert len(r) == B_GLOBAL // world
    model = dp.create_ddp_model(model)
    assert isinstance(model, torch.nn.parallel.DistributedDataParallel)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    loss = step.train_step(model, feats[r.start:r.stop], slice_batch(batch, r), opt, H, W, 0, "focal_l2",
                           render_fn=cpu_render_views, clip_grad=None)
    grads = [p.grad.clone() for p in model.module.parameters()]
    mean_loss = dp.all_reduce_mean(loss.clone())
    dp.synchronize()
    torch.save({"grads": grads, "loss": loss, "mean_loss": mean_loss}, os.path.join(out_dir, f"rank{rank}.pt"))


def sampler_worker(out_dir):
    s = dp.EpochShardSampler(10, shuffle=True, seed=3)
    s.set_epoch(4)
    torch.save(list(iter(s)), os.path.join(out_dir, f"sampler{dp.get_rank()}.pt"))
