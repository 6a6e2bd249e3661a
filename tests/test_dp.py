"""DP runtime on CPU: world_size-2 gloo processes.  DDP-averaged gradients of the render-loss step must
equal single-process gradients on the concatenated batch (the self-check SURVEY 8c names for R9)."""
import os

import torch

import dp_worker
from unipre3d_amd import dp, step


def test_ddp_gradients_equal_single_process_on_concatenated_batch(tmp_path):
    dp.launch(dp_worker.ddp_worker, 2, cfg=(str(tmp_path),), backend="gloo")
    r0, r1 = torch.load(os.path.join(tmp_path, "rank0.pt")), torch.load(os.path.join(tmp_path, "rank1.pt"))
    batch, feats, model = dp_worker.make_inputs()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    loss = step.train_step(model, feats, batch, opt, dp_worker.H, dp_worker.W, 0, "focal_l2",
                           render_fn=dp_worker.cpu_render_views, clip_grad=None)
    for g0, g1, p in zip(r0["grads"], r1["grads"], model.parameters()):
        assert torch.allclose(g0, g1, rtol=0, atol=0)                       # all-reduced: identical on both ranks
        assert torch.allclose(g0, p.grad, rtol=1e-4, atol=1e-7)             # == gradient of the global-batch loss
    assert any(p.grad.abs().sum() > 0 for p in model.parameters())
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - loss) < 1e-6
    assert torch.allclose(r0["mean_loss"], r1["mean_loss"]) and abs(r0["mean_loss"] - loss) < 1e-6


def test_shard_sampler_partitions_the_epoch(tmp_path):
    dp.launch(dp_worker.sampler_worker, 2, cfg=(str(tmp_path),), backend="gloo")
    a, b = torch.load(os.path.join(tmp_path, "sampler0.pt")), torch.load(os.path.join(tmp_path, "sampler1.pt"))
    assert len(a) == len(b) == 5 and sorted(a + b) == list(range(10))
    ref = torch.utils.data.distributed.DistributedSampler(list(range(10)), num_replicas=2, rank=0, shuffle=True, seed=3)
    ref.set_epoch(4)
    assert a == list(iter(ref))                                               # same permutation rule as the reference's sampler


def test_single_process_helpers():
    assert dp.get_world_size() == 1 and dp.get_rank() == 0 and dp.is_main_process()
    dp.synchronize()
    m = torch.nn.Linear(2, 2)
    assert dp.create_ddp_model(m) is m
    assert list(dp.shard_range(32, 3, 8)) == [12, 13, 14, 15] and list(dp.shard_range(5, 1, 2)) == [2, 3]
    x = torch.tensor([3.0])
    assert dp.all_reduce_mean(x).item() == 3.0
    h = dp.GaussianHead()
    assert h(torch.zeros(2, 128, 384)).shape == (2, 23, 128)
    assert sum(p.numel() for p in h.parameters()) == 384 * 128 + 128 + 128 * 23 + 23   # 52,247
