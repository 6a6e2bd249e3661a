"""RCCL itself (R9): needs at least two MI355X in the box -- skipped on the 1-GPU test boxes, runs wherever the suite meets a
multi-GPU node.  DDP-averaged gradients of the head + render-loss step over RCCL must equal the single-process gradients on the
concatenated batch (the gloo version of this check runs on CPU in tests/test_dp.py)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rccl_code_paths_on_one_gpu():
    """World of one over backend "nccl": the collective library comes up and every RCCL branch of dp.py executes on a single
    MI355X (communicator, device barrier, DDP buckets + SyncBN around the fused step); only the inter-GPU transport is left to a
    multi-GPU node.  Own process: the process group must not leak into the other tests."""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_world1.py")
    res = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["backend"] == "nccl"


def _worker(out_dir):
    from unipre3d_amd import dp, step, synthetic
    rank, world = dp.get_rank(), dp.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    assert world == 2 and torch.distributed.get_backend() == "nccl" and dev.index == rank
    batch = synthetic.make_batch(4, 128, 2, 64, 64, seed=5)
    feats = torch.randn(4, 128, 384, generator=torch.Generator().manual_seed(6))
    torch.manual_seed(7)
    model = dp.GaussianHead().to(dev)
    r = dp.shard_range(4)
    sl = synthetic.SyntheticBatch(**{k: (v[r.start:r.stop] if torch.is_tensor(v) and v.dim() > 1 else v) for k, v in batch.__dict__.items()}).to(dev)
    model = dp.create_ddp_model(model)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    loss = step.train_step(model, feats[r.start:r.stop].to(dev), sl, opt, 64, 64, 0, "focal_l2", clip_grad=None, fused=True)
    torch.save({"grads": [p.grad.cpu() for p in model.module.parameters()], "loss": loss.cpu()}, os.path.join(out_dir, f"rccl{rank}.pt"))
    dp.synchronize()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: two MI355X")
def test_rccl_ddp_gradients_equal_single_process(tmp_path):
    from unipre3d_amd import dp, step, synthetic
    dp.launch(_worker, 2, cfg=(str(tmp_path),), backend="nccl")
    r0, r1 = torch.load(os.path.join(tmp_path, "rccl0.pt")), torch.load(os.path.join(tmp_path, "rccl1.pt"))
    dev = torch.device("cuda:0")
    batch = synthetic.make_batch(4, 128, 2, 64, 64, seed=5).to(dev)
    feats = torch.randn(4, 128, 384, generator=torch.Generator().manual_seed(6)).to(dev)
    torch.manual_seed(7)
    model = dp.GaussianHead().to(dev)
    loss = step.train_step(model, feats, batch, torch.optim.SGD(model.parameters(), lr=0.0), 64, 64, 0, "focal_l2", clip_grad=None, fused=True)
    for g0, g1, p in zip(r0["grads"], r1["grads"], model.parameters()):
        assert torch.equal(g0, g1)                                            # all-reduced: identical on both ranks
        assert torch.allclose(g0, p.grad.cpu(), rtol=1e-4, atol=1e-7)          # == gradient of the global-batch loss
    assert abs(0.5 * (r0["loss"].item() + r1["loss"].item()) - loss.item()) < 1e-6
