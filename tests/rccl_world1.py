"""One rank, backend "nccl" (= RCCL): every collective code path of unipre3d_amd/dp.py on a single MI355X -- communicator creation,
device barrier, host-side gloo control group, DDP (bucketed all-reduce, SyncBN) around the fused render-loss step -- everything but the
inter-GPU transport.  Run by tests/test_gpu_rccl.py in its own process."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from unipre3d_amd import dp, step, synthetic  # noqa: E402

os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(dp._find_free_port()))
rank, local_rank, world = dp.init_from_env("nccl", force_group=True)
assert dist.get_backend() == "nccl" and world == 1 and dp._CTL is not None
dev = torch.device("cuda", torch.cuda.current_device())
batch = synthetic.make_batch(4, 128, 2, 64, 64, seed=5).to(dev)
feats = torch.randn(4, 128, 384, generator=torch.Generator().manual_seed(6)).to(dev)


class Net(torch.nn.Module):                     # head with a BatchNorm in front so that the SyncBN conversion has something to convert
    def __init__(self):
        super().__init__()
        self.bn = torch.nn.BatchNorm1d(384)
        self.head = dp.GaussianHead()

    def forward(self, f, point_major=False):
        return self.head(self.bn(f.transpose(1, 2)).transpose(1, 2), point_major=point_major)


def run(wrap):
    torch.manual_seed(7)
    net = Net().to(dev)
    model = dp.create_ddp_model(net, force_ddp=True) if wrap else net
    if wrap:
        assert isinstance(model, torch.nn.parallel.DistributedDataParallel)
        assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in model.modules())
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    loss = step.train_step(model, feats, batch, opt, 64, 64, 0, "focal_l2", clip_grad=None, fused=True)
    return loss.item(), [p.grad.clone() for p in net.parameters()]


l_ddp, g_ddp = run(True)
l_ref, g_ref = run(False)
dp.synchronize()                                 # RCCL barrier (device_ids)
dp.host_barrier()                                # gloo control group
m = dp.all_reduce_mean(torch.tensor([3.0], device=dev))
mx = dp.host_all_reduce_max(1.5)
ok = abs(l_ddp - l_ref) < 1e-6 and all(torch.allclose(a, b, rtol=1e-4, atol=1e-7) for a, b in zip(g_ddp, g_ref)) and m.item() == 3.0 and mx == 1.5
dp.shutdown()
print(json.dumps({"ok": bool(ok), "loss_ddp": l_ddp, "loss_ref": l_ref, "backend": "nccl"}))
sys.exit(0 if ok else 1)
