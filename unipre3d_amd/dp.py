"""Data-parallel runtime of the render-loss path: one process per MI355X, RCCL over xGMI.

Counterpart of pointcept/engines/launch.py:36-157 (launcher + init_process_group + local group + barrier),
pointcept/engines/defaults.py:22-43 (`create_ddp_model`), pointcept/utils/comm.py:23-88 (rank helpers, barrier)
and the sharding arithmetic of train_network.py:55-71.  Design differences (MI355X-first, not a translation):

  * the process model is "one rank per GPU, started by torch.distributed.run (or `launch` below)"; ranks read
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Backend "nccl" IS RCCL on ROCm; "gloo" on CPU.
    A world of one does NOT create a process group (the reference does, launch.py:89-97) -- there is nothing to
    exchange and RCCL initialisation costs seconds.
  * the render path needs no collective: renders are independent per (object, view).  The single exchange per step
    is DDP's bucketed gradient all-reduce of the PyTorch modules feeding the rasterizer; rasterizer gradients are
    never communicated.  Bucket cap defaults to 32 MiB: four buckets for the 117.9 MB transformer config, so the all-reduce of the
    late layers' gradients overlaps the backward of the early ones (a single 128 MiB bucket could only start when the whole backward
    is done), while each ring collective is still large enough to run at link bandwidth (xGMI is point-to-point, 7 links x ~153 GB/s:
    message size, not message count, is what a ring needs -- PyTorch's 25 MiB default is in the same range); gradient_as_bucket_view
    avoids a copy per step.
  * the object-level reference path does not shard its batch across ranks (train_network.py:114-128 builds loaders
    with the global batch and no sampler); `shard_range` shards both levels.
"""
from __future__ import annotations

import os
import socket
from datetime import timedelta
from typing import Callable, Optional, Tuple

# The host driver only supports dmabuf IPC: RCCL (and sharing HIP tensors across processes) needs this BEFORE the HIP/HSA
# runtime initialises, i.e. before the first torch.cuda call of the process -- so it is set at import, not in init_from_env.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

DEFAULT_TIMEOUT = timedelta(minutes=60)  # pointcept/engines/launch.py:21
_LOCAL_RANK = 0
_CTL = None   # gloo control group beside an RCCL default group (host-side barriers / scalar reductions)


# ---- comm helpers (pointcept/utils/comm.py:23-88) -------------------------------------------
def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_local_rank() -> int:
    return _LOCAL_RANK


def is_main_process() -> bool:
    return get_rank() == 0


def synchronize() -> None:
    """Barrier among all ranks (no-op without a process group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_backend() == dist.Backend.NCCL:
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def host_barrier() -> None:
    """Barrier over the gloo control group (host side): rank rendezvous that does not touch RCCL -- timing brackets and the
    launcher use it so that a broken collective library cannot take the collective-free render path down with it."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    dist.barrier(group=_CTL) if _CTL is not None else synchronize()


def host_all_reduce_max(value: float) -> float:
    """MAX over ranks of a host scalar (the 'max over ranks' of a timed region) through the gloo control group."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if _CTL is None:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == dist.Backend.NCCL else torch.device("cpu")
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_CTL)
    return float(t.item())


def all_reduce_mean(x: torch.Tensor) -> torch.Tensor:
    """SUM then divide by world: the validation-PSNR reduction of train_network.py:253-256."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        x = x / get_world_size()
    return x


# ---- process group -------------------------------------------------------------------------
def init_from_env(backend: Optional[str] = None, timeout: timedelta = DEFAULT_TIMEOUT, force_group: bool = False) -> Tuple[int, int, int]:
    """Join the job described by the torch.distributed.run environment.  Returns (rank, local_rank, world).
    force_group: create the process group even for a world of one (the reference always does, launch.py:89-97).  With backend
    "nccl" on one GPU this runs every RCCL code path of this module -- communicator, device barrier, DDP buckets, SyncBN -- except
    the inter-GPU transport (tests/test_gpu_rccl.py); MASTER_ADDR / MASTER_PORT must then be set like for any job."""
    global _LOCAL_RANK, _CTL
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    _LOCAL_RANK = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(_LOCAL_RANK % max(torch.cuda.device_count(), 1))
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            raise RuntimeError("WORLD_SIZE > 1 needs MASTER_PORT (torch.distributed.run and dp.launch both set it): there is no "
                               "safe default port for several jobs on one node")
        dist.init_process_group(backend=backend or ("nccl" if use_cuda else "gloo"), init_method="env://",
                                world_size=world, rank=rank, timeout=timeout)
        if dist.get_backend() == dist.Backend.NCCL:
            # RCCL communicators are created lazily by the first device collective (DDP's parameter broadcast); rendezvous
            # and host-side scalars go through a gloo group so that they never depend on it
            _CTL = dist.new_group(backend="gloo", timeout=timeout)
        host_barrier()
    return rank, _LOCAL_RANK, world


def _find_free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(local_rank: int, main_func: Callable, world: int, port: int, backend: Optional[str], args: tuple) -> None:
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    init_from_env(backend)
    try:
        main_func(*args)
    finally:
        shutdown()


def shutdown() -> None:
    """Leave the job: destroy the process group(s) of this rank (no-op for a world of one)."""
    global _CTL
    if dist.is_available() and dist.is_initialized():
        _CTL = None
        dist.destroy_process_group()


def launch(main_func: Callable, num_gpus_per_machine: int, cfg: tuple = (), backend: Optional[str] = None) -> None:
    """Single-node launcher with the reference's call shape `launch(main_worker, n, cfg=(cfg,))`
    (train_network.py:572-579 -> pointcept/engines/launch.py:36): n == 1 runs in-process, n > 1 spawns one
    process per GPU with a localhost rendezvous on a free port."""
    if num_gpus_per_machine <= 1:
        init_from_env(backend)
        main_func(*cfg)
        return
    mp.spawn(_worker, nprocs=num_gpus_per_machine, args=(main_func, num_gpus_per_machine, _find_free_port(), backend, cfg),
             daemon=False)


def launch_script(script: str, argv, nproc: int, backend: Optional[str] = None, share_devices: bool = False,
                  env_extra: Optional[dict] = None) -> int:
    """Start `nproc` ranks of a SCRIPT on this node (one per GPU) and wait for them: what `python -m torch.distributed.run
    --nproc-per-node N script` does, for entry points that take `--gpus N` themselves (bench.py) -- the script form of
    `launch` above / of the reference's launcher (pointcept/engines/launch.py:75-88).  Each rank is a fresh interpreter with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT=<free port> in its environment (read by
    `init_from_env`); rank 0 keeps the caller's stdout, the others' is dropped; the first failure ends the job.
    share_devices: allow more ranks than HIP devices (test hook; RCCL refuses two ranks on one device, so the backend
    defaults to gloo then).  Returns the job's exit code."""
    import subprocess
    import sys
    if nproc > 1 and torch.cuda.is_available() and torch.cuda.device_count() < nproc and not share_devices:
        print(f"[dp.launch_script] {nproc} ranks requested but only {torch.cuda.device_count()} HIP device(s) are visible", file=sys.stderr)
        return 2
    port = _find_free_port()
    procs = []
    for r in range(nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend or share_devices:
            env["U3D_BENCH_BACKEND"] = backend or "gloo"
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        import time
        live = list(procs)
        while live:
            for pr in list(live):
                code = pr.poll()
                if code is not None:
                    live.remove(pr)
                    if code != 0:
                        rc = rc or code
                        for other in live:          # one rank died: the others would wait in a collective forever
                            other.kill()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


# ---- model wrap ----------------------------------------------------------------------------
def create_ddp_model(model: nn.Module, *, sync_bn: bool = True, bucket_cap_mb: int = 32, force_ddp: bool = False, **kwargs) -> nn.Module:
    """SyncBN conversion + DistributedDataParallel as ModelManager.setup_distributed does
    (train_network.py:180-186 -> pointcept/engines/defaults.py:22-43; the reference's kwargs broadcast_buffers=False and
    find_unused_parameters=True are the defaults here too: its predictor has branches that do not run in every step, and DDP
    hangs on a parameter that never produces a gradient unless it is told to look for them).  Pass
    find_unused_parameters=False for a module known to use every parameter (saves DDP's per-step graph walk)."""
    if get_world_size() == 1 and not (force_ddp and dist.is_available() and dist.is_initialized()):
        return model        # (force_ddp: wrap even a world of one -- exercises the collective library on a single GPU)
    if sync_bn:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    on_gpu = next(model.parameters()).is_cuda
    if on_gpu and "device_ids" not in kwargs:
        kwargs["device_ids"] = [torch.cuda.current_device()]
    kwargs.setdefault("broadcast_buffers", False)
    kwargs.setdefault("find_unused_parameters", True)
    kwargs.setdefault("gradient_as_bucket_view", True)
    return nn.parallel.DistributedDataParallel(model, bucket_cap_mb=bucket_cap_mb, **kwargs)


def shard_range(global_batch: int, rank: Optional[int] = None, world: Optional[int] = None) -> range:
    """Objects [lo, hi) of a global batch owned by `rank`: bs_per_gpu = batch // world (train_network.py:66-71);
    the remainder (global_batch % world) is dropped like a DistributedSampler with drop_last."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    per = global_batch // world
    return range(rank * per, (rank + 1) * per)


class EpochShardSampler(torch.utils.data.Sampler):
    """DistributedSampler semantics used at train_network.py:55-64,326-327: a seeded permutation per epoch
    (set_epoch(iteration)), padded to a multiple of world, strided by rank."""

    def __init__(self, n: int, rank: Optional[int] = None, world: Optional[int] = None, shuffle: bool = True, seed: int = 0):
        self.n, self.shuffle, self.seed, self.epoch = n, shuffle, seed, 0
        self.rank = get_rank() if rank is None else rank
        self.world = get_world_size() if world is None else world
        self.num_samples = -(-n // self.world)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        total = self.num_samples * self.world
        idx += idx[: total - len(idx)]
        return iter(idx[self.rank:total:self.world])

    def __len__(self):
        return self.num_samples


class GaussianHead(nn.Module):
    """The per-point MLP that produces the rasterizer's inputs: `final` of model/point_predictor.py:78-85
    (transformer: 384 -> 128 -> 23) followed by the permute of :100 -> (B, 23, N).  It is the trainable module
    directly upstream of the hot path; its gradients (plus the backbone's) are what DDP all-reduces (R9)."""

    def __init__(self, in_dim: int = 384, hidden: int = 128, out_dim: int = 23):
        super().__init__()
        self.final = nn.Sequential(nn.Linear(in_dim, hidden), nn.ReLU(), nn.Linear(hidden, out_dim))

    def forward(self, feats: torch.Tensor, point_major: bool = False) -> torch.Tensor:
        out = self.final(feats)                      # (B, N, 23) contiguous
        return out if point_major else out.permute(0, 2, 1)
