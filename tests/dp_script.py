"""Script-form rank for tests/test_dp.py::test_launch_script_*: joins the job from the environment dp.launch_script prepared."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from unipre3d_amd import dp  # noqa: E402

out_dir, mode = sys.argv[1], sys.argv[2]
rank, local_rank, world = dp.init_from_env(os.environ.get("U3D_BENCH_BACKEND"))
if mode == "fail" and rank == 1:
    sys.exit(7)
dp.host_barrier()
mx = dp.host_all_reduce_max(10.0 + rank)
x = torch.tensor([float(rank + 1)])
dist.all_reduce(x)
with open(os.path.join(out_dir, f"script_rank{rank}.json"), "w") as f:
    json.dump({"rank": rank, "local_rank": local_rank, "world": world, "max": mx, "sum": x.item(),
               "port": os.environ["MASTER_PORT"], "addr": os.environ["MASTER_ADDR"]}, f)
if rank == 0:
    print("rank0-stdout")
dp.host_barrier()
dp.shutdown()
