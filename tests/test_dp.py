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


def test_gradclip_library_exports_every_declared_symbol():
    """include/unipre3d_gradclip.h vs libunipre3d_gradclip.so (no compute without a GPU: argument checks only)."""
    import ctypes, os, re
    from unipre3d_amd import gradcheck
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "unipre3d_gradclip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(u3d_[a-z_0-9]+)\s*\(", hdr)))
    assert set(names) == set(gradcheck.EXPORTS)
    lib = gradcheck.load()
    null = ctypes.c_void_p(0)
    assert lib.u3d_gradclip_stats(null, null, null, 0, 0, null, null) == 0          # nothing to do
    assert lib.u3d_gradclip_stats(null, null, null, 3, 5, null, null) == 1          # missing tables
    assert lib.u3d_gradclip_stats(null, null, null, -1, 0, null, null) == 1
    assert lib.u3d_gradclip_finalize(null, 0, ctypes.c_float(1.0), null, null) == 1  # no state block
    assert lib.u3d_gradclip_scale(null, null, null, 2, 2, null, null) == 1
    assert int(re.search(r"#define U3D_GC_CHUNK (\d+)", open(os.path.join(root, "include", "unipre3d_gradclip.h")).read()).group(1)) == gradcheck.GC_CHUNK


def test_check_and_clip_gradients_matches_reference_semantics():
    """train_network.py:368-390: False on any NaN/Inf (grads untouched), else clip_grad_norm_(max_norm=1.0)."""
    from unipre3d_amd.gradcheck import check_and_clip_gradients
    torch.manual_seed(0)
    m1, m2 = torch.nn.Linear(7, 5), torch.nn.Linear(7, 5)
    m2.load_state_dict(m1.state_dict())
    x = torch.randn(9, 7)
    for m in (m1, m2):
        (m(x) ** 2).sum().mul(10).backward()
    assert check_and_clip_gradients(m1.parameters(), 1.0)
    has_invalid = any(torch.isnan(p.grad).any() or torch.isinf(p.grad).any() for p in m2.parameters())
    assert not has_invalid
    torch.nn.utils.clip_grad_norm_(m2.parameters(), max_norm=1.0)
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-8)
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m1.parameters()))
    assert abs(total.item() - 1.0) < 1e-4                                # it was above the threshold and got clipped
    # small gradients are left alone
    m1.zero_grad(); (m1(x).sum() * 1e-4).backward(); g0 = [p.grad.clone() for p in m1.parameters()]
    assert check_and_clip_gradients(m1.parameters(), 1.0) and all(torch.equal(a, p.grad) for a, p in zip(g0, m1.parameters()))
    # any non-finite entry -> False and the step is to be skipped
    for bad in (float("nan"), float("inf"), -float("inf")):
        m1.weight.grad[2, 3] = bad
        assert not check_and_clip_gradients(m1.parameters(), 1.0)
        m1.weight.grad[2, 3] = 0.0
    assert check_and_clip_gradients([torch.nn.Parameter(torch.zeros(3))], 1.0)   # no .grad at all
    # finite gradients whose fp32 sum of squares overflows (norm ~ 4e20): the reference's isnan/isinf scan passes them and
    # clip_grad_norm_ rescales -- so must this (the max-abs decides finiteness, not the overflowed L2 norm)
    big = torch.nn.Parameter(torch.zeros(64))
    big.grad = torch.full((64,), 5e19)
    assert check_and_clip_gradients([big], 1.0)
    assert torch.isfinite(big.grad).all() and abs(big.grad.double().norm().item() - 1.0) < 1e-4


def test_standin_predictor_has_the_reference_parameter_counts():
    """encoder 29,066,880 + final 52,247 + fusion Linear 295,296 + image 1x1 conv/GN 49,792 = 117.9 MB fp32."""
    from unipre3d_amd.standin import PointTransformerStandIn, object_intrinsics
    m = PointTransformerStandIn()
    assert m.encoder_parameters() == 29_066_880
    assert sum(p.numel() for p in m.final.parameters()) == 52_247
    assert sum(p.numel() for p in m.fusion_mlps.parameters()) == 295_296
    assert sum(p.numel() for p in m.image_conv.parameters()) == 49_792
    total = sum(p.numel() for p in m.parameters())
    assert total == 29_464_215 and abs(total * 4 / 1e6 - 117.9) < 0.1
    k = object_intrinsics(49.13434264120263, 128)
    assert abs(k[0, 0] - 140.0) < 1e-3 and k[0, 2] == 64.0                 # fx = fy = 140, cx = cy = 64


def test_launch_script_starts_one_rank_per_process_and_joins_them(tmp_path):
    """The `--gpus N` self-launch of bench.py: N fresh interpreters with the torch.distributed.run environment, localhost
    rendezvous on a free port, gloo world of 2 on CPU; host-side barrier / MAX helpers; a failing rank ends the job."""
    import json
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_script.py")
    assert dp.launch_script(script, [str(tmp_path), "ok"], 2, backend="gloo") == 0
    r = [json.load(open(os.path.join(tmp_path, f"script_rank{i}.json"))) for i in range(2)]
    assert [x["rank"] for x in r] == [0, 1] and all(x["world"] == 2 and x["addr"] == "127.0.0.1" for x in r)
    assert r[0]["port"] == r[1]["port"] and all(x["max"] == 11.0 and x["sum"] == 3.0 for x in r)
    assert dp.launch_script(script, [str(tmp_path), "fail"], 2, backend="gloo") == 7
