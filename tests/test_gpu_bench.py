"""bench.py as the driver runs it: the contract's JSON line, and the `--gpus N` self-launch (two ranks sharing cuda:0 over gloo --
RCCL refuses two ranks on one device, so the collective library itself can only run on a multi-GPU node)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline")


LINE_MAX = 6000          # the driver parses the LAST stdout line; round 5's 29 KB object came back `parsed: null`


def _check_compact(stdout):
    """The LAST stdout line is the contract object: compact JSON, <= 6 KB, with roofline / cpu_baseline / config.workload."""
    last = stdout.rstrip("\n").splitlines()[-1]
    assert len(last) < LINE_MAX, len(last)
    line = json.loads(last)
    for k in CONTRACT:
        assert k in line, k
    assert line["roofline"]["frac"] > 0 and line["config"]["workload"] and "extra_keys" in line
    assert [l for l in stdout.splitlines() if l.startswith("{")] == [last], "exactly ONE JSON line on stdout"
    return line


def _run(args, env=None, timeout=900, compact=False):
    """Runs bench.py; returns the FULL record (bench_full.json, where the per-operator detail lives) after checking the compact
    line the driver parses -- whose contract keys must equal the record's."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        full = os.path.join(td, "bench_full.json")
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                             env=dict(os.environ, U3D_BENCH_FULL_JSON=full, **(env or {})))
        assert res.returncode == 0, res.stderr[-3000:]
        line = _check_compact(res.stdout)
        out = json.load(open(full))
    assert line["full_record"] == full
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "data"):
        assert line[k] == out[k], k
    assert abs(line["value"] - out["value"]) <= 1e-5 * out["value"] and abs(line["roofline"]["frac"] - out["roofline"]["frac"]) <= 1e-4 * out["roofline"]["frac"]
    if "cpu_baseline" in out:
        assert line["cpu_baseline"]["kind"] == out["cpu_baseline"]["kind"] and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    for k in ("scale_ok", "scale_verdict", "rccl_ranks"):
        assert line[k] == out[k], k
    return (out, line) if compact else out


def test_bench_line_single_gpu_small_config():
    out = _run(["--config", "C1", "--steps", "5", "--warmup", "2", "--cpu-seconds", "1", "--no-e2e"])
    for k in CONTRACT + ("cpu_baseline",):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["scaling"] == "weak" and out["dtype"] == "f32"
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    assert abs(out["value"] - 8 * 5 / (out["ms_per_step"] * 5e-3)) < 1e-6 * out["value"]
    assert out["config"]["autograd_entry"] == "loss.backward()"            # the headline is quoted on the standard entry
    assert "frac_consumed" in r and 0 < r["frac_consumed"] <= r["frac"] and r["consumed_bytes_per_launch"] <= r["algorithmic_bytes_per_launch"]
    assert out["config"]["list_consumption"]["instances_consumed_per_view"] <= out["config"]["num_rendered_per_view"]
    # SCALE keys exist at every N; --no-e2e leaves the region with the exchange out, and the line says so
    assert out["scale_ok"] is False and "skipped" in out["scale_note"] and out["rccl_ranks"] == 0


def test_bench_self_launches_two_ranks():
    out, line = _run(["--gpus", "2", "--config", "C1", "--steps", "4", "--warmup", "1", "--cpu-seconds", "1"],
                     env={"U3D_BENCH_SHARE_GPU": "1", "U3D_BENCH_EXTRAS_BUDGET_S": "240"}, compact=True)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["views_per_step"] == 16
    # an N > 1 line is field for field as complete as the N = 1 line: roofline AND cpu_baseline (rank 0 times it
    # while the other ranks wait at the host barrier), plus the rows N1 / N4a
    for k in CONTRACT + ("cpu_baseline", "pointops", "fusion"):
        assert k in out, k
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0 and out["roofline"]["frac"] > 0
    assert "error" not in out["pointops"] and out["pointops"]["all_equal_oracle"] and out["fusion"]["forward"]["equals_oracle"]
    assert abs(out["value"] - 16 * 4 / (out["ms_per_step"] * 4e-3)) < 1e-6 * out["value"]
    e2e = out["train_step_e2e_standin"]
    assert "error" not in e2e, e2e
    assert e2e["collective_backend"] == "gloo" and e2e["gradient_bytes_all_reduced_per_step"] == 4 * 29_464_215
    assert e2e["n1_same_region"]["ms_per_step"] > 0 and e2e["value"] > 0
    assert "error" not in out["train_step_with_head"], out["train_step_with_head"]
    # SCALE-proofing: the figures of the step that contains the exchange sit at the top level of the line, and scale_ok is False
    # here because the two ranks share one device over gloo (rccl_ranks 0 != n_gpus 2)
    for k in ("rccl_ranks", "collective_backend", "gradient_bytes_all_reduced_per_step", "train_region_value", "n1_same_region",
              "speedup_over_n1_same_region", "scale_ok"):
        assert k in out, k
    assert out["rccl_ranks"] == 0 and out["collective_backend"] == "gloo" and out["scale_ok"] is False
    assert "gloo" in out["scale_note"] and "rccl_ranks = 0" in out["scale_note"]          # ... and the line says why
    assert line["scale_ok"] is False and line["scale_verdict"].startswith("fail: rccl_ranks 0 != n_gpus 2") and "gloo" in line["scale_verdict"]
    for k in ("rccl_ranks", "train_region_value", "n1_same_region", "speedup_over_n1_same_region", "scale_verdict"):
        assert k in line, k                                                               # ... at the top level of the COMPACT line
    assert line["extra_keys"]["pointops_all_equal_oracle"] is True and line["extra_keys"]["fusion"]["forward"]["equals_oracle"] is True
    assert out["train_region_value"] == e2e["value"] and out["n1_same_region"] == e2e["n1_same_region"]["value"]
    assert out["gradient_bytes_all_reduced_per_step"] == 4 * 29_464_215 and out["speedup_over_n1_same_region"] > 0
    # the same launch quoting the step WITH the exchange in it as `value`
    out2 = _run(["--gpus", "2", "--config", "C1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--value-region", "train"],
                env={"U3D_BENCH_SHARE_GPU": "1", "U3D_BENCH_EXTRAS_BUDGET_S": "240"})
    e = out2["train_step_e2e_standin"]
    assert out2["value"] == e["value"] and out2["ms_per_step"] == e["ms_per_step"] and out2["hot_path"]["value"] > out2["value"]
    assert out2["config"]["workload"].startswith("END-TO-END")


def test_bench_under_torch_distributed_run():
    """The driver's multi-GPU form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- ranks come from the environment.  Two ranks share cuda:0 over gloo here (RCCL needs
    one device per rank)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--no-cpu-baseline", "--no-e2e", "--no-next-rows"]
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        full = os.path.join(td, "bench_full.json")
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, U3D_BENCH_BACKEND="gloo", U3D_BENCH_FULL_JSON=full))
        assert res.returncode == 0, res.stderr[-3000:]
        line = _check_compact(res.stdout)
        out = json.load(open(full))
    assert line["n_gpus"] == 2 and line["extra_keys"]["other_configs"]["C3"][0] > 0
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and "error" not in out["train_step_with_head"]
    # the default config at N > 1 carries the hot-only step of the config BASELINE assigns to that GPU count (here C3), every rank on its
    # own per-GPU batch, max-over-ranks time
    c3 = out["other_configs"]["C3"]
    assert "error" not in c3, c3
    assert c3["n_gpus"] == 2 and c3["ms_per_step"] > 0 and abs(c3["views_s"] - 2 * 64 / (c3["ms_per_step"] * 1e-3)) < 1e-6 * c3["views_s"]
    assert set(out["other_configs"]) == {"C3"}


def test_default_line_carries_the_other_configs_and_the_noop_control():
    """The driver's own command (`python bench.py --gpus 1`, here with fewer steps): after the contractual C2 region the line
    holds the hot-only steps of C3 / C4 / C5 / C2-compact (`other_configs`) and the per-view route's NO-OP-operator control."""
    out, line = _run(["--steps", "10", "--warmup", "3", "--cpu-seconds", "1", "--no-e2e"], compact=True)
    ex = line["extra_keys"]
    assert "extra_keys_dropped" not in line, line.get("extra_keys_dropped")
    assert set(ex["other_configs"]) == {"C3", "C4", "C5", "C2_compact", "C4_fused", "C5_fused"} and len(ex["pointops_us"]) >= 14
    assert ex["per_view"]["operator_gpu_ms"] > 0 and ex["per_view"]["operator_host_ms"] > 0
    assert line["roofline"]["kernel"] == "render_fb" and line["roofline"]["traffic"] and line["roofline"]["frac_consumed"] > 0
    assert line["roofline"]["traffic_source"].startswith("profiles/r06/")                   # this round's committed PMC pass
    # the tile kernel's issue floors, none fitted to it: every VALU at the guide's 2 cycles < transcendentals at quarter rate < measured opcode costs <= 1
    iss = ex["issue"]
    assert 0 < iss["frac_valu_2cyc"] < iss["frac_valu_trans_quarter"] < iss["frac_opcode_costs"] <= 1.0 and 0.5 < iss["wave_slot_occupancy"] <= 1.0
    assert line["scale_verdict"].startswith("n/a")
    oc = out["other_configs"]
    for k in ("C3", "C4", "C5", "C2_compact", "C4_fused", "C5_fused"):
        assert "error" not in oc[k], oc[k]
        for f in ("ms_per_step", "views_s", "tile_kernel_ms", "frac_consumed", "walked_mean"):
            assert oc[k][f] > 0, (k, f)
        assert oc[k]["tile_kernel_ms"] < oc[k]["ms_per_step"] and 0 < oc[k]["frac_consumed"] < 1
    pv = out["per_view_dropin"]
    assert "error" not in pv, pv
    assert pv["noop_operator_ms"] > 0 and abs(pv["operator_share_ms"] - (pv["ms_per_step"] - pv["noop_operator_ms"])) < 1e-9
    # the operator's own share, measured directly: its kernels' HIP-event time per step and the host time of its 2 x B*V bare binding calls
    assert "operator_error" not in pv, pv.get("operator_error")
    assert 0 < pv["operator_gpu_ms"] < pv["ms_per_step"] and pv["operator_host_ms"] > 0 and pv["operator_gpu_ms"] < pv["graph_replay_ms"]
    fr = out["forward_rasterizer"]
    assert 0 < fr["frac_pmc_bytes"] <= fr["frac_of_8TBs"]
    # rows N1 / N4a in the driver's own line: per operator microseconds, roofline, CPU-oracle baseline, equality with the oracle
    po, fu = out["pointops"], out["fusion"]
    assert "error" not in po and "error" not in fu, (po, fu)
    assert po["all_equal_oracle"] and len(po["ops"]) >= 14
    for name, o in po["ops"].items():
        assert o["us"] > 0 and o["roofline"]["frac"] > 0 and o["cpu_baseline"]["value"] > 0 and o["cpu_baseline"]["kind"] == "port", name
    assert po["ops"]["fps_1024_to_128"]["us_per_selection"] < 0.6                          # (0.84 before round 5; 0.44 measured)
    for half in ("forward", "backward"):
        assert fu[half]["us"] > 0 and fu[half]["equals_oracle"] and fu[half]["cpu_baseline"]["value"] > 0
