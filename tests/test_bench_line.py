"""The driver parses the LAST stdout line of bench.py; round 5's 29 KB object came back `parsed: null`.  `bench.compact_line` is pure Python, so the
contract on that line -- one JSON object, < 6 000 bytes, with the contract keys, `roofline` and `cpu_baseline` -- is held on the CPU as well, on the committed
full record of the driver-form run and on a record inflated far beyond anything the bench produces."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def _bench():
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_compact_line_of_the_committed_default_run():
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_full_default_run.json")))
    line = b.compact_line(full, os.path.join(ROOT, "bench_full.json"))
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < b.COMPACT_LINE_MAX <= 6000 and "\n" not in text
    for k in CONTRACT:
        assert k in line, k
    assert line["metric"] == "rendered_views_per_sec" and line["unit"] == "views/s" and line["dtype"] == "f32" and line["vs_baseline"] is None
    assert line["config"]["workload"].startswith("C2:") and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and r["traffic"] > 0 and r["kernel"] == "render_fb"
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / 1e9 / (r["avg_launch_us"] * 1e-6)) < 1e-3 * r["achieved"]     # bytes / live launch duration
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert line["scale_verdict"].startswith("n/a") and line["full_record"] == "bench_full.json" and "extra_keys_dropped" not in line
    # the committed compact line of the same run is this object (up to the path of the full record)
    committed = json.loads(open(os.path.join(ROOT, "profiles", "r06", "bench_line_default_run.json")).read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "roofline", "cpu_baseline", "config"):
        assert committed[k] == line[k], k


def test_compact_line_stays_under_the_cap_whatever_the_record_holds():
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_full_default_run.json")))
    ops = full["pointops"]["ops"]
    one = next(iter(ops.values()))
    for i in range(400):                                   # a record with hundreds of operators and configs, and prose in every field
        ops[f"made_up_operator_with_a_long_name_{i}"] = dict(one)
        full["other_configs"][f"made_up_config_{i}"] = dict(next(iter(full["other_configs"].values())))
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    full["scale_note"] = "z" * 5000
    line = b.compact_line(full, "/tmp/bench_full.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= b.COMPACT_LINE_MAX
    for k in CONTRACT:
        assert k in line, k
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line.get("extra_keys_dropped")


def test_scale_verdict_sentences():
    b = _bench()
    assert b._scale_verdict(1, {"value": 1.0}).startswith("n/a")
    assert b._scale_verdict(8, {}).startswith("fail: the region with the exchange did not complete")
    assert b._scale_verdict(2, {"value": 1.0, "rccl_ranks": 0, "collective_backend": "gloo"}).startswith("fail: rccl_ranks 0 != n_gpus 2")
    assert b._scale_verdict(8, {"value": 1.0, "rccl_ranks": 8, "speedup_over_n1_same_region": 7.41}) == "pass: 7.41 x over one rank's same region >= 0.75 x 8"
    assert b._scale_verdict(8, {"value": 1.0, "rccl_ranks": 8, "speedup_over_n1_same_region": 5.2}).startswith("fail: 5.20 x")
    assert b._scale_verdict(4, {"value": 1.0, "rccl_ranks": 4}).startswith("fail: no single-rank time")
