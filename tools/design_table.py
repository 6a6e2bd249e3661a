"""Prints the §5 table of DESIGN.md from the committed files of a profiles/<round>/ directory (so that the document's numbers are the files' numbers).
usage: python tools/design_table.py profiles/r06"""
import csv
import json
import os
import re
import sys

P = sys.argv[1] if len(sys.argv) > 1 else "profiles/r06"
names = {"C2": "**C2** (128 views, P = 128, 256²)", "C2_compact": "C2 `--compact`", "C3": "C3 (64 views, P = 2048, 256²)", "C4": "C4 (16 views, P = 40 000, 480×640)",
         "C5": "C5 (8 views, P = 200 000, 480×640)"}
print("| config | ms/step (repeat min / median) | views/s | tile kernel | other kernels of the step (µs) | walked per tile (mean / max) | §8(d) `frac` | `frac_consumed` | VALU at 2 cycles | wave slots occupied | PMC HBM per launch (tile + reduce) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for cfg in ("C2", "C2_compact", "C3", "C4", "C5"):
    d = json.load(open(os.path.join(P, f"bench_full_{cfg}.json")))
    ks = {}
    for r in csv.DictReader(open(os.path.join(P, f"kernel_stats_bench_{cfg}.csv"))):
        m = re.search(r"(\w+_kernel)", r["Name"])
        if m:
            ks[m.group(1)] = float(r["AverageNs"]) / 1e3
    tile = ks.get("render_fb_wave_kernel")
    own = ("preprocess_fwd_kernel", "quat_norms_kernel", "depth_sort_block_radix_kernel", "msd_hist_kernel", "msd_scatter_kernel", "bucket_sort_kernel", "bwd_reduce1_kernel",
           "bwd_reduce_kernel", "preprocess_bwd_kernel", "quat_fixup_kernel")
    others = " + ".join(f"{ks[k]:.1f}" for k in own if k in ks)
    rep, rf, lc = d["repeatability"], d["roofline"], d["config"]["list_consumption"]
    sw = json.load(open(os.path.join(P, f"sq_wait_{cfg}.json"))).get("derived", {})
    ri = d.get("roofline_issue") or {}
    pt = json.load(open(os.path.join(P, f"pmc_traffic_{cfg}.json")))["per_launch"]
    tr = pt.get("render_fb_wave_kernel", {}).get("hbm_bytes_corrected", 0) / 1e6
    rd = sum(pt[k]["hbm_bytes_corrected"] for k in ("bwd_reduce1_kernel", "bwd_reduce_kernel") if k in pt) / 1e6
    frac = rf["frac"]
    print(f"| {names[cfg]} | {d['ms_per_step']:.4f} ({rep['min']:.4f} / {rep['median']:.4f}) | {d['value'] / 1e3:.0f} k | {tile:.1f} µs | {others} | "
          f"{lc['sorted_positions_walked_per_tile_mean']:.1f} / {lc['sorted_positions_walked_per_tile_max']:.0f} | {frac:.2f}{' (†)' if frac > 1 else ''} | {rf.get('frac_consumed', 0):.2f} | "
          f"{ri.get('frac_valu_2cyc', float('nan')):.2f} | {sw.get('wave_slot_occupancy', float('nan')):.2f} | {tr:.0f} + {rd:.0f} MB |")
dd = os.path.join(P, "bench_full_default_run.json")
if os.path.exists(dd):
    d = json.load(open(dd))
    for k, v in (d.get("other_configs") or {}).items():
        if k in ("C4_fused", "C5_fused") and "error" not in v:
            print(f"| {k} | {v['ms_per_step']:.3f} ({v['repeat_min_ms']:.3f}) | {v['views_s'] / 1e3:.1f} k | {1e3 * v['tile_kernel_ms']:.1f} µs incl. reduce (HIP events) | — | {v['walked_mean']:.1f} / {v['walked_max']:.0f} | — | "
                  f"{v['frac_consumed']:.2f} | — | — | — |")
    print("\nother_configs of the default run (ms/step):", {k: round(v["ms_per_step"], 4) for k, v in (d.get("other_configs") or {}).items() if "error" not in v})
    pv = d["per_view_dropin"]
    print("per_view:", {k: round(pv[k], 3) for k in ("ms_per_step", "noop_operator_ms", "op_by_op_wrapper_body_ms", "graph_replay_ms", "operator_host_ms", "operator_gpu_ms") if k in pv})
    fr = d["forward_rasterizer"]
    print("forward_rasterizer:", round(1e3 * fr["avg_ms"], 1), "us", round(fr["frac_of_8TBs"], 3), round(fr["frac_pmc_bytes"], 3), "| no invdepth:", round(1e3 * fr["without_inverse_depth"]["avg_ms"], 1),
          round(fr["without_inverse_depth"]["frac_of_8TBs"], 3), round(fr["without_inverse_depth"]["frac_pmc_bytes"], 3))
    print("headline:", round(d["value"]), "views/s", round(d["ms_per_step"], 4), "ms; cpu", round(d["cpu_baseline"]["value"], 1), "on", d["cpu_baseline"]["cores"], "threads; e2e", round(d["train_step_e2e_standin"]["ms_per_step"], 2),
          "ms; with head", round(d["train_step_with_head"]["ms_per_step"], 3), "; backward_unit", round(d["hot_step_backward_unit"]["ms_per_step"], 4), "vs", round(d["hot_step_backward_unit"]["plain_loss_backward_same_moment_ms_per_step"], 4))
    print("gradclip:", {k: round(v, 3) for k, v in d["train_step_e2e_standin"]["gradclip_n4b"].items() if isinstance(v, float)})
    print("issue:", {k: round(v, 3) for k, v in d["roofline_issue"].items() if isinstance(v, float) and k.startswith("frac")})
