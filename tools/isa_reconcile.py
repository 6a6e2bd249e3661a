"""profiles/r06/isa_histogram_render_fb.json: the tile kernel's loop bodies by opcode (tools/isa_histogram.py on the product build's ISA), their
dynamic weight from the measured walk statistics, the reconciliation with the SQ instruction counters of the same launch, and the issue
floor the measured per-opcode costs give (tools/ub/opcost.hip).  usage: isa_reconcile.py <fb1.json> <profiles/r06 dir>"""
import json
import os
import sys

hist, prof = json.load(open(sys.argv[1])), sys.argv[2]
sq = json.load(open(os.path.join(prof, "sq_issue_C2.json")))
line = json.load(open(os.path.join(prof, "bench_full_C2.json")))
costs = [json.loads(l) for l in open(os.path.join(prof, "opcode_issue_costs.jsonl")) if l.startswith("{")]
SW = json.load(open(os.path.join(prof, "sq_wait_C2.json"))).get("derived", {})
cost = {c["op"]: c["cycles_per_slot_at_2_per_fma"] for c in costs}

loops = hist["loops"]
nd = lambda lp, k: lp["valu_subclasses"].get(k, 0)
# backward PLAIN body: the loop with the 26 DPP adds of the reduction and the fewest transcendentals (2 exp2 of alpha_run + 4 rcp); its
# inner "skip" loop (entries that contribute nothing to the tile) is the one it contains with LDS reads and no DPP
bwd = min((lp for lp in loops if nd(lp, "dpp") == 26), key=lambda lp: (nd(lp, "trans"), lp["instructions"]))
# forward PLAIN body: no DPP, one v_min3 (the hoisted saturation test), 2 transcendentals, the smallest such loop with LDS reads
fwd = min((lp for lp in loops if nd(lp, "dpp") == 0 and "v_min3_f32" in lp["opcodes"] and nd(lp, "trans") == 2 and lp["classes"].get("LDS", 0) >= 3),
          key=lambda lp: lp["instructions"])


def fast_path(lp):
    """The loop's common path: its span minus the blocks that only repair masks (>= 4 selects, no transcendental) -- in the forward body the
    per-pixel stop masks and the saturation bookkeeping, taken on the last few entries of a walk only."""
    keep = [b for b in lp["block_list"] if not (b["valu_subclasses"].get("cndmask", 0) >= 4 and b["valu_subclasses"].get("trans", 0) == 0)]
    cls, sub = {}, {}
    for b in keep:
        for k, v in b["classes"].items(): cls[k] = cls.get(k, 0) + v
        for k, v in b["valu_subclasses"].items(): sub[k] = sub.get(k, 0) + v
    return {"head": lp["head"], "tail_block": lp["tail_block"], "instructions": sum(b["instructions"] for b in keep), "classes": cls, "valu_subclasses": sub,
            "blocks_kept": [b["label"] for b in keep], "blocks_on_the_rare_path": [b["label"] for b in lp["block_list"] if b not in keep],
            "opcodes_of_the_whole_span": lp["opcodes"]}


def price(lp):
    """cycles of one trip through the body at the measured per-opcode issue costs (VALU only + what a scalar instruction costs the SIMD)"""
    c_fma = cost["v_fma_f32"]
    c = {"fma_class": (cost["v_fmac_f32_e32"] + cost["v_mul_f32_e32"] + cost["v_sub_f32_e32"] + c_fma) / 4, "mov": cost["v_mov_b32_e32"],
         "trans": (cost["v_exp_f32_e32"] + cost["v_rcp_f32_e32"]) / 2, "cmp": (cost["v_cmp_ge_f32_e32 -> vcc"] + cost["v_cmp_gt_f32_e64 -> sgpr pair"]) / 2,
         "cndmask": cost["v_cndmask_b32_e64 (sgpr pair)"], "dpp": cost["v_add_f32_dpp quad_perm"], "minmax": cost["v_min_f32_e32"],
         "lane": cost["v_readfirstlane_b32"], "other": cost["v_cmp_lt_u32_e64 -> sgpr pair"]}
    valu = sum(n * c[k] for k, n in lp["valu_subclasses"].items())
    salu = lp["classes"].get("SALU", 0) * max(cost["v_fma_f32 + s_and_b64"] - c_fma, 0.0)
    return {"valu_cycles": valu, "salu_cycles": salu, "class_costs_used": c}


cfg = line["config"]
tiles = 128 * 256
walked = cfg["list_consumption"]["sorted_positions_walked_per_tile_mean"]
trips = tiles * walked
per = sq["per_launch"]
fwd_span = fwd
fwd = fast_path(fwd)
pf, pb = price(fwd), price(bwd)
valu_static = fwd["classes"]["VALU"] + bwd["classes"]["VALU"]
salu_static = fwd["classes"].get("SALU", 0) + bwd["classes"].get("SALU", 0)
cycles = per["GRBM_GUI_ACTIVE"] / 8.0
body_cycles = (pf["valu_cycles"] + pf["salu_cycles"] + pb["valu_cycles"] + pb["salu_cycles"]) * trips / 1024.0
overhead_valu = per["SQ_INSTS_VALU"] - valu_static * trips
out = {
    "kernel": "render_fb_wave_kernel<1> (C2: 128 views x 256 tiles, P = 128)",
    "how": "tools/isa_render_fb.sh: hipcc -S with the product's flags -> tools/isa_histogram.py (basic blocks, back edges) -> this reconciliation",
    "forward_PLAIN_body": fwd,
    "backward_PLAIN_body": {k: bwd[k] for k in ("head", "tail_block", "instructions", "classes", "valu_subclasses", "opcodes")},
    "trip_counts": {"tiles_per_launch": tiles, "sorted_positions_walked_per_tile_mean": walked, "body_trips_per_launch": trips,
                    "note": "a tile walks the view's sorted list up to its last contributing position in BOTH directions; with the reference's activations "
                            "nearly every walked entry hits the tile (R ~ P T), so body trips ~ tiles x walked positions"},
    "reconciliation": {
        "VALU_static_per_trip_fwd_plus_bwd": valu_static, "VALU_predicted_loop_bodies": valu_static * trips, "SQ_INSTS_VALU_measured": per["SQ_INSTS_VALU"],
        "VALU_outside_the_two_bodies": overhead_valu, "VALU_outside_per_tile": overhead_valu / tiles,
        "SALU_static_per_trip": salu_static, "SALU_predicted_loop_bodies": salu_static * trips, "SQ_INSTS_SALU_measured": per["SQ_INSTS_SALU"],
        "reading": "the two bodies account for the stated share of the measured instruction counts; the rest is per tile: prologue, staging (tile_stage, "
                   "once per 64 sorted entries and direction), loss / seed, the partial-row copy-out, the saturation bookkeeping on the last entries of a walk"},
    "opcode_costs": {"source": "profiles/r06/opcode_issue_costs.jsonl (tools/ub/opcost.hip on the same box; cycles = 2 x time / time of a v_fma_f32 stream)", "table": cost},
    "issue_floor": {
        "shader_cycles_per_launch_measured": cycles, "kernel_us_rocprof": 1e3 * sq["kernel_avg_ms"],
        "guide_rate_every_VALU_2_cycles": {"cycles": per["SQ_INSTS_VALU"] * 2.0 / 1024.0, "frac": per["SQ_INSTS_VALU"] * 2.0 / 1024.0 / cycles},
        "guide_rate_transcendentals_quarter": {"cycles": ((per["SQ_INSTS_VALU"] - per["SQ_INSTS_VALU_TRANS_F32"]) * 2.0 + per["SQ_INSTS_VALU_TRANS_F32"] * 8.0) / 1024.0,
                                                "frac": ((per["SQ_INSTS_VALU"] - per["SQ_INSTS_VALU_TRANS_F32"]) * 2.0 + per["SQ_INSTS_VALU_TRANS_F32"] * 8.0) / 1024.0 / cycles},
        "measured_opcode_costs_loop_bodies_only": {"cycles": body_cycles, "frac": body_cycles / cycles, "fwd_body": pf, "bwd_body": pb},
        "measured_opcode_costs_whole_kernel": {
            "cycles": body_cycles + overhead_valu * 2.4 / 1024.0,
            "frac": (body_cycles + overhead_valu * 2.4 / 1024.0) / cycles,
            "note": "loop bodies at their opcode costs + the per-tile remainder at the bodies' mean VALU cost (2.4 cycles)"},
        "reading": ("no build of THIS instruction mix can run below the opcode-cost floor.  What separates the kernel from it: the {slots} wave slots (7 per SIMD) are occupied "
                    "{occ:.2f} of the launch (SQ_WAVE_CYCLES x 4 / (cycles x slots), profiles/r06/sq_wait_C2.json) -- dispatching the tiles heaviest-first moves the kernel "
                    "by -2 %, lightest-first by +3 %, a persistent-wave form with a software dequeue by +30 % (EXPERIMENTS.md, round 6) -- and inside a resident wave "
                    "{wi:.0f} % of the time is spent waiting for an issue slot or a dependency, {ww:.0f} % in s_waitcnt on LDS round trips").format(
                        slots=SW.get("wave_slots", 7168), occ=SW.get("wave_slot_occupancy", float("nan")), wi=100 * SW.get("share_of_wave_time_waiting_for_issue", float("nan")),
                        ww=100 * SW.get("share_of_wave_time_in_s_waitcnt", float("nan")))
    },
}
print(json.dumps(out, indent=1))
