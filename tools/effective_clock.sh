#!/bin/bash
# Effective shader clock of the tile kernel (MI355X_MICROARCH.md "DVFS give-back": GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time OF THE SAME PASS).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06t; mkdir -p $O; export TMPDIR=/tmp
for c in ${1:-C2}; do
  rm -rf /tmp/clk_$c
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/clk_$c -- python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --hot-only > /tmp/clk_$c.log 2>&1
  python - $c <<'PY' | tee $O/effective_clock_$c.json
import csv, glob, json, sys
c = sys.argv[1]
cc = glob.glob(f"/tmp/clk_{c}/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(f"/tmp/clk_{c}/**/*kernel_trace.csv", recursive=True)[0]
wall = {}
for r in csv.DictReader(open(kt)):
    if "render_fb_wave_kernel" in r["Kernel_Name"]: wall[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
per = {}
for r in csv.DictReader(open(cc)):
    if "render_fb_wave_kernel" in r["Kernel_Name"]: per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
rows = [{"wall_us": round(wall[d], 2), "shader_cycles": round(v["GRBM_GUI_ACTIVE"] / 8), "clock_ghz": round(v["GRBM_GUI_ACTIVE"] / 8 / wall[d] * 1e-3, 3),
         "wave_slot_occupancy_7168": round(4 * v["SQ_WAVE_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 7168), 3), "valu": v.get("SQ_INSTS_VALU")} for d, v in per.items() if d in wall]
n = len(rows)
print(json.dumps({"config": c, "command": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace -- python bench.py --config %s --steps 6 --warmup 2 --no-cpu-baseline --hot-only" % c,
                  "kernel": "render_fb_wave_kernel", "launches": n, "mean_wall_us_in_this_profiled_pass": round(sum(r["wall_us"] for r in rows) / n, 2),
                  "mean_shader_cycles": round(sum(r["shader_cycles"] for r in rows) / n), "mean_clock_ghz": round(sum(r["clock_ghz"] for r in rows) / n, 3), "per_launch": rows[:12]}, indent=1))
PY
done
