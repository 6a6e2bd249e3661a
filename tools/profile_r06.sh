#!/bin/bash
# Regenerates the judged artifacts of one bench configuration: tools/profile_r06.sh <C2|C3|C4|C5> [sq]
#   gpurun_out/prof_r06/bench_line_<cfg>.json          python bench.py --config <cfg>: the compact contract line (stdout); bench_full_<cfg>.json: the full record
#   gpurun_out/prof_r06/kernel_stats_bench_<cfg>.csv   rocprofv3 --kernel-trace --stats of the hot-only run
#   gpurun_out/prof_r06/pmc_traffic_<cfg>.json         separate --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH doubled: gfx950 calibration)
#   gpurun_out/prof_r06/sq_issue_<cfg>.json            (with `sq`) SQ instruction counters of the tile kernel, two --pmc passes
# Copy what is to be judged into profiles/r06/.
# A third argument is passed on to bench.py (e.g. --compact) and names the files <cfg>_<tag>: tools/profile_r06.sh C2 sq --compact
CFGNAME=${1:-C2}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06; mkdir -p $O
EXTRA=$3; CFG=$CFGNAME; if [ -n "$EXTRA" ]; then CFG=${CFGNAME}_${EXTRA#--}; fi
if [ -n "$U3D_LIB_DIRNAME" ] && [ "$U3D_LIB_DIRNAME" != "lib" ]; then CFG=${CFG}_${U3D_LIB_DIRNAME}; fi
cd /tmp && export TMPDIR=/tmp
ARGS="--config $CFGNAME $EXTRA --steps 50 --warmup 10"
if [ "$CFG" = "C2" ]; then U3D_BENCH_FULL_JSON=$O/bench_full_$CFG.json python $R/bench.py $ARGS > $O/bench_line_$CFG.json 2> $O/bench_$CFG.err
else U3D_BENCH_FULL_JSON=$O/bench_full_$CFG.json python $R/bench.py $ARGS --hot-only --cpu-seconds 10 > $O/bench_line_$CFG.json 2> $O/bench_$CFG.err; fi
if [ -n "$ONLY_LINE" ]; then tail -c 300 $O/bench_line_$CFG.json; echo; exit 0; fi   # (second pass: lines regenerated once this round's PMC / SQ files are committed)
rm -rf /tmp/kt_$CFG; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$CFG -- python $R/bench.py $ARGS --no-cpu-baseline --hot-only > $O/kt_$CFG.log 2>&1
cp $(find /tmp/kt_$CFG -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_$CFG.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --config $CFGNAME $EXTRA --steps 10 --warmup 3 --no-cpu-baseline --hot-only > $O/pmc_${c}_$CFG.log 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) /tmp/cc_$c.csv
done
python - $CFG <<'PY' > $O/pmc_traffic_$CFG.json
import csv, json, re, collections, sys
per = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f"/tmp/cc_{c}.csv")):
        if r["Counter_Name"] != c: continue
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if not m: continue
        acc[m.group(1)] += float(r["Counter_Value"]); n[m.group(1)] += 1
    for k in acc:
        per[k][c + "_KB"] = acc[k] / n[k]; per[k]["launches_" + c] = n[k]
for k, v in per.items():
    v["hbm_bytes_corrected"] = (2 * v.get("FETCH_SIZE_KB", 0.0) + v.get("WRITE_SIZE_KB", 0.0)) * 1024
print(json.dumps({
    "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --config {sys.argv[1]} --steps 10 --warmup 3 --no-cpu-baseline --hot-only  (two separate passes)",
    "workload": f"{sys.argv[1]} single-pass fused path",
    "calibration": "FETCH_SIZE reads exactly 1/2 of the bytes on gfx950 (x2 applied), WRITE_SIZE exact (MI355X_MICROARCH.md; re-verified in round 1 on a 256 MiB copy)",
    "formula": "hbm_bytes_corrected = (2*FETCH_SIZE_KB + WRITE_SIZE_KB) * 1024",
    "per_launch": per}, indent=1))
PY
if [ "$CFG" = "C2" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcf_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcf_$c -- python $R/bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline --hot-only > $O/pmcf_${c}.log 2>&1
    cp $(find /tmp/pmcf_$c -name "*counter_collection.csv" | head -1) /tmp/ccf_$c.csv
  done
  python - <<'PY' > $O/pmc_traffic_fwd_C2.json
import csv, json, collections
per = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f"/tmp/ccf_{c}.csv")):
        if r["Counter_Name"] != c or "render_fwd_wave_kernel" not in r["Kernel_Name"]: continue
        k = "render_fwd_wave_kernel<true>" if "ILb1E" in r["Kernel_Name"] or "<true>" in r["Kernel_Name"] else "render_fwd_wave_kernel<false>"
        acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        per[k][c + "_KB"] = acc[k] / n[k]; per[k]["launches_" + c] = n[k]
for k, v in per.items():
    v["hbm_bytes_corrected"] = (2 * v.get("FETCH_SIZE_KB", 0.0) + v.get("WRITE_SIZE_KB", 0.0)) * 1024
print(json.dumps({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline --hot-only (two passes); rows of the forward-only tile kernel over the 128 views of the batch (bench region forward_rasterizer; <true> = with the inverse-depth plane, <false> = u3d_render_view_forward)",
                  "formula": "hbm_bytes_corrected = (2*FETCH_SIZE_KB + WRITE_SIZE_KB) * 1024", "per_launch": per}, indent=1))
PY
fi
if [ "$2" = "sq" ]; then
  P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_WAVES"
  P2="SQ_INSTS_LDS SQ_INSTS_VALU_INT32 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
  i=0
  for P in "$P1" "$P2"; do
    i=$((i+1)); rm -rf /tmp/sq_$i
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/sq_$i -- python $R/bench.py --config $CFGNAME $EXTRA --steps 6 --warmup 2 --no-cpu-baseline --hot-only > $O/sq_${i}_$CFG.log 2>&1
    cp $(find /tmp/sq_$i -name "*counter_collection.csv" | head -1) /tmp/sq_$i.csv
  done
  python - $CFG $O <<'PY' > $O/sq_issue_$CFG.json
import csv, json, re, collections, sys
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in ("/tmp/sq_1.csv", "/tmp/sq_2.csv"):
    for r in csv.DictReader(open(f)):
        if "render_fb_wave_kernel" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
per = {k: acc[k] / n[k] for k in acc}
avg_ms = None
for r in csv.DictReader(open(f"{sys.argv[2]}/kernel_stats_bench_{sys.argv[1]}.csv")):
    if "render_fb_wave_kernel" in r["Name"]: avg_ms = float(r["AverageNs"]) / 1e6
print(json.dumps({"command": f"rocprofv3 --pmc <SQ counters, two passes> --kernel-trace -- python bench.py --config {sys.argv[1]} --steps 6 --warmup 2 --no-cpu-baseline --hot-only",
                  "kernel": "render_fb_wave_kernel", "workload": f"{sys.argv[1]} single-pass fused path", "per_launch": per, "launches_sampled": max(n.values()) if n else 0,
                  "kernel_avg_ms": avg_ms, "kernel_avg_ms_source": f"kernel_stats_bench_{sys.argv[1]}.csv (rocprofv3 --kernel-trace --stats, no counters)"}, indent=1))
PY
  # what the tile kernel's waves wait on (round 6): wave / wait / busy cycles (quad-cycles, MI355X_MICROARCH.md), LDS activity, instruction fetch
  P3="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
  P4="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU"
  i=2
  for P in "$P3" "$P4"; do
    i=$((i+1)); rm -rf /tmp/sq_$i
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/sq_$i -- python $R/bench.py --config $CFGNAME $EXTRA --steps 6 --warmup 2 --no-cpu-baseline --hot-only > $O/sq_${i}_$CFG.log 2>&1
    cp $(find /tmp/sq_$i -name "*counter_collection.csv" | head -1) /tmp/sq_$i.csv
  done
  python - $CFG <<'PY' > $O/sq_wait_$CFG.json
import csv, json, collections, sys
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in ("/tmp/sq_3.csv", "/tmp/sq_4.csv"):
    for r in csv.DictReader(open(f)):
        if "render_fb_wave_kernel" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
per = {k: acc[k] / n[k] for k in acc}
cyc = per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
d = {"command": f"rocprofv3 --pmc <two passes> --kernel-trace -- python bench.py --config {sys.argv[1]} --steps 6 --warmup 2 --no-cpu-baseline --hot-only",
     "kernel": "render_fb_wave_kernel", "per_launch": per,
     "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE sums the 8 XCDs"}
if cyc and per.get("SQ_WAVE_CYCLES"):
    wc = per["SQ_WAVE_CYCLES"]
    d["derived"] = {"shader_cycles_per_launch": cyc, "wave_slot_occupancy": 4.0 * wc / (cyc * 7168.0), "wave_slots": 7168,
                    "share_of_wave_time_waiting_for_issue": per.get("SQ_WAIT_INST_ANY", 0.0) / wc, "share_of_wave_time_in_s_waitcnt": per.get("SQ_WAIT_ANY", 0.0) / wc,
                    "share_of_wave_time_issuing": per.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "lds_pipeline_busy_share_per_cu": per.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256.0 / cyc,
                    "cycles_per_valu_instruction_per_simd": cyc * 1024.0 / per["SQ_INSTS_VALU"] if per.get("SQ_INSTS_VALU") else None}
print(json.dumps(d, indent=1))
PY
fi
tail -c 400 $O/bench_line_$CFG.json; echo; head -4 $O/kernel_stats_bench_$CFG.csv | cut -c1-200
