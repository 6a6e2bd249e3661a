#!/bin/bash
# What are render_fb's HBM writes?  (56.8 MB of PMC writes per C2 launch against 27.6 MB of partial rows
# bwd_reduce reads back.)  The single-pass tile kernel is pinned to 8 waves per SIMD (64 VGPRs) and spills 23 VGPRs to scratch
# (private_segment_fixed_size 64 B per lane = 4 KB per wave); scratch lines that are evicted from L2 are written to HBM.  This probe
# measures WRITE_SIZE / FETCH_SIZE of the kernel for the shipped build and for the same source compiled without the occupancy pin
# (no spills), plus the durations of both:  tools/pmc_spill_probe.sh [C2]
CFG=${1:-C2}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
probe() {   # $1 = tag
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/sp_$1_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/sp_$1_$c -- python $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --hot-only > /tmp/sp_$1_$c.log 2>&1
    cp $(find /tmp/sp_$1_$c -name "*counter_collection.csv" | head -1) /tmp/sp_$1_$c.csv
  done
  rm -rf /tmp/sp_$1_kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$1_kt -- python $R/bench.py --config $CFG --steps 30 --warmup 5 --no-cpu-baseline --hot-only > /tmp/sp_$1_kt.log 2>&1
  cp $(find /tmp/sp_$1_kt -name "*kernel_stats.csv" | head -1) /tmp/sp_$1_kt.csv
}
probe pinned
make -C $R/unipre3d_amd/csrc clean > /dev/null; make -C $R/unipre3d_amd/csrc -j8 EXTRA=-DU3D_NO_OCC_PIN > /tmp/sp_build.log 2>&1
probe unpinned
make -C $R/unipre3d_amd/csrc clean > /dev/null; make -C $R/unipre3d_amd/csrc -j8 > /tmp/sp_build2.log 2>&1
python - $CFG <<'PY' > $O/pmc_spill_probe_$CFG.json
import csv, json, re, sys, collections
out = {"what": "HBM traffic and duration of render_fb_wave_kernel: shipped build (amdgpu_waves_per_eu(8,8): 64 VGPRs, 23 spilled, 64 B/lane scratch) against the "
               "same source without the pin (no scratch)", "config": sys.argv[1], "formula": "hbm bytes = WRITE_SIZE_KB * 1024 (exact) ; 2 * FETCH_SIZE_KB * 1024 (gfx950 calibration)"}
for tag in ("pinned", "unpinned"):
    d = {}
    for c in ("WRITE_SIZE", "FETCH_SIZE"):
        acc = n = 0
        for r in csv.DictReader(open(f"/tmp/sp_{tag}_{c}.csv")):
            if r["Counter_Name"] == c and "render_fb_wave_kernel" in r["Kernel_Name"]:
                acc += float(r["Counter_Value"]); n += 1
        d[c + "_bytes_per_launch"] = (acc / max(n, 1)) * 1024 * (2 if c == "FETCH_SIZE" else 1)
    for r in csv.DictReader(open(f"/tmp/sp_{tag}_kt.csv")):
        if "render_fb_wave_kernel" in r["Name"]:
            d["avg_us"] = float(r["AverageNs"]) / 1e3
    out[tag] = d
out["scratch_share_of_writes_bytes"] = out["pinned"]["WRITE_SIZE_bytes_per_launch"] - out["unpinned"]["WRITE_SIZE_bytes_per_launch"]
print(json.dumps(out, indent=1))
PY
cat $O/pmc_spill_probe_$CFG.json
