#!/bin/bash
# Rows N1 (point ops) and N4a (fusion): tools/profile_r06_rows.sh
#   gpurun_out/prof_r06/bench_rows.json            python bench.py --next-rows-only   (per-operator us, roofline, CPU-oracle baseline)
#   gpurun_out/prof_r06/kernel_stats_pointops.csv  rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/prof_r06/pmc_traffic_pointops.json  separate --pmc FETCH_SIZE / WRITE_SIZE passes, per kernel (FETCH doubled: gfx950 calibration)
# Copy what is to be judged into profiles/r06/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --next-rows-only > $O/bench_rows.json 2> $O/bench_rows.err
rm -rf /tmp/kt_rows; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_rows -- python $R/bench.py --next-rows-only > $O/kt_rows.log 2>&1
cp $(find /tmp/kt_rows -name "*kernel_stats.csv" | head -1) $O/kernel_stats_pointops.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcr_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcr_$c -- python $R/bench.py --next-rows-only > $O/pmc_rows_$c.log 2>&1
  cp $(find /tmp/pmcr_$c -name "*counter_collection.csv" | head -1) /tmp/ccr_$c.csv
done
python - <<'PY' > $O/pmc_traffic_pointops.json
import csv, json, re, collections
per = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f"/tmp/ccr_{c}.csv")):
        if r["Counter_Name"] != c: continue
        m = re.search(r"(\w+_kernel\w*(<[^>]*>)?)", r["Kernel_Name"])
        if not m: continue
        # launches of one kernel differ by shape: keep them apart by grid size
        k = f"{m.group(1)} grid={r.get('Grid_Size', '?')} wg={r.get('Workgroup_Size', '?')}"
        acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        per[k][c + "_KB"] = acc[k] / n[k]; per[k]["launches_" + c] = n[k]
for k, v in per.items():
    v["hbm_bytes_corrected"] = (2 * v.get("FETCH_SIZE_KB", 0.0) + v.get("WRITE_SIZE_KB", 0.0)) * 1024
print(json.dumps({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --next-rows-only (two separate passes)",
                  "formula": "hbm_bytes_corrected = (2*FETCH_SIZE_KB + WRITE_SIZE_KB) * 1024 (gfx950: FETCH_SIZE reads half the bytes)",
                  "per_launch": per}, indent=1))
PY
python -c "
import json; d=json.load(open('$O/bench_rows.json'))
for k,v in d['pointops']['ops'].items(): print(f\"{k:28s} {v['us']:9.2f} us  frac {v['roofline']['frac']:.4f}  cpu {v['cpu_baseline']['value']:10.1f} us  eq {v['equals_oracle']}\")
f=d['fusion']; print('fusion fwd', f['forward']['us'], f['forward']['roofline']['frac'], f['forward']['equals_oracle'], 'bwd', f['backward']['us'], f['backward']['us_plain_zero_fill_of_the_same_bytes'], f['backward']['equals_oracle'])
"
tail -3 $O/bench_rows.err
grep -i "fps\|ball\|group\|three\|zbuf\|gather\|scatter" $O/kernel_stats_pointops.csv | cut -c1-220
