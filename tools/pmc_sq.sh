#!/bin/bash
# usage: tools/pmc_sq.sh "<counters>" tag  -> gpurun_out/pmc_<tag>.txt (per-kernel averages)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$2
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d /tmp/pmc_$2 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --hot-only $PMC_BENCH_ARGS > /tmp/pmc_$2.log 2>&1
f=$(find /tmp/pmc_$2 -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/pmc_$2.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    import re
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
    if "render_fb" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    print(k, {c: round(acc[k][c] / n[k][c], 1) for c in acc[k]}, "launches", max(n[k].values()))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_$2.txt
