#!/bin/bash
# GPU-side account of the per-view drop-in route: rocprofv3 kernel trace of tools/per_view_route.py -> per-kernel totals
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pv; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $GRAFT_REPO_ROOT/tools/per_view_route.py > /tmp/pv.log 2>&1
tail -2 /tmp/pv.log
f=$(find /tmp/pv -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cp $f $GRAFT_REPO_ROOT/gpurun_out/per_view_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("kernels: %d launches, %.2f ms total GPU kernel time (7 steps incl. warm-up)" % (calls, tot / 1e6))
ours = [r for r in rows if "u3d" in r["Name"] or "render_" in r["Name"] or "preprocess" in r["Name"] or "bwd_reduce" in r["Name"] or "quat" in r["Name"]]
print("ours: %d launches %.2f ms" % (sum(int(r["Calls"]) for r in ours), sum(float(r["TotalDurationNs"]) for r in ours) / 1e6))
for r in rows[:25]:
    print("%8d calls %9.1f us avg %8.2f ms total  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Name"][:110]))
PY
