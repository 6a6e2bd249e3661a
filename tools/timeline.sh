#!/bin/bash
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
rm -rf /tmp/tl_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --hot-only "$@" > /tmp/tl_$tag.log 2>&1
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python - "$f" "$tag" <<'PY'
import csv, sys, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); n = m.group(1) if m else r["Kernel_Name"][:25]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.replace("_kernel", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == "preprocess_fwd"]
per = [rows[b][0] - rows[a][0] for a, b in zip(starts, starts[1:])]
per = sorted(per)[: max(1, len(per) * 3 // 4)]
print(sys.argv[2], "steps", len(starts), "median-ish period %.1f us" % (per[len(per) // 2] / 1e3))
i0 = starts[len(starts) * 3 // 4]; i1 = starts[len(starts) * 3 // 4 + 1]
t0 = rows[i0][0]
for r in rows[i0:i1]:
    print("   %-28s start %8.1f  dur %7.1f" % (r[2], (r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3))
PY
