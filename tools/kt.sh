#!/bin/bash
# per-kernel average durations (rocprofv3 kernel trace) of the bench: tools/kt.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e "$@" > /tmp/kt_$tag.log 2>&1
f=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" "$tag" <<'PY'
import csv, sys, re
out = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_kernel)", r["Name"])
    if m and m.group(1) in ("render_fb_wave_kernel", "bwd_reduce_kernel", "bwd_reduce1_kernel", "preprocess_fwd_kernel", "preprocess_bwd_kernel", "quat_norms_kernel", "quat_fixup_kernel", "render_fwd_wave_kernel", "render_bwd_wave_kernel", "radix_hist_kernel", "radix_scatter_kernel", "depth_sort_block_radix_kernel", "msd_hist_kernel", "msd_scatter_kernel", "bucket_sort_kernel"):
        out.append("%s %.1f(x%s)" % (m.group(1).replace("_kernel", ""), float(r["AverageNs"]) / 1e3, r["Calls"]))
print(sys.argv[2], " ".join(out))
PY
