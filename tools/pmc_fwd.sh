#!/bin/bash
# SQ wait / issue counters of the tile kernels of one bench run: tools/pmc_fwd.sh [config]
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1)); rm -rf /tmp/pf_$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pf_$i -- python $R/bench.py --config ${1:-C2} --steps 6 --warmup 2 --no-cpu-baseline --hot-only > /tmp/pf_$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int)); dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pf_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "render_" not in k: continue
        name = ("fwd<depth>" if "render_fwd" in k and ("ILb1" in k or "<true>" in k) else "fwd<nodepth>" if "render_fwd" in k else "fb" if "render_fb" in k else "bwd")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[name][r["Counter_Name"]] += 1
        dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for name in acc:
    c = {k: acc[name][k] / n[name][k] for k in acc[name]}
    d = sum(dur[name]) / len(dur[name]) / 1e3
    print(name, "launch us (under counters) %.1f" % d, {k: round(v / 1e6, 2) for k, v in c.items()})
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        print("    of wave-cycles: waiting %.2f, issue-stalled %.2f, issuing %.2f ; VALU per wave %.0f" % (c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1)))
PY
