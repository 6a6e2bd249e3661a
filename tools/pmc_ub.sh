#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_ub
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_ub -- $GRAFT_REPO_ROOT/tools/ub/ops > /tmp/pmc_ub.log 2>&1
f=$(find /tmp/pmc_ub -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    rows[(r["Dispatch_Id"], r["Kernel_Name"][:40])][r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in sorted(rows.items(), key=lambda x: int(x[0][0])):
    iv = v.get("SQ_INSTS_VALU", 1)
    print(k[0], k[1], "insts %.3g active_q/inst %.2f gui/8 %.0f cyc/inst/simd %.2f" % (iv, v.get("SQ_ACTIVE_INST_VALU", 0) / iv, v.get("GRBM_GUI_ACTIVE", 0) / 8, v.get("GRBM_GUI_ACTIVE", 0) / 8 / (iv / 1024)))
PY
