#!/bin/bash
# A/B of variant builds on the TWO-PASS route (operator forward + backward kernels): tools/ab_two_pass.sh "<libdirs>" "<configs>"
LIBS=${1:-"lib"}; CFGS=${2:-"C2"}
cd $GRAFT_REPO_ROOT
for c in $CFGS; do for l in $LIBS; do
  U3D_BENCH_FULL_JSON=/tmp/ab2_full.json U3D_LIB_DIRNAME=$l python bench.py --config $c --two-pass --hot-only --no-cpu-baseline --steps 50 --warmup 10 >/dev/null 2>&1
  python -c "
import json
d=json.load(open('/tmp/ab2_full.json'))
print('%-4s %-8s two-pass ms/step %.4f  repeat-min %.4f  ' % ('$c', '$l', d['ms_per_step'], d['repeatability']['min']), {k: round(v['avg_ms']*1e3,1) for k,v in d['render_loss_step_ms']['kernels'].items()})"
done; done
