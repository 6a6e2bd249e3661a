#!/bin/bash
# A/B of kernel-variant builds (make LIBDIR=../lib_x EXTRA=-D...): tools/ab_variants.sh "<libdirs>" "<configs>" [bench args]
# prints, per (variant, config): ms/step of the timed region, min of the repeats, per-kernel HIP-event averages (us)
# (reads the FULL record of the run -- bench_full.json -- the stdout line is the compact contract object since round 6)
LIBS=${1:-"lib"}; CFGS=${2:-"C2"}; shift 2
for c in $CFGS; do for l in $LIBS; do
  U3D_BENCH_FULL_JSON=/tmp/ab_full.json U3D_LIB_DIRNAME=$l python bench.py --config $c --hot-only --no-cpu-baseline --steps 50 --warmup 10 "$@" >/dev/null 2>&1
  python -c "
import json
d=json.load(open('/tmp/ab_full.json'))
print('%-4s %-10s ms/step %.4f  repeat-min %.4f  ' % ('$c', '$l', d['ms_per_step'], d['repeatability']['min']), {k: round(v['avg_ms']*1e3,1) for k,v in d['render_loss_step_ms']['kernels'].items()})"
done; done
