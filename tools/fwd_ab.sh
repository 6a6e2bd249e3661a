#!/bin/bash
# A/B of the forward-only tile kernel (bench region forward_rasterizer): tools/fwd_ab.sh "<libdirs>" [config]
cd $GRAFT_REPO_ROOT
for l in ${1:-lib}; do
  U3D_LIB_DIRNAME=$l python bench.py --config ${2:-C2} --hot-only --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['forward_rasterizer']
print('$l', 'fwd us %.2f frac_alg %.3f frac_pmc %.3f | no-depth us %.2f frac_pmc %.3f | step ms %.4f' % (1e3*f['avg_ms'], f['frac_of_8TBs'], f['frac_pmc_bytes'], 1e3*f['without_inverse_depth']['avg_ms'], f['without_inverse_depth']['frac_pmc_bytes'], d['ms_per_step']))"
done
