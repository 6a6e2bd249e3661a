#!/bin/bash
# A/B of furthest-point-sampling builds: tools/fps_ab.sh "<libdirs>"   (make LIBDIR=../lib_x EXTRA=-DU3D_FPS_...)
cd $GRAFT_REPO_ROOT
for l in ${1:-lib}; do
  U3D_LIB_DIRNAME=$l python bench.py --next-rows-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['pointops']['ops']
print('$l', {k: (round(v['us'],1), round(v['us_per_selection'],3)) for k,v in d.items() if k.startswith('fps')}, all(v['equals_oracle'] for v in d.values()))"
done
