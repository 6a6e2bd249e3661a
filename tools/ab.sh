#!/bin/bash
# A/B two builds of the rasterizer library on the same box: tools/ab.sh <libA> [configs...]
cd $GRAFT_REPO_ROOT; A=$1; shift; CFGS=${@:-C2}
cp unipre3d_amd/lib/libunipre3d_rasterizer.so /tmp/new.so
for rep in 1 2; do
for v in A B; do
  if [ $v = A ]; then cp $A unipre3d_amd/lib/libunipre3d_rasterizer.so; else cp /tmp/new.so unipre3d_amd/lib/libunipre3d_rasterizer.so; fi
  for c in $CFGS; do python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-e2e 2>&1 | tail -1 | python tools/brief.py $v-$c; done
done; done
cp /tmp/new.so unipre3d_amd/lib/libunipre3d_rasterizer.so
