#!/bin/bash
# round-6 measurement batch c: fusion-backward variants (plane-major, bitmap), group_points_grad channel blocks, interpolate shapes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_fusion.py tests/test_gpu_pointops.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
row() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['fusion']; po=d['pointops']['ops']
print('$1', 'fusion bwd', round(f['backward']['us'],1), 'fill', round(f['backward']['us_plain_zero_fill_of_the_same_bytes'],1), f['backward']['equals_oracle'], 'fwd', round(f['forward']['us'],1), f['forward']['equals_oracle'],
      '| gg', round(po['group_points_grad_c384']['us'],1), po['group_points_grad_c384']['equals_oracle'], '| interp', round(po['three_interpolate']['us'],1), po['three_interpolate']['equals_oracle'],
      '| igrad', round(po['three_interpolate_grad']['us'],1), '| all', d['pointops']['all_equal_oracle'])"; }
for v in 6 8 9 10 11 12; do U3D_FUSION_BWD_VARIANT=$v python bench.py --next-rows-only 2>/dev/null | row "fusion_variant=$v" | tee -a $O/variants.log; done
for cb in 4 8 16; do U3D_FUSION_BWD_VARIANT=8 U3D_GG_CB=$cb python bench.py --next-rows-only 2>/dev/null | row "gg_cb=$cb" | tee -a $O/variants.log; done
for l in lib_il2 lib_il8 lib_ilp8; do U3D_FUSION_BWD_VARIANT=8 U3D_LIB_DIRNAME=$l python bench.py --next-rows-only 2>/dev/null | row "$l" | tee -a $O/variants.log; done
