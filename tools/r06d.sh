#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06d; mkdir -p $O
row() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['fusion']; po=d['pointops']['ops']
print('$1', '| gg', round(po['group_points_grad_c384']['us'],1), po['group_points_grad_c384']['equals_oracle'], '| igrad', round(po['three_interpolate_grad']['us'],1), po['three_interpolate_grad']['equals_oracle'], '| gather', round(po['gather_points_c384']['us'],1), '| all', d['pointops']['all_equal_oracle'])"; }
for cb in 1 2 4; do U3D_GG_CB=$cb python bench.py --next-rows-only 2>/dev/null | row "gg_cb=$cb" | tee -a $O/variants.log; done
for cb in 1 2 4 8 16; do U3D_IG_CB=$cb python bench.py --next-rows-only 2>/dev/null | row "ig_cb=$cb" | tee -a $O/variants.log; done
