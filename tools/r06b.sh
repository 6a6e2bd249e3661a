#!/bin/bash
# round-6 measurement batch b: tests of the rewritten kernels, render_fb A/B, fusion-backward variants, interpolate A/B, opcode costs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_fusion.py tests/test_gpu_pointops.py tests/test_gpu_more_parity.py tests/test_gpu_parity.py tests/test_gpu_batched.py tests/test_gpu_ragged.py -m gpu -q -x > $O/tests.log 2>&1; tail -12 $O/tests.log
echo "== render_fb A/B (lib_base: r05 loops; lib_fh: forward hoist; lib_bl: backward limit-free; lib: both)"
for rep in 1 2; do bash tools/ab_variants.sh "lib_base lib_fh lib_bl lib" "C2 C3 C5" 2>&1 | tee -a $O/ab_render.log; done
echo "== fusion backward variants"
for v in 0 1 2 3 4 5 6 7; do U3D_FUSION_BWD_VARIANT=$v python bench.py --next-rows-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['fusion']['backward']; print('variant $v bwd us', round(f['us'],1), 'fill', round(f['us_plain_zero_fill_of_the_same_bytes'],1), 'equal', f['equals_oracle'], 'fwd', round(d['fusion']['forward']['us'],1))" | tee -a $O/fusion_variants.log; done
echo "== pointops (LDS interpolate / global interpolate)"
for e in 0 1; do if [ $e = 1 ]; then export U3D_INTERP_GLOBAL=1; fi; python bench.py --next-rows-only 2>/dev/null > $O/rows_$e.json; python -c "
import json
d=json.loads(open('$O/rows_$e.json').read().strip().splitlines()[-1]); po=d['pointops']['ops']
print('interp_global=$e', {k: (round(v['us'],1), round(v['roofline']['frac'],3), v['equals_oracle']) for k,v in po.items() if not k.startswith('fps')}, d['pointops']['all_equal_oracle'])"; done
unset U3D_INTERP_GLOBAL
echo "== opcode issue costs"
tools/ub/opcost | tee $O/opcost.jsonl
