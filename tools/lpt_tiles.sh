#!/bin/bash
# Does the ORDER in which the 32768 tiles of a C2 launch are dispatched matter (ramp / tail of one-wave workgroups whose walks differ 2 x)?
# lib_lpt = make LIBDIR=../lib_lpt EXTRA=-DU3D_LPT_EXPERIMENT: after 30 launches the launcher reads every tile's walk length of the (identical)
# previous step, builds a permutation on the host and the kernel maps its block id through it.
# U3D_LPT_MODE: 0 product order, 1 heaviest tile first (global), 2 lightest first, 3 random (control: what losing the XCD-aware map costs),
#               4 heaviest first within each view, views interleaved (what a production form could compute per view in preprocess_fwd)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06g; mkdir -p $O
for rep in 1 2; do for m in 0 1 2 3 4; do for c in C2; do
  U3D_LPT_MODE=$m U3D_LIB_DIRNAME=lib_lpt U3D_BENCH_FULL_JSON=/tmp/g.json python bench.py --config $c --hot-only --no-cpu-baseline --steps 50 --warmup 10 > /dev/null 2> /tmp/g.err
  grep "\[lpt\]" /tmp/g.err | tail -1
  python -c "
import json
d=json.load(open('/tmp/g.json'))
print('lpt_mode=$m %-3s ms/step %.4f  repeat-min %.4f ' % ('$c', d['ms_per_step'], d['repeatability']['min']), {k: round(v['avg_ms']*1e3,1) for k,v in d['render_loss_step_ms']['kernels'].items()})" | tee -a $O/lpt.log
done; done; done
