#!/bin/bash
# Timeline of one render_fb launch: build  make -C unipre3d_amd/csrc LIBDIR=../lib_tl EXTRA=-DU3D_TIMELINE ../lib_tl/libunipre3d_rasterizer.so
# then on the GPU box:  bash tools/tile_timeline.sh "C2 C3 C5"
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06t; mkdir -p $O; cp -n unipre3d_amd/lib/_u3d_torch.so unipre3d_amd/lib_tl/ 2>/dev/null
for c in ${1:-C2}; do
  U3D_TIMELINE_OUT=$O/timeline_$c.bin U3D_LIB_DIRNAME=lib_tl U3D_BENCH_FULL_JSON=/tmp/t.json python bench.py --config $c --hot-only --no-cpu-baseline --steps 50 --warmup 10 > /dev/null 2> /tmp/t.err
  grep "\[timeline\]" /tmp/t.err | tail -1 || tail -5 /tmp/t.err
  python tools/tile_timeline.py $O/timeline_$c.bin 7 > $O/timeline_$c.json; cat $O/timeline_$c.json
done
