#!/bin/bash
# Static ISA of render_fb_wave_kernel<1> with the product's flags -> /tmp/isa/render.s, then the per-loop opcode histogram and its
# reconciliation with the measured SQ counters and opcode costs: profiles/r06/isa_histogram_render_fb.json  (CPU only: hipcc cross-compiles)
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -I$R/unipre3d_amd/csrc -munsafe-fp-atomics -fno-slp-vectorize --cuda-device-only -S \
  -o /tmp/isa/render.s $R/unipre3d_amd/csrc/u3d_render.hip 2>/dev/null
python $R/tools/isa_histogram.py /tmp/isa/render.s "_ZN12_GLOBAL__N_121render_fb_wave_kernelILi1E" --json /tmp/isa/fb1.json > /tmp/isa/fb1_loops.txt
python $R/tools/isa_reconcile.py /tmp/isa/fb1.json $R/profiles/r06 > $R/profiles/r06/isa_histogram_render_fb.json && tail -c 1500 $R/profiles/r06/isa_histogram_render_fb.json
