#!/bin/bash
# bwd_reduce: tiles in flight per thread (RU): rebuild per variant, rocprofv3 kernel averages
R=$GRAFT_REPO_ROOT
for v in 16 32 64; do
  make -C $R/unipre3d_amd/csrc clean > /dev/null
  make -C $R/unipre3d_amd/csrc -j8 EXTRA="-DRU=$v" > /tmp/sweep_build.log 2>&1 || { tail -5 /tmp/sweep_build.log; continue; }
  echo "RU $v"
  for c in C2 C3 C4 C5; do bash $R/tools/kt.sh x --config $c --hot-only | grep -o "bwd_reduce[0-9]* [0-9.]*"; done
done
make -C $R/unipre3d_amd/csrc clean > /dev/null; make -C $R/unipre3d_amd/csrc -j8 > /dev/null 2>&1
