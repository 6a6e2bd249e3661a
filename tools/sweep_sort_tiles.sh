#!/bin/bash
# msd_hist / msd_scatter keys per workgroup (U3D_RADIX_NT_* x U3D_RADIX_IT_*): rebuild per variant, rocprofv3 kernel averages at C4 / C5
R=$GRAFT_REPO_ROOT
for v in "1024 1 1024 2" "1024 2 1024 4" "1024 4 1024 8" "512 2 512 4" "512 4 512 8"; do
  set -- $v
  make -C $R/unipre3d_amd/csrc clean > /dev/null
  make -C $R/unipre3d_amd/csrc -j8 EXTRA="-DU3D_RADIX_NT_SMALL=$1 -DU3D_RADIX_IT_SMALL=$2 -DU3D_RADIX_NT_LARGE=$3 -DU3D_RADIX_IT_LARGE=$4" > /tmp/sweep_build.log 2>&1 || { tail -5 /tmp/sweep_build.log; continue; }
  echo "variant small $1 x $2, large $3 x $4"
  bash $R/tools/kt.sh c4 --config C4 --hot-only | grep -o "msd_[a-z]* [0-9.]*\|bucket_sort [0-9.]*" | tr "\n" " "; echo
  bash $R/tools/kt.sh c5 --config C5 --hot-only | grep -o "msd_[a-z]* [0-9.]*\|bucket_sort [0-9.]*" | tr "\n" " "; echo
done
make -C $R/unipre3d_amd/csrc clean > /dev/null; make -C $R/unipre3d_amd/csrc -j8 > /dev/null 2>&1
