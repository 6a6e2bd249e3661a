#!/bin/bash
# after a change to the point operators only: their tests, the rows profile, and the two bench lines that carry the rows' digest
cd $GRAFT_REPO_ROOT; O=gpurun_out/prof_r06; mkdir -p $O
python -m pytest tests/test_gpu_pointops.py tests/test_fusion.py -m gpu -q 2>&1 | tail -3
bash tools/profile_r06_rows.sh | tail -25
ONLY_LINE=1 bash tools/profile_r06.sh C2 sq
U3D_BENCH_FULL_JSON=$O/bench_full_default_run.json python bench.py > $O/bench_line_default_run.json 2> $O/bench_default_run.err; tail -c 300 $O/bench_line_default_run.json; echo
