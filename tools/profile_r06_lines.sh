#!/bin/bash
# Round-6 judged artifacts, pass 2: bench lines (compact stdout line + full record) regenerated AFTER profiles/r06/ holds this round's PMC / SQ /
# ISA files, the driver-form default run, the full GPU test log, the whole-batch parity log and the soak logs.
cd $GRAFT_REPO_ROOT; O=gpurun_out/prof_r06; mkdir -p $O
ONLY_LINE=1 bash tools/profile_r06.sh C2 sq
ONLY_LINE=1 bash tools/profile_r06.sh C2 sq --compact
for c in C3 C4 C5; do ONLY_LINE=1 bash tools/profile_r06.sh $c sq; done
U3D_BENCH_FULL_JSON=$O/bench_full_default_run.json python bench.py > $O/bench_line_default_run.json 2> $O/bench_default_run.err; tail -c 600 $O/bench_line_default_run.json; echo
python -m pytest tests -m gpu -q > $O/gputest_full_suite.log 2>&1; tail -12 $O/gputest_full_suite.log
python tools/whole_batch_parity.py C3 C4 C5 C4_fused C5_fused > $O/whole_batch_parity.log 2>&1; tail -5 $O/whole_batch_parity.log
(python tools/stress.py 11 150; python tools/stress_ragged.py 13 100) > $O/soak_stress.log 2>&1; tail -4 $O/soak_stress.log
