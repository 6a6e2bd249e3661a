#!/bin/bash
# Round-6 judged artifacts, pass 1 (counters, kernel stats, opcode costs): everything lands in gpurun_out/prof_r06/; copy to profiles/r06/,
# commit, then run pass 2 (tools/profile_r06_lines.sh) so that every bench line names this round's committed sources.
cd $GRAFT_REPO_ROOT
bash tools/profile_r06.sh C2 sq
bash tools/profile_r06.sh C2 sq --compact
for c in C3 C4 C5; do bash tools/profile_r06.sh $c sq; done
bash tools/profile_r06_rows.sh
tools/ub/opcost > gpurun_out/prof_r06/opcode_issue_costs.jsonl 2>&1
ls -la gpurun_out/prof_r06 | head -80
