#!/bin/bash
# round-6 batch e: what do render_fb's waves wait on?  SQ wait / busy / LDS / ifetch counters of the C2 hot-only run, one --pmc pass per set
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$O/avail.txt 2>&1)
grep -o "SQ_[A-Z0-9_]*" $O/avail.txt | sort -u | tr '\n' ' ' | head -c 6000 > $O/sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVES SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU SQ_WAVE32_INSTS SQ_VALU_MFMA_BUSY_CYCLES SQ_LEVEL_WAVES"; do
  i=$((i+1)); bash tools/pmc_sq.sh "$set" r06e_$i > $O/pass_$i.txt 2>&1; tail -2 $O/pass_$i.txt; tail -3 /tmp/pmc_r06e_$i.log | cut -c1-300
done
