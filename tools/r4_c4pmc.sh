#!/bin/bash
# round 4: instruction-issue counters of the two-phase Jacobian pass (k_local_terms2) in a Newton step at 512^2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  bash tools/pmc.sh r4c4_$tag "$c" -- python tools/c4_step.py 512 2>&1 | grep -A8 "${KERNEL:-k_local_terms2}" | tee -a gpurun_out/prof_r4_pmc_c4.txt
done
