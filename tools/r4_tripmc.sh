#!/bin/bash
# round 4: issue counters of the three-role C2 kernel (k_p1hex_tri) beside the skewed one's (profiles/r04_c2_kernels.md section 2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for k in tri skew; do
for c in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-36)
  NH_P1HEX_KERNEL=$k bash tools/pmc.sh r4${k}_$tag "$c" -- python tools/c2_time.py 128 60 2>&1 | grep -A3 "k_p1hex_$k"
done
done
