#!/bin/bash
# PMC passes of the vector owner kernel on 96^3 trilinear elasticity (GPU box): issue, waits, LDS, occupancy
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES"; do
  tag=own_$(echo $c | tr ' ' '_' | cut -c1-40)
  bash tools/pmc.sh $tag "$c" -- python tools/generic_probe.py "3D P1 elasticity" 2>&1 | grep -A4 "k_owner_rows_v"
  rm -rf gpurun_out/pmc_$tag
done
