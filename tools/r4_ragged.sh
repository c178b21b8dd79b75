#!/bin/bash
# round 4: the ragged rational workload (tools/ragged_probe.py 256 10): kernel trace and issue counters of its element kernel
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/ragged_probe.py 256 10 2>&1 | tail -3
rm -rf gpurun_out/prof_r4_ragged
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4_ragged -o r -- python tools/ragged_probe.py 256 10 > gpurun_out/prof_r4_ragged.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_r4_ragged/r_results.db | head -8 | cut -c1-200
for c in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  bash tools/pmc.sh r4rg_$tag "$c" -- python tools/ragged_probe.py 256 10 2>&1 | grep -A3 "k_matrix_generic"
done
