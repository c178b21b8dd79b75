#!/bin/bash
# round-4 evidence run (GPU box, from the repo root): kernel trace of the bench command, PMC passes of the C2 headline kernel (instruction issue) and of
# the three C2 matrix kernels' traffic, the deterministic owner-block path and the C3 kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4 -o f -- python bench.py --no-cpu > gpurun_out/prof_r4.json 2> gpurun_out/prof_r4.err
python tools/rocpd_summary.py $(find gpurun_out/prof_r4 -name "*results.db" | head -1) | head -16 | tee gpurun_out/prof_r4_summary.txt
tail -c 3000 gpurun_out/prof_r4.json
for c in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  bash tools/pmc.sh r4c2_$tag "$c" -- python tools/c2_time.py 128 60 2>&1 | grep -A8 "k_p1hex_skew" | tee -a gpurun_out/prof_r4_pmc_c2.txt
done
for k in skew tiles tri; do
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== $k $c" | tee -a gpurun_out/prof_r4_traffic.txt
    NH_P1HEX_KERNEL=$k bash tools/pmc.sh r4_${k}_$c $c -- python tools/c2_time.py 128 40 2>&1 | grep -A2 "k_p1hex_" | tee -a gpurun_out/prof_r4_traffic.txt
  done
done
