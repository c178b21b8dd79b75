#!/bin/bash
# round-3 evidence run (GPU box, from the repo root): kernel traces of the bench command + PMC traffic of the two write-once kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3_final -o f -- python bench.py --no-cpu > gpurun_out/prof_r3_final.json 2> gpurun_out/prof_r3_final.err
python tools/rocpd_summary.py $(find gpurun_out/prof_r3_final -name "*results.db" | head -1) | head -14
tail -c 1500 gpurun_out/prof_r3_final.json
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc.sh r3c2_$c $c -- python tools/c2_time.py 2>&1 | grep -A3 "k_p1hex_skew"
  bash tools/pmc.sh r3c3_$c $c -- python tools/c3_bench.py 64 5 2>&1 | grep -A3 "k_p2hex_inreg"
done
