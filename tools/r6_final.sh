#!/bin/bash
# evidence of the round's state (GPU box, from the repo root): the default bench line, the kernel trace of the same command, PMC traffic (separate passes) of kernels the
# static traffic file lacks -> gpurun_out/r6f_*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r6f_bench.json 2> gpurun_out/r6f_bench.err
tail -c 300 gpurun_out/r6f_bench.json
KT_LINES=26 KT_TAIL=1 bash tools/kt.sh r6f_bench python bench.py --no-cpu --traffic static > /dev/null
cat gpurun_out/kt_r6f_bench.txt | cut -c1-200 | head -8
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== c3 uniform $c" | tee -a gpurun_out/r6f_traffic.txt
  bash tools/pmc.sh r6u_$c $c -- python tools/c3_bench.py 64 6 uniform 2>&1 | grep -A2 "k_p2hex_rows_uniform" | tee -a gpurun_out/r6f_traffic.txt
  echo "== c5 $c" | tee -a gpurun_out/r6f_traffic.txt
  bash tools/pmc.sh r6c5_$c $c -- python tools/ragged_probe.py 256 6 2>&1 | grep -A2 "k_gram_sym\|k_gather_values_2x2_tri\|k_mirror_2x2" | tee -a gpurun_out/r6f_traffic.txt
  echo "== owner_v $c" | tee -a gpurun_out/r6f_traffic.txt
  bash tools/pmc.sh r6ov_$c $c -- python tools/vector_probe.py 96 6 2>&1 | grep -A2 "k_owner_rows_v" | tee -a gpurun_out/r6f_traffic.txt
  rm -rf gpurun_out/pmc_r6u_$c gpurun_out/pmc_r6c5_$c gpurun_out/pmc_r6ov_$c
done
