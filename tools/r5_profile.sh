#!/bin/bash
# round-5 evidence run (GPU box, from the repo root): the bench line, its kernel trace, kernel traces and HBM traffic (PMC, separate passes) of the owner kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
tail -c 600 gpurun_out/r5_bench.json
KT_LINES=16 KT_TAIL=1 bash tools/kt.sh r5_bench python bench.py --no-cpu --traffic static > /dev/null
KT_LINES=10 KT_TAIL=2 bash tools/kt.sh r5_vector python tools/vector_probe.py 96 10 > /dev/null
NUTILS_AMD_NO_FAST_PATH=1 KT_LINES=8 KT_TAIL=1 bash tools/kt.sh r5_generic python tools/generic_probe.py "3D P1 128" > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== owner $c" | tee -a gpurun_out/r5_traffic.txt
  bash tools/pmc.sh r5o_$c $c -- python tools/vector_probe.py 96 4 2>&1 | grep -A2 "k_owner_rows_v" | tee -a gpurun_out/r5_traffic.txt
  echo "== fused $c" | tee -a gpurun_out/r5_traffic.txt
  bash tools/pmc.sh r5f_$c $c -- python tools/generic_probe.py "3D P1 128" 2>&1 | grep -A2 "k_fused_p1hex" | tee -a gpurun_out/r5_traffic.txt
  rm -rf gpurun_out/pmc_r5o_$c gpurun_out/pmc_r5f_$c
done
