#!/bin/bash
# round 4: the skewed C2 kernel with and without the per-line pipelined flush, ablation builds without phase timers:
# NH_P1HEX_DEBUG 0 = all, 4 = no element math (memory role alone), 2 = no HBM stores, 6 = skeleton
L=gpurun_out/r4_pf_diag.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout 60 python tools/c2_time.py 128 200 2>&1 | grep -v amdgpu.ids >> $L; }
for rep in 1 2; do
for v in ab0 ab4; do
  for d in 0 4 2 6; do run NUTILS_AMD_LIB=$GRAFT_REPO_ROOT/nutils_amd/libnutils_hip_$v.so NH_P1HEX_DEBUG=$d; done
done
done
cat $L
