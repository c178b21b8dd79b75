#!/bin/bash
# round 4: what bounds the memory role of the skewed C2 kernel alone (ablation build without phase timers):
# 4 = no math; +8 = no vertex loads / staging; +16 = half of the store bytes
L=gpurun_out/r4_pf_diag2.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout 60 python tools/c2_time.py 128 200 2>&1 | grep -v amdgpu.ids >> $L; }
for rep in 1 2; do
  for d in 4 12 20 28 6 14 0 16 8; do run NUTILS_AMD_LIB=$GRAFT_REPO_ROOT/nutils_amd/libnutils_hip_ab0.so NH_P1HEX_DEBUG=$d; done
done
cat $L
