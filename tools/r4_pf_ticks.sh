#!/bin/bash
# round 4: phase timers (cycles per wave and launch) of the skewed C2 kernel with / without the per-line pipelined flush
L=gpurun_out/r4_pf_ticks.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout 120 python tools/c2_time.py 128 3 2>&1 | grep -v amdgpu.ids | tail -2 >> $L; }
for v in tk0 tk4; do
  for d in 0 4 2; do run NUTILS_AMD_LIB=$GRAFT_REPO_ROOT/nutils_amd/libnutils_hip_$v.so NH_P1HEX_TIMERS=1 NH_P1HEX_DEBUG=$d; done
done
cat $L
