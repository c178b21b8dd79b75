#!/bin/bash
# round 4: A/B of the C2 matrix kernels (production builds), alternating, 300 timed steps each
L=gpurun_out/r4_ab.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout 120 python tools/c2_time.py 128 300 2>&1 | grep -v amdgpu.ids >> $L; }
for rep in 1 2; do
run NH_P1HEX_KERNEL=skew
run NH_P1HEX_KERNEL=skew NH_P1HEX_STAGE_LATE=1
run NH_P1HEX_KERNEL=skew NUTILS_AMD_LIB=$PWD/nutils_amd/libnutils_hip_wpe.so
run NH_P1HEX_KERNEL=tiles
run NH_P1HEX_KERNEL=tiles NUTILS_AMD_LIB=$PWD/nutils_amd/libnutils_hip_wpe.so
done
cat $L
