#!/bin/bash
export NUTILS_AMD_LIB=$PWD/nutils_amd/libnutils_hip_abl.so
L=gpurun_out/r4_diag2.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout 60 python tools/c2_time.py 128 200 2>&1 | grep -v amdgpu.ids >> $L; }
for d in 0 8 24 2 28 506 510; do run NH_P1HEX_KERNEL=tiles NH_P1HEX_DEBUG=$d; done
run NH_P1HEX_KERNEL=skew
cat $L
