#!/bin/bash
export NUTILS_AMD_LIB=$PWD/nutils_amd/libnutils_hip_abl.so
L=gpurun_out/r4_diag3.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout 300 python tools/c2_time.py 128 200 2>&1 | grep -v amdgpu.ids >> $L; }
run NH_P1HEX_KERNEL=tiles NH_P1HEX_TIMERS=1
run NH_P1HEX_KERNEL=tiles NH_P1HEX_TIMERS=1 NH_P1HEX_DEBUG=8
run NH_P1HEX_KERNEL=tiles NH_P1HEX_TIMERS=1 NH_P1HEX_DEBUG=2
run NH_P1HEX_KERNEL=tiles NH_P1HEX_TIMERS=1 NH_P1HEX_DEBUG=4
run NH_P1HEX_KERNEL=skew NH_P1HEX_TIMERS=1
cat $L | cut -c1-500
