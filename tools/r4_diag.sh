#!/bin/bash
# round 4: parity of the p1hex kernels, A/B timing, phase timers and ablations of the exact-tile kernel (ablation build)
L=gpurun_out/r4_diag.log; : > $L
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "p1hex" 2>&1 | tail -5 >> $L
run() { echo "== $*" >> $L; env "$@" timeout 300 python tools/c2_time.py 128 200 2>&1 | grep -v amdgpu.ids >> $L; }
run NH_P1HEX_KERNEL=tiles
run NH_P1HEX_KERNEL=skew
run NH_P1HEX_KERNEL=tiles
export NUTILS_AMD_LIB=$PWD/nutils_amd/libnutils_hip_abl.so
run NH_P1HEX_KERNEL=tiles NH_P1HEX_TIMERS=1
for d in 0 8 24 2 28 506; do run NH_P1HEX_KERNEL=tiles NH_P1HEX_DEBUG=$d; done
run NH_P1HEX_KERNEL=tiles NH_P1HEX_NOXCD=1
run NH_P1HEX_KERNEL=skew
cat $L | cut -c1-700
