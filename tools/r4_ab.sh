#!/bin/bash
# round 4: A/B of the C2 matrix kernels (production builds), alternating, 300 timed steps each + parity of the p1hex kernels
L=gpurun_out/r4_ab.log; : > $L
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "p1hex" 2>&1 | tail -3 >> $L
run() { echo "== $*" >> $L; env "$@" timeout 60 python tools/c2_time.py 128 300 2>&1 | grep -v amdgpu.ids >> $L; }
for rep in 1 2; do
run NH_P1HEX_KERNEL=skew
run NH_P1HEX_KERNEL=tri
run NH_P1HEX_KERNEL=tri NH_P1HEX_TRI_DELAY=0
run NH_P1HEX_KERNEL=tri NH_P1HEX_TRI_NOPRIO=1
done
cat $L
