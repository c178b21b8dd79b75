#!/bin/bash
# round 4: A/B of builds of the skewed C2 kernel (NUTILS_AMD_LIB), alternating, 300 timed steps each; parity of the variant first
L=gpurun_out/r4_ab2.log; : > $L
for v in $VARIANTS; do
  NUTILS_AMD_LIB=$GRAFT_REPO_ROOT/nutils_amd/libnutils_hip_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "p1hex" 2>&1 | tail -2 >> $L
done
run() { echo "== $*" >> $L; env "$@" timeout 60 python tools/c2_time.py 128 300 2>&1 | grep -v amdgpu.ids >> $L; }
for rep in 1 2 3; do
  run X=base
  for v in $VARIANTS; do run NUTILS_AMD_LIB=$GRAFT_REPO_ROOT/nutils_amd/libnutils_hip_$v.so; done
done
cat $L
