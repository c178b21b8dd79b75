#!/bin/bash
# round 4: first run of the exact-tile kernel: parity tests of the p1hex kernels, then A/B timing against the skew kernel
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "p1hex" > gpurun_out/r4_first_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_first_tests.log
tail -15 gpurun_out/r4_first_tests.log
for k in tiles skew tiles skew; do
  echo "== $k" >> gpurun_out/r4_first_time.log
  NH_P1HEX_KERNEL=$k timeout 300 python tools/c2_time.py 128 300 >> gpurun_out/r4_first_time.log 2>&1
done
cat gpurun_out/r4_first_time.log
