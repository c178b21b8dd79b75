#!/bin/bash
# round 4: two-phase Jacobian pass (k_local_terms2): parity tests that reach it, then the kernel trace of a Newton step at 512^2, old and new kernel
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_examples.py tests/test_gpu_plans.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4_c4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_c4_tests.log
tail -5 gpurun_out/r4_c4_tests.log
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mode in 0 1; do
  rm -rf gpurun_out/prof_r4_c4_$mode
  NUTILS_AMD_LOCAL_TERMS=$mode timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4_c4_$mode -o c4 -- python tools/c4_step.py 512 > gpurun_out/prof_r4_c4_$mode.log 2>&1
  echo "== NUTILS_AMD_LOCAL_TERMS=$mode"
  tail -4 gpurun_out/prof_r4_c4_$mode.log
  python tools/rocpd_summary.py gpurun_out/prof_r4_c4_$mode 2>/dev/null | head -12
done
