#!/bin/bash
# round 4: k_local_terms2 register budgets (waves per SIMD 3 = default build, 2, 4): kernel trace of tools/c4_step.py 512
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-"" _lt2w2 _lt2w4}; do
  rm -rf gpurun_out/prof_r4_c4v$v
  NUTILS_AMD_LIB=$GRAFT_REPO_ROOT/nutils_amd/libnutils_hip$v.so timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4_c4v$v -o c4 -- python tools/c4_step.py 512 > gpurun_out/prof_r4_c4v$v.log 2>&1
  echo "== lib '$v'"
  grep -v "^W2026\|^E2026" gpurun_out/prof_r4_c4v$v.log | tail -2
  python tools/rocpd_summary.py gpurun_out/prof_r4_c4v$v/c4_results.db 2>/dev/null | grep "k_local_terms\|k_terms_multi" | cut -c1-160
done
