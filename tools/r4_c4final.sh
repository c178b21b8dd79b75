#!/bin/bash
# round 4: evidence run for the configs[3] kernels: kernel trace of tools/c4_step.py 512 with the round-4 thread passes (default) and with the round-3 kernels
# (NUTILS_AMD_LOCAL_TERMS=1 NUTILS_AMD_LOCAL_VTERMS=1), then the issue counters of the two new kernels
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r4_c4new gpurun_out/prof_r4_c4old
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4_c4new -o c4 -- python tools/c4_step.py 512 > gpurun_out/prof_r4_c4new.log 2>&1
NUTILS_AMD_LOCAL_TERMS=1 NUTILS_AMD_LOCAL_VTERMS=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4_c4old -o c4 -- python tools/c4_step.py 512 > gpurun_out/prof_r4_c4old.log 2>&1
for v in new old; do
  echo "== $v"; grep -v "^W2026\|^E2026" gpurun_out/prof_r4_c4$v.log | tail -5
  python tools/rocpd_summary.py gpurun_out/prof_r4_c4$v/c4_results.db | head -14 | cut -c1-220
done > gpurun_out/prof_r4_c4_summary.txt
rm -f gpurun_out/prof_r4_pmc_c4.txt
KERNEL="k_local_" bash tools/r4_c4pmc.sh > /dev/null 2>&1
cat gpurun_out/prof_r4_c4_summary.txt
grep -A3 "k_local_vterms2\|k_local_terms2" gpurun_out/prof_r4_pmc_c4.txt
