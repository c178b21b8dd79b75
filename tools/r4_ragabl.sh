#!/bin/bash
# round 4: ablation of the generic element kernel on the ragged rational workload (variant builds: results are wrong, the parity assert of the probe is skipped by the grep)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" _gSKIPQ _gSKIPST _gSKIPFILL _gall; do
  rm -rf gpurun_out/prof_r4_rg$v
  NUTILS_AMD_LIB=$GRAFT_REPO_ROOT/nutils_amd/libnutils_hip$v.so rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4_rg$v -o r -- python tools/ragged_probe.py 256 10 > gpurun_out/prof_r4_rg$v.log 2>&1
  echo "== $v"; python tools/rocpd_summary.py gpurun_out/prof_r4_rg$v/r_results.db | grep "k_matrix_generic\|k_gather_values_v" | cut -c1-150
done
