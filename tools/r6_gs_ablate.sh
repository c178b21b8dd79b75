#!/bin/bash
# ablation of k_gram_sym (tools/ragged_probe.py through variant libraries built with `make var NAME=gs_<x> FLAGS=-DNH_GS_<x> FILE=nh_assemble_generic`): kernel-trace average of the big size class
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" ${GS_VARIANTS:-gs_NOLOAD gs_NOSTORE gs_NOLOOP gs_all}; do
  lib=nutils_amd/libnutils_hip${v:+_$v}.so
  mkdir -p gpurun_out/ab_$v
  NUTILS_AMD_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$v -o f -- python tools/ragged_probe.py 256 20 > /dev/null 2>&1
  echo "variant ${v:-base}: $(python tools/rocpd_summary.py $(find gpurun_out/ab_$v -name '*results.db' | head -1) | grep "k_gram_sym\|k_gather_values" | cut -d'|' -f2,3,5,6 | tr '\n' ' ')"
  rm -rf gpurun_out/ab_$v
done
