#!/bin/bash
# kernel trace of one command on the GPU box: tools/kt.sh <tag> <command...>  -> gpurun_out/kt_<tag>.txt (per-kernel table)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$tag -o f -- "$@" > gpurun_out/kt_$tag.out 2> gpurun_out/kt_$tag.err
python tools/rocpd_summary.py $(find gpurun_out/kt_$tag -name "*results.db" | head -1) | head -${KT_LINES:-14} | tee gpurun_out/kt_$tag.txt
tail -n ${KT_TAIL:-6} gpurun_out/kt_$tag.out
rm -rf gpurun_out/kt_$tag
