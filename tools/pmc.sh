#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <tag> "<counters>" -- <command...>
# runs rocprofv3 --pmc in its own pass (kernel-trace only) and prints per-kernel averages
tag=$1; ctrs=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc $ctrs -d gpurun_out/pmc_$tag -o p -f csv -- "$@" > gpurun_out/pmc_$tag/run.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_$tag
