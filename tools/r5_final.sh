#!/bin/bash
# evidence of the round's last state (GPU box, from the repo root): the default bench line and the kernel trace of the same command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r5f_bench.json 2> gpurun_out/r5f_bench.err
tail -c 300 gpurun_out/r5f_bench.json
KT_LINES=22 KT_TAIL=1 bash tools/kt.sh r5f_bench python bench.py --no-cpu --traffic static > /dev/null
cat gpurun_out/kt_r5f_bench.txt | cut -c1-200 | head -8
