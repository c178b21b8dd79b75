#!/bin/bash
# round 4, final evidence run: full GPU suite, smoke, the default bench line (as the driver runs it) and the kernel trace of the same command
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/r4_final_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_final_tests.log
tail -6 gpurun_out/r4_final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 900 python bench.py ) > gpurun_out/r4_final_bench.log 2>&1
grep '^{' gpurun_out/r4_final_bench.log > gpurun_out/r4_final_bench.json
tail -4 gpurun_out/r4_final_bench.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r4f
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4f -o f -- python bench.py --no-cpu --traffic static > gpurun_out/prof_r4f.json 2> gpurun_out/prof_r4f.err
python tools/rocpd_summary.py $(find gpurun_out/prof_r4f -name "*results.db" | head -1) | head -16 > gpurun_out/prof_r4f_summary.txt
cat gpurun_out/prof_r4f_summary.txt | cut -c1-200
