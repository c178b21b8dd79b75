#!/bin/bash
# round 4: full GPU suite, smoke, default bench line (as the driver runs them)
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/r4_full_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_full_tests.log
tail -8 gpurun_out/r4_full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_full_smoke.log 2>&1
tail -3 gpurun_out/r4_full_smoke.log
( time timeout 900 python bench.py ) > gpurun_out/r4_full_bench.log 2>&1
tail -4 gpurun_out/r4_full_bench.log
