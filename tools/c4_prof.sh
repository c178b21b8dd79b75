cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3_c4 -o c4 -- python tools/c4_step.py 512 > gpurun_out/prof_r3_c4.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_r3_c4 2>/dev/null | head -30 || ls gpurun_out/prof_r3_c4
