#!/bin/bash
# round 4: kernel trace of the any-mesh path on 96^3 trilinear elasticity (tools/generic_probe.py row)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r4_el
GENERIC_PROBE_ONLY="3D P1 elasticity" rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4_el -o e -- python tools/generic_probe.py > gpurun_out/prof_r4_el.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_r4_el/e_results.db | head -8 | cut -c1-200
