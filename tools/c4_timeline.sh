cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace -d gpurun_out/tl -o f -- env C4_STEPS=${C4_STEPS:-3} python tools/c4_step.py 512 > gpurun_out/tl.out 2> gpurun_out/tl.err
db=$(find gpurun_out/tl -name "*results.db" | head -1)
python -c "
import sqlite3; db=sqlite3.connect('$db'); print([r[1] for r in db.execute('pragma table_info(kernels)')])"
python tools/rocpd_timeline.py $db 3.2 | tee gpurun_out/c4_timeline.txt
tail -5 gpurun_out/tl.out
rm -rf gpurun_out/tl
