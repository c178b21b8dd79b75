#!/usr/bin/env python3
'''Timeline of the LAST `ms` milliseconds of a rocprofv3 rocpd database (kernel-trace run): start offset, duration, stream/queue and name per dispatch.
usage: rocpd_timeline.py results.db [ms = 3.0]'''
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 3e6
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = db.execute(f'select start, end, name{", " + qcol if qcol else ""} from kernels order by start').fetchall()
t1 = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t1 - span]
t0 = rows[0][0]
for r in rows:
    print(f'{(r[0] - t0) / 1e3:9.1f} us  +{(r[1] - r[0]) / 1e3:8.1f} us  q{r[3] if qcol else "":<4} {r[2][:100]}')
