#!/usr/bin/env python3
'''Timeline of the last kernels of a rocprofv3 rocpd database: python tools/timeline.py <results.db> [window_ms]
(start / end relative to the first listed event, queue, duration; memory copies need --memory-copy-trace)'''
import sys, sqlite3
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 8.
ev = [(s, e, f'q{q}', n[:70]) for s, e, q, n in db.execute('select start, end, queue_id, name from kernels')]
try:
    ev += [(s, e, 'copy', f'memcpy {sz} B') for s, e, sz in db.execute('select start, end, size from rocpd_memory_copy')]
except sqlite3.Error:
    pass
ev.sort()
t1 = ev[-1][1]
ev = [x for x in ev if x[0] >= t1 - win * 1e6]
t0 = ev[0][0]
for s, e, q, n in ev:
    print(f'{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  {q:5s} {n}')
