#!/usr/bin/env python3
'''Summarise a rocprofv3 rocpd database (kernel-trace --stats run) into the per-kernel
table that gets committed under profiles/.  usage: rocpd_summary.py results.db [out.md]'''
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), '
                  'max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc').fetchall()
total = sum(r[2] for r in rows) or 1
lines = ['| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch B | grid | wg |', '|---|---|---|---|---|---|---|---|---|---|---|---|---|---|']
for r in rows:
    name = r[0] if len(r[0]) < 90 else r[0][:87] + '...'
    lines.append(f'| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.1f} | {r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} | {100 * r[2] / total:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} | {r[12]} |')
out = '\n'.join(lines) + '\n'
if len(sys.argv) > 2:
    open(sys.argv[2], 'a').write(out)
print(out)
