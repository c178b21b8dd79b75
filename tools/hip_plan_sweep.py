#!/usr/bin/env python3
'''GPU box: replay the plans written by tools/hip_plan_capture.py through seam.execute (the C ABI) and compare with the stored results
(index arrays exact, values 1e-12 of the largest entry).  python tools/hip_plan_sweep.py DIR'''
import os
import sys
import traceback
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy  # noqa: E402
from nutils_amd import seam  # noqa: E402

d = sys.argv[1]
bad = 0
names = sorted(f for f in os.listdir(d) if f.endswith('.npz'))
for f in names:
    plan, expect = seam.load(os.path.join(d, f))
    args = {k[4:]: v for k, v in expect.items() if k.startswith('arg_')}
    try:
        out = seam.execute(plan, args)
        again = seam.execute(plan, args)  # (a re-assembly may take another path: owner-side reduction instead of atomics)
        for a, b in zip(out if plan['kind'] == 'matrix' else [out], again if plan['kind'] == 'matrix' else [again]):
            a, b = numpy.asarray(a, dtype=float), numpy.asarray(b, dtype=float)
            floor = 0. if plan['kind'] == 'matrix' else max([float(numpy.abs(v).max()) for v in args.values() if numpy.size(v)] + [0.])  # (residue: see below)
            if a.shape != b.shape or numpy.abs(a - b).max() > 1e-12 * max(numpy.abs(a).max(), floor, 1e-300):
                raise AssertionError(f're-execution differs: {numpy.abs(a - b).max():.3e} of {numpy.abs(a).max():.3e}')
        if plan['kind'] == 'matrix':
            ok = numpy.array_equal(out[1], expect['rowptr']) and numpy.array_equal(out[2], expect['colidx'])
            err = numpy.abs(out[0] - expect['values']).max() / max(numpy.abs(expect['values']).max(), 1e-300) if ok else numpy.inf
        elif plan['kind'] == 'vector':
            ref = expect['vector']
            err = numpy.abs(numpy.asarray(out).reshape(ref.shape) - ref).max() / max(numpy.abs(ref).max(), 1e-300)
        else:
            err = abs(float(out) - float(expect['scalar'])) / max(abs(float(expect['scalar'])), 1e-300)
        # results that are pure rounding residue in the reference (a residual at its own solution, a squared distance at the projection) have no
        # relative accuracy: the scale of a result is at least the scale of its arguments
        if err >= 1e-12 and plan['kind'] != 'matrix':
            ref = numpy.asarray(expect['vector'] if plan['kind'] == 'vector' else expect['scalar'], dtype=float)
            mine = numpy.asarray(out, dtype=float).reshape(ref.shape)
            coef = max([1.] + [abs(float(t['fac'])) * max([float(numpy.abs(numpy.asarray(t[k], dtype=float)).max()) for k in ('B', 'L', 'f0') if t.get(k) is not None] + [0.])
                               for t in plan['terms']])  # largest coefficient of a term: the terms cancel at this magnitude
            scale = coef * max([numpy.abs(ref).max()] + [float(numpy.abs(v).max()) ** (2 if plan['kind'] == 'scalar' else 1) for v in args.values() if numpy.size(v)])
            err2 = numpy.abs(mine - ref).max() / max(scale, 1e-300)
            status = 'ok' if err2 < 1e-12 else f'MISMATCH rel {err:.3e}, against the argument scale {err2:.3e} (|result| {numpy.abs(ref).max():.3e})'
        else:
            status = 'ok' if err < 1e-12 else f'MISMATCH {err:.3e}'
    except Exception as e:
        status = f'ERROR {type(e).__name__}: {str(e)[:200]}'
        if os.environ.get('SWEEP_TRACE'):
            traceback.print_exc()
    if status != 'ok':
        bad += 1
        print(f'{f}: {plan["kind"]}, {len(plan["terms"])} terms: {status}')
print(f'{len(names)} plans, {bad} not ok')
