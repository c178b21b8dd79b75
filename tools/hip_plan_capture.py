#!/usr/bin/env python3
'''TEST INFRASTRUCTURE (build container only).  Sweep for the composition that cannot run in one process (the reference exists only in the
build container, the GPU only on the box): the UNMODIFIED examples run with the seam installed and the CPU evaluator tests/af_oracle.py as executor; every
distinct plan the hooks hand to the executor is written -- with the arguments of its first evaluation and the result of THE REFERENCE for the function-level
array the plan was matched from (the un-hooked nutils.function.evaluate / as_csr saved by seam.install: function.py:2427-2452) -- to a directory that
tools/hip_plan_sweep.py / tests/test_gpu_plans.py replay through seam.execute (the C ABI) on the GPU box.  The CPU evaluator's own result must agree with the
reference's to 1e-13 of the largest entry, else the capture stops.

  python tools/hip_plan_capture.py OUTDIR [example ...]'''
import os
import sys
import tempfile
import unittest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
sys.path[:0] = [os.path.join(ROOT, 'oracle', 'refshim'), REF + '/src', REF, ROOT, os.path.join(ROOT, 'tests')]
os.environ.setdefault('NUTILS_NPROCS', '1')
os.environ.setdefault('NUTILS_MATRIX', 'scipy')
import numpy  # noqa: E402
import nutils  # noqa: E402
nutils.__path__.append(os.path.join(ROOT, 'oracle', 'refshim', 'nutils_ext'))
import matplotlib  # noqa: E402
matplotlib.use('Agg')
from nutils_amd import seam  # noqa: E402
import af_oracle  # noqa: E402
import plan_exec  # noqa: E402
import nutils.testing  # noqa: E402


MEDIUM = [  # (module, tag, main arguments): sizes at which the executor takes its other paths (coloured launches, thread passes, owner-side reduction)
    ('laplace', 'spline3_64', dict(nelems=64, etype='square', btype='spline', degree=3)),
    ('laplace', 'std1_48', dict(nelems=48, etype='square', btype='std', degree=1)),
    ('laplace', 'std2_66', dict(nelems=66, etype='square', btype='std', degree=2)),
    ('elasticity', 'std2_24', dict(nelems=24, etype='square', btype='std', degree=2)),
    ('drivencavity', 'th2_10', dict(nelems=10, etype='square', degree=2, reynolds=100., compatible=False, strongbc=False)),
    ('adaptivity', 'hstd2_4', dict(etype='square', btype='h-std', degree=2, nrefine=4)),
    ('adaptivity', 'thspline2_4', dict(etype='square', btype='th-spline', degree=2, nrefine=4)),
]


def main(out, modules):
    import importlib
    out = os.path.abspath(out)
    os.makedirs(out, exist_ok=True)
    os.chdir(tempfile.mkdtemp())
    problems = []
    for name in modules:
        contents = set()
        seen, count, pcount = {}, [0], [0]  # (point-evaluation plans are numbered apart: `<example>_pts_NNN`, so that the integral plans keep their names)

        def reference(plan, arguments, nrows=None):
            '''the reference's own result for the array the plan was matched from, through its un-hooked entry points'''
            import nutils.function as rf
            st = seam._STATE
            arr = plan['_source']
            if plan['kind'] != 'matrix':
                return numpy.asarray(st['evaluate'](arr, arguments=arguments)[0], dtype=float)
            # rows = the axes of the test argument, columns = those of the trial argument (solver.py:252-258 flattens its blocks the same way)
            if arr.ndim != 2:
                arr = numpy.reshape(arr, (nrows, -1))  # (the reference's own reshape of its Array: function.py:3485)
            v, rp, ci = st['evaluate'](*st['as_csr'](arr), arguments=arguments)
            return numpy.asarray(v, dtype=float), numpy.asarray(rp), numpy.asarray(ci)

        def executor(plan, arguments):
            try:
                return checked(plan, arguments)
            except Exception as e:  # (the seam would turn it into a silent fall-back to the reference path)
                problems.append(f'{name}: {plan["kind"]} plan: {type(e).__name__}: {e}')
                raise

        def checked(plan, arguments):
            res = seam.run(plan, arguments, lambda integral, args, kind: af_oracle.evaluate(integral, args))
            ref = reference(plan, arguments, len(res[1]) - 1 if plan['kind'] == 'matrix' else None)
            if plan['kind'] == 'matrix':
                assert numpy.array_equal(res[1], ref[1]) and numpy.array_equal(res[2], ref[2]), 'index arrays differ from the reference'
                scale = numpy.abs(ref[0]).max() if len(ref[0]) else 1.
                assert not len(ref[0]) or numpy.abs(res[0] - ref[0]).max() <= 1e-13 * max(scale, 1e-300), ('values', numpy.abs(res[0] - ref[0]).max() / scale)
                expect = dict(values=ref[0], rowptr=ref[1], colidx=ref[2])
                ret = res
            elif plan['kind'] == 'points':
                ret = numpy.asarray(res, dtype=float)
                expect = dict(points=ref)
            elif plan['kind'] == 'scalar':
                ret = float(res)
                expect = dict(scalar=numpy.asarray(float(ref)))
            else:
                ret = numpy.asarray(res, dtype=float).reshape(plan['shape'])
                expect = dict(vector=ref.reshape(plan['shape']))
            if plan['kind'] in ('vector', 'scalar', 'stack'):
                # what every entry is a sum OF: the same integral with every factor replaced by its absolute value (af_oracle absolute=True) -- the scale of the rounding
                # error of an entry whatever cancels in it (a residual at its own solution, an energy difference); stored beside the result, 32 ulp of it is the floor of the comparison
                ab = seam.run(plan, arguments, lambda integral, args, kind: af_oracle.evaluate(integral, args, absolute=True))
                expect['abssum'] = numpy.asarray(ab, dtype=float).reshape(numpy.shape(expect['scalar' if plan['kind'] == 'scalar' else 'vector']))
            if plan['kind'] != 'matrix':
                err = plan_exec.compare_example(plan, ret, expect, arguments)
                if os.environ.get('CAPTURE_VERBOSE'):
                    print(f'  {name} {plan["kind"]}: |ref| {float(numpy.abs(ref).max()):.3e}, error / tolerance {err:.2e}')
            numeric = {k: numpy.asarray(v, dtype=float) for k, v in (arguments or {}).items() if numpy.asarray(v).dtype.kind in 'fiub'}
            if id(plan) not in seen:
                for k, v in numeric.items():
                    expect['arg_' + k] = v
                ctr, tag = (pcount, 'pts_') if plan['kind'] == 'points' else (count, '')
                # (an example whose tests repeat a computation hands over equal plans with equal arguments again: one fixture per content)
                digest = content_digest(plan, expect)
                if digest in contents:
                    seen[id(plan)] = dict(plan=plan, path=None, first=expect, args=numeric, later=True)
                    return ret
                contents.add(digest)
                seen[id(plan)] = dict(plan=plan, path=os.path.join(out, f'{name}_{tag}{ctr[0]:03d}.npz'), first=expect, args=numeric, later=False)  # (plan kept alive: ids are not recycled)
                seam.save(seen[id(plan)]['path'], {k: v for k, v in plan.items() if not k.startswith('_')}, expect)
                ctr[0] += 1
            else:
                # a LATER evaluation of the same plan with other arguments (a Newton iteration, the next time step): stored once, as `*2` / `arg2_*`,
                # so that the replay re-executes the built plan with new arguments
                rec = seen[id(plan)]
                same = set(numeric) == set(rec['args']) and all(numpy.array_equal(numeric[k], rec['args'][k]) for k in numeric)
                if not rec['later'] and not same and plan['kind'] != 'matrix' or not rec['later'] and not same and any(t.get('fpoly') is not None for t in plan.get('terms', [])):
                    rec['later'] = True
                    both = dict(rec['first'])
                    for k, v in expect.items():
                        both[k + '2'] = v
                    for k, v in numeric.items():
                        both['arg2_' + k] = v
                    seam.save(rec['path'], {k: v for k, v in plan.items() if not k.startswith('_')}, both)
            return ret
        if isinstance(name, tuple):  # a run of main(**kwargs) at a larger size (no embedded vectors there: the reference's result per plan is the only check)
            module, tag, kwargs = name
            name = f'{module}_{tag}'
            mod = importlib.import_module('examples.' + module)
            st = seam.install(executor)
            try:
                mod.main(**kwargs)
            finally:
                seam.uninstall()
            print(f'{name}: main({kwargs}); {count[0]} + {pcount[0]} (points) distinct plans written')
            continue
        mod = importlib.import_module('examples.' + name)
        st = seam.install(executor)
        try:
            res = unittest.TextTestRunner(verbosity=0, stream=open(os.devnull, 'w')).run(unittest.defaultTestLoader.loadTestsFromTestCase(mod.test))
        finally:
            seam.uninstall()
        print(f'{name}: {res.testsRun} reference tests, {len(res.failures)} failures, {len(res.errors)} errors; {count[0]} + {pcount[0]} (points) distinct plans written')
    for p in problems:
        print('MISMATCH', p)
    if problems:
        raise SystemExit(1)


def signature(plan):
    '''What a plan IS, independent of the order in which the reference happened to visit its terms and samples (that order follows object hashes inside the reference's
    simplifier and changes from process to process): kind, shape, derivatives, the multiset of its terms (tensor, factor, exposure, coefficient kinds) and of its tables.'''
    import json

    if plan['kind'] == 'stack':  # an array of scalar plans
        return json.dumps(['stack', [int(n) for n in plan['shape']], [None if part is None else signature(part) for part in plan['parts']]])

    def arr(x):
        return None if x is None else [round(float(v), 10) for v in numpy.asarray(x, dtype=float).ravel()]
    terms = []
    for t in plan.get('terms', []):
        k = 'B' if t.get('B') is not None else 'L' if t.get('L') is not None else 'f0'
        v = numpy.asarray(t[k], dtype=float) * float(t['fac'])
        fp = t.get('fpoly')
        terms.append(json.dumps([k, list(v.shape), arr(v), bool(t['rows']), bool(t['cols']), t.get('scale') is not None, int(t['geom']) >= 0,
                                 None if fp is None else sorted(zip(map(tuple, numpy.asarray(fp['powers']).tolist()), arr(fp['coeffs']))), len(t.get('pvars') or [])]))
    for t in plan.get('pterms', []):
        terms.append(json.dumps(['P', [int(n) for n in t['Ashape']], arr(t['A']), len(t['factors']), t.get('scale') is not None]))
    tables = sorted([json.dumps(['topo', t['kind']]) for t in plan['topos']] + [json.dumps(['basis', b['kind']]) for b in plan['bases']]
                    + [json.dumps(['geom', g['kind']]) for g in plan['geoms']]
                    + [json.dumps(['sample', len(s['weights']), int(s.get('bnd_axis', -1)), None if s.get('elist') is None else len(s['elist'])]) for s in plan['samples']])
    return json.dumps([plan['kind'], [int(n) for n in plan['shape']], list(plan.get('derivs', [])), sorted(terms), tables])


def content_digest(plan, expect):
    import hashlib
    h = hashlib.sha1(signature(plan).encode())
    for k in sorted(expect):
        h.update(k.encode())
        h.update(numpy.ascontiguousarray(numpy.asarray(expect[k])).tobytes())
    return h.hexdigest()


def check(committed, modules):
    '''--check: capture the plans of `modules` again (into a scratch directory) and compare every file with the committed fixture of the same name: the plan
    (structure and every array) and the reference's results stored beside it.  -> number of files compared; raises SystemExit on the first difference.'''
    scratch = tempfile.mkdtemp()
    main(scratch, modules)
    names = sorted(os.listdir(scratch))
    bad = []
    for f in names:
        path = os.path.join(committed, f)
        if not os.path.exists(path):
            bad.append(f'{f}: not among the committed fixtures')
            continue
        (pn, en), (po, eo) = seam.load(os.path.join(scratch, f)), seam.load(path)
        if signature(pn) != signature(po) or sorted(en) != sorted(eo):
            bad.append(f'{f}: plan structure differs')
            continue
        for k in en:  # the reference's results and the arguments they belong to
            a, b = numpy.asarray(en[k]), numpy.asarray(eo[k])
            if a.shape != b.shape or a.dtype != b.dtype:
                bad.append(f'{f}: array {k} differs in shape / type')
            elif a.dtype.kind == 'f':
                if not numpy.array_equal(numpy.isnan(a), numpy.isnan(b)) or (a.size and numpy.abs(numpy.nan_to_num(a - b)).max() > 1e-13 * max(1e-300, numpy.abs(numpy.nan_to_num(b)).max())):
                    bad.append(f'{f}: array {k} differs by more than 1e-13 of its largest entry')
            elif not numpy.array_equal(a, b):
                bad.append(f'{f}: index array {k} differs')
    committed_names = {f for f in os.listdir(committed) if any(f.startswith(m + '_') for m in modules)}
    missing = sorted(committed_names - set(names))
    bad += [f'{f}: committed, not reproduced' for f in missing]
    for b in bad:
        print('DIFFERENT', b)
    print(f'check: {len(names)} plans of {", ".join(modules)} captured again, {len(bad)} differences from {committed}')
    if bad:
        raise SystemExit(1)
    return len(names)


ALL = ['laplace', 'elasticity', 'poisson', 'platewithhole', 'adaptivity', 'cahnhilliard', 'drivencavity', 'burgers', 'finitestrain', 'cylinderflow']

if __name__ == '__main__':
    if sys.argv[1] == '--check':  # python tools/hip_plan_capture.py --check tests/golden/plans_examples [example ...]
        check(os.path.abspath(sys.argv[2]), sys.argv[3:] or ALL)
    else:
        main(sys.argv[1], MEDIUM if sys.argv[2:] == ['--medium'] else sys.argv[2:] or ALL)
