#!/usr/bin/env python3
'''GPU box: differential fuzzing of the assembly path.  Random integrals built at the level the seam builds them (function.Integrand: test / trial
argument, form tensor B [nct][S][ncr][S] / linear form L / constant, sample, geometry) on random structured meshes -- dimension 1-3, std / spline bases of
degree 1-4, periodic axes, scalar and vector valued, rectilinear / graded / isoparametric geometry, volume samples and boundary sides, several terms and
several samples per integral -- evaluated through the C ABI (function.eval) and by the CPU evaluator tests/af_oracle.py (the checker of the plan tests).
Index arrays must be equal, values within 1e-12 of the largest entry.   python tools/fuzz_frontend.py [ncases] [seed]'''
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy  # noqa: E402
from nutils_amd import mesh, function  # noqa: E402
import af_oracle  # noqa: E402


def random_case(rng):
    nd = int(rng.choice([1, 2, 2, 3, 3]))
    big = rng.random() < .15
    hi = {1: 300, 2: 40, 3: 11}[nd] if big else {1: 9, 2: 7, 3: 5}[nd]
    shape = [int(rng.integers(1, hi + 1)) for _ in range(nd)]
    btype = str(rng.choice(['std', 'spline']))
    degree = int(rng.integers(1, 4 if btype == 'std' or nd == 3 else 5))
    periodic = tuple(i for i in range(nd) if rng.random() < .15 and shape[i] > degree + 1)
    gkind = str(rng.choice(['unit', 'scaled', 'graded', 'iso'])) if not periodic else str(rng.choice(['unit', 'scaled']))
    if gkind == 'unit':
        domain, geom = mesh.rectilinear(shape, periodic=periodic)
    elif gkind == 'scaled':
        domain, geom = mesh.rectilinear([numpy.linspace(rng.normal(), rng.normal() + 1 + 3 * rng.random(), n + 1) for n in shape], periodic=periodic)
    elif gkind == 'graded':
        domain, geom = mesh.rectilinear([numpy.cumsum(numpy.concatenate([[rng.normal()], .3 + rng.random(n)])) for n in shape])
    else:
        domain, geom0 = mesh.rectilinear(shape)
        gb = domain.basis('std', degree=1)
        grid = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, nd)
        geom = gb @ (grid + rng.uniform(-.2, .2, grid.shape))
    basis = domain.basis(btype, degree=degree)
    ncomp = int(rng.choice([1, 1, nd, 2]))
    kind = str(rng.choice(['matrix', 'matrix', 'vector', 'scalar']))
    S = 1 + nd
    test, trial = function.Arg(basis, ncomp, 'v'), function.Arg(basis, ncomp, 'u')
    sides = ['left', 'right', 'bottom', 'top', 'front', 'back'][:2 * nd]
    samples = [domain.sample('gauss', int(rng.integers(1, 2 * degree + 3)))]
    if nd > 1 and rng.random() < .4 and not periodic:
        for side in rng.choice(sides, size=int(rng.integers(1, 3)), replace=False):
            samples.append(domain.boundary[str(side)].sample('gauss', int(rng.integers(1, 2 * degree + 2))))
        if rng.random() < .3:
            samples = samples[1:]  # boundary terms only
    terms = []
    for smp in samples:
        for _ in range(int(rng.integers(1, 3))):
            pat = str(rng.choice(['full', 'mass', 'stiff', 'block', 'sparse']))
            B = rng.normal(size=(ncomp, S, ncomp, S))
            if pat == 'mass':
                B[:, 1:] = 0
                B[:, :, :, 1:] = 0
            elif pat == 'stiff':
                B[:, 0] = 0
                B[:, :, :, 0] = 0
            elif pat == 'block':
                keep = rng.random((ncomp, ncomp)) < .5
                keep[int(rng.integers(ncomp)), int(rng.integers(ncomp))] = True
                B *= keep[:, None, :, None]
            elif pat == 'sparse':
                B *= rng.random(B.shape) < .3
                if not B.any():
                    B[0, 0, 0, 0] = 1.
            fac = float(rng.normal())
            if kind == 'matrix':
                itg = function.Integrand(test=test, trial=trial, B=B, geom=geom, measure=geom, rows=True, cols=True)
            elif kind == 'vector':
                if rng.random() < .5:
                    itg = function.Integrand(test=test, trial=trial, B=B, geom=geom, measure=geom, rows=True, cols=False)
                else:
                    itg = function.Integrand(test=test, L=B[:, :, 0, 0].copy(), geom=geom, measure=geom, rows=True, cols=False)
            else:
                itg = function.Integrand(test=test, trial=trial, B=B, geom=geom, measure=geom, rows=False, cols=False)
            terms.append((smp, itg, fac))
    nd_ = len(basis)
    args = dict(u=rng.normal(size=(nd_, ncomp) if ncomp > 1 else nd_), v=rng.normal(size=(nd_, ncomp) if ncomp > 1 else nd_))
    desc = f'nd={nd} shape={shape} {btype}{degree} periodic={periodic} geom={gkind} ncomp={ncomp} {kind} samples={[(s.nlist, s.points.npoints, s.bnd_axis) for s in samples]} terms={len(terms)}'
    return function.Integral(terms), args, kind, desc


def main(ncases, seed):
    rng = numpy.random.default_rng(seed)
    bad = 0
    for i in range(ncases):
        sub = numpy.random.default_rng(rng.integers(1 << 62))
        try:
            integral, args, kind, desc = random_case(sub)
        except Exception as e:
            print(f'case {i}: generator: {type(e).__name__}: {e}')
            continue
        try:
            ref = af_oracle.evaluate(integral, args)
        except Exception as e:
            print(f'case {i}: {desc}: CPU evaluator: {type(e).__name__}: {str(e)[:150]}')
            continue
        try:
            status = 'ok'
            for rep in range(2):  # (a re-assembly may take another path)
                out = function.eval(function.as_csr(integral), args) if kind == 'matrix' else function.eval(integral, args)
                if kind == 'matrix':
                    if not (numpy.array_equal(out[1], ref[1]) and numpy.array_equal(out[2], ref[2])):
                        status = f'PATTERN differs (nnz {len(out[2])} / {len(ref[2])})'
                        break
                    err = numpy.abs(out[0] - ref[0]).max() / max(numpy.abs(ref[0]).max(), 1e-300)
                else:
                    r = numpy.asarray(ref, dtype=float)
                    err = numpy.abs(numpy.asarray(out, dtype=float).reshape(r.shape) - r).max() / max(numpy.abs(r).max(), 1e-300)
                if not err < 1e-11:
                    status = f'MISMATCH {err:.3e} (execution {rep})'
                    break
        except Exception as e:
            status = f'ERROR {type(e).__name__}: {str(e)[:200]}'
            if os.environ.get('FUZZ_TRACE'):
                traceback.print_exc()
        if status != 'ok':
            bad += 1
            print(f'case {i}: {desc}: {status}', flush=True)
    print(f'{ncases} cases, {bad} not ok (seed {seed})')


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
