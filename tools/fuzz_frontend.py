#!/usr/bin/env python3
'''GPU box: differential fuzzing of the assembly path.  Random integrals built at the level the seam builds them (function.Integrand: test / trial
argument, form tensor B [nct][S][ncr][S] / linear form L / constant, sample, geometry) on random structured meshes -- dimension 1-3, std / spline bases of
degree 1-4, periodic axes, scalar and vector valued, rectilinear / graded / isoparametric geometry, volume samples and boundary sides, several terms and
several samples per integral -- evaluated through the C ABI (function.eval) and by the CPU evaluator tests/af_oracle.py (the checker of the plan tests).
Index arrays must be equal, values within 1e-12 of the largest entry.   python tools/fuzz_frontend.py [ncases] [seed]   (FUZZ_SECONDS=<s>: time budget, FUZZ_HUGE=<fraction> of meshes past the executor's size thresholds)'''
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
    if os.environ.get('FUZZ_HUGE') and rng.random() < float(os.environ['FUZZ_HUGE']):  # past the size thresholds of the executor (coloured launches, owner-side reduction)
        shape = [int(rng.integers(lo, hi + 1)) for lo, hi in {1: [(5000, 9000)], 2: [(64, 90)] * 2, 3: [(16, 21)] * 3}[nd]]
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
    wbasis = basis
    unstructured = None
    if not periodic and gkind in ('unit', 'scaled', 'graded') and rng.random() < .3:
        # the same functions as a PLAIN basis on an element list (the form in which hierarchical / imported bases arrive): uniform, or RAGGED -- some
        # elements carry extra functions (copies of their own with new dof numbers) --, optionally made rational
        from nutils_amd import topology, basis as _basis, sample as _sample
        origin, size = geom.element_boxes()
        topo = topology.ElementList(origin, size)
        ne = len(origin)
        coeffs = [numpy.array(basis.get_coefficients(e)) for e in range(ne)]
        dofs = [numpy.array(basis.get_dofs(e)) for e in range(ne)]
        ndofs = len(basis)
        if rng.random() < .6:
            for e in range(ne):
                for _ in range(int(rng.integers(0, 3))):
                    k = int(rng.integers(len(dofs[e])))
                    coeffs[e] = numpy.concatenate([coeffs[e], coeffs[e][k:k + 1] * rng.normal()])
                    dofs[e] = numpy.concatenate([dofs[e], [ndofs]])
                    ndofs += 1
        pb = _basis.PlainBasis(coeffs, dofs, ndofs, nd)
        if rng.random() < .3:
            pb = _basis.RationalBasis(pb, .5 + rng.random(ndofs))
        unstructured = dict(topo=topo, Sample=_sample.Sample)
        basis = wbasis = pb
        geom = topo.geom
        domain = None
    ncomp = int(rng.choice([1, 1, nd, 2]))
    kind = str(rng.choice(['matrix', 'matrix', 'vector', 'scalar']))
    S = 1 + nd
    test, trial = function.Arg(basis, ncomp, 'v'), function.Arg(basis, ncomp, 'u')
    sides = ['left', 'right', 'bottom', 'top', 'front', 'back'][:2 * nd]
    from nutils_amd import points as _points
    if unstructured is not None:
        topo = unstructured['topo']
        samples = [unstructured['Sample'](topo, _points.gauss(int(rng.integers(1, 2 * degree + 3)), nd))]
        if rng.random() < .4 and topo.nelems > 1:  # a subset of the elements (in list order), also as a boundary side
            el = numpy.sort(rng.choice(topo.nelems, size=int(rng.integers(1, topo.nelems)), replace=False))
            bnd = int(rng.integers(-1, nd)) if nd > 1 else -1
            pts = _points.gauss(int(rng.integers(1, 2 * degree + 2)), nd)
            if bnd >= 0:  # points on the face xi_bnd = 0 or 1
                face = _points.gauss(int(rng.integers(1, 2 * degree + 2)), nd - 1)
                coords = numpy.insert(face.coords, bnd, float(rng.integers(0, 2)), axis=1)
                pts = _points.Points(coords, face.weights)
            samples.append(unstructured['Sample'](topo, pts, elist=el, bnd_axis=bnd))
            if rng.random() < .3:
                samples = samples[1:]
    else:
        samples = [domain.sample('gauss', int(rng.integers(1, 2 * degree + 3)))]
        if rng.random() < .2 and len(domain) > 1:  # element subset of a structured topology
            from nutils_amd import sample as _sample
            el = numpy.sort(rng.choice(len(domain), size=int(rng.integers(1, len(domain))), replace=False))
            samples.append(_sample.Sample(domain, _points.gauss(int(rng.integers(1, 2 * degree + 2)), nd), elist=el))
    if unstructured is None and nd > 1 and rng.random() < .4 and not periodic:
        for side in rng.choice(sides, size=int(rng.integers(1, 3)), replace=False):
            samples.append(domain.boundary[str(side)].sample('gauss', int(rng.integers(1, 2 * degree + 2))))
        if rng.random() < .3:
            samples = samples[1:]  # boundary terms only
    warg = function.Arg(wbasis, 1, 'w')
    terms = []
    for smp in samples:
        for _ in range(int(rng.integers(1, 3))):
            pat = str(rng.choice(['full', 'mass', 'stiff', 'block', 'sparse', 'laplace', 'laplace', 'elastic']))
            B = rng.normal(size=(ncomp, S, ncomp, S))
            if pat == 'laplace':  # kappa grad . grad + mu value value per component: the forms the structured / closed-form kernels recognise
                m, k = (float(rng.normal()) if rng.random() < .5 else 0.), float(rng.normal())
                B = numpy.einsum('cd,ab->cadb', numpy.eye(ncomp), numpy.diag([m] + [k] * nd))
            elif pat == 'elastic' and ncomp == nd:  # lambda div div + 2 mu eps : eps
                lam, mu = rng.random(2) + .1
                B = numpy.zeros((ncomp, S, ncomp, S))
                G = lam * numpy.einsum('ca,db->cadb', numpy.eye(nd), numpy.eye(nd)) + mu * (numpy.einsum('cd,ab->cadb', numpy.eye(nd), numpy.eye(nd)) + numpy.einsum('cb,ad->cadb', numpy.eye(nd), numpy.eye(nd)))
                B[:, 1:, :, 1:] = G
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
            extra = {}
            if rng.random() < .25:  # coefficient given by its values at the points of the sample
                extra['scale'] = function.PointTable(.5 + rng.random((smp.nlist, smp.points.npoints)))
            if rng.random() < .25:  # polynomial of the values of a scalar field
                mono = {(int(rng.integers(1, 4)),): float(rng.normal())}
                if rng.random() < .5:
                    mono[(0,)] = float(rng.normal())
                extra['fscale'] = function.FieldPoly([warg], mono)
            if extra:
                itg = itg._copy(**extra)
            terms.append((smp, itg, fac))
    nd_ = basis.ndofs if hasattr(basis, 'ndofs') else len(basis)
    args = dict(u=rng.normal(size=(nd_, ncomp) if ncomp > 1 else nd_), v=rng.normal(size=(nd_, ncomp) if ncomp > 1 else nd_), w=rng.normal(size=nd_))
    desc = f'nd={nd} shape={shape} {btype}{degree} periodic={periodic} geom={gkind} ncomp={ncomp} {kind} samples={[(s.nlist, s.points.npoints, s.bnd_axis) for s in samples]} terms={len(terms)}'
    integral = function.Integral(terms)
    if kind == 'vector' and ncomp == 1 and rng.random() < .3 and all(itg.fscale is not None for _, itg, _ in terms):
        # Jacobian of a residual whose coefficient depends on a field: the product-rule terms of function.derivative
        try:
            integral = function.derivative(integral, 'w')
            kind = 'matrix'
            desc += ' d/dw'
        except NotImplementedError:
            pass
    desc += f' unstructured={None if unstructured is None else type(basis).__name__} extras={sum(itg.scale is not None for _, itg, _ in terms)}+{sum(itg.fscale is not None for _, itg, _ in terms)}'
    return integral, args, kind, desc


def main(ncases, seed):
    import time
    rng = numpy.random.default_rng(seed)
    bad, done, t0 = 0, 0, time.perf_counter()
    budget = float(os.environ.get('FUZZ_SECONDS', 0))  # stop after this many seconds (the summary says how many cases ran: a run cut off from outside prints none)
    for i in range(ncases):
        if budget and time.perf_counter() - t0 > budget:
            break
        done += 1
        sub = numpy.random.default_rng(rng.integers(1 << 62))
        if os.environ.get('FUZZ_ONLY') and i != int(os.environ['FUZZ_ONLY']):  # (replay of one case of a run: FUZZ_ONLY=<case>, same count and seed)
            continue
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
                    # (a vector whose entries all cancel -- a gradient form on a periodic mesh -- is rounding residue of its terms in the reference too: the error is
                    # measured against the result, but not against less than 1e-4 of the largest term coefficient)
                    coef = max([abs(fac) * max([float(numpy.abs(numpy.asarray(getattr(itg, k), dtype=float)).max()) for k in ('B', 'L', 'f0') if getattr(itg, k) is not None] + [0.])
                                for _, itg, fac in integral.terms] + [0.])
                    err = numpy.abs(numpy.asarray(out, dtype=float).reshape(r.shape) - r).max() / max(numpy.abs(r).max(), 1e-4 * coef, 1e-300)
                # (a scalar is a sum of terms of either sign: relative to the result there is no accuracy to speak of when they cancel)
                if not err < (1e-9 if kind == 'scalar' else 1e-11):
                    status = f'MISMATCH {err:.3e} (execution {rep})'
                    if os.environ.get('FUZZ_ONLY'):
                        o, r = (out[0], ref[0]) if kind == 'matrix' else (numpy.asarray(out, dtype=float).ravel(), numpy.asarray(ref, dtype=float).ravel())
                        d = numpy.flatnonzero(numpy.abs(o - r) > 1e-11 * numpy.abs(r).max())
                        print(f'  {len(d)} of {len(r)} entries differ; first: {d[:12]}; mine {o[d[:6]]}; reference {r[d[:6]]}; |ref| {numpy.abs(r).max():.3e}')
                        for _, itg, fac in integral.terms:
                            print('  term:', {k: (getattr(v, 'shape', v) if not isinstance(v, (int, float, type(None))) else v) for k, v in vars(itg).items()}, fac)
                    break
        except Exception as e:
            status = f'ERROR {type(e).__name__}: {str(e)[:200]}'
            if os.environ.get('FUZZ_TRACE'):
                traceback.print_exc()
        if status != 'ok':
            bad += 1
            print(f'case {i}: {desc}: {status}', flush=True)
    print(f'{done} of {ncases} cases in {time.perf_counter() - t0:.0f} s, {bad} not ok (seed {seed})')


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
