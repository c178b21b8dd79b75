'''TEST INFRASTRUCTURE (not product code): numpy evaluation of the integrals that nutils_amd's front end describes
(nutils_amd/function.py Integral / Integrand: constant forms B / L / f0, tabulated coefficients, polynomial coefficients of field
values, the product-rule tensors qform / qscalar that function.derivative produces) -- element by element, with the polynomial and
dedup routines of oracle/ (the CPU restatement of the reference's hot path).  Used to check seam plans (nutils_amd/seam.py) on the
CPU: plan -> front-end objects -> this evaluator must reproduce the reference's own result stored beside the plan.  It follows the
conventions of the reference that the oracle pins: tables D[q][m][0] = N_m, D[q][m][1+i] = dN_m/dx_i (function.py:1221-1231), flat
dof = scalar dof * ncomp + comp (function.py:2598-2627), CSR sorted with structural zeros kept and symbolically absent component
blocks pruned (evaluable.py:588-616).'''
import numpy

from oracle import assemble as oa
from nutils_amd import function as af
from nutils_amd.basis import StructuredBasis, PlainBasis, RationalBasis


def _ref_tables(basis, ie, points):
    '''N[q][m], dN[q][m][j] w.r.t. the element coordinates'''
    if isinstance(basis, RationalBasis):
        N, dN = _ref_tables(basis.parent, ie, points)
        w = basis.weights[basis.parent.get_dofs(ie)]
        if basis.W is not None:
            W, dW = basis.W[ie], basis.dW[ie]
        else:
            W, dW = N @ w, numpy.einsum('qmj,m->qj', dN, w)
        Nw, dNw = N * w, dN * w[None, :, None]
        return Nw / W[:, None], (dNw * W[:, None, None] - Nw[:, :, None] * dW[:, None, :]) / (W ** 2)[:, None, None]
    return oa.tabulate(numpy.asarray(basis.get_coefficients(ie)), points)


def _dofs(basis, ie):
    return numpy.asarray((basis.parent if isinstance(basis, RationalBasis) else basis).get_dofs(ie), dtype=numpy.int64)


class _Geo:
    def __init__(self, smp, geom):
        pts = smp.points.coords
        nl = smp.nlist
        el = numpy.arange(nl) if smp.elist is None else smp.elist
        nd = smp.ndims
        if isinstance(geom, af.TabulatedGeometry):
            self.J = geom.jac
        elif isinstance(geom, af.IsoGeometry):
            J = numpy.empty((nl, len(pts), nd, nd))
            for l, ie in enumerate(el):
                N, dN = _ref_tables(geom.basis, int(ie), pts)
                X = geom.verts[_dofs(geom.basis, int(ie))]
                J[l] = numpy.einsum('ai,qaj->qij', X, dN)
            self.J = J
        else:
            origin, size = geom.element_boxes()
            J = numpy.zeros((nl, len(pts), nd, nd))
            for i in range(nd):
                J[:, :, i, i] = size[el, i][:, None]
            self.J = J
        self.Jinv = oa.inv(self.J)
        det = numpy.abs(numpy.linalg.det(self.J))
        if smp.bnd_axis >= 0:  # surface measure of the face xi_axis = const: |det J| |J^-T e_axis|
            det = det * numpy.sqrt((self.Jinv[..., smp.bnd_axis, :] ** 2).sum(-1))
        self.wdet = det * smp.points.weights


_ABS = False  # evaluate(..., absolute=True): every factor by its absolute value -- the result bounds the sum of the |products| an entry is made of


def _a(x):
    return numpy.abs(x) if _ABS else x


def _tables(smp, basis, geo, l):
    ie = l if smp.elist is None else int(smp.elist[l])
    N, dN = _ref_tables(basis, ie, smp.points.coords)
    G = numpy.einsum('qmj,qji->qmi', _a(dN), _a(geo.Jinv[l]))
    return numpy.concatenate([_a(N)[..., None], G], axis=-1), _dofs(basis, ie)


def _field(smp, arg, geo, l, arguments):
    D, dofs = _tables(smp, arg.basis, geo, l)
    u = numpy.asarray(arguments[arg.name], dtype=float).reshape(arg.basis.ndofs, arg.ncomp)
    return numpy.einsum('qns,nc->qcs', D, _a(u[dofs]))


def evaluate_points(smp, expr, arguments=None):
    '''Sample.eval of a function.PointExpr on the CPU: [npoints, *shape], elements in list order, each with its points'''
    arguments = arguments or {}
    nq, nd = smp.points.npoints, smp.ndims
    out = numpy.zeros((smp.nlist, nq) + expr.shape)
    geos = {}
    from nutils_amd import sample as _s
    for term in expr.terms:
        sc = None if term.scale is None else term.scale().reshape(smp.nlist, nq)
        nf = len(term.factors)
        nfree = term.A.ndim - 2 * nf
        for l in range(smp.nlist):
            ops = []
            for fac in term.factors:
                geom = fac[-1] if fac[-1] is not None else _s._default_geometry(smp.topo)
                if id(geom) not in geos:
                    geos[id(geom)] = (geom, _Geo(smp, geom))
                geo = geos[id(geom)][1]
                if fac[0] == 'field':
                    ops.append(_field(smp, fac[1], geo, l, arguments))  # [q][c][s]
                else:
                    ops.append(_coords(smp, geom, l)[:, :, None])  # [q][axis][1]
            # value[q][free] = sum A[free, c0, s0, c1, s1, ...] prod_k ops_k[q][c_k][s_k]
            labels = list(range(1, 1 + term.A.ndim))
            args = [term.A, labels]
            for k, op in enumerate(ops):
                args += [op, [0, 1 + nfree + 2 * k, 2 + nfree + 2 * k]]
            val = numpy.einsum(*args, [0] + labels[:nfree]) if ops else numpy.broadcast_to(term.A, (nq,) + term.A.shape)
            out[l] += val * (1. if sc is None else sc[l].reshape((nq,) + (1,) * nfree))
    return out.reshape((smp.nlist * nq,) + expr.shape)


def _coords(smp, geom, l):
    '''x[q][axis] of list element l'''
    pts = smp.points.coords
    ie = l if smp.elist is None else int(smp.elist[l])
    if isinstance(geom, af.TabulatedGeometry):
        x = geom.x
        return x[l] if (smp.elist is not None and len(x) == smp.nlist != smp.nelems) else x[ie]
    if isinstance(geom, af.IsoGeometry):
        N, _ = _ref_tables(geom.basis, ie, pts)
        return N @ geom.verts[_dofs(geom.basis, ie)]
    origin, size = geom.element_boxes()
    return origin[ie] + size[ie] * pts


def evaluate(integral, arguments=None, absolute=False):
    '''-> float (no dof axis), array [ndofs(, ncomp)] (one), or (values, rowptr, colidx) (two dof axes; int64 index arrays); a (sample, PointExpr) pair: evaluate_points.
    absolute: the same sums with every factor replaced by its absolute value (tables, inverse Jacobians, coefficients, argument values): an upper bound of the sum of the
    |products| each entry is made of -- the scale of its rounding error whatever cancels in it (tests/plan_exec.py: the floor of vector / scalar comparisons).'''
    global _ABS
    if absolute:
        _ABS = True
        try:
            return evaluate(integral, arguments)
        finally:
            _ABS = False
    if isinstance(integral, tuple):
        return evaluate_points(integral[0], integral[1], arguments)
    arguments = arguments or {}
    kinds = {(itg.rows, itg.cols) for _, itg, _ in integral.terms}
    if len(kinds) != 1:
        raise ValueError('terms of different kinds')
    rows, cols = kinds.pop()
    geos = {}
    scalar, vec, coo = 0., None, []
    masks = {}
    if rows and cols:
        # component blocks that a sample's terms couple: the reference runs one loop per sample and concatenates the triplets (evaluable.py:6841-6895,
        # :588-616), so a block that only the boundary terms couple is structurally absent from the rows of interior elements (checked against the
        # reference: tools/hip_plan.py `mixed_blocks_*`)
        t0, r0 = integral.terms[0][1].test, integral.terms[0][1].trial
        for smp, itg, _ in integral.terms:
            m = masks.setdefault(id(smp), numpy.zeros((t0.ncomp, r0.ncomp), dtype=bool))
            m |= True if itg.qform is not None else (numpy.abs(itg.B).sum(axis=(1, 3)) != 0)
    for smp, itg, fac in integral.terms:
        mask = masks.get(id(smp))
        gkey = (id(smp), id(itg.measure))
        if gkey not in geos:
            geos[gkey] = _Geo(smp, itg.measure)
        geo = geos[gkey]
        if itg.geom is not None and itg.geom is not itg.measure:
            raise NotImplementedError('gradient geometry differs from the measure')
        sc = None if itg.scale is None else _a(itg.scale().reshape(smp.nlist, -1))
        for l in range(smp.nlist):
            w = _a(geo.wdet[l] * fac)
            if sc is not None:
                w = w * sc[l]
            if itg.fscale is not None:
                vals = [_field(smp, a, geo, l, arguments)[:, 0, 0] for a in itg.fscale.args]
                poly = 0.
                for pw, c in itg.fscale.terms.items():
                    term = _a(c)
                    for v, p in zip(vals, pw):
                        term = term * v ** p
                    poly = poly + term
                w = w * poly
            for arg, comp, slot in itg.pvars:  # point variables: components of values / gradients of bound fields as factors
                w = w * _field(smp, arg, geo, l, arguments)[:, comp, slot]
            if itg.qscalar is not None:
                Bs, at, ar = itg.qscalar
                w = w * numpy.einsum('cadb,qca,qdb->q', _a(Bs), _field(smp, at, geo, l, arguments), _field(smp, ar, geo, l, arguments))
            if itg.test is None:  # constant integrand
                scalar += float(numpy.sum(w) * _a(float(itg.f0)))
                continue
            Dt, tdofs = _tables(smp, itg.test.basis, geo, l)
            nct = itg.test.ncomp
            if rows and cols:
                Dr, rdofs = _tables(smp, itg.trial.basis, geo, l)
                ncr = itg.trial.ncomp
                nq = len(w)
                if itg.qform is None:
                    Cq = numpy.broadcast_to(_a(itg.B), (nq,) + itg.B.shape)
                elif itg.qform[0] == 'trial':
                    U = _field(smp, itg.qform[1], geo, l, arguments)
                    Cq = numpy.zeros((nq,) + itg.B.shape[:2] + (ncr, Dr.shape[-1]))
                    Cq[:, :, :, 0, 0] = numpy.einsum('cadb,qdb->qca', _a(itg.B), U)
                else:
                    U = _field(smp, itg.qform[1], geo, l, arguments)
                    Cq = numpy.einsum('ca,xydb,qxy->qcadb', _a(numpy.asarray(itg.qform[2], dtype=float)), _a(itg.B), U)
                A = numpy.einsum('q,qma,qcadb,qnb->mcnd', w, Dt, Cq, Dr)
                for c in range(nct):
                    for d in range(ncr):
                        if mask[c, d]:
                            r = numpy.repeat(tdofs * nct + c, len(rdofs))
                            k = numpy.tile(rdofs * ncr + d, len(tdofs))
                            coo.append((A[:, c, :, d].ravel(), r, k))
                continue
            if itg.B is not None:
                U = _field(smp, itg.trial, geo, l, arguments)
                F = numpy.einsum('cadb,qdb->qca', _a(itg.B), U)
            else:
                F = numpy.broadcast_to(_a(itg.L), (len(w),) + itg.L.shape)
            if rows:
                r = numpy.einsum('q,qma,qca->mc', w, Dt, F)
                if vec is None:
                    vec = numpy.zeros((itg.test.basis.ndofs, nct))
                numpy.add.at(vec, tdofs, r)
            else:
                Ut = _field(smp, itg.test, geo, l, arguments)
                scalar += float(numpy.einsum('q,qca,qca->', w, Ut, F))
    if rows and cols:
        t0, r0 = integral.terms[0][1].test, integral.terms[0][1].trial
        v = numpy.concatenate([c[0] for c in coo])
        r = numpy.concatenate([c[1] for c in coo])
        k = numpy.concatenate([c[2] for c in coo])
        return oa.dedup_csr(v, r, k, t0.basis.ndofs * t0.ncomp, r0.basis.ndofs * r0.ncomp)
    if rows:
        return vec if vec.shape[1] > 1 else vec[:, 0]
    return scalar
