#!/usr/bin/env python3
'''Seam matcher on the REAL reference graph (build container only: imports /root/reference/src through oracle/refshim).

What a Nutils-side integration does where `function.evaluate` hands its arrays to `evaluable.compile`
(/root/reference/src/nutils/function.py:2427-2429, evaluable.py:6532): it inspects the function-level tree of every
integral -- `sample._Integral` (sample.py:944-956), `_Derivative` (function.py:1184), `_Gradient` (:1207), `_Jacobian` (:1266),
`_Wrapper` around evaluable.add / multiply / Sum / InsertAxis ... (:989, :3202-3449), `_Transpose` (:1076), `Argument` (:1030)
contracted with a `Basis` (function.field, :2598-2627) -- and, if the integrand is a constant-coefficient bilinear / linear form
in field values and first derivatives times J(geom) (optionally times a coefficient function of the coordinates), translates it
into the operand algebra of nutils_amd/function.py.  The result is an assembly PLAN for the C ABI: per-element dof / coefficient
tables of test and trial basis (Basis.get_dofs / get_coefficients), quadrature points and weights of the sample, tabulated geometry
(x, dx/dxi at the points), the form tensor C[c,a,d,b] (or L[c,a]), component counts, an optional pointwise scale and bound argument
values.  Anything the matcher does not recognise raises Unmatched: the caller falls back to the reference path.

`python tools/hip_plan.py` runs the UNMODIFIED examples/laplace.py and examples/elasticity.py, captures the integrals they hand to
`solver.System` (by wrapping System.__init__, the examples are not edited), matches them, and writes one `.npz` per plan under
tests/golden/plans/ together with the reference's own result for it; tests/test_gpu_plans.py executes every plan through the C ABI
only and compares, tests/test_plans_host.py checks the plans against what nutils_amd's own front end derives for the same problem.
'''
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
if not os.path.isdir(REF + '/src/nutils'):
    raise SystemExit('the reference is not present: plans can only be generated in the build container')
sys.path[:0] = [os.path.join(ROOT, 'oracle', 'refshim'), REF + '/src', REF, ROOT]

import numpy  # noqa: E402
from nutils import function as rf, evaluable, transform  # noqa: E402
from nutils import sample as rsample, solver as rsolver  # noqa: E402
from nutils_amd import function as af  # noqa: E402  (pure numpy: the operand algebra; no device code is touched here)

OUT = os.environ.get('NUTILS_AMD_PLAN_OUT') or os.path.join(ROOT, 'tests', 'golden', 'plans')


class Unmatched(Exception):
    pass


# ---- stand-ins that carry reference objects through nutils_amd's algebra ----------------------------------------------

class BasisRef:
    '''nutils_amd-side handle of a reference Basis (identity = the reference object)'''
    _cache = {}

    def __new__(cls, ref, ndims):
        key = id(ref)
        if key not in cls._cache:
            self = super().__new__(cls)
            self.ref, self.ndims, self.ndofs = ref, ndims, len(ref)
            cls._cache[key] = self
        return cls._cache[key]


class GeomRef(af.Geometry):
    '''nutils_amd-side handle of a reference geometry array'''
    _cache = {}

    def __new__(cls, ref):
        key = id(ref)
        if key not in cls._cache:
            self = object.__new__(cls)
            self.ref, self.ndims = ref, ref.shape[-1]
            cls._cache[key] = self
        return cls._cache[key]

    def __init__(self, ref):
        pass


class Pointwise:
    '''subtree without arguments and bases: a coefficient function of the point (evaluated by the reference at the sample points)'''

    def __init__(self, node):
        self.node = node


def _wrapper_name(node):
    f = node._lower
    return getattr(f, '__name__', None) or getattr(getattr(f, 'func', None), '__name__', repr(f))


def _children(node):
    t = type(node).__name__
    if t == '_Wrapper':
        return list(node._args)
    if t in ('_WithoutPoints', '_Transpose'):
        return [node._arg]
    if t == '_Gradient':
        return [node._func, node._geom]
    if t == '_Jacobian':
        return [node._geom]
    if t == '_Derivative':
        return [node._arg]
    if t == '_Integral':
        return [node._integrand]
    return []


def has_symbols(node):
    '''does the subtree contain an Argument or a Basis?'''
    if isinstance(node, (rf.Argument, rf.Basis)):
        return True
    return any(has_symbols(c) for c in _children(node))


def strip_broadcast(node, ndim):
    '''undo numpy.broadcast_to wrappers (InsertAxis / transposes) around an array of rank `ndim`'''
    while getattr(node, 'ndim', 0) > ndim:
        t = type(node).__name__
        if t == '_Transpose' or (t == '_Wrapper' and _wrapper_name(node) in ('InsertAxis',)):
            node = _children(node)[0]
        else:
            break
    return node


def _field(node, ndims):
    '''function.field(name, basis, shape=...) (function.py:2624-2627): Sum(multiply(transposed/appended Argument, basis))'''
    if type(node).__name__ != '_Wrapper' or _wrapper_name(node) != 'Sum':
        return None
    prod = node._args[0]
    if type(prod).__name__ != '_Wrapper' or _wrapper_name(prod) != 'multiply':
        return None
    a, b = prod._args
    arg = strip_to_argument(a)
    basis = strip_to_basis(b)
    if arg is None or basis is None:
        arg, basis = strip_to_argument(b), strip_to_basis(a)
    if arg is None or basis is None or arg.shape[0] != len(basis) or len(arg.shape) > 2:
        return None
    ncomp = arg.shape[1] if len(arg.shape) == 2 else 1
    return af.field(arg.name, BasisRef(basis, ndims), (ncomp,) if len(arg.shape) == 2 else ())


def strip_to_argument(node):
    while not isinstance(node, rf.Argument):
        t = type(node).__name__
        if t == '_Transpose' or (t == '_Wrapper' and _wrapper_name(node) == 'InsertAxis'):
            node = _children(node)[0]
        else:
            return None
    return node


def strip_to_basis(node):
    while not isinstance(node, rf.Basis):
        t = type(node).__name__
        if t == '_Transpose' or (t == '_Wrapper' and _wrapper_name(node) == 'InsertAxis'):
            node = _children(node)[0]
        else:
            return None
    return node


# ---- axis operations on operands / integrands (free axes lead, `keep` trailing axes are the form slots) ----------------

def _tensor_keep(x):
    if isinstance(x, af.Operand):
        return x.P, 2
    return x._tensor, x._keep


def _with_tensor(x, t):
    if isinstance(x, af.Operand):
        return af.Operand(x.arg, t, x.geom)
    return x._set(t)


def _distribute(op, x, *a):
    if isinstance(x, af.IntegrandSum):
        return af.IntegrandSum([op(t, *a) for t in x.terms])
    return None


def insert_axis(x, length):
    d = _distribute(insert_axis, x, length)
    if d is not None:
        return d
    if isinstance(x, (af.Measure, Pointwise)):
        return x
    if isinstance(x, numpy.ndarray):
        return numpy.array(numpy.broadcast_to(x[..., None], x.shape + (length,)))
    t, keep = _tensor_keep(x)
    nfree = t.ndim - keep
    t = numpy.broadcast_to(numpy.expand_dims(t, nfree), t.shape[:nfree] + (length,) + t.shape[nfree:])
    return _with_tensor(x, numpy.array(t))


def transpose(x, axes):
    d = _distribute(transpose, x, axes)
    if d is not None:
        return d
    if isinstance(x, (af.Measure, Pointwise)):
        return x
    if isinstance(x, numpy.ndarray):
        return numpy.transpose(x, axes)
    t, keep = _tensor_keep(x)
    nfree = t.ndim - keep
    if len(axes) != nfree:
        raise Unmatched(f'transpose of rank {len(axes)} on {nfree} free axes')
    return _with_tensor(x, numpy.transpose(t, tuple(axes) + tuple(range(nfree, t.ndim))))


def take_diag(x):
    '''evaluable.TakeDiag: the diagonal of the last two free axes becomes the last free axis'''
    d = _distribute(take_diag, x)
    if d is not None:
        return d
    if isinstance(x, numpy.ndarray):
        return numpy.diagonal(x, axis1=-2, axis2=-1)
    t, keep = _tensor_keep(x)
    nfree = t.ndim - keep
    d = numpy.diagonal(t, axis1=nfree - 2, axis2=nfree - 1)  # diagonal axis goes last
    return _with_tensor(x, numpy.ascontiguousarray(numpy.moveaxis(d, -1, nfree - 2)))


# ---- the walk ------------------------------------------------------------------------------------------------------------

class Walker:

    def __init__(self, sample):
        self.sample = sample
        self.ndims = sample.ndims if hasattr(sample, 'ndims') else None

    def conv(self, node):
        t = type(node).__name__
        if not has_symbols(node):
            core = strip_broadcast(node, 0)  # a scalar broadcast over the free axes (numpy.broadcast_to wrappers): the scalar itself
            if core is not node and core.ndim == 0:
                node, t = core, type(core).__name__
            if t == '_Jacobian':
                geom = strip_broadcast(node._geom, 1)
                m = af.Measure(GeomRef(geom))
                m.tip_dim = node._tip_dim
                return m
            if not node.spaces:  # a constant
                return numpy.asarray(rf.eval(node), dtype=float)
            return Pointwise(node)
        f = _field(node, self.nd)
        if f is not None:
            return f
        if t == '_Gradient':
            func = self.conv(node._func)
            if not isinstance(func, af.Operand):
                raise Unmatched('gradient of a non-linear expression')
            geom = GeomRef(strip_broadcast(node._geom, 1))
            return func.grad(geom)
        if t == '_Transpose':
            return transpose(self.conv(node._arg), node._axes)
        if t == '_Wrapper':
            name = _wrapper_name(node)
            args = node._args
            if name == 'multiply':
                return self.mul(self.conv(args[0]), self.conv(args[1]))
            if name == 'add':
                return self.conv(args[0]) + self.conv(args[1])
            if name == 'subtract':
                return self.conv(args[0]) - self.conv(args[1])
            if name == 'negative':
                return -self.conv(args[0])
            if name == 'divide':
                den = self.conv(args[1])
                if not isinstance(den, numpy.ndarray):
                    raise Unmatched('division by a non-constant')
                return self.mul(self.conv(args[0]), 1. / den)
            if name == 'Sum':
                return self.conv(args[0]).sum(-1)
            if name == 'InsertAxis':
                length = int(rf.eval(args[1]._arg)) if type(args[1]).__name__ == '_WithoutPoints' else int(rf.eval(args[1]))
                return insert_axis(self.conv(args[0]), length)
            if name == 'astype':
                return self.conv(args[0])
            if name == 'TakeDiag':
                return take_diag(self.conv(args[0]))
            if name == 'power':
                base, exp = self.conv(args[0]), self.conv(args[1])
                if isinstance(exp, numpy.ndarray) and numpy.all(exp == 2) and isinstance(base, af.Operand):
                    return base * base
                raise Unmatched('power')
            raise Unmatched(f'operation {name}')
        raise Unmatched(f'node {t}')

    def mul(self, a, b):
        if isinstance(a, af.IntegrandSum):
            return af.IntegrandSum([self.mul(t, b) for t in a.terms])
        if isinstance(b, af.IntegrandSum):
            return af.IntegrandSum([self.mul(a, t) for t in b.terms])
        if isinstance(a, Pointwise) or isinstance(b, Pointwise):
            pw, other = (a, b) if isinstance(a, Pointwise) else (b, a)
            if isinstance(other, Pointwise):
                return Pointwise(pw.node * other.node)
            if isinstance(other, numpy.ndarray):
                return Pointwise(pw.node * rf.Array.cast(other))
            if pw.node.ndim:
                raise Unmatched('array-valued coefficient function')
            itg = af._as_integrand(other) if not isinstance(other, af.Measure) else None
            if itg is None:
                raise Unmatched('coefficient function times a bare measure')
            if itg.scale is not None:
                return itg._copy(scale=Pointwise(itg.scale.node * pw.node))
            return itg._copy(scale=pw)
        if isinstance(a, numpy.ndarray) and isinstance(b, numpy.ndarray):
            return a * b
        if isinstance(a, numpy.ndarray):
            a, b = b, a
        if isinstance(b, numpy.ndarray) and isinstance(a, af.Measure):
            raise Unmatched('constant times a bare measure')
        if isinstance(a, af.Measure):
            a, b = b, a
        if isinstance(b, af.Measure):
            itg = af._as_integrand(a)
            out = itg.with_measure(b.geom)
            out.tip_dim = getattr(b, 'tip_dim', None)
            return out
        return a * b

    def integral(self, node):
        '''sum of integrals / derivatives of integrals -> list of (walker sample, af.Integral)'''
        t = type(node).__name__
        if t == '_Integral':
            self.sample = node._sample
            self.nd = node._sample.ndims + (1 if _is_boundary(node._sample) else 0)
            itg = self.conv(node._integrand)
            if isinstance(itg, af.IntegrandSum):
                terms = itg.terms
            else:
                terms = [af._as_integrand(itg)]
            return af.Integral([(node._sample, term, 1.) for term in terms])
        if t == '_Derivative':
            return af.derivative(self.integral(node._arg), node._var.name)
        if t == '_Wrapper':
            name = _wrapper_name(node)
            if name == 'add':
                return self.integral(node._args[0]) + self.integral(node._args[1])
            if name == 'subtract':
                return self.integral(node._args[0]) - self.integral(node._args[1])
            if name == 'negative':
                return -self.integral(node._args[0])
            if name == 'multiply':
                for i in (0, 1):
                    if not has_symbols(node._args[i]) and not node._args[i].spaces and not _contains_integral(node._args[i]):
                        return self.integral(node._args[1 - i]) * float(rf.eval(node._args[i]))
        raise Unmatched(f'integral-level node {t}')


def _contains_integral(node):
    return type(node).__name__ == '_Integral' or any(_contains_integral(c) for c in _children(node))


def _is_boundary(smp):
    tr = smp.transforms[0]
    return tr.fromdims < tr.todims


def match(array):
    '''function-level array (sum of integrals, possibly differentiated) -> nutils_amd Integral over the reference's samples'''
    return Walker(None).integral(array)


# ---- plan emission ---------------------------------------------------------------------------------------------------------

def _basis_tables(basis, ielems):
    dofs = [numpy.asarray(basis.get_dofs(int(e)), dtype=numpy.int64) for e in ielems]
    coeffs = [numpy.asarray(basis.get_coefficients(int(e)), dtype=float) for e in ielems]
    nbs = {len(d) for d in dofs}
    if len(nbs) != 1 or len({c.shape for c in coeffs}) != 1:
        raise Unmatched('ragged basis in a plan (emit offsets)')
    return numpy.stack(dofs), numpy.stack(coeffs)


def _sample_tables(smp, domain_transforms):
    '''parent element index, quadrature points in PARENT element coordinates (one table: all elements alike) and weights'''
    pts = smp.points
    p0 = pts[0]
    for i in range(1, len(pts)):
        if pts[i] != p0 and not (numpy.array_equal(pts[i].coords, p0.coords) and numpy.array_equal(pts[i].weights, p0.weights)):
            raise Unmatched('elements with different quadrature tables')
    tr = smp.transforms[0]
    ielems, coords, axis = [], None, -1
    for i in range(len(tr)):
        ie, tail = domain_transforms.index_with_tail(tr[i])
        c = transform.apply(tail, numpy.asarray(p0.coords, dtype=float))
        if coords is None:
            coords = c
        elif not numpy.allclose(c, coords, atol=1e-14):
            raise Unmatched('faces of different orientation in one sample')
        ielems.append(ie)
    if tr.fromdims < tr.todims:
        const = [a for a in range(coords.shape[1]) if numpy.ptp(coords[:, a]) == 0 and coords[0, a] in (0., 1.)]
        if len(const) != 1:
            raise Unmatched('cannot identify the face axis')
        axis = const[0]
    return numpy.array(ielems, dtype=numpy.int64), coords, numpy.asarray(p0.weights, dtype=float), axis


def emit(name, kind, ref_array, arguments, domain, expect):
    '''match `ref_array` (matrix: two exposed dof axes; vector: one) and write tests/golden/plans/<name>.npz'''
    integral = match(ref_array)
    space = domain.space
    terms = []
    for smp, itg, fac in integral.terms:
        if itg.measure is None:
            raise Unmatched('term without J(geom)')
        geom = itg.measure.ref
        nd = domain.ndims
        ielems, points, weights, bnd_axis = _sample_tables(smp, domain.transforms)
        nl, nq = len(ielems), len(weights)
        xi = rf.transforms_coords(space, domain.transforms)  # coordinates in the parent element (function.py:1162-1181)
        x = numpy.asarray(smp.eval(geom)).reshape(nl, nq, nd)
        jac = numpy.asarray(smp.eval(rf.grad(geom, xi))).reshape(nl, nq, nd, nd)  # dx/dxi
        term = dict(ielems=ielems, points=points, weights=weights, bnd_axis=bnd_axis, x=x, jac=jac, fac=float(fac))
        for side, arg, exposed in (('test', itg.test, itg.rows), ('trial', itg.trial, itg.cols)):
            if arg is None:
                continue
            d, c = _basis_tables(arg.basis.ref, ielems)
            term[side + '_dofs'], term[side + '_coeffs'] = d, c
            term[side + '_ncomp'], term[side + '_ndofs'] = arg.ncomp, arg.basis.ndofs
            term[side + '_exposed'] = bool(exposed)
            if not exposed:
                if arg.name not in arguments:
                    raise Unmatched(f'argument {arg.name} is bound but has no value')
                term[side + '_value'] = numpy.asarray(arguments[arg.name], dtype=float).reshape(arg.basis.ndofs, arg.ncomp)
        if itg.B is not None:
            term['B'] = numpy.asarray(itg.B, dtype=float)
        if itg.L is not None:
            term['L'] = numpy.asarray(itg.L, dtype=float)
        if itg.f0 is not None:
            term['f0'] = numpy.asarray(itg.f0, dtype=float)
        if itg.scale is not None:
            term['scale'] = numpy.asarray(smp.eval(itg.scale.node), dtype=float).reshape(nl, nq)
        if itg.fscale is not None or itg.qform is not None or itg.qscalar is not None:
            raise Unmatched('field-dependent coefficients are not emitted as plans yet')
        terms.append(term)
    out = dict(kind=kind, nterms=len(terms), ndims=domain.ndims)
    for i, term in enumerate(terms):
        for k, v in term.items():
            out[f't{i}_{k}'] = v
    for k, v in expect.items():
        out['expect_' + k] = v
    os.makedirs(OUT, exist_ok=True)
    numpy.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    nb = {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and k.startswith('t0_') and ('B' in k or 'L' in k)}
    print(f'{name}: {kind}, {len(terms)} term(s), {nb}')
    return out


# ---- capture of the integrals of the unmodified examples ------------------------------------------------------------------

def run_example(module, **kwargs):
    '''run examples.<module>.main(**kwargs) unmodified and return the (residual / functional, trial, test) triples it hands to System'''
    import importlib
    captured = []
    orig = rsolver.System.__init__

    def spy(self, residual, /, trial, test=None):
        captured.append((residual, trial, test))
        orig(self, residual, trial=trial, test=test)
    rsolver.System.__init__ = spy
    try:
        import matplotlib
        matplotlib.use('Agg')
        import tempfile
        cwd = os.getcwd()
        os.chdir(tempfile.mkdtemp())
        mod = importlib.import_module('examples.' + module)
        mod.System = rsolver.System  # (the example imported the class by name: same object, patched in place)
        result = mod.main(**kwargs)
        os.chdir(cwd)
    finally:
        rsolver.System.__init__ = orig
    return captured, result


def _domain_of(array):
    for node in _walk(array):
        if type(node).__name__ == '_Integral':
            smp = node._sample
            tr = smp.transforms[0]
            if tr.fromdims == tr.todims:
                return smp
    raise Unmatched('no volume integral')


def _walk(node):
    yield node
    for c in _children(node):
        yield from _walk(c)


class _Domain:
    '''what emit() needs of the topology: space name, dimension and the transforms the bases are indexed by'''

    def __init__(self, array):
        smp = _domain_of(array)
        self.space = smp.spaces[0]
        self.ndims = smp.ndims
        self.transforms = smp.transforms[0]


def csr_of(array, arguments):
    n0, n1 = array.shape[0] * int(numpy.prod(array.shape[1:array.ndim // 2])), None
    flat = rf.Array.cast(numpy.reshape(array, (int(numpy.prod(array.shape[:array.ndim // 2])), -1)))
    values, rowptr, colidx = rf.eval(rf.as_csr(flat), arguments)
    return dict(values=values, rowptr=rowptr, colidx=colidx)


def main():
    # examples/laplace.py, unmodified: residual functional of the second System; std p=1 and spline p=2
    for tag, kw in (('laplace_std1', dict(nelems=4, etype='square', btype='std', degree=1)), ('laplace_spline2', dict(nelems=5, etype='square', btype='spline', degree=2))):
        captured, (cons, lhs, err) = run_example('laplace', **kw)
        res = captured[1][0]
        dom = _Domain(res)
        args = dict(u=lhs, v=numpy.zeros_like(lhs))
        dres = rf.derivative(res, 'v')
        jac = rf.derivative(dres, 'u')
        emit(tag + '_matrix', 'matrix', jac, args, dom, csr_of(jac, args))
        # the residual at a non-solution: volume term + the Neumann boundary term with its coefficient function of x
        u = numpy.random.default_rng(1).normal(size=lhs.shape)
        args = dict(u=u, v=numpy.zeros_like(u))
        emit(tag + '_residual', 'vector', dres, args, dom, dict(vector=numpy.asarray(rf.eval(dres, args))))
    # examples/elasticity.py, unmodified: energy functional of the first equilibrium System; P1 and P2 vector fields
    for tag, kw in (('elasticity_p1', dict(nelems=4, etype='square', btype='std', degree=1)), ('elasticity_p2', dict(nelems=3, etype='square', btype='std', degree=2))):
        captured, (cons, eargs) = run_example('elasticity', **kw)
        energy = captured[1][0]
        dom = _Domain(energy)
        u = numpy.random.default_rng(2).normal(size=eargs['u'].shape)
        args = dict(u=u)
        dE = rf.derivative(energy, 'u')
        H = rf.derivative(dE, 'u')
        emit(tag + '_matrix', 'matrix', H, args, dom, csr_of(H, args))
        emit(tag + '_residual', 'vector', dE, args, dom, dict(vector=numpy.asarray(rf.eval(dE, args))))


if __name__ == '__main__':
    main()
