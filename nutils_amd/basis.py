'''Basis tables (inputs of the hot path): per-element dof lists and polynomial
coefficient tables, mirroring the accessors of the reference's ``function.Basis``
(/root/reference/src/nutils/function.py:2794-2837 ``get_dofs``/``get_coefficients``)
for the two families the assembly path meets:

* :class:`StructuredBasis` -- tensor-product spline / C0 'std' bases on
  ``mesh.rectilinear`` (function.py:3030-3100; per-axis tables from
  topology.py:2209-2312 ``basis_spline``);
* :class:`PlainBasis` -- explicit per-element tables, ragged nbasis per element
  (function.py:2881-2913), e.g. hierarchical or imported bases.

The tables are produced on the host (they are O(#distinct element classes)); the
per-element expansion -- dof maps for all elements, tabulation at the quadrature
points -- happens on the device (nh_structured_dofs, nh_poly_tabulate).
'''

import numpy

from . import poly


def _local_bsplines(lknots):
    '''The p+1 B-splines that are nonzero on the middle span [lknots[p-1], lknots[p]]
    of the 2p local knots, as polynomials in the element coordinate xi in [0,1]
    (rows: functions, columns: poly1d order).  Cox-de Boor recursion (Piegl & Tiller,
    "The NURBS book", algorithm A2.2) carried out on coefficient vectors
    (cf. topology.py:2326-2361).'''
    lknots = numpy.asarray(lknots, dtype=float)
    p = len(lknots) // 2
    P = numpy.polynomial.polynomial  # ascending-power helpers
    N = [numpy.array([1.])]
    if p:
        a, h = lknots[p - 1], lknots[p] - lknots[p - 1]
        if not h > 0:
            raise ValueError('element size should be positive')
        x = numpy.array([a, h])  # physical coordinate as polynomial in xi
        for k in range(1, p + 1):
            new = [numpy.zeros(1) for _ in range(k + 1)]
            for r in range(k):
                # N[r] is the degree k-1 function whose support starts r-(k-1) spans left of the element
                tl, tr = lknots[p + r - k], lknots[p + r]
                term = N[r] / (tr - tl)
                new[r] = P.polyadd(new[r], P.polymul(P.polysub(numpy.array([tr]), x), term))
                new[r + 1] = P.polyadd(new[r + 1], P.polymul(P.polysub(x, numpy.array([tl])), term))
            N = new
    out = numpy.zeros((p + 1, p + 1))
    for i, c in enumerate(N):
        c = numpy.concatenate([c, numpy.zeros(p + 1 - len(c))])
        out[i] = c[::-1]
    return out


def spline_axis(n, degree, continuity=-1, periodic=False):
    '''Per-axis tables for `n` uniform elements: (coeffs list of (p+1,p+1) arrays,
    start_dofs int array, ndofs).  continuity=-1 is the maximally smooth spline, 0 the
    C0 'std' basis (topology.py:2243-2312).  Open knot vector, or -- periodic
    (topology.py:2279-2291) -- the knot vector continued with the period: every
    element has the interior table, the dof ranges start_dofs[i] + 0..p wrap around
    modulo ndofs = n (p - c).'''
    p = degree
    c = continuity + p if continuity < 0 else continuity
    if not -1 <= c < max(p, 1):
        raise ValueError('invalid continuity')
    step = p - c
    if periodic and step != p + 1:
        reps = 1
        while (reps - 1) * n * step < p - step + 2:  # periods of knots needed behind the last element
            reps *= 2
        knots = numpy.repeat(numpy.arange(reps * n, dtype=float), step)
        if p > step:
            knots = numpy.concatenate([knots[-(p - step):] - reps * n, knots])
        start = step * numpy.arange(n)
        ndofs = n * step
    else:
        mult = numpy.full(n + 1, step, dtype=int)
        mult[0] = mult[-1] = p
        knots = numpy.repeat(numpy.arange(n + 1, dtype=float), mult)
        start = numpy.cumsum(mult[:n]) - mult[0]
        ndofs = int(mult[:n].sum()) + 1
    cache = {}
    coeffs = []
    for o in start:
        lk = knots[o:o + 2 * p]
        key = tuple(numpy.round((lk - lk[0]) * 1024).astype(int)) if p else ()
        if key not in cache:
            cache[key] = _local_bsplines(lk)
        coeffs.append(cache[key])
    return coeffs, start.astype(numpy.int64), ndofs


class Basis:
    '''Common accessors (function.py:2794-2837).'''

    ndofs: int
    nelems: int
    ndims: int

    def __len__(self):
        return self.ndofs

    def get_dofs(self, ielem):
        raise NotImplementedError

    def get_coefficients(self, ielem):
        raise NotImplementedError

    def get_coeffshape(self, ielem):
        return numpy.asarray(self.get_coefficients(ielem).shape[1:])

    # a basis used as an array (shape (ndofs,)) in an integrand: delegate to the operand algebra
    __array_ufunc__ = None

    def _operand(self):
        from . import function
        return function._as_operand(self)

    def __mul__(self, other):
        return self._operand() * other

    def __rmul__(self, other):
        return other * self._operand()

    def __neg__(self):
        return -self._operand()

    def grad(self, geom):
        return self._operand().grad(geom)

    def __matmul__(self, other):
        from . import function
        return function.dot_basis(self, other)


class StructuredBasis(Basis):

    def __init__(self, shape, btype, degree, periodic=()):
        if btype not in ('std', 'spline'):
            raise ValueError(f'unsupported structured basis type {btype!r}')
        self.shape = tuple(int(n) for n in shape)
        self.ndims = len(self.shape)
        self.btype, self.degree = btype, int(degree)
        self.periodic = tuple(sorted(int(i) for i in periodic))
        axes = [spline_axis(n, self.degree, 0 if btype == 'std' else -1, i in self.periodic) for i, n in enumerate(self.shape)]
        self.axis_coeffs = [a[0] for a in axes]
        self.start_dofs = [a[1] for a in axes]
        self.dofs_shape = tuple(a[2] for a in axes)
        self.nloc = (self.degree + 1,) * self.ndims
        self.nb = int(numpy.prod(self.nloc))
        self.ndofs = int(numpy.prod(self.dofs_shape))
        self.nelems = int(numpy.prod(self.shape))
        # element classes: distinct per-axis tables -> class index per axis position
        self.axis_class = []
        self.axis_tables = []
        for coeffs in self.axis_coeffs:
            uniq, cls = [], []
            for c in coeffs:
                for k, u in enumerate(uniq):
                    if u is c or numpy.array_equal(u, c):
                        cls.append(k)
                        break
                else:
                    cls.append(len(uniq))
                    uniq.append(c)
            self.axis_class.append(numpy.array(cls, dtype=numpy.int32))
            self.axis_tables.append(uniq)
        self.nclasses_axis = tuple(len(t) for t in self.axis_tables)
        self.nclasses = int(numpy.prod(self.nclasses_axis))

    def _unravel(self, ielem):
        if not 0 <= ielem < self.nelems:
            raise IndexError('element index out of range')
        return numpy.unravel_index(ielem, self.shape)

    def get_dofs(self, ielem):
        idx = self._unravel(ielem)
        dofs = numpy.zeros(1, dtype=numpy.int64)
        for start, nd, i in zip(self.start_dofs, self.dofs_shape, idx):
            rng = (start[i] + numpy.arange(self.degree + 1)) % nd
            dofs = (dofs[:, None] * nd + rng[None, :]).ravel()
        return dofs

    def class_of(self, ielem):
        idx = self._unravel(ielem)
        return int(numpy.ravel_multi_index([cls[i] for cls, i in zip(self.axis_class, idx)], self.nclasses_axis))

    def class_coefficients(self, icls):
        '''Coefficient table (nb, ncoeffs) of element class `icls`.'''
        sub = numpy.unravel_index(icls, self.nclasses_axis)
        return poly.tensor_product([tabs[k] for tabs, k in zip(self.axis_tables, sub)])

    def get_coefficients(self, ielem):
        return self.class_coefficients(self.class_of(ielem))

    def element_classes(self):
        '''int32 class index per element (last axis fastest).'''
        grids = numpy.meshgrid(*self.axis_class, indexing='ij')
        return numpy.ravel_multi_index([g.ravel() for g in grids], self.nclasses_axis).astype(numpy.int32)

    def all_class_coefficients(self):
        return numpy.concatenate([self.class_coefficients(k) for k in range(self.nclasses)], axis=0)



class PlainBasis(Basis):
    '''Explicit tables: `coefficients[e]` (nb_e, ncoeffs), `dofs[e]` (nb_e,).'''

    def __init__(self, coefficients, dofs, ndofs, ndims):
        if len(coefficients) != len(dofs):
            raise ValueError('coefficients and dofs differ in length')
        self._coeffs = [numpy.asarray(c, dtype=float) for c in coefficients]
        self._dofs = [numpy.asarray(d, dtype=numpy.int64) for d in dofs]
        for c, d in zip(self._coeffs, self._dofs):
            if c.ndim != 2 or d.ndim != 1 or len(c) != len(d):
                raise ValueError('inconsistent element tables')
        ncs = {c.shape[1] for c in self._coeffs}
        if len(ncs) > 1:
            raise NotImplementedError('mixed polynomial degrees within one basis')
        self.ncoeffs = ncs.pop() if ncs else 1
        self.ndofs, self.ndims, self.nelems = int(ndofs), int(ndims), len(dofs)
        self.offsets = numpy.cumsum([0] + [len(d) for d in self._dofs]).astype(numpy.int64)

    def get_dofs(self, ielem):
        return self._dofs[ielem]

    def get_coefficients(self, ielem):
        return self._coeffs[ielem]

    def concatenated(self):
        return (numpy.concatenate(self._dofs) if self._dofs else numpy.zeros(0, numpy.int64),
                numpy.concatenate(self._coeffs, axis=0) if self._coeffs else numpy.zeros((0, self.ncoeffs)))


class RationalBasis(Basis):
    '''NURBS-type basis  N_i = w_i B_i / W  over a polynomial parent basis (examples/platewithhole.py:71-72,85:
    ``bsplinebasis * controlweights / weightfunc``).  `weights` are the dof weights w.  The weight function W is either the
    parent combination sum_j w_j B_j itself (W=None) or supplied by the mesh producer at the points of ONE sample:
    W[nelems][nq] and dW[nelems][nq][ndims] (derivatives w.r.t. the element coordinates), e.g. a coarse-level weight function
    seen through refinement transforms.  Tables are per element (nh_rationalize on the device).'''

    def __init__(self, parent, weights, W=None, dW=None):
        weights = numpy.asarray(weights, dtype=float)
        if weights.shape != (parent.ndofs,):
            raise ValueError('one weight per dof expected')
        if (W is None) != (dW is None):
            raise ValueError('W and dW go together')
        self.parent, self.weights = parent, weights
        self.W = None if W is None else numpy.ascontiguousarray(W, dtype=float)
        self.dW = None if dW is None else numpy.ascontiguousarray(dW, dtype=float)
        self.ndofs, self.nelems, self.ndims = parent.ndofs, parent.nelems, parent.ndims

    def get_dofs(self, ielem):
        return self.parent.get_dofs(ielem)

    def get_coefficients(self, ielem):
        raise NotImplementedError('a rational basis has no polynomial coefficients; see parent.get_coefficients and weights')
