'''function.factor: pre-integrate a polynomial functional of a field ONCE
into sparse Taylor coefficient tensors, then evaluate it / its derivatives per solver step with
the Monomial kernels (gather-multiply-scatter over nnz) instead of re-running the element loop.

Reference: function.factor (/root/reference/src/nutils/function.py:2630-2642) ->
evaluable.factor (evaluable.py:5785-5874) + Monomial (evaluable.py:5693-5751).  The reference
expands to arbitrary polynomial degree.  Here: c + f.u + 1/2 u.K.u with K kept as device-resident
CSR tensors (one per sample: volume and boundary terms have different patterns, exactly as the
reference keeps a sum of Monomials), plus -- round 4 -- the rank-3 and rank-4 tensors of value
polynomials  c int s(x) u^k dV  (the double-well potential of examples/cahnhilliard.py:175, a cubic
functional), BUILT on the device (nh_factor_tensor: element moments, radix sort of the flat keys,
run sums, zeros pruned -- evaluable.py:5846-5856) in the reference's own layout and evaluated per
step by nh_monomial.  Terms of degree >= 3 that carry gradient slots (kappa(u) |grad u|^2) are
refused, not truncated.'''

import numpy

from . import device, function, kernels, sample as _sample


class Factored:
    '''value(u) = c + f.u + 1/2 u.K.u'''

    def __init__(self, integral, name=None):
        if not isinstance(integral, function.Integral):
            raise TypeError('factor expects an Integral')
        names = {a.name for _, itg, _ in integral.terms for a in (itg.test, itg.trial) if a is not None and a.name is not None}
        names |= {a.name for _, itg, _ in integral.terms if itg.fscale is not None for a in itg.fscale.args if a.name is not None}  # (fields inside polynomial factors)
        if name is None:
            if len(names) != 1:
                raise NotImplementedError(f'factor needs exactly one field argument, found {sorted(names)}')
            name, = names
        for _, itg, _ in integral.terms:
            if itg.rows or itg.cols:
                raise ValueError('factor expects a scalar functional')
        self.name = name
        grad = function.derivative(integral, name)
        hess = function.derivative(grad, name)
        # the argument: a test / trial slot that carries the name, else the field of a polynomial factor (a pure value functional such as the integral of u^4)
        arg = next((a for _, itg, _ in integral.terms for a in (itg.test, itg.trial) if a is not None and a.name == name), None)
        if arg is None:
            arg = next((a for _, itg, _ in integral.terms if itg.fscale is not None for a in itg.fscale.args if a.name == name), None)
        if arg is None:
            raise NotImplementedError(f'factor: argument {name!r} occurs neither as a test / trial slot nor in a polynomial factor')
        self.arg = arg
        self.shape = (arg.basis.ndofs, arg.ncomp) if arg.ncomp > 1 else (arg.basis.ndofs,)
        self.size = int(numpy.prod(self.shape))
        zero = {name: numpy.zeros(self.shape)}
        # Taylor coefficients at u = 0, each evaluated ONCE with the element loop (evaluable.py:5846)
        self.c = float(_sample.evaluate(integral, zero))
        self.f = device.to_dev(numpy.asarray(_sample.evaluate(grad, zero)).ravel(), 'float64') if grad.terms else None
        self.K = []
        by_sample = {}
        for term in hess.terms:
            by_sample.setdefault(id(term[0]), []).append(term)
        for terms in by_sample.values():
            values, rowptr, colidx, ncols = _sample._MatrixPlan(terms).run(zero)  # (terms whose coefficient is a polynomial of the field: evaluated at u = 0)
            self.K.append((values, rowptr, colidx))
        # tensors of rank >= 3: value polynomials of the field in scalar terms (nothing of them survives in c, f, K: their derivatives vanish at u = 0)
        self.T = []  # (rank, values[nnz], indices[rank][nnz]) on the device
        for smp, itg, fac in integral.terms:
            fp = itg.fscale
            if fp is None or not fp.depends_on(name):
                continue
            i = next(j for j, a in enumerate(fp.args) if a.name == name)
            bound = sum(a is not None and a.name == name for a in (itg.test, itg.trial))
            for key, coef in fp.terms.items():
                k = key[i] + bound
                if k <= 2:
                    continue
                if bound or itg.qform is not None or itg.qscalar is not None:
                    raise NotImplementedError(f'factor: a term of degree {k} in {name!r} with gradient slots (rank-{k} tensor of a quasi-linear form)')
                farg = fp.args[i]  # (the polynomial's OWN argument: the same dof space may come through another basis object -- rational tables per sample, a restricted view)
                if sum(key) != key[i] or farg.ncomp != 1:
                    raise NotImplementedError('factor: polynomial of several fields / a vector field in a term of degree >= 3')
                if k > 4:
                    raise NotImplementedError(f'factor: degree {k} (tensors up to rank 4 are built)')
                pt = smp.tables(farg.basis)
                if not pt.nb:
                    raise NotImplementedError('factor: rank >= 3 tensors on a ragged basis')
                geom = itg.measure if itg.measure is not None else _sample._default_geometry(smp.topo)
                values, indices = kernels.factor_tensor(nelems=smp.nlist, ndims=smp.ndims, nq=smp.points.npoints, rank=k, weights=smp._weights_dev, geom=smp.geometry(geom),
                                                        basis=pt.struct, ndofs=farg.basis.ndofs, coeff=float(coef) * float(numpy.asarray(itg.f0)) * float(fac),
                                                        scale=smp.scale(itg.scale), elist=smp._elist_dev)
                if values.numel():
                    self.T.append((k, values, indices))

    def _u(self, arguments):
        if self.name not in arguments:
            raise KeyError(f'argument {self.name!r} missing')
        u = numpy.asarray(arguments[self.name], dtype=float)
        if u.shape != self.shape:
            raise ValueError(f'argument {self.name!r} has shape {u.shape}, expected {self.shape}')
        return device.to_dev(u.ravel(), 'float64')

    def gradient_dev(self, u):
        '''f + K u on the device (Monomial with the CSR rows as output index).'''
        g = self.f.clone() if self.f is not None else device.zeros(self.size, 'float64')
        for values, rowptr, colidx in self.K:
            kernels.monomial_csr(rowptr, colidx, values, u, g)
        for k, values, indices in self.T:  # d/du_m of T[u, .., u] = k T[m, u, .., u] (all permutations are stored: evaluable.py:5727-5735, powers[0])
            kernels.monomial(values, [u] * (k - 1), [indices[a] for a in range(1, k)], g, out_index=indices[0], alpha=float(k))
        return g

    def eval(self, **arguments):
        u = self._u(arguments)
        out = device.zeros(1, 'float64')
        idx = device.to_dev(numpy.arange(self.size), 'int64')
        if self.f is not None:
            kernels.monomial(self.f, [u], [idx], out)
        for values, rowptr, colidx in self.K:
            ku = device.zeros(self.size, 'float64')
            kernels.monomial_csr(rowptr, colidx, values, u, ku)
            kernels.monomial(ku, [u], [idx], out, alpha=.5)
        for k, values, indices in self.T:
            kernels.monomial(values, [u] * k, [indices[a] for a in range(k)], out)
        return self.c + float(device.to_host(out)[0])

    def derivative(self, name):
        if name != self.name:
            raise NotImplementedError('derivative with respect to another argument')
        return FactoredVector(self)


class FactoredVector:
    '''d value / d u = f + K u'''

    def __init__(self, parent):
        self.parent = parent

    def eval(self, **arguments):
        p = self.parent
        return device.to_host(p.gradient_dev(p._u(arguments))).reshape(p.shape)

    def derivative(self, name):
        if name != self.parent.name:
            raise NotImplementedError
        return FactoredMatrix(self.parent)


class FactoredMatrix:
    '''d2 value / d u2 = K (constant): the pre-integrated CSR tensors themselves.'''

    def __init__(self, parent):
        self.parent = parent

    def as_csr(self):
        if self.parent.T:
            raise NotImplementedError('Hessian of a factored functional of degree >= 3 (depends on the argument): integrate the second derivative instead')
        if len(self.parent.K) != 1:
            raise NotImplementedError('sum of tensors with different patterns: assemble per sample and add')
        values, rowptr, colidx = self.parent.K[0]
        return device.to_host(values), device.to_host(rowptr), device.to_host(colidx)


def factor(integral, name=None):
    return Factored(integral, name)
