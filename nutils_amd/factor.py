'''function.factor: pre-integrate a polynomial functional of a field ONCE
into sparse Taylor coefficient tensors, then evaluate it / its derivatives per solver step with
the Monomial kernels (gather-multiply-scatter over nnz) instead of re-running the element loop.

Reference: function.factor (/root/reference/src/nutils/function.py:2630-2642) ->
evaluable.factor (evaluable.py:5785-5874) + Monomial (evaluable.py:5693-5751).  The reference
expands to arbitrary polynomial degree; the accelerated class of integrands (function.py here)
is at most quadratic in a field, so the expansion is  c + f.u + 1/2 u.K.u  with K kept as
device-resident CSR tensors (one per sample: volume and boundary terms have different
patterns, exactly as the reference keeps a sum of Monomials).'''

import numpy

from . import device, function, kernels, sample as _sample


class Factored:
    '''value(u) = c + f.u + 1/2 u.K.u'''

    def __init__(self, integral, name=None):
        if not isinstance(integral, function.Integral):
            raise TypeError('factor expects an Integral')
        names = {a.name for _, itg, _ in integral.terms for a in (itg.test, itg.trial) if a is not None and a.name is not None}
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
        arg = next(a for _, itg, _ in integral.terms for a in (itg.test, itg.trial) if a is not None and a.name == name)
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
            values, rowptr, colidx, ncols = _sample._MatrixPlan(terms).run()
            self.K.append((values, rowptr, colidx))

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
        if len(self.parent.K) != 1:
            raise NotImplementedError('sum of tensors with different patterns: assemble per sample and add')
        values, rowptr, colidx = self.parent.K[0]
        return device.to_host(values), device.to_host(rowptr), device.to_host(colidx)


def factor(integral, name=None):
    return Factored(integral, name)
