'''System: the caller of the assembly path, mirroring the part of
/root/reference/src/nutils/solver.py:189-431 that orchestrates (re)assembly:
``System(functional_or_residual, trial, test)`` builds residual and Jacobian by
differentiation (solver.py:238,253), assembles them on the GPU, applies constraints
on the host (NaN = free dof, solver.py:273-315) and hands the matrix to the host
solver (``matrix.assemble_csr`` -> scipy).  Only linear problems (the class of forms
the accelerated path represents) -- the Newton/minimize drivers stay with the
reference.'''

import numpy

from . import function, matrix as _matrix, sample as _sample, device


class SolverError(Exception):
    pass


class System:

    def __init__(self, residual, /, trial, test=None):
        if not isinstance(residual, function.Integral):
            raise TypeError('System expects an Integral (sum of sample.integral terms)')
        self.trials = tuple(trial.split(',')) if isinstance(trial, str) else tuple(trial)
        if len(self.trials) != 1:
            raise NotImplementedError('multi-field systems (block Jacobians) are not on the accelerated path yet')
        tests = self.trials if test is None else (tuple(test.split(',')) if isinstance(test, str) else tuple(test))
        self.is_symmetric = tests == self.trials
        self.value = residual if self.is_symmetric else None
        self.residual = function.derivative(residual, tests[0])
        self.jacobian = function.derivative(self.residual, self.trials[0])
        if not self.jacobian.terms:
            raise ValueError('the functional does not depend on the trial argument')
        arg = self.jacobian.terms[0][1].trial
        self.trial_arg = arg
        self.trial_shape = (arg.basis.ndofs, arg.ncomp) if arg.ncomp > 1 else (arg.basis.ndofs,)
        self.is_linear = True
        self.is_constant_matrix = True
        self._jac = None

    # -- assembly (solver.py:318-386) --

    def assemble_jacobian(self):
        '''Constant Jacobian, cached like solver.py:321-331; terms on different samples (volume + boundary) are
        assembled separately and added on the host.'''
        if self._jac is None:
            by_sample = {}
            for term in self.jacobian.terms:
                by_sample.setdefault(id(term[0]), []).append(term)
            total = None
            n = int(numpy.prod(self.trial_shape))
            for terms in by_sample.values():
                values, rowptr, colidx = _sample.evaluate(function.as_csr(function.Integral(terms)), {})
                m = _matrix.assemble_csr(values, rowptr, colidx, n)
                total = m if total is None else _matrix.ScipyMatrix(total.core + m.core)
            self._jac = total
        return self._jac

    def assemble_residual(self, arguments):
        '''Residual at the given trial value (linear: res(0) + jac @ x, solver.py:364-378).'''
        zero = dict(arguments)
        zero[self.trials[0]] = numpy.zeros(self.trial_shape)
        res0 = numpy.zeros(int(numpy.prod(self.trial_shape)))
        for term in self.residual.terms:
            r = _sample.evaluate(function.Integral([term]), zero)
            res0 += numpy.asarray(r).ravel()
        x = numpy.asarray(arguments.get(self.trials[0], zero[self.trials[0]]), dtype=float).ravel()
        return res0 + self.assemble_jacobian() @ x

    def assemble_jacobian_residual(self, arguments):
        return self.assemble_jacobian(), self.assemble_residual(arguments)

    def assemble_value(self, arguments):
        if not self.is_symmetric:
            raise Exception('value is not defined')
        return function.eval(self.value, arguments)

    # -- solves --

    def solve(self, *, arguments=None, constrain=None):
        '''Direct solve of the linear system; `constrain[trial]` holds NaN for free dofs (solver.py:440-500, Direct).'''
        arguments = dict(arguments or {})
        t = self.trials[0]
        cons = None if not constrain or t not in constrain else numpy.asarray(constrain[t], dtype=float).ravel()
        x0 = numpy.zeros(int(numpy.prod(self.trial_shape)))
        if cons is not None:
            x0[~numpy.isnan(cons)] = cons[~numpy.isnan(cons)]
        arguments[t] = x0.reshape(self.trial_shape)
        jac, res = self.assemble_jacobian_residual(arguments)
        free = numpy.ones(len(x0), dtype=bool) if cons is None else numpy.isnan(cons)
        dx = -jac.solve(res, constrain=~free)
        arguments[t] = (x0 + dx).reshape(self.trial_shape)
        return arguments

    def solve_constraints(self, *, droptol, arguments=None, constrain=None):
        '''Dirichlet constraints by boundary projection (solver.py:562-612): solve on the dofs whose matrix column has an
        entry above droptol, return NaN for all others.'''
        arguments = dict(arguments or {})
        t = self.trials[0]
        n = int(numpy.prod(self.trial_shape))
        arguments[t] = numpy.zeros(self.trial_shape)
        jac, res = self.assemble_jacobian_residual(arguments)
        data, colidx, _ = jac.export('csr')
        mycons = numpy.ones(n, dtype=bool)
        mycons[colidx[abs(data) > droptol]] = False
        x = -jac.solve(res, constrain=mycons)
        x[mycons] = numpy.nan
        out = dict(constrain or {})
        prev = out.get(t)
        if prev is not None:
            prev = numpy.asarray(prev, dtype=float).ravel()
            x = numpy.where(numpy.isnan(prev), x, prev)
        out[t] = x.reshape(self.trial_shape)
        return out
