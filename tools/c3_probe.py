#!/usr/bin/env python3
'''C3 probe: 3-D linear elasticity, P2 vector basis, n^3 elements (BASELINE.json configs[2]) through the generic path.'''
import sys, time
sys.path.insert(0, '.')
import numpy, torch
from nutils_amd import mesh, function, sample, device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iso = len(sys.argv) > 2 and sys.argv[2] == 'iso'
domain, geom = mesh.rectilinear([n] * 3)
if iso:
    gb = domain.basis('std', degree=1)
    rng = numpy.random.default_rng(0)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (len(gb), 3))
    geom = gb @ verts
u = domain.field('u', btype='std', degree=2, shape=[3])
v = domain.field('v', btype='std', degree=2, shape=[3])
lam, mu = 1., .5 / .3 - 1
eps = lambda w: function.symgrad(w, geom)
sigma = lam * function.div(u, geom) * function.eye(3) + 2 * mu * eps(u)
res = domain.integral(function.inner(eps(v), sigma) * function.J(geom), degree=4)
jac = function.derivative(function.derivative(res, 'v'), 'u')
plan = sample._MatrixPlan(jac.terms)
t0 = time.perf_counter(); values, rowptr, colidx, ncols = plan.run(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f'n={n} nelems={n**3} ndofs={ncols} nnz={values.numel()} first assembly (pattern + tables + values) {t1 - t0:.3f} s')
for _ in range(2):
    t0 = time.perf_counter(); values, rowptr, colidx, ncols = plan.run(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f'  re-assembly incl. expand {t1 - t0:.4f} s -> {n**3 / (t1 - t0):.3e} elements/s')
print('checksum', float(values.sum()), float(values.abs().max()))
