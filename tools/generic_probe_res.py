#!/usr/bin/env python3
'''Residual (linear-form) times of the generic path: r_m = int grad(phi_m) . grad(u) (+ elasticity) for a few element types: python tools/generic_probe_res.py'''
import os, sys, time
os.environ['NUTILS_AMD_NO_FAST_PATH'] = '1'
sys.path.insert(0, '.')
import numpy, torch
from nutils_amd import mesh, function

CASES = [('3D P1 128^3', [128] * 3, 'std', 1, 1), ('3D P2 scalar 64^3', [64] * 3, 'std', 2, 1), ('3D spline2 scalar 64^3', [64] * 3, 'spline', 2, 1),
         ('2D P1 2048^2', [2048] * 2, 'std', 1, 1), ('2D P2 1024^2', [1024] * 2, 'std', 2, 1), ('2D spline3 1024^2', [1024] * 2, 'spline', 3, 1),
         ('3D spline3 32^3', [32] * 3, 'spline', 3, 1), ('3D P1 elasticity 96^3', [96] * 3, 'std', 1, 3), ('2D P1 elasticity 1024^2', [1024] * 2, 'std', 1, 2),
         ('3D P2 elasticity 32^3', [32] * 3, 'std', 2, 3)]
only = sys.argv[1:]
for name, shape, btype, degree, nc in CASES:
    if only and not any(o in name for o in only):
        continue
    nd = len(shape)
    domain, geom = mesh.rectilinear(shape)
    rng = numpy.random.default_rng(0)
    gb = domain.basis('std', degree=1)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, nd) + rng.uniform(-.2, .2, (len(gb), nd))
    X = gb @ verts
    dV = function.J(X)
    if nc == 1:
        u = domain.field('u', btype=btype, degree=degree)
        energy = domain.integral(.5 * (function.grad(u, X) * function.grad(u, X)).sum(-1) * dV, degree=2 * degree)
    else:
        u = domain.field('u', btype=btype, degree=degree, shape=[nd])
        eps = function.symgrad(u, X)
        energy = domain.integral((.5 * function.div(u, X) ** 2 + .65 * function.inner(eps, eps)) * dV, degree=2 * degree)
    res = function.derivative(energy, 'u')
    args = {'u': rng.normal(size=(len(u.arg.basis), nd) if nc > 1 else len(u.arg.basis))}
    function.eval(res, arguments=args); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        function.eval(res, arguments=args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    ne = int(numpy.prod(shape))
    print(f'{name:28s} nelems {ne:9d}: residual incl. upload / download {1e3*dt:8.3f} ms = {ne/dt:.3e} elements/s')
    torch.cuda.empty_cache()
