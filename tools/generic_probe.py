#!/usr/bin/env python3
'''Re-assembly times of the generic (any-mesh) matrix path for a few element types: python tools/generic_probe.py
(NUTILS_AMD_NO_FAST_PATH=1 is set: the structured write-once kernels are bypassed)'''
import os, sys, time
os.environ['NUTILS_AMD_NO_FAST_PATH'] = '1'
sys.path.insert(0, '.')
import numpy, torch
from nutils_amd import mesh, function, sample

CASES = [('3D P1 128^3', [128] * 3, 'std', 1, 1), ('3D P2 scalar 64^3', [64] * 3, 'std', 2, 1), ('3D spline2 scalar 64^3', [64] * 3, 'spline', 2, 1),
         ('2D P1 2048^2', [2048] * 2, 'std', 1, 1), ('2D P2 1024^2', [1024] * 2, 'std', 2, 1), ('2D spline3 1024^2', [1024] * 2, 'spline', 3, 1),
         ('2D spline2 1024^2', [1024] * 2, 'spline', 2, 1), ('3D spline3 32^3', [32] * 3, 'spline', 3, 1), ('2D P3 512^2', [512] * 2, 'std', 3, 1), ('2D P4 512^2', [512] * 2, 'std', 4, 1),
         ('3D P1 elasticity 96^3', [96] * 3, 'std', 1, 3), ('2D P1 elasticity 1024^2', [1024] * 2, 'std', 1, 2), ('2D P2 elasticity 512^2', [512] * 2, 'std', 2, 2),
         ('3D P2 elasticity 32^3', [32] * 3, 'std', 2, 3)]
only = sys.argv[1:] 
ONLY = os.environ.get("GENERIC_PROBE_ONLY")
for name, shape, btype, degree, nc in CASES:
    if ONLY and ONLY not in name:
        continue
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
        basis = domain.basis(btype, degree=degree)
        K = domain.integral(function.outer(function.grad(basis, X)).sum(-1) * dV, degree=2 * degree)
    else:
        u = domain.field('u', btype=btype, degree=degree, shape=[nd])
        v = domain.field('v', btype=btype, degree=degree, shape=[nd])
        eps = lambda w: function.symgrad(w, X)
        sigma = function.div(u, X) * function.eye(nd) + 1.3 * eps(u)
        K = function.derivative(function.derivative(domain.integral(function.inner(eps(v), sigma) * dV, degree=2 * degree), 'v'), 'u')
    plan = sample._MatrixPlan(K.terms)
    t0 = time.perf_counter(); out = plan.run({}); torch.cuda.synchronize(); t1 = time.perf_counter()
    plan.run({}); torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(5):
        plan.run({})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t2) / 5
    ne = int(numpy.prod(shape))
    print(f'{name:28s} nelems {ne:9d} nnz {out[0].numel():11d}: first {1e3*(t1-t0):8.1f} ms, re-assembly {1e3*dt:8.3f} ms = {ne/dt:.3e} elements/s, {out[0].numel()*8/dt/1e9:7.1f} GB/s of values')
    del plan, out
    torch.cuda.empty_cache()
