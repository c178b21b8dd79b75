import sys, time, os
sys.path.insert(0, '.')
os.environ['NUTILS_AMD_NO_FAST_PATH'] = '1'
import numpy, torch
from nutils_amd import mesh, function
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = numpy.random.default_rng(0)
domain, geom0 = mesh.rectilinear([n] * 3)
basis = domain.basis('std', degree=1)
verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (len(basis), 3))
geom = basis @ verts
u = domain.field('u', btype='std', degree=1)
res = domain.integral(((function.grad(basis, geom) * function.grad(u, geom)).sum(-1) + basis * function.value(u) - basis) * function.J(geom), degree=2)
args = {'u': rng.normal(size=len(basis))}
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = function.eval(res, arguments=args)
    torch.cuda.synchronize(); print(f'generic residual {n}^3 P1: {1e3 * (time.perf_counter() - t0):.2f} ms')
