import sys, time, runpy, os
sys.path.insert(0, '.')
import torch
from nutils_amd import solver, sample as _sample
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T.setdefault(name, []).append((time.perf_counter() - t0) * 1e6)
    setattr(obj, name, g)
wrap(_sample, 'prefetch_arguments')
wrap(solver.System, '_start_jacobian')
wrap(solver.System, 'assemble_residual')
wrap(solver.System, '_finish_jacobian')
wrap(solver.System, '_dyn_values')
wrap(solver._HostMirror, 'publish')
wrap(_sample, 'start_blocks')
wrap(solver.System, 'assemble_jacobian_residual')
os.environ['C4_STEPS'] = '8'
sys.argv = ['c4_step.py', '512']
runpy.run_path('tools/c4_step.py', run_name='__main__')
for k, v in T.items():
    print(f'{k:28s} n={len(v):3d} last5 us: ' + ' '.join(f'{x:7.0f}' for x in v[-5:]))
