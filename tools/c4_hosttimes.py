'''Host timestamps inside one C4 Newton step (no profiler): python tools/c4_hosttimes.py'''
import sys, time
sys.path.insert(0, '.')
sys.argv = ['c4_step.py', '512']
import runpy
ns = runpy.run_path('tools/c4_step.py')
system, args = ns['system'], ns['args']
import torch
from nutils_amd import kernels, device, sample, solver
marks = []
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        marks.append((name + ' >', time.perf_counter()))
        r = orig(*a, **k)
        marks.append((name + ' <', time.perf_counter()))
        return r
    setattr(mod, name, f)
for mod, name in [(kernels, 'assemble_terms_multi'), (device, 'to_host'), (device, 'to_dev'), (kernels, 'index_copy'), (kernels, 'assemble_matrix_terms'), (kernels, 'monomial'),
                  (sample, '_vector_blocks'), (solver._HostMirror, 'publish')]:
    wrap(mod, name)
for it in range(4):
    marks.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    system.assemble_jacobian_residual(args, copy=False)
    t1 = time.perf_counter()
print(f'step {1e3 * (t1 - t0):.2f} ms')
for n, t in marks:
    print(f'{1e6 * (t - t0):8.0f} us  {n}')
