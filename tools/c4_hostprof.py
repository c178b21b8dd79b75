'''Host-side profile of the C4 Newton step (cProfile over 20 calls): python tools/c4_hostprof.py [residual|step]'''
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
what = sys.argv[1] if len(sys.argv) > 1 else 'step'
sys.argv = ['c4_probe.py', '512']
import runpy
ns = runpy.run_path('tools/c4_probe.py')
system, args = ns['system'], ns['args']
import torch
call = system.assemble_residual if what == 'residual' else system.assemble_jacobian_residual
for _ in range(3): call(args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): call(args)
torch.cuda.synchronize()
print(f'{what}: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms per call (20 calls back to back)')
pr = cProfile.Profile(); pr.enable()
for _ in range(20): call(args)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
