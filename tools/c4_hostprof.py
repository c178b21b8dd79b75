import sys, time, cProfile, pstats
sys.path.insert(0, '.')
sys.argv = ['c4_probe.py', '512']
import runpy
ns = runpy.run_path('tools/c4_probe.py')
system, args = ns['system'], ns['args']
import torch
for _ in range(3): system.assemble_residual(args)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): system.assemble_residual(args)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
