#!/usr/bin/env python3
'''Kernel time of the headline kernel without the result check (ablation switches make the matrix wrong): python tools/c2_time.py [n] [steps]'''
import sys
sys.path.insert(0, '.')
import torch
from nutils_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = workloads.PoissonSlab(n=n)
wl.setup()
wl.build_pattern()
for _ in range(400):
    wl.step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(steps):
    wl.step()
b.record()
torch.cuda.synchronize()
print(f'{a.elapsed_time(b) / steps:.4f} ms per step (eager launches)')
