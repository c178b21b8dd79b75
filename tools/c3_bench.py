#!/usr/bin/env python3
'''Kernel-level timing of BASELINE.json configs[2] (64^3 P2 vector elasticity): python tools/c3_bench.py [n] [steps] [iso|uniform]'''
import sys
sys.path.insert(0, '.')
import torch
from nutils_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wl = workloads.ElasticityP2(n=n, variant=sys.argv[3] if len(sys.argv) > 3 else 'iso')
wl.setup()
wl.build_pattern()
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
for i in range(steps):
    wl.step(kernel_events=ev[i])
torch.cuda.synchronize()
ms = sorted(s.elapsed_time(e) for s, e in ev)
gb = wl.algorithmic_bytes_per_element() * wl.nelems / 1e9
print(f'n={n} nnz={wl.nnz} kernel ms: min {ms[0]:.3f} median {ms[len(ms) // 2]:.3f} max {ms[-1]:.3f}  -> {wl.nelems / ms[len(ms) // 2] * 1e3:.3e} el/s, '
      f'{gb / ms[len(ms) // 2] * 1e3:.0f} GB/s algorithmic ({gb:.2f} GB), {wl.algorithmic_flops_per_element() * wl.nelems / ms[len(ms) // 2] * 1e3 / 1e12:.2f} TFLOP/s')
v = wl.values
print('checksum', float(v.sum()), float(v.abs().max()))
