#!/usr/bin/env python3
'''Kernel-level timing of the any-mesh entry on BASELINE.json configs[1] (128^3 trilinear, iso geometry): owner blocks (NH_MATRIX_FUSED) against the two-pass
owner-side reduction (NH_MATRIX_GATHER).  python tools/fused_probe.py [n] [steps]'''
import sys
sys.path.insert(0, '.')
import torch
from nutils_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ref = None
for kernel in ('gather', 'fused'):
    wl = workloads.PoissonSlab(n=n, rank=0, world=1, variant='iso', kernel=kernel)
    wl.setup()
    wl.build_pattern()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    wl.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(steps):
        wl.step(kernel_events=ev[i])
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    gb = wl.algorithmic_bytes_per_element() * wl.nelems / 1e9
    info = wl.pattern.fused_info() if kernel == 'fused' else None
    print(f'{kernel}: first assembly (plan / map build included) {1e3 * (t1 - t0):.1f} ms; kernel ms min {ms[0]:.3f} median {ms[len(ms) // 2]:.3f} -> '
          f'{gb / ms[len(ms) // 2] * 1e3:.0f} GB/s algorithmic ({gb:.3f} GB)', info or '')
    v = wl.values.clone()
    if ref is None:
        ref = v
    else:
        print('max |fused - gather| / max |K| =', float((v - ref).abs().max() / ref.abs().max()))
    del wl
    torch.cuda.empty_cache()
