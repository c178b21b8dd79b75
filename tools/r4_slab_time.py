#!/usr/bin/env python3
'''Round 4: kernel time of ONE slab of the strong-scaling split of the 128^3 Poisson mesh on one GPU (no peers needed: rank / world are bookkeeping) --
rank 0 (no ghost layer) against an inner rank with halo='recompute' (one ghost layer assembled too) and halo='reduce' (kernel only, exchange not timed),
20 steps per HIP graph.  python tools/r4_slab_time.py [config c2|c3]'''
import sys
sys.path.insert(0, '.')
import torch
from nutils_amd import workloads

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
n = 128 if cfg == 'c2' else 64
for world in (2, 4, 8):
    layers = n // world
    row = []
    for rank, halo in ((0, 'recompute'), (1, 'recompute'), (1, 'reduce')):
        wl = (workloads.PoissonSlab if cfg == 'c2' else workloads.ElasticityP2)(n=n, layers=layers, rank=rank, world=world, halo=halo)
        wl.setup()
        wl.build_pattern()
        for _ in range(50 if cfg == 'c2' else 5):
            wl.step(exchange=False)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        per = 20 if cfg == 'c2' else 4
        with torch.cuda.graph(g):
            for _ in range(per):
                wl.step(exchange=False)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        reps = 20 if cfg == 'c2' else 5
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        row.append(a.elapsed_time(b) / reps / per)
        del wl, g
        torch.cuda.empty_cache()
    print(f'{cfg} world {world}: {layers} layers per rank: rank 0 {row[0]*1e3:.1f} us | inner rank, ghost layer recomputed {row[1]*1e3:.1f} us (x{row[1]/row[0]:.3f}; (layers+1)/layers = {(layers+1)/layers:.3f}) | '
          f'inner rank, reduce mode kernel only {row[2]*1e3:.1f} us')
