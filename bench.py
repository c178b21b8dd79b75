#!/usr/bin/env python3
'''bench.py -- elements assembled/sec for the global stiffness matrix K.

Workload (BASELINE.json configs[1]): 3-D Poisson on a 128^3 structured hex mesh,
p=1, 2x2x2 Gauss, stiffness-matrix assembly on one MI355X.  Geometry variant:
isoparametric (vertices perturbed by default_rng(0).uniform(-.2,.2))
so that Jacobians, inverses and the local contraction are really computed per
element; `--variant uniform` times the exact-uniform mesh instead.

A "step" is one full (re)assembly of the CSR values with all inputs resident in HBM
(connectivity, vertex coordinates, tabulated bases, sparsity pattern): zero-fill (if
the kernel accumulates) + element kernel.  The one-time pattern build is reported
separately (`pattern_ms`).  With --gpus N every rank assembles its own 128^3-element
slab of a (128 N) x 128 x 128 mesh (weak scaling) and the shared dof plane between
neighbouring slabs is reduced over RCCL.

Prints ONE JSON line on rank 0.
'''
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--settle', type=int, default=300, help='untimed steps before the warm-up: the first ~50 launches after idle run ~10 %% slower (clock ramp)')
    ap.add_argument('--n', '--elements-per-axis', dest='n', type=int, default=128,
                    help='elements per axis per GPU (use the long form behind torch.distributed.run, whose parser claims --n)')
    ap.add_argument('--variant', choices=['iso', 'uniform'], default='iso')
    ap.add_argument('--kernel', choices=['auto', 'generic', 'fast'], default='auto')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-graph', dest='graph', action='store_false', help='launch every step eagerly instead of replaying a captured HIP graph')
    return ap.parse_args()


def measured_traffic(kernel_name, n):
    '''HBM bytes per launch measured with rocprofv3 PMC counters for this kernel (profiles/r01_traffic.json), or None.'''
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'r01_traffic.json'))).get(kernel_name)
        return d['fetch_bytes'] + d['write_bytes'] if d and d['n'] == n else None
    except Exception:
        return None


def timed_steps(wl, steps, world, dist, use_graph):
    '''Time EXACTLY `steps` steps between barrier + synchronize on both sides (max over ranks).  On one GPU the launch-bound
    step (one kernel of 0.1-0.3 ms behind ~0.1 ms of Python/ctypes launch work) is captured once in a HIP graph and replayed;
    the kernel's own duration is measured in a separate pass with HIP events on the launch stream.'''
    import torch
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(steps):
        wl.step(kernel_events=kev[i])
    torch.cuda.synchronize()
    kernel_ms = sum(s.elapsed_time(e) for s, e in kev) / steps
    graph = None
    if use_graph and world == 1:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                wl.step()
            graph.replay()
            torch.cuda.synchronize()
        except Exception:
            graph = None
            torch.cuda.synchronize()
    def run(replay):
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            if replay:
                graph.replay()
            else:
                wl.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device='cuda', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed, launch = run(False), 'eager'
    if graph is not None:  # both are complete executions of `steps` steps; report the faster launch mode
        eg = run(True)
        if eg < elapsed:
            elapsed, launch = eg, 'hipGraph replay'
    return elapsed, kernel_ms, launch


def cpu_baseline(variant):
    '''Oracle port (oracle/c, kind "port") on the host cores: bounded sample of the same workload.'''
    import numpy
    from oracle import assemble as oa, port
    if not port.available():
        return None
    cores = len(os.sched_getaffinity(0))
    pts, w = oa.gauss(2, 3)
    _, coeffs, _ = oa.structured_basis((1, 1, 1), 'std', 1)
    N, dN = oa.tabulate(coeffs[0], pts)
    T = numpy.concatenate([N.T[:, :, None], dN.transpose(1, 0, 2)], axis=2)

    def run(n):
        verts = None
        if variant == 'iso':
            rng = numpy.random.default_rng(0)
            verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, ((n + 1) ** 3, 3))
        t0 = time.perf_counter()
        v, rp, ci, (tl, td) = port.laplace3d((n, n, n), 1, T, T, w, verts, threads=cores)
        return time.perf_counter() - t0, tl, td

    t32, _, _ = run(32)
    n = 32
    for cand in (64, 96, 128):
        if t32 * (cand / 32) ** 3 <= 25.:
            n = cand
    t, tl, td = (t32, 0, 0) if n == 32 else run(n)
    return {'value': n ** 3 / t, 'unit': 'elements/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n}^3-element {variant} P1 hex Laplace, full COO->sort->CSR assembly, {t:.2f} s '
                      f'(element loop {tl:.2f} s on {cores} OpenMP threads, serial radix-sort dedup {td:.2f} s)'}


def main():
    a = parse()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: required by RCCL / cross-process device memory on this driver
    import numpy
    import torch
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
    # test hook (tests/test_gpu_partition.py): all ranks on ONE GPU with gloo transport, to run the multi-process path on a 1-GPU box
    one_gpu = os.environ.get('NUTILS_AMD_BENCH_ONE_GPU') == '1'
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    from nutils_amd import workloads
    wl = workloads.PoissonSlab(n=a.n, rank=rank, world=world, variant=a.variant, kernel=a.kernel)
    t0 = time.perf_counter()
    wl.setup()
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    wl.build_pattern()
    torch.cuda.synchronize()
    pattern_ms = (time.perf_counter() - t0) * 1e3

    for _ in range(a.settle + a.warmup):  # settle: bring the GPU out of its idle power state (0.236 ms per step over the first 20
        wl.step()                          # launches, 0.215 ms in steady state); then the W warm-up steps of the contract
    torch.cuda.synchronize()
    elapsed, kernel_ms, launch = timed_steps(wl, a.steps, world, dist if world > 1 else None, a.graph)

    # size-independent check of the assembled matrix on every rank (rows it owns are complete after the interface reduce):
    # K 1 = 0, i.e. every owned row sums to zero; catches a lost or doubled halo contribution without moving the matrix
    wl.finish()
    s = wl.slab
    a_row, b_row = s.own_plane_begin * s.plane, s.own_plane_end * s.plane
    lo = int(wl.rowptr[a_row])
    owned = wl.values[lo:int(wl.rowptr[b_row])]
    csum = torch.cat([torch.zeros(1, dtype=torch.float64, device=owned.device), torch.cumsum(owned, 0)])  # stays O(|K|): every row sums to ~0
    rowsum = csum[wl.rowptr[a_row + 1:b_row + 1] - lo] - csum[wl.rowptr[a_row:b_row] - lo]
    check = torch.stack([rowsum.abs().max() / owned.abs().max()])
    del csum, rowsum
    if world > 1:
        dist.all_reduce(check, op=dist.ReduceOp.MAX)
    row_sum_rel = float(check.item())
    if not row_sum_rel < 1e-10:  # a broken kernel or halo reduce must not be recorded as a (possibly faster) valid result
        print(f'ERROR: owned rows do not sum to zero (relative {row_sum_rel:.2e}): the assembled matrix is wrong', file=sys.stderr)
        if rank == 0:
            print(json.dumps({'metric': 'elements assembled/sec (global stiffness K)', 'value': None, 'unit': 'elements/s', 'n_gpus': world,
                              'error': f'correctness gate failed: owned row sums relative {row_sum_rel:.3e}'}))
        if world > 1:
            dist.destroy_process_group()
        sys.exit(1)

    if rank == 0:
        nelems_total = wl.nelems * world
        value = nelems_total * a.steps / elapsed
        bytes_per_elem = wl.algorithmic_bytes_per_element()
        achieved = bytes_per_elem * wl.nelems / (kernel_ms * 1e-3) / 1e9
        out = {
            'metric': 'elements assembled/sec (global stiffness K)', 'value': value, 'unit': 'elements/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'3D Poisson stiffness, {a.n}^3 structured hex per GPU, p=1, 2x2x2 Gauss, {a.variant} geometry '
                                   f'(BASELINE.json configs[1])', 'nelems_per_gpu': wl.nelems, 'nnz_per_gpu': wl.nnz, 'kernel': wl.kernel_name,
                       'parallelism': f'element slabs x{world}, halo-plane reduce' if world > 1 else 'single GPU', 'launch': launch,
                       'settle_steps': a.settle},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': measured_traffic(wl.kernel_name, a.n) if world == 1 else None, 'kernel': wl.kernel_name, 'kernel_ms': kernel_ms, 'algorithmic_bytes_per_element': bytes_per_elem},
            'pattern_ms': pattern_ms, 'setup_s': setup_s, 'checks': {'owned_row_sums_rel': row_sum_rel},
        }
        if world == 1 and a.variant == 'iso':
            # secondary variant of the same config: exact-uniform mesh (what mesh.rectilinear gives the reference; there the
            # element matrix is hoisted out of the loop and the work is index generation + dedup).  Not the headline value.
            w2 = workloads.PoissonSlab(n=a.n, rank=0, world=1, variant='uniform', kernel=a.kernel)
            w2.setup()
            w2.build_pattern()
            for _ in range(a.settle + a.warmup):
                w2.step()
            torch.cuda.synchronize()
            el2, kms, _ = timed_steps(w2, a.steps, 1, None, a.graph)
            b2 = w2.algorithmic_bytes_per_element()
            ach = b2 * w2.nelems / (kms * 1e-3) / 1e9
            out['variants'] = {'uniform': {'value': w2.nelems * a.steps / el2, 'unit': 'elements/s', 'ms_per_step': el2 / a.steps * 1e3, 'kernel': w2.kernel_name,
                                           'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                                                        'traffic': measured_traffic(w2.kernel_name, a.n),
                                                        'kernel_ms': kms, 'algorithmic_bytes_per_element': b2}}}
            del w2
        if not a.no_cpu and world == 1:
            cb = cpu_baseline(a.variant)
            if cb:
                out['cpu_baseline'] = cb
                out['speedup_vs_cpu_port'] = value / cb['value']
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
