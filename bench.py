#!/usr/bin/env python3
'''bench.py -- elements assembled/sec for the global stiffness matrix K.

Default workload (BASELINE.json configs[1], `--config c2`): 3-D Poisson on a 128^3 structured hex mesh, p=1,
2x2x2 Gauss, stiffness-matrix assembly on one MI355X.  Geometry variant: isoparametric (vertices perturbed by
default_rng(0).uniform(-.2,.2)) so that Jacobians, inverses and the local contraction are really
computed per element; `--variant uniform` times the exact-uniform mesh instead.
`--config c3` (BASELINE.json configs[2]): 3-D linear elasticity, P2 vector basis, 64^3 elements, 3x3x3 Gauss.

A "step" is one full (re)assembly of the CSR values with all inputs resident in HBM (vertex coordinates, tabulated
bases, sparsity pattern): zero-fill (if the kernel accumulates) + element kernel; `value` is measured on that.  The
regions of BASELINE.md 3 are reported next to it: `first_assembly_ms` (pattern + values, device resident),
`host_ready_ms` (pattern + values + the copy of (values, rowptr, colidx) to host memory, i.e. "CSR ready for scipy")
and `reassembly_host_ready_ms` (values only); the CPU leg times pattern + values on the host, so
`speedup_vs_cpu_port.first_assembly_host_ready` is the like-for-like ratio.

With --gpus N (default `--scaling strong`, SURVEY.md 8e / BASELINE.json north_star: "elements are partitioned across the
GPUs of one node"): the ONE n^3 mesh is split into N slabs of n/N element layers; the line also carries the weak-scaling
figure of the same launch (`weak`: every rank assembles its own n^3-element slab of an (n N) x n x n mesh).  `--scaling weak`
makes the weak figure the headline instead.  With the default `--halo recompute` no data moves between the ranks (each assembles its ghost element layer and
writes the rows it owns); `--halo reduce` reduces the shared dof plane over RCCL (point to point, interface rows only).  At N = 1 the default line also carries `variants.c3` (BASELINE.json configs[2], measured in the
same run: kernel time by HIP events, HBM fraction, CPU port), `variants.c4` (configs[3]: one Newton step of the 512^2 Cahn-Hilliard system, tools/c4_step.py) and
`variants.c5` (the configs[4] class: 63 488 ragged rational hierarchical elements, parity-checked against the reference fixture, tools/ragged_probe.py) and
`variants.vector_any_mesh` (96^3 trilinear elasticity through the any-mesh entry: the owner kernel for vector-valued blocks, tools/vector_probe.py).

Prints ONE JSON line on rank 0.
'''
import argparse
import json
import os
import sys
import time
import subprocess

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.     # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
F64_MFMA_PEAK_TF = 78.6  # dense f64 MFMA (= f64 vector) peak, 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz
TRAFFIC_FILE = 'profiles/r06_traffic.json'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--config', choices=['c2', 'c3'], default='c2', help='c2: 128^3 P1 Poisson (headline); c3: 64^3 P2 vector elasticity')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default=None, help='default: strong (one mesh split over the GPUs) when the element layers divide evenly, else weak')
    ap.add_argument('--halo', choices=['recompute', 'reduce'], default='recompute',
                    help='N > 1: recompute = every rank also assembles its one ghost element layer and writes only the rows it owns (no exchange; the step is one '
                         'kernel, graph-captured); reduce = RCCL point-to-point reduce of the interface-plane rows.  With --compare-halo the line carries the other mode as `halo_<mode>`')
    ap.add_argument('--compare-halo', action='store_true', help='N > 1: also time the other halo mode (secondary figure).  Off by default: the reduce mode is the only part of the '
                    'bench that moves data between ranks, and it has never run on multi-GPU hardware -- a stall there would take the headline line with it')
    ap.add_argument('--traffic', choices=['measure', 'static'], default='measure',
                    help='roofline.traffic: measure = two rocprofv3 --pmc passes of a short probe of the same kernel inside this run (when rocprofv3 is on PATH), '
                         'static = the committed profile file')
    ap.add_argument('--settle', type=int, default=None, help='untimed steps before the warm-up: the first ~50 launches after idle run ~10 %% slower (clock ramp)')
    ap.add_argument('--n', '--elements-per-axis', dest='n', type=int, default=None,
                    help='elements per axis (per GPU with weak scaling; use the long form behind torch.distributed.run, whose parser claims --n)')
    ap.add_argument('--variant', choices=['iso', 'uniform'], default='iso')
    ap.add_argument('--kernel', choices=['auto', 'generic', 'gather', 'fused', 'batched', 'fast'], default='auto')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-c3', action='store_true', help='skip the configs[2] variant of the default line')
    ap.add_argument('--no-graph', dest='graph', action='store_false', help='launch every step eagerly instead of replaying a captured HIP graph')
    a = ap.parse_args()
    c3 = a.config == 'c3'
    a.n = a.n or (64 if c3 else 128)
    a.steps = a.steps if a.steps is not None else (30 if c3 else 200)
    a.warmup = a.warmup if a.warmup is not None else (3 if c3 else 20)
    a.settle = a.settle if a.settle is not None else (10 if c3 else 300)
    if a.scaling is None:
        a.scaling = 'strong' if a.n % max(a.gpus, 1) == 0 else 'weak'
    return a


def measured_traffic(kernel_name, n):
    '''HBM bytes per launch measured with rocprofv3 PMC counters for this kernel (TRAFFIC_FILE: separate --pmc passes of this
    command, FETCH_SIZE + WRITE_SIZE), or None.  Static: a profile of the same kernel at the same size, not measured in this run.'''
    try:
        d = json.load(open(os.path.join(ROOT, TRAFFIC_FILE))).get(kernel_name)
        return d['fetch_bytes'] + d['write_bytes'] if d and d['n'] == n else None
    except Exception:
        return None


def measured_traffic_inrun(kernel_substr, probe_cmd):
    '''HBM bytes per launch of the kernel measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, and counters are
    collected with --kernel-trace only, as the guide prescribes) of a short probe command that launches the same kernel on the same workload.  Returns
    None when rocprofv3 is not on PATH or a pass fails (then the static profile file is used and labelled so).  Units: KB x 1024; FETCH_SIZE is
    reported as counted AND doubled (the guide: on gfx950 a wide streaming read is tallied at half its bytes).'''
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which('rocprofv3') or os.environ.get('NUTILS_AMD_NO_INRUN_TRAFFIC'):
        return None
    out = {}
    try:
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = tempfile.mkdtemp(prefix='nh_pmc_', dir='/tmp')
            env = dict(os.environ, TMPDIR='/tmp', NUTILS_AMD_NO_INRUN_TRAFFIC='1')
            r = subprocess.run(['rocprofv3', '--kernel-trace', '--pmc', ctr, '-d', d, '-o', 'p', '-f', 'csv', '--'] + probe_cmd, cwd=ROOT, env=env,
                               capture_output=True, text=True, timeout=240)
            vals = []
            for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
                for row in csv.DictReader(open(fn)):
                    if kernel_substr in row['Kernel_Name'] and row['Counter_Name'] == ctr:
                        vals.append(float(row['Counter_Value']))
            shutil.rmtree(d, ignore_errors=True)
            if r.returncode != 0 or not vals:
                return None
            vals = vals[len(vals) // 2:]  # (the first launches of the probe run cold)
            out[ctr] = sum(vals) / len(vals) * 1024
        return {'fetch_bytes': out['FETCH_SIZE'], 'write_bytes': out['WRITE_SIZE'], 'bytes': out['FETCH_SIZE'] + out['WRITE_SIZE'],
                'bytes_fetch_doubled': 2 * out['FETCH_SIZE'] + out['WRITE_SIZE'],
                'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) of `' + ' '.join(probe_cmd) + '`, measured in this run'}
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
    graph, per_graph = None, 1
    if use_graph and (world == 1 or getattr(wl, 'halo', None) is None):  # (a step without exchange is a single kernel: capturable on every rank)
        # several steps per graph (a divisor of `steps`, at most 20): the replay of a one-kernel graph still costs ~8 us of launch latency per step
        per_graph = next(g for g in (20, 10, 8, 5, 4, 2, 1) if steps % g == 0)
        try:
            graph = torch.cuda.CUDAGraph()
            # (thread-local capture mode: with N > 1 the RCCL watchdog thread of torch.distributed issues event queries of its own while this thread captures)
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                for _ in range(per_graph):
                    wl.step()
            for _ in range(3):  # (untimed: the first replays of a graph carry its upload to the device)
                graph.replay()
            torch.cuda.synchronize()
        except Exception:
            graph, per_graph = None, 1
            torch.cuda.synchronize()
        if world > 1:  # (every rank times the same launch modes -- the timed regions are bracketed by collectives: one rank without a graph means no rank replays)
            ok = torch.tensor([1 if graph is not None else 0], device='cuda', dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                graph, per_graph = None, 1

    def run(replay):
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if replay:
            for i in range(steps // per_graph):  # (per_graph divides steps: exactly `steps` steps)
                graph.replay()
        else:
            for i in range(steps):
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

    # Both launch modes are complete executions of exactly `steps` steps and both are reported (`ms_per_step_eager`, `ms_per_step_graph`);
    # the headline is the graph replay whenever the step can be captured -- that is how a time loop would issue a launch-bound step -- and
    # is NOT chosen after the fact.
    eager = run(False)
    modes = {'ms_per_step_eager': eager / steps * 1e3}
    elapsed, launch = eager, 'eager'
    if graph is not None:
        graph.replay()  # (untimed: back from the eager launch path to the graph's)
        torch.cuda.synchronize()
        elapsed, launch = run(True), f'hipGraph replay ({per_graph} steps per graph)'
        modes['ms_per_step_graph'] = elapsed / steps * 1e3
    timed_steps.last_modes = modes
    return elapsed, kernel_ms, launch


def host_regions(make_workload, copy=True):
    '''BASELINE.md 3 regions on a fresh workload object: first assembly (pattern + values) device resident, the same with the CSR
    copied to host memory, and a re-assembly (values only) copied to host.  The copies go through page-locked memory
    (device.to_host); one untimed round first, so that the pinned blocks exist.'''
    import torch
    from nutils_amd import device
    out = {}
    for timed in (False, True):
        wl = make_workload()
        wl.setup()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wl.build_pattern()
        wl.step(exchange=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if not copy:  # (configs[2]: 19.5 GB of CSR -- the copy is not timed, the size is reported)
            if timed:
                out = {'first_assembly_ms': (t1 - t0) * 1e3, 'host_bytes': int(8 * (2 * wl.values.numel() + wl.rowptr.numel()))}
            del wl
            continue
        host = [device.to_host(wl.values), device.to_host(wl.rowptr), device.to_host(wl.colidx)]
        t2 = time.perf_counter()
        wl.step(exchange=False)
        v = device.to_host(wl.values)
        t3 = time.perf_counter()
        if timed:
            out = {'first_assembly_ms': (t1 - t0) * 1e3, 'host_ready_ms': (t2 - t0) * 1e3, 'reassembly_host_ready_ms': (t3 - t2) * 1e3,
                   'host_bytes': int(sum(h.nbytes for h in host))}
        del wl, host, v
    return out


def cpu_baseline_c2(variant):
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
            'sample': f'{n}^3-element {variant} P1 hex Laplace, full COO->sort->CSR assembly (pattern + values, on the host), {t:.2f} s '
                      f'(element loop {tl:.2f} s on {cores} OpenMP threads, serial radix-sort dedup {td:.2f} s)'}


def cpu_baseline_c3(wl):
    '''Oracle port of the vector form (port_form3d): element loop on all cores, serial stable-sort dedup as in the reference.'''
    import numpy
    from oracle import assemble as oa, port
    if not port.available():
        return None
    cores = len(os.sched_getaffinity(0))
    pts, w = oa.gauss(4, 3)
    _, coeffs, _ = oa.structured_basis((1, 1, 1), 'std', 2)
    _, gcoeffs, _ = oa.structured_basis((1, 1, 1), 'std', 1)
    N, dN = oa.tabulate(coeffs[0], pts)
    gN, gdN = oa.tabulate(gcoeffs[0], pts)
    T = numpy.concatenate([N.T[:, :, None], dN.transpose(1, 0, 2)], axis=2)
    gT = numpy.concatenate([gN.T[:, :, None], gdN.transpose(1, 0, 2)], axis=2)

    def run(n):
        rng = numpy.random.default_rng(0)
        verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, ((n + 1) ** 3, 3))
        t0 = time.perf_counter()
        v, rp, ci, (tl, td) = port.form3d((n, n, n), 2, wl.C, T, gT, w, verts, threads=cores)
        return time.perf_counter() - t0, tl, td

    t8, _, _ = run(8)
    n = 8
    for cand in (12, 16, 20, 24):
        if t8 * (cand / 8) ** 3 <= 25.:
            n = cand
    t, tl, td = (t8, 0, 0) if n == 8 else run(n)
    return {'value': n ** 3 / t, 'unit': 'elements/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n}^3-element iso P2 vector elasticity, full COO->sort->CSR assembly (pattern + values, on the host), {t:.2f} s '
                      f'(element loop {tl:.2f} s on {cores} OpenMP threads, serial radix-sort dedup {td:.2f} s)'}


def row_sum_check(values, rowptr, a_row, b_row, world, dist, stride=1):
    '''max |sum of row| / max |K| over rows [a_row, b_row) (constants lie in the kernel of the Laplace stiffness matrix)'''
    import torch
    lo = int(rowptr[a_row])
    owned = values[lo:int(rowptr[b_row])]
    csum = torch.cat([torch.zeros(1, dtype=torch.float64, device=owned.device), torch.cumsum(owned, 0)])  # stays O(|K|): every row sums to ~0
    rowsum = csum[rowptr[a_row + 1:b_row + 1] - lo] - csum[rowptr[a_row:b_row] - lo]
    check = torch.stack([rowsum.abs().max() / owned.abs().max()])
    if world > 1:
        dist.all_reduce(check, op=dist.ReduceOp.MAX)
    return float(check.item())


def fail(msg, rank, world, dist, metric):
    print('ERROR: ' + msg, file=sys.stderr)
    if rank == 0:  # a broken kernel or halo reduce must not be recorded as a (possibly faster) valid result
        print(json.dumps({'metric': metric, 'value': None, 'unit': 'elements/s', 'n_gpus': world, 'error': 'correctness gate failed: ' + msg}))
    if world > 1:
        dist.destroy_process_group()
    sys.exit(1)


def make_workload(a, config, scaling, rank, world, halo=None):
    from nutils_amd import workloads
    halo = halo or a.halo
    n = a.n if config == a.config else (64 if config == 'c3' else 128)
    strong = scaling == 'strong' and world > 1
    if strong and n % world:
        raise SystemExit(f'--scaling strong: {n} element layers do not split into {world} slabs')
    layers = n // world if strong else n
    if config == 'c3':
        return lambda r=rank, w=world: workloads.ElasticityP2(n=n, layers=layers if w > 1 else n, rank=r, world=w, variant=a.variant, halo=halo)
    return lambda r=rank, w=world: workloads.PoissonSlab(n=n, layers=layers if w > 1 else n, rank=r, world=w, variant=a.variant, kernel=a.kernel, halo=halo)


def settle_and_time(wl, steps, warmup, settle, world, dist, use_graph):
    """Untimed settle steps (idle power state -> steady clocks), W warm-up steps, then EXACTLY `steps` timed steps."""
    import torch
    for _ in range(settle):
        wl.step()
    torch.cuda.synchronize()
    settle_extra = 0
    if world == 1 and settle:
        # keep settling while batches of steps still get faster (a box that has idled for long ramps its clocks over more than the fixed
        # number of steps: one bench run of round 2 read 0.193 ms where three runs on the next box read 0.168): at most 2 s
        prev, t_end = None, time.perf_counter() + 2.
        while time.perf_counter() < t_end:
            t1 = time.perf_counter()
            for _ in range(max(20, settle // 2)):
                wl.step()
            settle_extra += max(20, settle // 2)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            if prev is not None and dt > .99 * prev:
                break
            prev = dt
    for _ in range(warmup):
        wl.step()
    torch.cuda.synchronize()
    elapsed, kernel_ms, launch = timed_steps(wl, steps, world, dist, use_graph)
    return elapsed, kernel_ms, launch, settle + settle_extra


def c3_roofline(wl, kernel_ms, traffic):
    """configs[2]: the kernel writes 37 kB of CSR values per element once -- it is priced against the HBM peak (the write stream is its floor:
    9.7 GB / ~6 TB/s); the matrix-pipe share rides along."""
    bpe = wl.algorithmic_bytes_per_element()
    gbs = bpe * wl.nelems / (kernel_ms * 1e-3) / 1e9
    tf = wl.algorithmic_flops_per_element() * wl.nelems / (kernel_ms * 1e-3) / 1e12
    return {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'traffic': traffic,
            'kernel': wl.kernel_name, 'kernel_ms': kernel_ms, 'algorithmic_bytes_per_element': bpe,
            'mfma': {'achieved': tf, 'peak': F64_MFMA_PEAK_TF, 'unit': 'TFLOP/s', 'frac': tf / F64_MFMA_PEAK_TF,
                     'algorithmic_flops_per_element': wl.algorithmic_flops_per_element(), 'mfma_instructions_per_element': wl.mfma_per_element(),
                     'matrix_pipe_busy': wl.mfma_per_element() * wl.nelems * 64 / 1024 / (kernel_ms * 1e-3 * 2.4e9)}}


def main():
    a = parse()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: required by RCCL / cross-process device memory on this driver
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
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    metric = 'elements assembled/sec (global stiffness K)'
    from nutils_amd import workloads
    strong = a.scaling == 'strong' and world > 1
    make = make_workload(a, a.config, a.scaling, rank, world)
    wl = make()
    t0 = time.perf_counter()
    wl.setup()
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    wl.build_pattern()
    torch.cuda.synchronize()
    pattern_ms = (time.perf_counter() - t0) * 1e3
    elapsed, kernel_ms, launch, settled = settle_and_time(wl, a.steps, a.warmup, a.settle, world, dist, a.graph)
    launch_modes = dict(getattr(timed_steps, 'last_modes', {}))

    # size-independent check of the assembled matrix on every rank (rows it owns are complete after the interface reduce)
    wl.finish()
    checks = wl.check(world, dist)
    bad = [k for k, v in checks.items() if not v < 1e-10]
    if bad:
        fail(', '.join(f'{k} = {checks[k]:.3e}' for k in bad) + ': the assembled matrix is wrong', rank, world, dist, metric)

    other = other_halo = None
    if world > 1 and a.compare_halo:
        # the other halo mode of the same launch and scaling mode (secondary figure)
        h2 = 'reduce' if a.halo == 'recompute' else 'recompute'
        wh = make_workload(a, a.config, a.scaling, rank, world, halo=h2)()
        wh.setup()
        wh.build_pattern()
        elh, kmsh, launchh, _ = settle_and_time(wh, a.steps, a.warmup, min(a.settle, 50), world, dist, a.graph)
        wh.finish()
        ch = wh.check(world, dist)
        if not all(v < 1e-10 for v in ch.values()):
            fail(f'halo={h2}: the assembled matrix is wrong ({ch})', rank, world, dist, metric)
        other_halo = {'halo': h2, 'value': wh.nelems * world * a.steps / elh, 'unit': 'elements/s', 'ms_per_step': elh / a.steps * 1e3, 'kernel_ms': kmsh, 'launch': launchh,
                      'checks': ch, **getattr(timed_steps, 'last_modes', {})}
        del wh
        torch.cuda.empty_cache()
    if world > 1:
        # the other scaling mode of the same launch (secondary figure; fewer settle steps -- the clocks are up)
        mode2 = 'weak' if strong else 'strong'
        n = a.n
        if not (mode2 == 'strong' and n % world):
            nelems1, layers1, nnz1 = wl.nelems, wl.layers, wl.nnz
            del wl
            torch.cuda.empty_cache()
            w2 = make_workload(a, a.config, mode2, rank, world)()
            w2.setup()
            w2.build_pattern()
            el2, kms2, _, _ = settle_and_time(w2, a.steps, a.warmup, min(a.settle, 50), world, dist, a.graph)
            w2.finish()
            c2 = w2.check(world, dist)
            if not all(v < 1e-10 for v in c2.values()):
                fail(f'{mode2} scaling: the assembled matrix is wrong ({c2})', rank, world, dist, metric)
            other = {'scaling': mode2, 'value': w2.nelems * world * a.steps / el2, 'unit': 'elements/s', 'ms_per_step': el2 / a.steps * 1e3, 'kernel_ms': kms2,
                     'nelems_per_gpu': w2.nelems, 'checks': c2}
            wl = w2  # (only sizes of the primary run are reported below)
            wl_nelems, wl_layers, nnz = nelems1, layers1, nnz1
        else:
            wl_nelems, wl_layers, nnz = wl.nelems, wl.layers, wl.nnz
    else:
        wl_nelems, wl_layers, nnz = wl.nelems, wl.layers, wl.nnz

    if rank == 0:
        nelems_total = wl_nelems * world
        value = nelems_total * a.steps / elapsed
        bytes_per_elem = wl.algorithmic_bytes_per_element()
        gbs = bytes_per_elem * wl_nelems / (kernel_ms * 1e-3) / 1e9
        traffic = measured_traffic(wl.kernel_name, a.n) if world == 1 else None
        inrun = None
        if world == 1 and a.traffic == 'measure':
            probe = [sys.executable, 'tools/c3_bench.py', str(a.n), '3'] if a.config == 'c3' else [sys.executable, 'tools/c2_time.py', str(a.n), '30']
            if a.config == 'c3' or (a.variant == 'iso' and a.kernel in ('auto', 'fast')):
                inrun = measured_traffic_inrun(wl.kernel_name.split('<')[0], probe)
            if inrun is not None:
                traffic = inrun['bytes']
        if a.config == 'c3':
            roofline = c3_roofline(wl, kernel_ms, traffic)
            workload = f'3D linear elasticity stiffness, {a.n}^3 structured hex, p=2 vector basis (81 local dofs), 3x3x3 Gauss, {a.variant} geometry (BASELINE.json configs[2])'
        else:
            roofline = {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'traffic': traffic,
                        'kernel': wl.kernel_name, 'kernel_ms': kernel_ms, 'algorithmic_bytes_per_element': bytes_per_elem}
            if a.variant == 'iso' and a.kernel in ('auto', 'fast'):
                # second roofline of the same kernel: EXECUTED f64 arithmetic against the f64 vector peak.  Per element-thread the routine issues 1060 f64
                # instructions = 1489 flop (ISA of nh_p1hex_math.inc + the 36 entries: 301 mul, 330 add, 429 fma; tools/isa_hist.sh) plus 8 reciprocals; the
                # kernel computes every element of its 16 x 16 tiles incl. the one-element lateral halo and one extra layer per run of a workgroup.
                nb = -(-(a.n + 1) // 15)
                threads = nb * nb * 256 * (wl_layers + max(1, round(256 / (nb * nb))))
                tf = 1489. * threads / (kernel_ms * 1e-3) / 1e12
                roofline['f64_valu'] = {'achieved': tf, 'peak': F64_MFMA_PEAK_TF, 'unit': 'TFLOP/s', 'frac': tf / F64_MFMA_PEAK_TF, 'flop_per_element_thread': 1489,
                                        'f64_instructions_per_element_thread': 1060, 'element_threads': threads,
                                        'note': 'executed arithmetic incl. halo recompute (1.27 x the elements); counters: profiles/r04_c2_kernels.md section 2'}
            per = f'{wl_layers} x {a.n} x {a.n} per GPU' if world > 1 else f'{a.n}^3'
            workload = f'3D Poisson stiffness, {per} structured hex, p=1, 2x2x2 Gauss, {a.variant} geometry (BASELINE.json configs[1])'
        if inrun is not None:
            roofline['traffic_source'] = inrun['source']
            roofline['traffic_detail'] = {k: inrun[k] for k in ('fetch_bytes', 'write_bytes', 'bytes_fetch_doubled')}
        elif traffic is not None:
            roofline['traffic_source'] = TRAFFIC_FILE + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command, separate passes; not measured in this run)'
        out = {
            'metric': metric, 'value': value, 'unit': 'elements/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True, 'scaling': a.scaling,  # (one GPU: both modes coincide; the series N = 1, 2, 4, 8 carries one label)
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': workload, 'nelems_per_gpu': wl_nelems, 'nnz_per_gpu': nnz, 'kernel': wl.kernel_name,
                       'parallelism': (f'element slabs x{world}, ' + ('one ghost element layer recomputed per rank, no exchange' if a.halo == 'recompute' else 'halo-plane reduce (RCCL point to point)'))
                       if world > 1 else 'single GPU', 'halo': a.halo if world > 1 else None, 'launch': launch,
                       'settle_steps': settled, 'timed_region': 'device-resident re-assembly of the CSR values (pattern, tables, vertices in HBM)'},
            'roofline': roofline, 'pattern_ms': pattern_ms, 'setup_s': setup_s, 'checks': checks,
        }
        out.update(launch_modes)
        if other is not None:
            out[other['scaling']] = other
        if other_halo is not None:
            out['halo_' + other_halo['halo']] = other_halo
        if world == 1:
            del wl
            torch.cuda.empty_cache()
            out.update(host_regions(lambda: make(0, 1), copy=a.config == 'c2'))
        if world == 1 and a.variant == 'iso' and a.config == 'c2':
            # secondary variant of the same config: exact-uniform mesh (what mesh.rectilinear gives the reference; there the
            # element matrix is hoisted out of the loop and the work is index generation + dedup).  Not the headline value.
            w2 = workloads.PoissonSlab(n=a.n, rank=0, world=1, variant='uniform', kernel=a.kernel)
            w2.setup()
            w2.build_pattern()
            for _ in range(a.settle + a.warmup):
                w2.step()
            torch.cuda.synchronize()
            el2, kms, l2 = timed_steps(w2, a.steps, 1, None, a.graph)
            b2 = w2.algorithmic_bytes_per_element()
            ach = b2 * w2.nelems / (kms * 1e-3) / 1e9
            out['variants'] = {'uniform': {'value': w2.nelems * a.steps / el2, 'unit': 'elements/s', 'ms_per_step': el2 / a.steps * 1e3, 'kernel': w2.kernel_name, 'launch': l2,
                                           'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                                                        'traffic': measured_traffic(w2.kernel_name, a.n),
                                                        'kernel_ms': kms, 'algorithmic_bytes_per_element': b2}}}
            del w2
            torch.cuda.empty_cache()
            if a.kernel == 'auto':
                # the same config through the GENERIC entry (any mesh, ragged bases): thread-per-element pass + owner-side reduction
                # (NH_MATRIX_GATHER | NH_MATRIX_STORE, profiles/r02_generic_gather.md).  Not the headline value either.
                w3 = workloads.PoissonSlab(n=a.n, rank=0, world=1, variant='iso', kernel='gather')
                w3.setup()
                w3.build_pattern()
                for _ in range(20):
                    w3.step()
                torch.cuda.synchronize()
                nst = max(10, a.steps // 4)
                el3, kms3, l3 = timed_steps(w3, nst, 1, None, a.graph)  # (same launch mode as the headline)
                b3 = w3.algorithmic_bytes_per_element()
                out['variants']['generic_gather'] = {'value': w3.nelems * nst / el3, 'unit': 'elements/s', 'ms_per_step': el3 / nst * 1e3, 'kernel': w3.kernel_name,
                                                     'launch': l3, 'kernel_ms': kms3, 'hbm_frac': b3 * w3.nelems / (kms3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                     'algorithmic_bytes_per_element': b3, 'traffic': measured_traffic(w3.kernel_name, a.n),
                                                     'note': 'nh_assemble_matrix with NH_MATRIX_GATHER | NH_MATRIX_STORE; bit-reproducible'}
                del w3
                torch.cuda.empty_cache()
                # ... and through the owner blocks of NH_MATRIX_FUSED: one pass, no scratch, no global atomics, ordered sums (the default of the any-mesh entry for this block)
                w5 = workloads.PoissonSlab(n=a.n, rank=0, world=1, variant='iso', kernel='fused')
                w5.setup()
                w5.build_pattern()
                for _ in range(20):
                    w5.step()
                torch.cuda.synchronize()
                el5, kms5, l5 = timed_steps(w5, nst, 1, None, a.graph)
                out['variants']['generic_fused'] = {'value': w5.nelems * nst / el5, 'unit': 'elements/s', 'ms_per_step': el5 / nst * 1e3, 'kernel': w5.kernel_name,
                                                    'launch': l5, 'kernel_ms': kms5, 'hbm_frac': b3 * w5.nelems / (kms5 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                    'algorithmic_bytes_per_element': b3, 'traffic': measured_traffic(w5.kernel_name, a.n),
                                                    'owner_blocks': dict(zip(('blocks', 'rows_per_block', 'element_visits'), w5.pattern.fused_info())),
                                                    'note': 'nh_assemble_matrix with NH_MATRIX_FUSED | NH_MATRIX_STORE; ordered rounds: bit-reproducible, bit-identical to the gather'}
                del w5
                torch.cuda.empty_cache()
            if not a.no_c3 and a.n == 128:
                # BASELINE.json configs[2] in the same run: 64^3 P2 vector elasticity through nh_p2hex_matrix (its own line: --config c3)
                w4 = workloads.ElasticityP2(n=64, rank=0, world=1, variant='iso')
                w4.setup()
                w4.build_pattern()
                el4, kms4, l4, _ = settle_and_time(w4, 10, 3, 10, 1, None, a.graph)
                c4 = w4.check(1, None)
                v4 = {'value': w4.nelems * 10 / el4, 'unit': 'elements/s', 'steps': 10, 'ms_per_step': el4 / 10 * 1e3, 'kernel': w4.kernel_name, 'launch': l4,
                      'workload': '3D linear elasticity stiffness, 64^3 structured hex, p=2 vector basis, 3x3x3 Gauss, iso geometry (BASELINE.json configs[2])',
                      'roofline': c3_roofline(w4, kms4, measured_traffic(w4.kernel_name, 64)), 'checks': c4}
                if not all(x < 1e-10 for x in c4.values()):
                    v4['error'] = 'correctness gate failed'
                del w4
                torch.cuda.empty_cache()
                if not a.no_cpu:
                    cb3 = cpu_baseline_c3(workloads.ElasticityP2(n=64, rank=0, world=1, variant='iso'))
                    if cb3:
                        v4['cpu_baseline'] = cb3
                        v4['speedup_vs_cpu_port'] = v4['value'] / cb3['value']
                out['variants']['c3'] = v4
                # the same matrix on UNIFORM cells (mesh.rectilinear with equidistant vertices, what examples/elasticity.py builds): all element matrices equal, the rows of
                # the 2 x 2 x 2 mesh replicated by node class -- a write stream (nh_p2hex_rows_uniform)
                w6 = workloads.ElasticityP2(n=64, rank=0, world=1, variant='uniform')
                w6.setup()
                w6.build_pattern()
                el6, kms6, l6, _ = settle_and_time(w6, 10, 3, 10, 1, None, a.graph)
                c6 = w6.check(1, None)
                b6 = w6.algorithmic_bytes_per_element()
                v6 = {'value': w6.nelems * 10 / el6, 'unit': 'elements/s', 'steps': 10, 'ms_per_step': el6 / 10 * 1e3, 'kernel': w6.kernel_name, 'launch': l6,
                      'workload': '3D linear elasticity stiffness, 64^3 structured hex, p=2 vector basis, 3x3x3 Gauss, uniform cells (BASELINE.json configs[2] on the geometry of examples/elasticity.py)',
                      'roofline': {'bound': 'hbm', 'achieved': b6 * w6.nelems / (kms6 * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                   'frac': b6 * w6.nelems / (kms6 * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': measured_traffic(w6.kernel_name, 64), 'kernel': w6.kernel_name,
                                   'kernel_ms': kms6, 'algorithmic_bytes_per_element': b6}, 'checks': c6}
                if not all(x < 1e-10 for x in c6.values()):
                    v6['error'] = 'correctness gate failed'
                del w6
                torch.cuda.empty_cache()
                out['variants']['c3_uniform'] = v6
                # BASELINE.json configs[3] and the configs[4] class: the probes of tools/ in their own processes (their parity checks run there); a failure drops the entry
                for key, cmd in (('c4', [sys.executable, 'tools/c4_step.py', '512']), ('c5', [sys.executable, 'tools/ragged_probe.py', '256', '10']),
                                 ('vector_any_mesh', [sys.executable, 'tools/vector_probe.py', '96', '10'])):
                    try:
                        r = subprocess.run(cmd, capture_output=True, text=True, timeout=180, cwd=os.path.dirname(os.path.abspath(__file__)))
                        line = next(l for l in r.stdout.splitlines() if l.startswith('RESULT '))
                        out['variants'][key] = json.loads(line[7:])
                        if key == 'c5':  # (PMC traffic of the whole step -- element kernels of both size classes, gather, mirror: static, profiles/r06_c5_gram.md)
                            out['variants'][key]['traffic'] = measured_traffic('c5 step: k_gram_sym (both size classes) + k_gather_values_2x2_tri + k_mirror_2x2', 256)
                    except Exception as e:  # noqa: BLE001 (secondary figures)
                        out['variants'][key] = {'error': f'{type(e).__name__}: {e}'[:200]}
        if not a.no_cpu and world == 1:
            cb = cpu_baseline_c3(make(0, 1)) if a.config == 'c3' else cpu_baseline_c2(a.variant)
            if cb:
                out['cpu_baseline'] = cb
                nel = a.n ** 3
                out['speedup_vs_cpu_port'] = {'device_resident_reassembly': value / cb['value'],
                                              'first_assembly_device_resident': nel / (out['first_assembly_ms'] * 1e-3) / cb['value'],
                                              'note': 'the CPU leg times pattern + values on the host (BASELINE.md 3)'}
                if 'host_ready_ms' in out:  # the like-for-like ratio: pattern + values + CSR in host memory on both sides
                    out['speedup_vs_cpu_port']['first_assembly_host_ready'] = nel / (out['host_ready_ms'] * 1e-3) / cb['value']
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
