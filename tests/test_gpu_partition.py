'''Two slabs of the multi-GPU partition simulated on ONE GPU (sequentially): each "rank" assembles its
slab with the real kernels, the interface-plane reduce is applied with the HaloPlan indices (without the
network hop), and the concatenated owned row blocks must equal the single-mesh assembly.'''
import numpy
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('halo', ['recompute', 'reduce'])
@pytest.mark.parametrize('kernel', ['fast', 'generic'])
@pytest.mark.parametrize('variant', ['iso', 'uniform'])
def test_two_slabs_equal_single_mesh(kernel, variant, halo, monkeypatch):
    '''halo='reduce': the interface rows travel as HaloPlan prescribes; halo='recompute': every rank assembles its ghost layer as well and writes
    only the rows it owns -- nothing travels (the default of the workloads).  Either way the concatenated blocks are the single-mesh matrix.'''
    monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')  # the single-mesh reference goes through the generic kernel
    from nutils_amd import workloads, partition, device
    n, world = 9, 3
    blocks = []
    prev_tail = None
    for rank in range(world):
        wl = workloads.PoissonSlab(n=n, rank=rank, world=world, variant=variant, kernel=kernel, halo=halo)
        wl.setup()
        wl.build_pattern()
        if halo == 'recompute':
            wl.values.fill_(float('nan'))  # (the owned rows are all written)
        wl.step(exchange=False)
        assert (wl.halo is None) == (halo == 'recompute')
        if wl.slab.recvs:  # what irecv + index_add_ do in HaloPlan.exchange
            assert prev_tail.numel() == wl.halo.recv_buf.numel()
            wl.values.index_add_(0, wl.halo.recv_idx, prev_tail)
        if wl.slab.sends:
            prev_tail = wl.values[wl.halo.send_a:wl.halo.send_b].clone()
        blocks.append(wl.owned_csr())
    v, rp, ci = partition.concatenate(blocks)
    # single mesh (n*world) x n x n through the generic path
    from nutils_amd import mesh, function
    domain, geom = mesh.rectilinear([n * world, n, n])
    basis = domain.basis('std', degree=1)
    if variant == 'iso':
        rng = numpy.random.default_rng(0)
        verts = numpy.stack(numpy.meshgrid(numpy.arange(n * world + 1.), numpy.arange(n + 1.), numpy.arange(n + 1.), indexing='ij'), -1) \
            + rng.uniform(-.2, .2, (n * world + 1, n + 1, n + 1, 3))
        geom = basis @ verts.reshape(-1, 3)
    K = domain.integral(function.outer(function.grad(basis, geom)).sum(-1) * function.J(geom), degree=2)
    vo, rpo, cio = function.eval(function.as_csr(K))
    assert numpy.array_equal(rp, rpo) and numpy.array_equal(ci, cio)
    assert numpy.abs(v - vo).max() <= 1e-13 * numpy.abs(vo).max()


@pytest.mark.parametrize('halo', ['recompute', 'reduce'])
@pytest.mark.parametrize('world,layers', [(2, 3), (3, 2)])
def test_p2_elasticity_slabs_equal_single_mesh(world, layers, halo):
    '''configs[2] partitioned: every "rank" assembles its slab of the quadratic vector elasticity matrix with the HIP kernel
    (nh_p2hex_matrix with layer / owner ranges), the rows of the interface plane travel as HaloPlan prescribes, and the concatenated
    owned row blocks equal the single-mesh assembly ENTRY BY ENTRY (index arrays bit-exact).'''
    from nutils_amd import workloads, partition
    n = 4
    blocks, prev_tail = [], None
    for rank in range(world):
        wl = workloads.ElasticityP2(n=n, layers=layers, rank=rank, world=world, halo=halo)
        wl.setup()
        wl.build_pattern()
        wl.step(exchange=False)
        assert (wl.halo is None) == (halo == 'recompute')
        if wl.slab.recvs:  # what irecv + index_add_ do in HaloPlan.exchange
            assert prev_tail.numel() == wl.halo.recv_buf.numel()
            wl.values.index_add_(0, wl.halo.recv_idx, prev_tail)
        if wl.slab.sends:
            prev_tail = wl.values[wl.halo.send_a:wl.halo.send_b].clone()
        blocks.append(wl.owned_csr())
        assert wl.check()['owned_row_sums_rel'] < 1e-12
    v, rp, ci = partition.concatenate(blocks)
    from nutils_amd import device
    single = workloads.ElasticityP2(n=n, layers=layers * world, rank=0, world=1)
    single.setup()
    single.build_pattern()
    single.step()
    vo, rpo, cio = device.to_host(single.values), device.to_host(single.rowptr), device.to_host(single.colidx)
    assert numpy.array_equal(rp, rpo) and numpy.array_equal(ci, cio)
    assert numpy.abs(v - vo).max() <= 1e-13 * numpy.abs(vo).max()
    # a misplaced interface row would also show here: the six rigid body modes lie in the kernel of the global matrix
    import scipy.sparse
    K = scipy.sparse.csr_matrix((v, ci, rp), shape=(len(rp) - 1,) * 2)
    X = numpy.stack(numpy.meshgrid(numpy.arange(2 * layers * world + 1.), numpy.arange(2 * n + 1.), numpy.arange(2 * n + 1.), indexing='ij'), -1).reshape(-1, 3)
    assert abs(K @ numpy.ones(K.shape[0])).max() < 1e-12 * abs(v).max() * 100


def test_pipelined_exchange_matches_serial():
    '''The double-buffered, side-stream exchange of PoissonSlab (world > 1) orchestrated on ONE GPU: the network hop is replaced
    by a local stand-in with the same stream semantics (reads the send range, adds into the receive rows on the current
    stream); several pipelined steps must leave exactly what serial steps leave, in both value arrays.'''
    import torch
    from nutils_amd import workloads

    class LocalHalo:
        def __init__(self, wl):
            self.a, self.b = wl.nnz // 3, wl.nnz // 3 + 1000
            self.buf = torch.empty(self.b - self.a, dtype=torch.float64, device=wl.values.device)
            self.idx = torch.arange(0, 2 * (self.b - self.a), 2, device=wl.values.device)
            self.calls = 0

        def exchange(self, values):
            self.calls += 1
            self.buf.copy_(values[self.a:self.b])  # "irecv" of what the neighbour "isend"s
            for _ in range(20):                    # keep the side stream busy for a while
                self.buf.mul_(1.0)
            values.index_add_(0, self.idx, self.buf)

    def run(pipelined):
        wl = workloads.PoissonSlab(n=24, rank=0, world=1, variant='iso')
        wl.setup()
        wl.build_pattern()
        wl.halo = LocalHalo(wl)
        if pipelined:
            wl.enable_pipeline()
        outs = []
        for i in range(5):
            wl.step()
        wl.finish()
        torch.cuda.synchronize()
        assert wl.halo.calls == 5
        arrays = wl._vals if pipelined else [wl.values]
        return [a.clone() for a in arrays], wl.values.clone()

    (sa,), slast = run(False)
    (pa, pb), plast = run(True)
    scale = float(sa.abs().max())
    for got in (pa, pb, plast):
        assert float((got - sa).abs().max()) <= 1e-13 * scale
    assert plast.data_ptr() != 0


@pytest.mark.parametrize('world', [2, 3, 8])  # (8: the driver's widest launch, `--gpus 8`, eight processes on the one GPU of this box)
def test_bench_multiprocess_on_one_gpu(world):
    '''`bench.py --gpus N` exactly as the driver launches it (torch.distributed.run, one process per rank), except that all ranks
    share the one GPU of this box and the interface rows travel through gloo (host staging) instead of RCCL: exercises rank
    bookkeeping, ghost layers, the pipelined exchange, the timing protocol and the owned-row-sum check of every rank.'''
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NUTILS_AMD_BENCH_ONE_GPU='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1', '--master-port', str(29500 + world),
           'bench.py', '--gpus', str(world), '--steps', '4', '--warmup', '2', '--elements-per-axis', '32', '--no-cpu', '--compare-halo']
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    # default mode: the ONE mesh split into slabs when the layers divide evenly -- the line then also carries the weak figure
    strong = 32 % world == 0
    assert rec['n_gpus'] == world and rec['steps'] == 4 and rec['scaling'] == ('strong' if strong else 'weak')
    assert rec['config']['nelems_per_gpu'] == (32 ** 3 // world if strong else 32 ** 3) and rec['value'] > 0
    assert rec['checks']['owned_row_sums_rel'] < 1e-12
    # default halo mode: one ghost layer recomputed per rank, no exchange, the step captured in a HIP graph on every rank; the line carries the
    # figure of the RCCL-style reduce of the interface rows as well
    assert rec['config']['halo'] == 'recompute' and 'hipGraph' in rec['config']['launch'] and 'ms_per_step_graph' in rec and 'ms_per_step_eager' in rec
    assert rec['halo_reduce']['value'] > 0 and rec['halo_reduce']['checks']['owned_row_sums_rel'] < 1e-12 and rec['halo_reduce']['launch'] == 'eager'
    if strong:
        assert rec['weak']['nelems_per_gpu'] == 32 ** 3 and rec['weak']['value'] > 0 and rec['weak']['checks']['owned_row_sums_rel'] < 1e-12
    assert 'WARNING' not in out.stderr


@pytest.mark.parametrize('args,nel', [(['--scaling', 'weak', '--elements-per-axis', '32'], 32 ** 3),
                                      (['--config', 'c3', '--scaling', 'weak', '--elements-per-axis', '8'], 8 ** 3),
                                      (['--config', 'c3', '--elements-per-axis', '8'], 8 ** 3 // 2)])
def test_bench_modes_multiprocess_on_one_gpu(args, nel):
    '''strong scaling (one mesh split into slabs) and the configs[2] workload, two ranks on the one GPU of this box'''
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NUTILS_AMD_BENCH_ONE_GPU='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1', '--master-port', '29511',
           'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--settle', '2', '--no-cpu'] + args
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert rec['n_gpus'] == 2 and rec['scaling'] == ('weak' if '--scaling' in args else 'strong')
    assert ('strong' if '--scaling' in args else 'weak') in rec  # the other mode rides along
    assert rec['config']['nelems_per_gpu'] == nel and rec['value'] > 0
    assert rec['checks']['owned_row_sums_rel'] < 1e-12


@pytest.mark.parametrize('halo', ['reduce', 'recompute'])
@pytest.mark.parametrize('shape,degree,ncomp,world', [([7, 6, 5], 1, 1, 3), ([10, 9], 2, 2, 4)])
def test_element_partition_of_an_unstructured_mesh(shape, degree, ncomp, world, halo):
    '''partition.ElementPartition on an element list in shuffled order (what an imported unstructured mesh gives): every "rank" assembles its local mesh -- own + ghost elements,
    GLOBAL dof numbers -- with the real kernels through the front end; with halo='reduce' the ghosts contribute structural zeros (a 0 / 1 coefficient per element) and the partial
    shared rows travel as SharedRowPlan prescribes (nh_index_copy to pack, nh_monomial with an output index to add; the network hop is left out), with halo='recompute' the ghosts
    are assembled as well.  The owners' rows merged in row order equal the single-mesh matrix: index arrays bit-exact, values to 1e-13.'''
    import torch
    from nutils_amd import mesh, function, topology, partition, device
    rng = numpy.random.default_rng(4)
    axes = [numpy.cumsum(numpy.r_[0., rng.uniform(.5, 1.5, n)]) for n in shape]
    sdomain, sgeom = mesh.rectilinear(axes)
    sbasis = sdomain.basis('std', degree=degree)
    perm = rng.permutation(len(sdomain))
    origin, size = sgeom.element_boxes()
    coeffs, dofs = [sbasis.get_coefficients(e) for e in perm], [numpy.asarray(sbasis.get_dofs(e)) for e in perm]
    nd = len(shape)

    def matrix(elements, live):
        topo = topology.ElementList(origin[perm][elements], size[perm][elements])
        basis = topo.plain_basis([coeffs[e] for e in elements], [dofs[e] for e in elements], sbasis.ndofs)
        smp = topo.sample('gauss', 2 * degree)
        if ncomp == 1:
            itg = function.outer(function.grad(basis, topo.geom)).sum(-1) + .3 * function.outer(basis)
            K = smp.integral(itg * function.J(topo.geom) * function.PointTable(numpy.repeat(live.astype(float), smp.points.npoints).reshape(len(elements), -1)))
        else:
            u, v = function.field('u', basis, shape=[ncomp]), function.field('v', basis, shape=[ncomp])
            sig = 1.3 * function.div(u, topo.geom) * function.eye(nd) + 2 * function.symgrad(u, topo.geom)
            res = smp.integral(function.inner(function.symgrad(v, topo.geom), sig) * function.J(topo.geom)
                               * function.PointTable(numpy.repeat(live.astype(float), smp.points.npoints).reshape(len(elements), -1)))
            K = function.derivative(function.derivative(res, 'v'), 'u')
        return function.eval(function.as_csr(K))

    offsets = numpy.cumsum([0] + [len(d) for d in dofs])
    part = partition.ElementPartition(offsets, numpy.concatenate(dofs), sbasis.ndofs, world, ncomp=ncomp, halo=halo)
    plans, vals = [], []
    for rank in range(world):
        el, live = part.local_elements(rank)
        v, rp, ci = matrix(el, live)
        plans.append(partition.SharedRowPlan(part, rank, rp, ci))
        vals.append(device.to_dev(v, 'float64'))
    offers = [p.offer() for p in plans]
    for p in plans:
        p.accept(offers)
    assert any(p.send for p in plans) == (halo == 'reduce')
    for dst in range(world):  # what batch_isend_irecv moves in SharedRowPlan.exchange; the owner adds its sources in rank order
        for src in sorted(plans[dst].recv):
            plans[dst].add(vals[dst], src, plans[src].pack(vals[src], dst))
    torch.cuda.synchronize()
    v, rp, ci = partition.merge_rows([p.owned_block(x) for p, x in zip(plans, vals)], sbasis.ndofs * ncomp)
    ne = len(perm)
    vo, rpo, cio = matrix(numpy.arange(ne), numpy.ones(ne, dtype=bool))
    assert numpy.array_equal(rp, rpo) and numpy.array_equal(ci, cio)
    assert numpy.abs(v - vo).max() <= 1e-13 * numpy.abs(vo).max()


@pytest.mark.parametrize('halo', ['reduce', 'recompute'])
def test_element_partition_multiprocess_on_one_gpu(halo, tmp_path):
    '''tools/partition_run.py as `torch.distributed.run` launches it -- three processes, all on the one GPU of this box, shared rows through gloo (host staging) --:
    setup (all_gather_object of the column lists), pack / send / receive / add, owned blocks; merged in row order they equal the single-process matrix of the same
    script (index arrays exact, values 1e-13).'''
    import os, subprocess, sys
    from nutils_amd import partition
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NUTILS_AMD_BENCH_ONE_GPU='1', MASTER_ADDR='127.0.0.1')
    outs = {}
    for world in (3, 1):
        d = tmp_path / f'w{world}'
        d.mkdir()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1', '--master-port', str(29520 + world),
               'tools/partition_run.py', '--halo', halo, '--shape', '7,6,5', '--out', str(d)]
        out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
        blocks = []
        for r in range(world):
            z = numpy.load(d / f'rows{r}.npz')
            blocks.append((z['rows'], z['lens'], z['cols'], z['vals']))
        outs[world] = partition.merge_rows(blocks, 8 * 7 * 6 * 3)
        if world == 3:
            assert ('entries sent' in out.stdout) and ((' 0 of ' in out.stdout) == (halo == 'recompute') or halo == 'reduce')
    v, rp, ci = outs[3]
    vo, rpo, cio = outs[1]
    assert numpy.array_equal(rp, rpo) and numpy.array_equal(ci, cio)
    assert numpy.abs(v - vo).max() <= 1e-13 * numpy.abs(vo).max()
