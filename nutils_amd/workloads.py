'''Synthetic workloads of BASELINE.json, set up through the product path and kept
resident in HBM, for bench.py and the large-size GPU tests.

PoissonSlab = configs[1]: 3-D Poisson stiffness on a structured hex mesh, p=1.
With world > 1 the global mesh is (n*world) x n x n; rank r owns element layers
[r*n, (r+1)*n) and the dof planes [r*n, (r+1)*n) (the last rank also the final
plane); see nutils_amd/partition.py for the exchange.
'''

import os

import numpy

from . import device, function, kernels, mesh, partition


class PoissonSlab:

    def __init__(self, n=128, rank=0, world=1, variant='iso', kernel='auto', seed=0, layers=None, halo='recompute'):
        self.n, self.rank, self.world, self.variant = int(n), int(rank), int(world), variant
        self.layers = int(layers) if layers else self.n  # element layers per rank (strong scaling: n / world)
        self.kernel = kernel
        self.seed = seed
        # halo='recompute': the ghost layer below is assembled too and only owned rows are written -- no exchange (partition.py)
        self.slab = partition.Slab(self.layers, self.rank, self.world, shape_jk=(self.n, self.n), halo=halo)

    def setup(self):
        s = self.slab
        n = self.n
        self.domain, geom0 = mesh.rectilinear([s.local_layers, n, n])
        self.basis = self.domain.basis('std', degree=1)
        if self.variant == 'iso':
            # global vertex field: (I, J, K) + U(-.2, .2)^3 with default_rng(seed) over the GLOBAL mesh;
            # every rank draws the same stream and keeps its planes
            rng = numpy.random.default_rng(self.seed)
            nI = self.layers * self.world + 1
            pert = rng.uniform(-.2, .2, (nI, n + 1, n + 1, 3))
            I0 = s.first_global_plane
            idx = numpy.stack(numpy.meshgrid(numpy.arange(I0, I0 + s.local_layers + 1, dtype=float), numpy.arange(n + 1.), numpy.arange(n + 1.), indexing='ij'), -1)
            verts = (idx + pert[I0:I0 + s.local_layers + 1]).reshape(-1, 3)
            self.verts = verts
            self.geom = self.basis @ verts
        else:
            self.verts = None
            self.geom = function.RectilinearGeometry(self.domain, [float(s.first_global_plane), 0., 0.], [1., 1., 1.])
        self.smp = self.domain.sample('gauss', 2)
        self.tables = self.smp.tables(self.basis)
        self.kgeom = self.smp.geometry(self.geom)
        self.nelems = s.own_layers * n * n
        self.C = numpy.zeros((1, 4, 1, 4))
        for i in range(3):
            self.C[0, 1 + i, 0, 1 + i] = 1.

    @property
    def fast(self):
        return self.kernel in ('auto', 'fast')

    def build_pattern(self):
        s = self.slab
        if self.fast:
            # closed-form structured pattern (K6 without sorting), write-once kernel
            self.pattern = None
            self.rowptr, self.colidx = kernels.p1hex_pattern((s.local_layers, self.n, self.n))
            self.kernel_name = 'k_p1hex_skew<16,16,false,false>' if self.variant == 'iso' else 'k_p1hex_uniform'
        else:
            self.pattern = self.smp.pattern(self.basis, self.basis)
            self.rowptr, self.colidx = self.pattern.expand()
            self.kernel_name = {'batched': 'k_mterms<3>', 'gather': 'k_local_scalar<3,8,8> + k_gather_values', 'fused': 'k_fused_p1hex<false>'}.get(self.kernel, 'k_matrix_generic<3>')
        self.values = device.zeros(self.colidx.numel(), 'float64')  # rows of a ghost plane are never written: keep them zero
        self.nnz = int(self.colidx.numel())
        self.halo = partition.HaloPlan(self.slab, self.rowptr) if self.world > 1 and self.slab.halo == 'reduce' else None
        # with an exchange in flight next to the kernel, 8 CUs are left to the RCCL send/recv workgroups: a marching workgroup takes
        # the whole LDS of its CU, so on a fully occupied chip the transfer could only start when the assembly has finished
        self._max_wg = 0
        if self.halo is not None:
            import torch
            self._max_wg = max(1, torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count - 8)
        self._vals = None
        if self.halo is not None and not os.environ.get('NUTILS_AMD_SERIAL_EXCHANGE'):
            self.enable_pipeline()
        self._verts_dev = device.to_dev(self.verts, 'float64') if self.verts is not None else None
        self._ke = None
        if self.fast and self.verts is None:  # uniform mesh: the element matrix is a per-mesh constant (hoisted, as in the reference)
            self._ke = kernels.p1hex_unit_matrix(shape=(s.local_layers, self.n, self.n), gauss_x=self._gauss_x1(), gauss_w=self._gauss_w1())

    def enable_pipeline(self):
        '''Overlap the interface-plane reduce of step i with the assembly of step i+1: consecutive steps write alternating value
        arrays, the exchange (isend/irecv + index_add_) runs on a side stream behind an event of the kernel that produced its
        array, and a kernel only reuses an array once the exchange that read it has finished.  xGMI is point to point and the 2.4 MB
        message costs tens of microseconds next to a 0.25 ms kernel -- hidden completely instead of added to every step.'''
        import torch
        self._vals = [self.values, torch.zeros_like(self.values)]
        self._comm_stream = torch.cuda.Stream()
        self._comm_done = [None, None]
        self._comm_ready, self._comm_fin = [None, None], [None, None]
        self._it = 0

    def _begin_step(self, exchange):
        '''Select the value array of this step; returns the pipeline slot or None (serial exchange).'''
        if self._vals is None or not exchange or self.halo is None:
            return None
        import torch
        slot = self._it & 1
        self._it += 1
        self.values = self._vals[slot]
        if self._comm_done[slot] is not None:
            torch.cuda.current_stream().wait_event(self._comm_done[slot])
        return slot

    def _end_step(self, slot, exchange):
        if self.halo is None or not exchange:
            return
        if slot is None:
            self.halo.exchange(self.values)
            return
        import torch
        if self._comm_ready[slot] is None:  # events are re-recorded every step (a wait refers to the record that precedes it)
            self._comm_ready[slot], self._comm_fin[slot] = torch.cuda.Event(), torch.cuda.Event()
        ready, done = self._comm_ready[slot], self._comm_fin[slot]
        ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm_stream):
            self._comm_stream.wait_event(ready)
            self.halo.exchange(self.values)
            done.record(self._comm_stream)
        self._comm_done[slot] = done

    def finish(self):
        '''Make the current stream wait for the exchanges still in flight (before the values are read).'''
        if self._vals is not None:
            import torch
            for ev in self._comm_done:
                if ev is not None:
                    torch.cuda.current_stream().wait_event(ev)

    def _own_views(self):
        '''Structures restricted to the rank's own element layers (skip the ghost layer).'''
        e0 = self.slab.value_layers[0] * self.n * self.n  # (halo='recompute': the ghost layer is assembled as well)
        if not hasattr(self, '_views'):
            t = self.tables
            test = kernels.basis(t.T, t.dofs[e0 * 8:], nb=8)
            if self.variant == 'iso':
                g = kernels.geometry_iso(8, t.T, t.dofs[e0 * 8:], self.kgeom._keep[2])
            else:
                g = kernels.geometry_box(self.kgeom._keep[0].reshape(-1)[e0 * 3:], self.kgeom._keep[1].reshape(-1)[e0 * 3:])
            self._views = test, g, e0
        return self._views

    def step(self, kernel_events=None, exchange=True):
        slot = self._begin_step(exchange)
        if self.fast:
            s = self.slab
            if getattr(self, '_launch', None) is None:
                self._launch = kernels.P1HexLaplace(shape=(s.local_layers, self.n, self.n), gauss_x=self._gauss_x1(), gauss_w=self._gauss_w1(), verts=self._verts_dev,
                                                    origin=(float(s.first_global_plane), 0., 0.), layers=s.value_layers,
                                                    planes=s.written_planes, unit_matrix=self._ke, max_workgroups=self._max_wg)
            if kernel_events:
                kernel_events[0].record()
            self._launch(self.values)
            if kernel_events:
                kernel_events[1].record()
            self._end_step(slot, exchange)
            return
        test, g, e0 = self._own_views()
        nelems = (self.slab.value_layers[1] - self.slab.value_layers[0]) * self.n * self.n  # elements assembled (self.nelems: elements owned)
        gather = self.kernel == 'gather' and e0 == 0 and nelems == self.pattern.nelems
        fused = self.kernel == 'fused' and e0 == 0 and nelems == self.pattern.nelems
        if not gather and not fused:
            self.values.zero_()  # (the gather and fused paths store every entry: NH_MATRIX_STORE)
        if kernel_events:
            kernel_events[0].record()
        if self.kernel == 'batched' and e0 == 0:
            kernels.assemble_matrix_terms(nelems=nelems, ndims=3, nq=8, weights=self.smp._weights_dev, geom=g, test=test, trial=test, nct=1, ncr=1,
                                          mask=None, pattern=self.pattern, values=self.values, terms=[dict(C=self.C)])
        else:
            kernels.assemble_matrix(nelems=nelems, ndims=3, nq=8, weights=self.smp._weights_dev, geom=g, test=test, trial=test, nct=1, ncr=1,
                                    C=self.C, mask=None, pattern=self.pattern, values=self.values, emap_offset=e0 * 64,
                                    gather=gather, fused=fused, store=gather or fused)
        if kernel_events:
            kernel_events[1].record()
        self._end_step(slot, exchange)

    def check(self, world=1, dist=None):
        '''Size-independent check on the rows this rank owns (complete after the interface reduce): K 1 = 0.'''
        import torch
        s = self.slab
        a_row, b_row = s.own_plane_begin * s.plane, s.own_plane_end * s.plane
        lo = int(self.rowptr[a_row])
        owned = self.values[lo:int(self.rowptr[b_row])]
        csum = torch.cat([torch.zeros(1, dtype=torch.float64, device=owned.device), torch.cumsum(owned, 0)])  # stays O(|K|): every row sums to ~0
        rowsum = csum[self.rowptr[a_row + 1:b_row + 1] - lo] - csum[self.rowptr[a_row:b_row] - lo]
        check = torch.stack([rowsum.abs().max() / owned.abs().max()])
        if world > 1:
            dist.all_reduce(check, op=dist.ReduceOp.MAX)
        return {'owned_row_sums_rel': float(check.item())}

    def _gauss_w1(self):
        from . import points
        return list(points.gauss1(2)[1])

    def _gauss_x1(self):
        from . import points
        return list(points.gauss1(2)[0])

    def algorithmic_bytes_per_element(self):
        '''connectivity + unique vertex coordinates + CSR values written once.'''
        n = self.n
        nverts_per_elem = (n + 1) ** 3 / n ** 3
        nnz_per_elem = (3 * n + 1) ** 3 / n ** 3
        conn = 0 if self.fast else 8 * 4  # structured connectivity is generated in-kernel on the fast path; int32 otherwise
        coords = nverts_per_elem * 24 if self.variant == 'iso' else 0.
        return conn + coords + nnz_per_elem * 8

    def owned_csr(self):
        '''(values, rowptr, colidx) of the rows this rank owns, global numbering, on the host.'''
        self.finish()
        return partition.owned_rows(self.slab, device.to_host(self.values), device.to_host(self.rowptr), device.to_host(self.colidx))


class ElasticityP2:
    '''configs[2]: 3-D linear elasticity (examples/elasticity.py scaled to 3-D: lambda = 1, mu = .5/nu - 1, nu = .3),
    quadratic C0 vector basis (27 nodes x 3 components per element), 3x3x3 Gauss, isoparametric P1 geometry with the vertices
    perturbed by default_rng(seed).uniform(-.2, .2).  Stiffness-matrix (re)assembly through nh_p2hex_matrix; the pattern is the
    closed-form one of nh_p2hex_pattern (bit-equal to the generic row-wise build: tests/test_gpu_p2hex.py).  With world > 1 rank r
    owns `layers` element layers of a (layers * world) x n x n mesh plus one ghost layer below (pattern of the interface plane only);
    the rows of its top node plane go to rank r + 1 (partition.HaloPlan).'''

    ncomp = 3

    def __init__(self, n=64, rank=0, world=1, variant='iso', seed=0, lam=1., mu=.5 / .3 - 1, layers=None, halo='recompute'):
        self.n, self.rank, self.world, self.variant, self.seed = int(n), int(rank), int(world), variant, seed
        self.layers = int(layers) if layers else self.n
        self.C = self.form_tensor(lam, mu)
        self.kernel_name = 'k_p2hex_inreg<3,1,1>' if variant == 'iso' else 'k_p2hex_rows_uniform'  # ('uniform': unit cells, every element matrix the same)
        self.slab = partition.Slab(self.layers, self.rank, self.world, shape_jk=(self.n, self.n), degree=2, ncomp=3, halo=halo)

    @staticmethod
    def form_tensor(lam, mu):
        C = numpy.zeros((3, 4, 3, 4))
        for c in range(3):
            for i in range(3):
                for d in range(3):
                    for j in range(3):
                        C[c, 1 + i, d, 1 + j] = lam * (c == i) * (d == j) + mu * ((c == d) * (i == j) + (c == j) * (d == i))
        return C

    def setup(self):
        n, s = self.n, self.slab
        self.shape = (s.local_layers, n, n)
        self.domain, _ = mesh.rectilinear(list(self.shape))
        self.basis = self.domain.basis('std', degree=2)
        gb = self.domain.basis('std', degree=1)
        # global vertex field over the (layers * world) x n x n mesh: every rank draws the same stream and keeps its planes
        nI = self.layers * self.world + 1
        I0 = s.first_global_plane // 2
        idx = numpy.stack(numpy.meshgrid(numpy.arange(I0, I0 + s.local_layers + 1, dtype=float), numpy.arange(n + 1.), numpy.arange(n + 1.), indexing='ij'), -1)
        if self.variant == 'iso':
            pert = numpy.random.default_rng(self.seed).uniform(-.2, .2, (nI, n + 1, n + 1, 3))
            idx = idx + pert[I0:I0 + s.local_layers + 1]
        self.verts = idx.reshape(-1, 3)
        self.geom = gb @ self.verts
        self.smp = self.domain.sample('gauss', 4)
        self.tables = self.smp.tables(self.basis)
        self.kgeom = self.smp.geometry(self.geom)
        self.nelems = s.own_layers * n * n

    def build_pattern(self):
        s = self.slab
        self.rowptr, self.colidx = kernels.p2hex_pattern(self.shape, 3)
        self.nnz = int(self.colidx.numel())
        self.values = device.zeros(self.nnz, 'float64')  # write-once kernel: no zero-fill per step (rows of a ghost plane are never written)
        owners = s.written_planes[0] // 2, (s.written_planes[1] - 1) // 2
        if self.variant == 'uniform':
            if self.world > 1 and s.halo == 'reduce':
                # (the uniform-cell kernel replicates COMPLETE rows into every written plane: the received partial rows of the interface would be counted twice)
                raise ValueError("variant='uniform' writes complete rows: use halo='recompute' with more than one rank")
            self._launch = kernels.P2HexUniform(shape=self.shape, nq=self.smp.points.npoints, weights=self.smp._weights_dev, T=self.tables.T, ncomp=3, C=self.C,
                                                cell=(1., 1., 1.), owners=owners)
        else:
            self._launch = kernels.P2HexMatrix(shape=self.shape, nq=self.smp.points.npoints, weights=self.smp._weights_dev, geom=self.kgeom, T=self.tables.T, ncomp=3,
                                               C=self.C, layers=s.value_layers, owners=owners)
        self.halo = partition.HaloPlan(s, self.rowptr) if self.world > 1 and s.halo == 'reduce' else None

    def step(self, kernel_events=None, exchange=True):
        if kernel_events:
            kernel_events[0].record()
        self._launch(self.values)
        if kernel_events:
            kernel_events[1].record()
        if self.halo is not None and exchange:
            self.halo.exchange(self.values)

    def finish(self):
        pass

    check = PoissonSlab.check  # rigid translations lie in the kernel of K: the owned rows sum to zero

    def algorithmic_bytes_per_element(self):
        '''unique vertex coordinates + CSR values written once (structured connectivity is generated in-kernel).'''
        n = self.n
        return (0 if self.variant == 'uniform' else (n + 1) ** 3 / n ** 3 * 24) + (8 * n + 1) ** 3 * 9 / n ** 3 * 8

    def algorithmic_flops_per_element(self):
        '''Gram-matrix formulation: G = D^T W D over (27 nodes x 3 gradient slots)^2 x 27 points, multiply-add = 2 flops; the form
        tensor costs 21 flops per node pair on top.  (SURVEY 8d's 2.3 Mflop counts the B^T C B product per point, which this
        formulation does not execute.)'''
        return 2. * 81 * 81 * 27 + 21. * 27 * 27

    @staticmethod
    def mfma_per_element():
        '''v_mfma_f64_16x16x4_f64 per element: 8 units of 4 row nodes x 2 column tiles x 3 trial slots x 7 k-steps'''
        return 8 * 2 * 3 * 7

    def owned_csr(self):
        '''(values, rowptr, colidx) of the rows this rank owns, global numbering, on the host.'''
        return partition.owned_rows(self.slab, device.to_host(self.values), device.to_host(self.rowptr), device.to_host(self.colidx))
