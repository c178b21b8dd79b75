'''Multi-GPU element partition for structured slabs.

The reference's only parallelism is fork + shared mmap arrays over the element
loop (/root/reference/src/nutils/parallel.py:27-154; locks around in-place adds,
evaluable.py:7116-7133).  Here: one process per GPU; the global mesh is cut into
contiguous slabs of element layers along the slowest axis, so that -- with the
reference's element order (last axis fastest) and dof order (first axis slowest)
-- the rows a rank owns are one contiguous block of the global CSR and the global
matrix is the concatenation of the per-rank blocks in rank order.

rank r assembles its own element layers only.  Its local mesh additionally holds
ONE ghost element layer below (r > 0) that contributes to the local sparsity
PATTERN but not to the values: this makes the pattern of the interface dof plane
complete on its owner (rank r owns the plane it shares with rank r-1) without any
communication.  The only exchange is the reduce of the partial VALUES of the
interface plane: rank r sends the rows of its top plane (a contiguous tail of its
values array: cols in planes I-1 and I) to rank r+1, where they are the leading 2/3
of each owned interface row (cols sorted plane-first).  Point-to-point over
RCCL/xGMI (torch.distributed backend nccl; gloo on CPU tensors in the tests).

halo='recompute' (the default of the structured workloads since round 4): the ghost
layer contributes VALUES as well -- rank r > 0 assembles one element layer more than
it owns (6 % at 16 layers, 12.5 % at 8) and writes only the rows of its owned planes,
which are then complete: no exchange, no collective, no CUs set aside, and the step is
a single kernel that a HIP graph (or a K-steps loop) can carry.  For slabs of a
structured mesh the interface plane is 2.4 MB (P1, 128 x 128) or ~90 MB (P2 vector,
64 x 64) per step and neighbour, against ~20 us / 0.7 ms of kernel: recomputing is
cheaper than any transfer at these sizes (DESIGN.md section 6 has the model).
halo='reduce' keeps the exchange described above (unstructured partitions, where a
ghost layer is not one layer thick, need it).
'''

import numpy


class Slab:
    '''Index bookkeeping of one rank's slab of n element layers for a C0 ('std') basis of the given degree with ncomp components:
    `degree` dof planes per element layer, the plane on a slab boundary is shared; a row of that plane couples 2 degree + 1 planes,
    degree + 1 of them through the elements below.'''

    def __init__(self, n, rank, world, shape_jk, degree=1, ncomp=1, halo='reduce'):
        if halo not in ('reduce', 'recompute'):
            raise ValueError("halo must be 'reduce' or 'recompute'")
        self.halo = halo
        self.n, self.rank, self.world = int(n), int(rank), int(world)
        self.nj, self.nk = (int(x) for x in shape_jk)
        self.degree, self.ncomp = int(degree), int(ncomp)
        p = self.degree
        if not 0 <= rank < world:
            raise ValueError('rank out of range')
        self.ghost_layers = 1 if rank > 0 else 0
        self.own_layers = self.n
        self.local_layers = self.n + self.ghost_layers
        self.first_global_plane = p * (rank * self.n - self.ghost_layers)  # global index of local dof plane 0
        self.plane = (p * self.nj + 1) * (p * self.nk + 1) * self.ncomp    # rows (= flat dofs) per plane
        self.own_plane_begin = p * self.ghost_layers
        self.own_plane_end = p * (self.ghost_layers + self.n) + (1 if rank == world - 1 else 0)
        self.sends = rank < world - 1   # top plane -> rank+1
        self.recvs = rank > 0           # into the first owned plane <- rank-1
        self.send_plane = p * (self.ghost_layers + self.n)
        self.recv_plane = p * self.ghost_layers
        self.lower_planes, self.coupled_planes = p + 1, 2 * p + 1
        if halo == 'recompute':
            self.sends = self.recvs = False

    @property
    def value_layers(self):
        '''local element layers [a, b) that contribute values: the own layers, with halo='recompute' also the ghost layer'''
        return (0 if self.halo == 'recompute' else self.ghost_layers, self.local_layers)

    @property
    def written_planes(self):
        '''local dof planes [a, b) whose rows the assembly writes: the owned planes, with halo='reduce' also the top plane that is sent on'''
        return (self.own_plane_begin, self.own_plane_end if self.halo == 'recompute' else self.degree * self.local_layers + 1)


class HaloPlan:
    '''Pre-computed send range and receive scatter indices for the interface-plane reduce.
    Works on CUDA tensors (RCCL) and CPU tensors (gloo).'''

    def __init__(self, slab, rowptr):
        import torch
        self.slab = slab
        self._ops = {}
        p = slab.plane
        if slab.sends:
            r0 = slab.send_plane * p
            self.send_a, self.send_b = int(rowptr[r0]), int(rowptr[r0 + p])
        if slab.recvs:
            r0 = slab.recv_plane * p
            rp = rowptr[r0:r0 + p + 1]
            lens = rp[1:] - rp[:-1]
            if int((lens % slab.coupled_planes).abs().sum()) != 0:
                raise ValueError(f'interface rows are expected to couple {slab.coupled_planes} dof planes')
            slen = lens * slab.lower_planes // slab.coupled_planes
            starts = torch.cumsum(slen, 0) - slen
            total = int(slen.sum())
            within = torch.arange(total, device=rowptr.device, dtype=rowptr.dtype) - torch.repeat_interleave(starts, slen)
            self.recv_idx = torch.repeat_interleave(rp[:-1], slen) + within
            self.recv_buf = torch.empty(total, dtype=torch.float64, device=rowptr.device)

    def exchange(self, values):
        import torch.distributed as dist
        s = self.slab
        staged = values.is_cuda and dist.get_backend() == 'gloo'  # gloo moves host memory only (tests on a 1-GPU box)
        send = values[self.send_a:self.send_b] if s.sends else None
        recv = self.recv_buf if s.recvs else None
        if staged:
            send = send.cpu() if s.sends else None
            recv = recv.cpu() if s.recvs else None
        key = (values.data_ptr(), staged)
        ops = None if staged else self._ops.get(key)  # the descriptors of a value array are built once (two arrays when pipelined)
        if ops is None:
            ops = []
            if s.sends:
                ops.append(dist.P2POp(dist.isend, send, s.rank + 1))
            if s.recvs:
                ops.append(dist.P2POp(dist.irecv, recv, s.rank - 1))
            if not staged:
                self._ops[key] = ops
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if s.recvs:
            if staged:
                self.recv_buf.copy_(recv)
            if values.is_cuda:  # owner-side reduce of the interface rows: nh_monomial with an output index (unique positions), on the current stream
                from . import kernels
                kernels.monomial(self.recv_buf, [], [], values, out_index=self.recv_idx)
            else:  # (CPU tensors: the gloo runs of tests/test_partition.py)
                values.index_add_(0, self.recv_idx, self.recv_buf)


def owned_rows(slab, values, rowptr, colidx):
    '''Host arrays of the CSR rows owned by this rank in GLOBAL numbering
    (rows re-based to the first owned row; cols shifted by the slab's first plane).'''
    a, b = slab.own_plane_begin * slab.plane, slab.own_plane_end * slab.plane
    lo, hi = rowptr[a], rowptr[b]
    return values[lo:hi], rowptr[a:b + 1] - lo, colidx[lo:hi] + slab.first_global_plane * slab.plane


def concatenate(blocks):
    '''Global CSR from the per-rank owned blocks in rank order.'''
    values = numpy.concatenate([b[0] for b in blocks])
    colidx = numpy.concatenate([b[2] for b in blocks])
    rowptr = [numpy.zeros(1, dtype=numpy.int64)]
    off = 0
    for b in blocks:
        rowptr.append(b[1][1:] + off)
        off += b[1][-1]
    return values, numpy.concatenate(rowptr), colidx


# =====================================================================================================================================
# any mesh: element ranges after a dof-locality sort, shared rows from host index lists (last sentences)
# =====================================================================================================================================

class ElementPartition:
    '''Partition of ANY mesh given by its element -> dof connectivity (ragged allowed: hierarchical bases) over `world` ranks -- what replaces the reference's fork over the
    element loop (parallel.py:128-154: a shared element counter, locks around the in-place adds of evaluable.py:7116-7133) when the mesh has no layers to cut.

      * elements are sorted by their smallest dof (stable: a dof-locality sort -- meshes are numbered with locality, and elements that share rows end up next to each other) and
        cut into `world` contiguous ranges of equal weight (default: nb^2 per element, the size of the local matrix);
      * a row (dof) is OWNED by the lowest rank whose elements touch it; rank r's local mesh is its own elements E_r plus the GHOST elements G_r (elements of other ranks that
        touch a row r owns): with them the sparsity pattern of every owned row is complete on its owner, without communication (the connectivity is replicated);
      * halo='reduce': ghosts contribute to the pattern only (their values are structural zeros); every rank computes the partial rows its own elements add to rows of other
        owners and SENDS them to the owner, which adds them into its rows through index lists computed once on the host (SharedRowPlan) -- the RCCL reduce over the shared-dof rows
        only that BASELINE.json's north_star names; halo='recompute': ghosts contribute values as well, owned rows are complete, nothing travels;
      * the global CSR is the owners' rows in row order (merge): index arrays bit-equal to the single-process assembly, structural zeros included, because an owner's pattern is
        the sorted union over ALL elements that touch the row, exactly what the dedup of evaluable.py:588-616 sees.

    All numbering stays GLOBAL on every rank (local CSR arrays have nrows + 1 row pointers, rows nobody local touches are empty): no renumbering, the kernels see an element
    list of the one mesh.  Rows are flat dofs: scalar dof * ncomp + component.'''

    def __init__(self, offsets, dofs, ndofs, world, ncomp=1, weights=None, halo='reduce'):
        if halo not in ('reduce', 'recompute'):
            raise ValueError("halo must be 'reduce' or 'recompute'")
        self.halo = halo
        self.offsets = numpy.asarray(offsets, dtype=numpy.int64)
        self.dofs = numpy.asarray(dofs, dtype=numpy.int64)
        self.ndofs, self.world, self.ncomp = int(ndofs), int(world), int(ncomp)
        ne = len(self.offsets) - 1
        nb = numpy.diff(self.offsets)
        if ne < self.world:
            raise ValueError('fewer elements than ranks')
        if (nb <= 0).any() or self.dofs.min(initial=0) < 0 or self.dofs.max(initial=-1) >= self.ndofs:
            raise ValueError('invalid connectivity')
        elem_of = numpy.repeat(numpy.arange(ne), nb)
        mindof = numpy.full(ne, self.ndofs, dtype=numpy.int64)
        numpy.minimum.at(mindof, elem_of, self.dofs)
        self.order = numpy.argsort(mindof, kind='stable')          # position -> element
        w = numpy.asarray(weights, dtype=float) if weights is not None else nb.astype(float) ** 2
        cw = numpy.cumsum(w[self.order])
        cuts = numpy.searchsorted(cw, cw[-1] * numpy.arange(1, self.world) / self.world, side='left') + 1
        cuts = numpy.maximum.accumulate(numpy.clip(cuts, numpy.arange(1, self.world), ne - self.world + numpy.arange(1, self.world)))  # (no empty range)
        self.bounds = numpy.concatenate([[0], cuts, [ne]]).astype(numpy.int64)
        self.rank_of_element = numpy.empty(ne, dtype=numpy.int64)
        self.rank_of_element[self.order] = numpy.repeat(numpy.arange(self.world), numpy.diff(self.bounds))
        # owner of a dof: the lowest rank that touches it (dofs no element touches: nobody, -1)
        owner = numpy.full(self.ndofs, self.world, dtype=numpy.int64)
        numpy.minimum.at(owner, self.dofs, self.rank_of_element[elem_of])
        owner[owner == self.world] = -1
        self.owner = owner
        self._elem_of = elem_of

    def own_elements(self, rank):
        '''elements of rank `rank`, ascending (the element order inside a rank is the mesh's: the sums of a row keep the order of the single-process loop)'''
        return numpy.sort(self.order[self.bounds[rank]:self.bounds[rank + 1]])

    def ghost_elements(self, rank):
        '''elements of OTHER ranks that touch a row `rank` owns, ascending'''
        touches = (self.owner[self.dofs] == rank) & (self.rank_of_element[self._elem_of] != rank)
        return numpy.unique(self._elem_of[touches])

    def local_elements(self, rank):
        '''-> (elements of the local mesh, ascending; mask: True where the element contributes VALUES)'''
        own, ghost = self.own_elements(rank), self.ghost_elements(rank)
        el = numpy.concatenate([own, ghost])
        live = numpy.concatenate([numpy.ones(len(own), dtype=bool), numpy.full(len(ghost), self.halo == 'recompute')])
        o = numpy.argsort(el, kind='stable')
        return el[o], live[o]

    def owned_rows(self, rank):
        '''flat rows owned by `rank`, ascending'''
        d = numpy.flatnonzero(self.owner == rank)
        return (d[:, None] * self.ncomp + numpy.arange(self.ncomp)).ravel()

    def foreign_rows(self, rank):
        '''{owner: flat rows of that owner which the OWN elements of `rank` add to, ascending} -- what `rank` sends with halo='reduce' '''
        mine = self.rank_of_element[self._elem_of] == rank
        d = numpy.unique(self.dofs[mine])
        d = d[self.owner[d] != rank]
        out = {}
        for o in numpy.unique(self.owner[d]):
            dd = d[self.owner[d] == o]
            out[int(o)] = (dd[:, None] * self.ncomp + numpy.arange(self.ncomp)).ravel()
        return out


class SharedRowPlan:
    '''halo='reduce' on an ElementPartition: which entries of this rank's value array travel to which owner, and where the entries it receives are added.

    Setup (once per pattern): every rank publishes, per destination, the rows it sends and their column lists (`offer()`); the owner finds the position of every received entry
    in its own -- complete -- rows (`accept(offers)`: one searchsorted per sender).  A step: `exchange(values)` packs the rows (gather by index), sends / receives the packed
    value buffers point to point (torch.distributed: RCCL on device tensors, gloo on host tensors) and adds what arrives at the prepared positions (nh_monomial with an
    output index on the device).  Payload = the values of the shared rows only.'''

    def __init__(self, part, rank, rowptr, colidx):
        self.part, self.rank = part, int(rank)
        self.rowptr = numpy.asarray(rowptr, dtype=numpy.int64)
        self.colidx = numpy.asarray(colidx, dtype=numpy.int64)
        self.send = {}  # dst -> (rows, positions in this rank's value array)
        if part.halo == 'reduce':
            for dst, rows in part.foreign_rows(rank).items():
                a, b = self.rowptr[rows], self.rowptr[rows + 1]
                lens = b - a
                pos = numpy.repeat(a, lens) + (numpy.arange(int(lens.sum())) - numpy.repeat(numpy.cumsum(lens) - lens, lens))
                self.send[dst] = rows, pos
        self.recv = {}  # src -> positions in this rank's value array, in the sender's packing order
        self._dev = {}

    def offer(self):
        '''what this rank tells the owners at setup: {dst: (rows, row lengths, column indices of the packed entries)}'''
        return {dst: (rows, self.rowptr[rows + 1] - self.rowptr[rows], self.colidx[pos]) for dst, (rows, pos) in self.send.items()}

    def accept(self, offers):
        '''offers[src] = that rank's offer(); keeps the positions of the entries each source sends to THIS rank'''
        for src, offer in enumerate(offers):
            if src == self.rank or self.rank not in offer:
                continue
            rows, lens, cols = offer[self.rank]
            if (self.part.owner[rows // self.part.ncomp] != self.rank).any():
                raise ValueError(f'rank {src} offers rows that rank {self.rank} does not own')
            pos = numpy.empty(len(cols), dtype=numpy.int64)
            at = 0
            for r, n in zip(rows, lens):
                a, b = self.rowptr[r], self.rowptr[r + 1]
                p = numpy.searchsorted(self.colidx[a:b], cols[at:at + n])
                if (p >= b - a).any() or (self.colidx[a:b][numpy.minimum(p, b - a - 1)] != cols[at:at + n]).any():
                    raise ValueError(f'row {r}: an entry sent by rank {src} is missing from the owner\'s pattern')
                pos[at:at + n] = a + p
                at += n
            self.recv[src] = pos

    def setup(self):
        '''offer / accept over torch.distributed (host objects, once per pattern)'''
        import torch.distributed as dist
        offers = [None] * self.part.world
        dist.all_gather_object(offers, self.offer())
        self.accept(offers)

    def _index(self, key, idx, like):
        import torch
        k = key, like.device
        if k not in self._dev:
            self._dev[k] = torch.from_numpy(numpy.ascontiguousarray(idx)).to(like.device)
        return self._dev[k]

    def pack(self, values, dst):
        '''the packed value buffer for owner `dst` (device tensors: nh_index_copy; host tensors: index_select)'''
        import torch
        idx = self._index(('s', dst), self.send[dst][1], values)
        if values.is_cuda:
            from . import kernels
            buf = torch.empty(idx.numel(), dtype=values.dtype, device=values.device)
            kernels.index_copy(values, buf, src_index=idx)
            return buf
        return values.index_select(0, idx)

    def add(self, values, src, buf):
        '''owner-side reduce of what `src` sent (unique positions per source: plain adds; the sources are added in rank order)'''
        idx = self._index(('r', src), self.recv[src], values)
        if values.is_cuda:
            from . import kernels
            kernels.monomial(buf, [], [], values, out_index=idx)
        else:
            values.index_add_(0, idx, buf)

    def exchange(self, values):
        import torch
        import torch.distributed as dist
        if not self.send and not self.recv:
            return
        staged = values.is_cuda and dist.get_backend() == 'gloo'  # (gloo moves host memory only)
        sends = {dst: self.pack(values, dst) for dst in sorted(self.send)}
        recvs = {src: torch.empty(len(self.recv[src]), dtype=values.dtype, device='cpu' if staged else values.device) for src in sorted(self.recv)}
        if staged:
            sends = {d: b.cpu() for d, b in sends.items()}
        ops = [dist.P2POp(dist.isend, b, d) for d, b in sends.items()] + [dist.P2POp(dist.irecv, b, s) for s, b in recvs.items()]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for src in sorted(recvs):  # (fixed order: reproducible sums)
            self.add(values, src, recvs[src].to(values.device) if staged else recvs[src])

    def owned_block(self, values):
        '''(rows, row lengths, column indices, values) of the rows this rank owns -- host arrays, global numbering'''
        rows = self.part.owned_rows(self.rank)
        a, b = self.rowptr[rows], self.rowptr[rows + 1]
        lens = b - a
        pos = numpy.repeat(a, lens) + (numpy.arange(int(lens.sum())) - numpy.repeat(numpy.cumsum(lens) - lens, lens))
        v = values.detach().cpu().numpy() if hasattr(values, 'detach') else numpy.asarray(values)
        return rows, lens, self.colidx[pos], v[pos]


def merge_rows(blocks, nrows):
    '''Global CSR (values, rowptr, colidx) from the owners' row blocks (SharedRowPlan.owned_block of every rank): every row comes from exactly one owner.'''
    lens = numpy.zeros(nrows, dtype=numpy.int64)
    for rows, l, _, _ in blocks:
        if lens[rows].any():
            raise ValueError('a row is owned twice')
        lens[rows] = l
    rowptr = numpy.concatenate([[0], numpy.cumsum(lens)]).astype(numpy.int64)
    colidx = numpy.empty(rowptr[-1], dtype=numpy.int64)
    values = numpy.empty(rowptr[-1])
    for rows, l, c, v in blocks:
        pos = numpy.repeat(rowptr[rows], l) + (numpy.arange(int(l.sum())) - numpy.repeat(numpy.cumsum(l) - l, l))
        colidx[pos], values[pos] = c, v
    return values, rowptr, colidx
