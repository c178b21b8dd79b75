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
