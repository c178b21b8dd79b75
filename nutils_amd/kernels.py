'''Python faces of the C-ABI entry points (one function per kernel family).
Arguments are device tensors (see device.py); nothing here computes.'''

import ctypes
import os
import numpy

from . import _lib, device


def tabulate(coeffs_dev, nfn, ncoeffs, points_dev, nq, ndims):
    '''K2 (nh_poly_tabulate): T[nfn][nq][1+ndims].'''
    T = device.empty(nfn * nq * (1 + ndims), 'float64')
    _lib.call('nh_poly_tabulate', device.ptr(coeffs_dev), nfn, ncoeffs, device.ptr(points_dev), nq, ndims, device.ptr(T), device.stream())
    return T


def structured_dofs(shape, nloc, ndofs_axis, start_concat_dev, elem_begin, nelems):
    '''K1 (nh_structured_dofs).'''
    nd = len(shape)
    arr = ctypes.c_int * nd
    nb = int(numpy.prod(nloc))
    out = device.empty(nelems * nb, 'int32')
    _lib.call('nh_structured_dofs', nd, arr(*shape), arr(*nloc), arr(*ndofs_axis), device.ptr(start_concat_dev), elem_begin, nelems,
              device.ptr(out), device.stream())
    return out


GATHER_SCRATCH_LIMIT = 8 << 30
VECTOR_THREAD_PASS = {(3, 8, 3), (2, 4, 2), (2, 9, 2)}  # (ndims, functions per element, components) of nh_local_vector (nh_gather.hip)


class Pattern:
    '''K6 (nh_pattern_*): scalar sparsity pattern + element map, device resident.'''

    def __init__(self, nelems, nrows, ncols, tdofs, rdofs, nbt=0, nbr=0, toff=None, roff=None):
        args = _lib.PatternArgs(nelems, nrows, ncols, nbt, nbr, device.ptr(tdofs), device.ptr(rdofs), device.ptr(toff), device.ptr(roff))
        handle = ctypes.c_void_p()
        _lib.call('nh_pattern_build', ctypes.byref(args), ctypes.byref(handle), device.stream())
        self._handle = handle
        self._keep = (tdofs, rdofs, toff, roff)
        nnz = ctypes.c_int64()
        emap_len = ctypes.c_int64()
        srowptr, scol, emap, eoff = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.call('nh_pattern_info', handle, ctypes.byref(nnz), ctypes.byref(srowptr), ctypes.byref(scol), ctypes.byref(emap),
                  ctypes.byref(emap_len), ctypes.byref(eoff))
        self.nrows, self.ncols, self.nelems = nrows, ncols, nelems
        self.nnz_scalar = nnz.value
        self.srowptr_ptr, self.scol_ptr, self.emap_ptr, self.eoff_ptr = srowptr, scol, emap, eoff
        self.emap_len = emap_len.value

    def fused_info(self):
        '''(row blocks, rows per block, element visits) of the owner blocks built for NH_MATRIX_FUSED so far; (0, 0, 0): none'''
        nb, rpb, nv, rt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64(), ctypes.c_int()
        _lib.call('nh_pattern_fused_info', self._handle, ctypes.byref(nb), ctypes.byref(rpb), ctypes.byref(nv), ctypes.byref(rt))
        self.fused_routine = rt.value  # -1 none yet, 0 tabulated any-element routine, 1 / 2 sum-factorised trilinear routine (2: with a mass term)
        return nb.value, rpb.value, nv.value

    def owner_info(self):
        '''(row blocks, node rows per block, element visits, chunks of 64 contributions) of the row tasks built for vector-valued NH_MATRIX_FUSED so far; zeros: none'''
        nb, rpb, nv, nc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64()
        _lib.call('nh_pattern_owner_info', self._handle, ctypes.byref(nb), ctypes.byref(rpb), ctypes.byref(nv), ctypes.byref(nc))
        return nb.value, rpb.value, nv.value, nc.value

    def expanded_nnz(self, nct, ncr, mask=None):
        nnz = ctypes.c_int64()
        m = None if mask is None else numpy.ascontiguousarray(mask, dtype=numpy.uint8)
        _lib.call('nh_pattern_expanded_nnz', self._handle, nct, ncr, device.host_ptr(m), ctypes.byref(nnz))
        return nnz.value

    def expand(self, nct=1, ncr=1, mask=None):
        '''(rowptr, colidx) int64 device tensors of the component-expanded CSR (cached: re-assemblies reuse them).'''
        m = None if mask is None else numpy.ascontiguousarray(mask, dtype=numpy.uint8)
        key = nct, ncr, None if m is None else m.tobytes()
        cache = self.__dict__.setdefault('_expanded', {})
        if key in cache:
            return cache[key]
        cache[key] = self._expand(nct, ncr, m)
        return cache[key]

    def _expand(self, nct, ncr, m):
        nnz = self.expanded_nnz(nct, ncr, m)
        rowptr = device.empty(self.nrows * nct + 1, 'int64')
        colidx = device.empty(nnz, 'int64')
        _lib.call('nh_pattern_expand', self._handle, nct, ncr, device.host_ptr(m), device.ptr(rowptr), device.ptr(colidx), device.stream())
        return rowptr, colidx

    def __del__(self):
        h = getattr(self, '_handle', None)
        if h:
            try:
                _lib.load().nh_pattern_free(h)
            except Exception:
                pass
            self._handle = None


UNION_MAX = 8  # NH_UNION_MAX of nh_pattern.hip


def pattern_union(parts):
    '''Union of sorted-unique CSR patterns [(rowptr, colidx)] with equal row counts (nh_pattern_union_*): (rowptr, colidx, [position of every entry of part i in
    the union]).  Row-wise merge of sorted lists on the device; replaces the sort-based unique of the reference (evaluable.py:5560-5682).'''
    n = len(parts)
    nrows = parts[0][0].numel() - 1
    if any(rp.numel() - 1 != nrows for rp, _ in parts):
        raise ValueError('pattern_union: parts with different numbers of rows')
    if n > UNION_MAX:
        # more parts than one merge takes (NH_UNION_MAX): fold -- the union of the first UNION_MAX parts, then that union with the next UNION_MAX - 1, ...; the positions of a
        # part in an intermediate union are carried through the later ones
        rowptr, colidx, pos = pattern_union(parts[:UNION_MAX])
        at = UNION_MAX
        while at < n:
            more = parts[at:at + UNION_MAX - 1]
            rowptr2, colidx2, pos2 = pattern_union([(rowptr, colidx)] + more)
            pos = [pos2[0][q] for q in pos] + pos2[1:]
            rowptr, colidx = rowptr2, colidx2
            at += len(more)
        return rowptr, colidx, pos
    RP = (ctypes.c_void_p * n)(*[rp.data_ptr() for rp, _ in parts])
    CI = (ctypes.c_void_p * n)(*[ci.data_ptr() for _, ci in parts])
    rowptr = device.empty(nrows + 1, 'int64')
    nnz = ctypes.c_int64()
    _lib.call('nh_pattern_union_count', n, nrows, RP, CI, device.ptr(rowptr), ctypes.byref(nnz), device.stream())
    colidx = device.empty(nnz.value, 'int64')
    pos = [device.empty(ci.numel(), 'int64') for _, ci in parts]
    PP = (ctypes.c_void_p * n)(*[p.data_ptr() for p in pos])
    _lib.call('nh_pattern_union_fill', n, nrows, RP, CI, device.ptr(rowptr), device.ptr(colidx), PP, device.stream())
    return rowptr, colidx, pos


def geometry_iso(ngb, gT, gdofs, verts, bnd_axis=-1):
    g = _lib.Geometry(_lib.GEOM_ISO, ngb, device.ptr(gT), device.ptr(gdofs), device.ptr(verts), None, None, None, None, bnd_axis)
    g._keep = (gT, gdofs, verts)
    return g


def geometry_box(origin, size, bnd_axis=-1):
    g = _lib.Geometry(_lib.GEOM_BOX, 0, None, None, None, device.ptr(origin), device.ptr(size), None, None, bnd_axis)
    g._keep = (origin, size)
    return g


def geometry_tab(jac, x=None, bnd_axis=-1):
    g = _lib.Geometry(_lib.GEOM_TAB, 0, None, None, None, None, None, device.ptr(jac), device.ptr(x), bnd_axis)
    g._keep = (jac, x)
    return g


def factor_tensor(*, nelems, ndims, nq, rank, weights, geom, basis, ndofs, coeff=1., scale=None, elist=None):
    '''Taylor coefficient tensor of rank 3 or 4 of  coeff int scale u^rank dV  built on the device (nh_factor_tensor / nh_factor_fetch): (values[nnz],
    indices[rank][nnz]) as the reference's Monomial holds them (evaluable.py:5693-5751, 5785-5874).'''
    args = _lib.FactorArgs(nelems, device.ptr(elist), ndims, nq, rank, device.ptr(weights), geom, basis, device.ptr(scale), float(coeff), int(ndofs))
    nnz = ctypes.c_int64(0)
    _lib.call('nh_factor_tensor', ctypes.byref(args), ctypes.byref(nnz), device.stream())
    values = device.empty(nnz.value, 'float64')
    indices = device.empty(rank * nnz.value, 'int64')
    _lib.call('nh_factor_fetch', device.ptr(values), device.ptr(indices), device.stream())
    return values, indices.reshape(rank, nnz.value)


def rationalize(T, nelems, nb, dofs, weights, nq, ndims, W=None, dW=None, off=None):
    '''N_i = w_i B_i / W in place on per-element tables (nh_rationalize).'''
    _lib.call('nh_rationalize', device.ptr(T), nelems, nb, device.ptr(off), device.ptr(dofs), device.ptr(weights), device.ptr(W), device.ptr(dW), nq, ndims,
              device.stream())


def basis(T, dofs, nb=0, off=None, tab=None):
    b = _lib.Basis(nb, device.ptr(T), device.ptr(dofs), device.ptr(off), device.ptr(tab))
    b._keep = (T, dofs, off, tab)
    return b


FUSED_SIZES = {(1, 2), (1, 3), (2, 3), (2, 4), (2, 9), (3, 4), (3, 8)}  # (dimension, functions per element) of the owner-block kernels (NH_MATRIX_FUSED)
# ... and those for which they are the DEFAULT: measured faster than the gather with the ordered sums (tools/generic_probe.py, round 5, ordered rounds: 128^3 trilinear
# 0.55 against 0.90 ms; 2048^2 bilinear 0.43 against 0.45, 1024^2 quadratic splines 0.43 against 0.59, but 1024^2 biquadratic 0.88 against 0.69 ms -- the 2-D sizes stay
# with the gather)
FUSED_DEFAULT = {(3, 8)}
# (dimension, functions per element, components) of the owner kernel for vector-valued blocks (nh_owner.hip: Gram sums per scalar entry from D tables in LDS, one pass)
OWNER_VECTOR = {(3, 8, 3), (2, 4, 2), (2, 9, 2), (3, 27, 3)}


def assemble_matrix(*, nelems, ndims, nq, weights, geom, test, trial, nct, ncr, C, mask, pattern, values, elist=None, emap_offset=0, scale=None, flags=0,
                    cq=None, first_touch=None, gather=None, store=False, fused=False, fresh=False):
    '''K3+K4+K5 (nh_assemble_matrix); accumulates into `values`.  `first_touch=(grid_shape, nodes_per_axis)`: NH_MATRIX_FIRST_TOUCH.
    gather: NH_MATRIX_GATHER (deterministic owner-side reduction instead of atomics); None = from the second assembly on a pattern on (the gather
    map costs one device sort of the element map, which a one-off assembly does not earn back).  fused: NH_MATRIX_FUSED (owner blocks: one pass
    without scratch or global atomics for scalar blocks on small uniform bases, contributions added in visit order: bit-reproducible, bit-identical to the gather; excludes gather;
    the default for those blocks -- NUTILS_AMD_NO_FUSED=1 restores the gather / atomics choice).  fresh: `values` is uninitialised -- the gather and
    owner-block paths STORE their sums (no zero fill, no read of the old values), every other path gets the array zero-filled first.'''
    C = numpy.ascontiguousarray(C, dtype=float)
    if C.shape != (nct, 1 + ndims, ncr, 1 + ndims):
        raise ValueError(f'coefficient tensor has shape {C.shape}, expected {(nct, 1 + ndims, ncr, 1 + ndims)}')
    m = None if mask is None else numpy.ascontiguousarray(mask, dtype=numpy.uint8)
    if cq is not None and cq.numel() != nelems * nq * C.size:
        raise ValueError('per-point coefficient tensor must have shape [nelems][nq] + C.shape')
    args = _lib.MatrixArgs(nelems, device.ptr(elist), ndims, nq, device.ptr(weights), geom, test, trial, nct, ncr, device.host_ptr(C),
                           device.host_ptr(m), pattern.srowptr_ptr, ctypes.c_void_p(pattern.emap_ptr.value + 4 * emap_offset), pattern.eoff_ptr,
                           device.ptr(values), device.ptr(scale), int(flags) | (32 if first_touch else 0), device.ptr(cq))
    if first_touch:
        args.grid_shape[:] = list(first_touch[0]) + [1] * (3 - len(first_touch[0]))
        args.nodes_per_axis = int(first_touch[1])
    if elist is None and emap_offset == 0 and not os.environ.get('NUTILS_AMD_NO_BUCKETS'):
        args.pattern = pattern._handle  # ragged bases: launches per size class of the pattern
    whole = emap_offset == 0 and not flags and not first_touch and nelems == pattern.nelems
    if (not fused and gather is None and not os.environ.get('NUTILS_AMD_NO_FUSED') and whole and elist is None and nct == ncr == 1 and cq is None and ((ndims, test.nb) in FUSED_DEFAULT or os.environ.get('NUTILS_AMD_FUSED') and (ndims, test.nb) in FUSED_SIZES)
            and test.nb == trial.nb and test.dofs_dev == trial.dofs_dev and not test.off_dev):
        # (default since round 4 for the blocks the owner-block kernels cover: one pass, 1.5 x instead of 4.6 x the algorithmic traffic, and -- with the
        # turns of the block plan -- bit-reproducible like the gather)
        fused = True
    if (not fused and gather is None and not os.environ.get('NUTILS_AMD_NO_FUSED') and whole and elist is None and nct == ncr and cq is None and (ndims, test.nb, nct) in OWNER_VECTOR
            and test.nb == trial.nb and test.T_dev == trial.T_dev and test.dofs_dev == trial.dofs_dev and test.tab_dev == trial.tab_dev and not test.off_dev):
        fused = True  # (vector-valued blocks on small uniform bases: one pass instead of the thread pass + gather, 2.3 x the algorithmic bytes)
    if fused:
        if not whole:
            raise ValueError('fused needs all elements of the pattern in one call')
        if gather:
            raise ValueError('fused excludes gather')
        gather = False
    if gather is None:
        # (automatic only while the scratch of the local matrices stays below GATHER_SCRATCH_LIMIT bytes: 1.07 GB for the 128^3 trilinear mesh)
        # and for blocks with a thread pass (scalar blocks; the vector-valued sizes of VECTOR_THREAD_PASS): the local matrices of other vector-valued
        # blocks go through the one-wave-per-element kernel either way, and writing + gathering them costs more than its atomics -- 96^3 trilinear
        # elasticity 15.7 against 7.9 ms, 32^3 triquadratic 14.5 against 3.9 ms, tools/generic_probe.py)
        thread_pass = nct == ncr == 1 or ((ndims, test.nb, nct) in VECTOR_THREAD_PASS and nct == ncr and cq is None and test.nb == trial.nb
                                          and test.T_dev == trial.T_dev and test.dofs_dev == trial.dofs_dev and test.tab_dev == trial.tab_dev and not test.off_dev)
        # ... and blocks on ragged bases with a constant form.  Vector-valued: the element kernel sums Gram matrices per node pair (nh_assemble_generic.hip), which leaves
        # the global atomics as its bound -- 63 488 rational hierarchical elements 2.36 ms with atomics, 1.76 ms through the gather.  Scalar: 1.20 against 1.22 ms -- the
        # same time, and the sums become bit-reproducible (tools/ragged_probe.py, RAGGED_PROBE_SCALAR=1)
        thread_pass = thread_pass or bool(test.off_dev and cq is None)
        gather = (whole and thread_pass and getattr(pattern, '_assemblies', 0) >= 1 and not os.environ.get('NUTILS_AMD_NO_GATHER')
                  and 8 * pattern.emap_len * nct * ncr <= GATHER_SCRATCH_LIMIT and pattern.emap_len < 2 ** 32 and pattern.nnz_scalar < 2 ** 32)
    if whole:
        pattern._assemblies = getattr(pattern, '_assemblies', 0) + 1
    if fresh:
        if gather or fused:
            store = True
        else:
            values.zero_()
    if gather:
        if not whole:
            raise ValueError('gather needs all elements of the pattern in one call')
        args.pattern = pattern._handle
        args.flags |= 64 | (128 if store else 0)
    elif fused:
        args.pattern = pattern._handle
        args.flags |= 256 | (128 if store else 0)
    elif store:
        raise ValueError('store is an option of the gather and fused paths')
    _lib.call('nh_assemble_matrix', ctypes.byref(args), device.stream())


class ScatterPlan:
    '''Map of the deterministic vector scatter (nh_scatter_plan_build): for every dof the positions of its contributions in an element-major
    array of local vectors, in the order of the reference's loop.  One per (basis tables, element list).'''

    def __init__(self, *, nelems, nrows, nb, dofs, off=None, elist=None, nlist=None):
        handle = ctypes.c_void_p()
        _lib.call('nh_scatter_plan_build', int(nelems), int(nrows), int(nb), device.ptr(dofs), device.ptr(off), device.ptr(elist),
                  int(nelems if elist is None else nlist), ctypes.byref(handle), device.stream())
        self._handle, self._keep = handle, (dofs, off, elist)
        self.npositions = int(dofs.numel())  # length of a local array per component

    def __del__(self):
        h = getattr(self, '_handle', None)
        if h:
            try:
                _lib.load().nh_scatter_plan_free(h)
            except Exception:
                pass
            self._handle = None


def scatter_gather(pairs, ncomp, out, accumulate=True):
    '''out[dof][c] (+)= sum over the (plan, local array) pairs, in order, of the contributions of dof (nh_scatter_gather).'''
    n = len(pairs)
    P = (ctypes.c_void_p * n)(*[p._handle.value for p, _ in pairs])
    L = (ctypes.c_void_p * n)(*[loc.data_ptr() for _, loc in pairs])
    _lib.call('nh_scatter_gather', n, P, L, int(ncomp), device.ptr(out), int(bool(accumulate)), device.stream())


def assemble_vector(*, nelems, ndims, nq, weights, geom, test, trial, nct, ncr, C=None, f=None, u=None, out=None, f0=0., out_scalar=None,
                    elist=None, scale=None, local=None):
    C = None if C is None else numpy.ascontiguousarray(C, dtype=float)
    f = None if f is None else numpy.ascontiguousarray(f, dtype=float)
    if C is not None and C.shape != (nct, 1 + ndims, ncr, 1 + ndims):
        raise ValueError(f'coefficient tensor has shape {C.shape}')
    if f is not None and f.shape != (nct, 1 + ndims):
        raise ValueError(f'source tensor has shape {f.shape}')
    args = _lib.VectorArgs(nelems, device.ptr(elist), ndims, nq, device.ptr(weights), geom, test, trial, nct, ncr, device.host_ptr(C),
                           device.host_ptr(f), device.ptr(u), device.ptr(out), float(f0), device.ptr(out_scalar), device.ptr(scale), device.ptr(local))
    _lib.call('nh_assemble_vector', ctypes.byref(args), device.stream())


def _terms_args(keep, *, nelems, ndims, nq, weights, geom, fields, blocks, terms, polys=(), elist=None):
    S = 1 + ndims
    F, P = _fields_polys(fields, polys, keep)
    B = (_lib.Block * len(blocks))()
    for i, blk in enumerate(blocks):  # (test struct, nct, out[, local array of the deterministic scatter])
        B[i] = _lib.Block(blk[0], int(blk[1]), device.ptr(blk[2]), device.ptr(blk[3]) if len(blk) > 3 else None)
    T = (_lib.Term * len(terms))()
    for i, t in enumerate(terms):
        blk, fld = int(t['block']), int(t.get('field', -1))
        nct = blocks[blk][1]
        C, f = t.get('C'), t.get('f')
        if C is not None:
            C = numpy.ascontiguousarray(C, dtype=float)
            if fld < 0 or C.shape != (nct, S, fields[fld][2], S):
                raise ValueError(f'term {i}: coefficient tensor has shape {C.shape}')
        if f is not None:
            f = numpy.ascontiguousarray(f, dtype=float)
            if f.shape != (nct, S):
                raise ValueError(f'term {i}: source tensor has shape {f.shape}')
        qB, qt, qr = _qs(t.get('qs'), S)
        keep += [C, f, qB]
        T[i] = _lib.Term(blk, fld, int(t.get('poly', -1)), device.host_ptr(C), device.host_ptr(f), device.ptr(t.get('scale')), qt, qr, device.host_ptr(qB))
    keep += [F, P, B, T]
    return _lib.TermsArgs(nelems, device.ptr(elist), ndims, nq, device.ptr(weights), geom, len(fields), F, len(blocks), B, len(terms), T, len(polys), P)


def assemble_terms(**kwargs):
    '''All linear-form terms of a residual on one sample in ONE element loop (nh_assemble_terms).  Keywords: nelems, ndims, nq, weights, geom, elist=None,
    fields: [(basis struct, u, ncomp)], blocks: [(test struct, nct, out)], polys: [(vars [(field, comp)], coeffs, powers [nterms][nvars])],
    terms: [dict(block, field=-1, poly=-1, C=None, f=None, scale=None)].'''
    keep = []
    args = _terms_args(keep, **kwargs)
    _lib.call('nh_assemble_terms', ctypes.byref(args), device.stream())


def assemble_terms_multi(lists):
    '''The term lists of several samples (`lists`: keyword dicts of assemble_terms) in one launch (nh_assemble_terms_multi).'''
    if not lists:
        return
    keep = []
    args = [_terms_args(keep, **kw) for kw in lists]
    ptrs = (ctypes.POINTER(_lib.TermsArgs) * len(args))(*[ctypes.pointer(a) for a in args])
    _lib.call('nh_assemble_terms_multi', len(args), ptrs, device.stream())


def _qs(qs, S):
    '''term option qs = (B [S][S], field_t, field_r): point factor U_t . B . U_r'''
    if qs is None:
        return None, -1, -1
    B = numpy.ascontiguousarray(qs[0], dtype=float)
    if B.shape != (S, S):
        raise ValueError(f'point factor tensor has shape {B.shape}')
    return B, int(qs[1]), int(qs[2])


def _fields_polys(fields, polys, keep):
    F = (_lib.Field * max(len(fields), 1))()
    for i, (b, u, nc) in enumerate(fields):
        F[i] = _lib.Field(b, device.ptr(u), int(nc))
    P = (_lib.PointPoly * max(len(polys), 1))()
    for i, (vars_, coeffs, powers) in enumerate(polys):
        cf = numpy.ascontiguousarray(coeffs, dtype=float)
        pw = numpy.ascontiguousarray(powers, dtype=numpy.int32).reshape(len(cf), len(vars_))
        keep += [cf, pw]
        fld = (ctypes.c_int * 4)(*([v[0] for v in vars_] + [0] * (4 - len(vars_))))
        cmp_ = (ctypes.c_int * 4)(*([v[1] for v in vars_] + [0] * (4 - len(vars_))))
        P[i] = _lib.PointPoly(len(vars_), len(cf), fld, cmp_, device.host_ptr(cf), pw.ctypes.data if pw.size else None)
    return F, P


def assemble_matrix_terms(*, nelems, ndims, nq, weights, geom, test, trial, nct, ncr, mask, pattern, values, terms, fields=(), polys=(), elist=None, flags=0,
                          gather=None):
    '''All bilinear-form terms of a matrix block in ONE element loop, several elements per workgroup (nh_assemble_matrix_terms); accumulates
    into `values`.  terms: [dict(C, kind=0, field=-1, poly=-1, L=None, scale=None)], fields / polys as in assemble_terms.'''
    S = 1 + ndims
    keep = []
    F, P = _fields_polys(fields, polys, keep)
    m = None if mask is None else numpy.ascontiguousarray(mask, dtype=numpy.uint8)
    T = (_lib.MatrixTerm * len(terms))()
    for i, t in enumerate(terms):
        C = numpy.ascontiguousarray(t['C'], dtype=float)
        if C.shape != (nct, S, ncr, S):
            raise ValueError(f'term {i}: coefficient tensor has shape {C.shape}, expected {(nct, S, ncr, S)}')
        L = t.get('L')
        if L is not None:
            L = numpy.ascontiguousarray(L, dtype=float)
            if L.shape != (nct, S):
                raise ValueError(f'term {i}: L has shape {L.shape}')
        qB, qt, qr = _qs(t.get('qs'), S)
        keep += [C, L, qB]
        T[i] = _lib.MatrixTerm(int(t.get('kind', 0)), int(t.get('field', -1)), int(t.get('poly', -1)), device.host_ptr(C), device.host_ptr(L), device.ptr(t.get('scale')),
                               qt, qr, device.host_ptr(qB))
    whole = not flags and nelems == pattern.nelems
    if gather is None:  # (as in assemble_matrix: the owner-side reduction from the second assembly on a pattern on; blocks that do not qualify ignore it)
        gather = (whole and getattr(pattern, '_assemblies', 0) >= 1 and not os.environ.get('NUTILS_AMD_NO_GATHER')
                  and 8 * pattern.emap_len <= GATHER_SCRATCH_LIMIT and pattern.emap_len < 2 ** 32 and pattern.nnz_scalar < 2 ** 32)
    if whole:
        pattern._assemblies = getattr(pattern, '_assemblies', 0) + 1
    args = _lib.MatrixTermsArgs(nelems, device.ptr(elist), ndims, nq, device.ptr(weights), geom, test, trial, nct, ncr, device.host_ptr(m), pattern.srowptr_ptr,
                                pattern.emap_ptr, pattern.eoff_ptr, device.ptr(values), int(flags) | (64 if gather and whole else 0), len(fields), F, len(terms), T,
                                len(polys), P, pattern._handle)
    _lib.call('nh_assemble_matrix_terms', ctypes.byref(args), device.stream())


def sample_eval(*, nelems, ndims, nq, geom, trial=None, ncr=1, points=None, u=None, x=None, detj=None, U=None, elist=None):
    if trial is None:
        trial = _lib.Basis(0, None, None, None, None)
    args = _lib.EvalArgs(nelems, device.ptr(elist), ndims, nq, geom, trial, ncr, device.ptr(points), device.ptr(u), device.ptr(x), device.ptr(detj), device.ptr(U))
    _lib.call('nh_sample_eval', ctypes.byref(args), device.stream())


def p1hex_pattern(shape, row_begin=0, row_end=None):
    '''Closed-form CSR index arrays of the trilinear basis on a structured hex mesh (nh_p1hex_pattern).'''
    n0, n1, n2 = (int(n) for n in shape)
    nrows = (n0 + 1) * (n1 + 1) * (n2 + 1)
    if row_end is None:
        row_end = nrows

    def cum(X, N):
        return 0 if X == 0 else (3 * N - 2 if X >= N else 3 * X - 1)

    def rp(r):
        if r >= nrows:
            return (3 * n0 + 1) * (3 * n1 + 1) * (3 * n2 + 1)
        K, J, I = r % (n2 + 1), (r // (n2 + 1)) % (n1 + 1), r // ((n2 + 1) * (n1 + 1))
        ln = lambda X, N: (X > 0) + 1 + (X < N - 1)
        return cum(I, n0 + 1) * (3 * n1 + 1) * (3 * n2 + 1) + ln(I, n0 + 1) * (cum(J, n1 + 1) * (3 * n2 + 1) + ln(J, n1 + 1) * cum(K, n2 + 1))

    nnz = rp(row_end) - rp(row_begin)
    rowptr = device.empty(row_end - row_begin + 1, 'int64')
    colidx = device.empty(nnz, 'int64')
    arr = (ctypes.c_int * 3)(n0, n1, n2)
    _lib.call('nh_p1hex_pattern', arr, row_begin, row_end, device.ptr(rowptr), device.ptr(colidx), device.stream())
    return rowptr, colidx


def _p1hex_args(shape, values, gauss_x, gauss_w, verts, origin, scale, kappa, layers, planes, unit_matrix=None, qscale=None, max_workgroups=0, mass=0.,
                qmass=None):
    n0 = int(shape[0])
    layers = (0, n0) if layers is None else layers
    planes = (0, n0 + 1) if planes is None else planes
    a = _lib.P1HexArgs()
    a.shape[:] = [int(n) for n in shape]
    a.layer_begin, a.layer_end = layers
    a.plane_begin, a.plane_end = planes
    a.verts_dev = device.ptr(verts)
    a.origin[:] = [float(x) for x in origin]
    a.scale[:] = [float(x) for x in scale]
    a.gauss_x[:] = [float(x) for x in gauss_x]
    a.gauss_w[:] = [float(x) for x in gauss_w]
    a.kappa = float(kappa)
    a.values_dev = device.ptr(values)
    a.unit_matrix_dev = device.ptr(unit_matrix)
    if qscale is not None and qscale.numel() != 8 * int(shape[0]) * int(shape[1]) * int(shape[2]):
        raise ValueError('qscale must hold 8 values per element')
    a.qscale_dev = device.ptr(qscale)
    a.max_workgroups = int(max_workgroups)
    if qmass is not None and qmass.numel() != 8 * int(shape[0]) * int(shape[1]) * int(shape[2]):
        raise ValueError('qmass must hold 8 values per element')
    a.mass = float(mass) if qmass is None or mass else 1.
    a.qmass_dev = device.ptr(qmass)
    return a


def p1hex_unit_matrix(*, shape, gauss_x, gauss_w, origin=(0., 0., 0.), scale=(1., 1., 1.), kappa=1., mass=0.):
    '''8x8 element matrix of the uniform cell (nh_p1hex_unit_matrix): computed once per mesh.'''
    ke = device.empty(64, 'float64')
    a = _p1hex_args(shape, None, gauss_x, gauss_w, None, origin, scale, kappa, None, None, mass=mass)
    _lib.call('nh_p1hex_unit_matrix', ctypes.byref(a), device.ptr(ke), device.stream())
    return ke


def p1hex_laplace(*, shape, values, gauss_x, gauss_w, verts=None, origin=(0., 0., 0.), scale=(1., 1., 1.), kappa=1., layers=None, planes=None,
                  unit_matrix=None, qscale=None, max_workgroups=0, mass=0., qmass=None):
    '''Write-once structured P1-hex assembly of kappa grad.grad + mass phi phi (nh_p1hex_laplace).'''
    a = _p1hex_args(shape, values, gauss_x, gauss_w, verts, origin, scale, kappa, layers, planes, unit_matrix, qscale, max_workgroups, mass, qmass)
    _lib.call('nh_p1hex_laplace', ctypes.byref(a), device.stream())


class P1HexLaplace:
    '''nh_p1hex_laplace with the argument block filled ONCE: a re-assembly is one ctypes call (the per-step path of a Newton loop or of
    bench.py; building the block costs more host time than the launch).  Call with the value array of the step.'''

    def __init__(self, **kwargs):
        self._keep = kwargs  # the tensors whose addresses the block holds
        values = kwargs.pop('values', None)
        self._args = _p1hex_args(kwargs['shape'], values, kwargs['gauss_x'], kwargs['gauss_w'], kwargs.get('verts'), kwargs.get('origin', (0., 0., 0.)),
                                 kwargs.get('scale', (1., 1., 1.)), kwargs.get('kappa', 1.), kwargs.get('layers'), kwargs.get('planes'), kwargs.get('unit_matrix'),
                                 kwargs.get('qscale'), kwargs.get('max_workgroups', 0), kwargs.get('mass', 0.), kwargs.get('qmass'))
        self._ref = ctypes.byref(self._args)
        self._name = 'nh_p1hex_laplace'
        self._fn = getattr(_lib.load(), self._name)

    def __call__(self, values):
        self._args.values_dev = values.data_ptr()
        if _lib.TRACE is not None:
            _lib.TRACE.append(self._name)
        _lib.check(self._fn(self._ref, device.stream()))


def p1hex_apply(*, shape, u, out, gauss_x, gauss_w, verts, kappa=1., layers=None, planes=None, qscale=None, accumulate=True, max_workgroups=0, mass=0.,
                qmass=None):
    '''out (+)= K u for the P1-hex Laplace form without forming K (nh_p1hex_apply).'''
    n = (int(shape[0]) + 1) * (int(shape[1]) + 1) * (int(shape[2]) + 1)
    if u.numel() != n or out.numel() != n:
        raise ValueError('u and out must hold one value per vertex')
    a = _p1hex_args(shape, None, gauss_x, gauss_w, verts, (0., 0., 0.), (1., 1., 1.), kappa, layers, planes, None, qscale, max_workgroups, mass, qmass)
    _lib.call('nh_p1hex_apply', ctypes.byref(a), device.ptr(u), device.ptr(out), int(bool(accumulate)), device.stream())


def p2hex_rowptr(shape, node):
    '''Scalar row pointer of a node of the structured C0 quadratic basis (closed form, nh_p2hex_rowptr).'''
    out = ctypes.c_int64()
    _lib.call('nh_p2hex_rowptr', (ctypes.c_int * 3)(*[int(n) for n in shape]), int(node), ctypes.byref(out))
    return out.value


def p2hex_pattern(shape, ncomp):
    '''Closed-form CSR index arrays of the structured C0 quadratic hex basis with ncomp fully coupled components (nh_p2hex_pattern).'''
    n0, n1, n2 = (int(n) for n in shape)
    nnodes = (2 * n0 + 1) * (2 * n1 + 1) * (2 * n2 + 1)
    nnz = p2hex_rowptr(shape, nnodes) * ncomp * ncomp
    rowptr = device.empty(nnodes * ncomp + 1, 'int64')
    colidx = device.empty(nnz, 'int64')
    _lib.call('nh_p2hex_pattern', (ctypes.c_int * 3)(n0, n1, n2), int(ncomp), device.ptr(rowptr), device.ptr(colidx), device.stream())
    return rowptr, colidx


class P2HexMatrix:
    '''Write-once assembly of a constant-coefficient form on the structured C0 quadratic hex basis (nh_p2hex_matrix), argument block
    filled once: a re-assembly is one ctypes call.  Call with the value array of the step.'''

    def __init__(self, *, shape, nq, weights, geom, T, ncomp, C, scale=None, layers=None, owners=None, max_workgroups=0):
        C = numpy.ascontiguousarray(C, dtype=float)
        if C.shape != (ncomp, 4, ncomp, 4):
            raise ValueError(f'coefficient tensor has shape {C.shape}, expected {(ncomp, 4, ncomp, 4)}')
        n0 = int(shape[0])
        a = _lib.P2HexArgs()
        a.shape[:] = [int(n) for n in shape]
        a.nq = int(nq)
        a.weights_dev = device.ptr(weights)
        a.geom = geom
        a.T_dev = device.ptr(T)
        a.ncomp = int(ncomp)
        a.C_host = device.host_ptr(C)
        a.scale_dev = device.ptr(scale)
        a.layer_begin, a.layer_end = (0, n0) if layers is None else layers
        a.owner_begin, a.owner_end = (0, n0) if owners is None else owners
        a.max_workgroups = int(max_workgroups)
        a.weights_positive = int(bool((weights > 0).all().item()))  # (once per plan: the object lives as long as the weight array it keeps)
        self._keep = (weights, geom, T, C, scale)
        self._args = a
        self._ref = ctypes.byref(a)
        self._name = 'nh_p2hex_matrix'
        self._fn = getattr(_lib.load(), self._name)

    def __call__(self, values):
        self._args.values_dev = values.data_ptr()
        if _lib.TRACE is not None:
            _lib.TRACE.append(self._name)
        _lib.check(self._fn(self._ref, device.stream()))


class P2HexUniform:
    '''The same matrix on a mesh of UNIFORM cells (`cell`: the three edge lengths; constant form, no pointwise factor): the mesh of 2 x 2 x 2 such cells is assembled once by
    nh_p2hex_matrix, a (re-)assembly replicates its rows (nh_p2hex_rows_uniform: one write stream).  `owners` as for P2HexMatrix (slabs: owner io holds the node planes
    2 io and 2 io + 1).'''

    def __init__(self, *, shape, nq, weights, T, ncomp, C, cell, owners=None):
        idx = numpy.stack(numpy.meshgrid(*[numpy.arange(2.)] * 3, indexing='ij'), -1).reshape(-1, 3)
        cell = numpy.asarray(cell, dtype=float).reshape(3)
        small = (2, 2, 2)
        rp = ctypes.c_int64()
        _lib.call('nh_p2hex_rowptr', (ctypes.c_int * 3)(*small), 125, ctypes.byref(rp))
        self.table = device.empty(rp.value * ncomp * ncomp, 'float64')
        geom = geometry_box(device.to_dev(idx * cell, 'float64'), device.to_dev(numpy.broadcast_to(cell, idx.shape), 'float64'))
        P2HexMatrix(shape=small, nq=nq, weights=weights, geom=geom, T=T, ncomp=ncomp, C=C)(self.table)
        n0 = int(shape[0])
        ob, oe = (0, n0) if owners is None else owners
        self._shape = (ctypes.c_int * 3)(*[int(n) for n in shape])
        self._planes = 2 * int(ob), min(2 * int(oe) + 2, 2 * n0 + 1)
        self._nc = int(ncomp)
        self._name = 'nh_p2hex_rows_uniform'
        self._fn = getattr(_lib.load(), self._name)

    def __call__(self, values):
        if _lib.TRACE is not None:
            _lib.TRACE.append(self._name)
        _lib.check(self._fn(self._shape, self._nc, self.table.data_ptr(), values.data_ptr(), self._planes[0], self._planes[1], device.stream()))


def p2hex_matrix(*, values, **kwargs):
    P2HexMatrix(**kwargs)(values)


def monomial_csr(rowptr, colidx, values, x, y, alpha=1.):
    '''y[r] += alpha * sum_k values[k] x[colidx[k]] (nh_monomial_csr).'''
    _lib.call('nh_monomial_csr', rowptr.numel() - 1, device.ptr(rowptr), device.ptr(colidx), device.ptr(values), device.ptr(x), float(alpha), device.ptr(y),
              device.stream())


def monomial(values, args, indices, out, out_index=None, alpha=1.):
    '''out[out_index[i]] += alpha values[i] prod_k args[k][indices[k][i]] (nh_monomial).'''
    n = len(args)
    A = (ctypes.c_void_p * max(n, 1))(*[a.data_ptr() for a in args])
    I = (ctypes.c_void_p * max(n, 1))(*[i.data_ptr() for i in indices])
    _lib.call('nh_monomial', values.numel(), device.ptr(values), n, A, I, device.ptr(out_index), float(alpha), device.ptr(out), device.stream())


def index_copy(src, dst, src_index=None, dst_index=None):
    '''dst[dst_index[i]] = src[src_index[i]] (nh_index_copy); `dst` is a device tensor or a PINNED host tensor (device-mapped).'''
    if not (dst.is_cuda or dst.is_pinned()):
        raise ValueError('index_copy: the destination must be device memory or page-locked host memory')
    n = (src_index if src_index is not None else dst_index if dst_index is not None else src).numel()
    if not n:
        return
    _lib.call('nh_index_copy', n, device.ptr(src), device.ptr(src_index), device.ptr(dst_index), device.ptr(dst), device.stream())


def pointwise_poly(xs, strides, coeffs, powers, n):
    '''out[i] = sum_t coeffs[t] prod_v xs[v][i*strides[v]]^powers[t][v] (nh_pointwise_poly).'''
    nv, nt = len(xs), len(coeffs)
    out = device.empty(n, 'float64')
    X = (ctypes.c_void_p * max(nv, 1))(*[x.data_ptr() for x in xs])
    S = (ctypes.c_int * max(nv, 1))(*[int(s) for s in strides])
    C = (ctypes.c_double * max(nt, 1))(*[float(c) for c in coeffs])
    P = (ctypes.c_int * max(nt * nv, 1))(*[int(p) for row in powers for p in row])
    _lib.call('nh_pointwise_poly', n, nv, X, S, nt, C, P, device.ptr(out), device.stream())
    return out


def point_expr(xs, strides, out_index, offsets, coef, n, nout, scale=None, out=None):
    '''out[i][f] (+)= scale_i sum_{t: out_index[t] == f} coef[t] prod_v xs[v][i*strides[v] + offsets[t][v]] (nh_point_expr): the contraction of field values at
    the points of a sample with a sparse constant tensor.  out_index (int32, ascending), offsets (int32 [nentries][nvars]), coef: device tables; `out`: added to if given.'''
    nv = len(xs)
    fresh = out is None
    if fresh:
        out = device.empty(n * nout, 'float64')
    X = (ctypes.c_void_p * max(nv, 1))(*[x.data_ptr() for x in xs])
    S = (ctypes.c_int64 * max(nv, 1))(*[int(s) for s in strides])
    _lib.call('nh_point_expr', n, nv, X, S, int(coef.numel()), device.ptr(out_index), device.ptr(offsets) if nv else None, device.ptr(coef),
              device.ptr(scale) if scale is not None else None, nout, device.ptr(out), 0 if fresh else 1, device.stream())
    return out


def point_forms(kind, Ut, B, Ur=None, L=None, scale=None):
    '''Per-point forms of field values (nh_point_forms): kind 0 -> [npoints] point factor Ut.B.Ur; kind 1 / 2 -> [npoints][S][S] coefficient
    tensors of the product-rule terms (sample._MatrixPlan.run).  Ut, Ur: [npoints][S] on the device; B [S][S], L [S] on the host.'''
    n, S = int(Ut.shape[0]), int(Ut.shape[1])
    B = numpy.ascontiguousarray(B, dtype=float).reshape(S, S)
    out = device.empty(n if kind == 0 else n * S * S, 'float64')
    Bc = (ctypes.c_double * (S * S))(*B.ravel())
    Lc = (ctypes.c_double * S)(*numpy.asarray(L, dtype=float).ravel()) if L is not None else None
    _lib.call('nh_point_forms', kind, n, S, device.ptr(Ut), device.ptr(Ur) if Ur is not None else None, Bc, Lc,
              device.ptr(scale) if scale is not None else None, device.ptr(out), device.stream())
    return out
