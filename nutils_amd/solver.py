'''System: the caller of the assembly path, mirroring the part of
/root/reference/src/nutils/solver.py:189-431 that orchestrates (re)assembly:
``System(functional_or_residual, trial, test)`` builds residual and Jacobian blocks by
differentiation (solver.py:238,253), (re)assembles them on the GPU for the current
arguments, merges the blocks (``matrix.assemble_block_csr``, matrix/__init__.py:103-151),
applies constraints on the host (NaN = free dof, solver.py:273-315) and hands the
matrix to the host solver (scipy).  Linear systems are solved directly, nonlinear ones
(field-dependent coefficient functions) by plain Newton iteration (solver.py Newton
without line search); minimisation / pseudo-time drivers stay with the reference.'''

import numpy

from . import function, matrix as _matrix, sample as _sample


class SolverError(Exception):
    pass


def _names(spec):
    return tuple(spec.split(',')) if isinstance(spec, str) else tuple(spec)


def _field_dependent(itg):
    '''Does the term's coefficient depend on the arguments of the call (a polynomial of field values, point variables, product-rule tensors)?'''
    return itg.fscale is not None or bool(itg.pvars) or itg.qform is not None or itg.qscalar is not None


class _HostMirror:
    '''The CSR value array of the (merged, possibly reduced) Jacobian in page-locked host memory, kept up to date in place: the first
    assembly copies everything, a later one lets the device write only the entries of the field-dependent blocks (compact array `dyn`,
    entries `sel` of it go to positions `out`) through the mapping of the pinned pages (nh_index_copy) -- for Cahn-Hilliard a quarter
    of the 0.2 GB.  Two buffers alternate, so the matrix of a step stays valid while the next one is assembled, but not longer.  That
    is the contract of the INTERNAL path only (`System._start_jacobian`, used by `solve`, which consumes each Jacobian inside its own
    iteration): the public `assemble_jacobian*` methods hand out value arrays of their own (`copy=True`), like the reference
    (evaluable.py:6813-6815: fresh arrays per evaluation).'''

    def __init__(self, values_dev, sel, out, copies=2):
        from . import device
        t = device.torch()
        self.bufs = [t.empty(values_dev.shape, dtype=values_dev.dtype, pin_memory=True) for _ in range(copies)]
        for buf in self.bufs:
            buf.copy_(values_dev)
        self.sel, self.out, self.turn, self.side = sel, out, 0, None

    def first(self):
        self.turn = 1 % len(self.bufs)
        return self.bufs[0].numpy()

    def publish(self, dyn):
        '''Starts the copy of the changed entries on a side stream (the residual of the same Newton step is assembled meanwhile: its kernels
        run beside the PCIe writes); returns finish() -> host value array.'''
        from . import device, kernels
        t = device.torch()
        buf = self.bufs[self.turn]
        self.turn = (self.turn + 1) % len(self.bufs)
        if not self.out.numel():
            return buf.numpy
        if self.side is None:
            self.side = t.cuda.Stream()
        self.side.wait_stream(t.cuda.current_stream())
        with t.cuda.stream(self.side):
            kernels.index_copy(dyn, buf, src_index=self.sel, dst_index=self.out)
        # `dyn` comes from the caching allocator of the launch stream: tell it that the side stream reads the block, so that it is
        # not handed out again (and overwritten) before the copy has run, whether or not finish() is ever called
        dyn.record_stream(self.side)

        def finish():
            self.side.synchronize()
            return buf.numpy()
        return finish

    def drain(self):
        '''Wait for a copy in flight (error paths: the pinned buffer must not be written behind the caller's back).'''
        if self.side is not None:
            self.side.synchronize()


class System:

    def __init__(self, residual, /, trial, test=None):
        if not isinstance(residual, function.Integral):
            raise TypeError('System expects an Integral (sum of sample.integral terms)')
        self.trials = _names(trial)
        tests = self.trials if test is None else _names(test)
        if len(tests) != len(self.trials):
            raise ValueError('as many test as trial arguments are required')
        self.is_symmetric = tests == self.trials
        self.value = residual if self.is_symmetric else None
        self.block_residual = [function.derivative(residual, t) for t in tests]
        self.block_jacobian = [[function.derivative(r, t) for t in self.trials] for r in self.block_residual]
        # trial argument objects: found in the Jacobian (column space) or residual terms
        self.trial_args = []
        for j, t in enumerate(self.trials):
            arg = None
            for row in self.block_jacobian:
                for _, itg, _ in row[j].terms:
                    arg = itg.trial
            if arg is None:
                raise ValueError(f'the system does not depend on trial argument {t!r}')
            self.trial_args.append(arg)
        self.trial_shapes = [(a.basis.ndofs, a.ncomp) if a.ncomp > 1 else (a.basis.ndofs,) for a in self.trial_args]
        sizes = [int(numpy.prod(s)) for s in self.trial_shapes]
        self.offsets = numpy.cumsum([0] + sizes)
        self.size = int(self.offsets[-1])
        # linear <=> no Jacobian term carries a coefficient function of a trial field (solver.py:255-256)
        # (point variables -- function.Integrand.pvars: values / gradients of bound fields as factors -- are coefficient functions like fscale)
        self.is_linear = not any(itg.fscale is not None and any(itg.fscale.depends_on(t) for t in self.trials) or any(a.name in self.trials for a, _, _ in itg.pvars)
                                 for row in self.block_jacobian for blk in row for _, itg, _ in blk.terms)
        self.is_constant_matrix = not any(_field_dependent(itg) for row in self.block_jacobian for blk in row for _, itg, _ in blk.terms)
        self._jac = None

    # -- assembly (solver.py:318-386) --

    def _build_merge_plan(self, arguments):
        '''Symbolic block merge, ONCE per system (the patterns do not change between Newton steps): every (block, sample)
        group of matrix terms is one device assembly; the position of each of its CSR entries in the merged block matrix is
        precomputed.  The groups without a coefficient function are assembled here, once, into `_base`; the others (the
        field-dependent blocks) are re-assembled per step into the compact array of the entries they touch (`_dynpos`), which is
        all that crosses PCIe afterwards (`_HostMirror`).  Replaces the per-row Python loop of matrix.assemble_block_csr
        (matrix/__init__.py:141-147) on the per-step path.'''
        from . import device, kernels
        groups, keys = [], []
        for i, row in enumerate(self.block_jacobian):
            for j, blk in enumerate(row):
                by_sample = {}
                for term in blk.terms:
                    by_sample.setdefault(id(term[0]), []).append(term)
                for terms in by_sample.values():
                    plan = _sample._MatrixPlan(terms)
                    values, rowptr, colidx, ncols = plan.run(arguments)
                    rp, ci = device.to_host(rowptr), device.to_host(colidx)
                    rows = numpy.repeat(numpy.arange(len(rp) - 1, dtype=numpy.int64), numpy.diff(rp)) + int(self.offsets[i])
                    key = rows * self.size + ci + int(self.offsets[j])
                    constant = not any(_field_dependent(itg) for _, itg, _ in terms)
                    groups.append(dict(plan=plan, constant=constant, values=values if constant else None, n=len(key)))
                    keys.append(key)
        allkeys = numpy.concatenate(keys) if keys else numpy.zeros(0, dtype=numpy.int64)
        ukeys = numpy.unique(allkeys)
        rows, cols = numpy.divmod(ukeys, self.size)
        self._merged_rowptr = numpy.searchsorted(rows, numpy.arange(self.size + 1)).astype(numpy.int64)
        self._merged_colidx = cols.astype(numpy.int64)
        slots = [numpy.searchsorted(ukeys, key) for key in keys]
        dyn = [slot for grp, slot in zip(groups, slots) if not grp['constant']]
        self._dynpos = numpy.unique(numpy.concatenate(dyn)) if dyn else numpy.zeros(0, dtype=numpy.int64)
        self._dynpos_dev = device.to_dev(self._dynpos, 'int64')
        self._base = device.zeros(len(ukeys), 'float64')
        for grp, slot in zip(groups, slots):
            if grp['constant']:
                kernels.monomial(grp.pop('values'), [], [], self._base, out_index=device.to_dev(slot, 'int64'))
            else:
                dslot = numpy.searchsorted(self._dynpos, slot)
                grp['dslot'] = device.to_dev(dslot, 'int64')
                # the group's entries ARE the dynamic entries, in the same order (one field-dependent block on one sample, Cahn-Hilliard): it is assembled
                # straight into the compact array (no array of its own, no pass that adds it)
                grp['direct'] = len(dslot) == len(self._dynpos) and bool((dslot == numpy.arange(len(dslot))).all())
        self._dyn_base = device.empty(len(self._dynpos), 'float64')
        kernels.index_copy(self._base, self._dyn_base, src_index=self._dynpos_dev)
        self._groups = [grp for grp in groups if not grp['constant']]
        self._mirror = None

    def _dyn_values(self, arguments):
        '''Values of the merged Jacobian at `_dynpos` (the entries that field-dependent blocks contribute to), on the device.'''
        from . import kernels
        dyn = self._dyn_base.clone()
        for grp in self._groups:
            if grp['direct']:
                grp['plan'].run(arguments, into=dyn)
            else:
                kernels.monomial(grp['plan'].run(arguments)[0], [], [], dyn, out_index=grp['dslot'])
        return dyn

    def _merged_values(self, arguments):
        '''Values of the merged block Jacobian on the device (pattern: _merged_rowptr, _merged_colidx).'''
        from . import kernels
        merged = self._base.clone()
        if len(self._dynpos):
            kernels.index_copy(self._dyn_values(arguments), merged, dst_index=self._dynpos_dev)
        return merged

    def assemble_jacobian(self, arguments, copy=True):
        '''The merged block Jacobian as a Matrix.  `copy=True` (default): the matrix owns its value array, like every evaluation of the
        reference (fresh arrays, evaluable.py:6813-6815) -- Jacobians of different iterates can be kept side by side.  `copy=False`: the
        matrix wraps the page-locked array the device writes into; it is overwritten by the assembly after the next one (two arrays
        alternate), which is all a Newton driver needs and saves a host copy of the values (C4: 0.2 GB).'''
        return self._finish_jacobian(self._start_jacobian(arguments, None), copy)

    def assemble_jacobian_free(self, arguments, free, copy=True):
        '''jac.submatrix(free, free) of the reference (solver.py:332,386) WITHOUT the host-side slicing: the positions of the free-free
        entries in the merged value array and the reduced (rowptr, colidx) are computed once per constraint set; a Newton step is the
        device assembly of the field-dependent blocks + the copy of THEIR free-free entries into the host value array.
        `copy`: see assemble_jacobian.'''
        return self._finish_jacobian(self._start_jacobian(arguments, free), copy)

    def _finish_jacobian(self, finish, copy):
        return finish(copy and not self.is_constant_matrix)  # (a constant matrix is assembled once and never written again)

    def _drain(self):
        '''Error path of a Newton step: no device copy may still be writing into a host array when the exception reaches the caller.'''
        for m in (self._mirror, (getattr(self, '_free_plan', None) or {}).get('mirror')):
            if m is not None:
                m.drain()

    def _start_jacobian(self, arguments, free, enqueued=None):
        '''Launches the device work of a (reduced, if `free` is given) Jacobian and returns finish() -> Matrix.  Between the two calls the
        changed entries travel to the host on a side stream; `assemble_jacobian_residual` puts the residual of the step there.
        `enqueued`: called once the assembly kernels are enqueued, before the copy to the host is.'''
        from . import device, kernels
        if free is None and self._jac is not None and self.is_constant_matrix:
            return lambda copy=False: self._jac
        if not hasattr(self, '_groups'):
            self._build_merge_plan(arguments)
        if free is None:
            if self._mirror is None:
                self._mirror = _HostMirror(self._merged_values(arguments), None, self._dynpos_dev, 1 if self.is_constant_matrix else 2)
                first = self._mirror.first()
                jac = _matrix.assemble_csr(first, self._merged_rowptr, self._merged_colidx, self.size)  # (validates the index pair once)
                if self.is_constant_matrix:
                    self._jac = jac
                    return lambda copy=False: jac
                return lambda copy=False: _matrix.reassemble_csr(numpy.array(first), self._merged_rowptr, self._merged_colidx, self.size) if copy else jac
            dyn = self._dyn_values(arguments)
            if enqueued:
                enqueued()
            pending = self._mirror.publish(dyn)
            return lambda copy=False: _matrix.reassemble_csr(numpy.array(pending()) if copy else pending(), self._merged_rowptr, self._merged_colidx, self.size)
        key = free.tobytes()
        plan = getattr(self, '_free_plan', None)
        if plan is None or plan['key'] != key:
            rp, ci = self._merged_rowptr, self._merged_colidx
            rows = numpy.repeat(numpy.arange(self.size, dtype=numpy.int64), numpy.diff(rp))
            keepmask = free[rows] & free[ci]
            keep = numpy.flatnonzero(keepmask)
            newcol = numpy.cumsum(free, dtype=numpy.int64) - 1
            counts = numpy.bincount(rows[keep], minlength=self.size)[free]
            frp = numpy.zeros(len(counts) + 1, dtype=numpy.int64)
            numpy.cumsum(counts, out=frp[1:])
            sel = numpy.flatnonzero(keepmask[self._dynpos])  # entries of the compact dynamic array that survive, and where they go
            newpos = numpy.cumsum(keepmask, dtype=numpy.int64) - 1
            plan = self._free_plan = dict(key=key, keep=device.to_dev(keep, 'int64'), rowptr=frp, colidx=newcol[ci[keep]], n=int(free.sum()), matrix=None,
                                          sel=device.to_dev(sel, 'int64'), out=device.to_dev(newpos[self._dynpos[sel]], 'int64'), mirror=None)
        if plan['matrix'] is not None and self.is_constant_matrix:
            return lambda copy=False: plan['matrix']
        if plan['mirror'] is None:
            values = device.empty(len(plan['colidx']), 'float64')
            kernels.index_copy(self._merged_values(arguments), values, src_index=plan['keep'])
            plan['mirror'] = _HostMirror(values, plan['sel'], plan['out'], 1 if self.is_constant_matrix else 2)
            first = plan['mirror'].first()
            jac = _matrix.assemble_csr(first, plan['rowptr'], plan['colidx'], plan['n'])
            if self.is_constant_matrix:
                plan['matrix'] = jac
                return lambda copy=False: jac
            return lambda copy=False: _matrix.reassemble_csr(numpy.array(first), plan['rowptr'], plan['colidx'], plan['n']) if copy else jac
        dyn = self._dyn_values(arguments)
        if enqueued:
            enqueued()
        pending = plan['mirror'].publish(dyn)
        return lambda copy=False: _matrix.reassemble_csr(numpy.array(pending()) if copy else pending(), plan['rowptr'], plan['colidx'], plan['n'])

    def assemble_residual(self, arguments):
        return self._start_residual(arguments)()

    def _start_residual(self, arguments):
        '''Enqueues the residual on the current stream, returns finish() -> concatenated residual vector.'''
        sizes = [int(n) for n in numpy.diff(self.offsets)]
        try:  # all blocks in one pass: the terms that share a sample go through one element loop (nh_assemble_terms)
            live = [r for r in self.block_residual if r.terms]
            if len(live) == len(sizes):  # one device buffer, one copy: the blocks arrive concatenated
                return _sample.start_blocks(live, arguments, flat=True)
            vals = iter(_sample.evaluate_blocks(live, arguments))
            res = numpy.concatenate([numpy.asarray(next(vals), dtype=float).ravel() if r.terms else numpy.zeros(n) for r, n in zip(self.block_residual, sizes)])
            return lambda: res
        except NotImplementedError:
            pass
        parts = []
        for r, size in zip(self.block_residual, sizes):
            try:  # all terms of a block share the test space: one device accumulator, one copy back
                v = numpy.asarray(_sample.evaluate(r, arguments), dtype=float).ravel() if r.terms else numpy.zeros(size)
            except NotImplementedError:
                v = numpy.zeros(size)
                for term in r.terms:
                    v += numpy.asarray(_sample.evaluate(function.Integral([term]), arguments)).ravel()
            parts.append(v)
        res = numpy.concatenate(parts)
        return lambda: res

    def assemble_jacobian_residual(self, arguments, free=None, copy=True):
        '''Jacobian (reduced to the free dofs if `free` is given) and residual of one Newton step, as the reference evaluates them: in one go
        (solver.py:358-387; its Newton drivers call nothing else, :633,659,751-760).  `copy`: see assemble_jacobian (`solve` passes False).'''
        with _sample.upload_scope():  # (every field is copied to the device once, before the Jacobian entries start to travel the other way)
            # (the fields the Jacobian kernels read go first; the others follow on a stream of their own while those kernels run -- all of them before the copy starts)
            groups = getattr(self, '_groups', None)
            join = []
            if groups:
                _sample.prefetch_arguments([grp['plan'] for grp in groups], arguments)
                late = lambda: join.append(_sample.prefetch_beside(self.block_residual, arguments))
            else:
                _sample.prefetch_arguments(self.block_residual, arguments)
                late = None
            # Jacobian kernels, then -- while the changed entries are written to the host by a side stream -- the residual.  (Measured alternatives,
            # profiles/r02_c4.md: the residual on a third stream beside the Jacobian kernels gains nothing, both fill the register files; kernels
            # next to the host-bound stores run at half speed, the stores back up the write queues of the L2 channels -- still the best order.)
            finish = self._start_jacobian(arguments, free, late)
            try:
                if late and not join:  # (a path of _start_jacobian that enqueues nothing)
                    late()
                for wait in join:
                    wait()
                res = self.assemble_residual(arguments)
            except BaseException:
                self._drain()
                raise
        return self._finish_jacobian(finish, copy), res

    def assemble_value(self, arguments):
        if not self.is_symmetric:
            raise Exception('value is not defined')
        return function.eval(self.value, arguments)

    def assemble_jacobian_residual_value(self, arguments, free=None, copy=True):
        '''Jacobian, residual and value of the functional in one call (solver.py:389-425): the functional goes through the same upload
        scope as the other two, its launches run beside the copy of the Jacobian entries.'''
        if not self.is_symmetric:
            raise Exception('value is not defined')
        with _sample.upload_scope():
            # (the fields the Jacobian kernels read go first; the others follow on a stream of their own while those kernels run -- all of them before the copy starts)
            groups = getattr(self, '_groups', None)
            join = []
            if groups:
                _sample.prefetch_arguments([grp['plan'] for grp in groups], arguments)
                late = lambda: join.append(_sample.prefetch_beside(self.block_residual, arguments))
            else:
                _sample.prefetch_arguments(self.block_residual, arguments)
                late = None
            finish = self._start_jacobian(arguments, free, late)
            try:
                if late and not join:  # (a path of _start_jacobian that enqueues nothing)
                    late()
                for wait in join:
                    wait()
                res = self.assemble_residual(arguments)
                val = function.eval(self.value, arguments)
            except BaseException:
                self._drain()
                raise
        return self._finish_jacobian(finish, copy), res, val

    def assemble(self, arguments, free=None, copy=True):
        '''(jacobian, residual, value or None), solver.py:427-431'''
        if self.is_symmetric:
            return self.assemble_jacobian_residual_value(arguments, free, copy)
        return (*self.assemble_jacobian_residual(arguments, free, copy), None)

    # -- argument packing (solver.py:273-315) --

    def _pack(self, arguments, constrain):
        x = numpy.zeros(self.size)
        free = numpy.ones(self.size, dtype=bool)
        for t, shape, a, b in zip(self.trials, self.trial_shapes, self.offsets, self.offsets[1:]):
            if t in arguments:
                x[a:b] = numpy.asarray(arguments[t], dtype=float).ravel()
            c = (constrain or {}).get(t)
            if c is not None:
                c = numpy.asarray(c).ravel()
                if c.dtype == bool:
                    free[a:b] = ~c
                else:
                    fixed = ~numpy.isnan(c)
                    x[a:b][fixed] = c[fixed]
                    free[a:b] = ~fixed
        return x, free

    def _unpack(self, arguments, x):
        out = dict(arguments)
        for t, shape, a, b in zip(self.trials, self.trial_shapes, self.offsets, self.offsets[1:]):
            out[t] = x[a:b].reshape(shape)
        return out

    # -- solves --

    def solve(self, *, arguments=None, constrain=None, tol=0., maxiter=25):
        '''Direct (linear) or Newton (nonlinear) solve; `constrain[trial]` holds NaN for free dofs (solver.py:440-500).'''
        x, free = self._pack(dict(arguments or {}), constrain)
        args = self._unpack(dict(arguments or {}), x)
        if not self.is_linear and tol <= 0:
            raise ValueError('iterative solver requires a strictly positive tolerance')
        sub = None if free.all() else free  # constraint elimination on the device: only the free-free block travels to the host solver
        for it in range(maxiter + 1):
            if self.is_linear:
                res, jac = self.assemble_residual(args), None
            else:  # residual and Jacobian of the iterate in one go, like the reference (the last Jacobian is not used)
                jac, res = self.assemble_jacobian_residual(args, sub, copy=False)  # (consumed before the next assembly)
            resnorm = numpy.linalg.norm(res[free])
            if it and (self.is_linear or resnorm <= tol):
                break
            if not it and not self.is_linear and resnorm <= tol:
                break
            if it == maxiter:
                raise SolverError(f'failed to converge in {maxiter} iterations (residual norm {resnorm:.1e})')
            if jac is None:
                jac = self._start_jacobian(args, sub)()  # (zero-copy: used before the next assembly)
            if sub is None:
                x = x - jac.solve(res)
            else:
                x[free] -= jac.solve(res[free])
            args = self._unpack(args, x)
        return args

    def step(self, *, arguments, suffix, timestep=None, timesteparg=None, **solveargs):
        '''Advance a time step (solver.py:503-560): copies trial arguments to name+suffix, then solves.'''
        arguments = dict(arguments)
        for t in self.trials:
            if t in arguments:
                arguments[t + suffix] = arguments[t]
        if timesteparg:
            arguments[timesteparg] = timestep
        return self.solve(arguments=arguments, **solveargs)

    def solve_constraints(self, *, droptol, arguments=None, constrain=None):
        '''Dirichlet constraints by boundary projection (solver.py:562-612): solve on the dofs whose matrix column has an
        entry above droptol, return NaN for all others.'''
        if not self.is_linear:
            raise ValueError('system is not linear')
        x, free = self._pack(dict(arguments or {}), constrain)
        args = self._unpack(dict(arguments or {}), x)
        jac, res = self.assemble_jacobian_residual(args)
        data, colidx, _ = jac.export('csr')
        mycons = numpy.ones(self.size, dtype=bool)
        mycons[colidx[abs(data) > droptol]] = False
        mycons |= ~free
        dx = -jac.solve(res, constrain=mycons)
        x = x + dx
        x[mycons & free] = numpy.nan
        out = dict(constrain or {})
        for t, shape, a, b in zip(self.trials, self.trial_shapes, self.offsets, self.offsets[1:]):
            out[t] = x[a:b].reshape(shape)
        return out
