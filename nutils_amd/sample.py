'''Sample: quadrature points on a topology, and the evaluator that turns
integrals into kernel launches.

Mirrors the reference's ``Sample.integrate / integral / eval / bind``
(/root/reference/src/nutils/sample.py:160-232).  Where the reference lowers
``_Integral`` to ``loop_sum(einsum(weights, integrand))`` (sample.py:951-956) and
runs a generated Python loop, this class keeps the per-sample device tables
(quadrature, tabulated bases, geometry, sparsity patterns) in HBM and calls
libnutils_hip.so.  No CPU fallback.
'''

import os

import numpy

from . import device, function, kernels, matrix as _matrix
from .basis import StructuredBasis, PlainBasis, RationalBasis


class _BasisTables:
    '''Device tables of one basis at the sample's points.'''

    def __init__(self, smp, basis):
        pts = smp._points_dev
        nq, nd = smp.points.npoints, smp.ndims
        self.basis = basis
        if isinstance(basis, StructuredBasis):
            coeffs = basis.all_class_coefficients()
            self.T = kernels.tabulate(device.to_dev(coeffs, 'float64'), len(coeffs), coeffs.shape[1], pts, nq, nd)
            start = device.to_dev(numpy.concatenate(basis.start_dofs), 'int32')
            self.dofs = kernels.structured_dofs(basis.shape, basis.nloc, basis.dofs_shape, start, 0, basis.nelems)
            self.tab = device.to_dev(basis.element_classes(), 'int32') if basis.nclasses > 1 else None
            self.off = None
            self.nb = basis.nb
        elif isinstance(basis, PlainBasis):
            dofs, coeffs = basis.concatenated()
            self.T = kernels.tabulate(device.to_dev(coeffs, 'float64'), len(coeffs), coeffs.shape[1], pts, nq, nd)
            self.dofs = device.to_dev(dofs, 'int32')
            sizes = numpy.diff(basis.offsets)
            if len(sizes) and (sizes == sizes[0]).all() and sizes[0] > 0:
                # one element type throughout (an unstructured mesh of hexahedra, say): the uniform representation -- nb functions per element, the
                # table of element e at e * nb -- which the thread-per-element passes and the structured-size kernels take; ragged bases keep offsets
                self.off, self.nb = None, int(sizes[0])
                self.tab = device.to_dev(numpy.arange(len(sizes)), 'int32')
            else:
                self.off = device.to_dev(basis.offsets, 'int64')
                self.tab = None
                self.nb = 0
        elif isinstance(basis, RationalBasis):
            pt = smp.tables(basis.parent)
            ne = basis.nelems
            w = device.to_dev(basis.weights, 'float64')
            W = dW = None
            if basis.W is not None:
                Wt, dWt = smp._per_element(basis.W, 'weight function table', by_list=False), smp._per_element(basis.dW, 'weight function table', by_list=False)
                if Wt.shape != (ne, nq) or dWt.shape != (ne, nq, nd):
                    raise ValueError('weight function tables do not match this sample')
                if smp.elist is not None:
                    Wt = numpy.where(Wt == 0, 1., Wt)  # (rows of unlisted elements: never read)
                W, dW = device.to_dev(Wt, 'float64'), device.to_dev(dWt, 'float64')
            if pt.off is None:  # uniform nb: expand the class tables to per-element tables, then transform in place
                S = 1 + nd
                cls = pt.tab.long() if pt.tab is not None else device.torch().zeros(ne, dtype=device.torch().long, device='cuda')
                self.T = pt.T.view(-1, pt.nb * nq * S)[cls].reshape(-1).contiguous()
                self.tab = device.to_dev(numpy.arange(ne), 'int32')
                self.off, self.nb = None, pt.nb
                kernels.rationalize(self.T, ne, pt.nb, pt.dofs, w, nq, nd, W, dW)
            else:
                self.T = pt.T.clone()
                self.tab, self.off, self.nb = None, pt.off, 0
                kernels.rationalize(self.T, ne, 0, pt.dofs, w, nq, nd, W, dW, off=pt.off)
            self.dofs = pt.dofs
        else:
            raise TypeError(f'unsupported basis type {type(basis).__name__}')
        self.struct = kernels.basis(self.T, self.dofs, nb=self.nb, off=self.off, tab=self.tab)


class Sample:

    def __init__(self, topo, points, elist=None, bnd_axis=-1):
        self.topo = topo
        self.points = points
        self.ndims = topo.ndims
        self.nelems = topo.nelems
        # a sample on part of the topology (boundary faces): element subset + the reference axis that is constant on the face
        self.elist = None if elist is None else numpy.ascontiguousarray(elist, dtype=numpy.int32)
        self.bnd_axis = int(bnd_axis)
        self.nlist = self.nelems if self.elist is None else len(self.elist)
        self.npoints = self.nlist * points.npoints
        self._scales = {}
        self._tables = {}
        self._p1verts = {}
        self._geoms = {}
        self._patterns = {}
        self.__dev = None

    # -- device caches --

    @property
    def _points_dev(self):
        if self.__dev is None:
            self.__dev = device.to_dev(self.points.coords, 'float64'), device.to_dev(self.points.weights, 'float64')
        return self.__dev[0]

    @property
    def _weights_dev(self):
        self._points_dev
        return self.__dev[1]

    @property
    def _elist_dev(self):
        if self.elist is None:
            return None
        if not hasattr(self, '_elist_t'):
            self._elist_t = device.to_dev(self.elist, 'int32')
        return self._elist_t

    def scale(self, pf, fp=None, arguments=None):
        '''Device array [nlist][nq]: coefficient function of x (PointFunc, cached per sample) times polynomial of field
        values (FieldPoly, re-evaluated on the device for the current arguments); None if neither is present.'''
        out = None
        if pf is not None:
            if isinstance(pf, _DeviceTable):
                if pf.dev.numel() != self.nlist * self.points.npoints:
                    raise ValueError('device-side coefficient does not match this sample')
                out = pf.dev
            elif isinstance(pf, function.PointTable):
                if pf.values.size != self.nlist * self.points.npoints:
                    raise ValueError('tabulated coefficient does not match this sample')
                out = _cached(self._scales, pf, lambda: device.to_dev(pf(), 'float64'), limit=16)
            else:
                out = _cached(self._scales, pf, lambda: device.to_dev(pf(self._eval_one(pf.geom, {})), 'float64'), limit=16)
        if fp is not None:
            nq, nd, ne = self.points.npoints, self.ndims, self.nlist
            xs = []
            for arg in fp.args:
                u = _argument_dev(arguments or {}, arg)
                U = device.empty(ne * nq * (1 + nd), 'float64')
                kernels.sample_eval(nelems=ne, ndims=nd, nq=nq, geom=self.geometry(_default_geometry(self.topo)), trial=self.tables(arg.basis).struct,
                                    ncr=1, points=self._points_dev, u=u, U=U, elist=self._elist_dev)
                xs.append(U)
            keys = list(fp.terms)
            f = kernels.pointwise_poly(xs, [1 + nd] * len(xs), [fp.terms[k] for k in keys], keys, ne * nq)
            out = f if out is None else out * f
        return out

    def tables(self, basis):
        if basis.nelems != self.nelems or basis.ndims != self.ndims:
            raise ValueError('basis does not live on the topology of this sample')
        t = self._tables.get(id(basis))
        if t is None:
            t = self._tables[id(basis)] = _BasisTables(self, basis)
        return t

    def _per_element(self, table, what, by_list=True):
        '''A table [elements][points]... as the kernels index it: by ELEMENT of the topology.  Producers that know a sample of an element subset
        only (a boundary side, the cells of one hierarchical level) hand it over by position in the sample's element list (`by_list`: what the
        producer means when both readings fit): spread to element rows.'''
        table = numpy.asarray(table)
        fits_elements = table.shape[:2] == (self.nelems, self.points.npoints)
        fits_list = self.elist is not None and table.shape[:2] == (self.nlist, self.points.npoints)
        if fits_list and (by_list or not fits_elements) and not (fits_elements and numpy.array_equal(self.elist, numpy.arange(self.nelems))):
            if len(numpy.unique(self.elist)) != self.nlist:
                raise NotImplementedError(f'{what} per list position on a sample that lists an element twice')
            if self.nelems * int(numpy.prod(table.shape[1:])) * 8 > 1 << 28:
                raise NotImplementedError(f'{what} per list position of a small subset of a large topology')
            full = numpy.zeros((self.nelems,) + table.shape[1:], dtype=table.dtype)
            if full.ndim == 4:  # (Jacobians of unlisted elements: identity, never read, but never singular either)
                full[:] = numpy.eye(table.shape[2], table.shape[3])
            full[numpy.asarray(self.elist)] = table
            return full
        if fits_elements:
            return table
        raise ValueError(f'{what} does not match this sample')

    def geometry(self, geom):
        def make():
            if isinstance(geom, function.IsoGeometry):
                t = self.tables(geom.basis)
                if not t.nb:
                    raise NotImplementedError('isoparametric geometry on a basis with a varying number of functions per element')
                return kernels.geometry_iso(t.nb, t.T, t.dofs, device.to_dev(geom.verts, 'float64'), self.bnd_axis)
            if isinstance(geom, (function.RectilinearGeometry, function.BoxGeometry)):
                origin, size = geom.element_boxes()
                if len(origin) != self.nelems:
                    raise ValueError('geometry does not match the sample')
                return kernels.geometry_box(device.to_dev(origin, 'float64'), device.to_dev(size, 'float64'), self.bnd_axis)
            if isinstance(geom, function.TabulatedGeometry):
                x, jac = self._per_element(geom.x, 'tabulated geometry'), self._per_element(geom.jac, 'tabulated geometry')
                return kernels.geometry_tab(device.to_dev(jac, 'float64'), device.to_dev(x, 'float64'), self.bnd_axis)
            raise TypeError(f'unsupported geometry {type(geom).__name__}')
        return _cached(self._geoms, geom, make)

    def pattern(self, test_basis, trial_basis):
        key = id(test_basis), id(trial_basis)
        p = self._patterns.get(key)
        if p is None:
            tt, tr = self.tables(test_basis), self.tables(trial_basis)
            if self.elist is None:
                p = kernels.Pattern(self.nelems, test_basis.ndofs, trial_basis.ndofs, tt.dofs, tr.dofs, nbt=tt.nb, nbr=tr.nb, toff=tt.off, roff=tr.off)
            else:  # pattern of the listed elements only (emap is then indexed by list position)
                if not (tt.nb and tr.nb):
                    raise NotImplementedError('element subsets with ragged bases')
                idx = self._elist_dev.long()
                td = tt.dofs.view(self.nelems, tt.nb)[idx].reshape(-1).contiguous()
                rd = tr.dofs.view(self.nelems, tr.nb)[idx].reshape(-1).contiguous()
                p = kernels.Pattern(self.nlist, test_basis.ndofs, trial_basis.ndofs, td, rd, nbt=tt.nb, nbr=tr.nb)
            self._patterns[key] = p
        return p

    # -- reference interface --

    def integral(self, func):
        '''Postponed integration (sample.py:177-190).'''
        if isinstance(func, function.IntegrandSum):
            total = None
            for t in func.terms:
                part = self.integral(t)
                total = part if total is None else total + part
            return total
        itg = function._as_integrand(func)
        if itg._tensor.ndim - itg._keep:
            raise NotImplementedError(f'integrand has unreduced axes {itg.shape}; sum or contract them first')
        return function.Integral([(self, itg, 1.)])

    def integrate(self, funcs, /, arguments=None, **kwargs):
        '''Integrate functions (sample.py:160-175).'''
        single = not isinstance(funcs, (tuple, list))
        items = (funcs,) if single else tuple(funcs)
        out = function.eval([self.integral(f) for f in items], dict(arguments or {}, **kwargs))
        return out[0] if single else out

    def eval(self, funcs, /, arguments=None, **kwargs):
        '''Evaluate at all sample points, element-major (sample.py:192-204 / bind
        :217-232, _ConcatenatePoints :966-975).  Supported: a geometry (coordinates),
        J(geom), a field, grad(field, geom).'''
        arguments = dict(arguments or {}, **kwargs)
        single = not isinstance(funcs, (tuple, list))
        items = (funcs,) if single else tuple(funcs)
        out = tuple(self._eval_one(f, arguments) for f in items)
        return out[0] if single else out

    def bind(self, func, /):
        '''Bind the sample to a function (sample.py:217-237): `function.eval(sample.bind(f), **args)` and `sample.bind(f).eval(**args)` are
        `sample.eval(f, **args)`, evaluated when asked for -- together with integrals in one `function.eval` call.'''
        return _Bound(self, func)

    def _eval_one(self, f, arguments):
        nq, nd, ne = self.points.npoints, self.ndims, self.nlist
        n = ne * nq
        el = self._elist_dev
        if isinstance(f, function.Geometry):
            x = device.empty(n * nd, 'float64')
            kernels.sample_eval(nelems=ne, ndims=nd, nq=nq, geom=self.geometry(f), points=self._points_dev, x=x, elist=el)
            return device.to_host(x).reshape(n, nd)
        if isinstance(f, function.Measure):
            dj = device.empty(n, 'float64')
            kernels.sample_eval(nelems=ne, ndims=nd, nq=nq, geom=self.geometry(f.geom), points=self._points_dev, detj=dj, elist=el)
            return device.to_host(dj)
        if isinstance(f, function.PointExpr):
            return self._eval_expr(f, arguments)
        if isinstance(f, function.Operand):
            if f.arg.name is None:
                raise NotImplementedError('evaluating a basis array at all points is a dense (npoints x ndofs) result; evaluate a field instead')
            u = _argument(arguments, f.arg)
            geom = f.geom
            if geom is None:
                geom = _default_geometry(self.topo)
            nc = f.arg.ncomp
            U = device.empty(n * nc * (1 + nd), 'float64')
            kernels.sample_eval(nelems=ne, ndims=nd, nq=nq, geom=self.geometry(geom), trial=self.tables(f.arg.basis).struct, ncr=nc,
                                points=self._points_dev, u=device.to_dev(u, 'float64'), U=U, elist=el)
            Uh = device.to_host(U).reshape(n, nc, 1 + nd)
            # tiny host contraction with the operand's constant tensor: value[free] = P[free,c,s] U[c,s]
            return numpy.einsum('...cs,ncs->n...', f.P, Uh)
        raise NotImplementedError(f'Sample.eval of {type(f).__name__}')


def _eval_expr(self, f, arguments):
    '''Sample.eval of a function.PointExpr: field values / coordinates at the points once per (field, geometry) (nh_sample_eval), then one contraction with the
    sparse coefficient tensor per term (nh_point_expr); -> host array [npoints, *shape].'''
    nq, nd, ne = self.points.npoints, self.ndims, self.nlist
    n, S = ne * nq, 1 + nd
    nout = int(numpy.prod(f.shape)) if f.shape else 1
    el = self._elist_dev
    cache, out = {}, None
    for term in f.terms:
        oidx, off, coef, free = term.entries(nd)
        if tuple(free) != f.shape:
            raise ValueError(f'term of shape {tuple(free)} in an expression of shape {f.shape}')
        if not len(coef):
            continue
        xs, strides = [], []
        for fac in term.factors:
            geom = fac[-1] if fac[-1] is not None else _default_geometry(self.topo)
            if fac[0] == 'field':
                arg = fac[1]
                key = 'f', arg.name, id(arg.basis), arg.ncomp, id(geom)
                if key not in cache:
                    U = device.empty(n * arg.ncomp * S, 'float64')
                    kernels.sample_eval(nelems=ne, ndims=nd, nq=nq, geom=self.geometry(geom), trial=self.tables(arg.basis).struct, ncr=arg.ncomp,
                                        points=self._points_dev, u=_argument_dev(arguments, arg), U=U, elist=el)
                    cache[key] = U
                xs.append(cache[key])
                strides.append(arg.ncomp * S)
            elif fac[0] == 'x':
                key = 'x', id(geom)
                if key not in cache:
                    X = device.empty(n * nd, 'float64')
                    kernels.sample_eval(nelems=ne, ndims=nd, nq=nq, geom=self.geometry(geom), points=self._points_dev, x=X, elist=el)
                    cache[key] = X
                xs.append(cache[key])
                strides.append(nd)
            else:
                raise ValueError(f'unknown factor kind {fac[0]!r}')
        tabs = term.__dict__.get('_dev')
        if tabs is None or tabs[0] != nd:
            tabs = term.__dict__['_dev'] = (nd, device.to_dev(oidx, 'int32'), device.to_dev(off.reshape(-1), 'int32'), device.to_dev(coef, 'float64'))
        out = kernels.point_expr(xs, strides, tabs[1], tabs[2], tabs[3], n, nout, scale=None if term.scale is None else self.scale(term.scale), out=out)
    if out is None:
        return numpy.zeros((n,) + f.shape)
    return device.to_host(out).reshape((n,) + f.shape)


Sample._eval_expr = _eval_expr


class _Bound:
    '''sample.bind(func): the values of `func` at all points of `sample` (point axis first), evaluated by function.eval.'''

    def __init__(self, sample, func):
        self.sample, self.func = sample, func

    def eval(self, arguments=None, **kwargs):
        return self.sample._eval_one(self.func, dict(arguments or {}, **kwargs))


def _default_geometry(topo):
    if not hasattr(topo, '_default_geom'):
        topo._default_geom = getattr(topo, 'geom', None) or function.RectilinearGeometry(topo, numpy.zeros(topo.ndims), numpy.ones(topo.ndims))
    return topo._default_geom


def _argument(arguments, arg):
    if arg.name not in arguments:
        raise KeyError(f'argument {arg.name!r} missing')
    u = numpy.asarray(arguments[arg.name], dtype=float)
    expect = (arg.basis.ndofs, arg.ncomp) if arg.ncomp > 1 or u.ndim == 2 else (arg.basis.ndofs,)
    if u.shape != expect:
        raise ValueError(f'argument {arg.name!r} has shape {u.shape}, expected {expect}')
    return u.reshape(arg.basis.ndofs, arg.ncomp)


_UPLOADS = None  # inside an upload_scope: argument name -> (host array, device tensor)


class upload_scope:
    '''Within the scope an argument array is copied to the device once (keyed by its name and the identity of the array): the Jacobian and the
    residual of one Newton step share the uploads of the fields they both depend on.'''

    def __enter__(self):
        global _UPLOADS
        self.outer = _UPLOADS
        if _UPLOADS is None:
            _UPLOADS = {}

    def __exit__(self, *exc):
        global _UPLOADS
        _UPLOADS = self.outer


def prefetch_arguments(integrals, arguments):
    '''Inside an upload_scope: start the copies of every bound field the terms of `integrals` read, before anything else of the step is enqueued.
    (A Newton step streams the changed Jacobian entries to the host while the residual is assembled; host-to-device copies issued during that
    time wait behind the posted PCIe writes.)'''
    for f in integrals:
        for _, itg, _ in getattr(f, 'terms', ()):
            need = [itg.trial if itg.B is not None else None, itg.test if not itg.rows else None]
            need += list(itg.fscale.args) if itg.fscale is not None else []
            need += [itg.qscalar[1], itg.qscalar[2]] if itg.qscalar is not None else []
            need += [a for a, _, _ in itg.pvars]
            for arg in need:
                if arg is not None and getattr(arg, 'name', None) in arguments:
                    _argument_dev(arguments, arg)


_SIDE_UPLOADS = None


def prefetch_beside(integrals, arguments):
    '''prefetch_arguments on a stream of its own: the copies run beside the kernels already enqueued on the launch stream (the Jacobian pass of a Newton step) instead of
    in front of them.  -> join(): makes the launch stream wait for the copies (call it before anything that reads the fields is enqueued).'''
    global _SIDE_UPLOADS
    t = device.torch()
    if _SIDE_UPLOADS is None:
        _SIDE_UPLOADS = t.cuda.Stream()
    main = t.cuda.current_stream()
    before = set(_UPLOADS or ())
    with t.cuda.stream(_SIDE_UPLOADS):
        prefetch_arguments(integrals, arguments)
    for name in set(_UPLOADS or ()) - before:  # (allocated on the side stream, read on the launch stream)
        _UPLOADS[name][1].record_stream(main)
    return lambda: main.wait_stream(_SIDE_UPLOADS)


def _argument_dev(arguments, arg):
    '''Device copy [ndofs][ncomp] of the array bound to `arg`.'''
    u = _argument(arguments, arg)
    if _UPLOADS is None:
        return device.to_dev(u, 'float64')
    src = arguments[arg.name]
    hit = _UPLOADS.get(arg.name)
    if hit is None or hit[0] is not src:
        hit = _UPLOADS[arg.name] = (src, device.to_dev(u, 'float64'))
    return hit[1]


# Element colouring (structured bases): elements whose multi-indices are congruent modulo the dof-overlap stride share no
# dof, so one launch per colour may add into the CSR values with plain loads/stores (NH_MATRIX_EXCLUSIVE): deterministic and
# at HBM rate instead of memory-side f64 atomics.  Used above COLOR_THRESHOLD elements when the local matrix is large enough
# for the MFMA kernel (below that the launch overhead of 2^d..(p+1)^d launches does not pay).
COLOR_THRESHOLD = 4096


def _colors(smp, basis):
    if not isinstance(basis, StructuredBasis) or smp.elist is not None:
        return None
    stride = 2 if basis.btype == 'std' else basis.degree + 1
    if any(basis.shape[i] % stride for i in basis.periodic):
        return None  # (periodic axis: the first and the last elements share dofs; congruent indices are disjoint only if the stride divides n)
    key = 'colors', id(basis)
    if key not in smp._tables:
        idx = numpy.arange(basis.nelems, dtype=numpy.int32).reshape(basis.shape)
        lists = []
        for off in numpy.ndindex(*(min(stride, n) for n in basis.shape)):
            sub = idx[tuple(slice(o, None, stride) for o in off)].ravel()
            if len(sub):
                lists.append(device.to_dev(sub, 'int32'))
        smp._tables[key] = lists
    return smp._tables[key]


def _cached(cache, obj, make, limit=4):
    '''Device-side companion of a host object (geometry, coefficient function), cached by identity.  The entry keeps the object alive
    -- so its id cannot be recycled for a different object while the entry exists -- and the cache is bounded: scripts that build a
    new geometry object per time step do not accumulate vertex arrays on the device.'''
    entry = cache.get(id(obj))
    if entry is None or entry[0] is not obj:
        value = make()
        if value is None:
            return None
        while len(cache) >= limit:
            cache.pop(next(iter(cache)))
        entry = cache[id(obj)] = (obj, value)
    return entry[1]


def _field_values(smp, arg, geom, arguments):
    '''U[e q][1 + nd]: value and gradient (w.r.t. `geom`) of a scalar field at the points of the sample, on the device.'''
    nq, S = smp.points.npoints, 1 + smp.ndims
    U = device.empty(smp.nlist * nq * S, 'float64')
    kernels.sample_eval(nelems=smp.nlist, ndims=smp.ndims, nq=nq, geom=smp.geometry(geom), trial=smp.tables(arg.basis).struct, ncr=1,
                        points=smp._points_dev, u=_argument_dev(arguments, arg), U=U, elist=smp._elist_dev)
    return U.reshape(smp.nlist * nq, S)


class _DeviceTable(function.PointTable):
    '''a coefficient [nlist][nq] that exists on the device only: the product of a term's point variables for the current arguments (times its tabulated coefficient)'''

    def __init__(self, dev):
        self.dev = dev
        self.geom = None

    @property
    def values(self):
        return device.to_host(self.dev)

    def __call__(self, x=None):
        return device.to_host(self.dev)


def _bind_pvars(terms, arguments):
    '''Terms with point variables (function.Integrand.pvars: components of values / gradients of bound fields as factors) -> the same terms with the product of the
    variables, evaluated on the device for the current arguments, as their pointwise coefficient: nh_sample_eval of every field the variables name (once per (sample,
    field, geometry) of the call), nh_pointwise_poly over the strided (component, slot) columns.  Everything downstream sees an ordinary tabulated coefficient.'''
    if not any(itg.pvars for _, itg, _ in terms):
        return terms
    out, fields = [], {}
    for smp, itg, fac in terms:
        if not itg.pvars:
            out.append((smp, itg, fac))
            continue
        geom = itg.geom if itg.geom is not None else itg.measure
        nq, S, n = smp.points.npoints, 1 + smp.ndims, smp.nlist * smp.points.npoints
        xs, strides = [], []
        for arg, comp, slot in itg.pvars:
            key = id(smp), id(arg.basis), arg.name, arg.ncomp, id(geom)
            U = fields.get(key)
            if U is None:
                U = device.empty(n * arg.ncomp * S, 'float64')
                kernels.sample_eval(nelems=smp.nlist, ndims=smp.ndims, nq=nq, geom=smp.geometry(geom), trial=smp.tables(arg.basis).struct, ncr=arg.ncomp,
                                    points=smp._points_dev, u=_argument_dev(arguments, arg), U=U, elist=smp._elist_dev)
                fields[key] = U
            xs.append(U[comp * S + slot:])
            strides.append(arg.ncomp * S)
        if itg.scale is not None:
            xs.append(smp.scale(itg.scale).reshape(-1))
            strides.append(1)
        while len(xs) > 4:  # (nh_pointwise_poly takes four variables)
            xs, strides = [kernels.pointwise_poly(xs[:4], strides[:4], [1.], [[1] * 4], n)] + xs[4:], [1] + strides[4:]
        prod = kernels.pointwise_poly(xs, strides, [1.], [[1] * len(xs)], n)
        out.append((smp, itg._copy(scale=_DeviceTable(prod), pvars=()), fac))
    return out


def _point_scale(smp, itg, arguments):
    '''scale_dev array of a term: coefficient function of x, polynomial of field values, and (energies) the point factor U_t . B . U_r.'''
    sc = smp.scale(itg.scale, itg.fscale, arguments)
    if itg.qscalar is not None:
        Bs, at, ar = itg.qscalar
        geom = itg.geom if itg.geom is not None else itg.measure
        Ut = _field_values(smp, at, geom, arguments)
        Ur = Ut if ar is at else _field_values(smp, ar, geom, arguments)
        sc = kernels.point_forms(0, Ut, Bs[0, :, 0, :], Ur=Ur, scale=None if sc is None else sc.reshape(-1))  # scalar fields: B [S][S]
    return sc


def _block_mask(B):
    return (numpy.abs(B).sum(axis=(1, 3)) != 0)


def _lib_error():
    from ._lib import NutilsHipError
    return NutilsHipError


_MERGES = {}  # (index arrays of the parts) -> union pattern + positions: symbolic merges of multi-sample matrix integrals, least recently used first
MERGE_CACHE_SIZE = 32


class _MatrixPlan:
    '''All matrix-type terms of one integral share test/trial basis; accumulate into one values buffer.'''

    def __init__(self, terms):
        smp0, itg0, _ = terms[0]
        by_sample = {}
        for term in terms:
            smp, itg, fac = term
            if itg.B is None or not (itg.rows and itg.cols):
                raise ValueError('as_csr needs a matrix-valued integral (both dof axes exposed)')
            group = by_sample.setdefault(id(smp), [])
            group.append(term)
            first = group[0][1]
            # one pair of basis objects per sample; across samples the same dof spaces (a rational basis comes with the weight function tabulated per
            # sample: one object each)
            if not (itg.test.basis is first.test.basis and itg.trial.basis is first.trial.basis and itg.test.ncomp == itg0.test.ncomp
                    and itg.trial.ncomp == itg0.trial.ncomp and itg.test.basis.ndofs == itg0.test.basis.ndofs and itg.trial.basis.ndofs == itg0.trial.basis.ndofs):
                raise NotImplementedError('matrix terms with different bases')
        self.terms = terms
        self.test, self.trial = itg0.test, itg0.trial
        self.parts = None
        if len(by_sample) > 1:
            # Terms on several samples (a volume integral + boundary terms in one form: evaluable.py:6841-6895 runs one loop per sample and the
            # dedup of evaluable.py:588-616 sees the triplets of all of them): one plan per sample, the CSR of the sum is the UNION of their
            # patterns.  Largest sample first -- its pattern usually contains the others (a boundary term couples dofs of one parent element).
            groups = sorted(by_sample.values(), key=lambda ts: -ts[0][0].nlist)
            self.parts = [_MatrixPlan(ts) for ts in groups]
            self.mask = numpy.logical_or.reduce([p.mask for p in self.parts])
            self.smp0 = self.parts[0].smp0
            return
        self.mask = numpy.zeros((self.test.ncomp, self.trial.ncomp), dtype=bool)
        for smp, itg, fac in terms:
            if itg.qform is not None:  # scalar fields: one block
                self.mask |= True
            else:
                self.mask |= _block_mask(itg.B)
        self.smp0 = smp0

    def _p1hex_laplace(self, arguments=None):
        '''Recognise the headline form -- scalar stiffness (+ mass) `(kappa grad(phi_m) . grad(phi_n) + mu phi_m phi_n) J(geom)` on the trilinear 'std'
        basis of a full 3-D structured topology, 2-point Gauss per axis, geometry either rectilinear or the isoparametric P1
        map -- and assemble it with the write-once structured kernel (nh_p1hex_pattern / nh_p1hex_laplace) instead of the
        generic one.  Anything else (other bases, coefficients, samples, extra terms) returns None.'''
        from . import points as _points
        basis, smp = self.test.basis, self.smp0
        if not (basis is self.trial.basis and isinstance(basis, StructuredBasis) and basis.btype == 'std' and basis.degree == 1
                and basis.ndims == 3 and basis.dofs_shape == tuple(n + 1 for n in basis.shape) and self.test.ncomp == self.trial.ncomp == 1):
            return None
        if smp.elist is not None or smp.bnd_axis >= 0 or smp.points.npoints != 8 or os.environ.get('NUTILS_AMD_NO_FAST_PATH'):
            return None
        x1, w1 = _points.gauss1(2)
        ref = _points.gauss(2, 3) if hasattr(_points, 'gauss') else None
        if ref is None or not (numpy.array_equal(smp.points.coords, ref.coords) and numpy.array_equal(smp.points.weights, ref.weights)):
            return None
        kappa, mass, geom, qscale, qmass = 0., 0., None, None, None
        rest = []  # terms of another shape (advection, product-rule tensors ...): added by the generic kernel afterwards
        for term in self.terms:
            _, itg, fac = term
            B = numpy.asarray(itg.B, dtype=float) * fac
            ok = B.shape == (1, 4, 1, 4) and itg.qform is None
            if ok:
                B = B[0, :, 0, :]
                m, k = B[0, 0], B[1, 1]  # mass and diffusion coefficient of the term: B = diag(m, k, k, k)
                ok = numpy.array_equal(B, numpy.diag([m, k, k, k])) and (geom is None or itg.measure is geom)
            if not ok:
                rest.append(term)
                continue
            geom = itg.measure
            if itg.scale is None and itg.fscale is None and itg.qscalar is None:
                kappa += k
                mass += m
            else:  # coefficient function (of position, or of a field): values at the Gauss points, summed over the terms
                sc = _point_scale(smp, itg, arguments)
                if k:
                    qscale = sc * k if qscale is None else qscale.add_(sc, alpha=k)
                if m:
                    qmass = sc * m if qmass is None else qmass.add_(sc, alpha=m)
        if geom is None:
            return None
        if qscale is not None:
            if kappa:
                qscale = qscale + kappa
            kappa = 1.
        if qmass is not None:
            if mass:
                qmass = qmass + mass
            mass = 1.
        verts, origin, scale = None, (0., 0., 0.), (1., 1., 1.)
        if isinstance(geom, function.IsoGeometry):
            g = geom.basis
            if not (isinstance(g, StructuredBasis) and g.btype == 'std' and g.degree == 1 and g.shape == basis.shape and g.dofs_shape == basis.dofs_shape):
                return None
            verts = _cached(smp._p1verts, geom, lambda: device.to_dev(geom.verts, 'float64'))
        elif isinstance(geom, function.RectilinearGeometry) and geom.topo.shape == basis.shape:
            origin, scale = tuple(geom.offset), tuple(geom.scale)
            if qscale is not None or qmass is not None:  # element matrices differ: explicit vertices for the isoparametric kernel
                verts = _cached(smp._p1verts, geom, lambda: _rectilinear_vertices(geom, basis.shape))
        elif isinstance(geom, function.GradedGeometry) and geom.topo.shape == basis.shape and not basis.periodic and geom.size.all():  # (flat elements: generic path, NaN rules of numeric.inv)
            verts = _cached(smp._p1verts, geom, lambda: device.to_dev(geom.vertices(), 'float64'))  # graded mesh: the isoparametric kernel on its vertices
        else:
            return None
        key = 'p1hex_pattern', basis.shape
        if key not in smp._tables:
            smp._tables[key] = kernels.p1hex_pattern(basis.shape)
        rowptr, colidx = smp._tables[key]
        values = device.empty(colidx.numel(), 'float64')  # write-once kernel: no zero-fill
        kernels.p1hex_laplace(shape=basis.shape, values=values, gauss_x=list(x1), gauss_w=list(w1), verts=verts, origin=origin, scale=scale, kappa=kappa,
                              qscale=qscale, mass=mass, qmass=qmass)
        return values, rowptr, colidx, basis.ndofs, rest

    def _p2hex(self, rowptr, colidx):
        '''Constant-coefficient forms (summed over the terms) on the quadratic 'std' basis of a full 3-D structured topology, scalar or
        vector valued with all component blocks coupled -- BASELINE.json configs[2], the 3-D elasticity stiffness matrix -- go to the
        write-once kernel nh_p2hex_matrix: every CSR value is formed in LDS from the (up to 8) elements that touch it and stored once; no
        zero-fill, no colours, no element map.  Returns the value array, or None when the integral is of another kind (then the
        generic path assembles it).'''
        basis, smp = self.test.basis, self.smp0
        nc = self.test.ncomp
        if not (basis is self.trial.basis and isinstance(basis, StructuredBasis) and basis.btype == 'std' and basis.degree == 2 and basis.ndims == 3
                and basis.dofs_shape == tuple(2 * n + 1 for n in basis.shape) and nc == self.trial.ncomp and nc <= 3 and self.mask.all()):
            return None
        if smp.elist is not None or smp.bnd_axis >= 0 or os.environ.get('NUTILS_AMD_NO_FAST_PATH'):
            return None
        C, geom = 0., None
        for _, itg, fac in self.terms:
            if not (itg.qform is None and itg.qscalar is None and itg.scale is None and itg.fscale is None and itg.measure is not None
                    and numpy.shape(itg.B) == (nc, 4, nc, 4) and (geom is None or itg.measure is geom)):
                return None
            geom = itg.measure
            C = C + numpy.asarray(itg.B, dtype=float) * fac
        # launchers by geometry OBJECT (the entry keeps it alive, so its id cannot be recycled for another geometry while the entry exists;
        # at most 4 geometries per sample: a script that builds a new one per step does not pile up vertex arrays on the device), then by form
        if not hasattr(smp, '_p2hex_fns'):
            smp._p2hex_fns = {}
        by_form = _cached(smp._p2hex_fns, geom, dict)
        if len(by_form) >= 8 and C.tobytes() not in by_form:
            by_form.pop(next(iter(by_form)))
        key = C.tobytes()
        fn = by_form.get(key)
        if fn is None:
            try:
                if isinstance(geom, function.RectilinearGeometry) and geom.topo.shape == basis.shape and not os.environ.get('NUTILS_AMD_NO_UNIFORM'):
                    # equidistant vertices: every element matrix is the same -- the rows of the 2 x 2 x 2 mesh of such cells, replicated (a write stream)
                    fn = kernels.P2HexUniform(shape=basis.shape, nq=smp.points.npoints, weights=smp._weights_dev, T=smp.tables(basis).T, ncomp=nc, C=C, cell=geom.scale)
                else:
                    fn = kernels.P2HexMatrix(shape=basis.shape, nq=smp.points.npoints, weights=smp._weights_dev, geom=smp.geometry(geom), T=smp.tables(basis).T,
                                             ncomp=nc, C=C)
                probe = device.empty(colidx.numel(), 'float64')
                fn(probe)  # NH_ELIMIT (tables do not fit the LDS for this quadrature) surfaces here
            except _lib_error() as e:
                if 'LDS' not in str(e):
                    raise
                by_form[key] = fn = False
            else:
                by_form[key] = fn
                return probe
        if fn is False:
            return None
        values = device.empty(colidx.numel(), 'float64')
        fn(values)
        return values

    def _first_touch(self, term):
        '''(elements per axis, local nodes per axis) if the first term is assembled colour by colour on a non-periodic C0 ('std') basis
        whose local order is the tensor order of its nodes: NH_MATRIX_FIRST_TOUCH applies.'''
        smp, itg, fac = term
        basis = itg.test.basis
        if not (isinstance(basis, StructuredBasis) and basis.btype == 'std' and itg.trial.basis is basis and itg.qform is None and itg.measure is not None):
            return None
        if basis.dofs_shape != tuple(n * basis.degree + 1 for n in basis.shape) or os.environ.get('NUTILS_AMD_NO_FIRST_TOUCH'):
            return None  # (periodic axes: the neighbour across the seam)
        if not (smp.nlist >= COLOR_THRESHOLD and basis.nb >= 16 and _colors(smp, basis)) or os.environ.get('NUTILS_AMD_NO_COLORS'):
            return None
        if self._rows_pass(smp, itg):
            return None  # (not assembled colour by colour)
        return basis.shape, basis.degree + 1

    def _rows_pass(self, smp, itg):
        '''Scalar blocks of 27 (3-D quadratic) / 16 (2-D cubic) functions: from the second assembly on, the owner-side reduction with its row-blocked
        thread pass (k_local_rows) instead of the coloured MFMA launches -- 64^3 triquadratic 3.9 against 13.0 ms, 1024^2 bicubic splines 3.3
        against 14.4 ms (tools/generic_probe.py).'''
        if smp.elist is not None or itg.qform is not None:
            return False
        tt, tr = smp.tables(itg.test.basis), smp.tables(itg.trial.basis)
        # ... and vector-valued blocks of 27 functions: the owner kernel (nh_owner.hip, points in chunks) from the FIRST assembly on -- one pass instead of 8 coloured launches
        if (self.test.ncomp == self.trial.ncomp and (smp.ndims, tt.nb, self.test.ncomp) in kernels.OWNER_VECTOR and tt.nb > 9 and itg.test.basis is itg.trial.basis
                and not os.environ.get('NUTILS_AMD_NO_FUSED') and itg.fscale is None and itg.qscalar is None):
            return True
        if self.test.ncomp != 1 or self.trial.ncomp != 1 or os.environ.get('NUTILS_AMD_NO_GATHER'):
            return False
        if (smp.ndims, tt.nb, tr.nb) not in ((2, 16, 16), (3, 27, 27), (2, 25, 25), (3, 64, 64)):  # (the instantiations of k_local_rows, nh_gather.hip)
            return False
        pat = smp.pattern(itg.test.basis, itg.trial.basis)
        return getattr(pat, '_assemblies', 0) >= 1 and 8 * pat.emap_len <= kernels.GATHER_SCRATCH_LIMIT

    def _batched(self, terms, values, mask, arguments, fresh=None):
        '''Bases with few functions per element: ALL terms of the block that share a measure go through one nh_assemble_matrix_terms launch
        (several elements per workgroup; coefficient functions, polynomial factors of field values and the point-dependent product-rule
        tensors are evaluated inside the kernel).  Returns the terms that remain for the per-term kernels.'''
        smp0 = self.smp0
        tt, tr = smp0.tables(self.test.basis), smp0.tables(self.trial.basis)
        if not (0 < tt.nb <= 27 and 0 < tr.nb <= 27):
            return terms
        if self.test.basis is self.trial.basis and smp0.nlist >= COLOR_THRESHOLD and tt.nb >= 16:
            return terms  # (the coloured MFMA path)
        # measured (profiles/r02_c4.md): the fused kernel pays when it replaces several launches -- more than one term, or pointwise factors that
        # would otherwise be evaluated by nh_sample_eval / nh_pointwise_poly / torch algebra first; one constant-coefficient term of a 3-D basis
        # is faster through the one-wave-per-element kernel (128^3 trilinear: 4.1 against 9.5 ms)
        if len(terms) < 2 and not any(itg.fscale is not None or itg.qform is not None or itg.qscalar is not None for _, itg, _ in terms):
            return terms
        nd, nq, S = smp0.ndims, smp0.points.npoints, 1 + smp0.ndims
        nct, ncr = self.test.ncomp, self.trial.ncomp
        groups, rest = {}, []
        for term in terms:
            smp, itg, fac = term
            ok = (itg.measure is not None and (itg.geom is None or itg.geom is itg.measure or itg.qform is None)
                  and (itg.fscale is None or (len(itg.fscale.args) <= 4 and len(itg.fscale.terms) <= 64 and all(a.ncomp == 1 for a in itg.fscale.args)))
                  and (itg.qform is None or (nct == ncr == 1 and itg.qform[1].ncomp == 1)))
            (groups.setdefault(id(itg.measure), []) if ok else rest).append(term)
        for items in groups.values():
            fkeys, pkeys, tl, ucache = [], [], [], {}

            def fidx(a):
                for i, k in enumerate(fkeys):
                    if a.same(k):
                        return i
                fkeys.append(a)
                return len(fkeys) - 1

            left = []
            for term in items:
                smp, itg, fac = term
                qs_ok = itg.qscalar is not None and itg.qscalar[1].ncomp == 1 and itg.qscalar[2].ncomp == 1 and (itg.geom is None or itg.geom is itg.measure)
                inkernel = itg.qscalar is None or qs_ok  # (else: the point factor stays a scale array from _point_scale)
                need = (([itg.qform[1]] if itg.qform is not None else []) + (list(itg.fscale.args) if itg.fscale is not None and inkernel else [])
                        + ([itg.qscalar[1], itg.qscalar[2]] if qs_ok else []))
                newf = []
                for a in need:
                    if not any(a.same(k) for k in fkeys + newf):
                        newf.append(a)
                newp = itg.fscale is not None and inkernel and not any(itg.fscale is k for k in pkeys)
                if len(tl) >= 32 or len(fkeys) + len(newf) > 6 or len(pkeys) + newp > 4:
                    left.append(term)
                    continue
                t = dict(C=numpy.asarray(itg.B, dtype=float) * fac)
                if not inkernel:
                    t['scale'] = _point_scale(smp, itg, arguments)
                else:
                    if qs_ok:  # energy Hessians: the point factor U_t . B . U_r, evaluated in the kernel
                        t['qs'] = (numpy.asarray(itg.qscalar[0], dtype=float)[0, :, 0, :], fidx(itg.qscalar[1]), fidx(itg.qscalar[2]))
                    if itg.scale is not None:
                        t['scale'] = smp.scale(itg.scale)
                    if itg.fscale is not None:
                        if newp:
                            pkeys.append(itg.fscale)
                        for a in itg.fscale.args:
                            fidx(a)
                        t['poly'] = next(i for i, k in enumerate(pkeys) if k is itg.fscale)
                if itg.qform is not None:
                    t['kind'], t['field'] = (1 if itg.qform[0] == 'trial' else 2), fidx(itg.qform[1])
                    if itg.qform[0] != 'trial':
                        t['L'] = numpy.asarray(itg.qform[2], dtype=float).reshape(1, S)
                tl.append(t)
            rest += left
            if not tl:
                continue
            smp = items[0][0]
            polys = []
            for fp in pkeys:
                keys = list(fp.terms)
                polys.append(([(fidx(a), 0) for a in fp.args], [fp.terms[k] for k in keys], keys))
            fields = [(smp.tables(a.basis).struct, _argument_dev(arguments, a), a.ncomp) for a in fkeys]
            if fresh is not None and fresh[0]:  # (this entry accumulates)
                values.zero_()
                fresh[0] = False
            kernels.assemble_matrix_terms(nelems=smp.nlist, elist=smp._elist_dev, ndims=nd, nq=nq, weights=smp._weights_dev, geom=smp.geometry(items[0][1].measure),
                                          test=tt.struct, trial=tr.struct, nct=nct, ncr=ncr, mask=mask, pattern=smp.pattern(self.test.basis, self.trial.basis),
                                          values=values, terms=tl, fields=fields, polys=polys)
        return rest

    def _run_parts(self, arguments):
        '''Sum of the per-sample matrices.  The symbolic part (union of the sorted-unique patterns, position of every entry of a part in it: kernels.pattern_union,
        a row-wise merge of sorted column lists on the device) is computed once per set of index arrays -- they are cached by the patterns, so re-assemblies and
        re-BUILT plans (function.eval makes a new _MatrixPlan per call) find the same tensors and hit _MERGES; the values of a re-assembly are added at the
        precomputed positions by nh_monomial (unique positions per part: no atomics, fixed order).'''
        results = [p.run(arguments) for p in self.parts]
        ncols = results[0][3]
        key = tuple((r[1].data_ptr(), r[2].data_ptr(), r[2].numel()) for r in results)
        m = _MERGES.get(key)
        if m is None:
            if any(nc != ncols or rowptr.numel() != results[0][1].numel() for _, rowptr, _, nc in results):
                raise ValueError('matrix terms of different shape in one integral')
            rowptr_u, colidx_u, pos = kernels.pattern_union([(r[1], r[2]) for r in results])
            if colidx_u.numel() == results[0][2].numel():  # the first pattern contains the others: its arrays ARE the union, its values the start of the sum
                rowptr_u, colidx_u, pos[0] = results[0][1], results[0][2], None
            m = dict(keep=[(r[1], r[2]) for r in results], rowptr=rowptr_u, colidx=colidx_u, pos=pos, n=colidx_u.numel())  # (keep: the addresses of the key stay taken)
            _MERGES[key] = m
            while len(_MERGES) > MERGE_CACHE_SIZE:
                _MERGES.pop(next(iter(_MERGES)))
        else:
            _MERGES[key] = _MERGES.pop(key)  # (most recently used last)
        if m['pos'][0] is None:
            out = results[0][0]  # (a fresh array of this assembly)
        else:
            out = device.zeros(m['n'], 'float64')
        for (values, _, _, _), pos in zip(results, m['pos']):
            if pos is not None:
                kernels.monomial(values, [], [], out, out_index=pos)
        return out, m['rowptr'], m['colidx'], ncols

    def run(self, arguments=None, into=None):
        '''-> values, rowptr, colidx, ncols.  `into`: a device array over the plan's own entries that the values are ADDED to (entry-wise: old + sum of the
        contributions, the same rounding as adding the result afterwards) -- the merged Jacobian's array of field-dependent entries (solver._dyn_values): no array
        of its own, no pass that adds it.'''
        if any(itg.pvars for _, itg, _ in self.terms):
            # point variables are products of field values at the points: bound per call (the Jacobian blocks of a native System arrive here unbound,
            # solver._build_merge_plan / _dyn_values), then an ordinary plan over tabulated coefficients
            return _MatrixPlan(_bind_pvars(self.terms, arguments or {})).run(arguments, into)
        if into is not None:
            return self._run_into(arguments, into)
        if self.parts is not None:
            return self._run_parts(arguments)
        fast = self._p1hex_laplace(arguments)
        if fast is not None and not fast[4]:
            return fast[:4]
        pat = self.smp0.pattern(self.test.basis, self.trial.basis)
        nct, ncr = self.test.ncomp, self.trial.ncomp
        mask = None if self.mask.all() else self.mask
        rowptr, colidx = pat.expand(nct, ncr, mask)
        return self._run_generic(arguments, fast, rowptr, colidx, None)

    def _run_into(self, arguments, into):
        generic = self.parts is None and not self._fast_candidates()
        if not generic:  # (paths that write a whole array of their own)
            values, rowptr, colidx, ncols = self.run(arguments)
            if into.numel() != values.numel():
                raise ValueError('run(into=...): the array does not match the pattern')
            into.add_(values)
            return into, rowptr, colidx, ncols
        rowptr, colidx = self.smp0.pattern(self.test.basis, self.trial.basis).expand(self.test.ncomp, self.trial.ncomp, None if self.mask.all() else self.mask)
        if into.numel() != colidx.numel():
            raise ValueError('run(into=...): the array does not match the pattern')
        return self._run_generic(arguments, None, rowptr, colidx, into)

    def _fast_candidates(self):
        '''Could a write-once kernel (nh_p1hex_laplace / nh_p2hex_elasticity) take this plan?  Their recognisers need constant forms: a plan with a
        coefficient function never qualifies.'''
        return not any(itg.fscale is not None or itg.qform is not None or getattr(itg, 'pvars', ()) for _, itg, _ in self.terms)

    def _run_generic(self, arguments, fast, rowptr, colidx, into):
        nct, ncr = self.test.ncomp, self.trial.ncomp
        mask = None if self.mask.all() else self.mask
        first_touch, fresh = None, [False]
        if into is not None:
            values, terms = into, self.terms
            if not os.environ.get('NUTILS_AMD_NO_BATCHED'):
                terms = self._batched(terms, values, mask, arguments, fresh)
            return self._run_terms(arguments, terms, self.terms, values, rowptr, colidx, first_touch, fresh)
        if fast is None:
            values = self._p2hex(rowptr, colidx)
            if values is not None:
                return values, rowptr, colidx, self.trial.basis.ndofs * ncr
        if fast is not None:  # same sorted-unique pattern: the generic kernel accumulates the remaining terms into the write-once result
            values, terms = fast[0], fast[4]
        else:
            # the first term of a coloured assembly on a C0 basis STORES the entries no earlier colour has touched (NH_MATRIX_FIRST_TOUCH):
            # no zero-fill, no read of the entries that receive a single contribution
            first_touch = self._first_touch(self.terms[0])
            # (a FRESH array: the first kernel that can STORES its sums -- gather / owner blocks: no zero fill, no read-modify-write of 8 bytes per entry --
            # every other path zero-fills it first)
            values, terms = device.empty(colidx.numel(), 'float64'), self.terms
            fresh = [not first_touch]
        if first_touch is None and not os.environ.get('NUTILS_AMD_NO_BATCHED'):
            terms = self._batched(terms, values, mask, arguments, fresh)
        return self._run_terms(arguments, terms, self.terms, values, rowptr, colidx, first_touch, fresh)

    def _run_terms(self, arguments, terms, allterms, values, rowptr, colidx, first_touch, fresh):
        nct, ncr = self.test.ncomp, self.trial.ncomp
        mask = None if self.mask.all() else self.mask
        for iterm, (smp, itg, fac) in enumerate(terms):
            if itg.measure is None:
                raise NotImplementedError('integrand without J(geom): reference-space integrals are outside the accelerated path')
            tt, tr = smp.tables(itg.test.basis), smp.tables(itg.trial.basis)
            common = dict(ndims=smp.ndims, nq=smp.points.npoints, weights=smp._weights_dev, geom=smp.geometry(itg.measure), test=tt.struct,
                          trial=tr.struct, nct=nct, ncr=ncr, C=itg.B * fac, mask=mask, pattern=smp.pattern(itg.test.basis, itg.trial.basis),
                          values=values)
            colors = None
            scale = _point_scale(smp, itg, arguments)
            if itg.qform is not None:
                # per-point coefficient tensors built on the device from U = (value, gradient) of the bound field (scalar fields)
                nq, S = smp.points.npoints, 1 + smp.ndims
                U = _field_values(smp, itg.qform[1], itg.geom if itg.geom is not None else itg.measure, arguments)
                Bq = numpy.ascontiguousarray(itg.B[0, :, 0, :]) * fac  # [S][S]
                sq = None if scale is None else scale.reshape(-1)
                if itg.qform[0] == 'trial':   # C_q[a][0] = sum_b B[a][b] U[b]
                    cq = kernels.point_forms(1, U, Bq, scale=sq)
                else:                         # 'test': C_q[a][b] = L[a] sum_x B[x][b] U[x]
                    cq = kernels.point_forms(2, U, Bq, L=itg.qform[2][0], scale=sq)
                common_q = dict(common, C=numpy.ones((nct, S, ncr, S)))
                kernels.assemble_matrix(nelems=smp.nlist, elist=smp._elist_dev, cq=cq, fresh=fresh[0], **common_q)
                fresh[0] = False
                continue
            if (itg.test.basis is itg.trial.basis and smp.nlist >= COLOR_THRESHOLD and tt.nb >= 16 and not self._rows_pass(smp, itg)
                    and not os.environ.get('NUTILS_AMD_NO_COLORS')):  # (small local matrices: 8 coloured launches measured slower than atomics, 4.7 vs 4.0 ms)
                colors = _colors(smp, itg.test.basis)
            if colors:
                if fresh[0]:
                    values.zero_()
                    fresh[0] = False
                ft = first_touch if iterm == 0 and terms is allterms else None
                for el in colors:
                    kernels.assemble_matrix(nelems=el.numel(), elist=el, flags=1 | 2, scale=scale, first_touch=ft, **common)
                common['pattern']._assemblies = getattr(common['pattern'], '_assemblies', 0) + 1  # (a re-assembly may take the gather path: _rows_pass)
            else:
                kernels.assemble_matrix(nelems=smp.nlist, elist=smp._elist_dev, scale=scale, fresh=fresh[0], **common)
                fresh[0] = False
        if fresh[0]:  # (no term at all)
            values.zero_()
        return values, rowptr, colidx, self.trial.basis.ndofs * ncr


def _rectilinear_vertices(geom, shape):
    idx = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, 3)
    return device.to_dev(geom.offset + geom.scale * idx, 'float64')


def _p1hex_setting(smp, basis, geom):
    '''(vertices on the device, 1-D Gauss points and weights) if `basis` is the trilinear 'std' basis of the full 3-D structured
    topology of `smp`, sampled with 2-point Gauss per axis, on a rectilinear or isoparametric-P1 geometry; else None.'''
    from . import points as _points
    if not (isinstance(basis, StructuredBasis) and basis.btype == 'std' and basis.degree == 1 and basis.ndims == 3
            and basis.dofs_shape == tuple(n + 1 for n in basis.shape)):
        return None
    if smp.elist is not None or smp.bnd_axis >= 0 or smp.points.npoints != 8 or os.environ.get('NUTILS_AMD_NO_FAST_PATH'):
        return None
    ref = _points.gauss(2, 3)
    if not (numpy.array_equal(smp.points.coords, ref.coords) and numpy.array_equal(smp.points.weights, ref.weights)):
        return None
    def make():
        if isinstance(geom, function.IsoGeometry):
            g = geom.basis
            if isinstance(g, StructuredBasis) and g.btype == 'std' and g.degree == 1 and g.shape == basis.shape and g.dofs_shape == basis.dofs_shape:
                return device.to_dev(geom.verts, 'float64')
        elif isinstance(geom, function.RectilinearGeometry) and geom.topo.shape == basis.shape:
            return _rectilinear_vertices(geom, basis.shape)
        elif isinstance(geom, function.GradedGeometry) and geom.topo.shape == basis.shape and geom.size.all():
            return device.to_dev(geom.vertices(), 'float64')
        return None
    verts = _cached(smp._p1verts, geom, make)
    if verts is None:
        return None
    x1, w1 = _points.gauss1(2)
    return verts, list(x1), list(w1)


def _p1hex_match(smp, itg, fac):
    '''(mass, kappa, setting) if the term is `(kappa grad(phi_m) . grad(u) + mu phi_m u) J(geom)` of the headline setting, else None.'''
    if not (itg.B is not None and itg.rows and not itg.cols and itg.test.basis is itg.trial.basis and itg.test.ncomp == itg.trial.ncomp == 1):
        return None
    B = numpy.asarray(itg.B, dtype=float) * fac
    if B.shape != (1, 4, 1, 4):
        return None
    B = B[0, :, 0, :]
    m, k = float(B[0, 0]), float(B[1, 1])
    if not numpy.array_equal(B, numpy.diag([m, k, k, k])):
        return None
    setting = _p1hex_setting(smp, itg.test.basis, itg.measure)
    return None if setting is None else (m, k, setting)


def _p1hex_apply_term(smp, itg, fac, arguments, out):
    '''Residual-type term of the headline setting: out += K u through nh_p1hex_apply (the element matrices are applied on the fly, no matrix,
    no global atomics).  Returns False if the term is anything else.'''
    match = _p1hex_match(smp, itg, fac)
    if match is None:
        return False
    m, k, (verts, x1, w1) = match
    u = _argument_dev(arguments, itg.trial)
    sc = _point_scale(smp, itg, arguments)
    kernels.p1hex_apply(shape=itg.test.basis.shape, u=u, out=out, gauss_x=x1, gauss_w=w1, verts=verts, kappa=k, mass=m,
                        qscale=sc if k else None, qmass=sc if m else None, accumulate=True)
    return True


def _vector_term(smp, itg, fac, arguments, out, scalar):
    nd, nq = smp.ndims, smp.points.npoints
    if itg.measure is None:
        raise NotImplementedError('integrand without J(geom)')
    if out is not None and _p1hex_apply_term(smp, itg, fac, arguments, out):
        return
    geom = smp.geometry(itg.measure)
    if out is not None and itg.rows and not itg.cols and _deterministic():
        # linear form into a vector: local vectors + owner-side reduction instead of atomics
        tt = smp.tables(itg.test.basis)
        plan, local = _scatter(smp, itg.test.basis, itg.test.ncomp)
        common = dict(elist=smp._elist_dev, scale=_point_scale(smp, itg, arguments), nelems=smp.nlist, ndims=nd, nq=nq, weights=smp._weights_dev, geom=geom, test=tt.struct,
                      nct=itg.test.ncomp, local=local)
        if itg.B is not None:
            kernels.assemble_vector(trial=smp.tables(itg.trial.basis).struct, ncr=itg.trial.ncomp, C=itg.B * fac, u=_argument_dev(arguments, itg.trial), **common)
        elif itg.L is not None:
            kernels.assemble_vector(trial=tt.struct, ncr=itg.test.ncomp, f=itg.L * fac, **common)
        else:
            raise NotImplementedError('vector term without a form')
        kernels.scatter_gather([(plan, local)], itg.test.ncomp, out, accumulate=True)
        return
    if itg.B is not None:
        tt, tr = smp.tables(itg.test.basis), smp.tables(itg.trial.basis)
        if itg.rows and not itg.cols:
            u = _argument_dev(arguments, itg.trial)
            kernels.assemble_vector(elist=smp._elist_dev, scale=_point_scale(smp, itg, arguments), nelems=smp.nlist, ndims=nd, nq=nq, weights=smp._weights_dev, geom=geom, test=tt.struct, trial=tr.struct,
                                    nct=itg.test.ncomp, ncr=itg.trial.ncomp, C=itg.B * fac, u=u, out=out)
            return
        if not itg.rows and not itg.cols:
            if itg.test.same(itg.trial):
                u = _argument_dev(arguments, itg.trial)
                kernels.assemble_vector(elist=smp._elist_dev, scale=_point_scale(smp, itg, arguments), nelems=smp.nlist, ndims=nd, nq=nq, weights=smp._weights_dev, geom=geom, test=tt.struct, trial=tr.struct,
                                        nct=itg.test.ncomp, ncr=itg.trial.ncomp, C=itg.B * (2 * fac), u=u, out_scalar=scalar[0])
            else:
                tmp = device.zeros(itg.test.basis.ndofs * itg.test.ncomp, 'float64')
                u = _argument_dev(arguments, itg.trial)
                kernels.assemble_vector(elist=smp._elist_dev, scale=_point_scale(smp, itg, arguments), nelems=smp.nlist, ndims=nd, nq=nq, weights=smp._weights_dev, geom=geom, test=tt.struct, trial=tr.struct,
                                        nct=itg.test.ncomp, ncr=itg.trial.ncomp, C=itg.B * fac, u=u, out=tmp)
                # v . r for two different bound fields: O(ndofs) post-processing on the host
                scalar[1] += float(numpy.dot(device.to_host(tmp), _argument(arguments, itg.test).ravel()))
            return
        raise NotImplementedError('unsupported combination of exposed axes')
    if itg.L is not None:
        tt = smp.tables(itg.test.basis)
        if itg.rows:
            kernels.assemble_vector(elist=smp._elist_dev, scale=_point_scale(smp, itg, arguments), nelems=smp.nlist, ndims=nd, nq=nq, weights=smp._weights_dev, geom=geom, test=tt.struct, trial=tt.struct,
                                    nct=itg.test.ncomp, ncr=itg.test.ncomp, f=itg.L * fac, out=out)
        else:
            u = _argument_dev(arguments, itg.test)
            kernels.assemble_vector(elist=smp._elist_dev, scale=_point_scale(smp, itg, arguments), nelems=smp.nlist, ndims=nd, nq=nq, weights=smp._weights_dev, geom=geom, test=tt.struct, trial=tt.struct,
                                    nct=itg.test.ncomp, ncr=itg.test.ncomp, f=itg.L * fac, u=u, out_scalar=scalar[0])
        return
    # constant integrand (volume-type functional): basis-free launch
    none = kernels.basis(None, None)
    kernels.assemble_vector(elist=smp._elist_dev, scale=_point_scale(smp, itg, arguments), nelems=smp.nlist, ndims=nd, nq=nq, weights=smp._weights_dev, geom=geom, test=none, trial=none, nct=1, ncr=1,
                            f0=float(itg.f0) * fac, out_scalar=scalar[0])


def _deterministic():
    '''Residual vectors are scattered with the owner-side reduction (element kernels store local vectors, nh_scatter_gather sums them per dof in the
    order of the reference's numpy.add.at loop: bit-identical from run to run) unless NUTILS_AMD_ATOMIC_RESIDUALS asks for global atomics.'''
    return not os.environ.get('NUTILS_AMD_ATOMIC_RESIDUALS')


def _scatter(smp, basis, nct, slot=0):
    '''(plan, local array [positions][nct]) of the deterministic scatter for vectors on `basis` assembled over `smp`; `slot` tells apart
    the arrays of several launches whose results are gathered together.  Kept with the sample: a Newton step reuses map and array.'''
    cache = smp.__dict__.setdefault('_scatter_plans', {})
    t = smp.tables(basis)
    plan = cache.get(id(basis))
    if plan is None or plan[0] is not basis:
        plan = cache[id(basis)] = (basis, kernels.ScatterPlan(nelems=smp.nelems, nrows=basis.ndofs, nb=t.nb, dofs=t.dofs, off=t.off, elist=smp._elist_dev, nlist=smp.nlist))
    arrays = smp.__dict__.setdefault('_scatter_local', {})
    key = id(basis), int(nct), slot
    if key not in arrays:
        arrays[key] = device.empty(plan[1].npositions * int(nct), 'float64')
    return plan[1], arrays[key]


def _fusable(itg):
    '''Linear-form term (test dofs exposed) that nh_assemble_terms takes: a form applied to a bound field or a source, times optional
    pointwise factors (coefficient function of x, polynomial of field values).'''
    return (itg.measure is not None and itg.rows and not itg.cols and itg.qform is None
            and (itg.qscalar is None or (itg.qscalar[1].ncomp == 1 and itg.qscalar[2].ncomp == 1 and (itg.geom is None or itg.geom is itg.measure)))
            and (itg.B is not None or itg.L is not None) and (itg.fscale is None or (len(itg.fscale.args) <= 4 and len(itg.fscale.terms) <= 64)))


def _plan_terms(items, blocks, lists, leftover):
    '''items: [(block index, sample, integrand, factor)] on ONE sample and measure; packed into as few term lists as the limits of the entry allow
    (2 blocks / 4 test components, 6 fields / 8 components, 4 polynomials, 32 terms).  Appends TEMPLATES of the keyword dicts of
    kernels.assemble_terms to `lists` -- fields as (tables, argument, ncomp), blocks as (tables, ncomp, block index): `_vector_blocks` fills in the
    device arrays of a Newton step -- and what no list takes to `leftover`.'''
    smp = items[0][1]
    geom = smp.geometry(items[0][2].measure)
    nd, nq = smp.ndims, smp.points.npoints
    pending = list(items)
    while pending:
        fkeys, bkeys, pkeys, terms, rest = [], [], [], [], []
        for it in pending:
            bi, _, itg, fac = it
            need = (([itg.trial] if itg.B is not None else []) + (list(itg.fscale.args) if itg.fscale is not None else [])
                    + ([itg.qscalar[1], itg.qscalar[2]] if itg.qscalar is not None else []))
            newf = []
            for a in need:
                if not any(a.same(k) for k in fkeys + newf):
                    newf.append(a)
            newb = bi not in bkeys
            newp = itg.fscale is not None and not any(itg.fscale is k for k in pkeys)
            if (len(terms) >= 32 or len(fkeys) + len(newf) > 6 or sum(a.ncomp for a in fkeys + newf) > 8 or len(bkeys) + newb > 2
                    or sum(blocks[b][0].ncomp for b in bkeys) + (blocks[bi][0].ncomp if newb else 0) > 4 or len(pkeys) + newp > 4):
                rest.append(it)
                continue
            fkeys += newf
            if newb:
                bkeys.append(bi)
            if newp:
                pkeys.append(itg.fscale)
            fidx = lambda a: next(i for i, k in enumerate(fkeys) if a.same(k))
            t = dict(block=bkeys.index(bi), scale=smp.scale(itg.scale) if itg.scale is not None else None)
            if itg.B is not None:
                t['field'], t['C'] = fidx(itg.trial), numpy.asarray(itg.B, dtype=float) * fac
            else:
                t['f'] = numpy.asarray(itg.L, dtype=float) * fac
            if itg.fscale is not None:
                t['poly'] = next(i for i, k in enumerate(pkeys) if k is itg.fscale)
            if itg.qscalar is not None:  # point factor U_t . B . U_r, evaluated in the kernel
                t['qs'] = (numpy.asarray(itg.qscalar[0], dtype=float)[0, :, 0, :], fidx(itg.qscalar[1]), fidx(itg.qscalar[2]))
            terms.append(t)
        if not terms:  # (a single term beyond the limits: the per-term path takes anything)
            leftover += pending
            return
        fidx = lambda a: next(i for i, k in enumerate(fkeys) if a.same(k))
        polys = []
        for fp in pkeys:
            keys = list(fp.terms)
            polys.append(([(fidx(a), 0) for a in fp.args], [fp.terms[k] for k in keys], keys))
        lists.append(dict(smp=smp, nelems=smp.nlist, ndims=nd, nq=nq, weights=smp._weights_dev, geom=geom, elist=smp._elist_dev,
                          fields=[(smp.tables(a.basis).struct, a, a.ncomp) for a in fkeys],
                          blocks=[(smp.tables(blocks[b][0].basis).struct, blocks[b][0].ncomp, b) for b in bkeys], terms=terms, polys=polys))
        pending = rest


_TERM_PLANS = {}


def _vector_blocks(blocks, arguments, scalar):
    '''Linear forms of one or more output blocks [(test argument, out tensor, terms)]: the structured P1-hex residual terms go through
    nh_p1hex_apply, everything else that shares a sample and a measure through ONE fused element loop -- and the loops of all samples through one
    launch (nh_assemble_terms_multi) --, the remainder term by term.  The split and the term lists depend on the integrals only: they are kept
    per tuple of term lists (the reference caches its compiled callables the same way, solver.py:321-331), a Newton step fills in the arrays.'''
    env = tuple(bool(os.environ.get('NUTILS_AMD_' + name)) for name in ('NO_BATCHED', 'NO_MULTI', 'NO_FAST_PATH', 'ATOMIC_RESIDUALS'))  # (debugging switches: part of the key)
    key = tuple(id(terms) for _, _, terms in blocks) + env
    plan = _TERM_PLANS.get(key)
    if plan is None or not all(a is b[2] for a, b in zip(plan['terms'], blocks)):
        groups, single, lists, leftover = {}, [], [], []
        for bi, (a0, _, terms) in enumerate(blocks):
            for smp, itg, fac in terms:
                if (not env[0] and not (itg.measure is not None and itg.rows and _p1hex_match(smp, itg, fac)) and _fusable(itg)
                        and all(a.ncomp == 1 for a in (itg.fscale.args if itg.fscale is not None else ()))):
                    groups.setdefault((id(smp), id(itg.measure)), []).append((bi, smp, itg, fac))
                else:
                    single.append((bi, smp, itg, fac))
        for items in groups.values():
            _plan_terms(items, blocks, lists, leftover)
        while len(_TERM_PLANS) >= 8:
            _TERM_PLANS.pop(next(iter(_TERM_PLANS)))
        plan = _TERM_PLANS[key] = dict(terms=[b[2] for b in blocks], single=single, lists=lists, leftover=leftover)
    for bi, smp, itg, fac in plan['single']:
        _vector_term(smp, itg, fac, arguments, blocks[bi][1], scalar)
    for bi, smp, itg, fac in plan['leftover']:
        _vector_term(smp, itg, fac, arguments, blocks[bi][1], None)
    ucache = {}

    def dev_u(arg):
        if arg.name not in ucache:
            ucache[arg.name] = _argument_dev(arguments, arg)
        return ucache[arg.name]

    det = _deterministic()
    gathers = {}  # block index -> [(plan, local array)] in list order
    lists = []
    for il, tpl in enumerate(plan['lists']):
        blks = []
        for t, nc, b in tpl['blocks']:
            if det:  # the kernel stores the local vectors of this list; they are summed per dof below, lists in order
                sp, local = _scatter(tpl['smp'], blocks[b][0].basis, nc, slot=(il, b))  # (blocks may share a basis: one array each)
                gathers.setdefault(b, []).append((sp, local))
                blks.append((t, nc, None, local))
            else:
                blks.append((t, nc, blocks[b][1]))
        kw = {k: v for k, v in tpl.items() if k != 'smp'}
        lists.append(dict(kw, fields=[(t, dev_u(a), nc) for t, a, nc in tpl['fields']], blocks=blks))
    if env[1]:  # one launch per sample (tests: the merged launch must give the same result)
        for kw in lists:
            kernels.assemble_terms(**kw)
    else:  # all samples of the residual (volume + the sides of the boundary) in one launch
        kernels.assemble_terms_multi(lists)
    for b, pairs in gathers.items():
        for i in range(0, len(pairs), 8):
            kernels.scatter_gather(pairs[i:i + 8], blocks[b][0].ncomp, blocks[b][1], accumulate=True)


def _bound_terms(f, arguments):
    '''the term list of an Integral with its point variables bound for this call (the list itself if it has none: the term plans are keyed by its identity)'''
    return _bind_pvars(f.terms, arguments)


def _exposed_test(f):
    '''Test argument of a vector-valued integral (all terms expose the same test space), or None.'''
    exposed = [itg for _, itg, _ in f.terms if itg.rows]
    if not exposed or len(exposed) != len(f.terms) or any(itg.cols for itg in exposed):
        return None
    a0 = exposed[0].test
    if not all(itg.test.basis is a0.basis and itg.test.ncomp == a0.ncomp for itg in exposed):
        return None
    return a0


def evaluate_blocks(fs, arguments, flat=False):
    '''The residual blocks of a multi-field system in one pass (solver.py:334-386 evaluates them as one compiled function): the terms
    of all blocks that share a sample go through one element loop, the blocks share one device buffer and one copy to the host.  Integrals
    that are not plain linear forms are evaluated one by one.  flat: the concatenation of the raveled blocks instead of the list.'''
    return start_blocks(fs, arguments, flat)()


def start_blocks(fs, arguments, flat=False):
    '''`evaluate_blocks` in two halves: enqueues the device work and the copy to page-locked host memory on the current stream and returns
    finish() -> result, which waits for that stream.  (All assembly launches of a Newton step share ONE stream -- the library's scratch
    buffers are not per stream, include/nutils_hip.h --; only the copy of the Jacobian entries to the host runs beside them.)'''
    t = device.torch()
    tests = [_exposed_test(f) if isinstance(f, function.Integral) and f.terms else None for f in fs]
    sizes = [a0.basis.ndofs * a0.ncomp if a0 is not None else 0 for a0 in tests]
    offsets = numpy.cumsum([0] + sizes)
    index, host, stream = {}, None, None
    if offsets[-1]:
        blocks = []
        buf = device.zeros(int(offsets[-1]), 'float64')
        for i, (f, a0) in enumerate(zip(fs, tests)):
            if a0 is not None:
                index[i] = len(blocks)
                blocks.append((a0, buf[int(offsets[i]):int(offsets[i + 1])], _bound_terms(f, arguments)))
        _vector_blocks(blocks, arguments, [device.zeros(1, 'float64'), 0.])
        host = t.empty(buf.shape, dtype=buf.dtype, pin_memory=True)
        host.copy_(buf, non_blocking=True)
        stream = t.cuda.current_stream()
    others = {i: evaluate(f, arguments) for i, f in enumerate(fs) if i not in index}

    def finish(buf=buf if offsets[-1] else None):  # (keeps the device buffer until the copy has run)
        if stream is not None:
            stream.synchronize()
        h = host.numpy() if host is not None else None
        if flat and len(index) == len(fs):
            return h
        out = []
        for i, a0 in enumerate(tests):
            if i in index:
                res = h[offsets[i]:offsets[i + 1]]
                out.append(res.reshape(a0.basis.ndofs, a0.ncomp) if a0.ncomp > 1 else res)
            else:
                out.append(others[i])
        return numpy.concatenate([numpy.asarray(o, dtype=float).ravel() for o in out]) if flat else out
    return finish


def _is_ragged(basis):
    b = basis.parent if isinstance(basis, RationalBasis) else basis
    return isinstance(b, PlainBasis) and len(set(numpy.diff(b.offsets).tolist())) > 1


class _SubsetView:
    '''A sample of an element SUBSET as a full sample of its own little topology: the listed elements, in list order (an element listed twice is two
    elements), with bases (same dof numbers), weight tables and geometries restricted to them.  The kernels take element lists with uniform bases only;
    bases with a varying number of functions per element (hierarchical refinements) reach them this way.  The dof spaces are unchanged, so the terms
    add into the same vectors / matrices (one basis object per sample: _MatrixPlan, evaluate).'''

    def __init__(self, smp):
        from . import topology
        self.el = numpy.asarray(smp.elist)
        box = getattr(smp.topo, 'geom', None)
        if isinstance(box, function.BoxGeometry):
            origin, size = numpy.asarray(box.origin)[self.el], numpy.asarray(box.size)[self.el]
        else:
            origin, size = numpy.zeros((len(self.el), smp.ndims)), numpy.ones((len(self.el), smp.ndims))
        self.topo = topology.ElementList(origin, size)
        self.smp = Sample(self.topo, smp.points, elist=None, bnd_axis=smp.bnd_axis)
        self.parent = smp
        self.objs = {}         # id(basis / argument / geometry) -> (object, its restriction)
        self.integrands = {}   # id(integrand) -> (integrand, its restriction): bounded

    def basis(self, b):
        if id(b) not in self.objs:
            if isinstance(b, RationalBasis):
                W = None if b.W is None else self.parent._per_element(b.W, 'weight function table', by_list=False)[self.el]
                dW = None if b.W is None else self.parent._per_element(b.dW, 'weight function table', by_list=False)[self.el]
                r = RationalBasis(self.basis(b.parent), b.weights, W=W, dW=dW)
            else:
                r = PlainBasis([b.get_coefficients(int(e)) for e in self.el], [b.get_dofs(int(e)) for e in self.el], b.ndofs, b.ndims)
            self.objs[id(b)] = (b, r)
        return self.objs[id(b)][1]

    def arg(self, a):
        if a is None:
            return None
        if id(a) not in self.objs:
            self.objs[id(a)] = (a, function.Arg(self.basis(a.basis), a.ncomp, a.name))
        return self.objs[id(a)][1]

    def geom(self, g):
        if g is None:
            return None
        if id(g) not in self.objs:
            if isinstance(g, function.TabulatedGeometry):
                per_list = g.x.shape[0] == len(self.el) and not (g.x.shape[0] == self.parent.nelems and numpy.array_equal(self.el, numpy.arange(self.parent.nelems)))
                r = g if per_list else function.TabulatedGeometry(g.x[self.el], g.jac[self.el])
            elif isinstance(g, function.IsoGeometry):
                r = function.IsoGeometry(self.basis(g.basis), g.verts)
            elif hasattr(g, 'element_boxes'):
                origin, size = g.element_boxes()
                r = function.BoxGeometry(numpy.asarray(origin)[self.el], numpy.asarray(size)[self.el])
            else:
                raise NotImplementedError(f'element subset with a {type(g).__name__}')
            self.objs[id(g)] = (g, r)
        return self.objs[id(g)][1]

    def integrand(self, itg):
        kw = dict(test=self.arg(itg.test), trial=self.arg(itg.trial), geom=self.geom(itg.geom), measure=self.geom(itg.measure))
        if itg.fscale is not None:
            kw['fscale'] = function.FieldPoly([self.arg(a) for a in itg.fscale.args], dict(itg.fscale.terms))
        if itg.qform is not None:
            kw['qform'] = (itg.qform[0], self.arg(itg.qform[1])) + tuple(itg.qform[2:])
        if itg.qscalar is not None:
            kw['qscalar'] = (itg.qscalar[0], self.arg(itg.qscalar[1]), self.arg(itg.qscalar[2]))
        if itg.pvars:
            kw['pvars'] = tuple((self.arg(a), comp, slot) for a, comp, slot in itg.pvars)
        return itg._copy(**kw)


def _restrict_ragged_subsets(terms):
    '''terms with samples of element subsets on ragged bases rewritten through _SubsetView (cached on the sample); the list itself if there are none'''
    out, changed = [], False
    for smp, itg, fac in terms:
        args = [a for a in (itg.test, itg.trial) if a is not None] + (list(itg.fscale.args) if itg.fscale is not None else [])
        args += [itg.qform[1]] if itg.qform is not None else []
        args += [itg.qscalar[1], itg.qscalar[2]] if itg.qscalar is not None else []
        args += [a for a, _, _ in itg.pvars]
        bases = [a.basis for a in args] + [g.basis for g in (itg.geom, itg.measure) if isinstance(g, function.IsoGeometry)]
        # also: a geometry tabulated per LIST position (a side of a NURBS patch) -- spreading it to element rows (Sample._per_element) costs the memory of
        # the whole topology for the sake of a side
        per_list = any(isinstance(g, function.TabulatedGeometry) and smp.elist is not None and g.x.shape[0] == smp.nlist != smp.nelems for g in (itg.geom, itg.measure))
        if (smp.elist is None or not (per_list or any(_is_ragged(b) for b in bases))
                or isinstance(itg.scale, function.PointFunc) and not isinstance(itg.scale, function.PointTable)):
            out.append((smp, itg, fac))
            continue
        view = smp.__dict__.get('_subset_view')
        if view is None:
            view = smp.__dict__['_subset_view'] = _SubsetView(smp)
        key = id(itg)
        if key not in view.integrands:
            while len(view.integrands) >= 256:  # (a script that builds new integrands every step must not pile them up on a long-lived sample)
                view.integrands.pop(next(iter(view.integrands)))
            view.integrands[key] = (itg, view.integrand(itg))
        out.append((view.smp, view.integrands[key][1], fac))
        changed = True
    return out if changed else terms


def _restricted(f):
    '''the Integral with its terms through _restrict_ragged_subsets, cached on the object (the term plans are keyed by the identity of the term lists)'''
    r = f.__dict__.get('_restricted')
    if r is None:
        terms = _restrict_ragged_subsets(f.terms)
        r = f.__dict__['_restricted'] = f if terms is f.terms else function.Integral(terms)
    return r


def evaluate(f, arguments):
    '''Evaluate one Integral / as_csr / as_coo wrapper.'''
    from . import factor as _factor0
    if isinstance(f, function.Integral) and type(f) is function.Integral:
        f = _restricted(f)
    elif isinstance(f, (function._AsCSR, function._AsCOO)) and type(getattr(f, 'integral', None)) is function.Integral:
        g = _restricted(f.integral)
        if g is not f.integral:
            f = type(f)(g)
    if isinstance(f, _Bound):
        return f.eval(arguments)
    if isinstance(f, function._AsCSR) and isinstance(f.integral, _factor0.FactoredMatrix):
        return f.integral.as_csr()
    if type(f) is function.Integral:
        bound = _bind_pvars(f.terms, arguments)
        if bound is not f.terms:
            f = function.Integral(bound)
    elif isinstance(f, (function._AsCSR, function._AsCOO)) and type(f.integral) is function.Integral:
        bound = _bind_pvars(f.integral.terms, arguments)
        if bound is not f.integral.terms:
            f = type(f)(function.Integral(bound))
    if isinstance(f, (function._AsCSR, function._AsCOO)):
        terms = f.integral.terms
        if not terms:
            raise ValueError('empty integral')
        values, rowptr, colidx, ncols = _MatrixPlan(terms).run(arguments)
        values, rowptr, colidx = device.to_host(values), device.to_host(rowptr), device.to_host(colidx)
        if isinstance(f, function._AsCOO):
            rowidx = numpy.repeat(numpy.arange(len(rowptr) - 1, dtype=numpy.int64), numpy.diff(rowptr))
            return values, rowidx, colidx
        return values, rowptr, colidx
    from . import factor as _factor
    if isinstance(f, (_factor.Factored, _factor.FactoredVector)):
        return f.eval(**arguments)
    if isinstance(f, function._AsCSR) and isinstance(f.integral, _factor.FactoredMatrix):
        return f.integral.as_csr()
    if not isinstance(f, function.Integral):
        raise TypeError(f'cannot evaluate {type(f).__name__}')
    if not f.terms:  # every term vanished (derivative with respect to an argument the integral does not depend on): the dof shape is unknown
        raise ValueError('empty integral: all terms vanished, the result shape is not defined (did you differentiate with respect to an absent argument?)')
    kinds = {(itg.rows, itg.cols) for _, itg, _ in f.terms}
    if kinds == {(True, True)}:
        values, rowptr, colidx, ncols = _MatrixPlan(f.terms).run(arguments)
        return _matrix.assemble_csr(device.to_host(values), device.to_host(rowptr), device.to_host(colidx), ncols).export('dense')
    if (True, True) in kinds:
        raise NotImplementedError('mixing matrix and vector terms in one integral')
    exposed = [itg for _, itg, _ in f.terms if itg.rows]
    out = None
    if exposed:
        a0 = exposed[0].test
        if len(exposed) != len(f.terms) or not all(itg.test.basis.ndofs == a0.basis.ndofs and itg.test.ncomp == a0.ncomp for itg in exposed):
            raise NotImplementedError('vector terms with different test spaces')
        if not all(itg.test.basis is a0.basis for itg in exposed):
            # the same dof space through several basis objects (a rational basis tabulated per sample): one pass per object, summed in term order
            groups = {}
            for term in f.terms:
                groups.setdefault(id(term[1].test.basis), []).append(term)
            total = None
            for terms in groups.values():
                part = evaluate(function.Integral(terms), arguments)
                total = part if total is None else total + part
            return total
        out = device.zeros(a0.basis.ndofs * a0.ncomp, 'float64')
    scalar = [device.zeros(1, 'float64'), 0.]  # device accumulator, host-side addend
    if out is not None:
        _vector_blocks([(a0, out, f.terms)], arguments, scalar)
    else:
        for smp, itg, fac in f.terms:
            _vector_term(smp, itg, fac, arguments, out, scalar)
    if out is not None:
        res = device.to_host(out)
        return res.reshape(a0.basis.ndofs, a0.ncomp) if a0.ncomp > 1 else res
    return float(device.to_host(scalar[0])[0]) + scalar[1]
