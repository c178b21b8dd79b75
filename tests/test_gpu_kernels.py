'''GPU parity tests at the C-ABI level: the INPUT tables come straight from the
golden files (real reference output), the result must match the reference's CSR:
rowptr/colidx bit-exact (int64), values within 1e-13 relative (the device sums
element contributions with atomics in a different order than the reference's
stable-sort bincount).'''
import os
import numpy
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-13
SCALAR = ['lap1d_p1_5', 'lap2d_p1_4x4', 'lap2d_p1_4x3_iso', 'lap2d_p2_3x4_iso', 'lap2d_spline2_4x4', 'lap2d_spline2_5x4_iso',
          'lap3d_p1_2', 'lap3d_p1_3', 'lap3d_p1_4', 'lap3d_p1_234', 'lap3d_p1_3_iso', 'lap3d_p1_543_iso', 'lap3d_p2_2_iso',
          'lap3d_spline2_3_iso', 'lap3d_spline3_3',
          'lap1d_spline3_6_per0', 'lap2d_spline2_5x4_per0', 'lap2d_p2_4x3_per1', 'lap3d_p1_345_per02']  # (last row: periodic axes)
ELAST = ['elast2d_p1_3x3', 'elast2d_p2_3x2_iso', 'elast3d_p1_2_iso', 'elast3d_p2_2', 'elast3d_p2_2_iso']


def close(a, b, scale=None):
    a = numpy.asarray(a)
    b = numpy.asarray(b)
    assert a.shape == b.shape
    s = numpy.abs(b).max() if scale is None else scale
    err = numpy.abs(a - b).max()
    assert err <= RTOL * max(s, 1e-300), err / s


class Case:
    '''Device-side setup of one golden case through the raw kernel wrappers.'''

    def __init__(self, g):
        from nutils_amd import device, kernels
        self.g = g
        shape = tuple(g['shape'])
        self.nd = nd = len(shape)
        self.nelems = ne = int(numpy.prod(shape))
        self.nb = nb = len(g['dofs']) // ne
        self.ndofs = int(g['dofs'].max() + 1)
        pts = g['gauss_coords']
        self.nq = nq = len(pts)
        self.points = device.to_dev(pts, 'float64')
        self.weights = device.to_dev(g['gauss_weights'], 'float64')
        coeffs = device.to_dev(g['coeffs'], 'float64')
        T = kernels.tabulate(coeffs, ne * nb, g['coeffs'].shape[1], self.points, nq, nd)
        self.T_host = device.to_host(T).reshape(ne, nb, nq, 1 + nd)
        self.dofs = device.to_dev(g['dofs'], 'int32')
        tab = device.to_dev(numpy.arange(ne), 'int32')
        self.basis = kernels.basis(T, self.dofs, nb=nb, tab=tab)
        if int(g['iso']):
            ngb = 2 ** nd
            gco = device.to_dev(g['gcoeffs'][:ngb], 'float64')
            gT = kernels.tabulate(gco, ngb, g['gcoeffs'].shape[1], self.points, nq, nd)
            self.geom = kernels.geometry_iso(ngb, gT, device.to_dev(g['gdofs'], 'int32'), device.to_dev(g['verts'], 'float64'))
        else:
            origin = numpy.array(list(numpy.ndindex(*shape)), dtype=float)
            self.geom = kernels.geometry_box(device.to_dev(origin, 'float64'), device.to_dev(numpy.ones_like(origin), 'float64'))
        self.pattern = kernels.Pattern(ne, self.ndofs, self.ndofs, self.dofs, self.dofs, nbt=nb, nbr=nb)

    def matrix(self, C, nc=1, mask=None):
        from nutils_amd import device, kernels
        rowptr, colidx = self.pattern.expand(nc, nc, mask)
        values = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=self.nelems, ndims=self.nd, nq=self.nq, weights=self.weights, geom=self.geom, test=self.basis,
                                trial=self.basis, nct=nc, ncr=nc, C=C, mask=mask, pattern=self.pattern, values=values)
        return device.to_host(values), device.to_host(rowptr), device.to_host(colidx)

    def vector(self, nc=1, C=None, f=None, u=None, functional=False, f0=0.):
        from nutils_amd import device, kernels
        out = device.zeros(self.ndofs * nc, 'float64')
        sc = device.zeros(1, 'float64') if functional else None
        ud = None if u is None else device.to_dev(u, 'float64')
        kernels.assemble_vector(nelems=self.nelems, ndims=self.nd, nq=self.nq, weights=self.weights, geom=self.geom, test=self.basis,
                                trial=self.basis, nct=nc, ncr=nc, C=C, f=f, u=ud, out=out, f0=f0, out_scalar=sc)
        return device.to_host(out).reshape(self.ndofs, nc), (float(device.to_host(sc)[0]) if functional else None)


@pytest.mark.parametrize('name', SCALAR)
def test_tabulate(golden, name):
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    N, dN = oa.tabulate(g['coeffs'].reshape(c.nelems, c.nb, -1), g['gauss_coords'])
    close(c.T_host[..., 0], N.transpose(0, 2, 1), 1.)
    close(c.T_host[..., 1:], dN.transpose(0, 2, 1, 3), numpy.abs(dN).max())


@pytest.mark.parametrize('name', SCALAR)
def test_scalar_forms(golden, name):
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    for key, C in (('K', oa.laplace_coefficient(c.nd)), ('M', oa.mass_coefficient(c.nd))):
        v, rp, ci = c.matrix(C)
        assert rp.dtype == ci.dtype == numpy.int64
        assert numpy.array_equal(rp, g[key + '_rowptr']) and numpy.array_equal(ci, g[key + '_colidx'])
        close(v, g[key + '_values'])
    r, _ = c.vector(C=oa.laplace_coefficient(c.nd), u=g['u'])
    close(r[:, 0], g['res_laplace'])
    r, _ = c.vector(C=oa.mass_coefficient(c.nd), u=g['u'])
    close(r[:, 0], g['res_mass'])
    f = numpy.zeros((1, 1 + c.nd))
    f[0, 0] = 1
    r, _ = c.vector(f=f)
    close(r[:, 0], g['load_one'])
    _, vol = c.vector(functional=True, f0=1.)
    close(vol, g['volume'])
    _, en = c.vector(C=oa.laplace_coefficient(c.nd), u=g['u'], functional=True)
    close(en, g['energy'])


@pytest.mark.parametrize('name', SCALAR)
def test_sample_eval(golden, name):
    from nutils_amd import device, kernels
    g = golden(name)
    c = Case(g)
    n = c.nelems * c.nq
    x = device.empty(n * c.nd, 'float64')
    dj = device.empty(n, 'float64')
    U = device.empty(n * (1 + c.nd), 'float64')
    kernels.sample_eval(nelems=c.nelems, ndims=c.nd, nq=c.nq, geom=c.geom, trial=c.basis, ncr=1, points=c.points,
                        u=device.to_dev(g['u'], 'float64'), x=x, detj=dj, U=U)
    U = device.to_host(U).reshape(n, 1 + c.nd)
    close(device.to_host(x).reshape(n, c.nd), g['eval_x'])
    close(device.to_host(dj), g['eval_detJ'])
    close(U[:, 0], g['eval_u'])
    close(U[:, 1:], g['eval_gradu'])


@pytest.mark.parametrize('name', ELAST)
def test_elasticity(golden, name):
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    C = oa.elasticity_coefficient(c.nd, float(g['lam']), float(g['mu']))
    v, rp, ci = c.matrix(C, nc=c.nd, mask=oa.block_mask(C))
    assert numpy.array_equal(rp, g['K_rowptr']) and numpy.array_equal(ci, g['K_colidx'])
    close(v, g['K_values'])
    r, en = c.vector(nc=c.nd, C=C, u=g['u'], functional=True)
    close(r, g['res'])
    close(en, g['energy'])


def test_vector_laplace_block_pruning(golden):
    '''Component-decoupled blocks are absent from the pattern (the simplifier prunes them symbolically).'''
    from oracle import assemble as oa
    g = golden('lap3d_p1_3_iso')
    c = Case(g)
    C = oa.laplace_coefficient(3, 3)
    mask = oa.block_mask(C)
    v, rp, ci = c.matrix(C, nc=3, mask=mask)
    assert len(v) == 3 * len(g['K_values'])
    pts, w = g['gauss_coords'], g['gauss_weights']
    dofs = g['dofs'].reshape(c.nelems, c.nb)
    N, dN = oa.tabulate(g['coeffs'].reshape(c.nelems, c.nb, -1), pts)
    gN, gdN = oa.tabulate(g['gcoeffs'].reshape(c.nelems, 8, -1), pts)
    x, J = oa.geometry_iso(g['verts'], g['gdofs'].reshape(c.nelems, 8), gN, gdN)
    D, det = oa.physical_tables(N, dN, J)
    vo, rpo, cio = oa.assemble_csr(oa.local_matrices(D, D, det * w, C), dofs, dofs, c.ndofs, c.ndofs, mask)
    assert numpy.array_equal(rp, rpo) and numpy.array_equal(ci, cio)
    close(v, vo)


@pytest.mark.parametrize('ndim,nnz_t,nnz_h', [(1, 60, 70), (2, 3012, 3424)])
def test_hierarchical_ragged(golden, ndim, nnz_t, nnz_h):
    '''Ragged nbasis-per-element (th-/h-spline on a locally refined mesh); known nnz from
    /root/reference/tests/test_basis.py:87-116.'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden(f'hier_spline2_{ndim}d')
    pts = device.to_dev(g['gauss_coords'], 'float64')
    w = device.to_dev(g['gauss_weights'], 'float64')
    nq = len(g['gauss_weights'])
    geom = kernels.geometry_box(device.to_dev(g['elem_origin'], 'float64'), device.to_dev(g['elem_size'], 'float64'))
    for key, nnz in (('t', nnz_t), ('h', nnz_h)):
        off_h = g[key + '_dof_offsets']
        ne = len(off_h) - 1
        ndofs = int(g[key + '_ndofs'])
        off = device.to_dev(off_h, 'int64')
        dofs = device.to_dev(g[key + '_dofs'], 'int32')
        T = kernels.tabulate(device.to_dev(g[key + '_coeffs'], 'float64'), len(g[key + '_dofs']), g[key + '_coeffs'].shape[1], pts, nq, ndim)
        b = kernels.basis(T, dofs, nb=0, off=off)
        pat = kernels.Pattern(ne, ndofs, ndofs, dofs, dofs, toff=off, roff=off)
        rowptr, colidx = pat.expand()
        values = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=ne, ndims=ndim, nq=nq, weights=w, geom=geom, test=b, trial=b, nct=1, ncr=1,
                                C=oa.laplace_coefficient(ndim), mask=None, pattern=pat, values=values)
        assert colidx.numel() == nnz
        assert numpy.array_equal(device.to_host(rowptr), g[key + 'K_rowptr']) and numpy.array_equal(device.to_host(colidx), g[key + 'K_colidx'])
        close(device.to_host(values), g[key + 'K_values'])


# ---- write-once structured fast path (nh_p1hex_*) -------------------------------------------------

P1HEX = ['lap3d_p1_2', 'lap3d_p1_3', 'lap3d_p1_4', 'lap3d_p1_234', 'lap3d_p1_3_iso', 'lap3d_p1_543_iso']


@pytest.mark.parametrize('name', P1HEX)
def test_p1hex_fast_path_golden(golden, name):
    from nutils_amd import device, kernels, points
    g = golden(name)
    shape = tuple(int(n) for n in g['shape'])
    rowptr, colidx = kernels.p1hex_pattern(shape)
    assert numpy.array_equal(device.to_host(rowptr), g['K_rowptr']) and numpy.array_equal(device.to_host(colidx), g['K_colidx'])
    values = device.empty(colidx.numel(), 'float64')
    values.fill_(float('nan'))  # write-once: every entry must be stored exactly once, no zero-fill needed
    x1, w1 = points.gauss1(2)
    verts = device.to_dev(g['verts'], 'float64') if int(g['iso']) else None
    kernels.p1hex_laplace(shape=shape, values=values, gauss_x=x1, gauss_w=w1, verts=verts)
    close(device.to_host(values), g['K_values'])


@pytest.mark.parametrize('shape,iso', [((17, 9, 23), True), ((7, 7, 7), False), ((1, 1, 1), True), ((1, 20, 3), True), ((30, 2, 1), False),
                                       ((40, 33, 70), True), ((3, 100, 100), True), ((100, 16, 15), False), ((65, 1, 130), True)])
def test_p1hex_fast_vs_generic(shape, iso, monkeypatch):
    '''Tile-boundary / ragged-edge coverage: sizes that are not multiples of the 15 x 15 dof column tile or of the two-layer step,
    several tiles per axis, meshes smaller than one tile; compared with the generic kernel
    (itself pinned to the golden vectors above).'''
    from nutils_amd import mesh, function, device, kernels, points
    rng = numpy.random.default_rng(3)
    domain, geom = mesh.rectilinear(list(shape))
    basis = domain.basis('std', degree=1)
    verts = None
    if iso:
        verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (len(basis), 3))
        geom = basis @ verts
    K = domain.integral(function.outer(function.grad(basis, geom)).sum(-1) * function.J(geom), degree=2)
    monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')  # reference values from the generic kernel
    vref, rp, ci = function.eval(function.as_csr(K))
    monkeypatch.delenv('NUTILS_AMD_NO_FAST_PATH')
    vapi, rpa, cia = function.eval(function.as_csr(2.5 * K))  # the same integral through the API takes the fast path on its own
    assert numpy.array_equal(rpa, rp) and numpy.array_equal(cia, ci)
    close(vapi, 2.5 * vref)
    rowptr, colidx = kernels.p1hex_pattern(shape)
    assert numpy.array_equal(device.to_host(rowptr), rp) and numpy.array_equal(device.to_host(colidx), ci)
    values = device.empty(colidx.numel(), 'float64')
    values.fill_(float('nan'))
    x1, w1 = points.gauss1(2)
    kernels.p1hex_laplace(shape=shape, values=values, gauss_x=x1, gauss_w=w1, verts=None if verts is None else device.to_dev(verts, 'float64'))
    close(device.to_host(values), vref)
    # row sub-range of the pattern (multi-GPU row blocks)
    nplane = (shape[1] + 1) * (shape[2] + 1)
    r0, r1 = nplane, min(3, shape[0] + 1) * nplane
    rps, cis = kernels.p1hex_pattern(shape, r0, r1)
    assert numpy.array_equal(device.to_host(rps), rp[r0:r1 + 1] - rp[r0]) and numpy.array_equal(device.to_host(cis), ci[rp[r0]:rp[r1]])


def test_error_paths_on_device():
    '''Invalid arguments come back as status codes + nh_last_error -> NutilsHipError, never a crash or a silent result.'''
    import ctypes
    from nutils_amd import _lib, device, kernels
    device.require_gpu()
    pts = device.to_dev(numpy.zeros((2, 3)), 'float64')
    co = device.to_dev(numpy.zeros((4, 7)), 'float64')
    with pytest.raises(_lib.NutilsHipError, match='not a valid coefficient count'):
        kernels.tabulate(co, 4, 7, pts, 2, 3)
    with pytest.raises(_lib.NutilsHipError, match='ndims must be 1..3'):
        kernels.tabulate(co, 4, 7, pts, 2, 4)
    d = device.to_dev(numpy.arange(8), 'int32')
    with pytest.raises(_lib.NutilsHipError, match='give either nbt or toff_dev'):
        kernels.Pattern(1, 8, 8, d, d)
    pat = kernels.Pattern(1, 8, 8, d, d, nbt=8, nbr=8)
    with pytest.raises(_lib.NutilsHipError, match='component counts'):
        pat.expand(9, 1)
    with pytest.raises(ValueError, match='coefficient tensor has shape'):
        kernels.assemble_matrix(nelems=1, ndims=3, nq=2, weights=pts, geom=kernels.geometry_box(pts, pts), test=kernels.basis(co, d, nb=8),
                                trial=kernels.basis(co, d, nb=8), nct=1, ncr=1, C=numpy.zeros((1, 3, 1, 3)), mask=None, pattern=pat, values=co)
    g = _lib.Geometry(7, 0, None, None, None, None, None, None, None, -1)
    with pytest.raises(_lib.NutilsHipError, match='unknown geometry kind'):
        kernels.sample_eval(nelems=1, ndims=3, nq=2, geom=g, points=pts, detj=co)
    with pytest.raises(_lib.NutilsHipError, match='layer range'):
        kernels.p1hex_laplace(shape=(2, 2, 2), values=co, gauss_x=[.2, .8], gauss_w=[.5, .5], layers=(0, 3))


@pytest.mark.parametrize('iso', [True, False])
def test_p1hex_fast_path_with_coefficients(iso, monkeypatch):
    '''Variable and field-dependent diffusivity through the front end: kappa(x) grad.grad + (1 + u^2) grad.grad + 0.5 grad.grad on
    the trilinear basis takes the write-once kernel with a per-Gauss-point coefficient array (qscale_dev); values must equal the
    generic kernel's, the Jacobian-type re-assembly must follow a change of the field.'''
    from nutils_amd import mesh, function
    shape = (9, 20, 17)
    rng = numpy.random.default_rng(5)
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1.8, shape[0] + 1), numpy.linspace(0, 2., shape[1] + 1), numpy.linspace(-1, 0.7, shape[2] + 1)])
    basis = domain.basis('std', degree=1)
    if iso:
        verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, 3) * .1 + rng.uniform(-.02, .02, (len(basis), 3))
        geom = basis @ verts
    kappa = function.PointFunc(lambda x: 1.5 + numpy.sin(3 * x[:, 0]) * x[:, 1] + x[:, 2] ** 2, geom)
    u = function.value(domain.field('u', btype='std', degree=1))
    gg = function.outer(function.grad(basis, geom)).sum(-1)
    dV = function.J(geom)
    K = domain.integral(kappa * gg * dV, degree=2) + domain.integral((1 + u ** 2) * gg * dV, degree=2) + domain.integral(.5 * gg * dV, degree=2)
    from nutils_amd import kernels
    calls = []
    orig = kernels.p1hex_laplace
    monkeypatch.setattr(kernels, 'p1hex_laplace', lambda **kw: (calls.append(kw.get('qscale') is not None), orig(**kw))[1])
    results = {}
    for mode in ('generic', 'fast'):
        if mode == 'generic':
            monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')
        else:
            monkeypatch.delenv('NUTILS_AMD_NO_FAST_PATH')
        for tag, scale in (('a', 1.), ('b', -2.)):
            args = {'u': scale * numpy.cos(numpy.arange(len(basis)) * .37)}
            results[mode, tag] = function.eval(function.as_csr(K), arguments=args)
    for tag in 'ab':
        vg, rpg, cig = results['generic', tag]
        vf, rpf, cif = results['fast', tag]
        assert numpy.array_equal(rpg, rpf) and numpy.array_equal(cig, cif)
        close(vf, vg)
    assert numpy.abs(results['fast', 'a'][0] - results['fast', 'b'][0]).max() > 1e-3  # the field really enters
    assert calls == [True, True]  # the write-once kernel ran exactly for the two 'fast' evaluations, with a coefficient array


@pytest.mark.parametrize('shape,iso', [((9, 20, 17), True), ((33, 16, 40), False), ((2, 1, 3), True), ((40, 33, 70), True)])
def test_p1hex_residual_fast_path(shape, iso, monkeypatch):
    '''Residual of the headline form, r_m = int kappa grad(phi_m) . grad(u): the front end applies the element matrices on the
    fly through nh_p1hex_apply (no matrix, no global atomics); must equal the generic vector kernel and K u.'''
    from nutils_amd import mesh, function, kernels
    rng = numpy.random.default_rng(7)
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1. + .1 * i, n + 1) for i, n in enumerate(shape)])
    basis = domain.basis('std', degree=1)
    if iso:
        verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, 3) * .1 + rng.uniform(-.02, .02, (len(basis), 3))
        geom = basis @ verts
    u = domain.field('u', btype='std', degree=1)
    kappa = function.PointFunc(lambda x: 1.5 + x[:, 0] * x[:, 1] - x[:, 2], geom)
    dV = function.J(geom)
    gu = (function.grad(basis, geom) * function.grad(u, geom)).sum(-1)
    res = domain.integral(kappa * gu * dV, degree=2) + domain.integral((1 + function.value(u) ** 2) * gu * dV, degree=2) + domain.integral(.5 * gu * dV, degree=2)
    args = {'u': numpy.cos(numpy.arange(len(basis)) * .37)}
    calls = []
    orig = kernels.p1hex_apply
    monkeypatch.setattr(kernels, 'p1hex_apply', lambda **kw: (calls.append(1), orig(**kw))[1])
    monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')
    rg = function.eval(res, arguments=args)
    assert not calls
    monkeypatch.delenv('NUTILS_AMD_NO_FAST_PATH')
    rf = function.eval(res, arguments=args)
    assert len(calls) == 3
    close(rf, rg)
    gg = function.outer(function.grad(basis, geom)).sum(-1)
    K = domain.integral(kappa * gg * dV, degree=2) + domain.integral((1 + function.value(u) ** 2) * gg * dV, degree=2) + domain.integral(.5 * gg * dV, degree=2)
    v, rp, ci = function.eval(function.as_csr(K), arguments=args)
    import scipy.sparse
    close(scipy.sparse.csr_matrix((v, ci, rp), (len(basis),) * 2) @ args['u'], rg)


@pytest.mark.parametrize('iso,coeff', [(True, True), (False, True), (False, False), (True, False)])
def test_p1hex_fast_path_mass_and_stiffness(iso, coeff, monkeypatch):
    '''Reaction-diffusion type forms on the trilinear basis, (kappa grad.grad + mu phi phi): matrix and residual through the front end
    take the write-once kernels (mass instantiation; uniform meshes with constant coefficients: hoisted unit matrix incl. mass);
    must equal the generic kernels.'''
    from nutils_amd import mesh, function, kernels
    shape = (9, 20, 17)
    rng = numpy.random.default_rng(8)
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1.8, shape[0] + 1), numpy.linspace(0, 2., shape[1] + 1), numpy.linspace(-1, 0.7, shape[2] + 1)])
    basis = domain.basis('std', degree=1)
    if iso:
        verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, 3) * .1 + rng.uniform(-.02, .02, (len(basis), 3))
        geom = basis @ verts
    u = domain.field('u', btype='std', degree=1)
    dV = function.J(geom)
    gg = function.outer(function.grad(basis, geom)).sum(-1)
    gu = (function.grad(basis, geom) * function.grad(u, geom)).sum(-1)
    K = domain.integral(.7 * gg * dV, degree=2) + domain.integral(2. * function.outer(basis) * dV, degree=2)
    r = domain.integral(.7 * gu * dV, degree=2) + domain.integral(2. * basis * u * dV, degree=2)
    if coeff:
        rho = function.PointFunc(lambda x: 2. + numpy.cos(x[:, 0]) * x[:, 2], geom)
        K = K + domain.integral(function.outer(basis) * rho * dV, degree=2) + domain.integral((1 + function.value(u) ** 2) * gg * dV, degree=2)
        r = r + domain.integral(basis * u * rho * dV, degree=2) + domain.integral((1 + function.value(u) ** 2) * gu * dV, degree=2)
    args = {'u': numpy.cos(numpy.arange(len(basis)) * .37)}
    calls = {'laplace': 0, 'apply': 0}
    for name in ('p1hex_laplace', 'p1hex_apply'):
        orig = getattr(kernels, name)
        monkeypatch.setattr(kernels, name, lambda _o=orig, _n=name.split('_')[1], **kw: (calls.__setitem__(_n, calls[_n] + 1), _o(**kw))[1])
    monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')
    vg, rpg, cig = function.eval(function.as_csr(K), arguments=args)
    rg = function.eval(r, arguments=args)
    assert calls == {'laplace': 0, 'apply': 0}
    monkeypatch.delenv('NUTILS_AMD_NO_FAST_PATH')
    vf, rpf, cif = function.eval(function.as_csr(K), arguments=args)
    rf = function.eval(r, arguments=args)
    assert calls == {'laplace': 1, 'apply': 4 if coeff else 2}
    assert numpy.array_equal(rpg, rpf) and numpy.array_equal(cig, cif)
    close(vf, vg)
    close(rf, rg)
    import scipy.sparse
    close(scipy.sparse.csr_matrix((vf, cif, rpf), (len(basis),) * 2) @ args['u'], rg)


def test_p1hex_fast_path_mixed_terms(monkeypatch):
    '''Diffusion + advection + reaction on the trilinear basis: the symmetric part takes the write-once kernel, the advection term (not
    of that shape) is added by the generic kernel into the same array; result equal to the all-generic assembly.'''
    from nutils_amd import mesh, function, kernels
    shape = (7, 18, 20)
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1, n + 1) for n in shape])
    basis = domain.basis('std', degree=1)
    dV = function.J(geom)
    gg = function.outer(function.grad(basis, geom)).sum(-1)
    adv = function.outer(basis, (function.grad(basis, geom) * numpy.array([1., -2., .5])).sum(-1))
    K = domain.integral(.3 * gg * dV, degree=2) + domain.integral(adv * dV, degree=2) + domain.integral(4. * function.outer(basis) * dV, degree=2)
    calls = []
    orig = kernels.p1hex_laplace
    monkeypatch.setattr(kernels, 'p1hex_laplace', lambda **kw: (calls.append(1), orig(**kw))[1])
    monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')
    vg, rpg, cig = function.eval(function.as_csr(K))
    monkeypatch.delenv('NUTILS_AMD_NO_FAST_PATH')
    vf, rpf, cif = function.eval(function.as_csr(K))
    assert calls == [1]
    assert numpy.array_equal(rpg, rpf) and numpy.array_equal(cig, cif)
    close(vf, vg)
    import scipy.sparse
    A = scipy.sparse.csr_matrix((vf, cif, rpf), (len(basis),) * 2)
    assert abs(A - A.T).max() > 1e-3  # the advection term is there


@pytest.mark.parametrize('case', ['2d_p3', '3d_p2_vector', '2d_p3_odd'])
def test_first_touch_coloured_assembly(case, monkeypatch):
    '''NH_MATRIX_FIRST_TOUCH (no zero-fill, entries untouched by earlier colours are stored): bit-identical to the zero-filled coloured
    assembly, and EVERY value is written -- the value array is handed over full of NaN.'''
    from nutils_amd import mesh, function, device, sample
    monkeypatch.setenv('NUTILS_AMD_NO_GATHER', '1')  # (16-function scalar blocks take the owner-side reduction from the second assembly on)
    if case == '3d_p2_vector':
        monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')  # this test is about the coloured generic path (the write-once kernel nh_p2hex_matrix would take the form,
        monkeypatch.setenv('NUTILS_AMD_NO_FUSED', '1')      # and the owner kernel for vector-valued 27-function blocks after it)
        shape = [16, 16, 17]
        domain, geom = mesh.rectilinear(shape)
        gb = domain.basis('std', degree=1)
        rng = numpy.random.default_rng(1)
        verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (len(gb), 3))
        geom = gb @ verts
        u = domain.field('u', btype='std', degree=2, shape=[3])
        v = domain.field('v', btype='std', degree=2, shape=[3])
        eps = lambda w: function.symgrad(w, geom)
        sigma = function.div(u, geom) * function.eye(3) + 1.3 * eps(u)
        res = domain.integral(function.inner(eps(v), sigma) * function.J(geom), degree=4)
    else:
        shape = [64, 64] if case == '2d_p3' else [65, 67]
        domain, geom = mesh.rectilinear([numpy.linspace(0, 1, shape[0] + 1), numpy.linspace(0, 2, shape[1] + 1)])
        u = domain.field('u', btype='std', degree=3)
        v = domain.field('v', btype='std', degree=3)
        res = domain.integral(((function.grad(v, geom) * function.grad(u, geom)).sum(-1) + v * u) * function.J(geom), degree=6)
    jac = function.derivative(function.derivative(res, 'v'), 'u')
    plan = sample._MatrixPlan(jac.terms)
    assert plan._first_touch(plan.terms[0]) == (tuple(shape), 3 if case == '3d_p2_vector' else 4)
    monkeypatch.setenv('NUTILS_AMD_NO_FIRST_TOUCH', '1')
    ref = [device.to_host(t) for t in sample._MatrixPlan(jac.terms).run()[:3]]
    monkeypatch.delenv('NUTILS_AMD_NO_FIRST_TOUCH')
    real_empty = device.empty
    monkeypatch.setattr(device, 'empty', lambda n, dtype: real_empty(n, dtype).fill_(float('nan')) if dtype == 'float64' else real_empty(n, dtype))
    got = [device.to_host(t) for t in plan.run()[:3]]
    assert numpy.array_equal(got[1], ref[1]) and numpy.array_equal(got[2], ref[2])
    assert not numpy.isnan(got[0]).any()
    assert numpy.array_equal(got[0], ref[0])


def test_p1hex_skewed_vs_marching_kernel(monkeypatch):
    '''The four matrix kernels -- role-skewed halo tiles (skew), two arithmetic waves + one memory wave per SIMD (tri), exact tiles with inter-workgroup
    face exchange (tiles), marching halo tiles (march) -- on random shapes, layer / plane ranges and workgroup limits: the same set of entries is written (the
    value array is handed over full of NaN), values agree to rounding.  Shapes reach several 16 x 16 tiles per axis, partial last tiles,
    single-line tiles and more units than workgroups (several runs per workgroup).'''
    from nutils_amd import kernels, device, points
    x1, w1 = points.gauss1(2)
    rng = numpy.random.default_rng(11)
    for it in range(20):
        shape = tuple(int(x) for x in rng.integers(2, 45, 3))
        if it == 17:
            shape = (5, 70, 33)
        if it == 18:
            shape = (40, 16, 17)
        if it == 19:
            shape = (3, 129, 50)
        n0 = shape[0]
        l0 = int(rng.integers(0, n0)); l1 = int(rng.integers(l0 + 1, n0 + 1))
        p0 = int(rng.integers(0, n0 + 1)); p1 = int(rng.integers(p0 + 1, n0 + 2))
        if it % 4 == 0 or it >= 17:
            l0, l1, p0, p1 = 0, n0, 0, n0 + 1
        nv = (shape[0] + 1) * (shape[1] + 1) * (shape[2] + 1)
        g = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, 3) + numpy.random.default_rng(it).uniform(-.2, .2, (nv, 3))
        verts = device.to_dev(g, 'float64')
        rowptr, colidx = kernels.p1hex_pattern(shape)
        got = []
        for kern in ('tiles', 'tri', 'skew', 'march'):
            monkeypatch.setenv('NH_P1HEX_KERNEL', kern)
            values = device.empty(colidx.numel(), 'float64')
            values.fill_(float('nan'))
            kernels.p1hex_laplace(shape=shape, values=values, gauss_x=x1, gauss_w=w1, verts=verts, layers=(l0, l1), planes=(p0, p1), max_workgroups=(it % 3) * 100 if it != 19 else 7)
            got.append(device.to_host(values))
        written = ~numpy.isnan(got[3])
        assert written.any()
        for k in (0, 1, 2):
            assert numpy.array_equal(~numpy.isnan(got[k]), written), (shape, k)
            assert numpy.abs(got[k][written] - got[3][written]).max() <= 1e-14 * numpy.abs(got[3][written]).max(), (shape, k)


def test_p1hex_tiles_repeated_launches_and_layout_changes(monkeypatch):
    '''The exact-tile kernel (NH_P1HEX_KERNEL=tiles) keeps flags across launches (epoch) and re-lays its exchange scratch when the mesh changes:
    alternate two meshes, several launches each, every result equal to the first of its mesh and free of NaN.'''
    from nutils_amd import kernels, device, points
    monkeypatch.setenv('NH_P1HEX_KERNEL', 'tiles')
    x1, w1 = points.gauss1(2)
    first = {}
    for it in range(8):
        shape = (20, 40, 35) if it % 3 else (9, 33, 18)
        nv = (shape[0] + 1) * (shape[1] + 1) * (shape[2] + 1)
        g = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, 3) + numpy.random.default_rng(5).uniform(-.2, .2, (nv, 3))
        rowptr, colidx = kernels.p1hex_pattern(shape)
        values = device.empty(colidx.numel(), 'float64')
        values.fill_(float('nan'))
        kernels.p1hex_laplace(shape=shape, values=values, gauss_x=x1, gauss_w=w1, verts=device.to_dev(g, 'float64'))
        v = device.to_host(values)
        assert not numpy.isnan(v).any()
        ref = first.setdefault(shape, v)
        assert numpy.abs(v - ref).max() <= 1e-14 * numpy.abs(ref).max()


# ---- fused linear forms: several elements per workgroup, all terms in one loop (nh_assemble_terms) ---------------------------------

@pytest.mark.parametrize('name', SCALAR)
def test_terms_scalar_golden(golden, name):
    '''Laplace residual + mass residual + load vector of the reference in ONE launch = the sum of the three golden vectors.'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    u = device.to_dev(g['u'], 'float64')
    out = device.zeros(c.ndofs, 'float64')
    f = numpy.zeros((1, 1 + c.nd))
    f[0, 0] = 1
    kernels.assemble_terms(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, fields=[(c.basis, u, 1)], blocks=[(c.basis, 1, out)],
                           terms=[dict(block=0, field=0, C=oa.laplace_coefficient(c.nd)), dict(block=0, field=0, C=2.5 * oa.mass_coefficient(c.nd)), dict(block=0, f=f)])
    ref = g['res_laplace'] + 2.5 * g['res_mass'] + g['load_one']
    close(device.to_host(out), ref, numpy.abs(g['res_laplace']).max() + numpy.abs(ref).max())


@pytest.mark.parametrize('name', ELAST)
def test_terms_elasticity_golden(golden, name):
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    C = oa.elasticity_coefficient(c.nd, float(g['lam']), float(g['mu']))
    u = device.to_dev(g['u'], 'float64')
    out = device.zeros(c.ndofs * c.nd, 'float64')
    kernels.assemble_terms(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, fields=[(c.basis, u, c.nd)], blocks=[(c.basis, c.nd, out)],
                           terms=[dict(block=0, field=0, C=C)])
    close(device.to_host(out).reshape(c.ndofs, c.nd), g['res'])


@pytest.mark.parametrize('name', ['lap2d_spline2_5x4_iso', 'lap3d_p1_3_iso'])
def test_terms_two_blocks_polynomial_factor(golden, name):
    '''Two output blocks, three fields, a pointwise polynomial of two of them and a scale array: equal to one nh_assemble_vector per term
    with the factors evaluated by nh_sample_eval + nh_pointwise_poly (the path this entry replaces).'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    rng = numpy.random.default_rng(5)
    nd, S, n = c.nd, 1 + c.nd, c.nelems * c.nq
    us = [device.to_dev(rng.normal(size=c.ndofs), 'float64') for _ in range(3)]
    sc = device.to_dev(rng.uniform(.5, 1.5, n), 'float64')
    f = rng.normal(size=(1, S))
    Cs = [rng.normal(size=(1, S, 1, S)) for _ in range(3)]
    coeffs, powers = [.25, -1., .5, 2.], [[2, 0], [0, 1], [1, 3], [0, 0]]
    out = [device.zeros(c.ndofs, 'float64') for _ in range(2)]
    kernels.assemble_terms(nelems=c.nelems, ndims=nd, nq=c.nq, weights=c.weights, geom=c.geom, fields=[(c.basis, u, 1) for u in us],
                           blocks=[(c.basis, 1, o) for o in out], polys=[([(0, 0), (2, 0)], coeffs, powers)],
                           terms=[dict(block=0, field=0, C=Cs[0]), dict(block=0, field=1, C=Cs[1], poly=0), dict(block=1, field=2, C=Cs[2], scale=sc),
                                  dict(block=1, f=f, poly=0, scale=sc), dict(block=0, f=2 * f)])
    # the same through the per-term path
    U = []
    for u in us:
        Ud = device.empty(n * S, 'float64')
        kernels.sample_eval(nelems=c.nelems, ndims=nd, nq=c.nq, geom=c.geom, trial=c.basis, ncr=1, points=c.points, u=u, U=Ud)
        U.append(Ud)
    pv = kernels.pointwise_poly([U[0], U[2]], [S, S], coeffs, powers, n)
    ref = [device.zeros(c.ndofs, 'float64') for _ in range(2)]
    kw = dict(nelems=c.nelems, ndims=nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=1, ncr=1)
    kernels.assemble_vector(C=Cs[0], u=us[0], out=ref[0], **kw)
    kernels.assemble_vector(C=Cs[1], u=us[1], out=ref[0], scale=pv, **kw)
    kernels.assemble_vector(C=Cs[2], u=us[2], out=ref[1], scale=sc, **kw)
    kernels.assemble_vector(f=f, out=ref[1], scale=pv * sc, **kw)
    kernels.assemble_vector(f=2 * f, out=ref[0], **kw)
    for o, r in zip(out, ref):
        close(device.to_host(o), device.to_host(r))


@pytest.mark.parametrize('name', ['lap2d_spline2_5x4_iso', 'lap3d_p1_3_iso'])
def test_terms_multi_equals_separate_launches(golden, name):
    '''nh_assemble_terms_multi: three term lists in one launch (all elements with a polynomial factor; an element subset through `elist` with 24 terms,
    whose table does not fit the parameter block and travels beside it; an empty list) = the same lists through nh_assemble_terms, one launch each.'''
    from nutils_amd import device, kernels
    g = golden(name)
    c = Case(g)
    rng = numpy.random.default_rng(11)
    nd, S = c.nd, 1 + c.nd
    us = [device.to_dev(rng.normal(size=c.ndofs), 'float64') for _ in range(2)]
    sub = numpy.sort(rng.choice(c.nelems, size=max(2, c.nelems // 3), replace=False)).astype(numpy.int32)
    elist = device.to_dev(sub, 'int32')
    sc = device.to_dev(rng.uniform(.5, 1.5, len(sub) * c.nq), 'float64')

    def lists(outs):
        common = dict(ndims=nd, nq=c.nq, weights=c.weights, geom=c.geom, fields=[(c.basis, u, 1) for u in us])
        a = dict(nelems=c.nelems, blocks=[(c.basis, 1, o) for o in outs], polys=[([(0, 0), (1, 0)], [.5, -1., 2.], [[2, 0], [1, 1], [0, 0]])],
                 terms=[dict(block=0, field=0, C=rng0.normal(size=(1, S, 1, S)), poly=0), dict(block=1, field=1, C=rng0.normal(size=(1, S, 1, S))),
                        dict(block=1, f=rng0.normal(size=(1, S)), poly=0)], **common)
        b = dict(nelems=len(sub), elist=elist, blocks=[(c.basis, 1, outs[1])],
                 terms=[dict(block=0, field=k % 2, C=rng0.normal(size=(1, S, 1, S)), f=rng0.normal(size=(1, S)), scale=sc if k % 3 == 0 else None) for k in range(24)], **common)
        e = dict(nelems=0, blocks=[(c.basis, 1, outs[0])], terms=[dict(block=0, f=numpy.ones((1, S)))], **common)
        return [a, b, e]

    out = [device.zeros(c.ndofs, 'float64') for _ in range(2)]
    ref = [device.zeros(c.ndofs, 'float64') for _ in range(2)]
    rng0 = numpy.random.default_rng(3)
    for _ in range(3):  # (the ring of parameter buffers is reused)
        kernels.assemble_terms_multi(lists(out))
    rng0 = numpy.random.default_rng(3)
    for _ in range(3):
        for kw in lists(ref):
            kernels.assemble_terms(**kw)
    for o, r in zip(out, ref):
        assert numpy.abs(device.to_host(r)).max() > 0
        close(device.to_host(o), device.to_host(r))


def test_terms_ragged_and_errors(golden):
    '''Ragged bases (functions per element from the offsets) and the argument checks of nh_assemble_terms.'''
    from nutils_amd import device, kernels, _lib
    from oracle import assemble as oa
    g = golden('hier_spline2_2d')
    pts = device.to_dev(g['gauss_coords'], 'float64')
    w = device.to_dev(g['gauss_weights'], 'float64')
    nq = len(g['gauss_weights'])
    geom = kernels.geometry_box(device.to_dev(g['elem_origin'], 'float64'), device.to_dev(g['elem_size'], 'float64'))
    off_h = g['t_dof_offsets']
    ne, ndofs = len(off_h) - 1, int(g['t_ndofs'])
    dofs = device.to_dev(g['t_dofs'], 'int32')
    T = kernels.tabulate(device.to_dev(g['t_coeffs'], 'float64'), len(g['t_dofs']), g['t_coeffs'].shape[1], pts, nq, 2)
    b = kernels.basis(T, dofs, nb=0, off=device.to_dev(off_h, 'int64'))
    u = device.to_dev(numpy.random.default_rng(1).normal(size=ndofs), 'float64')
    out, ref = device.zeros(ndofs, 'float64'), device.zeros(ndofs, 'float64')
    C = oa.laplace_coefficient(2) + oa.mass_coefficient(2)
    kernels.assemble_terms(nelems=ne, ndims=2, nq=nq, weights=w, geom=geom, fields=[(b, u, 1)], blocks=[(b, 1, out)], terms=[dict(block=0, field=0, C=C)])
    kernels.assemble_vector(nelems=ne, ndims=2, nq=nq, weights=w, geom=geom, test=b, trial=b, nct=1, ncr=1, C=C, u=u, out=ref)
    close(device.to_host(out), device.to_host(ref))
    with pytest.raises(_lib.NutilsHipError, match='missing block'):
        kernels.assemble_terms(nelems=ne, ndims=2, nq=nq, weights=w, geom=geom, fields=[(b, u, 1)], blocks=[(b, 1, out)], terms=[dict(block=0, field=1, f=numpy.ones((1, 3)))])
    with pytest.raises(_lib.NutilsHipError, match='neither a form nor a source'):
        kernels.assemble_terms(nelems=ne, ndims=2, nq=nq, weights=w, geom=geom, fields=[(b, u, 1)], blocks=[(b, 1, out)], terms=[dict(block=0, field=0)])


# ---- fused bilinear forms (nh_assemble_matrix_terms) ----------------------------------------------------------------------------

@pytest.mark.parametrize('name', SCALAR)
def test_matrix_terms_scalar_golden(golden, name):
    '''Stiffness + 2.5 x mass of the reference in ONE launch, several elements per workgroup.'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    rowptr, colidx = c.pattern.expand(1, 1, None)
    values = device.zeros(colidx.numel(), 'float64')
    kernels.assemble_matrix_terms(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=1, ncr=1, mask=None,
                                  pattern=c.pattern, values=values, terms=[dict(C=oa.laplace_coefficient(c.nd)), dict(C=2.5 * oa.mass_coefficient(c.nd))])
    assert numpy.array_equal(device.to_host(rowptr), g['K_rowptr']) and numpy.array_equal(device.to_host(colidx), g['K_colidx'])
    assert numpy.array_equal(g['K_colidx'], g['M_colidx'])
    close(device.to_host(values), g['K_values'] + 2.5 * g['M_values'], numpy.abs(g['K_values']).max())


@pytest.mark.parametrize('name', ELAST)
def test_matrix_terms_elasticity_golden(golden, name):
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    C = oa.elasticity_coefficient(c.nd, float(g['lam']), float(g['mu']))
    mask = oa.block_mask(C)
    rowptr, colidx = c.pattern.expand(c.nd, c.nd, mask)
    values = device.zeros(colidx.numel(), 'float64')
    kernels.assemble_matrix_terms(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=c.nd, ncr=c.nd, mask=mask,
                                  pattern=c.pattern, values=values, terms=[dict(C=.25 * C), dict(C=.75 * C)])
    assert numpy.array_equal(device.to_host(colidx), g['K_colidx'])
    close(device.to_host(values), g['K_values'])


@pytest.mark.parametrize('name', ['lap2d_spline2_5x4_iso', 'lap3d_p1_3_iso', 'lap1d_p1_5'])
def test_matrix_terms_point_dependent_forms(golden, name):
    '''Polynomial factor, scale array and the two product-rule kinds against nh_assemble_matrix with per-point tensors (cq_dev) built on the
    host from nh_sample_eval values: the path (and the torch glue in front of it) that the fused entry replaces.'''
    from nutils_amd import device, kernels
    g = golden(name)
    c = Case(g)
    rng = numpy.random.default_rng(7)
    nd, S, n = c.nd, 1 + c.nd, c.nelems * c.nq
    us = [device.to_dev(rng.normal(size=c.ndofs), 'float64') for _ in range(2)]
    sc = rng.uniform(.5, 1.5, n)
    C0, B1, B2 = (rng.normal(size=(1, S, 1, S)) for _ in range(3))
    L2 = rng.normal(size=(1, S))
    coeffs, powers = [1.5, -.5, .25], [[1, 0], [0, 2], [2, 1]]
    rowptr, colidx = c.pattern.expand(1, 1, None)
    values = device.zeros(colidx.numel(), 'float64')
    kernels.assemble_matrix_terms(nelems=c.nelems, ndims=nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=1, ncr=1, mask=None,
                                  pattern=c.pattern, values=values, fields=[(c.basis, u, 1) for u in us], polys=[([(0, 0), (1, 0)], coeffs, powers)],
                                  terms=[dict(C=C0, poly=0), dict(C=B1, kind=1, field=1, scale=device.to_dev(sc, 'float64')), dict(C=B2, kind=2, field=0, L=L2, poly=0)])
    U = []
    for u in us:
        Ud = device.empty(n * S, 'float64')
        kernels.sample_eval(nelems=c.nelems, ndims=nd, nq=c.nq, geom=c.geom, trial=c.basis, ncr=1, points=c.points, u=u, U=Ud)
        U.append(device.to_host(Ud).reshape(n, S))
    pv = sum(cf * U[0][:, 0] ** p0 * U[1][:, 0] ** p1 for cf, (p0, p1) in zip(coeffs, powers))
    cq = pv[:, None, None] * C0[0, :, 0, :]
    cq[:, :, 0] += sc[:, None] * (U[1] @ B1[0, :, 0, :].T)
    cq += pv[:, None, None] * L2[0][None, :, None] * (U[0] @ B2[0, :, 0, :])[:, None, :]
    ref = device.zeros(colidx.numel(), 'float64')
    kernels.assemble_matrix(nelems=c.nelems, ndims=nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=1, ncr=1, C=numpy.ones((1, S, 1, S)),
                            mask=None, pattern=c.pattern, values=ref, cq=device.to_dev(cq.reshape(-1), 'float64'))
    close(device.to_host(values), device.to_host(ref))


def test_matrix_terms_ragged(golden):
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden('hier_spline2_2d')
    pts = device.to_dev(g['gauss_coords'], 'float64')
    w = device.to_dev(g['gauss_weights'], 'float64')
    nq = len(g['gauss_weights'])
    geom = kernels.geometry_box(device.to_dev(g['elem_origin'], 'float64'), device.to_dev(g['elem_size'], 'float64'))
    for key in 'th':
        off_h = g[key + '_dof_offsets']
        ne, ndofs = len(off_h) - 1, int(g[key + '_ndofs'])
        off = device.to_dev(off_h, 'int64')
        dofs = device.to_dev(g[key + '_dofs'], 'int32')
        T = kernels.tabulate(device.to_dev(g[key + '_coeffs'], 'float64'), len(g[key + '_dofs']), g[key + '_coeffs'].shape[1], pts, nq, 2)
        b = kernels.basis(T, dofs, nb=0, off=off)
        pat = kernels.Pattern(ne, ndofs, ndofs, dofs, dofs, toff=off, roff=off)
        rowptr, colidx = pat.expand()
        values = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix_terms(nelems=ne, ndims=2, nq=nq, weights=w, geom=geom, test=b, trial=b, nct=1, ncr=1, mask=None, pattern=pat, values=values,
                                      terms=[dict(C=oa.laplace_coefficient(2))])
        close(device.to_host(values), g[key + 'K_values'])


# ---- deterministic owner-side reduction (NH_MATRIX_GATHER) ---------------------------------------------------------------------

@pytest.mark.parametrize('name', SCALAR + ELAST)
def test_gather_equals_atomics_and_is_reproducible(golden, name):
    '''Local matrices to scratch + one sum per CSR entry in (element, m, n) order: the golden values of the reference, and bit-identical
    results for repeated assemblies (the atomic scatter is only reproducible to rounding).'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    if name in ELAST:
        nc, C = c.nd, oa.elasticity_coefficient(c.nd, float(g['lam']), float(g['mu']))
        mask = oa.block_mask(C)
    else:
        nc, C, mask = 1, oa.laplace_coefficient(c.nd), None
    rowptr, colidx = c.pattern.expand(nc, nc, mask)
    out = []
    for it in range(3):
        values = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=nc, ncr=nc, C=C, mask=mask,
                                pattern=c.pattern, values=values, gather=True)
        out.append(device.to_host(values))
    close(out[0], g['K_values'])
    assert numpy.array_equal(out[0], out[1]) and numpy.array_equal(out[0], out[2])


@pytest.mark.parametrize('name', ['lap2d_p1_4x3_iso', 'lap3d_p2_2_iso', 'lap2d_spline2_5x4_iso', 'elast2d_p2_3x2_iso', 'elast3d_p1_2_iso'])
def test_vector_scatter_equals_the_reference_order_and_is_reproducible(golden, name):
    '''Residual vectors through the deterministic scatter (local vectors stored element-major + nh_scatter_gather: for every dof its contributions in
    ascending (element, local index) order, the order of the reference's numpy.add.at loop, evaluable.py:3405-3411): equal to the atomic scatter to
    rounding, BIT-IDENTICAL to the oracle's add.at restatement of the same local vectors summed in that order, and bit-identical from run to run.'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    nc = c.nd if name in ELAST else 1
    C = oa.elasticity_coefficient(c.nd, float(g['lam']), float(g['mu'])) if name in ELAST else oa.laplace_coefficient(c.nd)
    rng = numpy.random.default_rng(23)
    u = rng.normal(size=(c.ndofs, nc))
    ud = device.to_dev(u, 'float64')
    common = dict(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=nc, ncr=nc, C=C, u=ud)
    atomic = device.zeros(c.ndofs * nc, 'float64')
    kernels.assemble_vector(out=atomic, **common)
    plan = kernels.ScatterPlan(nelems=c.nelems, nrows=c.ndofs, nb=c.nb, dofs=c.dofs)
    outs, locals_ = [], []
    for it in range(3):
        local = device.empty(c.nelems * c.nb * nc, 'float64')
        kernels.assemble_vector(local=local, **common)
        out = device.zeros(c.ndofs * nc, 'float64')
        kernels.scatter_gather([(plan, local)], nc, out, accumulate=False)
        outs.append(device.to_host(out))
        locals_.append(device.to_host(local))
    a = device.to_host(atomic)
    assert numpy.abs(outs[0] - a).max() <= 1e-13 * numpy.abs(a).max()
    assert numpy.array_equal(outs[0], outs[1]) and numpy.array_equal(outs[0], outs[2])
    # the reference's scatter of the same local vectors: numpy.add.at element by element (ascending element, then local index)
    ref = numpy.zeros((c.ndofs, nc))
    numpy.add.at(ref, g['dofs'].reshape(-1), locals_[0].reshape(-1, nc))
    assert numpy.array_equal(outs[0].reshape(-1, nc), ref)
    # two plans / arrays gathered in one call add up in the order given; accumulate adds to what is there
    out2 = device.zeros(c.ndofs * nc, 'float64')
    kernels.scatter_gather([(plan, device.to_dev(locals_[0], 'float64')), (plan, device.to_dev(locals_[0], 'float64'))], nc, out2, accumulate=True)
    ref2 = ref.copy()  # (the second array continues the running sums of the first)
    numpy.add.at(ref2, g['dofs'].reshape(-1), locals_[0].reshape(-1, nc))
    assert numpy.array_equal(device.to_host(out2), ref2.reshape(-1))


def test_vector_scatter_on_an_element_list_and_a_ragged_basis(golden):
    '''the map of a boundary-type sample (element list: only the listed elements contribute, in list order) and of a ragged (hierarchical) basis with offsets'''
    from nutils_amd import device, kernels
    g = golden('hier_spline2_2d')
    dofs, off = g['t_dofs'], g['t_dof_offsets']
    ne, ndofs = len(off) - 1, int(g['t_ndofs'])
    rng = numpy.random.default_rng(4)
    local = rng.normal(size=(len(dofs), 2))
    ddofs, doff = device.to_dev(dofs, 'int32'), device.to_dev(off, 'int64')
    plan = kernels.ScatterPlan(nelems=ne, nrows=ndofs, nb=0, dofs=ddofs, off=doff)
    out = device.zeros(ndofs * 2, 'float64')
    kernels.scatter_gather([(plan, device.to_dev(local, 'float64'))], 2, out, accumulate=False)
    ref = numpy.zeros((ndofs, 2))
    numpy.add.at(ref, dofs, local)
    assert numpy.array_equal(device.to_host(out).reshape(-1, 2), ref)
    elist = numpy.array([5, 2, 7, 0][:min(4, ne)], dtype=numpy.int32)  # (not ascending: the order of the list is the order of the sum)
    plan2 = kernels.ScatterPlan(nelems=ne, nrows=ndofs, nb=0, dofs=ddofs, off=doff, elist=device.to_dev(elist, 'int32'), nlist=len(elist))
    kernels.scatter_gather([(plan2, device.to_dev(local, 'float64'))], 2, out, accumulate=False)
    ref = numpy.zeros((ndofs, 2))
    for e in elist:
        numpy.add.at(ref, dofs[off[e]:off[e + 1]], local[off[e]:off[e + 1]])
    assert numpy.array_equal(device.to_host(out).reshape(-1, 2), ref)


@pytest.mark.parametrize('name', ['lap3d_p2_2_iso', 'lap3d_spline2_3_iso', 'lap2d_spline2_5x4_iso', 'lap3d_p1_543_iso'])
def test_gather_with_a_full_coefficient_tensor(golden, name):
    '''A dense, non-symmetric form tensor C[a][b] (value and gradient slots mixed) and a scale array through both thread passes of NH_MATRIX_GATHER
    (k_local_scalar; k_local_rows for the 27-function elements) against the one-wave-per-element kernel with atomics.'''
    from nutils_amd import device, kernels
    g = golden(name)
    c = Case(g)
    rng = numpy.random.default_rng(17)
    S = 1 + c.nd
    C = rng.normal(size=(1, S, 1, S))
    scale = device.to_dev(rng.uniform(.5, 1.5, c.nelems * c.nq), 'float64')
    rowptr, colidx = c.pattern.expand(1, 1, None)
    out = []
    for gather in (False, True, True):
        values = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=1, ncr=1, C=C, mask=None,
                                pattern=c.pattern, values=values, gather=gather, scale=scale)
        out.append(device.to_host(values))
    close(out[1], out[0])
    assert numpy.array_equal(out[1], out[2])


@pytest.mark.parametrize('name', ['lap3d_p1_543_iso', 'lap2d_p1_4x3_iso', 'lap2d_spline2_5x4_iso'])
def test_vector_thread_pass_isotropic_form_in_closed_form(golden, name, monkeypatch):
    '''Vector-valued blocks on the small uniform bases of the thread pass (k_local_rows_v): the isotropic three-parameter family lam d_ca d_db + mu d_cd d_ab +
    mu2 d_cb d_ad is recognised on the host and applied in closed form (no form tensor in LDS); against the same kernel with the tensor read from LDS
    (NUTILS_AMD_NO_ISOFORM=1), against the one-wave-per-element kernel with atomics, and -- with three different parameters and a scale array -- a family member that
    is NOT symmetric.'''
    from nutils_amd import device, kernels
    g = golden(name)
    c = Case(g)
    nd, S = c.nd, 1 + c.nd
    rng = numpy.random.default_rng(23)
    lam, mu, mu2 = 1.3, .7, -.4
    C = numpy.zeros((nd, S, nd, S))
    for cc in range(nd):
        for a in range(nd):
            for d in range(nd):
                for b in range(nd):
                    C[cc, 1 + a, d, 1 + b] = lam * (cc == a and d == b) + mu * (cc == d and a == b) + mu2 * (cc == b and a == d)
    scale = device.to_dev(rng.uniform(.5, 1.5, c.nelems * c.nq), 'float64')
    rowptr, colidx = c.pattern.expand(nd, nd, None)

    def run(gather):
        values = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=c.nelems, ndims=nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=nd, ncr=nd, C=C, mask=None,
                                pattern=c.pattern, values=values, gather=gather, scale=scale)
        return device.to_host(values)
    atomics, closed = run(False), run(True)
    monkeypatch.setenv('NUTILS_AMD_NO_ISOFORM', '1')
    table = run(True)
    close(closed, atomics)
    close(closed, table)
    assert numpy.abs(closed - table).max() > 0  # (two different instruction sequences)


def test_gather_ragged_and_structured_large(golden):
    '''Ragged bases (size classes + gather) and a structured trilinear mesh of 24^3 elements against the atomic scatter.'''
    from nutils_amd import device, kernels, mesh, sample, points
    from oracle import assemble as oa
    g = golden('hier_spline2_2d')
    pts = device.to_dev(g['gauss_coords'], 'float64')
    w = device.to_dev(g['gauss_weights'], 'float64')
    nq = len(g['gauss_weights'])
    geom = kernels.geometry_box(device.to_dev(g['elem_origin'], 'float64'), device.to_dev(g['elem_size'], 'float64'))
    for key in 'th':
        off_h = g[key + '_dof_offsets']
        ne, ndofs = len(off_h) - 1, int(g[key + '_ndofs'])
        off = device.to_dev(off_h, 'int64')
        dofs = device.to_dev(g[key + '_dofs'], 'int32')
        T = kernels.tabulate(device.to_dev(g[key + '_coeffs'], 'float64'), len(g[key + '_dofs']), g[key + '_coeffs'].shape[1], pts, nq, 2)
        b = kernels.basis(T, dofs, nb=0, off=off)
        pat = kernels.Pattern(ne, ndofs, ndofs, dofs, dofs, toff=off, roff=off)
        rowptr, colidx = pat.expand()
        values = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=ne, ndims=2, nq=nq, weights=w, geom=geom, test=b, trial=b, nct=1, ncr=1, C=oa.laplace_coefficient(2), mask=None, pattern=pat,
                                values=values, gather=True)
        close(device.to_host(values), g[key + 'K_values'])
    # structured: the API path switches to the gather on the second assembly of a pattern
    from nutils_amd import function
    rng = numpy.random.default_rng(2)
    domain, geom0 = mesh.rectilinear([24, 24, 24])
    basis = domain.basis('std', degree=1)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(25.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (len(basis), 3))
    geomf = basis @ verts
    K = domain.integral((function.outer(function.grad(basis, geomf)).sum(-1) + 3. * function.outer(basis)) * function.J(geomf), degree=2)
    import os
    os.environ['NUTILS_AMD_NO_FAST_PATH'] = '1'
    try:
        v0, rp, ci = function.eval(function.as_csr(K))  # first assembly: atomics
        v1, _, _ = function.eval(function.as_csr(K))    # second: gather
        v2, _, _ = function.eval(function.as_csr(K))
    finally:
        del os.environ['NUTILS_AMD_NO_FAST_PATH']
    close(v1, v0)
    assert numpy.array_equal(v1, v2)


@pytest.mark.parametrize('name', ['lap2d_spline2_5x4_iso', 'lap2d_spline2_4x4', 'lap3d_p1_3_iso', 'lap2d_p1_4x3_iso', 'lap1d_p1_5'])
def test_matrix_terms_gather_path(golden, name):
    '''Scalar blocks on small uniform bases: thread-per-element pass + owner-side reduction (NH_MATRIX_GATHER through nh_assemble_matrix_terms)
    against the workgroup-batched kernel with atomics, with polynomial factors and both product-rule kinds; bit-identical when repeated.'''
    from nutils_amd import device, kernels
    g = golden(name)
    c = Case(g)
    rng = numpy.random.default_rng(9)
    nd, S, n = c.nd, 1 + c.nd, c.nelems * c.nq
    us = [device.to_dev(rng.normal(size=c.ndofs), 'float64') for _ in range(2)]
    sc = device.to_dev(rng.uniform(.5, 1.5, n), 'float64')
    C0, B1, B2 = (rng.normal(size=(1, S, 1, S)) for _ in range(3))
    L2 = rng.normal(size=(1, S))
    kw = dict(nelems=c.nelems, ndims=nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=1, ncr=1, mask=None, pattern=c.pattern,
              fields=[(c.basis, u, 1) for u in us], polys=[([(0, 0), (1, 0)], [1.5, -.5, .25], [[1, 0], [0, 2], [2, 1]])],
              terms=[dict(C=C0, poly=0), dict(C=B1, kind=1, field=1, scale=sc), dict(C=B2, kind=2, field=0, L=L2, poly=0), dict(C=.5 * C0)])
    rowptr, colidx = c.pattern.expand(1, 1, None)
    out = []
    for gather in (False, True, True):
        values = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix_terms(values=values, gather=gather, **kw)
        out.append(device.to_host(values))
    close(out[1], out[0])
    assert numpy.array_equal(out[1], out[2])


@pytest.mark.parametrize('name', ['lap2d_spline2_5x4_iso', 'lap3d_p1_3_iso'])
def test_terms_point_factor(golden, name):
    '''qs = (B, t, r): the term is multiplied by U_t . B . U_r at the point, in nh_assemble_terms and both paths of nh_assemble_matrix_terms;
    against the same term with that factor handed over as a scale array (nh_sample_eval values, host algebra).'''
    from nutils_amd import device, kernels
    g = golden(name)
    c = Case(g)
    rng = numpy.random.default_rng(13)
    nd, S, n = c.nd, 1 + c.nd, c.nelems * c.nq
    us = [device.to_dev(rng.normal(size=c.ndofs), 'float64') for _ in range(2)]
    U = []
    for u in us:
        Ud = device.empty(n * S, 'float64')
        kernels.sample_eval(nelems=c.nelems, ndims=nd, nq=c.nq, geom=c.geom, trial=c.basis, ncr=1, points=c.points, u=u, U=Ud)
        U.append(device.to_host(Ud).reshape(n, S))
    B = rng.normal(size=(S, S))
    sc = device.to_dev(numpy.einsum('qa,ab,qb->q', U[0], B, U[1]), 'float64')
    C = rng.normal(size=(1, S, 1, S))
    fields = [(c.basis, u, 1) for u in us]
    out = [device.zeros(c.ndofs, 'float64') for _ in range(2)]
    kw = dict(nelems=c.nelems, ndims=nd, nq=c.nq, weights=c.weights, geom=c.geom, fields=fields)
    kernels.assemble_terms(blocks=[(c.basis, 1, out[0])], terms=[dict(block=0, field=1, C=C, qs=(B, 0, 1))], **kw)
    kernels.assemble_terms(blocks=[(c.basis, 1, out[1])], terms=[dict(block=0, field=1, C=C, scale=sc)], **kw)
    close(device.to_host(out[0]), device.to_host(out[1]))
    rowptr, colidx = c.pattern.expand(1, 1, None)
    vals = []
    for terms, gather in (([dict(C=C, qs=(B, 0, 1))], False), ([dict(C=C, qs=(B, 0, 1))], True), ([dict(C=C, scale=sc)], False)):
        v = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix_terms(test=c.basis, trial=c.basis, nct=1, ncr=1, mask=None, pattern=c.pattern, values=v, terms=terms, gather=gather, **kw)
        vals.append(device.to_host(v))
    close(vals[0], vals[2])
    close(vals[1], vals[2])


# ---- owner blocks (NH_MATRIX_FUSED) ------------------------------------------------------------------------------------------------------

FUSED_SIZES = {(1, 2), (1, 3), (2, 3), (2, 4), (2, 9), (3, 4), (3, 8)}  # (dimension, functions per element) with a thread pass


@pytest.mark.parametrize('name', SCALAR)
def test_fused_owner_blocks_equal_the_reference(golden, name):
    '''One pass without scratch or global atomics: the golden values of the reference, accumulating into a zeroed array and STORING into a poisoned one; the
    block plan exists exactly for the element sizes that have a thread pass (the others take the default path behind the same flag).'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    C = oa.laplace_coefficient(c.nd)
    rowptr, colidx = c.pattern.expand(1, 1, None)
    for store in (False, True):
        values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64') if store else device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=1, ncr=1, C=C, mask=None,
                                pattern=c.pattern, values=values, fused=True, store=store)
        close(device.to_host(values), g['K_values'])
    nblocks, rpb, nvisits = c.pattern.fused_info()
    if (c.nd, c.nb) in FUSED_SIZES:
        assert nblocks >= 1 and nvisits >= c.nelems, (nblocks, rpb, nvisits)
        assert c.pattern.fused_routine == 0  # (Case passes per-element tables: the tabulated any-element routine; the trilinear one is tested below)
        # same element routine, and every entry summed over its slots in the order of the gather map: bit for bit the two-pass result
        gathered = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=1, ncr=1, C=C, mask=None,
                                pattern=c.pattern, values=gathered, gather=True)
        assert numpy.array_equal(device.to_host(values), device.to_host(gathered))
    else:
        assert nblocks == 0


@pytest.mark.parametrize('name', ['lap2d_p1_4x3_iso', 'lap2d_spline2_5x4_iso', 'lap3d_p1_543_iso'])
def test_fused_with_a_full_coefficient_tensor(golden, name):
    '''A dense, non-symmetric form tensor (value and gradient slots mixed) and a scale array: the general instantiation of the fused kernel against the
    one-wave-per-element kernel with atomics.'''
    from nutils_amd import device, kernels
    g = golden(name)
    c = Case(g)
    rng = numpy.random.default_rng(17)
    S = 1 + c.nd
    C = rng.normal(size=(1, S, 1, S))
    scale = device.to_dev(rng.uniform(.5, 1.5, c.nelems * c.nq), 'float64')
    rowptr, colidx = c.pattern.expand(1, 1, None)
    out = []
    for fused in (False, True):
        values = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=1, ncr=1, C=C, mask=None,
                                pattern=c.pattern, values=values, fused=fused, gather=False, scale=scale)
        out.append(device.to_host(values))
    assert c.pattern.fused_info()[0] >= 1
    close(out[1], out[0])


@pytest.mark.parametrize('shuffle', [False, True])
def test_fused_many_blocks_any_numbering(shuffle):
    '''28^3 trilinear elements on a perturbed mesh (24 389 rows: ~48 owner blocks, elements on block borders recomputed) against the deterministic gather path --
    with the natural numbering and with elements AND dofs renumbered at random (the clustering sees coordinates, not numbers).'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    n = 28
    rng = numpy.random.default_rng(5)
    dofs, coeffs, ndofs = oa.structured_basis((n, n, n), 'std', 1)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (ndofs, 3))
    dofs = numpy.asarray(dofs).reshape(-1, 8)
    if shuffle:
        perm = rng.permutation(ndofs)          # new number of old dof i
        dofs = perm[dofs][rng.permutation(len(dofs))]
        v2 = numpy.empty_like(verts)
        v2[perm] = verts
        verts = v2
    ne = len(dofs)
    pts, w = oa.gauss(2, 3)
    p = device.to_dev(pts, 'float64')
    T = kernels.tabulate(device.to_dev(coeffs[0], 'float64'), 8, coeffs.shape[2], p, len(pts), 3)
    d = device.to_dev(dofs.ravel(), 'int32')
    basis = kernels.basis(T, d, nb=8)
    geom = kernels.geometry_iso(8, T, d, device.to_dev(verts, 'float64'))
    pattern = kernels.Pattern(ne, ndofs, ndofs, d, d, nbt=8, nbr=8)
    rowptr, colidx = pattern.expand(1, 1, None)
    C = oa.laplace_coefficient(3) + 2. * oa.mass_coefficient(3)
    out = []
    for kw in (dict(gather=True), dict(fused=True, store=True), dict(fused=True, store=True)):
        values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64') if kw.get('store') else device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=ne, ndims=3, nq=len(pts), weights=device.to_dev(w, 'float64'), geom=geom, test=basis, trial=basis, nct=1, ncr=1, C=C, mask=None,
                                pattern=pattern, values=values, **kw)
        out.append(device.to_host(values))
    nblocks, rpb, nvisits = pattern.fused_info()
    assert pattern.fused_routine == 2  # (stiffness + mass on trilinear hexahedra: the sum-factorised routine with the mass term)
    assert nblocks >= -(-ndofs // rpb) and nblocks > 40, (nblocks, rpb)  # (Morton boxes of at most rpb rows)
    assert ne < nvisits < 2.6 * ne, nvisits / ne  # boxes of 8 x 4 x 4 nodes: 9 * 5 * 5 / 128 = 1.76 in the interior, more at this size
    close(out[1], out[0])
    close(out[2], out[0])
    # every contribution has its own slot, the slots of an entry are summed in (element, m, n) order: two assemblies agree bit for bit
    assert numpy.array_equal(out[1], out[2])


def test_fused_trilinear_routine_equals_the_tabulated_one(monkeypatch):
    '''The sum-factorised routine for trilinear hexahedra (chosen from the tables of the launch) against the tabulated any-element routine of the same owner blocks, with a
    coefficient array at the Gauss points, on an unstructured numbering; and forms the routine does not cover (anisotropic diffusion) keep the tabulated one.'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    n = 12
    rng = numpy.random.default_rng(11)
    dofs, coeffs, ndofs = oa.structured_basis((n, n, n), 'std', 1)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (ndofs, 3))
    perm = rng.permutation(ndofs)
    dofs = perm[numpy.asarray(dofs).reshape(-1, 8)][rng.permutation(n ** 3)]
    v2 = numpy.empty_like(verts)
    v2[perm] = verts
    ne = len(dofs)
    pts, w = oa.gauss(2, 3)
    T = kernels.tabulate(device.to_dev(coeffs[0], 'float64'), 8, coeffs.shape[2], device.to_dev(pts, 'float64'), len(pts), 3)
    d = device.to_dev(dofs.ravel(), 'int32')
    basis = kernels.basis(T, d, nb=8)
    geom = kernels.geometry_iso(8, T, d, device.to_dev(v2, 'float64'))
    scale = device.to_dev(rng.uniform(.5, 1.5, ne * 8), 'float64')
    wd = device.to_dev(w, 'float64')
    aniso = oa.laplace_coefficient(3).copy()
    aniso[0, 1, 0, 1] = 2.
    for C, routine in ((oa.laplace_coefficient(3), 1), (3. * oa.laplace_coefficient(3) + .5 * oa.mass_coefficient(3), 2), (aniso, 0)):
        out = []
        for no_fast in (False, True):
            if no_fast:
                monkeypatch.setenv('NUTILS_AMD_NO_FUSED_P1HEX', '1')
            else:
                monkeypatch.delenv('NUTILS_AMD_NO_FUSED_P1HEX', raising=False)
            pattern = kernels.Pattern(ne, ndofs, ndofs, d, d, nbt=8, nbr=8)
            rowptr, colidx = pattern.expand(1, 1, None)
            values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64')
            kernels.assemble_matrix(nelems=ne, ndims=3, nq=8, weights=wd, geom=geom, test=basis, trial=basis, nct=1, ncr=1, C=C, mask=None, pattern=pattern, values=values,
                                    fused=True, store=True, scale=scale)
            pattern.fused_info()
            assert pattern.fused_routine == (0 if no_fast else routine)
            out.append(device.to_host(values))
        close(out[0], out[1])


def test_ragged_blocks_become_reproducible_from_the_second_assembly(golden):
    '''Ragged bases (truncated hierarchical splines): the automatic choice of kernels.assemble_matrix takes the owner-side gather from the second assembly of a pattern on --
    the values of the reference every time, bit-identical from then on.'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    g = golden('hier_spline2_2d')
    pts = device.to_dev(g['gauss_coords'], 'float64')
    w = device.to_dev(g['gauss_weights'], 'float64')
    nq = len(g['gauss_weights'])
    geom = kernels.geometry_box(device.to_dev(g['elem_origin'], 'float64'), device.to_dev(g['elem_size'], 'float64'))
    off_h = g['t_dof_offsets']
    ne, ndofs = len(off_h) - 1, int(g['t_ndofs'])
    off = device.to_dev(off_h, 'int64')
    dofs = device.to_dev(g['t_dofs'], 'int32')
    T = kernels.tabulate(device.to_dev(g['t_coeffs'], 'float64'), len(g['t_dofs']), g['t_coeffs'].shape[1], pts, nq, 2)
    b = kernels.basis(T, dofs, nb=0, off=off)
    pat = kernels.Pattern(ne, ndofs, ndofs, dofs, dofs, toff=off, roff=off)
    rowptr, colidx = pat.expand()
    out = []
    for it in range(4):
        values = device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=ne, ndims=2, nq=nq, weights=w, geom=geom, test=b, trial=b, nct=1, ncr=1, C=oa.laplace_coefficient(2), mask=None, pattern=pat, values=values)
        out.append(device.to_host(values))
        close(out[-1], g['tK_values'])
    assert numpy.array_equal(out[1], out[2]) and numpy.array_equal(out[2], out[3])


@pytest.mark.parametrize('nd', [2, 3])
def test_fused_simplex_meshes(nd):
    '''Unstructured-type connectivity the golden fixtures do not hold: P1 triangles (quadrilaterals split in two) and P1 tetrahedra (Kuhn subdivision of hexahedra) with an
    isoparametric simplex geometry (nd + 1 geometry functions: the general vertex loop of the geometry routines), randomly renumbered -- owner blocks against the oracle's
    assembly and against the deterministic gather.'''
    import itertools
    from nutils_amd import device, kernels
    from oracle import assemble as oa, poly
    n = 14 if nd == 2 else 7
    rng = numpy.random.default_rng(21)
    grid = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * nd, indexing='ij'), -1)
    vid = numpy.arange((n + 1) ** nd).reshape((n + 1,) * nd)
    verts = grid.reshape(-1, nd) + rng.uniform(-.15, .15, ((n + 1) ** nd, nd))
    cells = []
    for c in itertools.product(range(n), repeat=nd):
        c = numpy.array(c)
        if nd == 2:
            v = lambda i, j: vid[c[0] + i, c[1] + j]
            cells += [[v(0, 0), v(1, 0), v(0, 1)], [v(1, 1), v(0, 1), v(1, 0)]]
        else:
            for perm in itertools.permutations(range(3)):
                p = [c.copy()]
                for ax in perm:
                    q = p[-1].copy()
                    q[ax] += 1
                    p.append(q)
                cells.append([vid[tuple(x)] for x in p])
    dofs = numpy.array(cells, dtype=numpy.int64)
    ndofs = len(verts)
    perm = rng.permutation(ndofs)
    dofs = perm[dofs][rng.permutation(len(dofs))]
    v2 = numpy.empty_like(verts)
    v2[perm] = verts
    ne, nb = dofs.shape
    P = poly.powers(nd, 1)  # exponent table of the degree-1 polynomials in the reference's coefficient order
    coeffs = numpy.zeros((nb, len(P)))
    for k, e in enumerate(P):
        if e.sum() == 0:
            coeffs[0, k] = 1.
        else:
            ax = int(numpy.argmax(e))
            coeffs[0, k], coeffs[1 + ax, k] = -1., 1.
    if nd == 2:
        pts = numpy.array([[1 / 6, 1 / 6], [2 / 3, 1 / 6], [1 / 6, 2 / 3]])
        w = numpy.full(3, 1 / 6)
    else:
        a, b = .5854101966249685, .1381966011250105
        pts = numpy.array([[b, b, b], [a, b, b], [b, a, b], [b, b, a]])
        w = numpy.full(4, 1 / 24)
    C = oa.laplace_coefficient(nd) + .7 * oa.mass_coefficient(nd)
    # oracle
    N, dN = oa.tabulate(coeffs, pts)
    x, J = oa.geometry_iso(v2, dofs, numpy.broadcast_to(N, (ne,) + N.shape), numpy.broadcast_to(dN, (ne,) + dN.shape))
    D, det = oa.physical_tables(numpy.broadcast_to(N, (ne,) + N.shape), numpy.broadcast_to(dN, (ne,) + dN.shape), J)
    vo, rpo, cio = oa.assemble_csr(oa.local_matrices(D, D, det * w, C), dofs, dofs, ndofs, ndofs)
    # device
    T = kernels.tabulate(device.to_dev(coeffs, 'float64'), nb, coeffs.shape[1], device.to_dev(pts, 'float64'), len(pts), nd)
    d = device.to_dev(dofs.ravel(), 'int32')
    basis = kernels.basis(T, d, nb=nb)
    geom = kernels.geometry_iso(nb, T, d, device.to_dev(v2, 'float64'))
    pattern = kernels.Pattern(ne, ndofs, ndofs, d, d, nbt=nb, nbr=nb)
    rowptr, colidx = pattern.expand(1, 1, None)
    assert numpy.array_equal(device.to_host(rowptr), rpo) and numpy.array_equal(device.to_host(colidx), cio)
    out = {}
    for name, kw in (('fused', dict(fused=True, store=True)), ('gather', dict(gather=True))):
        values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64') if kw.get('store') else device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nelems=ne, ndims=nd, nq=len(pts), weights=device.to_dev(w, 'float64'), geom=geom, test=basis, trial=basis, nct=1, ncr=1, C=C, mask=None,
                                pattern=pattern, values=values, **kw)
        out[name] = device.to_host(values)
        close(out[name], vo)
    nblocks, rpb, nvisits = pattern.fused_info()
    assert nblocks >= 1 and pattern.fused_routine == 0 and nvisits >= ne


def test_pattern_union_is_a_rowwise_merge():
    '''nh_pattern_union_*: union of sorted-unique CSR patterns and the position of every entry of every part in it, against numpy (the union pattern of matrix
    integrals on several samples: sample._run_parts; replaces the sort-based unique of evaluable.py:5560-5682).'''
    from nutils_amd import device, kernels
    rng = numpy.random.default_rng(4)
    nrows, ncols = 300, 500
    parts = []
    for dens in (.05, .01, .2, .0):
        mask = rng.uniform(size=(nrows, ncols)) < dens
        mask[7] = False  # (an empty row in every part)
        rp = numpy.concatenate([[0], numpy.cumsum(mask.sum(1))]).astype(numpy.int64)
        ci = numpy.nonzero(mask)[1].astype(numpy.int64)
        parts.append((mask, rp, ci))
    rowptr, colidx, pos = kernels.pattern_union([(device.to_dev(rp, 'int64'), device.to_dev(ci, 'int64')) for _, rp, ci in parts])
    union = numpy.logical_or.reduce([m for m, _, _ in parts])
    assert numpy.array_equal(device.to_host(rowptr), numpy.concatenate([[0], numpy.cumsum(union.sum(1))]))
    cu = device.to_host(colidx)
    assert numpy.array_equal(cu, numpy.nonzero(union)[1])
    ru = numpy.repeat(numpy.arange(nrows), union.sum(1))
    for (m, rp, ci), p in zip(parts, pos):
        ph = device.to_host(p)
        assert numpy.array_equal(cu[ph], ci) and numpy.array_equal(ru[ph], numpy.repeat(numpy.arange(nrows), numpy.diff(rp)))
