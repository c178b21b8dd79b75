'''Parity of the write-once quadratic-hex kernel (nh_p2hex_matrix, BASELINE.json configs[2]) with the oracle
(oracle/assemble.py: the reference's element loop + stable-sort dedup restated in numpy) at sizes between the golden
2^3 case and the 64^3 property checks: index arrays bit-exact, values within 1e-13 of max|K|.'''
import numpy
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-13


def _inputs(shape, iso, seed=0):
    from oracle import assemble as oa
    rng = numpy.random.default_rng(seed)
    dofs, coeffs, ndofs = oa.structured_basis(shape, 'std', 2)
    gdofs, gcoeffs, nverts = oa.structured_basis(shape, 'std', 1)
    pts, w = oa.gauss(4, 3)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, 3)
    if iso:
        verts = verts + rng.uniform(-.2, .2, verts.shape)
    return dict(shape=shape, dofs=dofs, coeffs=coeffs, ndofs=ndofs, gdofs=gdofs, gcoeffs=gcoeffs, pts=pts, w=w, verts=verts)


def _oracle(inp, C, scale=None):
    from oracle import assemble as oa
    N, dN = oa.tabulate(inp['coeffs'][0], inp['pts'])
    gN, gdN = oa.tabulate(inp['gcoeffs'][0], inp['pts'])
    ne = len(inp['dofs'])
    x, J = oa.geometry_iso(inp['verts'], inp['gdofs'], numpy.broadcast_to(gN, (ne,) + gN.shape), numpy.broadcast_to(gdN, (ne,) + gdN.shape))
    D, det = oa.physical_tables(numpy.broadcast_to(N, (ne,) + N.shape), numpy.broadcast_to(dN, (ne,) + dN.shape), J)
    wdet = det * inp['w']
    if scale is not None:
        wdet = wdet * scale
    A = oa.local_matrices(D, D, wdet, C)
    return oa.assemble_csr(A, inp['dofs'], inp['dofs'], inp['ndofs'], inp['ndofs'])


class Dev:
    def __init__(self, inp):
        from nutils_amd import device, kernels
        self.inp = inp
        nq = len(inp['pts'])
        p = device.to_dev(inp['pts'], 'float64')
        self.w = device.to_dev(inp['w'], 'float64')
        self.T = kernels.tabulate(device.to_dev(inp['coeffs'][0], 'float64'), 27, inp['coeffs'].shape[2], p, nq, 3)
        gT = kernels.tabulate(device.to_dev(inp['gcoeffs'][0], 'float64'), 8, inp['gcoeffs'].shape[2], p, nq, 3)
        self.geom = kernels.geometry_iso(8, gT, device.to_dev(inp['gdofs'], 'int32'), device.to_dev(inp['verts'], 'float64'))
        self.dofs = device.to_dev(inp['dofs'], 'int32')
        self.nq = nq
        ne = len(inp['dofs'])
        self.pattern = kernels.Pattern(ne, inp['ndofs'], inp['ndofs'], self.dofs, self.dofs, nbt=27, nbr=27)

    def fast(self, C, nc, scale=None, **kw):
        from nutils_amd import device, kernels
        rowptr, colidx = self.pattern.expand(nc, nc, None)
        values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64')  # write-once: every entry must be overwritten
        kernels.p2hex_matrix(values=values, shape=self.inp['shape'], nq=self.nq, weights=self.w, geom=self.geom, T=self.T, ncomp=nc, C=C,
                             scale=None if scale is None else device.to_dev(scale, 'float64'), **kw)
        return device.to_host(values), device.to_host(rowptr), device.to_host(colidx)


def _check(got, ref):
    v, rp, ci = got
    vo, rpo, cio = ref
    assert numpy.array_equal(rp, rpo) and numpy.array_equal(ci, cio)
    assert not numpy.isnan(v).any(), f'{numpy.isnan(v).sum()} entries never written'
    err = numpy.abs(v - vo).max() / numpy.abs(vo).max()
    assert err < RTOL, err


@pytest.mark.parametrize('shape,iso', [((1, 1, 1), True), ((2, 1, 3), True), ((3, 4, 5), True), ((6, 6, 6), True), ((7, 7, 7), True), ((4, 4, 4), False)])
def test_elasticity_vs_oracle(shape, iso):
    from oracle import assemble as oa
    inp = _inputs(shape, iso)
    C = oa.elasticity_coefficient(3, 1., .5 / .3 - 1)
    _check(Dev(inp).fast(C, 3), _oracle(inp, C))


def test_closed_form_rowptr():
    from nutils_amd import device, kernels
    inp = _inputs((3, 2, 4), True)
    d = Dev(inp)
    rowptr, _ = d.pattern.expand(1, 1, None)
    rp = device.to_host(rowptr)
    for node in list(range(0, inp['ndofs'], 7)) + [inp['ndofs'] - 1, inp['ndofs']]:
        assert kernels.p2hex_rowptr(inp['shape'], node) == rp[node]


@pytest.mark.parametrize('kind', ['laplace', 'laplace+mass', 'iso3', 'vector-laplace', 'dense3', 'dense2', 'dense3+mass'])
def test_other_forms_vs_oracle(kind):
    from oracle import assemble as oa
    inp = _inputs((3, 3, 4), True, seed=1)
    rng = numpy.random.default_rng(5)
    if kind == 'laplace':
        nc, C = 1, oa.laplace_coefficient(3)
    elif kind == 'laplace+mass':
        nc, C = 1, oa.laplace_coefficient(3) + 2.5 * oa.mass_coefficient(3)
    elif kind == 'iso3':  # the three-parameter isotropic family with distinct parameters
        nc = 3
        C = numpy.zeros((3, 4, 3, 4))
        d = numpy.eye(3)
        C[:, 1:, :, 1:] = .7 * numpy.einsum('ca,db->cadb', d, d) + 1.3 * numpy.einsum('cd,ab->cadb', d, d) - .4 * numpy.einsum('cb,ad->cadb', d, d)
    elif kind == 'vector-laplace':
        nc, C = 3, oa.laplace_coefficient(3, 3)
    elif kind == 'dense3':
        nc = 3
        C = numpy.zeros((3, 4, 3, 4))
        C[:, 1:, :, 1:] = rng.normal(size=(3, 3, 3, 3))
    elif kind == 'dense2':
        nc = 2
        C = numpy.zeros((2, 4, 2, 4))
        C[:, 1:, :, 1:] = rng.normal(size=(2, 3, 2, 3))
    else:
        nc, C = 3, rng.normal(size=(3, 4, 3, 4))
    mask = oa.block_mask(C)
    if not mask.all():  # decoupled blocks are pruned from the pattern by the front end: this kernel is for fully coupled forms
        C = C + 1e-3 * numpy.einsum('cd,ab->cadb', ~mask, numpy.eye(4)) * (numpy.arange(4) > 0)[None, :, None, None]
    _check(Dev(inp).fast(C, nc), _oracle(inp, C))


def test_pointwise_scale():
    from oracle import assemble as oa
    inp = _inputs((3, 2, 3), True, seed=2)
    rng = numpy.random.default_rng(7)
    scale = rng.uniform(-1., 2., (len(inp['dofs']), len(inp['pts'])))  # signed: the weight is not split into square roots
    C = oa.elasticity_coefficient(3, 2., .7)
    _check(Dev(inp).fast(C, 3, scale=scale), _oracle(inp, C, scale=scale))


def test_slabs_add_up():
    '''Element layers [0, L) and [L, n) assembled separately (the multi-GPU partition): rows of the planes below 2L come from the first
    call, above from the second, the rows of the interface plane 2L are the sum of both.'''
    from oracle import assemble as oa
    from nutils_amd import device, kernels
    inp = _inputs((5, 3, 4), True, seed=3)
    C = oa.elasticity_coefficient(3, 1., .5)
    d = Dev(inp)
    vo, rpo, cio = _oracle(inp, C)
    L = 2
    va, rp, ci = d.fast(C, 3, layers=(0, L), owners=(0, L))
    vb, _, _ = d.fast(C, 3, layers=(L, 5), owners=(L, 5))
    nplane = 7 * 9 * 3  # rows per node plane
    r0, r1 = rp[2 * L * nplane], rp[(2 * L + 1) * nplane]
    assert not numpy.isnan(va[:r1]).any() and numpy.isnan(va[r1:]).all()
    assert not numpy.isnan(vb[r0:]).any() and numpy.isnan(vb[:r0]).all()
    full = numpy.concatenate([va[:r0], va[r0:r1] + vb[r0:r1], vb[r1:]])
    assert numpy.abs(full - vo).max() < RTOL * numpy.abs(vo).max()


def test_through_the_api_matches_generic(monkeypatch):
    '''domain.integral(...) of the elasticity form on a P2 vector field: fast path and generic path (NUTILS_AMD_NO_FAST_PATH) agree.'''
    from nutils_amd import mesh, function
    n = 5
    rng = numpy.random.default_rng(0)

    def run():
        domain, geom = mesh.rectilinear([n, n - 1, n + 1])
        gb = domain.basis('std', degree=1)
        verts = numpy.stack(numpy.meshgrid(numpy.arange(n + 1.), numpy.arange(n + 0.), numpy.arange(n + 2.), indexing='ij'), -1).reshape(-1, 3)
        geom = gb @ (verts + numpy.random.default_rng(0).uniform(-.2, .2, verts.shape))
        u = domain.field('u', btype='std', degree=2, shape=[3])
        v = domain.field('v', btype='std', degree=2, shape=[3])
        eps = lambda w: function.symgrad(w, geom)
        sigma = 1. * function.div(u, geom) * function.eye(3) + 2 * .6 * eps(u)
        res = domain.integral(function.inner(eps(v), sigma) * function.J(geom), degree=4)
        return function.eval(function.as_csr(function.derivative(function.derivative(res, 'v'), 'u')))

    fast = run()
    monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')
    slow = run()
    assert numpy.array_equal(fast[1], slow[1]) and numpy.array_equal(fast[2], slow[2])
    assert numpy.abs(fast[0] - slow[0]).max() < RTOL * numpy.abs(slow[0]).max()
    assert numpy.abs(fast[0] - slow[0]).max() > 0  # (different summation orders: the two paths are really different kernels)


@pytest.mark.parametrize('n', [20, 32])
def test_elasticity_entrywise_vs_c_port(n):
    '''20^3 and 32^3 elements (interior lines with all four visits; 32^3: 823 875 dofs, 1.5e8 nonzeros, the C port on all host cores): every CSR entry against the
    C port of the oracle (oracle/c, port.form3d: element loop + stable-sort dedup as in the reference), index arrays bit-exact -- the sizes between the numpy
    oracle's 7^3 and the 64^3 property checks.'''
    from oracle import assemble as oa, port
    if not port.available():
        pytest.skip('oracle/c is not built')
    inp = _inputs((n, n, n), True, seed=3)
    C = oa.elasticity_coefficient(3, 1., .5 / .3 - 1)
    N, dN = oa.tabulate(inp['coeffs'][0], inp['pts'])
    gN, gdN = oa.tabulate(inp['gcoeffs'][0], inp['pts'])
    T = numpy.concatenate([N.T[:, :, None], dN.transpose(1, 0, 2)], axis=2)
    gT = numpy.concatenate([gN.T[:, :, None], gdN.transpose(1, 0, 2)], axis=2)
    vo, rpo, cio, _ = port.form3d((n, n, n), 2, C, T, gT, inp['w'], inp['verts'], threads=16 if n <= 20 else 0)
    from nutils_amd import _lib
    with _lib.trace() as calls:
        got = Dev(inp).fast(C, 3)
    assert 'nh_p2hex_matrix' in calls
    _check(got, (vo, rpo, cio))


@pytest.mark.parametrize('nc', [1, 2, 3])
def test_value_slot_window_vs_oracle(nc):
    '''Forms on the slot window {value, d/dx, d/dy} (no d/dz anywhere): the S0 = 0 instantiations -- in-register operands for one and two components (the value slot rides on
    sqrt(w |J|) itself), the table kernel for three (where the in-register variant would spill).'''
    from oracle import assemble as oa
    inp = _inputs((3, 2, 4), True, seed=4)
    rng = numpy.random.default_rng(9)
    C = numpy.zeros((nc, 4, nc, 4))
    C[:, :3, :, :3] = rng.normal(size=(nc, 3, nc, 3))
    _check(Dev(inp).fast(C, nc), _oracle(inp, C))


def _uniform_inputs(shape, cell):
    inp = _inputs(shape, False)
    inp['verts'] = inp['verts'] * numpy.asarray(cell, dtype=float)
    return inp


@pytest.mark.parametrize('shape,cell,nc', [((1, 1, 1), (1., 1., 1.), 3), ((1, 2, 3), (.5, 1., 2.), 3), ((2, 2, 2), (1., .25, .75), 3), ((3, 4, 5), (.3, .2, .7), 3),
                                           ((6, 5, 7), (1., 1., 1.), 3), ((4, 3, 2), (2., 1., .5), 2), ((3, 3, 3), (1., 2., 3.), 1)])
def test_uniform_cells_vs_oracle(shape, cell, nc):
    '''nh_p2hex_rows_uniform (equidistant vertices: the rows of the 2 x 2 x 2 mesh of the same cells, replicated by node class) against the oracle's element loop over
    EVERY element: index arrays bit-exact, every value written, values 1e-13; meshes with one element along an axis (no even interior node) included.'''
    from oracle import assemble as oa
    from nutils_amd import device, kernels, _lib
    inp = _uniform_inputs(shape, cell)
    rng = numpy.random.default_rng(5)
    C = oa.elasticity_coefficient(3, 1., .5 / .3 - 1) if nc == 3 else rng.normal(size=(nc, 4, nc, 4))
    d = Dev(inp)
    rowptr, colidx = d.pattern.expand(nc, nc, None)
    values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64')
    with _lib.trace() as calls:
        kernels.P2HexUniform(shape=shape, nq=d.nq, weights=d.w, T=d.T, ncomp=nc, C=C, cell=cell)(values)
    assert calls.count('nh_p2hex_rows_uniform') == 1
    _check((device.to_host(values), device.to_host(rowptr), device.to_host(colidx)), _oracle(inp, C))


def test_uniform_cells_slabs():
    '''owner ranges (multi-GPU slabs): the node planes of the owners are written, nothing else'''
    from oracle import assemble as oa
    from nutils_amd import device, kernels
    shape, cell = (5, 3, 4), (1., .5, 2.)
    inp = _uniform_inputs(shape, cell)
    C = oa.elasticity_coefficient(3, 1., .5)
    d = Dev(inp)
    vo, rpo, cio = _oracle(inp, C)
    rowptr, colidx = d.pattern.expand(3, 3, None)
    rp = device.to_host(rowptr)
    nplane = 7 * 9 * 3
    parts = []
    for owners in ((0, 1), (2, 5)):
        values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64')
        kernels.P2HexUniform(shape=shape, nq=d.nq, weights=d.w, T=d.T, ncomp=3, C=C, cell=cell, owners=owners)(values)
        parts.append(device.to_host(values))
    r = rp[4 * nplane]
    assert not numpy.isnan(parts[0][:r]).any() and numpy.isnan(parts[0][r:]).all()
    assert not numpy.isnan(parts[1][r:]).any() and numpy.isnan(parts[1][:r]).all()
    full = numpy.concatenate([parts[0][:r], parts[1][r:]])
    assert numpy.abs(full - vo).max() < RTOL * numpy.abs(vo).max()


def test_uniform_mesh_through_the_api(monkeypatch):
    '''mesh.rectilinear with equidistant vertices: the front end takes nh_p2hex_rows_uniform; the same call with NUTILS_AMD_NO_UNIFORM goes through nh_p2hex_matrix on
    the box geometry, NUTILS_AMD_NO_FAST_PATH through the generic kernels -- all three agree (index arrays exactly).'''
    from nutils_amd import mesh, function, _lib

    def run():
        domain, geom = mesh.rectilinear([numpy.linspace(0, 1, 5), numpy.linspace(0, 2, 4), numpy.linspace(-1, 1, 6)])
        u = domain.field('u', btype='std', degree=2, shape=[3])
        v = domain.field('v', btype='std', degree=2, shape=[3])
        eps = lambda w: function.symgrad(w, geom)
        sigma = 1. * function.div(u, geom) * function.eye(3) + 2 * .6 * eps(u)
        res = domain.integral(function.inner(eps(v), sigma) * function.J(geom), degree=4)
        with _lib.trace() as calls:
            out = function.eval(function.as_csr(function.derivative(function.derivative(res, 'v'), 'u')))
        return out, calls

    uni, calls = run()
    assert 'nh_p2hex_rows_uniform' in calls
    monkeypatch.setenv('NUTILS_AMD_NO_UNIFORM', '1')
    box, calls = run()
    assert 'nh_p2hex_rows_uniform' not in calls and 'nh_p2hex_matrix' in calls
    monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')
    gen, calls = run()
    assert 'nh_p2hex_matrix' not in calls
    for other in (box, gen):
        assert numpy.array_equal(uni[1], other[1]) and numpy.array_equal(uni[2], other[2])
        assert numpy.abs(uni[0] - other[0]).max() < RTOL * numpy.abs(other[0]).max()
