'''Owner kernel for vector-valued blocks (NH_MATRIX_FUSED, nh_owner.hip) at the C-ABI level: against the reference's golden CSR, against the deterministic
two-pass gather on larger meshes with any numbering, for closed-form and dense form tensors, masked component blocks, scale arrays; repeated assemblies
bit-identical.  Replaces numeric.accumulate / numpy.add.at (numeric.py:434-460, evaluable.py:3405-3411) for these blocks.'''
import numpy
import pytest
from test_gpu_kernels import Case, close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['elast2d_p1_3x3', 'elast2d_p2_3x2_iso', 'elast3d_p1_2_iso'])
def test_owner_rows_equal_the_reference(golden, name):
    from nutils_amd import device, kernels, _lib
    from oracle import assemble as oa
    g = golden(name)
    c = Case(g)
    C = oa.elasticity_coefficient(c.nd, float(g['lam']), float(g['mu']))
    mask = oa.block_mask(C)
    rowptr, colidx = c.pattern.expand(c.nd, c.nd, mask)
    out = []
    for it in range(2):
        values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64')
        with _lib.trace() as calls:
            kernels.assemble_matrix(nelems=c.nelems, ndims=c.nd, nq=c.nq, weights=c.weights, geom=c.geom, test=c.basis, trial=c.basis, nct=c.nd, ncr=c.nd, C=C, mask=mask,
                                    pattern=c.pattern, values=values, fused=True, store=True)
        out.append(device.to_host(values))
    assert c.pattern.owner_info()[0] > 0  # (the row tasks were built: the launch did not fall back)
    assert numpy.array_equal(device.to_host(rowptr), g['K_rowptr']) and numpy.array_equal(device.to_host(colidx), g['K_colidx'])
    close(out[0], g['K_values'])
    assert numpy.array_equal(out[0], out[1])


def _mesh(nd, n, degree, shuffle, seed=5):
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    rng = numpy.random.default_rng(seed)
    shape = (n,) * nd
    dofs, coeffs, ndofs = oa.structured_basis(shape, 'std', degree)
    nb = (degree + 1) ** nd
    gdofs, gcoeffs, nverts = oa.structured_basis(shape, 'std', 1)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * nd, indexing='ij'), -1).reshape(-1, nd) + rng.uniform(-.2, .2, (nverts, nd))
    dofs = numpy.asarray(dofs).reshape(-1, nb)
    gdofs = numpy.asarray(gdofs).reshape(-1, 2 ** nd)
    if shuffle:
        perm = rng.permutation(ndofs)          # new number of old dof i
        eperm = rng.permutation(len(dofs))
        dofs, gdofs = perm[dofs][eperm], gdofs[eperm]
    ne = len(dofs)
    pts, w = oa.gauss(2 * degree, nd)
    p = device.to_dev(pts, 'float64')
    T = kernels.tabulate(device.to_dev(coeffs[0], 'float64'), nb, coeffs.shape[2], p, len(pts), nd)
    gT = kernels.tabulate(device.to_dev(gcoeffs[0], 'float64'), 2 ** nd, gcoeffs.shape[2], p, len(pts), nd)
    d = device.to_dev(dofs.ravel(), 'int32')
    basis = kernels.basis(T, d, nb=nb)
    geom = kernels.geometry_iso(2 ** nd, gT, device.to_dev(gdofs.ravel(), 'int32'), device.to_dev(verts, 'float64'))
    pattern = kernels.Pattern(ne, ndofs, ndofs, d, d, nbt=nb, nbr=nb)
    return dict(nelems=ne, ndims=nd, nq=len(pts), weights=device.to_dev(w, 'float64'), geom=geom, test=basis, trial=basis, pattern=pattern), ndofs, rng


@pytest.mark.parametrize('nd,n,degree,shuffle', [(3, 14, 1, False), (3, 14, 1, True), (2, 40, 1, True), (2, 24, 2, False), (2, 24, 2, True), (3, 6, 2, False), (3, 5, 2, True)])
@pytest.mark.parametrize('form', ['elasticity', 'dense', 'masked'])
def test_owner_rows_many_blocks_any_numbering(nd, n, degree, shuffle, form):
    '''Meshes of many row blocks with a perturbed geometry, natural or random numbering of elements and dofs: the one-pass owner kernel against the two-pass gather
    (the reference's order of the sums) -- isotropic elasticity (closed form), a dense random form tensor with value slots and a scale array, and a form whose
    off-diagonal component blocks are absent from the pattern.'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    common, ndofs, rng = _mesh(nd, n, degree, shuffle)
    S, scale, mask = 1 + nd, None, None
    if form == 'elasticity':
        C = oa.elasticity_coefficient(nd, 1.3, .7)
    elif form == 'dense':
        C = rng.normal(size=(nd, S, nd, S))
        scale = device.to_dev(rng.uniform(-.5, 1.5, common['nelems'] * common['nq']), 'float64')
    else:
        C = numpy.zeros((nd, S, nd, S))
        for c in range(nd):
            C[c, 1:, c, 1:] = numpy.eye(nd) * (1. + c)
            C[c, 0, c, 0] = .3
        mask = oa.block_mask(C)
        assert mask.sum() == nd
    pattern = common['pattern']
    rowptr, colidx = pattern.expand(nd, nd, mask)
    out = []
    for kw in (dict(gather=True), dict(fused=True, store=True), dict(fused=True, store=True), dict(fused=True)):
        values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64') if kw.get('store') else device.zeros(colidx.numel(), 'float64')
        kernels.assemble_matrix(nct=nd, ncr=nd, C=C, mask=mask, values=values, scale=scale, **common, **kw)
        out.append(device.to_host(values))
    nblocks, rpb, nvisits, nchunks = pattern.owner_info()
    assert nblocks >= -(-ndofs // rpb) and nblocks > 20, (nblocks, rpb)  # (Morton boxes of at most rpb rows; triquadratic elements: points in chunks)
    assert nchunks * 64 >= common['nelems'] * common['test'].nb ** 2
    for o in out[1:]:
        close(o, out[0])
    assert numpy.array_equal(out[1], out[2])
    assert numpy.array_equal(out[1], out[3])  # (accumulating into zeros = storing)


@pytest.mark.parametrize('qdeg', [2, 4])  # (4: 27 points -- the D tables of a block exceed the LDS, the launch falls back inside the C entry)
def test_owner_rows_are_the_default_for_vector_blocks(qdeg, monkeypatch):
    '''Through the front end: trilinear elasticity on the any-mesh path takes the owner kernel from the first assembly on, bit-identical from run to run,
    equal to the gather path (NUTILS_AMD_NO_FUSED=1).'''
    from nutils_amd import mesh, function, _lib
    monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')
    rng = numpy.random.default_rng(3)
    n = 12
    domain, geom0 = mesh.rectilinear([n] * 3)
    gb = domain.basis('std', degree=1)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (len(gb), 3))
    X = gb @ verts
    u = domain.field('u', btype='std', degree=1, shape=[3])
    v = domain.field('v', btype='std', degree=1, shape=[3])
    eps = lambda w: function.symgrad(w, X)
    sigma = function.div(u, X) * function.eye(3) + 1.3 * eps(u)
    K = function.derivative(function.derivative(domain.integral(function.inner(eps(v), sigma) * function.J(X), degree=qdeg), 'v'), 'u')
    with _lib.trace() as calls:
        v0, rp, ci = function.eval(function.as_csr(K))
    v1, _, _ = function.eval(function.as_csr(K))
    assert qdeg != 2 or numpy.array_equal(v0, v1)
    monkeypatch.setenv('NUTILS_AMD_NO_FUSED', '1')
    K2 = function.derivative(function.derivative(domain.integral(function.inner(eps(v), sigma) * function.J(X), degree=qdeg), 'v'), 'u')
    function.eval(function.as_csr(K2))
    w1, rp2, ci2 = function.eval(function.as_csr(K2))  # (second assembly of the pattern: gather)
    assert numpy.array_equal(rp, rp2) and numpy.array_equal(ci, ci2)
    close(v0, w1)


def test_owner_plan_serves_other_forms():
    '''One pattern, the plan built for the closed-form elasticity tensor of its first launch: a later launch with a dense form that reads the value slot (wider D tables: more
    LDS per block than the plan budgeted, one workgroup per CU or the caller's other path) and a return to the first form still give the gather path's matrix.'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    common, ndofs, rng = _mesh(3, 10, 1, True)
    pattern, nd = common['pattern'], 3
    rowptr, colidx = pattern.expand(nd, nd, None)
    C1 = oa.elasticity_coefficient(3, 1.3, .7)
    C2 = rng.normal(size=(3, 4, 3, 4))
    for C in (C1, C2, C1):
        out = []
        for kw in (dict(gather=True), dict(fused=True, store=True)):
            values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64') if kw.get('store') else device.zeros(colidx.numel(), 'float64')
            kernels.assemble_matrix(nct=nd, ncr=nd, C=C, mask=None, values=values, **common, **kw)
            out.append(device.to_host(values))
        close(out[1], out[0])
    assert pattern.owner_info()[0] > 0


def test_owner_plan_follows_the_connectivity_of_the_call():
    '''The plan keeps the vertex numbers of the visiting elements (one level of dependent loads less per block), made from the connectivity array of the call that
    needed them first: a later call with ANOTHER connectivity array for the same pattern (the vertices renumbered) rebuilds them; back to the first array as well.'''
    from nutils_amd import device, kernels
    from oracle import assemble as oa
    common, ndofs, rng = _mesh(3, 10, 1, True)
    pattern, nd, geom = common['pattern'], 3, common['geom']
    rowptr, colidx = pattern.expand(nd, nd, None)
    C = oa.elasticity_coefficient(3, 1.3, .7)
    gT, gdofs, verts = geom._keep
    gdofs, verts = device.to_host(gdofs), device.to_host(verts).reshape(-1, 3)
    perm = rng.permutation(len(verts))  # new number of old vertex i
    verts2 = numpy.empty_like(verts)
    verts2[perm] = verts
    geom2 = kernels.geometry_iso(8, gT, device.to_dev(perm[gdofs].ravel(), 'int32'), device.to_dev(verts2, 'float64'))
    ref = None
    for g in (geom, geom2, geom):
        kw = dict(common, geom=g)
        out = []
        for mode in (dict(gather=True), dict(fused=True, store=True)):
            values = device.to_dev(numpy.full(colidx.numel(), numpy.nan), 'float64') if mode.get('store') else device.zeros(colidx.numel(), 'float64')
            kernels.assemble_matrix(nct=nd, ncr=nd, C=C, mask=None, values=values, **kw, **mode)
            out.append(device.to_host(values))
        close(out[1], out[0])
        if ref is None:
            ref = out[1]
        assert numpy.array_equal(out[1], ref)  # (the same elements with the same coordinates: the same sums in the same order)
    assert pattern.owner_info()[0] > 0
