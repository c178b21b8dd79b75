'''CPU tests of the host layer (no GPU, no oracle compute): table producers against the golden reference tables, the form
algebra, the matrix hand-off checks (error behaviour of /root/reference/src/nutils/matrix/__init__.py:30-151), and the
"no CPU fallback" rule.'''
import numpy
import pytest


@pytest.mark.parametrize('name,btype', [('lap1d_p1_5', 'std'), ('lap2d_spline2_4x4', 'spline'), ('lap3d_p1_234', 'std'), ('lap3d_p2_2_iso', 'std'),
                                        ('lap3d_spline3_3', 'spline'), ('lap2d_p2_3x4_iso', 'std'), ('lap1d_spline3_6_per0', 'spline'),
                                        ('lap2d_spline2_5x4_per0', 'spline'), ('lap2d_p2_4x3_per1', 'std'), ('lap3d_p1_345_per02', 'std')])
def test_structured_basis_tables(golden, name, btype):
    '''Basis.get_dofs / get_coefficients (function.py:2794-2837) of the product's table producer == the real reference's.'''
    from nutils_amd import mesh, points
    g = golden(name)
    shape = [int(n) for n in g['shape']]
    degree = int(g['degree'])
    domain, geom = mesh.rectilinear(shape, periodic=tuple(int(i) for i in g['periodic']) if 'periodic' in g else ())
    basis = domain.basis(btype, degree=degree)
    nb = len(g['dofs']) // len(domain)
    dofs, coeffs = g['dofs'].reshape(len(domain), nb), g['coeffs'].reshape(len(domain), nb, -1)
    for e in range(len(domain)):
        assert numpy.array_equal(basis.get_dofs(e), dofs[e])
        assert numpy.abs(basis.get_coefficients(e) - coeffs[e]).max() < 1e-14
    cls = basis.element_classes()
    assert [basis.class_of(e) for e in range(len(domain))] == list(cls)
    pts = points.gauss(2 * degree, len(shape))
    assert numpy.abs(pts.coords - g['gauss_coords']).max() < 1e-15 and numpy.abs(pts.weights - g['gauss_weights']).max() < 1e-15


def test_poly_layout():
    '''coefficient order of evaluable.py:4331-4340: 2 variables, degree 2: x1^2, x0 x1, x1, x0^2, x0, 1'''
    from nutils_amd import poly
    order = [(0, 2), (1, 1), (0, 1), (2, 0), (1, 0), (0, 0)]
    assert [poly.index(p, 2) for p in order] == list(range(6))
    assert poly.ncoeffs(3, 3) == 20 and poly.degree(3, 84) == 6
    with pytest.raises(ValueError):
        poly.degree(2, 7)
    t = poly.tensor_product([numpy.array([[-1., 1.], [1., 0.]])] * 2)  # (1-x0)(1-x1), (1-x0) x1, x0 (1-x1), x0 x1
    assert t.shape == (4, 6)
    assert numpy.array_equal(t[3], [0, 1, 0, 0, 0, 0]) and numpy.array_equal(t[0], [0, 1, -1, 0, -1, 1])


def test_poly_change_degree_keeps_the_polynomial():
    '''poly.change_degree (bases of triangles beside squares written at one degree, seam.Emitter.basis) against the oracle's restatement of nutils_poly.change_degree
    (pinned by the reference's own poly tests) and by evaluation at random points'''
    from oracle import poly as opoly
    from nutils_amd import poly
    rng = numpy.random.default_rng(3)
    for nv in (1, 2, 3):
        x = rng.uniform(size=(5, nv))
        for old in range(4):
            c = rng.normal(size=(3, poly.ncoeffs(nv, old)))
            for new in range(old, 5):
                up = poly.change_degree(c, nv, new)
                assert numpy.array_equal(up, opoly.change_degree(c, nv, new))
                for k in range(3):
                    assert numpy.allclose(opoly.eval_outer(up[k], x), opoly.eval_outer(c[k], x), rtol=1e-14, atol=1e-14)
    with pytest.raises(ValueError):
        poly.change_degree(numpy.ones(6), 2, 1)


def test_form_algebra():
    from nutils_amd import mesh, function
    domain, geom = mesh.rectilinear([2, 2, 2])
    u = domain.field('u', btype='std', degree=1, shape=[3])
    v = domain.field('v', btype='std', degree=1, shape=[3])
    lam, mu = 1.5, .7
    sigma = lam * function.div(u, geom) * function.eye(3) + 2 * mu * function.symgrad(u, geom)
    res = domain.integral(function.inner(function.symgrad(v, geom), sigma) * function.J(geom), degree=2)
    jac = function.derivative(function.derivative(res, 'v'), 'u')
    C = jac.terms[0][1].B
    ref = numpy.zeros((3, 4, 3, 4))
    for c in range(3):
        for i in range(3):
            for d in range(3):
                for j in range(3):
                    ref[c, 1 + i, d, 1 + j] = lam * (c == i) * (d == j) + mu * ((c == d) * (i == j) + (c == j) * (d == i))
    assert numpy.abs(C - ref).max() < 1e-15
    assert jac.shape == ()  # both dof axes belong to named fields
    # quadratic functional: d2/du2 of 1/2 eps:sigma gives the same tensor
    E = domain.integral(.5 * function.inner(function.symgrad(u, geom), sigma) * function.J(geom), degree=2)
    H = function.derivative(function.derivative(E, 'u'), 'u')
    assert numpy.abs(H.terms[0][1].B - ref).max() < 1e-15
    # unsupported expressions are refused, never evaluated on the CPU
    with pytest.raises(NotImplementedError):
        function.grad(function.grad(u, geom), geom)
    basis = domain.basis('std', degree=1)
    K = function.outer(function.grad(basis, geom)).sum(-1)
    assert K.shape == (27, 27)
    with pytest.raises(NotImplementedError):
        domain.sample('gauss', 2).integral(function.outer(function.grad(basis, geom)) * function.J(geom))  # unreduced axis


def test_fieldpoly_calculus():
    from nutils_amd import mesh, function
    domain, geom = mesh.unitsquare(2)
    phi = function.value(domain.field('φ', btype='std', degree=2))
    phi0 = function.value(domain.field('φ0', btype='std', degree=2))
    psi = .25 * (phi ** 2 - 1) ** 2
    assert psi.terms == {(4,): .25, (2,): -.5, (0,): .25}
    assert psi.derivative('φ').terms == {(3,): 1., (1,): -1.}
    d = (phi - phi0) ** 2
    assert d.derivative('φ0').terms == {(1, 0): -2., (0, 1): 2.}
    I = domain.integral(psi * function.J(geom), degree=8)
    r = function.derivative(I, 'φ')
    J = function.derivative(r, 'φ')
    assert r.terms[0][1].rows and r.terms[0][1].fscale.terms == {(3,): 1., (1,): -1.}
    assert J.terms[0][1].cols and J.terms[0][1].fscale.terms == {(2,): 3., (0,): -1.}
    with pytest.raises(NotImplementedError):
        function.derivative(J, 'φ')  # rank-3 tensor


def test_matrix_handoff_errors():
    '''assemble_csr validation (matrix/__init__.py:46-69) and block merge (:103-151).'''
    from nutils_amd import matrix
    v = numpy.array([1., 2., 3.])
    ok = matrix.assemble_csr(v, numpy.array([0, 2, 3]), numpy.array([0, 1, 1]), 2)
    assert numpy.array_equal(ok.export('dense'), [[1, 2], [0, 3]])
    for bad in ((v.reshape(3, 1), [0, 2, 3], [0, 1, 1]), (v, [1, 2, 3], [0, 1, 1]), (v, [0, 3, 2], [0, 1, 1]), (v, [0, 2, 3], [0, 2, 1]),
                (v, [0, 2, 3], [1, 0, 1]), (v, [0., 2., 3.], [0, 1, 1])):
        with pytest.raises(matrix.MatrixError):
            matrix.assemble_csr(numpy.asarray(bad[0]), numpy.asarray(bad[1]), numpy.asarray(bad[2]), 2)
    with pytest.raises(ValueError):
        with matrix.backend(object()):
            pass
    a = (numpy.array([1., 2.]), numpy.array([0, 1, 2]), numpy.array([0, 1]), 2)
    b = (numpy.array([5.]), numpy.array([0, 0, 1]), numpy.array([0]), 1)
    m = matrix.assemble_block_csr([[a, b], [b[:3] + (2,), (numpy.array([7., 8.]), numpy.array([0, 1, 2]), numpy.array([0, 0]), 1)]])
    assert numpy.array_equal(m.export('dense'), [[1, 0, 0], [0, 2, 5], [0, 0, 7], [5, 0, 8]])
    assert numpy.array_equal(matrix.compress_indices([0, 0, 2], 4), [0, 2, 2, 3, 3])
    with pytest.raises(ValueError):
        matrix.compress_indices([1, 0], 2)


def test_no_cpu_fallback():
    '''Without a GPU every evaluation raises NutilsHipError: the product path never computes on the CPU.'''
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from nutils_amd import mesh, function, _lib
    domain, geom = mesh.rectilinear([2, 2])
    basis = domain.basis('std', degree=1)
    K = domain.integral(function.outer(function.grad(basis, geom)).sum(-1) * function.J(geom), degree=2)
    with pytest.raises(_lib.NutilsHipError):
        function.eval(function.as_csr(K))


def test_reassemble_csr_matches_assemble_csr():
    '''Trusted re-assembly (values only) gives the same matrix as the validating entry point, and still rejects a wrong length.'''
    from nutils_amd import matrix
    rowptr = numpy.array([0, 2, 3, 5], dtype=numpy.int64)
    colidx = numpy.array([0, 2, 1, 0, 2], dtype=numpy.int64)
    vals = numpy.array([4., 1., 3., 1., 5.])
    a = matrix.assemble_csr(vals, rowptr, colidx, 3)
    b = matrix.reassemble_csr(vals * 2, rowptr, colidx, 3)
    assert numpy.array_equal(b.export('dense'), 2 * a.export('dense'))
    x = numpy.array([1., 2., 3.])
    assert numpy.allclose(b @ x, 2 * (a @ x))
    assert numpy.allclose(b.solve(numpy.ones(3), constrain=numpy.array([numpy.nan, numpy.nan, 1.])),
                          matrix.assemble_csr(vals * 2, rowptr, colidx, 3).solve(numpy.ones(3), constrain=numpy.array([numpy.nan, numpy.nan, 1.])))
    with pytest.raises(matrix.MatrixError):
        matrix.reassemble_csr(numpy.arange(4.), rowptr, colidx, 3)
    calls = []

    class Fake:  # backends without the trusted entry point still work (fake-backend precedent, reference tests/test_matrix.py:6-23)
        @staticmethod
        def assemble(values, rowptr, colidx, ncols):
            calls.append(len(values))
            return 'fake'
    with matrix.backend(Fake):
        assert matrix.reassemble_csr(numpy.arange(5.), rowptr, colidx, 3) == 'fake' and calls == [5]
