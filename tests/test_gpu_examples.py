'''BASELINE.json configs[0]: examples/laplace.py semantics (/root/reference/examples/laplace.py:50-105) end to end
through the accelerated path -- volume stiffness, Neumann boundary load with a coefficient function, Dirichlet
constraints by boundary projection (System.solve_constraints), linear solve (System.solve), L2 error -- compared
with the example's own return values (cons, lhs, err) captured from the real reference (tests/golden/examples_laplace.npz).'''
import numpy
import pytest

pytestmark = pytest.mark.gpu


def close(a, b, factor=1.):
    '''max |a - b| <= factor * 1e-13 * max |b| (the fp64 tolerance of the parity tests).'''
    a, b = numpy.asarray(a), numpy.asarray(b)
    assert a.shape == b.shape
    err = numpy.abs(a - b).max()
    assert err <= factor * 1e-13 * numpy.abs(b).max(), err / numpy.abs(b).max()


def laplace(nelems, btype, degree):
    from nutils_amd import mesh, function
    from nutils_amd.solver import System
    domain, geom = mesh.unitsquare(nelems, 'square')
    u = domain.field('u', btype=btype, degree=degree)
    v = domain.field('v', btype=btype, degree=degree)
    dV = function.J(geom)
    grad = lambda w: function.grad(w, geom)
    res = domain.integral((grad(v) * grad(u)).sum(-1) * dV, degree=degree * 2)
    flux = function.PointFunc(lambda x: numpy.cos(1) * numpy.cosh(x[:, 1]), geom)
    res -= domain.boundary['right'].integral(v * flux * dV, degree=degree * 2)
    g = function.PointFunc(lambda x: numpy.cosh(1) * numpy.sin(x[:, 0]), geom)
    sqr = domain.boundary['left'].integral(u * u * dV, degree=degree * 2)
    top = domain.boundary['top']
    sqr += top.integral(u * u * dV, degree=degree * 2) - 2 * top.integral(u * g * dV, degree=degree * 2) + top.integral(g * g * dV, degree=degree * 2)
    cons = System(sqr, trial='u').solve_constraints(droptol=1e-15)
    args = System(res, trial='u', test='v').solve(constrain=cons)
    uex = function.PointFunc(lambda x: numpy.sin(x[:, 0]) * numpy.cosh(x[:, 1]), geom)
    err2 = function.eval(domain.integral(u * u * dV, degree=degree * 2) - 2 * domain.integral(u * uex * dV, degree=degree * 2)
                         + domain.integral(uex * uex * dV, degree=degree * 2), args)
    return cons['u'], args['u'], err2 ** .5


@pytest.mark.parametrize('tag,nelems,btype,degree', [('default', 4, 'std', 1), ('spline', 4, 'spline', 2), ('c1', 32, 'std', 1)])
def test_laplace_example(golden, tag, nelems, btype, degree):
    g = golden('examples_laplace')
    cons, lhs, err = laplace(nelems, btype, degree)
    gc, gl, ge = g[f'laplace_{tag}_cons'], g[f'laplace_{tag}_lhs'], float(g[f'laplace_{tag}_err'])
    assert numpy.array_equal(numpy.isnan(cons), numpy.isnan(gc))
    assert numpy.nanmax(numpy.abs(cons - gc)) < 1e-12
    assert numpy.abs(lhs - gl).max() < 1e-10
    assert abs(err - ge) < 1e-9 * max(1, ge) + 1e-12


def cahnhilliard(g):
    '''examples/cahnhilliard.py:163-184 (unit-free): free energy with double-well potential, stabilisation term, wall energy.'''
    from nutils_amd import mesh, function
    size, eps, M, stens, wn, wp, dt = g['params']
    nelems, degree = int(g['nelems']), int(g['degree'])
    domain, geom = mesh.rectilinear([numpy.linspace(0, size, nelems + 1)] * 2)
    phi = domain.field('φ', btype='std', degree=degree)
    phi0 = domain.field('φ0', btype='std', degree=degree)
    eta = domain.field('η', btype='std', degree=degree) * (stens / eps)
    p, p0 = function.value(phi), function.value(phi0)
    dp = p - p0
    psi = .25 * (p ** 2 - 1) ** 2
    dpsi = .25 * dp ** 2 * (1 - p ** 2 + 2 * p * dp / 3 - dp ** 2 / 6)
    dV = function.J(geom)
    grad = lambda w: function.grad(w, geom)
    d4, d2 = degree * 4, degree * 2
    nrg = domain.integral((psi + dpsi) * (stens / eps) * dV, degree=d4) \
        + domain.integral(.5 * stens * eps * (grad(phi) * grad(phi)).sum(-1) * dV, degree=d4) \
        - domain.integral(eta * phi * dV, degree=d4) + domain.integral(eta * phi0 * dV, degree=d4) \
        - domain.integral(.5 * dt * M * (grad(eta) * grad(eta)).sum(-1) * dV, degree=d4) \
        + domain.boundary.integral((wp + wn) / 2 * dV, degree=d2) + domain.boundary.integral((wp - wn) / 2 * phi * dV, degree=d2)
    return domain, nrg


def test_cahnhilliard_residual_jacobian(golden):
    '''BASELINE.json configs[3]: nonlinear residual + Jacobian (re)assembly with field-dependent coefficient functions,
    against the reference's residual/Jacobian blocks at a random state, and one implicit time step (Newton).'''
    from nutils_amd import function
    from nutils_amd.solver import System
    g = golden('cahnhilliard_p2_4')
    domain, nrg = cahnhilliard(g)
    args = {'φ': g['arg_φ'], 'φ0': g['arg_φ0'], 'η': g['arg_η']}
    tol = lambda ref: 1e-13 * numpy.abs(ref).max()
    assert abs(function.eval(nrg, args) - float(g['energy'])) < 1e-13 * abs(float(g['energy']))
    system = System(nrg, trial='φ,η')
    assert not system.is_linear
    res = system.assemble_residual(args)
    n = len(g['arg_φ'])
    assert numpy.abs(res[:n] - g['res_φ']).max() < tol(g['res_φ'])
    assert numpy.abs(res[n:] - g['res_η']).max() < tol(g['res_η'])
    jac = system.assemble_jacobian(args).export('dense')
    import scipy.sparse
    for i, a in enumerate('φη'):
        for j, b in enumerate('φη'):
            ref = scipy.sparse.csr_matrix((g[f'jac_{a}{b}_values'], g[f'jac_{a}{b}_colidx'], g[f'jac_{a}{b}_rowptr']), (n, n)).toarray()
            assert numpy.abs(jac[i * n:(i + 1) * n, j * n:(j + 1) * n] - ref).max() < 1e-12 * max(numpy.abs(ref).max(), 1.), (a, b)
    # bit-exact pattern of a single-sample block: volume part of d2/dphi2
    vol = function.Integral([t for t in function.derivative(function.derivative(nrg, 'φ'), 'φ').terms if t[0].elist is None])
    v, rp, ci = function.eval(function.as_csr(vol), args)
    assert numpy.array_equal(rp, g['jac_φφ_rowptr']) and numpy.array_equal(ci, g['jac_φφ_colidx'])
    # one implicit step from (phi0, eta = 0): Newton on the GPU-assembled system
    sol = system.solve(arguments={'φ': g['arg_φ0'], 'φ0': g['arg_φ0'], 'η': numpy.zeros(n)}, tol=1e-8)
    assert numpy.abs(sol['φ'] - g['step_φ']).max() < 1e-8
    assert numpy.abs(sol['η'] - g['step_η']).max() < 1e-8


def test_nurbs_plate_with_hole(golden):
    '''BASELINE.json configs[4] ingredient -- NURBS mode of examples/platewithhole.py:66-86,126-153: rational basis
    bspline_i w_i / W on the refined structured topology, NURBS geometry map, plane-strain elasticity stiffness + residual.'''
    from nutils_amd import mesh, function, basis as _basis
    g = golden('nurbs_plate_r2')
    shape = [int(n) for n in g['shape']]
    domain, _ = mesh.rectilinear(shape)
    bspline = domain.basis('spline', degree=2)
    assert numpy.array_equal(numpy.concatenate([bspline.get_dofs(e) for e in range(len(domain))]), g['dofs'])
    h = 1. / 2 ** int(g['nrefine'])  # element size in the parameter domain of the coarse 1 x 2 patch
    nurbs = _basis.RationalBasis(bspline, g['weights'], W=g['W'], dW=g['dW_dparam'] * h)
    geom = function.TabulatedGeometry(g['x'], g['dx_dparam'] * h)
    smp = domain.sample('gauss', 10)
    assert numpy.abs(smp.points.coords - g['gauss_coords']).max() < 1e-15
    u = function.field('u', nurbs, shape=[2])
    v = function.field('v', nurbs, shape=[2])
    lam, mu = float(g['lam']), float(g['mu'])
    sigma = lam * function.div(u, geom) * function.eye(2) + 2 * mu * function.symgrad(u, geom)
    res = smp.integral(function.inner(function.grad(v, geom), sigma) * function.J(geom))
    values, rowptr, colidx = function.eval(function.as_csr(function.derivative(function.derivative(res, 'v'), 'u')))
    assert numpy.array_equal(rowptr, g['K_rowptr']) and numpy.array_equal(colidx, g['K_colidx'])
    assert numpy.abs(values - g['K_values']).max() < 1e-13 * numpy.abs(g['K_values']).max()
    r = function.eval(function.derivative(res, 'v'), u=g['u'])
    assert numpy.abs(r - g['res']).max() < 1e-13 * numpy.abs(g['res']).max()
    assert abs(smp.integrate(function.J(geom)) - float(g['area'])) < 1e-13
    # same-level NURBS (W = sum_j w_j B_j evaluated by the kernel): partition of unity and zero gradient sum
    nurbs2 = _basis.RationalBasis(bspline, g['weights'])
    one = numpy.ones(len(bspline))
    w = function.field('w', nurbs2)
    val, grad = smp.eval([w, function.grad(w, geom)], w=one)
    assert numpy.abs(val - 1).max() < 1e-14 and numpy.abs(grad).max() < 1e-12


def test_iga_plate_p3_ten_levels_one_workload(golden):
    '''BASELINE.json configs[4] as ONE workload (tests/golden/iga_plate_p3_l10.npz from the real reference): NURBS plate-with-hole
    geometry (examples/platewithhole.py:66-86), 10 hierarchical refinement levels towards the hole (refined_by, examples/adaptivity.py:
    58-70), p = 3 truncated hierarchical splines made rational with projected weights -- 16..24 functions per element, element sizes
    over a factor 2^9, rational, tabulated geometry, vector-valued -- plane-strain elasticity stiffness matrix, residual, area.'''
    from nutils_amd import function, topology, basis as _basis
    g = golden('iga_plate_p3_l10')
    topo = topology.ElementList(g['elem_origin'], g['elem_size'])
    smp = topo.sample('gauss', 8)
    assert numpy.abs(smp.points.coords - g['gauss_coords']).max() < 1e-15
    off = g['dof_offsets']
    nb = numpy.diff(off)
    assert nb.min() == 16 and nb.max() > 16 and int(g['levels']) == 10 and int(g['degree']) == 3
    hb = topo.plain_basis([g['coeffs'][a:b] for a, b in zip(off, off[1:])], [g['dofs'][a:b] for a, b in zip(off, off[1:])], int(g['ndofs']))
    size = g['elem_size']  # element coordinates xi in [0,1]^2: param = origin + size * xi
    nurbs = _basis.RationalBasis(hb, g['weights'], W=g['W'], dW=g['dW_dparam'] * size[:, None, :])
    geom = function.TabulatedGeometry(g['x'], g['dx_dparam'] * size[:, None, None, :])
    u = function.field('u', nurbs, shape=[2])
    v = function.field('v', nurbs, shape=[2])
    lam, mu = float(g['lam']), float(g['mu'])
    sigma = lam * function.div(u, geom) * function.eye(2) + 2 * mu * function.symgrad(u, geom)
    res = smp.integral(function.inner(function.grad(v, geom), sigma) * function.J(geom))
    values, rowptr, colidx = function.eval(function.as_csr(function.derivative(function.derivative(res, 'v'), 'u')))
    assert numpy.array_equal(rowptr, g['K_rowptr']) and numpy.array_equal(colidx, g['K_colidx'])
    assert numpy.abs(values - g['K_values']).max() < 1e-13 * numpy.abs(g['K_values']).max()
    r = function.eval(function.derivative(res, 'v'), u=g['u'])
    assert numpy.abs(r - g['res']).max() < 1e-13 * numpy.abs(g['res']).max()
    assert abs(smp.integrate(function.J(geom)) - float(g['area'])) < 1e-13


def test_nonlinear_diffusion_picard_p1hex(monkeypatch):
    '''Quasi-linear diffusion -div((1 + u^2) grad u) = 1 on a perturbed P1 hex mesh, u = 0 on x = 0, by fixed-point iteration
    K(u_k) u_{k+1} = f.  Every step re-assembles the stiffness matrix with the field-dependent coefficient and evaluates the residual
    K(u) u - f: both take the write-once structured kernels (nh_p1hex_laplace with qscale_dev, nh_p1hex_apply); the converged
    solution must equal the all-generic run.'''
    from nutils_amd import mesh, function, kernels, matrix
    n = 8
    rng = numpy.random.default_rng(11)
    domain, geom = mesh.rectilinear([n] * 3)
    basis = domain.basis('std', degree=1)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) / n + rng.uniform(-.02, .02, ((n + 1) ** 3, 3))
    X = basis @ verts
    u = domain.field('u', btype='std', degree=1)
    dV = function.J(X)
    kappa = 1 + function.value(u) ** 2
    K = domain.integral(kappa * function.outer(function.grad(basis, X)).sum(-1) * dV, degree=2)
    r = domain.integral(kappa * (function.grad(basis, X) * function.grad(u, X)).sum(-1) * dV, degree=2)
    f = function.eval(domain.integral(basis * dV, degree=2))
    cons = numpy.full((n + 1) ** 3, numpy.nan)
    cons.reshape(n + 1, n + 1, n + 1)[0] = 0.
    free = numpy.isnan(cons)
    calls = {'laplace': 0, 'apply': 0}
    for name in ('p1hex_laplace', 'p1hex_apply'):
        orig = getattr(kernels, name)
        monkeypatch.setattr(kernels, name, lambda _o=orig, _n=name.split('_')[1], **kw: (calls.__setitem__(_n, calls[_n] + 1), _o(**kw))[1])
    sol = {}
    for mode in ('generic', 'fast'):
        if mode == 'generic':
            monkeypatch.setenv('NUTILS_AMD_NO_FAST_PATH', '1')
        else:
            monkeypatch.delenv('NUTILS_AMD_NO_FAST_PATH')
        x = numpy.zeros((n + 1) ** 3)
        for it in range(50):
            resid = function.eval(r, arguments={'u': x}) - f
            if numpy.linalg.norm(resid[free]) < 1e-11:
                break
            x = matrix.assemble_csr(*function.eval(function.as_csr(K), arguments={'u': x}), len(x)).solve(f, constrain=cons)
        else:
            raise AssertionError('fixed-point iteration did not converge')
        sol[mode] = x
        assert (calls['laplace'] > 0 and calls['apply'] > 0) == (mode == 'fast')
    assert numpy.abs(sol['fast']).max() > .1
    assert numpy.abs(sol['fast'] - sol['generic']).max() <= 1e-9 * numpy.abs(sol['generic']).max()


@pytest.mark.parametrize('ndims,btype,degree', [(3, 'std', 1), (2, 'spline', 2)])
def test_quasilinear_newton(ndims, btype, degree, monkeypatch):
    '''Newton for -div((1 + u^2) grad u) = 1, u = 0 on x = 0, through System: the Jacobian contains the product-rule term
    2 u phi_n grad u . grad phi_m, a bilinear form whose coefficients depend on the point (cq_dev); checked against central
    differences of the residual, for symmetry breaking, and by quadratic convergence to the fixed-point solution.'''
    from nutils_amd import mesh, function
    from nutils_amd.solver import System
    n = 6 if ndims == 3 else 10
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1, n + 1)] * ndims)
    u = domain.field('u', btype=btype, degree=degree)
    v = domain.field('v', btype=btype, degree=degree)
    dV = function.J(geom)
    res = domain.integral((1 + function.value(u) ** 2) * (function.grad(v, geom) * function.grad(u, geom)).sum(-1) * dV, degree=2 * degree + 2) \
        - domain.integral(v * dV, degree=2 * degree)
    system = System(res, trial='u', test='v')
    assert not system.is_linear
    nd = system.size
    rng = numpy.random.default_rng(3)
    x0, d = rng.normal(0, .5, nd), rng.normal(0, 1, nd)
    jac = system.assemble_jacobian({'u': x0})
    h = 1e-5
    fd = (system.assemble_residual({'u': x0 + h * d}) - system.assemble_residual({'u': x0 - h * d})) / (2 * h)
    Jd = jac @ d
    assert numpy.abs(fd - Jd).max() <= 1e-8 * numpy.abs(Jd).max()
    A = jac.core
    assert abs(A - A.T).max() > 1e-3 * abs(A).max()  # the product-rule term makes the Jacobian non-symmetric
    basis = u.arg.basis
    cons = numpy.full(nd, numpy.nan)
    shape = basis.dofs_shape
    cons.reshape(shape)[0] = 0.
    sol = system.solve(constrain={'u': cons}, tol=1e-11)['u']
    r = system.assemble_residual({'u': sol})
    assert numpy.linalg.norm(r[numpy.isnan(cons)]) < 1e-11 and numpy.abs(sol).max() > .1


@pytest.mark.parametrize('name,btype', [('quasilin3d_p1_4', 'std'), ('quasilin2d_spline2_6', 'spline')])
def test_quasilinear_reference(golden, name, btype):
    '''Residual, Jacobian (CSR, index arrays bit-exact) and Newton solution of -div((1 + u^2) grad u) = 1 against the REAL reference
    (oracle/gen_golden.py:quasilinear_case): the Jacobian's product-rule term goes through per-point coefficient tensors.'''
    from nutils_amd import mesh, function
    from nutils_amd.solver import System
    g = golden(name)
    ndims, n, degree = int(g['ndims']), int(g['n']), int(g['degree'])
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1, n + 1)] * ndims)
    u = domain.field('u', btype=btype, degree=degree)
    v = domain.field('v', btype=btype, degree=degree)
    dV = function.J(geom)
    res = domain.integral((1 + function.value(u) ** 2) * (function.grad(v, geom) * function.grad(u, geom)).sum(-1) * dV, degree=2 * degree + 2) \
        - domain.integral(v * dV, degree=2 * degree + 2)
    rv = function.derivative(res, 'v')
    close(function.eval(rv, arguments={'u': g['x0']}), g['res'])
    values, rowptr, colidx = function.eval(function.as_csr(function.derivative(rv, 'u')), arguments={'u': g['x0']})
    assert numpy.array_equal(rowptr, g['jac_rowptr']) and numpy.array_equal(colidx, g['jac_colidx'])
    close(values, g['jac_values'])
    sol = System(res, trial='u', test='v').solve(constrain={'u': g['cons']}, tol=1e-11)['u']
    close(sol, g['sol'], 1e4)


@pytest.mark.parametrize('ndims,btype,degree', [(3, 'std', 1), (2, 'spline', 2)])
def test_quasilinear_energy(ndims, btype, degree):
    '''System built from an ENERGY, E(u) = int (1 + u^2) |grad u|^2 / 2 - u: residual and Hessian come from two derivatives of a
    field-dependent coefficient times a form quadratic in the field (point factor U.B.U, per-point tensors on either side).
    Residual = dE and Hessian = d residual by central differences; Hessian symmetric; the minimiser solves the quasi-linear PDE.'''
    from nutils_amd import mesh, function
    from nutils_amd.solver import System
    n = 5 if ndims == 3 else 9
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1, n + 1)] * ndims)
    u = domain.field('u', btype=btype, degree=degree)
    dV = function.J(geom)
    nrg = domain.integral((1 + function.value(u) ** 2) * (.5 * (function.grad(u, geom) * function.grad(u, geom)).sum(-1)) * dV, degree=2 * degree + 2) \
        - domain.integral(u * dV, degree=2 * degree)
    system = System(nrg, trial='u')
    nd = system.size
    rng = numpy.random.default_rng(4)
    x0, d = rng.normal(0, .5, nd), rng.normal(0, 1, nd)
    res = system.assemble_residual({'u': x0})
    h = 1e-5
    dE = (system.assemble_value({'u': x0 + h * d}) - system.assemble_value({'u': x0 - h * d})) / (2 * h)
    assert abs(dE - res @ d) <= 1e-8 * numpy.abs(res).max() * numpy.abs(d).max() * nd ** .5
    jac = system.assemble_jacobian({'u': x0})
    fd = (system.assemble_residual({'u': x0 + h * d}) - system.assemble_residual({'u': x0 - h * d})) / (2 * h)
    Jd = jac @ d
    assert numpy.abs(fd - Jd).max() <= 1e-8 * numpy.abs(Jd).max()
    A = jac.core
    assert abs(A - A.T).max() <= 1e-12 * abs(A).max()
    cons = numpy.full(nd, numpy.nan)
    cons.reshape(u.arg.basis.dofs_shape)[0] = 0.
    sol = system.solve(constrain={'u': cons}, tol=1e-11)['u']
    r = system.assemble_residual({'u': sol})
    assert numpy.linalg.norm(r[numpy.isnan(cons)]) < 1e-11 and numpy.abs(sol).max() > .1


@pytest.mark.parametrize('name,btype', [('quasilin_energy3d_p1_4', 'std'), ('quasilin_energy2d_spline2_6', 'spline')])
def test_quasilinear_energy_reference(golden, name, btype):
    '''Energy, residual, Hessian (CSR, index arrays bit-exact) and minimiser of E(u) = int (1 + u^2) |grad u|^2 / 2 - u against the
    REAL reference (oracle/gen_golden.py:quasilinear_case, energy=True).'''
    from nutils_amd import mesh, function
    from nutils_amd.solver import System
    g = golden(name)
    ndims, n, degree = int(g['ndims']), int(g['n']), int(g['degree'])
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1, n + 1)] * ndims)
    u = domain.field('u', btype=btype, degree=degree)
    dV = function.J(geom)
    nrg = domain.integral((1 + function.value(u) ** 2) * (.5 * (function.grad(u, geom) * function.grad(u, geom)).sum(-1)) * dV, degree=2 * degree + 2) \
        - domain.integral(u * dV, degree=2 * degree + 2)
    args = {'u': g['x0']}
    assert abs(function.eval(nrg, arguments=args) - float(g['energy'])) <= 1e-13 * abs(float(g['energy']))
    ru = function.derivative(nrg, 'u')
    close(function.eval(ru, arguments=args), g['res'])
    values, rowptr, colidx = function.eval(function.as_csr(function.derivative(ru, 'u')), arguments=args)
    assert numpy.array_equal(rowptr, g['jac_rowptr']) and numpy.array_equal(colidx, g['jac_colidx'])
    close(values, g['jac_values'])
    sol = System(nrg, trial='u').solve(constrain={'u': g['cons']}, tol=1e-11)['u']
    close(sol, g['sol'], 1e4)


@pytest.mark.parametrize('nelems', [10, 32])
def test_poisson_example(golden, nelems):
    '''/root/reference/examples/poisson.py:28-36 line by line ("direct function manipulation without namespace expressions"):
    constraints from the boundary functional, minimisation of int |grad u|^2 / 2 - u; solution against the example's own return
    value from the real reference (tests/golden/examples_poisson.npz).'''
    from nutils_amd import mesh, function
    from nutils_amd.solver import System
    topo, x = mesh.unitsquare(nelems, etype='square')
    u = topo.field('u', btype='std', degree=1)
    g = u.grad(x)
    J = function.J(x)
    sqr = topo.boundary.integral(u**2 * J, degree=2)
    cons = System(sqr, trial='u').solve_constraints(droptol=1e-12)
    energy = topo.integral((g @ g / 2 - u) * J, degree=1)
    args = System(energy, trial='u').solve(constrain=cons)
    ref = golden('examples_poisson')[f'poisson_{nelems}_u']
    assert numpy.abs(args['u'] - ref).max() <= 1e-12 * numpy.abs(ref).max()


@pytest.mark.parametrize('btype,degree,n', [('spline', 2, 16), ('std', 1, 24), ('spline', 3, 12)])
def test_periodic_helmholtz(btype, degree, n):
    '''-div grad u + u = f on the unit square, periodic in x (mesh.rectilinear(periodic=[0]), mesh.py:34-60: the bases wrap around, the
    boundary has only the bottom and top sides), homogeneous Neumann in y: the discrete solution converges to
    u = sin(2 pi x) cos(pi y) at the rate of the basis, the constant is reproduced exactly, and the matrix couples the first and the
    last element column.'''
    from nutils_amd import mesh, function
    from nutils_amd.solver import System
    errs = []
    for nel in (n, 2 * n):
        domain, geom = mesh.rectilinear([numpy.linspace(0, 1, nel + 1)] * 2, periodic=[0])
        assert [s.axis for s in domain.boundary.sides()] == [1, 1]
        with pytest.raises(KeyError):
            domain.boundary['left']
        u = domain.field('u', btype=btype, degree=degree)
        v = domain.field('v', btype=btype, degree=degree)
        assert len(u.arg.basis) == (nel * degree if btype == 'std' else nel) * (nel * degree + 1 if btype == 'std' else nel + degree)
        dV = function.J(geom)
        grad = lambda w: function.grad(w, geom)
        uex = lambda x: numpy.sin(2 * numpy.pi * x[:, 0]) * numpy.cos(numpy.pi * x[:, 1])
        f = function.PointFunc(lambda x: (1 + 5 * numpy.pi ** 2) * uex(x), geom)
        res = domain.integral(((grad(v) * grad(u)).sum(-1) + v * u) * dV, degree=2 * degree) - domain.integral(v * f * dV, degree=2 * degree + 2)
        system = System(res, trial='u', test='v')
        args = system.solve()
        ue = function.PointFunc(uex, geom)
        err2 = function.eval(domain.integral(u * u * dV, degree=2 * degree + 2) - 2 * domain.integral(u * ue * dV, degree=2 * degree + 2)
                             + domain.integral(ue * ue * dV, degree=2 * degree + 2), args)
        errs.append(max(err2, 0.) ** .5)
        K = system.assemble_jacobian(args).export('dense') if nel == n and nel <= 16 else None
        if K is not None:  # wrap-around coupling: the dofs of the last element column reach the first one
            nd1 = nel + degree if btype == 'spline' else nel * degree + 1
            assert numpy.abs(K[:nd1, -nd1:]).max() > 0
            assert numpy.abs(K - K.T).max() < 1e-12
            ones = numpy.ones(len(K))
            assert abs(ones @ K @ ones - 1.) < 1e-12  # int 1 * 1 dV: the constant is in the periodic space
    assert errs[1] < errs[0] * 2. ** -(degree + 1) * 1.3 and errs[1] < 2e-2


def test_ragged_rational_workload_tiled():
    '''BASELINE.json configs[4] at a size that measures something: the reference fixture iga_plate_p3_l10 (p = 3 NURBS, 10 hierarchical levels, ragged) tiled into one
    mesh of independent plates and assembled in ONE call (tools/ragged_probe.py: every diagonal block must equal the reference's matrix, indices bit-exact, values
    to 1e-13; the tool asserts it and prints the throughput -- 256 plates = 63 488 elements: profiles/r03_ragged_c4.md)'''
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # four arrangements of the element kernel, each against the reference's matrix: the default (two waves per element, symmetric node pairs m >= n mirrored on the
    # way out), all node pairs (NUTILS_AMD_NO_SYM_GRAM), one wave per element (NUTILS_AMD_GENERIC_WAVES=1), mirrored by the gather instead (NUTILS_AMD_SYM_SCRATCH=1)
    for extra in ({}, {'NUTILS_AMD_NO_SYM_GRAM': '1'}, {'NUTILS_AMD_GENERIC_WAVES': '1'}, {'NUTILS_AMD_SYM_SCRATCH': '1'}):
        out = subprocess.run([sys.executable, 'tools/ragged_probe.py', '12', '2'], cwd=root, capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert out.returncode == 0, (out.stdout + out.stderr)[-2000:]
        assert '2976 elements' in out.stdout and 'nh_assemble_matrix' in out.stdout
