'''Golden-vector generator (container-only; test infrastructure).

Imports the REAL Python reference from /root/reference/src through the import
shim in oracle/refshim (stand-ins for its absent third-party packages; the
polynomial stand-in is oracle/poly.py) and records, for a ladder of small cases,
the INPUT tables of the element-integration + sparse-assembly hot path and the
reference's OUTPUT (CSR triplets, residual vectors, sampled values).  The
results are committed as small ``.npz`` files under tests/golden/.  Nothing here
travels to the GPU box except those data files.

Usage:  python oracle/gen_golden.py            (regenerates tests/golden/*.npz)
        python oracle/gen_golden.py --check    (regenerates into a scratch directory and compares with the committed
                                                files: integer arrays identical, floats to 1e-13; exit status 1 on a difference)
'''

import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
if not os.path.isdir(REF + '/src/nutils'):
    raise SystemExit('the reference is not present; golden vectors can only be regenerated in the build container')
sys.path[:0] = [os.path.join(HERE, 'refshim'), REF + '/src', REF]

import numpy  # noqa: E402
from nutils import mesh, function  # noqa: E402
from nutils.expression_v2 import Namespace  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def basis_tables(basis, nelems):
    '''Per-element dofs and coefficient tables via the public accessors
    (function.py:2794-2837).  Ragged: concatenated with offsets.'''
    dofs = [numpy.asarray(basis.get_dofs(e), dtype=numpy.int64) for e in range(nelems)]
    coeffs = [numpy.asarray(basis.get_coefficients(e), dtype=float) for e in range(nelems)]
    offsets = numpy.cumsum([0] + [len(d) for d in dofs]).astype(numpy.int64)
    ncs = sorted({c.shape[1] for c in coeffs})
    out = dict(dof_offsets=offsets, dofs=numpy.concatenate(dofs))
    if len(ncs) == 1:
        out['coeffs'] = numpy.concatenate(coeffs, axis=0)
    else:  # mixed polynomial degree: pad to the largest (never happens in the cases below)
        raise NotImplementedError
    return out


def gauss_tables(domain, degree):
    smp = domain.sample('gauss', degree)
    pts = smp.points[0]
    return dict(gauss_coords=numpy.asarray(pts.coords, dtype=float), gauss_weights=numpy.asarray(pts.weights, dtype=float)), smp


def csr(prefix, integral, arguments=None):
    values, rowptr, colidx = function.eval(function.as_csr(integral), arguments or {})
    assert rowptr.dtype == numpy.int64 and colidx.dtype == numpy.int64
    return {prefix + '_values': values, prefix + '_rowptr': rowptr, prefix + '_colidx': colidx}


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    numpy.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path)} bytes; ' + ', '.join(f'{k}{tuple(numpy.shape(v))}' for k, v in arrays.items()))


def perturbed(shape, rng):
    nd = len(shape)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, nd)
    return verts + rng.uniform(-.2, .2, verts.shape)


def scalar_case(name, shape, btype, degree, iso, seed=0, periodic=()):
    '''Laplace stiffness, mass matrix, Laplace residual and load vector for a
    scalar basis on mesh.rectilinear(shape); geometry either the exact uniform
    one or an isoparametric P1 map with seeded vertex perturbation.'''
    rng = numpy.random.default_rng(seed)
    domain, geom0 = mesh.rectilinear(list(shape), periodic=periodic)
    nelems = len(domain)
    data = dict(shape=numpy.array(shape), degree=degree, iso=int(iso))
    if periodic:
        data['periodic'] = numpy.array(periodic)
    gbasis = domain.basis('std', degree=1) if not periodic else None  # (periodic: the geometry is the non-periodic rectilinear map)
    if iso:
        verts = perturbed(shape, rng)
        geom = gbasis @ verts
        data['verts'] = verts
        g = basis_tables(gbasis, nelems)
        data['gdofs'] = g['dofs']
        data['gdof_offsets'] = g['dof_offsets']
        data['gcoeffs'] = g['coeffs']
    else:
        geom = geom0
    basis = domain.basis(btype, degree=degree)
    data.update(basis_tables(basis, nelems))
    gt, smp = gauss_tables(domain, 2 * degree)
    data.update(gt)
    ns = Namespace()
    ns.x = geom
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    ns.basis = basis
    u = rng.normal(size=len(basis))
    ns.u = function.dotarg('u', basis)
    data['u'] = u
    data.update(csr('K', smp.integral('∇_i(basis_m) ∇_i(basis_n) dV' @ ns)))
    data.update(csr('M', smp.integral('basis_m basis_n dV' @ ns)))
    data['res_laplace'] = smp.integrate('∇_i(basis_m) ∇_i(u) dV' @ ns, arguments=dict(u=u))
    data['res_mass'] = smp.integrate('basis_m u dV' @ ns, arguments=dict(u=u))
    data['load_one'] = smp.integrate('basis_m dV' @ ns)
    data['volume'] = smp.integrate('dV' @ ns)
    data['energy'] = smp.integrate('.5 ∇_i(u) ∇_i(u) dV' @ ns, arguments=dict(u=u))
    # Sample.eval: field value, gradient and geometry at the gauss points
    data['eval_u'] = smp.eval('u' @ ns, arguments=dict(u=u))
    data['eval_gradu'] = smp.eval('∇_i(u)' @ ns, arguments=dict(u=u))
    data['eval_x'] = smp.eval('x_i' @ ns)
    data['eval_detJ'] = smp.eval('dV' @ ns)
    save(name, **data)


def singular_case(name, coords, seed=9):
    '''Rectilinear mesh with a repeated coordinate: the elements of zero width have an exactly singular dx/dxi, for which
    numeric.inv (numeric.py:221-241) warns ('singular matrix') and continues with NaN -- the gradients at the points of these
    elements, hence every stiffness entry and residual component they touch, are NaN; det = 0 keeps the mass matrix finite.'''
    rng = numpy.random.default_rng(seed)
    domain, geom = mesh.rectilinear(coords)
    basis = domain.basis('std', degree=1)
    data = dict(shape=numpy.array([len(c) - 1 for c in coords]), degree=1, iso=0)
    for i, c in enumerate(coords):
        data[f'coords{i}'] = c
    data.update(basis_tables(basis, len(domain)))
    gt, smp = gauss_tables(domain, 2)
    data.update(gt)
    ns = Namespace()
    ns.x = geom
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    ns.basis = basis
    u = rng.normal(size=len(basis))
    ns.u = function.dotarg('u', basis)
    data['u'] = u
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        data.update(csr('K', smp.integral('∇_i(basis_m) ∇_i(basis_n) dV' @ ns)))
        data.update(csr('M', smp.integral('basis_m basis_n dV' @ ns)))
        data['res_laplace'] = smp.integrate('∇_i(basis_m) ∇_i(u) dV' @ ns, arguments=dict(u=u))
        data['res_mass'] = smp.integrate('basis_m u dV' @ ns, arguments=dict(u=u))
        data['eval_gradu'] = smp.eval('∇_i(u)' @ ns, arguments=dict(u=u))
        data['eval_detJ'] = smp.eval('dV' @ ns)
    assert any('singular matrix' in str(w.message) for w in caught), 'the reference did not see a singular Jacobian'
    assert numpy.isnan(data['K_values']).any() and not numpy.isnan(data['M_values']).any()
    save(name, **data)


def elasticity_case(name, shape, degree, iso, poisson=.3, seed=1):
    '''Linear elasticity as in examples/elasticity.py:46-54 (lambda = 1,
    mu = .5/poisson - 1), vector field with ndims components interleaved
    (flat dof = iscalar * ncomp + comp, function.py:2598-2627).'''
    rng = numpy.random.default_rng(seed)
    nd = len(shape)
    domain, geom0 = mesh.rectilinear(list(shape))
    nelems = len(domain)
    data = dict(shape=numpy.array(shape), degree=degree, iso=int(iso), lam=1., mu=.5 / poisson - 1)
    gbasis = domain.basis('std', degree=1)
    if iso:
        verts = perturbed(shape, rng)
        geom = gbasis @ verts
        data['verts'] = verts
        g = basis_tables(gbasis, nelems)
        data['gdofs'] = g['dofs']
        data['gdof_offsets'] = g['dof_offsets']
        data['gcoeffs'] = g['coeffs']
    else:
        geom = geom0
    sbasis = domain.basis('std', degree=degree)
    data.update(basis_tables(sbasis, nelems))
    gt, smp = gauss_tables(domain, 2 * degree)
    data.update(gt)
    ns = Namespace()
    ns.δ = function.eye(nd)
    ns.x = geom
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    ns.u = domain.field('u', btype='std', degree=degree, shape=[nd])
    ns.v = domain.field('v', btype='std', degree=degree, shape=[nd])
    ns.λ = 1
    ns.μ = .5 / poisson - 1
    ns.ε_ij = '.5 (∇_i(u_j) + ∇_j(u_i))'
    ns.σ_ij = 'λ ε_kk δ_ij + 2 μ ε_ij'
    ns.η_ij = '.5 (∇_i(v_j) + ∇_j(v_i))'
    res = smp.integral('η_ij σ_ij dV' @ ns)
    u = rng.normal(size=(len(sbasis), nd))
    data['u'] = u
    jac = function.derivative(function.derivative(res, 'v'), 'u')  # shape (ns, nd, ns, nd)
    jac = function.Array.cast(jac)
    jac2 = function.Array.cast(numpy.reshape(jac, (len(sbasis) * nd, len(sbasis) * nd)))
    data.update(csr('K', jac2, dict(u=u * 0, v=u * 0)))
    r = function.eval(function.derivative(res, 'v'), dict(u=u, v=u * 0))
    data['res'] = numpy.asarray(r)
    data['energy'] = smp.integrate('.5 ε_ij σ_ij dV' @ ns, arguments=dict(u=u))
    save(name, **data)


def hierarchical_case(name, ndim):
    '''Ragged case: tests/test_basis.py:94-116 (th-spline / h-spline degree 2 on a
    locally refined 6^ndim mesh; known nnz 60/66/70 (1-D), 3012/3216/3424 (2-D)).'''
    import itertools
    topo, geom = mesh.rectilinear([6] * ndim)
    topo = topo.refined_by(set(map(topo.transforms.index, itertools.chain(topo[1:3].transforms, topo[-2:].transforms))))
    nelems = len(topo)
    data = dict(ndim=ndim)
    smp = topo.sample('gauss', 5)
    # per-element gauss tables are identical for all elements (same reference element)
    pts = smp.points[0]
    data['gauss_coords'] = numpy.asarray(pts.coords, dtype=float)
    data['gauss_weights'] = numpy.asarray(pts.weights, dtype=float)
    ns = Namespace()
    ns.x = geom
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    # element geometry: x = offset + scale * xi for each (axis-aligned) element
    x_at_0 = topo.sample('bezier', 2).eval(geom)
    npe = 2 ** ndim
    x_el = x_at_0.reshape(nelems, npe, ndim)
    data['elem_origin'] = x_el.min(axis=1)
    data['elem_size'] = x_el.max(axis=1) - x_el.min(axis=1)
    for key, btype, kw in (('t', 'th-spline', dict(truncation_tolerance=1e-14)), ('h', 'h-spline', {})):
        basis = topo.basis(btype, degree=2, **kw)
        tb = basis_tables(basis, nelems)
        data[key + '_dofs'] = tb['dofs']
        data[key + '_dof_offsets'] = tb['dof_offsets']
        data[key + '_coeffs'] = tb['coeffs']
        data[key + '_ndofs'] = len(basis)
        ns.b = basis
        data.update(csr(key + 'K', smp.integral('∇_k(b_i) ∇_k(b_j) dV' @ ns)))
    save(name, **data)


def hierarchical_p3_case(name, levels=4):
    '''Ragged nbasis-per-element at p=3 over several hierarchical refinement levels (BASELINE.json configs[4]:
    "p=3 ... hierarchical-refinement levels"; refinement loop as in examples/adaptivity.py:60-70, towards one corner).'''
    topo, geom = mesh.rectilinear([4, 4])
    for lvl in range(levels):
        n = len(topo)
        # refine the elements touching the origin corner region
        bez = topo.sample('gauss', 1).eval(geom).reshape(n, 2)
        sel = [i for i, x in enumerate(bez) if x.max() < 4 * .5 ** lvl]
        topo = topo.refined_by(sel)
    nelems = len(topo)
    data = dict(ndim=2, levels=levels)
    smp = topo.sample('gauss', 6)
    pts = smp.points[0]
    data['gauss_coords'] = numpy.asarray(pts.coords, dtype=float)
    data['gauss_weights'] = numpy.asarray(pts.weights, dtype=float)
    x_el = topo.sample('bezier', 2).eval(geom).reshape(nelems, 4, 2)
    data['elem_origin'] = x_el.min(axis=1)
    data['elem_size'] = x_el.max(axis=1) - x_el.min(axis=1)
    ns = Namespace()
    ns.x = geom
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    basis = topo.basis('th-spline', degree=3)
    tb = basis_tables(basis, nelems)
    data['t_dofs'], data['t_dof_offsets'], data['t_coeffs'], data['t_ndofs'] = tb['dofs'], tb['dof_offsets'], tb['coeffs'], len(basis)
    ns.b = basis
    data.update(csr('tK', smp.integral('∇_k(b_i) ∇_k(b_j) dV' @ ns)))
    data.update(csr('tM', smp.integral('b_i b_j dV' @ ns)))
    save(name, **data)


def cahnhilliard_case(name, nelems, degree=2, seed=3):
    '''Cahn-Hilliard free-energy functional of examples/cahnhilliard.py:163-184 (unit-free restatement: the shipped example
    imports nutils.units, which is not in the reference tree): residual blocks, Jacobian blocks and energy at a random state,
    and the result of one implicit time step (Newton solve).'''
    from nutils.solver import System
    rng = numpy.random.default_rng(seed)
    size, epsilon, mobility, stens, wtensn, wtensp, dt = 10., 1., 1., 50., 30., 20., .5
    domain, geom = mesh.unitsquare(nelems, 'square')
    ns = Namespace()
    ns.x = geom * size
    ns.define_for('x', gradient='∇', normal='n', jacobians=('dV', 'dS'))
    ns.φ = domain.field('φ', btype='std', degree=degree)
    ns.dφ = ns.φ - function.replace_arguments(ns.φ, 'φ:φ0')
    ns.η = domain.field('η', btype='std', degree=degree) * (stens / epsilon)
    ns.dt = dt
    ns.ε = epsilon
    ns.σ = stens
    ns.σmean = (wtensp + wtensn) / 2
    ns.σdiff = (wtensp - wtensn) / 2
    ns.σwall = 'σmean + φ σdiff'
    ns.ψ = '.25 (φ^2 - 1)^2'
    ns.δψ = '.25 dφ^2 (1 - φ^2 + 2 φ dφ / 3 - dφ^2 / 6)'
    ns.M = mobility
    ns.J_i = '-M ∇_i(η)'
    nrg = domain.integral('(ψ σ / ε + .5 σ ε ∇_k(φ) ∇_k(φ) + δψ σ / ε - η dφ + .5 dt J_k ∇_k(η)) dV' @ ns, degree=degree * 4) \
        + domain.boundary.integral('σwall dS' @ ns, degree=degree * 2)
    basis = domain.basis('std', degree=degree)
    n = len(basis)
    args = dict(φ=rng.normal(0, .5, n), φ0=rng.normal(0, .5, n), η=rng.normal(0, .1, n))
    data = dict(nelems=nelems, degree=degree, params=numpy.array([size, epsilon, mobility, stens, wtensn, wtensp, dt]), **{'arg_' + k: v for k, v in args.items()})
    data['energy'] = function.eval(nrg, args)
    for a in 'φη':
        ra = function.derivative(nrg, a)
        data['res_' + a] = function.eval(ra, args)
        for b in 'φη':
            v, rp, ci = function.eval(function.as_csr(function.derivative(ra, b)), args)
            data[f'jac_{a}{b}_values'], data[f'jac_{a}{b}_rowptr'], data[f'jac_{a}{b}_colidx'] = v, rp, ci
    sol = System(nrg, trial='φ,η').solve(arguments=dict(φ=args['φ0'], φ0=args['φ0'], η=numpy.zeros(n)), tol=1e-8)
    data['step_φ'] = sol['φ']
    data['step_η'] = sol['η']
    save(name, **data)


def quasilinear_case(name, ndims, btype, degree, n, seed=6, energy=False):
    '''Quasi-linear diffusion -div((1 + u^2) grad u) = 1 (the nonlinear example class of SURVEY 8f-1): residual and Jacobian of the
    weak form at a random state -- the Jacobian contains the product-rule term 2 u phi_n grad u . grad phi_m (a rank-3 tensor in the
    reference's `factor` language) -- and the Newton solution with u = 0 on the face x_0 = 0.'''
    from nutils.solver import System
    rng = numpy.random.default_rng(seed)
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1, n + 1)] * ndims)
    ns = Namespace()
    ns.x = geom
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    ns.u = domain.field('u', btype=btype, degree=degree)
    ns.v = domain.field('v', btype=btype, degree=degree)
    basis = domain.basis(btype, degree=degree)
    nd = len(basis)
    x0 = rng.normal(0, .5, nd)
    if energy:  # the same problem class from its energy (System(nrg, trial='u') differentiates twice)
        res = domain.integral('((1 + u^2) ∇_i(u) ∇_i(u) / 2 - u) dV' @ ns, degree=2 * degree + 2)
        rv = function.derivative(res, 'u')
    else:
        res = domain.integral('((1 + u^2) ∇_i(v) ∇_i(u) - v) dV' @ ns, degree=2 * degree + 2)
        rv = function.derivative(res, 'v')
    data = dict(ndims=ndims, n=n, degree=degree, x0=x0)
    if energy:
        data['energy'] = function.eval(res, dict(u=x0))
    data['res'] = function.eval(rv, dict(u=x0))
    v, rp, ci = function.eval(function.as_csr(function.derivative(rv, 'u')), dict(u=x0))
    data['jac_values'], data['jac_rowptr'], data['jac_colidx'] = v, rp, ci
    m = n + degree if btype == 'spline' else n * degree + 1  # dofs per axis; dof index = first axis slowest
    cons = numpy.full(nd, numpy.nan)
    cons.reshape((m,) * ndims)[0] = 0.
    data['cons'] = cons
    sol = (System(res, trial='u') if energy else System(res, trial='u', test='v')).solve(constrain=dict(u=cons), tol=1e-11)
    data['sol'] = sol['u']
    save(name, **data)


def nurbs_case(name, nrefine=2, radius=.5, poisson=.3, seed=4):
    '''BASELINE.json configs[4] ingredient: the NURBS mode of examples/platewithhole.py:66-86 -- rational basis
    bspline_i w_i / W(xi) on a refined structured topology, NURBS geometry -- and the plane-strain elasticity stiffness
    matrix / residual of :126-153.  Pointwise data of the coarse-level weight function W and geometry map are stored at the
    Gauss points (they are inputs of the hot path, produced by the reference's transform chains).'''
    from nutils.solver import System
    rng = numpy.random.default_rng(seed)
    topo, geom0 = mesh.rectilinear([1, 2])
    bsplinebasis = topo.basis('spline', degree=2)
    controlweights = numpy.ones(12)
    controlweights[1:3] = .5 + .25 * numpy.sqrt(2)
    weightfunc = bsplinebasis @ controlweights
    nurbsbasis = bsplinebasis * controlweights / weightfunc
    A = 0, 0, 0
    B = (2**.5 - 1) * radius, .3 * (radius + 1) / 2, 1
    C = radius, (radius + 1) / 2, 1
    controlpoints = numpy.array([[A, B, C, C], [C, C, B, A]]).T.reshape(-1, 2)
    geom = nurbsbasis @ controlpoints
    topo = topo.refine(nrefine)
    bsplinebasis = topo.basis('spline', degree=2)
    sqr = topo.integral((function.field('w', bsplinebasis) - weightfunc)**2, degree=9)
    controlweights = System(sqr, trial='w').solve()['w']
    nurbsbasis = bsplinebasis * controlweights / weightfunc
    degree = 5
    smp = topo.sample('gauss', degree * 2)
    nelems = len(topo)
    data = dict(shape=numpy.array([2**nrefine, 2 * 2**nrefine]), nrefine=nrefine, weights=controlweights, lam=2 * poisson, mu=1 - poisson)
    data.update(basis_tables(bsplinebasis, nelems))
    pts = smp.points[0]
    data['gauss_coords'] = numpy.asarray(pts.coords, dtype=float)
    data['gauss_weights'] = numpy.asarray(pts.weights, dtype=float)
    nq = len(pts.weights)
    data['W'] = smp.eval(weightfunc).reshape(nelems, nq)
    data['dW_dparam'] = smp.eval(function.grad(weightfunc, geom0)).reshape(nelems, nq, 2)
    data['x'] = smp.eval(geom).reshape(nelems, nq, 2)
    data['dx_dparam'] = smp.eval(function.grad(geom, geom0)).reshape(nelems, nq, 2, 2)
    data['param'] = smp.eval(geom0).reshape(nelems, nq, 2)
    ns = Namespace()
    ns.δ = function.eye(2)
    ns.x = geom
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    ns.λ = 2 * poisson
    ns.μ = 1 - poisson
    ns.u = function.field('u', nurbsbasis, shape=[2])
    ns.v = function.field('v', nurbsbasis, shape=[2])
    ns.ε_ij = '(∇_j(u_i) + ∇_i(u_j)) / 2'
    ns.σ_ij = 'λ ε_kk δ_ij + 2 μ ε_ij'
    res = smp.integral('∇_j(v_i) σ_ij dV' @ ns)
    n = len(bsplinebasis)
    u = rng.normal(size=(n, 2))
    data['u'] = u
    jac = function.derivative(function.derivative(res, 'v'), 'u')
    jac2 = function.Array.cast(numpy.reshape(function.Array.cast(jac), (n * 2, n * 2)))
    data.update(csr('K', jac2, dict(u=u * 0, v=u * 0)))
    data['res'] = numpy.asarray(function.eval(function.derivative(res, 'v'), dict(u=u, v=u * 0)))
    data['area'] = smp.integrate('dV' @ ns)
    save(name, **data)


def iga_plate_case(name, levels=10, degree=3, radius=.5, poisson=.3, seed=5):
    '''BASELINE.json configs[4] as ONE workload: the NURBS plate-with-hole geometry of examples/platewithhole.py:66-86 (quadratic
    NURBS map, coarse weight function W), `levels` hierarchical refinements towards the hole (topology.refined_by as in
    examples/adaptivity.py:58-70), a p = `degree` truncated hierarchical spline basis made rational with projected weights
    (N_i = w_i B_i / W, platewithhole.py:80-85) -- ragged functions per element, rational, hierarchical, p = 3 together -- and the
    plane-strain elasticity stiffness matrix / residual of platewithhole.py:126-153 on it.'''
    from nutils.solver import System
    rng = numpy.random.default_rng(seed)
    topo, geom0 = mesh.rectilinear([1, 2])
    b2 = topo.basis('spline', degree=2)
    cw = numpy.ones(12)
    cw[1:3] = .5 + .25 * numpy.sqrt(2)
    weightfunc = b2 @ cw
    A = 0, 0, 0
    B = (2**.5 - 1) * radius, .3 * (radius + 1) / 2, 1
    C = radius, (radius + 1) / 2, 1
    controlpoints = numpy.array([[A, B, C, C], [C, C, B, A]]).T.reshape(-1, 2)
    geom = (b2 * cw / weightfunc) @ controlpoints
    topo = topo.refine(1)
    for lvl in range(levels):  # towards the point (0, 1) of the parameter domain: the middle of the hole boundary
        n = len(topo)
        c = topo.sample('gauss', 1).eval(geom0).reshape(n, 2)
        sel = [i for i, x in enumerate(c) if max(abs(x[0]), abs(x[1] - 1.)) < .5 ** lvl]
        topo = topo.refined_by(sel)
    hb = topo.basis('th-spline', degree=degree)
    sqr = topo.integral((function.field('w', hb) - weightfunc)**2, degree=2 * degree + 3)
    w = System(sqr, trial='w').solve()['w']
    nurbs = hb * w / weightfunc
    nelems = len(topo)
    smp = topo.sample('gauss', 8)
    pts = smp.points[0]
    nq = len(pts.weights)
    data = dict(levels=levels, degree=degree, weights=w, lam=2 * poisson, mu=1 - poisson, gauss_coords=numpy.asarray(pts.coords, dtype=float),
                gauss_weights=numpy.asarray(pts.weights, dtype=float))
    tb = basis_tables(hb, nelems)
    data['dofs'], data['dof_offsets'], data['coeffs'], data['ndofs'] = tb['dofs'], tb['dof_offsets'], tb['coeffs'], len(hb)
    corners = topo.sample('bezier', 2).eval(geom0).reshape(nelems, 4, 2)
    data['elem_origin'] = corners.min(axis=1)
    data['elem_size'] = corners.max(axis=1) - corners.min(axis=1)
    data['W'] = smp.eval(weightfunc).reshape(nelems, nq)
    data['dW_dparam'] = smp.eval(function.grad(weightfunc, geom0)).reshape(nelems, nq, 2)
    data['x'] = smp.eval(geom).reshape(nelems, nq, 2)
    data['dx_dparam'] = smp.eval(function.grad(geom, geom0)).reshape(nelems, nq, 2, 2)
    ns = Namespace()
    ns.δ = function.eye(2)
    ns.x = geom
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    ns.λ = 2 * poisson
    ns.μ = 1 - poisson
    ns.u = function.field('u', nurbs, shape=[2])
    ns.v = function.field('v', nurbs, shape=[2])
    ns.ε_ij = '(∇_j(u_i) + ∇_i(u_j)) / 2'
    ns.σ_ij = 'λ ε_kk δ_ij + 2 μ ε_ij'
    res = smp.integral('∇_j(v_i) σ_ij dV' @ ns)
    n = len(hb)
    u = rng.normal(size=(n, 2))
    data['u'] = u
    jac = function.derivative(function.derivative(res, 'v'), 'u')
    jac2 = function.Array.cast(numpy.reshape(function.Array.cast(jac), (n * 2, n * 2)))
    data.update(csr('K', jac2, dict(u=u * 0, v=u * 0)))
    data['res'] = numpy.asarray(function.eval(function.derivative(res, 'v'), dict(u=u, v=u * 0)))
    data['area'] = smp.integrate('dV' @ ns)
    nb = numpy.diff(tb['dof_offsets'])
    print(f'  {nelems} elements, {n} dofs, functions per element {nb.min()}..{nb.max()}, element sizes {data["elem_size"].min():.2e}..{data["elem_size"].max():.2e}')
    save(name, **data)


def factor_rank3_case(name, n=4, degree=2, seed=8):
    '''SURVEY 8a row a14 with its own object: function.factor (function.py:2630-2642 -> evaluable.factor, evaluable.py:5785-5874) of a
    CUBIC functional E(u) = int (u^3 / 3 + |grad u|^2 / 2 - u) dV.  The sparse Taylor coefficient tensors of the reference's Monomials
    (evaluable.py:5693-5751: values + one index array per argument axis, powers) are the INPUT of the per-step work; stored with the
    reference's value and gradient at a random argument.'''
    from nutils import evaluable
    rng = numpy.random.default_rng(seed)
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1, n + 1)] * 2)
    ns = Namespace()
    ns.x = geom
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    ns.u = domain.field('u', btype='spline', degree=degree)
    E = domain.integral('(u^3 / 3 + ∇_i(u) ∇_i(u) / 2 - u) dV' @ ns, degree=3 * degree)
    F = function.factor(E)
    poly = F._array  # evaluable: sum of Sum(Monomial(...))
    monomials = []

    def walk(node, seen=set()):
        if id(node) in seen:
            return
        seen.add(id(node))
        if isinstance(node, evaluable.Monomial):
            monomials.append(node)
            return
        for dep in node.dependencies:
            walk(dep)
    walk(poly)
    ndofs = len(domain.basis('spline', degree=degree))
    u = rng.normal(size=ndofs)
    data = dict(nmonomials=len(monomials), ndofs=ndofs, u=u, value=numpy.asarray(function.eval(F, dict(u=u))),
                gradient=numpy.asarray(function.eval(function.derivative(F, 'u'), dict(u=u))),
                value_unfactored=numpy.asarray(function.eval(E, dict(u=u))))
    for i, m in enumerate(sorted(monomials, key=lambda m: len(m.args))):
        data[f'm{i}_values'] = numpy.asarray(evaluable.eval_once(m.values))
        data[f'm{i}_nargs'] = len(m.args)
        data[f'm{i}_powers'] = numpy.asarray(m.powers, dtype=numpy.int64)
        for k, indices in enumerate(m.indices):
            assert len(indices) == 1 and m.args[k].name == 'u'
            data[f'm{i}_idx{k}'] = numpy.asarray(evaluable.eval_once(indices[0]), dtype=numpy.int64)
    assert max(len(m.args) for m in monomials) == 3
    save(name, **data)


def example_vectors():
    '''Decoded assertAlmostEqual64 payloads of the reference examples
    (examples/laplace.py:111-152, examples/elasticity.py:89-146): the embedded
    base64 strings are data; store the decoded arrays with their tolerance.'''
    from nutils import numeric
    import examples.laplace as lap
    out = {}
    for nelems, etype, btype, degree, tag in ((4, 'square', 'std', 1, 'default'), (4, 'square', 'spline', 2, 'spline')):
        cons, lhs, err = lap.main(nelems=nelems, etype=etype, btype=btype, degree=degree)
        out[f'laplace_{tag}_cons'] = cons
        out[f'laplace_{tag}_lhs'] = lhs
        out[f'laplace_{tag}_err'] = err
    cons, lhs, err = lap.main(nelems=32, etype='square', btype='std', degree=1)
    out['laplace_c1_cons'] = cons
    out['laplace_c1_lhs'] = lhs
    out['laplace_c1_err'] = err
    save('examples_laplace', **out)
    import examples.poisson as poi
    import matplotlib
    matplotlib.use('Agg')
    import tempfile
    out, cwd = {}, os.getcwd()
    os.chdir(tempfile.mkdtemp())  # the example writes u.png into the working directory
    for nelems in (10, 32):
        out[f'poisson_{nelems}_u'] = poi.main(nelems=nelems)['u']
    os.chdir(cwd)
    save('examples_poisson', **out)


def generate_all():
    os.makedirs(OUT, exist_ok=True)
    scalar_case('lap1d_p1_5', (5,), 'std', 1, iso=False)
    scalar_case('lap2d_p1_4x4', (4, 4), 'std', 1, iso=False)
    scalar_case('lap2d_p1_4x3_iso', (4, 3), 'std', 1, iso=True)
    scalar_case('lap2d_p2_3x4_iso', (3, 4), 'std', 2, iso=True)
    scalar_case('lap2d_spline2_4x4', (4, 4), 'spline', 2, iso=False)
    scalar_case('lap2d_spline2_5x4_iso', (5, 4), 'spline', 2, iso=True)
    scalar_case('lap3d_p1_2', (2, 2, 2), 'std', 1, iso=False)
    scalar_case('lap3d_p1_3', (3, 3, 3), 'std', 1, iso=False)
    scalar_case('lap3d_p1_4', (4, 4, 4), 'std', 1, iso=False)
    scalar_case('lap3d_p1_234', (2, 3, 4), 'std', 1, iso=False)
    scalar_case('lap3d_p1_3_iso', (3, 3, 3), 'std', 1, iso=True)
    scalar_case('lap3d_p1_543_iso', (5, 4, 3), 'std', 1, iso=True)
    scalar_case('lap3d_p2_2_iso', (2, 2, 2), 'std', 2, iso=True)
    scalar_case('lap3d_spline2_3_iso', (3, 3, 3), 'spline', 2, iso=True)
    scalar_case('lap3d_spline3_3', (3, 4, 3), 'spline', 3, iso=False)
    scalar_case('lap1d_spline3_6_per0', (6,), 'spline', 3, iso=False, periodic=(0,))
    scalar_case('lap2d_spline2_5x4_per0', (5, 4), 'spline', 2, iso=False, periodic=(0,))
    scalar_case('lap2d_p2_4x3_per1', (4, 3), 'std', 2, iso=False, periodic=(1,))
    scalar_case('lap3d_p1_345_per02', (3, 4, 5), 'std', 1, iso=False, periodic=(0, 2))
    singular_case('lap2d_p1_singular', [numpy.array([0., 1., 1., 2.5]), numpy.array([0., .5, 2.])])
    singular_case('lap3d_p1_singular', [numpy.array([0., 1., 3.]), numpy.array([0., .5, .5, 2.]), numpy.array([0., 1., 1.5])])
    elasticity_case('elast2d_p1_3x3', (3, 3), 1, iso=False)
    elasticity_case('elast2d_p2_3x2_iso', (3, 2), 2, iso=True)
    elasticity_case('elast3d_p1_2_iso', (2, 2, 2), 1, iso=True)
    elasticity_case('elast3d_p2_2', (2, 2, 2), 2, iso=False)
    elasticity_case('elast3d_p2_2_iso', (2, 2, 2), 2, iso=True)
    cahnhilliard_case('cahnhilliard_p2_4', 4)
    nurbs_case('nurbs_plate_r2')
    hierarchical_p3_case('hier_thspline3_2d_l4')
    hierarchical_p3_case('hier_thspline3_2d_l10', levels=10)
    quasilinear_case('quasilin3d_p1_4', 3, 'std', 1, 4)
    quasilinear_case('quasilin2d_spline2_6', 2, 'spline', 2, 6)
    quasilinear_case('quasilin_energy3d_p1_4', 3, 'std', 1, 4, energy=True)
    quasilinear_case('quasilin_energy2d_spline2_6', 2, 'spline', 2, 6, energy=True)  # BASELINE.json configs[4]: ten refinement levels
    factor_rank3_case('factor_cubic2d_spline2_4')
    iga_plate_case('iga_plate_p3_l10')
    hierarchical_case('hier_spline2_1d', 1)
    hierarchical_case('hier_spline2_2d', 2)
    example_vectors()


def check():
    '''Regenerate every fixture in a scratch directory and compare it with the committed file.'''
    import tempfile
    global OUT
    committed = OUT
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        OUT = tmp
        generate_all()
        OUT = committed
        names = sorted(f for f in os.listdir(tmp) if f.endswith('.npz'))
        missing = sorted(set(f for f in os.listdir(committed) if f.endswith('.npz')) ^ set(names))
        bad += [f'{f}: present on one side only' for f in missing]
        for f in names:
            if f in missing:
                continue
            new, old = numpy.load(os.path.join(tmp, f), allow_pickle=False), numpy.load(os.path.join(committed, f), allow_pickle=False)
            if sorted(new.files) != sorted(old.files):
                bad.append(f'{f}: different keys')
                continue
            for k in new.files:
                a, b = new[k], old[k]
                if a.shape != b.shape or a.dtype != b.dtype:
                    bad.append(f'{f}[{k}]: shape / dtype')
                elif a.dtype.kind in 'iub' or a.dtype.kind in 'US':
                    if not numpy.array_equal(a, b):
                        bad.append(f'{f}[{k}]: integer data differs')
                elif a.size and not numpy.array_equal(numpy.isnan(a), numpy.isnan(b)):  # (constraint vectors: NaN = free dof)
                    bad.append(f'{f}[{k}]: NaN pattern differs')
                elif a.size and not numpy.nanmax(numpy.abs(a - b), initial=0.) <= 1e-13 * max(1., numpy.nanmax(numpy.abs(b), initial=0.)):
                    bad.append(f'{f}[{k}]: float data differs by {numpy.nanmax(numpy.abs(a - b)):.2e}')
    print(f'{len(names)} fixtures regenerated, {len(bad)} differences')
    for line in bad:
        print('  ' + line)
    return not bad


if __name__ == '__main__':
    if '--check' in sys.argv[1:]:
        raise SystemExit(0 if check() else 1)
    generate_all()
