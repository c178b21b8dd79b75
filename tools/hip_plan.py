#!/usr/bin/env python3
'''Plan fixtures for the seam tests (build container only: imports /root/reference/src through oracle/refshim).

Runs reference scripts -- the UNMODIFIED examples/laplace.py and examples/elasticity.py (their integrals are captured by wrapping
solver.System.__init__; nothing is edited), and scripts in the reference's own Namespace language for the BASELINE.json configurations
that have no shipped example of that shape: configs[1] `'∇_i(basis_m) ∇_i(basis_n) dV' @ ns` on a 3-D hex mesh (uniform and
isoparametric), configs[2] 3-D P2 vector elasticity, configs[3] the Cahn-Hilliard functional of examples/cahnhilliard.py:163-184 (unit-free:
the shipped example imports nutils.units, which is not in the reference tree), configs[4] the NURBS plate of examples/platewithhole.py:66-86 on
a hierarchically refined th-spline basis (examples/adaptivity.py:58-70) -- matches every array with nutils_amd.seam.match and writes
tests/golden/plans/<name>.npz: the plan, the arguments, and the REFERENCE's own result for it.

tests/test_plans_host.py evaluates every plan on the CPU (tests/af_oracle.py), tests/test_gpu_plans.py through the C ABI
(nutils_amd.seam.execute) and asserts which kernels ran; tests/test_reference_checks.py regenerates the fixtures and compares.
'''
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
if not os.path.isdir(REF + '/src/nutils'):
    raise SystemExit('the reference is not present: plans can only be generated in the build container')
sys.path[:0] = [os.path.join(ROOT, 'oracle', 'refshim'), REF + '/src', REF, ROOT]

import numpy  # noqa: E402
import nutils  # noqa: E402
nutils.__path__.append(os.path.join(ROOT, 'oracle', 'refshim', 'nutils_ext'))  # `nutils.units` (examples/cahnhilliard.py:10): re-export of the reference's nutils.SI
from nutils import function as rf, mesh, solver as rsolver  # noqa: E402
from nutils.expression_v2 import Namespace  # noqa: E402
from nutils_amd import seam  # noqa: E402

OUT = os.environ.get('NUTILS_AMD_PLAN_OUT') or os.path.join(ROOT, 'tests', 'golden', 'plans')


def emit(name, array, arguments, row_axes=None):
    '''match `array`, store the plan with the arguments and the reference's result (CSR of the array flattened to two axes for matrices: the first
    `row_axes` axes are the rows -- half of the axes unless given, which is wrong for blocks of a vector against a scalar field)'''
    plan = seam.match(array)
    expect = {}
    if plan['kind'] == 'matrix':
        flat = rf.Array.cast(numpy.reshape(rf.Array.cast(array), (int(numpy.prod(array.shape[:array.ndim // 2 if row_axes is None else row_axes])), -1)))
        expect['values'], expect['rowptr'], expect['colidx'] = rf.eval(rf.as_csr(flat), arguments)
    elif plan['kind'] == 'vector':
        expect['vector'] = numpy.asarray(rf.eval(array, arguments))
    else:
        expect['scalar'] = numpy.asarray(rf.eval(array, arguments))
    for k, v in arguments.items():
        expect['arg_' + k] = numpy.asarray(v)
    os.makedirs(OUT, exist_ok=True)
    seam.save(os.path.join(OUT, name + '.npz'), plan, expect)
    print(f'{name}: {plan["kind"]}, {len(plan["terms"])} term(s), topos {[t["kind"] for t in plan["topos"]]}, bases {[b["kind"] for b in plan["bases"]]}, '
          f'geometries {[g["kind"] for g in plan["geoms"]]}, derivatives {plan["derivs"]}')
    return plan


def run_example(module, **kwargs):
    '''run examples.<module>.main(**kwargs) unmodified and return the (residual / functional, trial, test) triples it hands to System'''
    import importlib
    captured = []
    orig = rsolver.System.__init__

    def spy(self, residual, /, trial, test=None):
        captured.append((residual, trial, test))
        orig(self, residual, trial=trial, test=test)
    rsolver.System.__init__ = spy
    try:
        import matplotlib
        matplotlib.use('Agg')
        import tempfile
        cwd = os.getcwd()
        os.chdir(tempfile.mkdtemp())
        mod = importlib.import_module('examples.' + module)
        mod.System = rsolver.System  # (the example imported the class by name: same object, patched in place)
        result = mod.main(**kwargs)
        os.chdir(cwd)
    finally:
        rsolver.System.__init__ = orig
    return captured, result


def hex_vertices(shape, rng):
    grid = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.) for n in shape], indexing='ij'), -1).reshape(-1, len(shape))
    return grid + rng.uniform(-.2, .2, grid.shape)


def main():
    # ---- configs[0]: examples/laplace.py, unmodified: residual functional of the second System; std p=1 and spline p=2 --------------------
    for tag, kw in (('laplace_std1', dict(nelems=4, etype='square', btype='std', degree=1)), ('laplace_spline2', dict(nelems=5, etype='square', btype='spline', degree=2))):
        captured, (cons, lhs, err) = run_example('laplace', **kw)
        res = captured[1][0]
        u = numpy.random.default_rng(1).normal(size=lhs.shape)
        args = dict(u=u, v=numpy.zeros_like(u))
        dres = rf.derivative(res, 'v')
        emit(tag + '_matrix', rf.derivative(dres, 'u'), args)
        emit(tag + '_residual', dres, args)  # volume term + the Neumann boundary term with its coefficient function cos(1) cosh(x_1)
        emit(tag + '_constraint_functional', captured[0][0], dict(u=u))  # boundary projection: (u - cosh(x_1) sin(x_0))^2 dS
    # ---- examples/elasticity.py, unmodified: energy functional of the first equilibrium System; P1 and P2 vector fields ---------------------
    for tag, kw in (('elasticity_p1', dict(nelems=4, etype='square', btype='std', degree=1)), ('elasticity_p2', dict(nelems=3, etype='square', btype='std', degree=2))):
        captured, (cons, eargs) = run_example('elasticity', **kw)
        energy = captured[1][0]
        args = dict(u=numpy.random.default_rng(2).normal(size=eargs['u'].shape))
        dE = rf.derivative(energy, 'u')
        emit(tag + '_matrix', rf.derivative(dE, 'u'), args)
        emit(tag + '_residual', dE, args)
    # ---- configs[1]: 3-D Poisson, p=1, the expression of SURVEY 8d with the basis used as an array -------------------------------------------
    rng = numpy.random.default_rng(0)
    for tag, shape, iso in (('c2_uniform_8', (8, 8, 8), False), ('c2_iso_8', (8, 8, 8), True), ('c2_iso_12x9x10', (12, 9, 10), True)):
        domain, geom = mesh.rectilinear(list(shape))
        ns = Namespace()
        ns.x = domain.basis('std', degree=1) @ hex_vertices(shape, rng) if iso else geom
        ns.define_for('x', gradient='∇', jacobians=('dV',))
        ns.basis = domain.basis('std', degree=1)
        emit(tag + '_matrix', domain.integral('∇_i(basis_m) ∇_i(basis_n) dV' @ ns, degree=2), {})
    # (Helmholtz-type form on the same mesh: stiffness + mass in one integrand)
    emit('c2_iso_12x9x10_helmholtz_matrix', domain.integral('(∇_i(basis_m) ∇_i(basis_n) + 3 basis_m basis_n) dV' @ ns, degree=2), {})
    # ---- configs[2]: 3-D linear elasticity, P2 vector field, isoparametric P1 geometry (examples/elasticity.py:51-54 in three dimensions) ------
    shape = (4, 3, 5)
    domain, geom = mesh.rectilinear(list(shape))
    ns = Namespace()
    ns.δ = rf.eye(3)
    ns.x = domain.basis('std', degree=1) @ hex_vertices(shape, rng)
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    ns.λ = 1.
    ns.μ = .5 / .3 - 1
    ns.u = domain.field('u', btype='std', degree=2, shape=[3])
    ns.ε_ij = '(∇_j(u_i) + ∇_i(u_j)) / 2'
    ns.σ_ij = 'λ ε_kk δ_ij + 2 μ ε_ij'
    energy = domain.integral('.5 ε_ij σ_ij dV' @ ns, degree=4)
    args = dict(u=rng.normal(size=(len(domain.basis('std', degree=2)), 3)))
    dE = rf.derivative(energy, 'u')
    emit('c3_p2_4x3x5_matrix', rf.derivative(dE, 'u'), args)
    emit('c3_p2_4x3x5_residual', dE, args)
    # ---- configs[3]: Cahn-Hilliard functional (examples/cahnhilliard.py:163-184, unit-free), residual and Jacobian blocks of a Newton step -----
    size, epsilon, mobility, stens, wtensn, wtensp, dt = 10., 1., 1., 50., 30., 20., .5
    degree = 2
    domain, geom = mesh.unitsquare(6, 'square')
    ns = Namespace()
    ns.x = geom * size
    ns.define_for('x', gradient='∇', normal='n', jacobians=('dV', 'dS'))
    ns.φ = domain.field('φ', btype='std', degree=degree)
    ns.dφ = ns.φ - rf.replace_arguments(ns.φ, 'φ:φ0')
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
    nb = len(domain.basis('std', degree=degree))
    rng4 = numpy.random.default_rng(3)
    args = dict(φ=rng4.normal(0, .5, nb), φ0=rng4.normal(0, .5, nb), η=rng4.normal(0, .1, nb))
    emit('c4_energy', nrg, args)
    for a in 'φη':
        ra = rf.derivative(nrg, a)
        emit(f'c4_residual_{a}', ra, args)
        for b in 'φη':
            emit(f'c4_jacobian_{a}{b}', rf.derivative(ra, b), args)
    # ---- configs[3] from the UNMODIFIED examples/cahnhilliard.py: `function.factor` made transparent by the installed seam (seam.install patches
    # function._Factor), the functional `nrg / tol` captured at its System; residual and Jacobian blocks at the example's own initial state ----------
    from nutils.units.typing import Length, Time, Density  # noqa: E402  (the stand-in)
    st = seam.install(lambda plan, arguments: (_ for _ in ()).throw(RuntimeError('plans are only matched here, not executed')))
    try:
        captured, chargs = run_example('cahnhilliard', epsilon=Length('5cm'), mobility=(Time / Density)('1μL*s/kg'), nelems=3, degree=2, timestep=Time('1h'), endtime=Time('1h'),
                                       circle=False)
    finally:
        seam.uninstall()
    assert st['matched'].count('factor') == 6, st['matched']  # the four energy terms of examples/cahnhilliard.py:181-184 and the two bound post-processing fields (:188,192: `factor(bezier.bind(...))`, point plans) were unwrapped
    nrgx = captured[0][0]
    rngx = numpy.random.default_rng(7)
    nbx = chargs['φ'].shape[0]
    args = dict(φ=rngx.normal(0, .5, nbx), φ0=rngx.normal(0, .5, nbx), η=rngx.normal(0, .1, nbx), dt=numpy.float64(1.))
    emit('c4x_example_energy', nrgx, args)
    for a in 'φη':
        ra = rf.derivative(nrgx, a)
        emit(f'c4x_example_residual_{a}', ra, args)
        for b in 'φη':
            emit(f'c4x_example_jacobian_{a}{b}', rf.derivative(ra, b), args)
    # ---- examples/drivencavity.py, unmodified (Taylor-Hood: velocity degree 2, pressure degree 1 in ONE functional; weak tangential conditions): the
    # Stokes residual captured at its System -- volume terms + the Nitsche boundary terms with n, uwall; rectangular blocks of mixed degree -------------
    captured, _ = run_example('drivencavity', nelems=3, etype='square', degree=2, reynolds=100., compatible=False, strongbc=False)
    stokes = next(res for res, trial, test in captured if trial == 'u,p')
    rngs = numpy.random.default_rng(11)
    shapes = {k: v.shape for k, v in rf.arguments_for(stokes).items()}
    args = {k: rngs.normal(size=shp) for k, shp in shapes.items()}
    for t in 'vq':
        rt = rf.derivative(stokes, t)
        emit(f'stokes_th_residual_{t}', rt, args)
        for a in 'up':
            if (t, a) != ('q', 'p'):  # (no pressure-pressure block)
                emit(f'stokes_th_jacobian_{t}{a}', rf.derivative(rt, a), args, row_axes=len(shapes[t]))
    # ---- examples/cylinderflow.py:110-146 in its own words (6 x 3 elements of the periodic annulus): velocity and pressure on Piola-transformed bases --
    # `function.field('u', function.vectorize([...]) @ J.T / detJ)`: the argument meets its basis only after the basis was multiplied by the Jacobian of the polar map --,
    # the potential-flow functional of the initial condition and the time-step residual (convection, stress, Nitsche terms on the rotating cylinder, `dt` a scalar argument)
    nel, deg = 6, 2
    angle = 2 * numpy.pi / nel
    domain, geom = mesh.rectilinear([3, nel], periodic=(1,))
    domain = domain.withboundary(inner='left', inflow=domain.boundary['right'][nel // 2:])
    ns = Namespace()
    ns.δ = rf.eye(domain.ndims)
    ns.Σ = rf.ones([domain.ndims])
    ns.ε = rf.levicivita(2)
    ns.uinf_i = 'δ_i0'
    ns.Re = 1000.
    ns.grid = geom * angle
    ns.x_i = '.5 exp(grid_0) (sin(grid_1) δ_i0 + cos(grid_1) δ_i1)'
    ns.define_for('x', gradient='∇', normal='n', jacobians=('dV', 'dS'))
    J = ns.x.grad(geom)
    detJ = numpy.linalg.det(J)
    ns.u = rf.field('u', rf.vectorize([domain.basis('spline', degree=(deg, deg - 1), removedofs=((0,), None)), domain.basis('spline', degree=(deg - 1, deg))]) @ J.T / detJ)
    ns.p = domain.field('p', btype='spline', degree=deg - 1) / detJ
    ns.v = rf.replace_arguments(ns.u, 'u:v')
    ns.q = rf.replace_arguments(ns.p, 'p:q')
    ns.du = ns.u - rf.replace_arguments(ns.u, 'u:u0')
    ns.dt = rf.field('dt')
    ns.σ_ij = '(∇_j(u_i) + ∇_i(u_j)) / Re - p δ_ij'
    ns.N = 10 * deg / angle
    ns.nitsche_i = '(N v_i - (∇_j(v_i) + ∇_i(v_j)) n_j) / Re'
    ns.rotation = 1.
    ns.uwall_i = 'rotation ε_ij x_j'
    sqr = domain.integral('(.5 Σ_i (u_i - uinf_i)^2 - ∇_k(u_k) p) dV' @ ns, degree=deg * 2)
    res = domain.integral('v_i du_i dV' @ ns, degree=deg * 3)
    res += domain.integral('(v_i ∇_j(u_i) u_j + ∇_j(v_i) σ_ij + q ∇_k(u_k)) dt dV' @ ns, degree=deg * 3)
    res += domain.boundary['inner'].integral('(nitsche_i (u_i - uwall_i) - v_i σ_ij n_j) dt dS' @ ns, degree=deg * 2)
    rngp = numpy.random.default_rng(17)
    args = {k: rngp.normal(size=v.shape) for k, v in rf.arguments_for(res).items() if k not in 'vq'}
    args['dt'] = numpy.array(.25)
    emit('cylinderflow_potential_residual_u', rf.derivative(sqr, 'u'), args)
    # (the residual block itself is among the plans captured from the unmodified example: tests/golden/plans_examples/cylinderflow_012 -- matching it here again costs 18 s of the CPU suite)
    emit('cylinderflow_step_jacobian_vu', rf.derivative(rf.derivative(res, 'v'), 'u'), args)
    # ---- the volume terms of examples/burgers.py:44-57 in its own words (periodic line, time step as two scalar arguments): `v du / dt` divides by an expression
    # of scalar parameters -- a derived scalar parameter of the plan (1 / (t - t0)); the interface terms of the example (jumps, means: `_Opposite`) are not matched --
    for btype, degree in (('std', 1), ('spline', 2)):
        domain, geom = mesh.line(numpy.linspace(-.5, .5, 21), periodic=True)
        ns = Namespace()
        ns.x = geom
        ns.define_for('x', gradient='∇', normal='n', jacobians=('dV', 'dS'))
        ns.u = domain.field('u', btype=btype, degree=degree)
        ns.du = ns.u - rf.replace_arguments(ns.u, 'u:u0')
        ns.v = domain.field('v', btype=btype, degree=degree)
        ns.t = rf.field('t')
        ns.dt = ns.t - rf.field('t0')
        ns.f = '.5 u^2'
        res = domain.integral('(v du / dt - ∇(v) f) dV' @ ns, degree=degree * 2)
        shapes = {k: v.shape for k, v in rf.arguments_for(res).items()}
        rngb = numpy.random.default_rng(13)
        args = {k: rngb.normal(size=shp) for k, shp in shapes.items() if k != 'v'}
        args['t'], args['t0'] = numpy.array(.7), numpy.array(.45)
        rv = rf.derivative(res, 'v')
        emit(f'burgers_volume_{btype}{degree}_residual', rv, args)
        emit(f'burgers_volume_{btype}{degree}_jacobian', rf.derivative(rv, 'u'), args)
    # ---- discontinuous Galerkin forms: jumps and means across interfaces (function.py:1121-1133 `_Opposite`, :1500-1600 jump / mean).  The complete Burgers residual
    # of examples/burgers.py:51-56 (upwind flux on the periodic line, non-uniform cells) and a symmetric interior-penalty Laplace form on a graded 2-D mesh -- volume
    # term, interface terms with {grad u}, [u] and the penalty, boundary terms -- as residual and Jacobian --------------------------------------------------
    for btype, degree in (('discont', 1), ('legendre', 2)):
        domain, geom = mesh.line(numpy.linspace(-.5, .5, 7) ** 3 * 4, periodic=True)
        ns = Namespace()
        ns.x = geom
        ns.define_for('x', gradient='∇', normal='n', jacobians=('dV', 'dS'))
        ns.u = domain.field('u', btype=btype, degree=degree)
        ns.du = ns.u - rf.replace_arguments(ns.u, 'u:u0')
        ns.v = domain.field('v', btype=btype, degree=degree)
        ns.t = rf.field('t')
        ns.dt = ns.t - rf.field('t0')
        ns.f = '.5 u^2'
        ns.C = 1
        res = domain.integral('(v du / dt - ∇(v) f) dV' @ ns, degree=degree * 2) - domain.interfaces.integral('[v] n ({f} - .5 C [u] n) dS' @ ns, degree=degree * 2)
        shapes = {k: v.shape for k, v in rf.arguments_for(res).items()}
        rngb = numpy.random.default_rng(17)
        args = {k: rngb.normal(size=shp) for k, shp in shapes.items() if k != 'v'}
        args['t'], args['t0'] = numpy.array(.7), numpy.array(.45)
        rv = rf.derivative(res, 'v')
        emit(f'dg_burgers_{btype}{degree}_residual', rv, args)
        emit(f'dg_burgers_{btype}{degree}_jacobian', rf.derivative(rv, 'u'), args)
    for degree in (1, 2):
        domain, geom = mesh.rectilinear([numpy.linspace(0, 1, 5) ** 2, numpy.linspace(0, 2, 4)])
        ns = Namespace()
        ns.x = geom
        ns.define_for('x', gradient='∇', normal='n', jacobians=('dV', 'dS'))
        ns.u = domain.field('u', btype='discont', degree=degree)
        ns.v = domain.field('v', btype='discont', degree=degree)
        ns.β = 7. * degree ** 2
        res = domain.integral('∇_i(v) ∇_i(u) dV' @ ns, degree=2 * degree) \
            - domain.interfaces.integral('([v] n_i {∇_i(u)} + {∇_i(v)} n_i [u] - β [v] [u]) dS' @ ns, degree=2 * degree) \
            + domain.boundary.integral('(β v u - v n_i ∇_i(u) - u n_i ∇_i(v)) dS' @ ns, degree=2 * degree)
        shapes = {k: v.shape for k, v in rf.arguments_for(res).items()}
        args = {k: numpy.random.default_rng(19).normal(size=shp) for k, shp in shapes.items() if k != 'v'}
        rv = rf.derivative(res, 'v')
        emit(f'dg_sipg_p{degree}_residual', rv, args)
        emit(f'dg_sipg_p{degree}_jacobian', rf.derivative(rv, 'u'), args)
    # ---- component blocks per sample: a block-diagonal volume form + a boundary form that couples all components.  The reference runs one loop per
    # sample and concatenates the triplets, so the off-diagonal blocks exist in the rows of the boundary elements only (nnz 364, not 520) --------------
    domain, geom = mesh.rectilinear([3, 4])
    ns = Namespace()
    ns.x = geom
    ns.Σ = rf.ones([2])
    ns.define_for('x', gradient='∇', normal='n', jacobians=('dV', 'dS'))
    ns.u = domain.field('u', btype='std', degree=1, shape=[2])
    ns.v = rf.replace_arguments(ns.u, 'u:v')
    mixed = domain.integral('∇_j(v_i) ∇_j(u_i) dV' @ ns, degree=2) + domain.boundary['right'].integral('v_i Σ_i u_j Σ_j dS' @ ns, degree=2)
    argsm = dict(u=numpy.random.default_rng(5).normal(size=(20, 2)), v=numpy.zeros((20, 2)))
    emit('mixed_blocks_matrix', rf.derivative(rf.derivative(mixed, 'v'), 'u'), argsm, row_axes=2)
    emit('mixed_blocks_residual', rf.derivative(mixed, 'v'), argsm)
    # ---- configs[4]: NURBS plate with hole, hierarchical refinement towards the hole, p = 3 truncated hierarchical splines made rational -------
    levels, degree, radius, poisson = 4, 3, .5, .3
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
        c = topo.sample('gauss', 1).eval(geom0).reshape(len(topo), 2)
        topo = topo.refined_by([i for i, x in enumerate(c) if max(abs(x[0]), abs(x[1] - 1.)) < .5 ** lvl])
    hb = topo.basis('th-spline', degree=degree)
    sqr = topo.integral((rf.field('w', hb) - weightfunc)**2, degree=2 * degree + 3)
    w = rsolver.System(sqr, trial='w').solve()['w']
    nurbs = hb * w / weightfunc
    smp = topo.sample('gauss', 8)
    ns = Namespace()
    ns.δ = rf.eye(2)
    ns.x = geom
    ns.define_for('x', gradient='∇', jacobians=('dV',))
    ns.λ = 2 * poisson
    ns.μ = 1 - poisson
    ns.u = rf.field('u', nurbs, shape=[2])
    ns.v = rf.field('v', nurbs, shape=[2])
    ns.ε_ij = '(∇_j(u_i) + ∇_i(u_j)) / 2'
    ns.σ_ij = 'λ ε_kk δ_ij + 2 μ ε_ij'
    res = smp.integral('∇_j(v_i) σ_ij dV' @ ns)
    u = numpy.random.default_rng(5).normal(size=(len(hb), 2))
    args = dict(u=u, v=u * 0)
    rv = rf.derivative(res, 'v')
    emit('c5_nurbs_hier_p3_matrix', rf.derivative(rv, 'u'), args)
    emit('c5_nurbs_hier_p3_residual', rv, args)
    emit('c5_nurbs_hier_p3_area', smp.integral('dV' @ ns), {})


if __name__ == '__main__':
    main()
