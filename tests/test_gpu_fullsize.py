'''Parity at BASELINE.json's FULL sizes through size-independent properties (and, where the C port of the oracle finishes in
seconds, entry by entry): C2 = 128^3 P1 hex Poisson stiffness, C3 = 64^3 P2 vector elasticity.'''
import numpy
import pytest

pytestmark = pytest.mark.gpu


def _csr_matvec(values, rowptr, colidx, x, transpose=False):
    import torch
    rows = torch.repeat_interleave(torch.arange(len(rowptr) - 1, device=values.device), rowptr[1:] - rowptr[:-1])
    src, dst = (rows, colidx) if transpose else (colidx, rows)
    return torch.zeros_like(x).index_add_(0, dst, values * x[src])


@pytest.mark.parametrize('variant', ['iso', 'uniform'])
def test_c2_full_size(variant):
    '''128^3 elements (57 066 625 nonzeros): index arrays bit-exact and values to 1e-13 against the oracle's C port run on the host
    cores, plus the known answers of SURVEY 8c (nnz law, K 1 = 0, symmetry, strictly increasing columns).'''
    import torch
    from nutils_amd import workloads
    n = 128
    wl = workloads.PoissonSlab(n=n, rank=0, world=1, variant=variant)
    wl.setup()
    wl.build_pattern()
    wl.values.fill_(float('nan'))  # write-once kernels: every entry must be stored
    from oracle_check import check_poisson_slab
    assert check_poisson_slab(wl) < 1e-13
    values, rowptr, colidx = wl.values, wl.rowptr, wl.colidx
    assert values.numel() == (3 * n + 1) ** 3 == int(rowptr[-1])
    assert rowptr.dtype == colidx.dtype == torch.int64
    inner = torch.ones(values.numel() - 1, dtype=torch.bool, device=values.device)
    inner[rowptr[1:-1] - 1] = False  # positions where a new row starts
    assert bool(((colidx[1:] - colidx[:-1])[inner] > 0).all())
    scale = float(values.abs().max())
    ones = torch.ones(len(rowptr) - 1, dtype=torch.float64, device=values.device)
    assert float(_csr_matvec(values, rowptr, colidx, ones).abs().max()) < 1e-12 * scale
    x = torch.rand(len(rowptr) - 1, dtype=torch.float64, device=values.device, generator=torch.Generator(device=values.device).manual_seed(1))
    y, yt = _csr_matvec(values, rowptr, colidx, x), _csr_matvec(values, rowptr, colidx, x, transpose=True)
    assert float((y - yt).abs().max()) < 1e-12 * scale
    if variant == 'uniform':
        assert abs(float(values[0]) - 1 / 3) < 1e-15


def test_c3_full_size_rigid_body_modes():
    '''64^3 P2 vector elasticity (6 440 067 dofs, 1.2e9 nonzeros) on the perturbed (isoparametric) mesh: the stiffness matrix
    annihilates the six rigid body modes -- translations and (for the interpolated P2 coordinates) rotations --, is symmetric,
    and obeys the nnz bound 9 (8 n + 1)^3 of SURVEY 8a; the values come from the write-once kernel nh_p2hex_matrix (asserted: the front end
    must not fall back to the generic nh_assemble_matrix for this form).'''
    import torch
    from nutils_amd import mesh, function, sample, device, _lib
    n = 64
    domain, geom = mesh.rectilinear([n] * 3)
    gb = domain.basis('std', degree=1)
    rng = numpy.random.default_rng(0)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (len(gb), 3))
    geom = gb @ verts
    u = domain.field('u', btype='std', degree=2, shape=[3])
    v = domain.field('v', btype='std', degree=2, shape=[3])
    lam, mu = 1., .5 / .3 - 1
    eps = lambda w: function.symgrad(w, geom)
    sigma = lam * function.div(u, geom) * function.eye(3) + 2 * mu * eps(u)
    res = domain.integral(function.inner(eps(v), sigma) * function.J(geom), degree=4)
    jac = function.derivative(function.derivative(res, 'v'), 'u')
    with _lib.trace() as calls:
        values, rowptr, colidx, ncols = sample._MatrixPlan(jac.terms).run()
    assert 'nh_p2hex_matrix' in calls and 'nh_assemble_matrix' not in calls and 'nh_assemble_matrix_terms' not in calls, calls
    ndofs = 3 * (2 * n + 1) ** 3
    assert ncols == ndofs == len(rowptr) - 1
    assert values.numel() == int(rowptr[-1]) <= 9 * (8 * n + 1) ** 3
    assert bool(torch.isfinite(values).all())
    scale = float(values.abs().max())
    # P2 nodal coordinates of the isoparametric map: the P1 geometry evaluated at the P2 nodes (midpoints are averages)
    X = torch.as_tensor(verts.reshape(n + 1, n + 1, n + 1, 3), device=values.device)
    for ax in range(3):
        lo, hi = X.narrow(ax, 0, n), X.narrow(ax, 1, n)
        shape = list(X.shape)
        shape[ax] = 2 * n + 1
        Y = torch.empty(shape, dtype=X.dtype, device=X.device)
        Y.index_copy_(ax, torch.arange(0, 2 * n + 1, 2, device=X.device), X)
        Y.index_copy_(ax, torch.arange(1, 2 * n, 2, device=X.device), .5 * (lo + hi))
        X = Y
    X = X.reshape(-1, 3)
    modes = []
    for c in range(3):
        t = torch.zeros_like(X)
        t[:, c] = 1.
        modes.append(t)
    for a, b in ((0, 1), (0, 2), (1, 2)):
        r = torch.zeros_like(X)
        r[:, a], r[:, b] = -X[:, b], X[:, a]
        modes.append(r)
    for m in modes:  # std P2 (Bernstein) coefficients of a LINEAR field are its values at the control points = nodal coordinates
        y = _csr_matvec(values, rowptr, colidx, m.reshape(-1))
        assert float(y.abs().max()) < 1e-10 * scale * float(m.abs().max()), float(y.abs().max())
    x = torch.rand(ndofs, dtype=torch.float64, device=values.device, generator=torch.Generator(device=values.device).manual_seed(2))
    y, yt = _csr_matvec(values, rowptr, colidx, x), _csr_matvec(values, rowptr, colidx, x, transpose=True)
    assert float((y - yt).abs().max()) < 1e-11 * scale


def test_c3_full_size_every_entry_vs_the_c_port_in_slabs():
    '''BASELINE.json configs[2] at full size, entry-wise: the 64^3 P2 vector-elasticity matrix of the perturbed mesh through the API (nh_p2hex_matrix) against the C port of
    the oracle (oracle/c: element loop + stable-sort dedup as the reference).  The full COO (1.7e9 entries) does not fit a test, so the port assembles SLABS of ten
    element layers of the same mesh (same vertices); a row of a slab matrix whose node lies on no cut plane has received all of its elements, and its columns are those of
    the full matrix minus the slab's dof offset (node numbering is layer-major).  The slabs overlap by two layers: every one of the 6 440 067 rows is compared -- row
    lengths and column indices exactly, values to 1e-13 of the largest entry.'''
    import torch
    from oracle import assemble as oa, port
    from nutils_amd import mesh, function, sample, _lib
    if not port.available():
        pytest.skip('oracle/c is not built')
    n = 64
    domain, geom = mesh.rectilinear([n] * 3)
    gb = domain.basis('std', degree=1)
    rng = numpy.random.default_rng(0)
    verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (len(gb), 3))
    geom = gb @ verts
    u = domain.field('u', btype='std', degree=2, shape=[3])
    v = domain.field('v', btype='std', degree=2, shape=[3])
    lam, mu = 1., .5 / .3 - 1
    eps = lambda w: function.symgrad(w, geom)
    sigma = lam * function.div(u, geom) * function.eye(3) + 2 * mu * eps(u)
    res = domain.integral(function.inner(eps(v), sigma) * function.J(geom), degree=4)
    jac = function.derivative(function.derivative(res, 'v'), 'u')
    with _lib.trace() as calls:
        values, rowptr, colidx, ncols = sample._MatrixPlan(jac.terms).run()
    assert 'nh_p2hex_matrix' in calls
    dev = values.device
    scale = float(values.abs().max())
    # the oracle's tables of one element (uniform over the mesh) and the form
    _, coeffs, _ = oa.structured_basis((1, 1, 1), 'std', 2)
    _, gcoeffs, _ = oa.structured_basis((1, 1, 1), 'std', 1)
    pts, w = oa.gauss(4, 3)
    N, dN = oa.tabulate(coeffs[0], pts)
    gN, gdN = oa.tabulate(gcoeffs[0], pts)
    T = numpy.concatenate([N.T[:, :, None], dN.transpose(1, 0, 2)], axis=2)
    gT = numpy.concatenate([gN.T[:, :, None], gdN.transpose(1, 0, 2)], axis=2)
    C = oa.elasticity_coefficient(3, 1., .5 / .3 - 1)
    V = verts.reshape(n + 1, n + 1, n + 1, 3)
    plane = 3 * (2 * n + 1) ** 2  # dofs per node plane
    step, checked, worst = 8, 0, 0.
    for a in range(0, n, step):
        lo, hi = max(0, a - 1), min(n, a + step + 1)  # element layers of the slab
        vo, rpo, cio, _ = port.form3d((hi - lo, n, n), 2, C, T, gT, w, V[lo:hi + 1].reshape(-1, 3))
        # node planes whose rows are complete in the slab matrix and belong to this step: [2 a, 2 (a + step)) (+ the last plane of the mesh)
        p0, p1 = 2 * a, (2 * (a + step) if a + step < n else 2 * n + 1)
        r0, r1 = (p0 - 2 * lo) * plane, (p1 - 2 * lo) * plane  # rows of the slab matrix
        R0, R1 = p0 * plane, p1 * plane                         # the same rows of the full matrix
        rps = torch.as_tensor(rpo[r0:r1 + 1], device=dev)
        rpf = rowptr[R0:R1 + 1]
        assert bool((rps[1:] - rps[:-1] == rpf[1:] - rpf[:-1]).all()), f'row lengths differ in node planes {p0}..{p1}'
        k0, k1, K0, K1 = int(rpo[r0]), int(rpo[r1]), int(rpf[0]), int(rpf[-1])
        assert bool((torch.as_tensor(cio[k0:k1], device=dev) + 2 * lo * plane == colidx[K0:K1]).all()), f'column indices differ in node planes {p0}..{p1}'
        err = float((torch.as_tensor(vo[k0:k1], device=dev) - values[K0:K1]).abs().max())
        worst = max(worst, err)
        checked += r1 - r0
        del vo, rpo, cio
    assert checked == len(rowptr) - 1 == 3 * (2 * n + 1) ** 3
    assert worst < 1e-13 * scale, worst / scale


def test_c2_full_size_through_the_api():
    '''The same 128^3 isoparametric Poisson stiffness written as a user script writes it (mesh.rectilinear, basis, `basis @ verts`,
    domain.integral, function.eval(as_csr)): the front end recognises the form and takes the write-once kernel; indices and values
    must equal the slab workload that bench.py times.'''
    import time
    import torch
    from nutils_amd import mesh, function, workloads, device
    n = 128
    wl = workloads.PoissonSlab(n=n, rank=0, world=1, variant='iso')
    wl.setup()
    wl.build_pattern()
    wl.step()
    domain, geom = mesh.rectilinear([n] * 3)
    basis = domain.basis('std', degree=1)
    X = basis @ wl.verts
    K = domain.integral(function.outer(function.grad(basis, X)).sum(-1) * function.J(X), degree=2)
    function.eval(function.as_csr(K))  # first call: tables, pattern, upload of the vertices
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    values, rowptr, colidx = function.eval(function.as_csr(K))
    dt = time.perf_counter() - t0
    assert numpy.array_equal(rowptr, device.to_host(wl.rowptr)) and numpy.array_equal(colidx, device.to_host(wl.colidx))
    ref = device.to_host(wl.values)
    assert numpy.abs(values - ref).max() <= 1e-13 * numpy.abs(ref).max()
    assert dt < 2., dt  # device assembly 0.25 ms; the rest is the copy of 1.4 GB of CSR arrays to the host


def test_c4_full_size_consistency():
    '''Cahn-Hilliard 512^2, p=2 splines, two fields (BASELINE.json configs[3]): at full size the residual must be the derivative of
    the energy and the Jacobian the derivative of the residual (central differences along a random direction; the functional is
    a polynomial of degree 4 in phi, so the truncation error is O(eps^2) and tiny), and the Jacobian must be symmetric.'''
    from nutils_amd import mesh, function
    from nutils_amd.solver import System
    n = 512
    size, eps_, M, stens, wn, wp, dt = 10., 1., 1., 50., 30., 20., .5
    domain, geom = mesh.rectilinear([numpy.linspace(0, size, n + 1)] * 2)
    phi = domain.field('φ', btype='spline', degree=2)
    phi0 = domain.field('φ0', btype='spline', degree=2)
    eta = domain.field('η', btype='spline', degree=2) * (stens / eps_)
    p, p0 = function.value(phi), function.value(phi0)
    dp = p - p0
    psi = .25 * (p ** 2 - 1) ** 2
    dpsi = .25 * dp ** 2 * (1 - p ** 2 + 2 * p * dp / 3 - dp ** 2 / 6)
    dV = function.J(geom)
    grad = lambda w: function.grad(w, geom)
    nrg = domain.integral((psi + dpsi) * (stens / eps_) * dV, degree=8) \
        + domain.integral(.5 * stens * eps_ * (grad(phi) * grad(phi)).sum(-1) * dV, degree=8) \
        - domain.integral(eta * phi * dV, degree=8) + domain.integral(eta * phi0 * dV, degree=8) \
        - domain.integral(.5 * dt * M * (grad(eta) * grad(eta)).sum(-1) * dV, degree=8) \
        + domain.boundary.integral((wp + wn) / 2 * dV, degree=4) + domain.boundary.integral((wp - wn) / 2 * phi * dV, degree=4)
    system = System(nrg, trial='φ,η')
    nd = len(phi.arg.basis)
    assert nd == (n + 2) ** 2
    rng = numpy.random.default_rng(0)
    x0 = {'φ': rng.normal(0, .5, nd), 'φ0': rng.normal(0, .5, nd), 'η': rng.normal(0, .1, nd)}
    v = {'φ': rng.normal(0, 1, nd), 'η': rng.normal(0, 1, nd)}
    shift = lambda s: {'φ': x0['φ'] + s * v['φ'], 'η': x0['η'] + s * v['η'], 'φ0': x0['φ0']}
    vv = numpy.concatenate([v['φ'], v['η']])
    res = system.assemble_residual(x0)
    # residual vectors take the owner-side reduction (local vectors + nh_scatter_gather, the order of the reference's add.at loop):
    # bit-identical from run to run -- with global f64 atomics (NUTILS_AMD_ATOMIC_RESIDUALS) equal to rounding only
    for _ in range(2):
        assert numpy.array_equal(system.assemble_residual(x0), res)
    jac = system.assemble_jacobian(x0)
    h = 1e-3
    dE = (system.assemble_value(shift(h)) - system.assemble_value(shift(-h))) / (2 * h)
    assert abs(dE - res @ vv) <= 1e-7 * (abs(dE) + numpy.abs(res).max() * numpy.abs(vv).max())
    dR = (system.assemble_residual(shift(h)) - system.assemble_residual(shift(-h))) / (2 * h)
    Jv = jac @ vv
    assert numpy.abs(dR - Jv).max() <= 1e-6 * numpy.abs(Jv).max()
    A = jac.core
    assert abs(A - A.T).max() <= 1e-12 * abs(A).max()
    assert A.nnz == 4 * (5 * (n + 2) - 6) ** 2  # four blocks of the 5-wide p=2 spline pattern: (5 N - 6)^2 per block, N = n + 2


def test_c2_full_size_generic_gather_path():
    '''The same 128^3 iso case through the GENERIC entry with the owner-side reduction (thread-per-element pass + gather, NH_MATRIX_GATHER |
    NH_MATRIX_STORE on a NaN-filled array): index arrays and values equal to the write-once fast path entry by entry, and bit-identical when repeated.'''
    import torch
    from nutils_amd import workloads
    fast = workloads.PoissonSlab(n=128, rank=0, world=1, variant='iso')
    fast.setup()
    fast.build_pattern()
    fast.step()
    gen = workloads.PoissonSlab(n=128, rank=0, world=1, variant='iso', kernel='gather')
    gen.setup()
    gen.build_pattern()
    assert torch.equal(gen.rowptr, fast.rowptr) and torch.equal(gen.colidx, fast.colidx)
    gen.values.fill_(float('nan'))
    gen.step()
    first = gen.values.clone()
    gen.values.fill_(float('nan'))
    gen.step()
    assert torch.equal(first, gen.values)
    assert float((gen.values - fast.values).abs().max()) <= 1e-13 * float(fast.values.abs().max())


def test_c4_full_size_entries_vs_the_cpu_evaluator_in_strips():
    '''BASELINE.json configs[3] at its full size, ENTRY BY ENTRY: the four Jacobian blocks and both residual blocks of the 512^2 Cahn-Hilliard functional (p = 2 splines, the
    double-well polynomial as coefficient, boundary terms on all sides) against the CPU evaluator of tests/af_oracle.py (numpy element loop, the reference's stable-sort
    dedup) on STRIPS of element rows -- the first rows of the mesh, two interior strips, the last rows: every CSR row whose elements all lie in a strip is complete there
    and is compared whole (column indices exact, values to 1e-13 of the block's largest entry; residual entries to 1e-13 of the largest + 32 ulp of the sum of their |terms|).
    The Python evaluator does ~1 500 elements per second and term, so the strips cover 6 % of the rows; test_c4_full_size_consistency holds the rest to the finite-difference
    and symmetry properties.'''
    import af_oracle
    from nutils_amd import mesh, function, sample as _sample
    n = 512
    size, eps_, M, stens, wn, wp, dt = 10., 1., 1., 50., 30., 20., .5
    domain, geom = mesh.rectilinear([numpy.linspace(0, size, n + 1)] * 2)
    phi = domain.field('φ', btype='spline', degree=2)
    phi0 = domain.field('φ0', btype='spline', degree=2)
    eta = domain.field('η', btype='spline', degree=2) * (stens / eps_)
    p, p0 = function.value(phi), function.value(phi0)
    dp = p - p0
    psi = .25 * (p ** 2 - 1) ** 2
    dpsi = .25 * dp ** 2 * (1 - p ** 2 + 2 * p * dp / 3 - dp ** 2 / 6)
    dV = function.J(geom)
    grad = lambda w: function.grad(w, geom)
    nrg = domain.integral((psi + dpsi) * (stens / eps_) * dV, degree=8) \
        + domain.integral(.5 * stens * eps_ * (grad(phi) * grad(phi)).sum(-1) * dV, degree=8) \
        - domain.integral(eta * phi * dV, degree=8) + domain.integral(eta * phi0 * dV, degree=8) \
        - domain.integral(.5 * dt * M * (grad(eta) * grad(eta)).sum(-1) * dV, degree=8) \
        + domain.boundary.integral((wp + wn) / 2 * dV, degree=4) + domain.boundary.integral((wp - wn) / 2 * phi * dV, degree=4)
    N = n + 2
    rng = numpy.random.default_rng(1)
    args = {'φ': rng.normal(0, .5, N * N), 'φ0': rng.normal(0, .5, N * N), 'η': rng.normal(0, .1, N * N)}
    names = ['φ', 'η']
    res = [function.derivative(nrg, t) for t in names]
    jac = [[function.derivative(r, t) for t in names] for r in res]

    def restrict(integral, a, b):
        '''the terms of `integral` on the elements of the element rows [a, b) (first axis) only'''
        subs, out = {}, []
        for smp, itg, fac in integral.terms:
            if id(smp) not in subs:
                el = numpy.arange(smp.nelems) if smp.elist is None else numpy.asarray(smp.elist)
                keep = el[(el // n >= a) & (el // n < b)]
                subs[id(smp)] = _sample.Sample(smp.topo, smp.points, elist=keep, bnd_axis=smp.bnd_axis) if len(keep) else None
            if subs[id(smp)] is not None:
                out.append((subs[id(smp)], itg, fac))
        return function.Integral(out)

    gpu_res = [numpy.asarray(function.eval(r, args)).ravel() for r in res]
    gpu_jac = [[function.eval(function.as_csr(blk), args) if blk.terms else None for blk in row] for row in jac]
    checked = 0
    for a, b in [(0, 6), (170, 176), (341, 347), (n - 6, n)]:
        lo, hi = (0 if a == 0 else a + 2), (N if b == n else b)  # dof rows (first axis) whose elements all lie in [a, b)
        rows = (numpy.arange(lo, hi)[:, None] * N + numpy.arange(N)).ravel()
        for i in range(2):
            ref = numpy.asarray(af_oracle.evaluate(restrict(res[i], a, b), args)).ravel()
            ab = numpy.asarray(af_oracle.evaluate(restrict(res[i], a, b), args, absolute=True)).ravel()
            tol = 1e-13 * numpy.abs(gpu_res[i]).max() + 32 * 2.3e-16 * ab[rows]
            assert (numpy.abs(gpu_res[i][rows] - ref[rows]) <= tol).all(), (names[i], a, b)
            for j in range(2):
                if gpu_jac[i][j] is None:
                    assert not restrict(jac[i][j], a, b).terms
                    continue
                v, rp, ci = af_oracle.evaluate(restrict(jac[i][j], a, b), args)
                gv, grp, gci = gpu_jac[i][j]
                scale = numpy.abs(gv).max()
                for r in rows[:: 7] if (a, b) != (0, 6) else rows:  # (every row of the first strip, every seventh of the others: the row loop is Python)
                    s0, s1, g0, g1 = rp[r], rp[r + 1], grp[r], grp[r + 1]
                    assert s1 - s0 == g1 - g0 and numpy.array_equal(ci[s0:s1], gci[g0:g1]), (names[i], names[j], r)
                    assert numpy.abs(v[s0:s1] - gv[g0:g1]).max() <= 1e-13 * scale, (names[i], names[j], r)
                    checked += 1
    assert checked > 4000
