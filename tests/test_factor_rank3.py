'''SURVEY 8a row a14 with its own object: the sparse Taylor coefficient tensors that the REFERENCE's evaluable.factor built for a
cubic functional (tests/golden/factor_cubic2d_spline2_4.npz, from oracle/gen_golden.py: Monomials of rank 1, 2 and 3, evaluable.py:
5693-5751, 5785-5874) evaluated per step by the Monomial kernel nh_monomial -- value with 1..3 gathered arguments, gradient with the
remaining index as output index and nargs = 0..2 (the reference's Monomial._derivative: multiplicity `powers[0]`, symmetric tensors).'''
import numpy
import pytest


def monomials(g):
    for i in range(int(g['nmonomials'])):
        n = int(g[f'm{i}_nargs'])
        yield g[f'm{i}_values'], [g[f'm{i}_idx{k}'] for k in range(n)], g[f'm{i}_powers']


def test_oracle_restatement_of_monomial(golden):
    g = golden('factor_cubic2d_spline2_4')
    u = g['u']
    val, grad = 0., numpy.zeros_like(u)
    for values, idx, powers in monomials(g):
        term = values.copy()  # evaluable.py:5741-5745: out = values.copy(); out *= arg[index] ...
        for ix in idx:
            term *= u[ix]
        val += term.sum()
        d = values * powers[0]
        for ix in idx[1:]:
            d = d * u[ix]
        numpy.add.at(grad, idx[0], d)
    assert abs(val - float(g['value'])) < 1e-13 * abs(float(g['value']))
    assert numpy.abs(grad - g['gradient']).max() < 1e-13 * numpy.abs(g['gradient']).max()
    assert max(len(idx) for _, idx, _ in monomials(g)) == 3


@pytest.mark.gpu
def test_rank3_monomials_through_nh_monomial(golden):
    from nutils_amd import device, kernels
    g = golden('factor_cubic2d_spline2_4')
    u = device.to_dev(g['u'], 'float64')
    value = device.zeros(1, 'float64')
    grad = device.zeros(len(g['u']), 'float64')
    ranks = []
    for values, idx, powers in monomials(g):
        v = device.to_dev(values, 'float64')
        ix = [device.to_dev(i, 'int64') for i in idx]
        kernels.monomial(v, [u] * len(ix), ix, value)                                          # nargs = 1, 2, 3, scalar result
        kernels.monomial(v, [u] * (len(ix) - 1), ix[1:], grad, out_index=ix[0], alpha=float(powers[0]))  # nargs = 0, 1, 2, scattered
        ranks.append(len(ix))
    assert sorted(ranks) == [1, 2, 3]
    val = float(device.to_host(value)[0])
    assert abs(val - float(g['value'])) < 1e-13 * abs(float(g['value']))
    assert numpy.abs(device.to_host(grad) - g['gradient']).max() < 1e-13 * numpy.abs(g['gradient']).max()


@pytest.mark.gpu
def test_rank3_tensor_built_on_the_device(golden):
    '''The rank-3 tensor of the same cubic functional BUILT by nh_factor_tensor (element moments, sort, run sums, zeros pruned) equals the reference's
    Monomial arrays of the fixture: index arrays bit-exact (same order: flat key ascending, every permutation stored), values to 1e-13; value and
    gradient of nutils_amd.function.factor at the fixture's argument equal the reference's.'''
    from nutils_amd import mesh, function, device
    g = golden('factor_cubic2d_spline2_4')
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1, 5)] * 2)
    u = domain.field('u', btype='spline', degree=2)
    uu = function.value(u)
    dV = function.J(geom)
    E = domain.integral((uu ** 3 / 3. + .5 * (function.grad(u, geom) * function.grad(u, geom)).sum(-1) - u) * dV, degree=6)
    F = function.factor(E)
    (rank, values, indices), = F.T
    ref = next(m for m in monomials(g) if len(m[1]) == 3)
    assert rank == 3 and numpy.array_equal(device.to_host(indices), numpy.stack(ref[1]))
    assert numpy.abs(device.to_host(values) - ref[0]).max() <= 1e-13 * numpy.abs(ref[0]).max()
    val = function.eval(F, u=g['u'])
    assert abs(val - float(g['value'])) <= 1e-13 * abs(float(g['value']))
    grad = function.eval(F.derivative('u'), u=g['u'])
    assert numpy.abs(grad - g['gradient']).max() <= 1e-13 * numpy.abs(g['gradient']).max()


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['quartic2d', 'quartic3d_iso', 'boundary_cubic'])
def test_rank4_factor_equals_direct_integration(case):
    '''Quartic (double-well) functionals: the factored form -- tensors of rank 2, 3 and 4 built once, nh_monomial per evaluation -- against the
    element loop with pointwise polynomial coefficients, value and gradient; structural checks of the rank-4 tensor (symmetric under permutation of
    its index arrays, keys strictly ascending).'''
    from nutils_amd import mesh, function, device
    rng = numpy.random.default_rng(3)
    if case == 'quartic3d_iso':
        domain, geom0 = mesh.rectilinear([3, 2, 4])
        gb = domain.basis('std', degree=1)
        verts = numpy.stack(numpy.meshgrid(numpy.arange(4.), numpy.arange(3.), numpy.arange(5.), indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (len(gb), 3))
        geom = gb @ verts
        btype, degree, gdeg = 'std', 1, 4
    else:
        domain, geom = mesh.rectilinear([numpy.linspace(0, 1, 6), numpy.linspace(0, 2, 5)])
        btype, degree, gdeg = 'spline', 2, 8
    u = domain.field('u', btype=btype, degree=degree)
    uu = function.value(u)
    dV = function.J(geom)
    if case == 'boundary_cubic':
        E = domain.integral((.5 * (function.grad(u, geom) * function.grad(u, geom)).sum(-1) + uu ** 2) * dV, degree=gdeg) + domain.boundary['left'].integral(uu ** 3 * dV, degree=gdeg)
    else:
        E = domain.integral((.25 * (uu ** 2 - 1.) ** 2 + .5 * (function.grad(u, geom) * function.grad(u, geom)).sum(-1) + .5 * uu ** 3) * dV, degree=gdeg)
    F = function.factor(E)
    assert sorted(t[0] for t in F.T) == ([3] if case == 'boundary_cubic' else [3, 4])
    n = len(domain.basis(btype, degree=degree))
    for k, values, indices in F.T:
        idx = device.to_host(indices)
        key = numpy.zeros(idx.shape[1], dtype=object)
        for a in range(k):
            key = key * n + idx[a].astype(object)
        assert all(b > a for a, b in zip(key[:-1], key[1:]))
        v = device.to_host(values)
        assert (v != 0).all()
        swapped = sorted(zip(map(tuple, idx[::-1].T), v))  # the tensor with its axes reversed is the same tensor
        assert numpy.abs(numpy.array([x[1] for x in swapped]) - v).max() <= 1e-14 * numpy.abs(v).max()
    for _ in range(2):
        uval = rng.normal(size=n)
        a, b = function.eval(F, u=uval), function.eval(E, u=uval)
        assert abs(a - b) <= 1e-12 * abs(b)
        ga, gb_ = function.eval(F.derivative('u'), u=uval), function.eval(function.derivative(E, 'u'), u=uval)
        assert numpy.abs(ga - gb_).max() <= 1e-12 * numpy.abs(gb_).max()


@pytest.mark.gpu
def test_factor_refuses_quasilinear_terms_of_degree_three():
    from nutils_amd import mesh, function
    domain, geom = mesh.rectilinear([4, 4])
    u = domain.field('u', btype='std', degree=1)
    uu = function.value(u)
    E = domain.integral((1. + uu) * (function.grad(u, geom) * function.grad(u, geom)).sum(-1) * function.J(geom), degree=3)
    with pytest.raises(NotImplementedError):
        function.factor(E)


@pytest.mark.gpu
def test_factor_of_a_pure_value_functional():
    '''The integral of u^4 has no test / trial slot that carries the name: the argument comes from the polynomial factor (advisor, round 4: StopIteration before).'''
    from nutils_amd import mesh, function
    rng = numpy.random.default_rng(11)
    domain, geom = mesh.rectilinear([numpy.linspace(0, 1, 5), numpy.linspace(0, 2, 4)])
    u = domain.field('u', btype='spline', degree=2)
    E = domain.integral(function.value(u) ** 4 * function.J(geom), degree=8)
    F = function.factor(E)
    assert [t[0] for t in F.T] == [4]
    n = len(domain.basis('spline', degree=2))
    uval = rng.normal(size=n)
    a, b = function.eval(F, u=uval), function.eval(E, u=uval)
    assert abs(a - b) <= 1e-12 * abs(b)
    ga, gb = function.eval(F.derivative('u'), u=uval), function.eval(function.derivative(E, 'u'), u=uval)
    assert numpy.abs(ga - gb).max() <= 1e-12 * numpy.abs(gb).max()
