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
