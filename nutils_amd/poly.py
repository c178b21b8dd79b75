'''Host-side polynomial bookkeeping for the coefficient layout shared with the
device tabulation kernel (nh_poly_tabulate).

The reference obtains these from the external package ``nutils_poly``; the layout
contract is the one documented at /root/reference/src/nutils/evaluable.py:4331-4340
(Polyval): C(n+p, n) coefficients per polynomial of degree p in n variables,
ordered so that the LAST variable is most significant and powers descend
(2 variables, degree 2:  x1^2, x0 x1, x1, x0^2, x0, 1).  Only the operations the
table producers of this package need are provided; evaluation happens on the GPU.
'''

import functools
import math
import numpy


def ncoeffs(nvars, degree):
    return math.comb(nvars + degree, nvars)


def degree(nvars, nc):
    '''Inverse of :func:`ncoeffs` (ValueError if `nc` is not a valid count).'''
    d = 0
    while ncoeffs(nvars, d) < nc:
        d += 1
    if ncoeffs(nvars, d) != nc:
        raise ValueError(f'{nc} coefficients do not describe a polynomial in {nvars} variables')
    return d


def index(powers, total):
    '''Position of the monomial prod x_i^powers[i] in a polynomial of degree `total`.

    Walks the variables from the most significant (last) one: all monomials with
    a higher power of that variable come first, and there are
    C(nrest + r - j, nrest) monomials of the remaining `nrest` variables with
    degree budget r - j for each skipped power j.'''
    pos = 0
    budget = total
    for nrest in range(len(powers) - 1, -1, -1):
        k = powers[nrest]
        if k > budget:
            raise ValueError('monomial exceeds the polynomial degree')
        for j in range(budget, k, -1):
            pos += ncoeffs(nrest, budget - j)
        budget -= k
    return pos


@functools.lru_cache(maxsize=None)
def _raise_targets(nvars, old, new):
    '''Position in a polynomial of degree `new` of every monomial of a polynomial of degree `old` <= `new`, in layout order.'''
    tgt = numpy.empty(ncoeffs(nvars, old), dtype=int)
    for powers in numpy.ndindex(*[old + 1] * nvars):
        if sum(powers) <= old:
            tgt[index(powers, old)] = index(powers, new)
    return tgt


def change_degree(coeffs, nvars, new):
    '''The same polynomials (last axis: coefficients) written at the higher degree `new`: the coefficients move to the positions of their monomials,
    the new ones are zero (what nutils_poly.change_degree does for the reference, evaluable.py:4470-4478).'''
    coeffs = numpy.asarray(coeffs, dtype=float)
    old = degree(nvars, coeffs.shape[-1])
    if old == new:
        return coeffs
    if old > new:
        raise ValueError('change_degree: cannot lower the degree')
    out = numpy.zeros(coeffs.shape[:-1] + (ncoeffs(nvars, new),))
    out[..., _raise_targets(nvars, old, new)] = coeffs
    return out


@functools.lru_cache(maxsize=None)
def _outer_targets(degrees):
    '''For 1-D polynomials of the given degrees in separate variables: flat index of
    x0^(p0-a0) ... x_{n-1}^(p_{n-1}-a_{n-1}) in the product polynomial, for every
    combination of coefficient positions (a0, ..., a_{n-1}) (poly1d order).'''
    total = sum(degrees)
    shape = tuple(p + 1 for p in degrees)
    tgt = numpy.empty(shape, dtype=int)
    for pos in numpy.ndindex(*shape):
        tgt[pos] = index([p - a for p, a in zip(degrees, pos)], total)
    return tgt


def tensor_product(tables):
    '''Coefficients of the tensor-product functions  f(x) = prod_i g^i_{l_i}(x_i).

    tables[i] has shape (n_i, p_i + 1): n_i univariate polynomials of degree p_i in
    numpy.poly1d order (highest power first; the reference's 1-D layout,
    topology.py:2361).  Result: (prod n_i, ncoeffs(nvars, sum p_i)), function index
    with the FIRST axis slowest -- the layout StructuredBasis produces through
    PolyMul(Left.., Right) + ravel (function.py:3093-3099).'''
    tables = [numpy.asarray(t, dtype=float) for t in tables]
    degrees = tuple(t.shape[1] - 1 for t in tables)
    nvars = len(tables)
    tgt = _outer_targets(degrees)
    prod = tables[0]
    for t in tables[1:]:
        prod = prod.reshape(prod.shape[0], 1, -1, 1) * t.reshape(1, t.shape[0], 1, t.shape[1])
        prod = prod.reshape(prod.shape[0] * prod.shape[1], -1)
    out = numpy.zeros((prod.shape[0], ncoeffs(nvars, sum(degrees))))
    out[:, tgt.ravel()] = prod  # targets are distinct monomials: plain assignment
    return out
