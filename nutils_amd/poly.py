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


@functools.lru_cache(maxsize=None)
def powers(nvars, total):
    '''(ncoeffs, nvars) integer array: the powers of the monomial at every coefficient position of a polynomial of degree `total`.'''
    out = numpy.zeros((ncoeffs(nvars, total), nvars), dtype=int)
    for pw in numpy.ndindex(*[total + 1] * nvars):
        if sum(pw) <= total:
            out[index(pw, total)] = pw
    return out


def monomials(points, nvars, total):
    '''(npoints, ncoeffs): the monomials of the layout at the given points'''
    points = numpy.asarray(points, dtype=float).reshape(-1, nvars)
    pw = powers(nvars, total)
    return numpy.prod(points[:, None, :] ** pw[None, :, :], axis=2) if nvars else numpy.ones((len(points), 1))


@functools.lru_cache(maxsize=None)
def _product_table(nvars, total):
    '''(nc, nc) positions of the products of two monomials of the layout (degree `total`), -1 where the product exceeds the degree'''
    pw = powers(nvars, total)
    nc = len(pw)
    tab = numpy.full((nc, nc), -1, dtype=int)
    for i in range(nc):
        for j in range(nc):
            q = pw[i] + pw[j]
            if q.sum() <= total:
                tab[i, j] = index(tuple(int(x) for x in q), total)
    return tab


def _mul(p, q, tab):
    out = numpy.zeros_like(p)
    i, j = numpy.nonzero(tab >= 0)
    numpy.add.at(out, tab[i, j], p[i] * q[j])
    return out


def compose_affine(coeffs, nvars, A, b):
    '''Coefficients of  x -> p(A x + b)  for the polynomials p of the last axis (same degree, same layout): what the reference obtains for a function seen from the
    other side of an interface by evaluating it in the opposite element's coordinates (function.py:1121-1133 `_Opposite`); here the change of coordinates is folded
    into the coefficients once.  Exact algebra: every monomial prod_i y_i^k_i of p is expanded with y_i = sum_j A_ij x_j + b_i by polynomial products in the layout
    (an affine map keeps the total degree).'''
    coeffs = numpy.asarray(coeffs, dtype=float)
    nc = coeffs.shape[-1]
    total = degree(nvars, nc)
    A, b = numpy.asarray(A, dtype=float).reshape(nvars, nvars), numpy.asarray(b, dtype=float).reshape(nvars)
    if not nvars or not total:
        return coeffs.copy()
    tab = _product_table(nvars, total)
    pw = powers(nvars, total)
    one = numpy.zeros(nc)
    one[index((0,) * nvars, total)] = 1.
    lin = []  # y_i as polynomials of x
    for i in range(nvars):
        y = b[i] * one
        for j in range(nvars):
            y[index(tuple(int(k == j) for k in range(nvars)), total)] += A[i, j]
        lin.append(y)
    pows = [[one] for _ in range(nvars)]  # pows[i][k] = y_i^k
    for i in range(nvars):
        for k in range(1, total + 1):
            pows[i].append(_mul(pows[i][-1], lin[i], tab))
    M = numpy.zeros((nc, nc))  # row: monomial of p, column: monomial of the composition
    for r in range(nc):
        term = one
        for i in range(nvars):
            if pw[r, i]:
                term = _mul(term, pows[i][pw[r, i]], tab)
        M[r] = term
    return coeffs @ M
