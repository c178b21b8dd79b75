'''ORACLE (test infrastructure, NOT product code): numpy restatement of the
polynomial kernels that the reference obtains from the external Rust package
``nutils_poly`` (pinned ``>=1,<2`` in /root/reference/pyproject.toml:12, absent
from /root/reference and not installable here).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import this module.  The product path (``nutils_amd``) has its own
host-side polynomial code and its own device kernels.

Published algorithm restated here (pinned by the reference call sites):

* coefficient order (evaluable.py:4331-4340, Polyval docstring): a polynomial in
  n variables of degree p has C(n+p, n) coefficients; the coefficient for powers
  j precedes the one for powers k iff j_i > k_i at the LAST index i where they
  differ ("reverse lexicographic": last variable most significant, descending
  powers).  1-D is numpy.poly1d order (topology.py:2361).
* ``eval_outer(coeffs, points)`` -> out[pt..., fn...] (evaluable.py:4366-4374,
  element.py:203).
* ``GradPlan(nvars, degree)(coeffs)`` -> (..., nvars, ncoeffs(degree-1))
  (evaluable.py:4629-4633).
* ``MulPlan(vars, degree_left, degree_right)(l, r)`` (evaluable.py:4557-4562).

Parity pin: see oracle/README.md (reference example golden vectors reproduced
through oracle/refshim with this module standing in for nutils_poly).
'''

import functools
import math
import numpy


@functools.lru_cache(maxsize=None)
def powers(nvars, degree):
    '''Exponent table, shape (ncoeffs, nvars), in the reference coefficient order.'''
    if nvars == 0:
        return numpy.zeros((1, 0), dtype=int)
    rows = []
    for klast in range(degree, -1, -1):
        sub = powers(nvars - 1, degree - klast)
        rows.append(numpy.concatenate([sub, numpy.full((len(sub), 1), klast, dtype=int)], axis=1))
    table = numpy.concatenate(rows, axis=0)
    table.setflags(write=False)
    return table


def ncoeffs(nvars, degree):
    return math.comb(nvars + degree, nvars)


def degree(nvars, nc):
    '''Inverse of ncoeffs; raises ValueError if nc is not a valid count.'''
    if nvars == 0:
        if nc != 1:
            raise ValueError('invalid number of coefficients for 0 variables')
        return 0
    d = 0
    while ncoeffs(nvars, d) < nc:
        d += 1
    if ncoeffs(nvars, d) != nc:
        raise ValueError(f'{nc} is not a valid number of coefficients for {nvars} variables')
    return d


def _index_of(nvars, degree):
    return {tuple(p): i for i, p in enumerate(powers(nvars, degree).tolist())}


def eval_outer(coeffs, points):
    coeffs = numpy.asarray(coeffs, dtype=float)
    points = numpy.asarray(points, dtype=float)
    nvars = points.shape[-1]
    deg = degree(nvars, coeffs.shape[-1])
    P = powers(nvars, deg)
    # monomials[pt..., k] = prod_i x_i ** P[k, i]; built by repeated
    # multiplication so that the result is exact for 0/1 powers.
    mono = numpy.ones(points.shape[:-1] + (len(P),))
    for i in range(nvars):
        xi = points[..., i, None]
        pw = numpy.ones(points.shape[:-1] + (deg + 1,))
        for k in range(1, deg + 1):
            pw[..., k] = pw[..., k - 1] * xi[..., 0]
        mono = mono * pw[..., P[:, i]]
    return numpy.tensordot(mono, coeffs, axes=([-1], [-1]))


def eval(coeffs, points):
    '''Pointwise evaluation: coeffs (..., nc), points (..., nvars) broadcast.'''
    coeffs = numpy.asarray(coeffs, dtype=float)
    points = numpy.asarray(points, dtype=float)
    nvars = points.shape[-1]
    deg = degree(nvars, coeffs.shape[-1])
    P = powers(nvars, deg)
    mono = numpy.prod(points[..., None, :] ** P, axis=-1)
    return (mono * coeffs).sum(-1)


class MulVar:
    '''Enumeration Left/Right/Both; repr must end in ``.Left`` etc. because the
    reference code generator embeds ``repr`` (evaluable.py:4557).'''

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return 'MulVar.' + self.name

    @property
    def __nutils_hash__(self):
        import hashlib
        return hashlib.sha1(('oracle.poly.MulVar.' + self.name).encode()).digest()

    def __reduce__(self):
        return (getattr, (MulVar, self.name))


MulVar.Left = MulVar('Left')
MulVar.Right = MulVar('Right')
MulVar.Both = MulVar('Both')


class MulPlan:

    def __init__(self, vars, degree_left, degree_right):
        vars = tuple(vars)
        dl, dr = int(degree_left), int(degree_right)
        left_vars = [i for i, v in enumerate(vars) if v is not MulVar.Right]
        right_vars = [i for i, v in enumerate(vars) if v is not MulVar.Left]
        PL = powers(len(left_vars), dl)
        PR = powers(len(right_vars), dr)
        full_l = numpy.zeros((len(PL), len(vars)), dtype=int)
        full_l[:, left_vars] = PL
        full_r = numpy.zeros((len(PR), len(vars)), dtype=int)
        full_r[:, right_vars] = PR
        lookup = _index_of(len(vars), dl + dr)
        total = full_l[:, None, :] + full_r[None, :, :]
        self.target = numpy.array([lookup[tuple(t)] for t in total.reshape(-1, len(vars)).tolist()], dtype=int)
        self.nout = ncoeffs(len(vars), dl + dr)

    def __call__(self, left, right):
        left = numpy.asarray(left, dtype=float)
        right = numpy.asarray(right, dtype=float)
        lead = numpy.broadcast_shapes(left.shape[:-1], right.shape[:-1])
        prod = (left[..., :, None] * right[..., None, :])
        prod = numpy.broadcast_to(prod, lead + prod.shape[-2:]).reshape(lead + (-1,))
        out = numpy.zeros(lead + (self.nout,))
        numpy.add.at(out, (..., self.target), prod)
        return out


def mul(left, right, vars):
    vars = tuple(vars)
    nl = sum(v is not MulVar.Right for v in vars)
    nr = sum(v is not MulVar.Left for v in vars)
    return MulPlan(vars, degree(nl, numpy.shape(left)[-1]), degree(nr, numpy.shape(right)[-1]))(left, right)


def mul_different_vars(left, right, nleft, nright):
    return mul(left, right, (MulVar.Left,) * nleft + (MulVar.Right,) * nright)


def mul_same_vars(left, right, nvars):
    return mul(left, right, (MulVar.Both,) * nvars)


class GradPlan:

    def __init__(self, nvars, deg):
        deg = int(deg)
        self.nvars = nvars
        P = powers(nvars, deg)
        lookup = _index_of(nvars, max(deg - 1, 0))
        self.nout = len(lookup)
        src, var, dst, fac = [], [], [], []
        for a, p in enumerate(P.tolist()):
            for v in range(nvars):
                if p[v]:
                    q = list(p)
                    q[v] -= 1
                    src.append(a)
                    var.append(v)
                    dst.append(lookup[tuple(q)])
                    fac.append(p[v])
        self.src = numpy.array(src, dtype=int)
        self.var = numpy.array(var, dtype=int)
        self.dst = numpy.array(dst, dtype=int)
        self.fac = numpy.array(fac, dtype=float)

    def __call__(self, coeffs):
        coeffs = numpy.asarray(coeffs, dtype=float)
        out = numpy.zeros(coeffs.shape[:-1] + (self.nvars, self.nout))
        if len(self.src):
            numpy.add.at(out, (..., self.var, self.dst), coeffs[..., self.src] * self.fac)
        return out


def grad(coeffs, nvars):
    return GradPlan(nvars, degree(nvars, numpy.shape(coeffs)[-1]))(coeffs)


def change_degree(coeffs, nvars, newdegree):
    coeffs = numpy.asarray(coeffs, dtype=float)
    old = degree(nvars, coeffs.shape[-1])
    if newdegree < old:
        raise ValueError('cannot lower the degree')
    lookup = _index_of(nvars, newdegree)
    dst = numpy.array([lookup[tuple(p)] for p in powers(nvars, old).tolist()], dtype=int)
    out = numpy.zeros(coeffs.shape[:-1] + (ncoeffs(nvars, newdegree),))
    out[..., dst] = coeffs
    return out


def composition_with_inner_matrix(inner, inner_nvars, outer_nvars, outer_degree):
    '''Matrix M such that coeffs(outer ∘ inner) = M @ coeffs(outer), where
    ``inner`` holds one polynomial (in inner_nvars variables) per outer variable
    (transform.py:181-186).'''
    inner = numpy.asarray(inner, dtype=float)
    if inner.shape[0] != outer_nvars:
        raise ValueError('inner must provide one polynomial per outer variable')
    inner_degree = degree(inner_nvars, inner.shape[-1])
    total = inner_degree * outer_degree
    PO = powers(outer_nvars, outer_degree)
    M = numpy.zeros((ncoeffs(inner_nvars, total), len(PO)))
    unit = numpy.ones(1)
    for col, p in enumerate(PO.tolist()):
        acc = unit
        for v, k in enumerate(p):
            for _ in range(k):
                acc = mul_same_vars(acc, inner[v], inner_nvars)
        M[:, col] = change_degree(acc, inner_nvars, total)
    return M
