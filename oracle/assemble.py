'''ORACLE (test infrastructure, NOT product code): numpy restatement of the
reference's element-integration + sparse-assembly path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import this module; ``nutils_amd`` never does.

Each function cites the reference lines (relative to /root/reference/src/nutils)
whose algorithm it restates.  The restatement is vectorised over elements where
the reference runs a generated per-element Python loop (evaluable.py:6773-6786),
but keeps the reference's data conventions and, for the sparse dedup, the
reference's exact sequence: element-major COO -> flat key row*ncols+col ->
STABLE argsort -> unique -> bincount-accumulate in sorted order ->
compress_indices (evaluable.py:588-616, 5560-5682; numeric.py:434-460, 687-711).

Parity pin (oracle/README.md): every function below is checked in
tests/test_oracle_golden.py against tests/golden/*.npz, which were produced by
the real reference (oracle/gen_golden.py).
'''

import numpy

from . import poly


# --- a10: quadrature tables ------------------------------------------------

def gauss1(degree):
    '''Gauss-Legendre on [0,1] exact for `degree` (points.py:343-355): Golub-
    Welsch eigenproblem with n = degree//2 + 1 points.'''
    n = degree // 2 + 1
    k = numpy.arange(1, n)
    beta = k / numpy.sqrt(4. * k * k - 1.)
    T = numpy.zeros((n, n))
    T[k, k - 1] = beta  # eigh reads the lower triangle
    x, v = numpy.linalg.eigh(T)
    return (x + 1.) * .5, v[0] ** 2


def gauss(degree, ndims):
    '''Tensor-product Gauss points, FIRST coordinate slowest (points.py:144-163).'''
    x1, w1 = gauss1(degree)
    coords = numpy.stack(numpy.meshgrid(*[x1] * ndims, indexing='ij'), -1).reshape(-1, ndims)
    weights = numpy.ones(())
    for _ in range(ndims):
        weights = numpy.multiply.outer(weights, w1)
    return coords, weights.ravel()


# --- a4: structured bases ---------------------------------------------------

def local_spline_coeffs(lknots):
    '''Polynomial pieces (in the element-local coordinate xi in [0,1], highest
    power first) of the p+1 B-splines that are nonzero on the middle knot span of
    the 2p local knots: Cox-de Boor / Piegl-Tiller A2.2 carried out on polynomial
    coefficient vectors (topology.py:2326-2361).'''
    lknots = numpy.asarray(lknots, dtype=float)
    p = len(lknots) // 2
    N = [numpy.poly1d([1.])] + [None] * p
    if p:
        xi = numpy.poly1d([lknots[p] - lknots[p - 1], lknots[p - 1]])
        left = [xi - lknots[p - i - 1] for i in range(p)]
        right = [-xi + lknots[p + i] for i in range(p)]
        for i in range(p):
            saved = 0.
            for r in range(i + 1):
                temp = N[r] / (lknots[p + r] - lknots[p + r - i - 1])
                N[r] = saved + right[r] * temp
                saved = left[i - r] * temp
            N[i + 1] = saved
    out = numpy.zeros((p + 1, p + 1))
    for a, Na in enumerate(N):
        c = numpy.atleast_1d(Na.coeffs)
        out[a, p + 1 - len(c):] = c
    return out


def structured_axis(n, p, continuity, periodic=False):
    '''Per-axis tables of topology.py:2243-2312 for uniform knots: returns
    (coeffs[i] (p+1,p+1), start_dofs[i], ndofs_axis).  continuity=-1 -> spline
    (C^{p-1}); continuity=0 -> 'std' (C^0).  periodic (topology.py:2279-2291):
    no repeated end knots, the knot vector continues with the period, the dof
    ranges wrap around (ndofs = n (p - c)).'''
    c = continuity + p if continuity < 0 else continuity
    k = numpy.arange(n + 1, dtype=float)
    m = numpy.repeat(p - c, n + 1)
    if periodic and m[0] != p + 1:
        dk = k[n] - k[0]
        m, k = m[:n], k[:n]
        ndofs = int(m.sum())
        while m[n:].sum() < p - m[0] + 2:
            k = numpy.concatenate([k, k + dk])
            m = numpy.concatenate([m, m])
            dk *= 2
        km = numpy.repeat(k, m)
        if p > m[0]:
            km = numpy.concatenate([km[-p + m[0]:] - dk, km])
    else:
        m[0] = m[-1] = p
        ndofs = int(m[:n].sum()) + 1
        km = numpy.repeat(k, m)
    offsets = numpy.cumsum(m[:n]) - m[0]
    coeffs = [local_spline_coeffs(km[o:o + 2 * p]) for o in offsets]
    return coeffs, offsets.astype(numpy.int64), ndofs


def structured_basis(shape, btype, degree, periodic=()):
    '''Per-element dof lists and coefficient tables of function.StructuredBasis
    (function.py:3080-3100): element index unravelled with the LAST axis fastest,
    dofs = RavelIndex of per-axis ranges (first axis slowest), coefficients =
    outer polynomial product of the per-axis tables (PolyMul Left..,Right) ravelled
    first axis slowest.  Returns dofs (nelems, nb) int64, coeffs (nelems, nb, nc),
    ndofs.'''
    nd = len(shape)
    axes = [structured_axis(n, degree, -1 if btype == 'spline' else 0, i in periodic) for i, n in enumerate(shape)]
    dofshape = [a[2] for a in axes]
    nelems = int(numpy.prod(shape))
    nb = (degree + 1) ** nd
    nc = poly.ncoeffs(nd, degree * nd)
    dofs = numpy.empty((nelems, nb), dtype=numpy.int64)
    coeffs = numpy.empty((nelems, nb, nc))
    cache = {}
    for e, idx in enumerate(numpy.ndindex(*shape)):
        d = numpy.zeros((1,), dtype=numpy.int64)
        for ax, i in zip(axes, idx):
            rng = (ax[1][i] + numpy.arange(degree + 1)) % ax[2]
            d = (d[:, None] * ax[2] + rng[None, :]).ravel()
        dofs[e] = d
        key = tuple(ax[0][i].tobytes() for ax, i in zip(axes, idx))
        if key not in cache:
            c = axes[0][0][idx[0]]
            for j in range(1, nd):
                cj = axes[j][0][idx[j]]
                c = poly.mul(c[:, None, :], cj[None, :, :], (poly.MulVar.Left,) * j + (poly.MulVar.Right,)).reshape(-1, poly.ncoeffs(j + 1, degree * (j + 1)))
            cache[key] = c
        coeffs[e] = cache[key]
    return dofs, coeffs, int(numpy.prod(dofshape))


# --- a5: basis tabulation ---------------------------------------------------

def tabulate(coeffs, points):
    '''Values and reference gradients at the points: Polyval / PolyGrad
    (evaluable.py:4328-4399, 4584-4653).  coeffs (..., nb, nc), points (nq, nd)
    -> N (..., nq, nb), dN (..., nq, nb, nd).'''
    coeffs = numpy.asarray(coeffs, dtype=float)
    nd = points.shape[1]
    lead = coeffs.shape[:-2]
    flat = coeffs.reshape((-1,) + coeffs.shape[-2:])
    N = numpy.empty((len(flat), len(points), flat.shape[1]))
    dN = numpy.empty((len(flat), len(points), flat.shape[1], nd))
    for i, c in enumerate(flat):
        N[i] = poly.eval_outer(c, points)
        g = poly.grad(c, nd)  # (nb, nd, nc')
        dN[i] = poly.eval_outer(g, points)  # (nq, nb, nd)
    return N.reshape(lead + N.shape[1:]), dN.reshape(lead + dN.shape[1:])


# --- a6: geometry -----------------------------------------------------------

def geometry_iso(verts, gdofs, gN, gdN):
    '''Isoparametric map x = sum_a N_a(xi) X_a: coordinates x (ne, nq, nd),
    Jacobian J[e,q,i,j] = d x_i / d xi_j (function.py:1284-1295 lowering of
    `geom = gbasis @ verts`), inverse via numpy.linalg.inv (numeric.py:221-241)
    and |det J| (evaluable.py:1463-1490, 6187-6192).'''
    X = verts[gdofs]  # (ne, ngb, nd)
    x = numpy.einsum('eqa,eai->eqi', gN, X)
    J = numpy.einsum('eai,eqaj->eqij', X, gdN)
    return x, J


def geometry_affine(origin, size, points):
    '''Axis-aligned box elements x = origin + size * xi (mesh.rectilinear,
    mesh.py:45-52; hierarchical refinements thereof).'''
    ne, nd = origin.shape
    x = origin[:, None, :] + size[:, None, :] * points[None, :, :]
    J = numpy.zeros((ne, len(points), nd, nd))
    for i in range(nd):
        J[:, :, i, i] = size[:, None, i]
    return x, J


def inv(A):
    '''numeric.inv (numeric.py:221-241): numpy.linalg.inv, but exactly singular matrices of the stack give NaN
    (and one RuntimeWarning 'singular matrix') instead of a LinAlgError.'''
    try:
        return numpy.linalg.inv(A)
    except numpy.linalg.LinAlgError:
        import warnings
        warnings.warn('singular matrix', RuntimeWarning)
        out = numpy.empty(A.shape, dtype=float)
        for index in numpy.ndindex(A.shape[:-2]):
            try:
                out[index] = numpy.linalg.inv(A[index])
            except numpy.linalg.LinAlgError:
                out[index] = numpy.nan
        return out


def physical_tables(N, dN, J):
    '''D[e,q,m,0] = N_m, D[e,q,m,1+i] = d N_m / d x_i = sum_j dN[q,m,j] Jinv[j,i]
    (function.py:1221-1231: einsum('Ai,Aij->Aj') with the inverse Jacobian);
    wdet excluded.'''
    Jinv = inv(J)
    G = numpy.einsum('eqmj,eqji->eqmi', dN, Jinv)
    return numpy.concatenate([N[..., None], G], axis=-1), numpy.abs(numpy.linalg.det(J))


# --- a7: local contraction ---------------------------------------------------

def laplace_coefficient(nd, ncomp=1):
    S = 1 + nd
    C = numpy.zeros((ncomp, S, ncomp, S))
    for c in range(ncomp):
        for i in range(nd):
            C[c, 1 + i, c, 1 + i] = 1.
    return C


def mass_coefficient(nd, ncomp=1):
    S = 1 + nd
    C = numpy.zeros((ncomp, S, ncomp, S))
    for c in range(ncomp):
        C[c, 0, c, 0] = 1.
    return C


def elasticity_coefficient(nd, lam, mu):
    '''eta_ij(v) sigma_ij(u), sigma = lam tr(eps) I + 2 mu eps
    (examples/elasticity.py:51-54): C[c,1+i,d,1+j] = lam d_ci d_dj +
    mu (d_cd d_ij + d_cj d_di).'''
    S = 1 + nd
    C = numpy.zeros((nd, S, nd, S))
    for c in range(nd):
        for i in range(nd):
            for d in range(nd):
                for j in range(nd):
                    C[c, 1 + i, d, 1 + j] = lam * (c == i) * (d == j) + mu * ((c == d) * (i == j) + (c == j) * (d == i))
    return C


def local_matrices(Dtest, Dtrial, wdet, C):
    '''A[e,m,c,n,d] = sum_q wdet[e,q] sum_ab Dtest[e,q,m,a] C[c,a,d,b] Dtrial[e,q,n,b]
    -- the einsum('B,ABC->AC', weights, integrand) of sample.py:951-956 applied to
    the bilinear integrand (evaluable.py:6414-6505).'''
    C = numpy.asarray(C)
    sa = slice(None) if C[:, 1:].any() else slice(0, 1)  # slots of the test / trial tables that the integrand contains at all: a mass
    sb = slice(None) if C[..., 1:].any() else slice(0, 1)  # integrand has no gradient node, NaN gradients of a singular element never enter it
    return numpy.einsum('eq,eqma,cadb,eqnb->emcnd', wdet, Dtest[..., sa], C[:, sa][..., sb], Dtrial[..., sb], optimize=True)


def local_vectors(Dtest, wdet, F):
    '''r[e,m,c] = sum_q wdet[e,q] sum_a Dtest[e,q,m,a] F[e,q,c,a].'''
    return numpy.einsum('eq,eqma,eqca->emc', wdet, Dtest, F, optimize=True)


def field_at_points(D, dofs, u):
    '''U[e,q,d,b] = sum_n D[e,q,n,b] u[dofs[e,n], d] (Basis.lower -> Inflate ->
    dot with the argument vector, function.py:2758-2762).'''
    ue = u[dofs] if u.ndim == 2 else u[dofs][..., None]
    return numpy.einsum('eqnb,end->eqdb', D, ue)


# --- a9 + a12: sparse dedup and CSR hand-off --------------------------------

def block_mask(C):
    '''(c,d) component blocks that are structurally present: the reference prunes
    symbolically-zero blocks in the simplifier; for the constant-coefficient forms
    here that coincides with "some coefficient of the block is nonzero".'''
    return numpy.abs(C).sum(axis=(1, 3)) != 0


def coo(A, rowdofs, coldofs, mask):
    '''Element-major COO triplets of the kept blocks, in the order the reference's
    loop_concatenate emits them (evaluable.py:5322-5343): for each element, local
    row m slowest, then test component c, then local column n, then trial
    component d.  Flat dof = scalar * ncomp + comp (function.py:2598-2627).'''
    ne, nbt, nct, nbr, ncr = A.shape
    rows = (rowdofs[:, :, None] * nct + numpy.arange(nct)[None, None, :])  # (ne, nbt, nct)
    cols = (coldofs[:, :, None] * ncr + numpy.arange(ncr)[None, None, :])  # (ne, nbr, ncr)
    R = numpy.broadcast_to(rows[:, :, :, None, None], A.shape)
    Cc = numpy.broadcast_to(cols[:, None, None, :, :], A.shape)
    keep = numpy.broadcast_to(mask[None, None, :, None, :], A.shape)
    return A[keep], R[keep], Cc[keep]


def dedup_csr(values, rows, cols, nrows, ncols):
    '''evaluable.py:588-616 + 5655-5682 + numeric.py:434-460,687-711.'''
    key = rows.astype(numpy.int64) * ncols + cols
    order = numpy.argsort(key, kind='stable')
    skey = key[order]
    first = numpy.ones(len(skey), dtype=bool)
    first[1:] = skey[1:] != skey[:-1]
    ukey = skey[first]
    inverse = numpy.empty(len(key), dtype=numpy.int64)
    inverse[order] = numpy.cumsum(first) - 1
    vals = numpy.bincount(inverse, weights=values, minlength=len(ukey)) if len(ukey) else numpy.zeros(0)
    urow, ucol = numpy.divmod(ukey, ncols)
    rowptr = numpy.searchsorted(urow, numpy.arange(nrows + 1)).astype(numpy.int64)
    return vals, rowptr, ucol.astype(numpy.int64)


def assemble_csr(A, rowdofs, coldofs, nrows, ncols, mask=None):
    if mask is None:
        mask = numpy.ones((A.shape[2], A.shape[4]), dtype=bool)
    v, r, c = coo(A, rowdofs, coldofs, mask)
    return dedup_csr(v, r, c, nrows * A.shape[2], ncols * A.shape[4])


def assemble_vector(r, rowdofs, nrows):
    '''Inflate -> numeric.accumulate (evaluable.py:3389-3411, numeric.py:434-460).'''
    ne, nb, nc = r.shape
    out = numpy.zeros((nrows, nc))
    numpy.add.at(out, rowdofs, r)
    return out


def validate_csr(values, rowptr, colidx, ncols):
    '''matrix/__init__.py:30-70 checks, restated.'''
    assert values.ndim == 1 and rowptr.ndim == 1 and colidx.ndim == 1
    assert rowptr[0] == 0 and rowptr[-1] == len(colidx) == len(values)
    assert (numpy.diff(rowptr) >= 0).all()
    assert len(colidx) == 0 or (colidx.min() >= 0 and colidx.max() < ncols)
    for i in range(len(rowptr) - 1):
        seg = colidx[rowptr[i]:rowptr[i + 1]]
        assert (numpy.diff(seg) > 0).all()


# --- ragged (hierarchical / PlainBasis) path --------------------------------

def ragged_stiffness(dofs, offsets, coeffs, origin, size, points, weights, ndofs):
    '''Per-element loop for bases with element-dependent nb (function.py:2881-2913
    PlainBasis.f_dofs_coeffs; Elemwise tables evaluable.py:3121-3146).'''
    vals, rows, cols = [], [], []
    C = laplace_coefficient(origin.shape[1])
    for e in range(len(offsets) - 1):
        sl = slice(offsets[e], offsets[e + 1])
        N, dN = tabulate(coeffs[sl], points)
        x, J = geometry_affine(origin[e:e + 1], size[e:e + 1], points)
        D, det = physical_tables(N[None], dN[None], J)
        A = local_matrices(D, D, det * weights[None], C)
        v, r, c = coo(A, dofs[sl][None], dofs[sl][None], numpy.ones((1, 1), dtype=bool))
        vals.append(v), rows.append(r), cols.append(c)
    return dedup_csr(numpy.concatenate(vals), numpy.concatenate(rows), numpy.concatenate(cols), ndofs, ndofs)
