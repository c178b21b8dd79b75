'''Pins the oracle (oracle/assemble.py, oracle/poly.py) to the real reference:
every golden file was produced by importing /root/reference (oracle/gen_golden.py).
Index arrays must be bit-exact; float data within 1e-13 relative (the oracle's
einsum association differs from the reference's generated einsum chain).'''
import numpy
import pytest
from oracle import assemble as oa

RTOL = 1e-13

SCALAR = ['lap1d_p1_5', 'lap2d_p1_4x4', 'lap2d_p1_4x3_iso', 'lap2d_p2_3x4_iso', 'lap2d_spline2_4x4', 'lap2d_spline2_5x4_iso',
          'lap3d_p1_2', 'lap3d_p1_3', 'lap3d_p1_4', 'lap3d_p1_234', 'lap3d_p1_3_iso', 'lap3d_p1_543_iso', 'lap3d_p2_2_iso',
          'lap3d_spline2_3_iso', 'lap3d_spline3_3',
          'lap1d_spline3_6_per0', 'lap2d_spline2_5x4_per0', 'lap2d_p2_4x3_per1', 'lap3d_p1_345_per02']  # (last row: periodic axes)
ELAST = ['elast2d_p1_3x3', 'elast2d_p2_3x2_iso', 'elast3d_p1_2_iso', 'elast3d_p2_2', 'elast3d_p2_2_iso']


def close(a, b, scale=None):
    a = numpy.asarray(a); b = numpy.asarray(b)
    assert a.shape == b.shape
    s = numpy.abs(b).max() if scale is None else scale
    assert numpy.abs(a - b).max() <= RTOL * max(s, 1e-300), numpy.abs(a - b).max() / s


def setup(g, name):
    shape = tuple(g['shape']); nd = len(shape); degree = int(g['degree'])
    btype = 'spline' if 'spline' in name else 'std'
    dofs, coeffs, ndofs = oa.structured_basis(shape, btype, degree, tuple(g['periodic']) if 'periodic' in g else ())
    pts, w = oa.gauss(2 * degree, nd)
    N, dN = oa.tabulate(coeffs, pts)
    if int(g['iso']):
        gd, gc, _ = oa.structured_basis(shape, 'std', 1)
        gN, gdN = oa.tabulate(gc, pts)
        x, J = oa.geometry_iso(g['verts'], gd, gN, gdN)
    else:
        origin = numpy.array(list(numpy.ndindex(*shape)), dtype=float)
        x, J = oa.geometry_affine(origin, numpy.ones_like(origin), pts)
    D, det = oa.physical_tables(N, dN, J)
    return nd, dofs, coeffs, ndofs, pts, w, x, D, det


@pytest.mark.parametrize('name', SCALAR + ELAST)
def test_tables(golden, name):
    g = golden(name)
    shape = tuple(g['shape']); degree = int(g['degree'])
    dofs, coeffs, ndofs = oa.structured_basis(shape, 'spline' if 'spline' in name else 'std', degree, tuple(g['periodic']) if 'periodic' in g else ())
    assert dofs.dtype == numpy.int64
    assert numpy.array_equal(dofs.ravel(), g['dofs'])
    close(coeffs.reshape(g['coeffs'].shape), g['coeffs'], 1.)
    pts, w = oa.gauss(2 * degree, len(shape))
    close(pts, g['gauss_coords']); close(w, g['gauss_weights'])
    if int(g['iso']):
        gd, gc, _ = oa.structured_basis(shape, 'std', 1)
        assert numpy.array_equal(gd.ravel(), g['gdofs'])
        close(gc.reshape(g['gcoeffs'].shape), g['gcoeffs'], 1.)


@pytest.mark.parametrize('name', SCALAR)
def test_scalar(golden, name):
    g = golden(name)
    nd, dofs, coeffs, ndofs, pts, w, x, D, det = setup(g, name)
    wdet = det * w
    for key, C in (('K', oa.laplace_coefficient(nd)), ('M', oa.mass_coefficient(nd))):
        A = oa.local_matrices(D, D, wdet, C)
        v, rp, ci = oa.assemble_csr(A, dofs, dofs, ndofs, ndofs)
        assert rp.dtype == ci.dtype == numpy.int64
        assert numpy.array_equal(rp, g[key + '_rowptr']) and numpy.array_equal(ci, g[key + '_colidx'])
        close(v, g[key + '_values'])
        oa.validate_csr(v, rp, ci, ndofs)
    u = g['u']
    U = oa.field_at_points(D, dofs, u)
    close(oa.assemble_vector(oa.local_vectors(D, wdet, numpy.einsum('cadb,eqdb->eqca', oa.laplace_coefficient(nd), U)), dofs, ndofs)[:, 0], g['res_laplace'])
    close(oa.assemble_vector(oa.local_vectors(D, wdet, numpy.einsum('cadb,eqdb->eqca', oa.mass_coefficient(nd), U)), dofs, ndofs)[:, 0], g['res_mass'])
    F = numpy.zeros(U.shape); F[..., 0, 0] = 1
    close(oa.assemble_vector(oa.local_vectors(D, wdet, F), dofs, ndofs)[:, 0], g['load_one'])
    close(wdet.sum(), g['volume'])
    close(.5 * (wdet * (U[:, :, 0, 1:] ** 2).sum(-1)).sum(), g['energy'])
    close(U[:, :, 0, 0].ravel(), g['eval_u'])
    close(U[:, :, 0, 1:].reshape(-1, nd), g['eval_gradu'])
    close(x.reshape(-1, nd), g['eval_x'])
    close(det.ravel(), g['eval_detJ'])


@pytest.mark.parametrize('name', ['lap2d_p1_singular', 'lap3d_p1_singular'])
def test_singular_jacobian(golden, name):
    '''Elements of zero width (repeated coordinate of mesh.rectilinear): the oracle follows numeric.inv (numeric.py:221-241) -- a warning and
    NaN gradients at the points of those elements.  Same NaN entries as the reference, all other entries and the mass matrix to RTOL.'''
    g = golden(name)
    shape = tuple(g['shape']); nd = len(shape)
    dofs, coeffs, ndofs = oa.structured_basis(shape, 'std', 1)
    assert numpy.array_equal(dofs.ravel(), g['dofs'])
    pts, w = oa.gauss(2, nd)
    N, dN = oa.tabulate(coeffs, pts)
    axes = [g[f'coords{i}'] for i in range(nd)]
    idx = numpy.array(list(numpy.ndindex(*shape)))
    origin = numpy.stack([axes[i][idx[:, i]] for i in range(nd)], 1)
    size = numpy.stack([numpy.diff(axes[i])[idx[:, i]] for i in range(nd)], 1)
    x, J = oa.geometry_affine(origin, size, pts)
    with pytest.warns(RuntimeWarning, match='singular matrix'):
        D, det = oa.physical_tables(N, dN, J)
    wdet = det * w
    v, rp, ci = oa.assemble_csr(oa.local_matrices(D, D, wdet, oa.laplace_coefficient(nd)), dofs, dofs, ndofs, ndofs)
    bad = numpy.isnan(g['K_values'])
    assert numpy.array_equal(rp, g['K_rowptr']) and numpy.array_equal(ci, g['K_colidx']) and bad.any() and numpy.array_equal(numpy.isnan(v), bad)
    close(v[~bad], g['K_values'][~bad])
    v, rp, ci = oa.assemble_csr(oa.local_matrices(D, D, wdet, oa.mass_coefficient(nd)), dofs, dofs, ndofs, ndofs)
    close(v, g['M_values'])
    U = oa.field_at_points(D, dofs, g['u'])
    r = oa.assemble_vector(oa.local_vectors(D, wdet, numpy.einsum('cadb,eqdb->eqca', oa.laplace_coefficient(nd), U)), dofs, ndofs)[:, 0]
    rbad = numpy.isnan(g['res_laplace'])
    assert numpy.array_equal(numpy.isnan(r), rbad)
    close(r[~rbad], g['res_laplace'][~rbad])
    close(oa.assemble_vector(oa.local_vectors(D[..., :1], wdet, U[..., :1]), dofs, ndofs)[:, 0], g['res_mass'])
    flat = numpy.abs(g['eval_detJ']) < 1e-15
    assert numpy.isnan(U[:, :, 0, 1:].reshape(-1, nd)[flat]).all()
    close(U[:, :, 0, 1:].reshape(-1, nd)[~flat], g['eval_gradu'][~flat])


@pytest.mark.parametrize('name', ELAST)
def test_elasticity(golden, name):
    g = golden(name)
    nd, dofs, coeffs, ndofs, pts, w, x, D, det = setup(g, name)
    wdet = det * w
    C = oa.elasticity_coefficient(nd, float(g['lam']), float(g['mu']))
    A = oa.local_matrices(D, D, wdet, C)
    v, rp, ci = oa.assemble_csr(A, dofs, dofs, ndofs, ndofs, oa.block_mask(C))
    assert numpy.array_equal(rp, g['K_rowptr']) and numpy.array_equal(ci, g['K_colidx'])
    close(v, g['K_values'])
    U = oa.field_at_points(D, dofs, g['u'])
    F = numpy.einsum('cadb,eqdb->eqca', C, U)
    close(oa.assemble_vector(oa.local_vectors(D, wdet, F), dofs, ndofs), g['res'])
    close(.5 * (wdet * numpy.einsum('eqca,eqca->eq', U, F)).sum(), g['energy'])


@pytest.mark.parametrize('ndim,nnz_t,nnz_h', [(1, 60, 70), (2, 3012, 3424)])
def test_hierarchical(golden, ndim, nnz_t, nnz_h):
    # known answers of /root/reference/tests/test_basis.py:87-116
    g = golden(f'hier_spline2_{ndim}d')
    for key, nnz in (('t', nnz_t), ('h', nnz_h)):
        v, rp, ci = oa.ragged_stiffness(g[key + '_dofs'], g[key + '_dof_offsets'], g[key + '_coeffs'], g['elem_origin'], g['elem_size'],
                                        g['gauss_coords'], g['gauss_weights'], int(g[key + '_ndofs']))
        assert len(v) == nnz
        assert numpy.array_equal(rp, g[key + 'K_rowptr']) and numpy.array_equal(ci, g[key + 'K_colidx'])
        close(v, g[key + 'K_values'])


def test_known_answers():
    # nnz law (3n+1)^3, K[0,0] = 1/3 for the unit-cube trilinear Laplacian, sum(values) = 0
    for n in (2, 5):
        dofs, coeffs, ndofs = oa.structured_basis((n,) * 3, 'std', 1)
        pts, w = oa.gauss(2, 3)
        N, dN = oa.tabulate(coeffs, pts)
        origin = numpy.array(list(numpy.ndindex(n, n, n)), dtype=float)
        x, J = oa.geometry_affine(origin, numpy.ones_like(origin), pts)
        D, det = oa.physical_tables(N, dN, J)
        v, rp, ci = oa.assemble_csr(oa.local_matrices(D, D, det * w, oa.laplace_coefficient(3)), dofs, dofs, ndofs, ndofs)
        assert len(v) == (3 * n + 1) ** 3
        assert abs(v[0] - 1 / 3) < 1e-15
        assert abs(v.sum()) < 1e-12


@pytest.mark.parametrize('name', ['hier_thspline3_2d_l4', 'hier_thspline3_2d_l10'])
def test_hierarchical_p3(golden, name):
    g = golden(name)
    v, rp, ci = oa.ragged_stiffness(g['t_dofs'], g['t_dof_offsets'], g['t_coeffs'], g['elem_origin'], g['elem_size'], g['gauss_coords'], g['gauss_weights'],
                                    int(g['t_ndofs']))
    assert numpy.array_equal(rp, g['tK_rowptr']) and numpy.array_equal(ci, g['tK_colidx'])
    close(v, g['tK_values'])
