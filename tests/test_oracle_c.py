'''The C port (oracle/c, bench.py's cpu_baseline) against the golden files and the numpy oracle.'''
import numpy
import pytest
from oracle import assemble as oa, port


@pytest.mark.parametrize('name', ['lap3d_p1_2', 'lap3d_p1_234', 'lap3d_p1_543_iso', 'lap3d_p2_2_iso'])
def test_port_vs_golden(golden, name):
    if not port.available():
        pytest.skip('oracle/c not built (run __graft_entry__.build())')
    g = golden(name)
    shape = tuple(g['shape']); degree = int(g['degree'])
    dofs, coeffs, ndofs = oa.structured_basis(shape, 'std', degree)
    pts, w = oa.gauss(2 * degree, 3)
    N, dN = oa.tabulate(coeffs[0], pts)
    T = numpy.concatenate([N.T[:, :, None], dN.transpose(1, 0, 2)], axis=2)
    gd, gc, _ = oa.structured_basis(shape, 'std', 1)
    gN, gdN = oa.tabulate(gc[0], pts)
    gT = numpy.concatenate([gN.T[:, :, None], gdN.transpose(1, 0, 2)], axis=2)
    v, rp, ci, _ = port.laplace3d(shape, degree, T, gT, w, g['verts'] if int(g['iso']) else None, threads=2)
    assert numpy.array_equal(rp, g['K_rowptr']) and numpy.array_equal(ci, g['K_colidx'])
    assert numpy.abs(v - g['K_values']).max() <= 1e-13 * numpy.abs(g['K_values']).max()
