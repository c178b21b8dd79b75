'''Checker shared by the large-size GPU tests: a PoissonSlab workload (the object bench.py times) against the oracle's C port.
Test infrastructure -- lives here, not in the product package, because it imports oracle/.'''
import numpy


def check_poisson_slab(wl):
    '''Run one step and compare index arrays (bit-exact) and values (relative max error, returned) with oracle/c through
    oracle.port on the same slab.'''
    from oracle import assemble as oa, port
    from nutils_amd import device
    pts, w = oa.gauss(2, 3)
    _, coeffs, _ = oa.structured_basis((1, 1, 1), 'std', 1)
    N, dN = oa.tabulate(coeffs[0], pts)
    T = numpy.concatenate([N.T[:, :, None], dN.transpose(1, 0, 2)], axis=2)
    s = wl.slab
    wl.step()
    wl.finish()
    v, rp, ci, _ = port.laplace3d((s.local_layers, wl.n, wl.n), 1, T, T, w, wl.verts)
    assert numpy.array_equal(device.to_host(wl.rowptr), rp) and numpy.array_equal(device.to_host(wl.colidx), ci)
    got = device.to_host(wl.values)
    err = numpy.abs(got - v).max() / numpy.abs(v).max()
    assert err < 1e-13, err
    return err
