'''Two slabs of the multi-GPU partition simulated on ONE GPU (sequentially): each "rank" assembles its
slab with the real kernels, the interface-plane reduce is applied with the HaloPlan indices (without the
network hop), and the concatenated owned row blocks must equal the single-mesh assembly.'''
import numpy
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('kernel', ['fast', 'generic'])
@pytest.mark.parametrize('variant', ['iso', 'uniform'])
def test_two_slabs_equal_single_mesh(kernel, variant):
    from nutils_amd import workloads, partition, device
    n, world = 9, 3
    blocks = []
    prev_tail = None
    for rank in range(world):
        wl = workloads.PoissonSlab(n=n, rank=rank, world=world, variant=variant, kernel=kernel)
        wl.setup()
        wl.build_pattern()
        wl.step(exchange=False)
        if wl.slab.recvs:  # what irecv + index_add_ do in HaloPlan.exchange
            assert prev_tail.numel() == wl.halo.recv_buf.numel()
            wl.values.index_add_(0, wl.halo.recv_idx, prev_tail)
        if wl.slab.sends:
            prev_tail = wl.values[wl.halo.send_a:wl.halo.send_b].clone()
        blocks.append(wl.owned_csr())
    v, rp, ci = partition.concatenate(blocks)
    # single mesh (n*world) x n x n through the generic path
    from nutils_amd import mesh, function
    domain, geom = mesh.rectilinear([n * world, n, n])
    basis = domain.basis('std', degree=1)
    if variant == 'iso':
        rng = numpy.random.default_rng(0)
        verts = numpy.stack(numpy.meshgrid(numpy.arange(n * world + 1.), numpy.arange(n + 1.), numpy.arange(n + 1.), indexing='ij'), -1) \
            + rng.uniform(-.2, .2, (n * world + 1, n + 1, n + 1, 3))
        geom = basis @ verts.reshape(-1, 3)
    K = domain.integral(function.outer(function.grad(basis, geom)).sum(-1) * function.J(geom), degree=2)
    vo, rpo, cio = function.eval(function.as_csr(K))
    assert numpy.array_equal(rp, rpo) and numpy.array_equal(ci, cio)
    assert numpy.abs(v - vo).max() <= 1e-13 * numpy.abs(vo).max()
