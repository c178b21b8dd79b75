#!/usr/bin/env python3
'''Any-mesh multi-GPU assembly as the ranks would run it (nutils_amd/partition.py: ElementPartition + SharedRowPlan; SURVEY 8e last sentences):

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/partition_run.py [--halo reduce|recompute] [--shape 12,10,8] [--out DIR]

One process per GPU (RCCL: backend nccl); with NUTILS_AMD_BENCH_ONE_GPU=1 all ranks share GPU 0 and the shared rows travel through gloo (host staging) -- how
tests/test_gpu_partition.py runs it on a one-GPU box.  The mesh is an element list in shuffled order (what an imported unstructured mesh gives: trilinear hexahedra
with explicit per-element tables and global dof numbers), the form plane / solid elasticity.  Every rank assembles its local mesh (own + ghost elements) with the HIP
kernels, the partial shared rows are exchanged, every rank writes the rows it owns to --out; rank 0 merges them and checks K . (rigid translation) = 0 on every row.'''
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--halo', default='reduce')
    ap.add_argument('--shape', default='9,8,7')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
    one_gpu = bool(os.environ.get('NUTILS_AMD_BENCH_ONE_GPU'))
    torch.cuda.set_device(0 if one_gpu else local)
    if world > 1:
        dist.init_process_group('gloo' if one_gpu else 'nccl')
    from nutils_amd import mesh, function, topology, partition, device
    shape = [int(x) for x in a.shape.split(',')]
    nd = len(shape)
    rng = numpy.random.default_rng(4)  # (the same mesh on every rank: the connectivity is replicated)
    axes = [numpy.cumsum(numpy.r_[0., rng.uniform(.5, 1.5, n)]) for n in shape]
    sdomain, sgeom = mesh.rectilinear(axes)
    sbasis = sdomain.basis('std', degree=1)
    perm = rng.permutation(len(sdomain))
    origin, size = sgeom.element_boxes()
    coeffs, dofs = [sbasis.get_coefficients(e) for e in perm], [numpy.asarray(sbasis.get_dofs(e)) for e in perm]
    offsets = numpy.cumsum([0] + [len(d) for d in dofs])
    part = partition.ElementPartition(offsets, numpy.concatenate(dofs), sbasis.ndofs, world, ncomp=nd, halo=a.halo)
    el, live = part.local_elements(rank)
    topo = topology.ElementList(origin[perm][el], size[perm][el])
    basis = topo.plain_basis([coeffs[e] for e in el], [dofs[e] for e in el], sbasis.ndofs)
    smp = topo.sample('gauss', 2)
    u, v = function.field('u', basis, shape=[nd]), function.field('v', basis, shape=[nd])
    sig = 1.3 * function.div(u, topo.geom) * function.eye(nd) + 2 * function.symgrad(u, topo.geom)
    res = smp.integral(function.inner(function.symgrad(v, topo.geom), sig) * function.J(topo.geom)
                       * function.PointTable(numpy.repeat(live.astype(float), smp.points.npoints).reshape(len(el), -1)))
    vals, rp, ci = function.eval(function.as_csr(function.derivative(function.derivative(res, 'v'), 'u')))
    plan = partition.SharedRowPlan(part, rank, rp, ci)
    tv = device.to_dev(vals, 'float64')
    if world > 1:
        plan.setup()
        plan.exchange(tv)
        torch.cuda.synchronize()
    rows, lens, cols, v = plan.owned_block(tv)
    # K applied to a rigid translation vanishes on every row (rows of the owned block are complete)
    x = numpy.zeros(sbasis.ndofs * nd)
    x[0::nd] = 1.
    starts = numpy.cumsum(lens) - lens
    rowsum = numpy.add.reduceat(v * x[cols], starts[lens > 0]) if len(v) else numpy.zeros(0)
    err = float(numpy.abs(rowsum).max() / numpy.abs(v).max()) if len(v) else 0.
    sent = sum(len(p) for _, p in plan.send.values())
    print(f'rank {rank}/{world}: {len(el)} local elements ({int(live.sum())} with values), {len(rows)} owned rows, {sent} of {len(vals)} entries sent, |K t| / |K| on owned rows {err:.1e}', flush=True)
    if a.out:
        numpy.savez(os.path.join(a.out, f'rows{rank}.npz'), rows=rows, lens=lens, cols=cols, vals=v)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if err > 1e-12:
        raise SystemExit(f'rank {rank}: owned rows are not complete ({err:.1e})')


if __name__ == '__main__':
    main()
