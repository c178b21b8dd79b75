#!/usr/bin/env python3
'''A ragged / rational workload at scale (BASELINE.json configs[4] class): the reference fixture tests/golden/iga_plate_p3_l10.npz -- NURBS plate with hole,
p = 3 truncated hierarchical splines over 10 refinement levels, 16-24 functions per element, rational, tabulated NURBS geometry, plane-strain elasticity
(2 components) -- tiled `copies` times into ONE mesh (dofs of copy i shifted by i * ndofs: a block-diagonal system of identical plates, 248 elements
each), assembled in one call through the front end (size-class launches of the ragged basis, nh_rationalize, NH_GEOM_TAB).  Parity: every diagonal block
equals the reference's matrix of the fixture (index arrays bit-exact, values to 1e-13).  python tools/ragged_probe.py [copies] [steps]'''
import os, sys, time
sys.path.insert(0, '.')
import numpy
import torch
from nutils_amd import function, topology, basis as _basis, device, sample as _sample, _lib

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = numpy.load(os.path.join('tests', 'golden', 'iga_plate_p3_l10.npz'))
off1 = g['dof_offsets']
ne1, nd1 = len(off1) - 1, int(g['ndofs'])
tile = lambda a: numpy.concatenate([a] * copies, axis=0)
topo = topology.ElementList(tile(g['elem_origin']), tile(g['elem_size']))
smp = topo.sample('gauss', 8)
coeffs1 = [g['coeffs'][a:b] for a, b in zip(off1, off1[1:])]
dofs1 = [g['dofs'][a:b] for a, b in zip(off1, off1[1:])]
hb = topo.plain_basis(coeffs1 * copies, [d + c * nd1 for c in range(copies) for d in dofs1], nd1 * copies)
size = tile(g['elem_size'])
nurbs = _basis.RationalBasis(hb, tile(g['weights']), W=tile(g['W']), dW=tile(g['dW_dparam']) * size[:, None, :])
geom = function.TabulatedGeometry(tile(g['x']), tile(g['dx_dparam']) * size[:, None, None, :])
SCALAR = os.environ.get('RAGGED_PROBE_SCALAR') == '1'  # (a scalar Laplace form on the same basis: timing only, the fixture holds the elasticity matrix)
if SCALAR:
    u, v = function.field('u', nurbs), function.field('v', nurbs)
    res = smp.integral(function.inner(function.grad(v, geom), function.grad(u, geom)) * function.J(geom))
else:
    u, v = function.field('u', nurbs, shape=[2]), function.field('v', nurbs, shape=[2])
    lam, mu = float(g['lam']), float(g['mu'])
    sigma = lam * function.div(u, geom) * function.eye(2) + 2 * mu * function.symgrad(u, geom)
    res = smp.integral(function.inner(function.grad(v, geom), sigma) * function.J(geom))
jac = function.derivative(function.derivative(res, 'v'), 'u')
plan = _sample._MatrixPlan(jac.terms)
t0 = time.perf_counter()
values, rowptr, colidx, ncols = plan.run()
torch.cuda.synchronize()
first = time.perf_counter() - t0
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
with _lib.trace() as calls:
    plan.run()
for s, e in ev:
    s.record()
    values, rowptr, colidx, ncols = plan.run()
    e.record()
torch.cuda.synchronize()
ms = sorted(s.elapsed_time(e) for s, e in ev)
if SCALAR:
    print(f'scalar Laplace on {ne1 * copies} ragged rational elements: re-assembly kernel ms min {ms[0]:.3f} median {ms[len(ms) // 2]:.3f}; entry points {sorted(set(calls))}')
    raise SystemExit(0)
# parity of every diagonal block with the reference's matrix of the fixture
v, rp, ci = device.to_host(values), device.to_host(rowptr), device.to_host(colidx)
n1, nnz1 = 2 * nd1, len(g['K_values'])
assert len(v) == nnz1 * copies and numpy.array_equal(rp, numpy.concatenate([[0]] + [g['K_rowptr'][1:] + c * nnz1 for c in range(copies)]))
assert numpy.array_equal(ci, numpy.concatenate([g['K_colidx'] + c * n1 for c in range(copies)]))
err = numpy.abs(v.reshape(copies, nnz1) - g['K_values']).max() / numpy.abs(g['K_values']).max()
assert err < 1e-13, err
nel = ne1 * copies
nb = numpy.diff(off1)
nq = len(g['gauss_weights'])
# algorithmic bytes per assembly: int32 connectivity + per-element rational tables T[nb][nq][3] (read) + tabulated geometry (x, J: 6 doubles per point) + CSR values (written once)
bytes_ = copies * (4 * int(nb.sum()) + 8 * 3 * nq * int(nb.sum()) + 8 * 6 * nq * ne1 + 8 * nnz1)
med = ms[len(ms) // 2]
print(f'{copies} plates: {nel} elements ({int(nb.min())}-{int(nb.max())} functions per element, {nq} points), {len(v)} nonzeros, max rel err vs the reference {err:.1e}')
print(f'first assembly (pattern + tables + values) {first * 1e3:.1f} ms; re-assembly kernel ms: min {ms[0]:.3f} median {med:.3f} max {ms[-1]:.3f} -> {nel / med * 1e3:.3e} elements/s, '
      f'{bytes_ / med / 1e6:.0f} GB/s of {bytes_ / 1e6:.0f} MB algorithmic = {bytes_ / med / 1e6 / 8000:.3f} of the HBM peak')
print('entry points of a re-assembly:', sorted(set(calls)))
import json
print('RESULT ' + json.dumps({'workload': f'{nel} ragged rational hierarchical elements (tests/golden/iga_plate_p3_l10.npz tiled {copies} x: p = 3 th-splines over 10 levels, {int(nb.min())}-{int(nb.max())} functions per '
                                          f'element, {nq} points, 2 components, tabulated NURBS geometry; BASELINE.json configs[4] class)', 'value': nel / med * 1e3, 'unit': 'elements/s',
                              'ms_per_step': med, 'steps': steps, 'launch': 'eager (HIP events around one assembly)', 'nnz': int(len(v)), 'max_rel_err_vs_reference': float(err),
                              'algorithmic_bytes': int(bytes_), 'hbm_frac': bytes_ / med / 1e6 / 8000, 'entry_points': sorted(set(calls))}))
