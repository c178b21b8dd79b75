#!/usr/bin/env python3
'''Vector-valued blocks through the any-mesh entry: n^3 trilinear hexahedra (perturbed vertices), 3-component linear elasticity -- the owner kernel of
nh_owner.hip (NH_MATRIX_FUSED: one pass, Gram sums per scalar entry from D tables in LDS) against the thread pass + gather it replaces as the default
(NUTILS_AMD_NO_FUSED=1).  Parity: entry by entry against the gather path (the reference's order of the sums), two assemblies bit-identical.
python tools/vector_probe.py [n] [steps]'''
import os, sys, json
os.environ['NUTILS_AMD_NO_FAST_PATH'] = '1'
sys.path.insert(0, '.')
import numpy
import torch
from nutils_amd import mesh, function, sample, device, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = numpy.random.default_rng(0)
domain, geom0 = mesh.rectilinear([n] * 3)
gb = domain.basis('std', degree=1)
verts = numpy.stack(numpy.meshgrid(*[numpy.arange(n + 1.)] * 3, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-.2, .2, (len(gb), 3))
X = gb @ verts
u = domain.field('u', btype='std', degree=1, shape=[3])
v = domain.field('v', btype='std', degree=1, shape=[3])
eps = lambda w: function.symgrad(w, X)
sigma = function.div(u, X) * function.eye(3) + 1.3 * eps(u)
K = function.derivative(function.derivative(domain.integral(function.inner(eps(v), sigma) * function.J(X), degree=2), 'v'), 'u')


def timed(plan):
    plan.run({})
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    with _lib.trace() as calls:
        out = plan.run({})
    for s, e in ev:
        s.record()
        out = plan.run({})
        e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    return out, ms[len(ms) // 2], sorted(set(calls))


plan = sample._MatrixPlan(K.terms)
out, ms, calls = timed(plan)
again = plan.run({})
torch.cuda.synchronize()
same = bool(torch.equal(out[0], again[0]))
pat = plan.smp0.pattern(plan.test.basis, plan.trial.basis)
info = dict(zip(('blocks', 'rows_per_block', 'element_visits', 'chunks'), pat.owner_info()))
vals = device.to_host(out[0])
os.environ['NUTILS_AMD_NO_FUSED'] = '1'
plan2 = sample._MatrixPlan(K.terms)
out2, ms2, calls2 = timed(plan2)
err = float(numpy.abs(vals - device.to_host(out2[0])).max() / numpy.abs(vals).max())
ne, nnz = n ** 3, out[0].numel()
# algorithmic bytes: int32 connectivity (geometry + basis share it) + unique vertex coordinates + CSR values written once
bytes_ = 4 * 8 * ne + 24 * (n + 1) ** 3 + 8 * nnz
print(f'{n}^3 trilinear elasticity, nnz {nnz}: owner kernel {ms:.3f} ms ({bytes_ / ms / 1e6:.0f} GB/s of {bytes_ / 1e6:.0f} MB algorithmic), thread pass + gather {ms2:.3f} ms; '
      f'max rel difference {err:.1e}, repeated assembly bit-identical: {same}; plan {info}')
assert err < 1e-13 and same and info['blocks'] > 0, (err, same, info)
print('RESULT ' + json.dumps({'workload': f'3D linear elasticity stiffness, {n}^3 trilinear hexahedra with perturbed vertices, 3 components, 2x2x2 Gauss, through the any-mesh entry '
                                          '(nh_assemble_matrix, NH_MATRIX_FUSED | NH_MATRIX_STORE: owner kernel for vector-valued blocks)', 'value': ne / ms * 1e3, 'unit': 'elements/s',
                              'ms_per_step': ms, 'steps': steps, 'launch': 'eager (HIP events around one assembly)', 'nnz': int(nnz), 'algorithmic_bytes': int(bytes_),
                              'hbm_frac': bytes_ / ms / 1e6 / 8000, 'thread_pass_plus_gather_ms': ms2, 'max_rel_diff_vs_gather': err, 'bit_reproducible': same, 'owner_plan': info,
                              'entry_points': calls}))
