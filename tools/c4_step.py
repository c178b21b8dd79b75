#!/usr/bin/env python3
'''C4 Newton steps only (timeline runs); C4 probe (BASELINE.json configs[3]): Cahn-Hilliard 512^2, p=2, nonlinear residual + Jacobian re-assembly per Newton step.'''
import os, sys, time
sys.path.insert(0, '.')
import numpy, torch
from nutils_amd import mesh, function, device
from nutils_amd.solver import System

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
btype = sys.argv[2] if len(sys.argv) > 2 else 'spline'
size, eps, M, stens, wn, wp, dt = 10., 1., 1., 50., 30., 20., .5
domain, geom = mesh.rectilinear([numpy.linspace(0, size, n + 1)] * 2)
phi = domain.field('φ', btype=btype, degree=2)
phi0 = domain.field('φ0', btype=btype, degree=2)
eta = domain.field('η', btype=btype, degree=2) * (stens / eps)
p, p0 = function.value(phi), function.value(phi0)
dp = p - p0
psi = .25 * (p ** 2 - 1) ** 2
dpsi = .25 * dp ** 2 * (1 - p ** 2 + 2 * p * dp / 3 - dp ** 2 / 6)
dV = function.J(geom)
grad = lambda w: function.grad(w, geom)
nrg = domain.integral((psi + dpsi) * (stens / eps) * dV, degree=8) \
    + domain.integral(.5 * stens * eps * (grad(phi) * grad(phi)).sum(-1) * dV, degree=8) \
    - domain.integral(eta * phi * dV, degree=8) + domain.integral(eta * phi0 * dV, degree=8) \
    - domain.integral(.5 * dt * M * (grad(eta) * grad(eta)).sum(-1) * dV, degree=8) \
    + domain.boundary.integral((wp + wn) / 2 * dV, degree=4) + domain.boundary.integral((wp - wn) / 2 * phi * dV, degree=4)
system = System(nrg, trial='φ,η')
nd = len(phi.arg.basis)
rng = numpy.random.default_rng(0)
args = {'φ': rng.normal(0, .5, nd), 'φ0': rng.normal(0, .5, nd), 'η': numpy.zeros(nd)}
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = system.assemble_residual(args)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    jac = system.assemble_jacobian(args, copy=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'n={n} {btype} p=2 nelems={n*n} ndofs/field={nd} nnz={jac.core.nnz}: residual {1e3*(t1-t0):.1f} ms, jacobian (4 blocks, D2H + host block merge) {1e3*(t2-t1):.1f} ms')
steps, res_ms, jac_ms = [], 1e3 * (t1 - t0), 1e3 * (t2 - t1)
for it in range(int(os.environ.get('C4_STEPS', 3))):  # the Newton step as the reference evaluates it (solver.py:358-387): Jacobian and residual in one call
    args['φ'] = rng.normal(0, .5, nd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    jac, res = system.assemble_jacobian_residual(args, copy=False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    steps.append(1e3 * (t1 - t0))
    print(f'  assemble_jacobian_residual (residual beside the PCIe copy of the changed Jacobian entries): {1e3*(t1-t0):.1f} ms')
print(f'STEPS min {min(steps):.3f} median {sorted(steps)[len(steps)//2]:.3f} ms')
import json
print('RESULT ' + json.dumps({'workload': f'Cahn-Hilliard {n}^2, p = 2 {btype} basis, 25 Gauss points, two fields (BASELINE.json configs[3]): one Newton step = residual (2 blocks) + Jacobian (4 blocks, '
                                          'merged CSR in host memory) in one call', 'value': n * n / min(steps) * 1e3, 'unit': 'elements/s', 'ms_per_step': min(steps), 'steps': len(steps),
                              'launch': 'eager (wall clock around System.assemble_jacobian_residual, device synchronised)', 'residual_alone_ms': res_ms, 'jacobian_alone_ms': jac_ms,
                              'nnz': int(jac.core.nnz), 'ndofs_per_field': int(nd),
                              # what a step has to move: the entries of the field-dependent block (the only ones that change) cross PCIe into the host CSR, the three
                              # fields go up, the two residual blocks come down; on the device the same entries are written once and the fields read once
                              'pcie_bytes': int(8 * (len(system._dynpos) + 3 * nd + 2 * nd)), 'pcie_peak_GBs': 63.0,
                              'pcie_frac': 8 * (len(system._dynpos) + 5 * nd) / (min(steps) * 1e-3) / 63e9,
                              'algorithmic_bytes': int(8 * (len(system._dynpos) + 3 * nd + 2 * nd)),
                              'hbm_frac': 8 * (len(system._dynpos) + 5 * nd) / (min(steps) * 1e-3) / 8e12,
                              'bound': 'pcie (the changed Jacobian entries travel to the host matrix of the scipy solver every step)'}))
