'''The seam plans (tests/golden/plans: nutils_amd.seam.match on arrays of the reference, written by tools/hip_plan.py) executed through the C ABI
(nutils_amd.seam.execute) and compared with the reference's own result stored beside each plan: CSR index arrays bit-exact (int64), values /
vectors / scalars to 1e-13 of the largest entry.  Because the plans are structural, the executor must reach the write-once kernels for the
BASELINE.json configurations that have them -- asserted on the traced C-ABI calls.'''
import pytest

import plan_exec

pytestmark = pytest.mark.gpu

GENERIC_MATRIX = {'nh_assemble_matrix', 'nh_assemble_matrix_terms'}


@pytest.mark.parametrize('name', plan_exec.names())
def test_plan_through_the_c_abi(name):
    plan, out, expect, calls = plan_exec.run_hip(name)
    plan_exec.compare(out, expect, rtol=1e-13)
    if name.startswith('c2_'):    # configs[1]: the structured P1-hex kernel, not the generic element loop
        assert 'nh_p1hex_laplace' in calls and not GENERIC_MATRIX & set(calls), calls
    if name == 'c3_p2_4x3x5_matrix':  # configs[2]: the write-once P2-hex kernel
        assert 'nh_p2hex_matrix' in calls and not GENERIC_MATRIX & set(calls), calls


def test_reexecution_reuses_the_built_plan():
    '''a plan is built once (front-end objects, device tables); a second execution with other arguments gives the other result'''
    import numpy
    from nutils_amd import seam
    plan, args, expect = plan_exec.load('c4_residual_φ')
    a = seam.execute(plan, args)
    built = plan['_built']
    b = seam.execute(plan, {k: 2. * v for k, v in args.items()})
    assert plan['_built'] is built
    assert numpy.abs(a - expect['vector']).max() < 1e-13 * numpy.abs(expect['vector']).max() and numpy.abs(a - b).max() > 1e-3


@pytest.mark.parametrize('name', plan_exec.example_names())
def test_example_plan_through_the_c_abi(name):
    '''Every distinct plan the installed seam emitted while the unit tests of nine UNMODIFIED reference examples ran (laplace, elasticity, poisson,
    platewithhole, adaptivity, cahnhilliard, drivencavity, burgers, finitestrain, cylinderflow; tools/hip_plan_capture.py in the build container), replayed through
    seam.execute = the C ABI: the half of "seam + HIP + reference in one process" that can run on the GPU box.  Boundary sides as element subsets with
    tabulated NURBS geometries, hierarchical (ragged) bases, rational bases tabulated per sample, Taylor-Hood blocks, DG projections.  Expected results: the
    reference's own (its un-hooked function.evaluate / as_csr of the array each plan was matched from), index arrays bit-exact, values to 1e-13 of the largest
    entry (vectors / scalars: plus the rounding floor of their terms, plan_exec.compare_example).'''
    from nutils_amd import seam
    plan, args, expect, later = plan_exec.load_example(name)
    out = seam.execute(plan, args)
    plan_exec.compare_example(plan, out, expect, args)
    if later is not None:  # the same built plan with the arguments of a later Newton iteration / time step
        out = seam.execute(plan, later[0])
        plan_exec.compare_example(plan, out, later[1], later[0])
