'''Oracle hygiene (build container only: skipped where /root/reference is absent, e.g. on the GPU box).

  * the reference's own tests and example golden vectors pass through the import shim with oracle/poly.py standing in for the
    Rust dependency nutils_poly (oracle/run_reference_checks.py: examples + the Polyval/PolyMul/PolyGrad checks of
    tests/test_evaluable.py; the 2130 tests of tests.test_basis run in the script's full mode);
  * the committed fixtures tests/golden/*.npz are what oracle/gen_golden.py produces from the real reference today
    (index arrays identical, floats to 1e-13).
'''
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/src/nutils'), reason='the reference only exists in the build container')


def run(*args):
    out = subprocess.run([sys.executable] + list(args), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    return out.stdout


def test_reference_examples_and_poly_tests_pass_through_the_shim():
    out = run('oracle/run_reference_checks.py', '--quick')
    assert '0 failures, 0 errors' in out


def test_golden_fixtures_reproduce():
    out = run('oracle/gen_golden.py', '--check')
    assert ' 0 differences' in out


def test_seam_installed_in_the_reference():
    '''nutils_amd.seam.install() patches function.evaluate / as_csr / solver.System of the importable reference; with the CPU evaluator of
    tests/af_oracle.py as executor the UNMODIFIED examples' own unit tests (embedded golden vectors) pass, their Systems assembled from plans'''
    out = run('tests/seam_hook_run.py')
    for name in ('laplace', 'elasticity', 'poisson', 'platewithhole', 'adaptivity'):
        line = next(l for l in out.splitlines() if l.startswith(name + ':'))
        assert ' 0 failures, 0 errors' in line and 'from plans: 0;' not in line, line
    assert 'equal to the reference: True' in out


def test_seam_installed_examples_with_declined_systems():
    '''Navier-Stokes on Taylor-Hood elements and the discontinuous Galerkin Burgers equation (upwind flux: jumps and means across interfaces): every System of the
    selected tests is assembled from plans, the examples' own unit tests pass unchanged'''
    # (finitestrain.py is matched entirely since round 5 -- 9 Systems, 0 declined; its 27 plans are replayed by tests/test_plans_host.py -- and its Newton minimisation through the CPU
    # evaluator alone takes 45 s: not run here)
    out = run('tests/seam_hook_run.py', 'drivencavity:test_baseline', 'burgers:test_1d_p1,test_1d_p2_legendre')
    for name in ('drivencavity', 'burgers'):
        line = next(l for l in out.splitlines() if l.startswith(name + ':'))
        assert ' 0 failures, 0 errors' in line and 'from plans: 0;' not in line and line.endswith('unmatched systems: 0'), line


def test_example_plan_fixtures_reproduce():
    '''tests/golden/plans_examples is what tools/hip_plan_capture.py writes TODAY: the unit tests of the unmodified examples laplace, elasticity and poisson run again with
    the seam installed, every plan the hooks hand out -- Systems, constraint functionals, the post-processing evaluations of Sample.eval -- is captured again with the
    reference's own result and compared with the committed fixture of the same name: the plan (order-free signature: the reference visits terms in hash order), the arguments
    and every reference result to 1e-13.  (The capture of ALL ten examples takes 15 minutes through the CPU evaluator -- cylinderflow and finitestrain 8 of them --:
    `python tools/hip_plan_capture.py --check tests/golden/plans_examples`, run before the fixtures were committed.)'''
    out = run('tools/hip_plan_capture.py', '--check', 'tests/golden/plans_examples', 'laplace', 'elasticity', 'poisson')
    line = next(l for l in out.splitlines() if l.startswith('check:'))
    assert ' 0 differences' in line and int(line.split()[1]) >= 40, line


def test_plans_of_the_reference_scripts_reproduce(tmp_path):
    '''tools/hip_plan.py matches the integrals of the unmodified examples/laplace.py and examples/elasticity.py (captured where they are handed to
    solver.System) and of the Namespace scripts for BASELINE.json configs[1..4], and must give the committed plans again'''
    import numpy
    out = subprocess.run([sys.executable, 'tools/hip_plan.py'], cwd=ROOT, capture_output=True, text=True, timeout=1500,
                         env=dict(os.environ, NUTILS_AMD_PLAN_OUT=str(tmp_path)))
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    committed = os.path.join(ROOT, 'tests', 'golden', 'plans')
    names = sorted(f for f in os.listdir(committed) if f.endswith('.npz'))
    assert names == sorted(f for f in os.listdir(tmp_path) if f.endswith('.npz')) and len(names) >= 6
    for f in names:
        a, b = numpy.load(os.path.join(committed, f)), numpy.load(tmp_path / f)
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            if a[k].dtype.kind in 'iubUS':
                assert numpy.array_equal(a[k], b[k]), (f, k)
            else:
                assert numpy.allclose(a[k], b[k], rtol=1e-12, atol=1e-14), (f, k)
