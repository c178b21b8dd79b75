'''Oracle hygiene (build container only: skipped where /root/reference is absent, e.g. on the GPU box).

  * the reference's own tests and example golden vectors pass through the import shim with oracle/poly.py standing in for the
    Rust dependency nutils_poly (oracle/run_reference_checks.py: examples + the Polyval/PolyMul/PolyGrad checks of
    tests/test_evaluable.py; the 2130 tests of tests.test_basis run in the script's full mode);
  * the committed fixtures tests/golden/*.npz are what oracle/gen_golden.py produces from the real reference today
    (index arrays identical, floats to 1e-12).
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
