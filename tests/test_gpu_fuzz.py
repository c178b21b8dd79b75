'''Differential fuzzing of the assembly path as a test: tools/fuzz_frontend.py with a fixed seed -- random integrals at the level the seam builds them
(structured / plain / ragged / rational bases, volume samples, sides, element subsets, several samples per form, point tables, field polynomials, derived
Jacobians) through the C ABI against the CPU evaluator tests/af_oracle.py.'''
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('seed', [101, 102])
def test_random_integrals_equal_the_cpu_evaluator(seed):
    out = subprocess.run([sys.executable, 'tools/fuzz_frontend.py', '40', str(seed)], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and f'40 of 40 cases' in out.stdout and f', 0 not ok (seed {seed})' in out.stdout, (out.stdout + out.stderr)[-3000:]
