'''Reference self-checks through the import shim (container-only; test infrastructure).

Pins the numpy stand-in for the reference's absent Rust dependency (oracle/poly.py, imported as `nutils_poly` through
oracle/refshim) with the reference's OWN tests and example golden vectors:
  * examples.laplace / elasticity / poisson / platewithhole / adaptivity  (embedded assertAlmostEqual64 vectors)
  * tests.test_basis           (partition of unity, polynomial reproduction, continuity, exact nnz counts: SURVEY 8c (ii))
  * tests.test_evaluable       the Polyval / PolyMul / PolyGrad checks (tests/test_evaluable.py:588-599)
The reference is imported from /root/reference where it lies; nothing is copied.

Usage:  python oracle/run_reference_checks.py [--quick]      exit status 0 iff everything passed
'''
import os
import sys
import unittest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
if not os.path.isdir(REF + '/src/nutils'):
    raise SystemExit('the reference is not present: these checks only run in the build container')
sys.path[:0] = [os.path.join(HERE, 'refshim'), REF + '/src', REF]
os.environ.setdefault('NUTILS_NPROCS', '1')
os.environ.setdefault('NUTILS_MATRIX', 'scipy')


def suite(quick):
    loader = unittest.TestLoader()
    s = unittest.TestSuite()
    for mod in ('examples.laplace', 'examples.elasticity', 'examples.poisson', 'examples.platewithhole', 'examples.adaptivity'):
        s.addTests(loader.loadTestsFromName(mod))
    # the polynomial checks of tests/test_evaluable.py (generated test classes named after the _check labels)
    import tests.test_evaluable as te
    for name in dir(te):
        if name.lower().startswith(('polyval', 'polymul', 'polygrad')) or name.startswith('check') and 'poly' in name.lower():
            obj = getattr(te, name)
            if isinstance(obj, type) and issubclass(obj, unittest.TestCase):
                s.addTests(loader.loadTestsFromTestCase(obj))
    if not quick:
        s.addTests(loader.loadTestsFromName('tests.test_basis'))
    return s


def main(quick=False):
    s = suite(quick)
    n = s.countTestCases()
    res = unittest.TextTestRunner(verbosity=0).run(s)
    print(f'{n} reference tests through the shim: {len(res.failures)} failures, {len(res.errors)} errors')
    return res.wasSuccessful() and n > 0


if __name__ == '__main__':
    raise SystemExit(0 if main('--quick' in sys.argv[1:]) else 1)
