'''Import shim (container-only, never shipped to the product path): lets the
read-only Python reference at /root/reference/src import in this container,
where its Rust dependency ``nutils_poly`` is absent.  Re-exports the oracle's
numpy restatement (oracle/poly.py).  Used only by oracle/gen_golden.py and
oracle/run_reference_checks.py.'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.poly import (MulVar, MulPlan, GradPlan, eval_outer, eval, degree, ncoeffs, mul, grad,  # noqa: F401
                         mul_different_vars, mul_same_vars, change_degree, composition_with_inner_matrix)
