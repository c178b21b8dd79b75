'''The plans the seam matcher extracted from the UNMODIFIED reference examples (tools/hip_plan.py -> tests/golden/plans) executed
through the C ABI only, compared with the reference's own result stored beside each plan: CSR index arrays bit-exact, values / vectors
to 1e-13 of the largest entry.'''
import pytest

import plan_exec

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode', ['term', 'fused', 'gather'])
@pytest.mark.parametrize('name', plan_exec.names())
def test_plan_through_the_c_abi(name, mode):
    '''per-term entries, the fused term-list entries (nh_assemble_terms / nh_assemble_matrix_terms) and the deterministic owner-side reduction'''
    out, expect = plan_exec.run_hip(name, mode)
    plan_exec.compare(out, expect, rtol=1e-13)
