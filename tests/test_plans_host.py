'''Host check of the seam matcher's plans (tools/hip_plan.py): every plan extracted from the unmodified reference examples,
restated with the numpy oracle, reproduces the reference's own result for the same integral (CSR index arrays bit-exact); and the form
tensors the matcher read off the reference graph equal what nutils_amd's own front end derives for the same problem.'''
import numpy
import pytest

import plan_exec


@pytest.mark.parametrize('name', plan_exec.names())
def test_plan_reproduces_the_reference(name):
    out, expect = plan_exec.run_oracle(name)
    plan_exec.compare(out, expect, rtol=1e-12)


def test_six_or_more_plans_cover_the_required_kinds():
    names = plan_exec.names()
    assert len(names) >= 6
    kinds = {n: plan_exec.load(n) for n in names}
    assert any(k[0] == 'matrix' and int(k[2][0]['test_ncomp']) == 1 for k in kinds.values())   # laplace
    assert any(k[0] == 'matrix' and int(k[2][0]['test_ncomp']) == 2 for k in kinds.values())   # elasticity
    assert any(k[0] == 'vector' and any(int(t['bnd_axis']) >= 0 and 'scale' in t for t in k[2]) for k in kinds.values())  # Neumann term with cos(1) cosh(x_1)
    assert any(k[0] == 'vector' and any('trial_value' in t for t in k[2]) for k in kinds.values())  # residual at a given u


def test_form_tensors_equal_the_front_end():
    '''the same integrands written for nutils_amd's front end (nutils_amd/function.py) give the same coefficient tensors'''
    from nutils_amd import function as af

    class B:  # a basis stand-in: the algebra only needs ndims / ndofs
        ndims, ndofs = 2, 7

    class G(af.Geometry):
        ndims = 2

        def __init__(self):
            pass
    geom = G()
    u, v = af.field('u', B(), ()), af.field('v', B(), ())
    itg = af._as_integrand((af.grad(v, geom) * af.grad(u, geom)).sum(-1) * af.J(geom))
    kind, nd, terms, _ = plan_exec.load('laplace_std1_matrix')
    assert numpy.array_equal(terms[0]['B'] * float(terms[0]['fac']), itg.B)
    lam, mu = 1., .5 / .3 - 1  # examples/elasticity.py defaults: poisson = .3
    u2 = af.field('u', B(), (2,))
    eps = af.symgrad(u2, geom)
    sigma = lam * af.div(u2, geom) * af.eye(2) + 2 * mu * eps
    E = af._as_integrand(af.inner(eps, sigma) * af.J(geom)) if hasattr(af, 'inner') else None
    kind, nd, terms, _ = plan_exec.load('elasticity_p1_matrix')
    Bp = terms[0]['B'] * float(terms[0]['fac'])
    if E is not None:
        H = E.B + numpy.moveaxis(E.B, (0, 1, 2, 3), (2, 3, 0, 1))  # second derivative of the quadratic energy
        assert numpy.allclose(Bp, H, atol=1e-15)
    assert numpy.allclose(Bp, numpy.moveaxis(Bp, (0, 1, 2, 3), (2, 3, 0, 1)))  # symmetric form
