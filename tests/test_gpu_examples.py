'''BASELINE.json configs[0]: examples/laplace.py semantics (/root/reference/examples/laplace.py:50-105) end to end
through the accelerated path -- volume stiffness, Neumann boundary load with a coefficient function, Dirichlet
constraints by boundary projection (System.solve_constraints), linear solve (System.solve), L2 error -- compared
with the example's own return values (cons, lhs, err) captured from the real reference (tests/golden/examples_laplace.npz).'''
import numpy
import pytest

pytestmark = pytest.mark.gpu


def laplace(nelems, btype, degree):
    from nutils_amd import mesh, function
    from nutils_amd.solver import System
    domain, geom = mesh.unitsquare(nelems, 'square')
    u = domain.field('u', btype=btype, degree=degree)
    v = domain.field('v', btype=btype, degree=degree)
    dV = function.J(geom)
    grad = lambda w: function.grad(w, geom)
    res = domain.integral((grad(v) * grad(u)).sum(-1) * dV, degree=degree * 2)
    flux = function.PointFunc(lambda x: numpy.cos(1) * numpy.cosh(x[:, 1]), geom)
    res -= domain.boundary['right'].integral(v * flux * dV, degree=degree * 2)
    g = function.PointFunc(lambda x: numpy.cosh(1) * numpy.sin(x[:, 0]), geom)
    sqr = domain.boundary['left'].integral(u * u * dV, degree=degree * 2)
    top = domain.boundary['top']
    sqr += top.integral(u * u * dV, degree=degree * 2) - 2 * top.integral(u * g * dV, degree=degree * 2) + top.integral(g * g * dV, degree=degree * 2)
    cons = System(sqr, trial='u').solve_constraints(droptol=1e-15)
    args = System(res, trial='u', test='v').solve(constrain=cons)
    uex = function.PointFunc(lambda x: numpy.sin(x[:, 0]) * numpy.cosh(x[:, 1]), geom)
    err2 = function.eval(domain.integral(u * u * dV, degree=degree * 2) - 2 * domain.integral(u * uex * dV, degree=degree * 2)
                         + domain.integral(uex * uex * dV, degree=degree * 2), args)
    return cons['u'], args['u'], err2 ** .5


@pytest.mark.parametrize('tag,nelems,btype,degree', [('default', 4, 'std', 1), ('spline', 4, 'spline', 2), ('c1', 32, 'std', 1)])
def test_laplace_example(golden, tag, nelems, btype, degree):
    g = golden('examples_laplace')
    cons, lhs, err = laplace(nelems, btype, degree)
    gc, gl, ge = g[f'laplace_{tag}_cons'], g[f'laplace_{tag}_lhs'], float(g[f'laplace_{tag}_err'])
    assert numpy.array_equal(numpy.isnan(cons), numpy.isnan(gc))
    assert numpy.nanmax(numpy.abs(cons - gc)) < 1e-12
    assert numpy.abs(lhs - gl).max() < 1e-10
    assert abs(err - ge) < 1e-9 * max(1, ge) + 1e-12
