'''Executors of the seam plans under tests/golden/plans (written by tools/hip_plan.py: nutils_amd.seam.match on arrays of the reference,
stored with their arguments and the reference's own result): `run_oracle` evaluates a plan on the CPU (tests/af_oracle.py: host check of the
matcher and of the plan format), `run_hip` through the C ABI (nutils_amd.seam.execute: the product path).'''
import os
import numpy

PLANS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'plans')


def names():
    return sorted(f[:-4] for f in os.listdir(PLANS) if f.endswith('.npz'))


def load(name):
    '''-> plan, arguments, expected result'''
    from nutils_amd import seam
    plan, expect = seam.load(os.path.join(PLANS, name + '.npz'))
    args = {k[4:]: v for k, v in expect.items() if k.startswith('arg_')}
    expect = {k: v for k, v in expect.items() if not k.startswith('arg_')}
    return plan, args, expect


def run_oracle(name):
    from nutils_amd import seam
    import af_oracle
    plan, args, expect = load(name)
    out = seam.run(plan, args, lambda integral, a, kind: af_oracle.evaluate(integral, a))
    return plan, _pack(plan, out), expect


def run_hip(name):
    from nutils_amd import seam, _lib
    plan, args, expect = load(name)
    with _lib.trace() as calls:
        out = seam.execute(plan, args)
    return plan, _pack(plan, out), expect, list(calls)


def _pack(plan, out):
    if plan['kind'] == 'matrix':
        return dict(values=out[0], rowptr=out[1], colidx=out[2])
    if plan['kind'] == 'points':
        return dict(points=numpy.asarray(out, dtype=float))
    if plan['kind'] in ('vector', 'stack'):
        return dict(vector=numpy.asarray(out, dtype=float))
    return dict(scalar=numpy.asarray(float(out)))


def compare(out, expect, rtol=1e-13):
    if 'values' in expect:
        assert numpy.array_equal(out['rowptr'], expect['rowptr']) and numpy.array_equal(out['colidx'], expect['colidx'])
        assert out['rowptr'].dtype == numpy.int64 and out['colidx'].dtype == numpy.int64
        err = numpy.abs(out['values'] - expect['values']).max() / numpy.abs(expect['values']).max()
    elif 'vector' in expect:
        ref = expect['vector']
        err = numpy.abs(out['vector'].reshape(ref.shape) - ref).max() / numpy.abs(ref).max()
    elif 'points' in expect:
        ref = expect['points']
        assert out['points'].shape == ref.shape
        err = numpy.abs(out['points'] - ref).max() / max(numpy.abs(ref).max(), 1e-300)
    else:
        err = abs(float(out['scalar']) - float(expect['scalar'])) / abs(float(expect['scalar']))
    assert err < rtol, err
    return err


# ---- plans captured from the unmodified examples (tools/hip_plan_capture.py): every distinct plan the installed seam handed to its executor while the
# examples' own unit tests ran, with the arguments of its first evaluation and THE REFERENCE'S result for the array the plan was matched from (its un-hooked
# function.evaluate / as_csr, in the same run) ----------------------------------------------------------------------------------------------------------
EXAMPLE_PLANS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'plans_examples')


def example_names():
    return sorted(f[:-4] for f in os.listdir(EXAMPLE_PLANS) if f.endswith('.npz'))


def load_example(name):
    '''-> plan, arguments, expected result, (arguments, expected result) of a later evaluation of the same plan or None'''
    from nutils_amd import seam
    plan, expect = seam.load(os.path.join(EXAMPLE_PLANS, name + '.npz'))
    args = {k[4:]: v for k, v in expect.items() if k.startswith('arg_')}
    args2 = {k[5:]: v for k, v in expect.items() if k.startswith('arg2_')}
    later = {k[:-1]: v for k, v in expect.items() if k in ('values2', 'rowptr2', 'colidx2', 'vector2', 'scalar2', 'points2', 'abssum2')}
    first = {k: v for k, v in expect.items() if k in ('values', 'rowptr', 'colidx', 'vector', 'scalar', 'points', 'abssum')}
    return plan, args, first, (args2, later) if later else None


def term_scale(plan, args):
    '''(largest term coefficient x pointwise factor x measure) x (argument scale): what a vector / scalar of this plan is made of, whatever cancels in it'''
    def measure(t):
        # what the term is integrated over, where the plan says so: sum of w |det J| (a reference-space integral -- the identity map per element -- has the number of
        # elements as its measure) and the largest value of its pointwise factor; at least 1
        g = plan['geoms'][int(t['measure'])]
        w = numpy.abs(numpy.asarray(plan['samples'][int(t['sample'])]['weights'], dtype=float))
        m = 1.
        if g['kind'] == 'tab':
            m = float((numpy.abs(numpy.linalg.det(numpy.asarray(g['jac'], dtype=float))) * w).sum())
        sc = t.get('scale')
        return max(1., m) * max(1., float(numpy.abs(numpy.asarray(sc, dtype=float)).max()) if sc is not None else 1.)
    coef = max([1.] + [abs(float(t['fac'])) * measure(t) * max([float(numpy.abs(numpy.asarray(t[k], dtype=float)).max()) for k in ('B', 'L', 'f0') if t.get(k) is not None] + [0.])
                       for t in plan['terms']])
    return coef * max([1.] + [float(numpy.abs(numpy.asarray(v, dtype=float)).max()) ** (2 if plan['kind'] == 'scalar' else 1) for v in (args or {}).values()
                              if numpy.size(v) and numpy.asarray(v).dtype.kind in 'fiub'])


def compare_example(plan, out, expect, args, rtol=1e-13, floor=32 * 2.3e-16):
    '''index arrays exact; matrix values to 1e-13 of the largest entry of the reference's result.  Vectors and scalars: to 1e-13 of the largest entry of the
    reference's result PLUS the rounding floor of a sum of terms of size term_scale (32 ulp of it): a residual at its own solution, an L2 error (the squared
    distance at the projection) or an energy difference are sums that cancel -- the reference's own value of them moves by ulps of the terms with the order of
    its additions, whatever their own size.  Returns error / tolerance.'''
    if plan['kind'] == 'matrix':
        assert numpy.array_equal(out[1], expect['rowptr']) and numpy.array_equal(out[2], expect['colidx'])
        err = numpy.abs(out[0] - expect['values']).max() / (rtol * max(numpy.abs(expect['values']).max(), 1e-300)) if len(expect['values']) else 0.
    elif plan['kind'] == 'points':
        # function values at the points of a sample (Sample.eval): the reference's shape exactly, every entry to 1e-13 of the largest entry PLUS 1e-13 of what the entry is a sum
        # of (x + u with x = O(1): the arguments' largest coefficient and 1 for the coordinates / coefficient functions) -- a difference u - uexact near zero cancels
        ref = numpy.asarray(expect['points'], dtype=float)
        mine = numpy.asarray(out, dtype=float)
        assert mine.shape == ref.shape, (mine.shape, ref.shape)
        # (a singular point of the map -- the corner of a NURBS patch -- is NaN in the reference: numeric.inv, numeric.py:221-241; NaN in the same places, the rest compared)
        assert numpy.array_equal(numpy.isnan(mine), numpy.isnan(ref))
        mine, ref = numpy.nan_to_num(mine), numpy.nan_to_num(ref)
        scale = max([numpy.abs(ref).max() if ref.size else 0., 1.] + [float(numpy.abs(numpy.asarray(v, dtype=float)).max()) for v in (args or {}).values()
                                                                    if numpy.size(v) and numpy.asarray(v).dtype.kind in 'fiub'])
        err = (numpy.abs(mine - ref).max() / (rtol * scale)) if ref.size else 0.
    else:
        ref = numpy.asarray(expect['scalar'] if plan['kind'] == 'scalar' else expect['vector'], dtype=float)  # ('stack': an array of scalar integrals, held to the bar of a vector)
        mine = numpy.asarray(out, dtype=float).reshape(ref.shape)
        if 'abssum' in expect:
            # per ENTRY: 1e-13 of the largest entry of the reference's result + 32 ulp of the sum of the |products| the entry is made of (stored at capture: the same integral
            # with every factor by its absolute value).  MARGINS records how much of each tolerance is that floor.
            ab = numpy.asarray(expect['abssum'], dtype=float).reshape(ref.shape)
            tol = rtol * numpy.abs(ref).max() + floor * ab
            err = float((numpy.abs(mine - ref) / numpy.maximum(tol, 1e-300)).max())
            MARGINS.append((plan['kind'], err, float(floor * ab.max() / max(rtol * numpy.abs(ref).max(), 1e-300))))
        else:  # (fixtures without the stored sums: the coarser bound from the plan's tensors and arguments)
            tol = rtol * numpy.abs(ref).max() + floor * term_scale(plan, args)
            err = numpy.abs(mine - ref).max() / max(tol, 1e-300)
    assert err < 1, err
    return err


MARGINS = []  # (kind, error / tolerance, floor / relative part of the tolerance) of every vector / scalar comparison with stored |product| sums
