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
    out = af_oracle.evaluate(seam.build(plan).integral, seam.prepare_arguments(plan, args))
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
    if plan['kind'] == 'vector':
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
    else:
        err = abs(float(out['scalar']) - float(expect['scalar'])) / abs(float(expect['scalar']))
    assert err < rtol, err
    return err
