'''Host check of the seam (nutils_amd/seam.py): every plan that the matcher wrote for arrays of the reference (tools/hip_plan.py: the unmodified
examples/laplace.py and examples/elasticity.py, and Namespace scripts for BASELINE.json configs[1..4]), evaluated on the CPU by tests/af_oracle.py,
reproduces the reference's own result stored beside it (CSR index arrays bit-exact, values to 1e-13); the plans describe structure where the
reference objects have it (structured bases, rectilinear / isoparametric geometry), so that the executor reaches the structured kernels.'''
import numpy
import pytest

import plan_exec


@pytest.mark.parametrize('name', plan_exec.names())
def test_plan_reproduces_the_reference(name):
    plan, out, expect = plan_exec.run_oracle(name)
    plan_exec.compare(out, expect, rtol=1e-13)


def test_plans_cover_the_baseline_configurations():
    names = plan_exec.names()
    plans = {n: plan_exec.load(n)[0] for n in names}
    for prefix in ('laplace_', 'elasticity_', 'c2_', 'c3_', 'c4_', 'c5_'):
        assert any(n.startswith(prefix) for n in names), prefix
    kinds = {p['kind'] for p in plans.values()}
    assert kinds == {'matrix', 'vector', 'scalar'}
    # configs[1]: the basis used as an array ('∇_i(basis_m) ∇_i(basis_n) dV' @ ns), 3-D, structural description throughout
    for n in ('c2_uniform_8_matrix', 'c2_iso_8_matrix', 'c2_iso_12x9x10_matrix'):
        p = plans[n]
        assert [t['kind'] for t in p['topos']] == ['structured'] and len(p['topos'][0]['shape']) == 3
        assert all(b['kind'] == 'structured' and b['btype'] == 'std' and b['degree'] == 1 for b in p['bases'])
        assert [g['kind'] for g in p['geoms']] == (['rectilinear'] if 'uniform' in n else ['iso'])
        assert p['args'][0]['name'] is None and p['terms'][0]['rows'] and p['terms'][0]['cols'] and not p['derivs']
        assert len(p['samples'][0]['weights']) == 8 and p['samples'][0]['elist'] is None
    # configs[2]: P2 vector field on isoparametric P1 hexahedra, Hessian of the energy
    p = plans['c3_p2_4x3x5_matrix']
    assert {(b['btype'], b['degree']) for b in p['bases']} == {('std', 1), ('std', 2)} and p['geoms'][0]['kind'] == 'iso' and p['derivs'] == ['u', 'u']
    assert p['args'][0]['ncomp'] == 3 and numpy.shape(p['terms'][0]['B']) == (3, 4, 3, 4)
    # configs[3]: polynomial coefficient functions of field values (the double-well potential), boundary terms on all four sides
    p = plans['c4_jacobian_φφ']
    assert any(t['fpoly'] is not None for t in p['terms']) and len({t['sample'] for t in p['terms']}) == 5
    # configs[4]: ragged (hierarchical) basis with offsets, rational, tabulated NURBS geometry
    p = plans['c5_nurbs_hier_p3_matrix']
    kinds = [b['kind'] for b in p['bases']]
    assert kinds == ['plain', 'rational'] and len(set(numpy.diff(p['bases'][0]['offsets']))) > 1 and p['geoms'][0]['kind'] == 'tab'
    # the Neumann term of examples/laplace.py: a boundary sample with its coefficient function cos(1) cosh(x_1) tabulated at the points
    p = plans['laplace_std1_residual']
    assert any(s['bnd_axis'] >= 0 for s in p['samples']) and any(t['scale'] is not None for t in p['terms'])


def test_plan_files_round_trip(tmp_path):
    from nutils_amd import seam
    plan, args, expect = plan_exec.load('c4_jacobian_φη')
    plan.pop('_built', None)
    seam.save(tmp_path / 'p.npz', plan, dict(expect, **{'arg_' + k: v for k, v in args.items()}))
    again, expect2 = seam.load(tmp_path / 'p.npz')
    assert again['derivs'] == plan['derivs'] and len(again['terms']) == len(plan['terms'])
    for a, b in zip(again['terms'], plan['terms']):
        for k in b:
            assert numpy.array_equal(numpy.asarray(a[k], dtype=object if a[k] is None or isinstance(a[k], dict) else None), numpy.asarray(b[k], dtype=object if b[k] is None or isinstance(b[k], dict) else None)) \
                if not isinstance(b[k], dict) else set(a[k]) == set(b[k])
    assert numpy.array_equal(expect2['values'], expect['values'])


@pytest.mark.parametrize('example', sorted({n.rsplit('_', 1)[0] for n in plan_exec.example_names()}))
def test_example_plans_on_the_cpu_evaluator(example):
    '''the plans captured from the unmodified examples (tools/hip_plan_capture.py; replayed through the C ABI by tests/test_gpu_plans.py) load, build and
    give, on the CPU evaluator, THE REFERENCE'S results stored beside them (the un-hooked function.evaluate / as_csr of the array each plan was matched from):
    af_oracle equals the same numbers the GPU replay is held to'''
    from nutils_amd import seam
    import af_oracle
    names = [n for n in plan_exec.example_names() if n.rsplit('_', 1)[0] == example]
    assert names
    del plan_exec.MARGINS[:]
    for name in names:
        plan, args, expect, later = plan_exec.load_example(name)
        for a, e in [(args, expect)] + ([later] if later else []):
            out = seam.run(plan, a, lambda integral, args, kind: af_oracle.evaluate(integral, args))
            plan_exec.compare_example(plan, out, e, a)  # (1e-13 of the reference's largest entry + 32 ulp of the sum of the |products| of an entry, stored at capture)
    if plan_exec.MARGINS:  # (shown with pytest -s / on failure: how close the comparisons came, and how much of their tolerance was the rounding floor)
        worst = max(m[1] for m in plan_exec.MARGINS)
        share = sorted(m[2] for m in plan_exec.MARGINS)
        print(f'{example}: {len(plan_exec.MARGINS)} vector / scalar comparisons, largest error / tolerance {worst:.2e}; floor / (1e-13 of the largest entry): '
              f'median {share[len(share) // 2]:.1e}, largest {share[-1]:.1e}')
