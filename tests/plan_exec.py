'''Executors of the assembly plans under tests/golden/plans (written by tools/hip_plan.py from the integrals of the unmodified
reference examples): `run_oracle` restates a plan with the numpy oracle (host check of the matcher), `run_hip` executes it through
the C ABI only (nutils_amd.kernels = ctypes wrappers; device.* = allocation / copies).'''
import os
import numpy

PLANS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'plans')


def names():
    return sorted(f[:-4] for f in os.listdir(PLANS) if f.endswith('.npz'))


def load(name):
    d = numpy.load(os.path.join(PLANS, name + '.npz'), allow_pickle=False)
    terms = []
    for i in range(int(d['nterms'])):
        pre = f't{i}_'
        terms.append({k[len(pre):]: d[k] for k in d.files if k.startswith(pre)})
    expect = {k[7:]: d[k] for k in d.files if k.startswith('expect_')}
    return str(d['kind']), int(d['ndims']), terms, expect


def _surface_factor(jac, axis):
    '''|J^-T e_axis|: the measure of the face xi_axis = const is |det J| times this'''
    Ji = numpy.linalg.inv(jac)
    return numpy.sqrt((Ji[..., axis, :] ** 2).sum(-1))


def run_oracle(name):
    from oracle import assemble as oa
    kind, nd, terms, expect = load(name)
    out = None
    for t in terms:
        pts, w, jac = t['points'], t['weights'], t['jac']
        Nt, dNt = oa.tabulate(t['test_coeffs'], pts)
        Dt, det = oa.physical_tables(Nt, dNt, jac)
        axis = int(t['bnd_axis'])
        wdet = det * w * (_surface_factor(jac, axis) if axis >= 0 else 1.) * (t['scale'] if 'scale' in t else 1.)
        fac = float(t['fac'])
        nct = int(t['test_ncomp'])
        if 'trial_dofs' in t:
            Nr, dNr = oa.tabulate(t['trial_coeffs'], pts)
            Dr, _ = oa.physical_tables(Nr, dNr, jac)
        if kind == 'matrix':
            A = oa.local_matrices(Dt, Dr, wdet, t['B'] * fac)
            v, rp, ci = oa.assemble_csr(A, t['test_dofs'], t['trial_dofs'], int(t['test_ndofs']), int(t['trial_ndofs']))
            if out is not None:
                raise NotImplementedError('several matrix terms: add the matrices')
            out = dict(values=v, rowptr=rp, colidx=ci)
        else:
            S = 1 + nd
            ne, nq = wdet.shape
            F = numpy.zeros((ne, nq, nct, S))
            if 'B' in t:
                U = oa.field_at_points(Dr, t['trial_dofs'], t['trial_value'])
                F += numpy.einsum('cadb,eqdb->eqca', t['B'], U)
            if 'L' in t:
                F += t['L']
            r = oa.assemble_vector(oa.local_vectors(Dt, wdet, F * fac), t['test_dofs'], int(t['test_ndofs']))
            out = dict(vector=r) if out is None else dict(vector=out['vector'] + r)
    return out, expect


def run_hip(name, mode='term'):
    '''mode 'term': nh_assemble_matrix / nh_assemble_vector per term; 'fused': the term-list entries nh_assemble_matrix_terms /
    nh_assemble_terms; 'gather': matrices with the owner-side reduction (NH_MATRIX_GATHER).'''
    from nutils_amd import device, kernels
    kind, nd, terms, expect = load(name)
    out = None
    S = 1 + nd
    for t in terms:
        nl, nq = t['jac'].shape[:2]
        pts = device.to_dev(t['points'], 'float64')
        w = device.to_dev(t['weights'], 'float64')

        def basis(side):
            co, dofs = t[side + '_coeffs'], t[side + '_dofs']
            nb = co.shape[1]
            T = kernels.tabulate(device.to_dev(co.reshape(nl * nb, -1), 'float64'), nl * nb, co.shape[2], pts, nq, nd)
            dd = device.to_dev(dofs.reshape(-1), 'int32')
            return kernels.basis(T, dd, nb=nb, tab=device.to_dev(numpy.arange(nl), 'int32')), dd, nb
        test, tdofs, nbt = basis('test')
        trial, rdofs, nbr = basis('trial') if 'trial_dofs' in t else (test, tdofs, nbt)
        geom = kernels.geometry_tab(device.to_dev(t['jac'], 'float64'), device.to_dev(t['x'], 'float64'), bnd_axis=int(t['bnd_axis']))
        scale = device.to_dev(t['scale'], 'float64') if 'scale' in t else None
        fac = float(t['fac'])
        nct = int(t['test_ncomp'])
        if kind == 'matrix':
            ncr = int(t['trial_ncomp'])
            pat = kernels.Pattern(nl, int(t['test_ndofs']), int(t['trial_ndofs']), tdofs, rdofs, nbt=nbt, nbr=nbr)
            rowptr, colidx = pat.expand(nct, ncr, None)
            values = device.zeros(colidx.numel(), 'float64')
            if mode == 'fused':
                kernels.assemble_matrix_terms(nelems=nl, ndims=nd, nq=nq, weights=w, geom=geom, test=test, trial=trial, nct=nct, ncr=ncr, mask=None, pattern=pat,
                                              values=values, terms=[dict(C=t['B'] * fac, scale=scale)], gather=False)
            else:
                kernels.assemble_matrix(nelems=nl, ndims=nd, nq=nq, weights=w, geom=geom, test=test, trial=trial, nct=nct, ncr=ncr, C=t['B'] * fac, mask=None,
                                        pattern=pat, values=values, scale=scale, gather=mode == 'gather')
            out = dict(values=device.to_host(values), rowptr=device.to_host(rowptr), colidx=device.to_host(colidx))
        else:
            if out is None:
                acc = device.zeros(int(t['test_ndofs']) * nct, 'float64')
                out = dict(acc=acc)
            if mode == 'fused':  # form and source of the term in ONE launch
                fields, tl = [], []
                if 'B' in t:
                    fields.append((trial, device.to_dev(t['trial_value'], 'float64'), int(t['trial_ncomp'])))
                    tl.append(dict(block=0, field=0, C=t['B'] * fac, scale=scale))
                if 'L' in t:
                    tl.append(dict(block=0, f=t['L'] * fac, scale=scale))
                kernels.assemble_terms(nelems=nl, ndims=nd, nq=nq, weights=w, geom=geom, fields=fields, blocks=[(test, nct, out['acc'])], terms=tl)
                continue
            if 'B' in t:
                kernels.assemble_vector(nelems=nl, ndims=nd, nq=nq, weights=w, geom=geom, test=test, trial=trial, nct=nct, ncr=int(t['trial_ncomp']), C=t['B'] * fac,
                                        u=device.to_dev(t['trial_value'], 'float64'), out=out['acc'], scale=scale)
            if 'L' in t:
                kernels.assemble_vector(nelems=nl, ndims=nd, nq=nq, weights=w, geom=geom, test=test, trial=test, nct=nct, ncr=nct, f=t['L'] * fac, out=out['acc'],
                                        scale=scale)
    if kind != 'matrix':
        out = dict(vector=device.to_host(out['acc']).reshape(-1, nct))
    return out, expect


def compare(out, expect, rtol=1e-13):
    if 'values' in expect:
        assert numpy.array_equal(out['rowptr'], expect['rowptr']) and numpy.array_equal(out['colidx'], expect['colidx'])
        err = numpy.abs(out['values'] - expect['values']).max() / numpy.abs(expect['values']).max()
    else:
        ref = expect['vector'].reshape(out['vector'].shape)
        err = numpy.abs(out['vector'] - ref).max() / numpy.abs(ref).max()
    assert err < rtol, err
    return err
