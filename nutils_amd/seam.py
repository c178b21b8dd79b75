'''Seam B of the reference: between `function.evaluate` / `solver.System` and `evaluable.compile`.

What a Nutils-side integration does where the reference hands its arrays to the evaluator
(/root/reference/src/nutils/function.py:2427-2429 `evaluate`; evaluable.py:6532 `compile`; solver.py:189-431 `System`):

  1. `match(array)` inspects the FUNCTION-LEVEL tree of the array -- `sample._Integral` (sample.py:944-956), `_Derivative` (function.py:1184),
     `_Gradient` (:1207), `_Jacobian` (:1266), `_Wrapper` around evaluable.add / multiply / Sum / InsertAxis / TakeDiag / power ... (:989,
     :3202-3449), `_Transpose` (:1076), `Argument` (:1030), `Basis` (:2700-3100) used through `function.field` (:2598-2627) OR directly as an array
     (`'∇_i(basis_m) ∇_i(basis_n) dV' @ ns`, expression_v2.py:668) -- expands the integrand into monomials that are multilinear in basis functions /
     fields with constant coefficient tensors, and writes an assembly PLAN.  The plan describes the problem STRUCTURALLY wherever the reference
     objects allow it: a structured topology by its shape, a `StructuredBasis` by (btype, degree, periodic axes), a rectilinear geometry by
     (offset, scale), an isoparametric geometry `gbasis @ verts` by its basis and vertex array, a Gauss sample by points and weights; only what
     has no such description (NURBS maps, hierarchical bases, coefficient functions of x) is tabulated: per-element tables through the public
     accessors `Basis.get_dofs / get_coefficients` (function.py:2794-2837), point data through `Sample.eval`.  Anything outside the class raises
     `Unmatched`: the caller falls back to the reference path.
  2. `execute(plan, arguments)` rebuilds the problem with nutils_amd's own front end (mesh / basis / sample / function objects) and evaluates it
     through the C ABI.  Because the plan is structural, the front end selects the same kernels as for a script written against nutils_amd
     directly: nh_p1hex_laplace for the trilinear Laplace form, nh_p2hex_matrix for quadratic hexahedra, the fused term lists for Newton steps,
     the generic kernels for everything else.  No CPU fallback.
  3. `install()` puts both at the seam of an importable reference: `nutils.function.evaluate` and `nutils.function.as_csr` route matched arrays
     through `execute`, `nutils.solver.System` gets the compiled callables of its cache (solver.py:321-386) replaced by plan executions;
     unmatched arrays take the reference's own evaluator.  Opt-in, like `matrix.backend`.

The matcher needs the reference (`import nutils`); plans and the executor do not: plans are plain data (`save` / `load`: one .npz), which is how the
GPU tests of this repository run them -- tests/golden/plans/*.npz are written in the build container by tools/hip_plan.py from reference scripts.
'''

import json

import numpy


PLAN_CACHE_SIZE = 128  # plans (and three times as many as_csr components) the installed hooks remember


class Unmatched(Exception):
    '''The array is outside the class the HIP backend assembles: use the reference path.'''


# =====================================================================================================================================
# plans: plain data
# =====================================================================================================================================

def save(path, plan, expect=None):
    '''One .npz: the nested plan with its arrays stored individually and the structure as JSON; `expect`: the reference's result for the plan.'''
    arrays = {}

    def enc(obj):
        if isinstance(obj, numpy.ndarray):
            key = f'a{len(arrays)}'
            arrays[key] = obj
            return {'__array__': key}
        if isinstance(obj, dict):
            return {k: enc(v) for k, v in obj.items() if k not in ('_built', '_source', '_failed')}  # (run-time attachments, also of the parts of a 'stack' plan)
        if isinstance(obj, (list, tuple)):
            return [enc(v) for v in obj]
        if isinstance(obj, (numpy.integer, numpy.bool_)):
            return int(obj)
        if isinstance(obj, numpy.floating):
            return float(obj)
        return obj
    spec = json.dumps(enc({k: v for k, v in plan.items() if not k.startswith('_')}))  # ('_built', '_source': run-time attachments)
    out = dict(arrays, spec=numpy.array(spec))
    for k, v in (expect or {}).items():
        out['expect_' + k] = numpy.asarray(v)
    numpy.savez_compressed(path, **out)


def load(path):
    d = numpy.load(path, allow_pickle=False)

    def dec(obj):
        if isinstance(obj, dict):
            if '__array__' in obj:
                return d[obj['__array__']]
            return {k: dec(v) for k, v in obj.items()}
        if isinstance(obj, list):
            return [dec(v) for v in obj]
        return obj
    plan = dec(json.loads(str(d['spec'])))
    expect = {k[7:]: d[k] for k in d.files if k.startswith('expect_')}
    return plan, expect


# =====================================================================================================================================
# executor: plan -> nutils_amd front end -> C ABI
# =====================================================================================================================================

class Built:
    '''The nutils_amd objects of a plan, built once (device tables live with the samples: re-evaluations of a plan reuse them, as the
    reference reuses its compiled callables, solver.py:321-331).'''

    def __init__(self, plan):
        from . import topology, basis as _basis, function, sample as _sample, points as _points
        self.plan = plan
        self.topos = []
        for t in plan['topos']:
            if t['kind'] == 'structured':
                self.topos.append(topology.StructuredTopology([int(n) for n in t['shape']], [int(i) for i in t.get('periodic', [])]))
            else:
                nel, nd = int(t['nelems']), int(t['ndims'])
                self.topos.append(topology.ElementList(numpy.zeros((nel, nd)), numpy.ones((nel, nd))))
        self.samples = []
        for s in plan['samples']:
            pts = _points.Points(numpy.asarray(s['points'], dtype=float), numpy.asarray(s['weights'], dtype=float))
            # a tensor Gauss scheme is replaced by the front end's own table of it (the reference computes its nodes by Golub-Welsch, points.py:343-355:
            # equal to rounding, and the structured kernels recognise their quadrature by identity with that table)
            n1 = round(pts.npoints ** (1. / pts.ndims)) if pts.ndims else 0
            for degree in (2 * n1 - 2, 2 * n1 - 1) if n1 >= 1 and n1 ** pts.ndims == pts.npoints else ():
                cand = _points.gauss(max(degree, 0), pts.ndims)
                if cand.coords.shape == pts.coords.shape and numpy.allclose(cand.coords, pts.coords, rtol=0, atol=1e-14) and numpy.allclose(cand.weights, pts.weights, rtol=0, atol=1e-14):
                    pts = cand
                    break
            elist = None if s.get('elist') is None else numpy.asarray(s['elist'])
            self.samples.append(_sample.Sample(self.topos[int(s['topo'])], pts, elist=elist, bnd_axis=int(s.get('bnd_axis', -1))))
        self.bases = []
        for b in plan['bases']:
            topo = self.topos[int(b['topo'])] if 'topo' in b else None
            if b['kind'] == 'structured':
                self.bases.append(topo.basis(str(b['btype']), int(b['degree'])))
            elif b['kind'] == 'plain':
                off = numpy.asarray(b['offsets'])
                dofs, coeffs = numpy.asarray(b['dofs']), numpy.asarray(b['coeffs'], dtype=float)
                self.bases.append(_basis.PlainBasis([coeffs[a:z] for a, z in zip(off, off[1:])], [dofs[a:z] for a, z in zip(off, off[1:])], int(b['ndofs']),
                                                    topo.ndims))
            elif b['kind'] == 'rational':
                W = b.get('W')
                self.bases.append(_basis.RationalBasis(self.bases[int(b['parent'])], numpy.asarray(b['weights'], dtype=float),
                                                       W=None if W is None else numpy.asarray(W, dtype=float),
                                                       dW=None if W is None else numpy.asarray(b['dW'], dtype=float)))
            else:
                raise ValueError(f'unknown basis kind {b["kind"]!r}')
        self.geoms = []
        for g in plan['geoms']:
            if g['kind'] == 'rectilinear':
                self.geoms.append(function.RectilinearGeometry(self.topos[int(g['topo'])], numpy.asarray(g['offset'], dtype=float), numpy.asarray(g['scale'], dtype=float)))
            elif g['kind'] == 'iso':
                self.geoms.append(function.IsoGeometry(self.bases[int(g['basis'])], numpy.asarray(g['verts'], dtype=float)))
            elif g['kind'] == 'tab':
                self.geoms.append(function.TabulatedGeometry(numpy.asarray(g['x'], dtype=float), numpy.asarray(g['jac'], dtype=float)))
            else:
                raise ValueError(f'unknown geometry kind {g["kind"]!r}')
        # an argument that is the concatenation of several coefficient vectors (function.vectorize: `part` = index, offset, length, total) is one field per part here,
        # named `name#index`; prepare_arguments slices the caller's array, derivatives with respect to `name` become one block per part (self.blocks)
        internal = lambda a: a['name'] if a.get('part') is None else f"{a['name']}#{int(a['part'][0])}"
        self.args = [function.Arg(self.bases[int(a['basis'])], int(a['ncomp']), internal(a)) for a in plan['args']]
        self.parts = {}
        for a in plan['args']:
            if a.get('part') is not None:
                k, off, n, total = (int(x) for x in a['part'])
                self.parts.setdefault(a['name'], {})[k] = (internal(a), off, n, total)
        self.scalar_args = sorted({a['name'] for a in plan['args'] if a.get('scalar')})  # bare scalar arguments: passed as the one coefficient of a constant basis
        # fields with CONSTANT coefficients (`nurbsbasis @ controlpoints`, `gbasis @ verts` evaluated at points: kind 'points'): bound like arguments, their values are in the plan
        self.consts = {a['name']: numpy.asarray(a['values'], dtype=float) for a in plan['args'] if a.get('values') is not None}

        # (samples of element subsets on ragged bases: the front end rewrites them as samples of their own element list, sample._SubsetView)
        terms = []
        for t in plan['terms']:
            fp = None
            if t.get('fpoly') is not None:
                p = t['fpoly']
                mono = {}
                for pw, c in zip(numpy.asarray(p['powers']).reshape(len(p['coeffs']), -1), p['coeffs']):
                    k = tuple(int(x) for x in pw)
                    mono[k] = mono.get(k, 0.) + float(c)
                fp = function.FieldPoly([self.args[int(i)] for i in p['args']], mono)
            sc = None if t.get('scale') is None else function.PointTable(numpy.asarray(t['scale'], dtype=float))
            arr = lambda k: None if t.get(k) is None else numpy.asarray(t[k], dtype=float)
            itg = function.Integrand(test=None if int(t['test']) < 0 else self.args[int(t['test'])], trial=None if int(t['trial']) < 0 else self.args[int(t['trial'])],
                                     B=arr('B'), L=arr('L'), f0=arr('f0'), geom=None if int(t['geom']) < 0 else self.geoms[int(t['geom'])],
                                     measure=self.geoms[int(t['measure'])], rows=bool(t['rows']), cols=bool(t['cols']), scale=sc, fscale=fp,
                                     pvars=[(self.args[int(a)], int(c), int(sl)) for a, c, sl in (t.get('pvars') or [])])
            terms.append((self.samples[int(t['sample'])], itg, float(t['fac'])))
        # kind 'points' (Sample.eval / Sample.bind, sample.py:192-232): one array-valued point expression per plan sample, `dest` = the rows of the result its points fill
        self.exprs = []
        if plan.get('kind') == 'points':
            shape = [int(n) for n in plan['shape'][1:]]
            by_sample = {}
            for t in plan['pterms']:
                facs = []
                for f in t['factors']:
                    if f.get('x') is not None:
                        facs.append(('x', self.geoms[int(f['x'])]))
                    else:
                        facs.append(('field', self.args[int(f['arg'])], None if int(f['geom']) < 0 else self.geoms[int(f['geom'])]))
                sc = None if t.get('scale') is None else function.PointTable(numpy.asarray(t['scale'], dtype=float))
                by_sample.setdefault(int(t['sample']), []).append(function.PointTerm(numpy.asarray(t['A'], dtype=float).reshape([int(n) for n in t['Ashape']]), facs, sc))
            for ps in plan['psamples']:
                si = int(ps['sample'])
                self.exprs.append((self.samples[si], function.PointExpr(shape, by_sample.get(si, [])), None if ps.get('dest') is None else numpy.asarray(ps['dest'], dtype=numpy.int64)))
        integral = function.Integral(terms)
        derivs = list(plan.get('derivs', []))
        self.split = any(name in self.parts for name in derivs)
        if not self.split:
            for name in derivs:
                integral = function.derivative(integral, name)
            self.integral = integral
            self.blocks = [((), integral)]
        else:
            # one block per combination of parts: (offsets along the dof axes, integral); dims: total length of every dof axis
            choices, self.dims = [], []
            for name in derivs:
                if name in self.parts:
                    ps = [self.parts[name][k] for k in sorted(self.parts[name])]
                    choices.append([(iname, off) for iname, off, n, total in ps])
                    self.dims.append(ps[0][3])
                else:
                    arg = next(a for a in self.args if a.name == name)
                    choices.append([(name, 0)])
                    self.dims.append(arg.basis.ndofs * arg.ncomp)
            self.blocks = []
            import itertools
            for combo in itertools.product(*choices):
                blk = integral
                for iname, _ in combo:
                    blk = function.derivative(blk, iname)
                if blk.terms:
                    self.blocks.append((tuple(off for _, off in combo), blk))
            self.integral = function.Integral([t for _, blk in self.blocks for t in blk.terms])  # (for `.terms`: is anything left?  not evaluable as one)


def build(plan):
    b = plan.get('_built') if isinstance(plan, dict) else None
    if b is None:
        b = Built(plan)
        plan['_built'] = b
    return b


def _scalar_value(ast, arguments):
    '''value of an expression of scalar arguments (Matcher.scalar_ast) for the caller's arguments'''
    op = ast[0]
    if op == 'const':
        return float(ast[1])
    if op == 'arg':
        if ast[1] not in arguments:
            raise KeyError(f'argument {ast[1]!r} of a derived scalar parameter is missing')
        return float(numpy.asarray(arguments[ast[1]], dtype=float).reshape(()))
    v = [_scalar_value(a, arguments) for a in ast[1:]]
    if op == 'negative':
        return -v[0]
    return {'add': lambda a, b: a + b, 'subtract': lambda a, b: a - b, 'multiply': lambda a, b: a * b, 'divide': lambda a, b: a / b, 'power': lambda a, b: a ** b}[op](*v)


def prepare_arguments(plan, arguments):
    '''arguments as the built integral expects them: a bare scalar argument becomes the coefficient vector (length 1) of its constant basis'''
    arguments = dict(arguments or {})
    b = build(plan)
    for name, ast in (plan.get('derived') or {}).items():  # derived scalar parameters (1 / (t - t0): Matcher.divide)
        arguments[name] = _scalar_value(ast, arguments)
    for name in b.scalar_args:
        if name in arguments:
            arguments[name] = numpy.reshape(numpy.asarray(arguments[name], dtype=float), (1,))
    arguments.update(b.consts)
    for name, parts in b.parts.items():  # concatenated coefficient vectors: one field per part
        if name in arguments:
            whole = numpy.asarray(arguments[name], dtype=float).ravel()
            for iname, off, n, total in parts.values():
                if len(whole) != total:
                    raise ValueError(f'argument {name!r} has {len(whole)} coefficients, the plan expects {total}')
                arguments[iname] = whole[off:off + n]
    return arguments


# pointwise functions of one argument a point plan may carry in `post` (names of the reference's wrappers = numpy's)
_POINTWISE = {n: getattr(numpy, n) for n in ('sqrt', 'abs', 'absolute', 'exp', 'log', 'log2', 'log10', 'sin', 'cos', 'tan', 'arcsin', 'arccos', 'arctan', 'sinh', 'cosh', 'tanh',
                                              'arctanh', 'sign', 'negative', 'square', 'reciprocal')}


def run(plan, arguments, evaluator):
    '''Evaluate a plan with `evaluator(integral, arguments, kind)` -> float | array | (values, rowptr, colidx): the whole array at once, or -- when it is differentiated
    to an argument that is a concatenation of coefficient vectors -- block by block, the blocks placed at their offsets (vectors) / merged into one CSR (matrices: the
    blocks are disjoint, every row is the concatenation of its blocks' rows in column order, as matrix/__init__.py:103-151 merges the blocks of a System).'''
    kind = plan['kind']

    def posted(x):
        for name in plan.get('post') or ():
            with numpy.errstate(all='ignore'):
                x = _POINTWISE[str(name)](x)
        return x
    if kind == 'stack':  # an array of integrals without dof axes: one scalar plan per entry (None: the entry is zero)
        out = numpy.zeros([int(n) for n in plan['shape']])
        for idx, part in zip(numpy.ndindex(*out.shape), plan['parts']):
            if part is not None:
                out[idx] = float(run(part, arguments, evaluator))
        return posted(out)
    b = build(plan)
    arguments = prepare_arguments(plan, arguments)
    if kind == 'points':
        # the point axis of the result: the elements of the sample in order, each with its points (loop_concatenate, sample.py:966-975), placed by the sample's point
        # indices (_ReorderPoints: Inflate, sample.py:978-989); a sample whose elements carry different point tables arrives as one plan sample per table
        out = numpy.zeros([int(n) for n in plan['shape']])
        for smp, expr, dest in b.exprs:
            vals = numpy.asarray(evaluator((smp, expr), arguments, kind), dtype=float)
            if dest is None:
                out += vals.reshape(out.shape)
            else:  # (the rows of one plan sample are distinct; terms located in different topologies arrive as different plan samples over the same rows)
                out[dest] += vals.reshape((len(dest),) + out.shape[1:])
        return posted(out)
    if not b.split:
        out = evaluator(b.integral, arguments, kind)
        return posted(numpy.float64(out)) if kind == 'scalar' and plan.get('post') else out
    if kind == 'vector':
        out = numpy.zeros(b.dims[0])
        for (off,), blk in b.blocks:
            r = numpy.asarray(evaluator(blk, arguments, kind), dtype=float).ravel()
            out[off:off + len(r)] = r
        return out
    vals, rows, cols = [], [], []
    for (roff, coff), blk in b.blocks:
        v, rp, ci = evaluator(blk, arguments, kind)
        rp = numpy.asarray(rp)
        vals.append(numpy.asarray(v, dtype=float))
        rows.append(numpy.repeat(numpy.arange(len(rp) - 1, dtype=numpy.int64), numpy.diff(rp)) + roff)
        cols.append(numpy.asarray(ci, dtype=numpy.int64) + coff)
    v, r, c = (numpy.concatenate(x) if x else numpy.zeros(0, dtype=t) for x, t in ((vals, float), (rows, numpy.int64), (cols, numpy.int64)))
    order = numpy.lexsort((c, r))
    rowptr = numpy.concatenate([[0], numpy.cumsum(numpy.bincount(r, minlength=b.dims[0]))]).astype(numpy.int64)
    return v[order], rowptr, c[order]


def execute(plan, arguments=None):
    '''Evaluate a plan through the C ABI.  kind 'matrix': (values, rowptr, colidx) as function.as_csr of the array flattened to two axes
    (function.py:2443-2452; index arrays int64); 'vector': the array in the reference's shape; 'scalar': float.'''
    from . import function
    out = run(plan, arguments, lambda integral, args, kind: integral[0].eval(integral[1], args) if kind == 'points' else
              function.eval(function.as_csr(integral) if kind == 'matrix' else integral, args))
    kind = plan['kind']
    if kind in ('matrix', 'points'):
        return out
    if kind == 'scalar':
        return float(out)
    return numpy.asarray(out, dtype=float).reshape([int(n) for n in plan['shape']])


# =====================================================================================================================================
# matcher: reference function tree -> plan   (needs `import nutils`)
# =====================================================================================================================================

def _name(node):
    f = node._lower
    return getattr(f, '__name__', None) or getattr(getattr(f, 'func', None), '__name__', repr(f))


def _kind(node):
    return type(node).__name__


def _children(node):
    t = _kind(node)
    if t == '_Wrapper':
        return list(node._args)
    if t in ('_WithoutPoints', '_Transpose'):
        return [node._arg]
    if t == '_Gradient':
        return [node._func, node._geom]
    if t == '_Jacobian':
        return [node._geom]
    if t in ('_Derivative', '_Opposite'):
        return [node._arg]
    if t == '_Integral':
        return [node._integrand]
    if t == '_Concatenate':
        return list(node.arrays)
    return []


class _ScalarBasis:
    '''stands for the "basis" of a bare scalar argument (`function.field('dt')`, shape ()): ONE function, identically one, shared by all elements --
    emitted as a plain basis of the term's home topology, the argument's value is passed as its single coefficient'''


_SCALAR = _ScalarBasis()


class _Coordinates:
    '''stands for the "basis" of a bare vector-valued function of the point that may be a GEOMETRY (`'x_i' @ ns` in Sample.eval): a pseudo-factor whose components are
    the coordinates; the emitter of a points plan describes it structurally (rectilinear map) or, failing that, tabulates its components like any coefficient function'''


_XGEOM = _Coordinates()


class _Factor:
    '''What a monomial is linear in: a basis (identity = the reference object), as exposed dof axis (`name` None), bound to a named argument, or
    bound to a constant coefficient array (`cvals` [ndofs][ncomp]: `gbasis @ verts`, `bsplinebasis @ controlweights`).'''

    def __init__(self, basis, name=None, ncomp=1, cvals=None, rational=None, part=None):
        self.basis, self.name, self.ncomp, self.cvals, self.rational = basis, name, int(ncomp), cvals, rational
        self.geom = None  # the reference geometry node the gradient slots refer to
        self.side = 0     # 1: the function is evaluated on the OPPOSITE side of an interface (function.py:1121-1133 `_Opposite`: jumps and means)
        self.part = part  # (index, offset, length, total): the argument is the concatenation of the coefficient vectors of several bases (function.vectorize), this is one of them

    def copy(self):
        f = _Factor(self.basis, self.name, self.ncomp, self.cvals, self.rational, self.part)
        f.geom = self.geom
        f.side = self.side
        if hasattr(self, 'xnode'):
            f.xnode = self.xnode
        return f


class _Mono:
    '''coefficient tensor A[free axes..., (comp, slot) per factor] x factors x pointwise scalar nodes x (measure).  `axes`: the array axes of the
    reference node in order: ('free', j) = free axis j of A, ('dof', f) = dof axis of factor f, ('cdof', f) = dof axis of factor f carrying
    per-dof constant coefficients (f.cvals) that a following Sum contracts.'''

    def __init__(self, A, axes, factors=(), pw=(), measure=None):
        self.A, self.axes, self.factors, self.pw, self.measure = numpy.asarray(A, dtype=float), list(axes), list(factors), list(pw), measure

    @property
    def nfree(self):
        return self.A.ndim - 2 * len(self.factors)

    def scaled(self, c):
        return _Mono(self.A * c, self.axes, self.factors, self.pw, self.measure)


class Matcher:
    '''One instance per matched array; collects the plan tables while walking.'''

    def __init__(self):
        import nutils.function as rf
        self.rf = rf
        self.S = None       # 1 + ndims of the integral being walked
        self.sample = None
        self._sym, self._symkeep = {}, []
        self.rename = {}
        self.derived = {}   # derived scalar parameters: name -> expression of scalar arguments (scalar_ast)
        self.points_mode = False  # match_points: functions evaluated at points (no measure, no exposed dof axes; coordinates kept as factors)

    # ---- helpers -------------------------------------------------------------------------------------------------------------------

    def has_symbols(self, node):
        '''does the subtree depend on unknowns: a named argument, or a basis whose dof axis is not contracted with a constant array?
        (`gbasis @ verts`, `bsplinebasis @ controlweights` are functions of the point; geometries under a gradient / measure do not count)'''
        rf = self.rf
        key = id(node)
        hit = self._sym.get(key)
        if hit is None:
            if isinstance(node, (rf.Argument, rf.Basis)) or (_kind(node) not in ('_Jacobian', '_Gradient') and getattr(node, 'arguments', None)):
                hit = True
            elif _kind(node) == '_Jacobian':
                hit = True  # (the measure is kept apart from coefficient functions: it is a symbol of its own)
            elif _kind(node) == '_Gradient':
                hit = self.has_symbols(node._func)
            elif self.basis_dot_constant(node):
                hit = False
            else:
                hit = any(self.has_symbols(c) for c in _children(node))
            self._sym[key] = hit
            self._symkeep.append(node)
        return hit

    def basis_dot_constant(self, node):
        '''Sum over the dof axis of (basis or rational basis) x (constant array): function.dot / matmul of a basis with coefficients'''
        if _kind(node) != '_Wrapper' or _name(node) != 'Sum':
            return False
        prod = node._args[0]
        while _kind(prod) == '_Transpose':
            prod = prod._arg
        if _kind(prod) != '_Wrapper' or _name(prod) != 'multiply':
            return False
        x, y = prod._args
        for b, c in ((x, y), (y, x)):
            bn = b
            while True:
                t = _kind(bn)
                if t == '_Transpose' or (t == '_Wrapper' and _name(bn) == 'InsertAxis'):
                    bn = _children(bn)[0]
                elif t == '_Wrapper' and _name(bn) == 'Take' and _kind(bn._args[0]) == '_Wrapper' and _name(bn._args[0]) == 'InsertAxis':
                    bn = bn._args[0]._args[0]
                else:
                    break
            if (isinstance(bn, self.rf.Basis) or (self.as_basis(bn) or (None, None))[1] is not None) and not c.spaces and not c.arguments:
                return True
        return False

    def const_value(self, node):
        '''numeric value of a subtree without spaces and arguments'''
        if _kind(node) == '_Constant':
            return numpy.asarray(node._value)
        return numpy.asarray(self.rf.eval(node))

    def strip_broadcast(self, node, ndim):
        '''undo numpy.broadcast_to wrappers (InsertAxis / transposes) around an array of rank `ndim`'''
        while getattr(node, 'ndim', 0) > ndim:
            t = _kind(node)
            if t == '_Transpose' or (t == '_Wrapper' and _name(node) == 'InsertAxis'):
                node = _children(node)[0]
            else:
                break
        return node

    # ---- conversion of integrand nodes to lists of monomials ----------------------------------------------------------------------------

    def conv(self, node):
        rf = self.rf
        t = _kind(node)
        if isinstance(node, rf.Basis):
            A = numpy.zeros((1, self.S))
            A[0, 0] = 1.
            return [_Mono(A, [('dof', 0)], [_Factor(node)])]
        if isinstance(node, rf.Argument):
            if node.shape == () and node.dtype == float:  # a scalar parameter of the integrand (examples/cahnhilliard.py:173 `dt`): coefficient polynomial
                A = numpy.zeros((1, self.S))
                A[0, 0] = 1.
                return [_Mono(A, [], [_Factor(_SCALAR, self.rename.get(node.name, node.name), 1)])]
            if len(node.shape) == 1 and node.dtype == float:
                # a coefficient vector used on its own (function.field(name, <array with a dof axis>): a basis transformed before it is contracted -- Piola maps,
                # examples/cylinderflow.py:122-125): the product with an exposed basis along the same array axis binds that basis to the argument (mul), the sum removes the axis
                return [_Mono(numpy.ones(()), [('avec', self.rename.get(node.name, node.name))])]
            raise Unmatched(f'argument {node.name!r} outside function.field')
        if t == '_Jacobian':
            return [_Mono(numpy.ones(()), [], measure=(self.strip_broadcast(node._geom, 1), node._tip_dim))]
        if not self.has_symbols(node) and not self.basis_dot_constant(node):
            # broadcast wrappers around a coefficient function are processed structurally (below), so that conv_plain sees its core
            if node.spaces and (t == '_Transpose' or (t == '_Wrapper' and _name(node) == 'InsertAxis')):
                pass
            else:
                return self.conv_plain(node)
        f = self.field(node)
        if f is not None:
            return f
        if t == '_Replace':  # function.replace_arguments(f, 'phi:phi0'): the same expression in a renamed argument
            ren = {}
            for old, new in node._replacements.items():
                if not isinstance(new, rf.Argument):
                    raise Unmatched('argument replaced by an expression')
                ren[old] = new.name
            saved, self.rename = self.rename, dict(self.rename, **ren)
            try:
                return self.conv(node._arg)
            finally:
                self.rename = saved
        if t == '_Opposite':
            # everything inside is evaluated in the element on the other side of the interface: the factors change sides (twice: back), coefficient functions of the
            # point are wrapped, so that the reference tabulates THEIR opposite values; measure and gradient geometry are the same map on both sides
            out = []
            for m in self.conv(node._arg):
                facs = []
                for f in m.factors:
                    g = f.copy()
                    if f.basis is not _SCALAR and f.cvals is None:
                        g.side = 1 - f.side
                    facs.append(g)
                out.append(_Mono(m.A, m.axes, facs, [self.rf.opposite(pn) for pn in m.pw], m.measure))
            return out
        if t == '_Gradient':
            geom = self.strip_broadcast(node._geom, 1)
            if geom.shape[-1] != self.S - 1:
                raise Unmatched('gradient with respect to a geometry of another dimension')
            return [g for m in self.conv(node._func) for g in self.grad(m, geom)]
        if t == '_Transpose':
            out = []
            for m in self.conv(node._arg):
                out.append(_Mono(m.A, [m.axes[i] for i in node._axes], m.factors, m.pw, m.measure))
            return out
        if t == '_Wrapper':
            name, args = _name(node), node._args
            if name == 'multiply':
                return self.merge([self.mul(a, b) for a in self.conv(args[0]) for b in self.conv(args[1])])
            if name == 'add':
                return self.conv(args[0]) + self.conv(args[1])
            if name == 'subtract':
                return self.conv(args[0]) + [m.scaled(-1.) for m in self.conv(args[1])]
            if name == 'negative':
                return [m.scaled(-1.) for m in self.conv(args[0])]
            if name == 'divide':
                return self.divide(node, args)
            if name == 'Sum':
                return self.merge([self.sum_last(m) for m in self.conv(args[0])])
            if name == 'InsertAxis':
                length = int(self.const_value(args[1]._arg if _kind(args[1]) == '_WithoutPoints' else args[1]))
                return [self.insert_axis(m, length) for m in self.conv(args[0])]
            if name == 'astype':
                return self.conv(args[0])
            if name == 'TakeDiag':
                return [self.take_diag(m) for m in self.conv(args[0])]
            if name == 'Take':
                return self.take(node, args)
            if name == 'power':
                if self.has_symbols(args[1]):
                    raise Unmatched('power with a non-constant exponent')
                e = self.const_value(self.strip_broadcast(args[1], 0))
                if e.ndim or e != int(e) or not 0 <= int(e) <= 8:
                    raise Unmatched(f'power {e!r}')
                base = self.conv(args[0])
                out = None
                for _ in range(int(e)):
                    out = base if out is None else [self.mul(a, b) for a in out for b in base]
                if out is None:
                    out = self.conv_plain(self.rf.Array.cast(numpy.ones(node.shape)))
                return out
            raise Unmatched(f'operation {name}')
        if t == '_Concatenate':
            parts = self.vector_parts(node)
            if parts is not None:
                # function.vectorize([basis_0, basis_1, ...]) as an array [dofs, components]: one exposed basis per part, each on its own component
                out = []
                for k, (pb, comp, off, n, total, nc) in enumerate(parts):
                    A = numpy.zeros((nc, 1, self.S))
                    A[comp, 0, 0] = 1.
                    out.append(_Mono(A, [('dof', 0), ('free', 0)], [_Factor(pb[0], None, 1, rational=pb[1], part=(k, off, n, total))]))
                return out
        raise Unmatched(f'node {t}')

    def conv_plain(self, node):
        '''subtree without bases and arguments: a constant, the measure, or a coefficient function of the point'''
        core = self.strip_broadcast(node, 0)
        if not node.spaces:
            v = numpy.asarray(self.const_value(node), dtype=float)
            return [_Mono(v, [('free', j) for j in range(v.ndim)])]
        if self.points_mode and node.ndim == 1 and node.shape[0] == self.S - 1 and node.dtype == float:
            nd = self.S - 1
            A = numpy.zeros((nd, nd, self.S))
            A[numpy.arange(nd), numpy.arange(nd), 0] = 1.
            f = _Factor(_XGEOM, None, nd)
            f.xnode = node
            return [_Mono(A, [('free', 0)], [f])]
        if core.ndim == 0:  # scalar coefficient function, broadcast over the array axes
            m = _Mono(numpy.ones(()), [], pw=[core])
            return [self.rebroadcast(m, node)]
        if node.size <= 16:
            # array-valued coefficient function (a normal, a traction vector, ...): one monomial per entry, each with its scalar function
            out = []
            for idx in numpy.ndindex(*node.shape):
                A = numpy.zeros(node.shape)
                A[idx] = 1.
                out.append(_Mono(A, [('free', j) for j in range(node.ndim)], pw=[node[idx]]))
            return out
        raise Unmatched('large array-valued coefficient function of the point')

    def rebroadcast(self, m, node):
        '''a scalar monomial broadcast to the shape of `node`'''
        for n in node.shape:
            m = self.insert_axis(m, int(n))
        return m

    def field(self, node):
        '''function.field(name, basis, shape=...) (function.py:2624-2627): Sum(multiply(transposed/appended Argument, basis))'''
        rf = self.rf
        if _kind(node) != '_Wrapper' or _name(node) != 'Sum':
            return None
        prod = node._args[0]
        transposed = False
        while _kind(prod) == '_Transpose':  # (the dof axis moved to the end for the sum: a vectorized basis [dofs, components] dotted with its argument)
            prod, transposed = prod._arg, True
        if _kind(prod) != '_Wrapper' or _name(prod) != 'multiply':
            return None

        def strip_to(n, cls):
            while not isinstance(n, cls):
                t = _kind(n)
                if t == '_Transpose' or (t == '_Wrapper' and _name(n) == 'InsertAxis'):
                    n = _children(n)[0]
                else:
                    return None
            return n
        a, b = prod._args
        for x, y in ((a, b), (b, a)):
            # function.vectorize of scalar bases (function.py:2556-2570: concatenate of kronecker-ed bases) dotted with ONE argument: the argument is the concatenation of
            # one coefficient vector per basis -- a scalar field per part, each on its own component of the result
            parts = self.vector_parts(y) if node.ndim == 1 else None
            varg = strip_to(x, rf.Argument) if parts is not None else None
            if varg is not None and len(varg.shape) == 1 and varg.shape[0] == parts[0][4]:
                nc = parts[0][5]
                out = []
                for k, (pb, comp, off, n, total, _) in enumerate(parts):
                    A = numpy.zeros((nc, 1, self.S))
                    A[comp, 0, 0] = 1.
                    out.append(_Mono(A, [('free', 0)], [_Factor(pb[0], self.rename.get(varg.name, varg.name), 1, rational=pb[1], part=(k, off, n, total))]))
                return out
        if transposed:
            return None
        arg, basis = strip_to(a, rf.Argument), self.as_basis(b)
        if arg is None or basis is None:
            arg, basis = strip_to(b, rf.Argument), self.as_basis(a)
        if arg is None or basis is None or arg.shape[0] != len(basis[0]) or len(arg.shape) > 2:
            return None
        if len(arg.shape) == 2:
            nc = int(arg.shape[1])
            A = numpy.zeros((nc, nc, self.S))
            A[numpy.arange(nc), numpy.arange(nc), 0] = 1.
            return [_Mono(A, [('free', 0)], [_Factor(basis[0], self.rename.get(arg.name, arg.name), nc, rational=basis[1])])]
        A = numpy.zeros((1, self.S))
        A[0, 0] = 1.
        return [_Mono(A, [], [_Factor(basis[0], self.rename.get(arg.name, arg.name), 1, rational=basis[1])])]

    def vector_parts(self, node):
        '''function.vectorize([basis_0, basis_1, ...]) -> [((basis, rational), component, offset, length, total length, components)] or None'''
        rf = self.rf
        if _kind(node) != '_Concatenate' or node.axis != 0 or node.ndim != 2:
            return None
        parts, off = [], 0
        for arr in node.arrays:
            if _kind(arr) != '_Wrapper' or _name(arr) != 'Inflate' or len(arr._args) != 3:
                return None
            pb = self.as_basis(arr._args[0])
            if pb is None or arr._args[0].ndim != 1:
                return None
            try:
                comp, nc = numpy.asarray(rf.eval(arr._args[1]._arg)), int(rf.eval(arr._args[2]._arg))
            except Exception:
                return None
            if comp.ndim != 0 or nc != node.shape[1]:
                return None
            n = int(arr.shape[0])
            parts.append([pb, int(comp), off, n, None, nc])
            off += n
        return [(pb, comp, o, n, off, nc) for pb, comp, o, n, _, nc in parts]

    def as_basis(self, node):
        '''(Basis, None) for a basis seen through broadcast wrappers; (Basis, (weights, W node)) for the rational form `basis * w / W`
        (examples/platewithhole.py:71-72,85); else None'''
        rf = self.rf
        while True:
            if isinstance(node, rf.Basis):
                return node, None
            t = _kind(node)
            if t == '_Transpose' or (t == '_Wrapper' and _name(node) == 'InsertAxis'):
                node = _children(node)[0]
            elif t == '_Wrapper' and _name(node) == 'divide' and node.ndim == 1:
                num, den = node._args
                W = self.strip_broadcast(den, 0)
                if W.ndim or _kind(num) != '_Wrapper' or _name(num) != 'multiply':
                    return None
                x, y = num._args
                for bnode, wnode in ((x, y), (y, x)):
                    if isinstance(bnode, rf.Basis) and not self.has_symbols(wnode) and not wnode.spaces:
                        w = numpy.asarray(self.const_value(wnode), dtype=float)
                        if w.shape == (len(bnode),) and not any(isinstance(c, rf.Argument) for c in self.walk(W)):
                            return bnode, (w, W)
                return None
            else:
                return None

    def walk(self, node):
        yield node
        for c in _children(node):
            yield from self.walk(c)

    # ---- operations on monomials -----------------------------------------------------------------------------------------------------

    def insert_axis(self, m, length):
        nf = m.nfree
        A = numpy.broadcast_to(numpy.expand_dims(m.A, nf), m.A.shape[:nf] + (length,) + m.A.shape[nf:])
        return _Mono(A, m.axes + [('free', nf)], m.factors, m.pw, m.measure)

    def constants_to_pointwise(self, m):
        '''A factor bound to a CONSTANT coefficient array that is not the geometry (`weightfunc = nurbsbasis @ weights` inside an integrand, examples/platewithhole.py:83) is a
        function of the point: one monomial per non-zero (component, slot) of the factor, the reference's own value / gradient node of the field as its pointwise factor.'''
        i = next((i for i, f in enumerate(m.factors) if f.cvals is not None), None)
        if i is None:
            return [m]
        f = m.factors[i]
        if f.rational is not None or any(k in ('dof', 'cdof') and j == i for k, j in m.axes):
            raise Unmatched('field with constant coefficients inside an integrand (other than the geometry)')
        nf = m.nfree
        val = f.basis @ f.cvals  # [ncomp]
        grad = self.rf.grad(val, f.geom) if f.geom is not None else None
        out = []
        for c in range(f.ncomp):
            for sl in range(self.S):
                A = m.A[(slice(None),) * (nf + 2 * i) + (c, sl)]
                if not numpy.abs(A).sum():
                    continue
                if sl and grad is None:
                    raise Unmatched('gradient slot of a constant field without a geometry')
                axes = [(k, j - 1 if k in ('dof', 'cdof') and j > i else j) for k, j in m.axes]
                out.extend(self.constants_to_pointwise(_Mono(A, axes, m.factors[:i] + m.factors[i + 1:], m.pw + [val[c] if sl == 0 else grad[c, sl - 1]], m.measure)))
        return out

    def sum_last(self, m):
        kind, j = m.axes[-1]
        if kind in ('cdof', 'adof'):  # basis @ constant / basis @ argument: the factor is now bound to its coefficients
            return _Mono(m.A, m.axes[:-1], m.factors, m.pw, m.measure)
        if kind != 'free':
            raise Unmatched('sum over a dof axis')
        A = m.A.sum(j)
        axes = [(k, i - 1 if k == 'free' and i > j else i) for k, i in m.axes[:-1]]
        return _Mono(A, axes, m.factors, m.pw, m.measure)

    def take_diag(self, m):
        (k1, j1), (k2, j2) = m.axes[-2:]
        if k1 != 'free' or k2 != 'free':
            raise Unmatched('diagonal over a dof axis')
        nf = m.nfree
        A = numpy.diagonal(m.A, axis1=j1, axis2=j2)          # the diagonal axis is appended
        A = numpy.moveaxis(A, -1, nf - 2)                    # ... and becomes the last free axis
        ren = {}
        for j in range(nf):
            if j not in (j1, j2):
                ren[j] = len(ren)
        axes = [(k, ren[i] if k == 'free' else i) for k, i in m.axes[:-2]] + [('free', nf - 2)]
        return _Mono(A, axes, m.factors, m.pw, m.measure)

    def take(self, node, args):
        '''Take(InsertAxis(x, 1), 0-index ...): the reshape inside `basis @ array` (function.py matmul -> dot); other takes are not matched'''
        x, idx = args[0], args[1]
        if _kind(x) == '_Wrapper' and _name(x) == 'InsertAxis' and node.shape == x._args[0].shape:
            return self.conv(x._args[0])
        # components of an array picked by constant indices along its last axis (`u_0`, `X_1` of expression_v2: function.get -> numpy.take, function.py take):
        # the same entries of the coefficient tensor
        inode = idx._arg if _kind(idx) == '_WithoutPoints' else idx
        if self.has_symbols(inode) or inode.spaces:
            raise Unmatched('Take with an index that is no constant')
        ind = numpy.asarray(self.const_value(inode))
        if ind.ndim > 1 or ind.dtype.kind not in 'iu':
            raise Unmatched('Take with an index array of rank > 1')
        if self.as_basis(x) is not None and self.as_basis(x)[1] is None and x.ndim == 1:
            # single functions of a basis, `basis[i]` (the patch indicators of a multipatch geometry: mesh.unitcircle, examples/cahnhilliard.py test_multipatchcircle): no unknown
            # left -- coefficient functions of the point, evaluated by the reference at the points of the sample like every other one (conv_plain)
            if ind.ndim == 0:
                return [_Mono(numpy.ones(()), [], pw=[node])]
            if ind.size > 16:
                raise Unmatched('large array-valued coefficient function of the point')
            out = []
            for k in range(ind.size):
                A = numpy.zeros(ind.shape)
                A[k] = 1.
                out.append(_Mono(A, [('free', 0)], pw=[node[k]]))
            return out
        out = []
        for m in self.conv(x):
            kind, j = m.axes[-1]
            if kind != 'free':
                raise Unmatched('Take along a dof axis')
            if (ind < 0).any() or (ind >= m.A.shape[j]).any():
                raise Unmatched('Take: index out of range')
            A = m.A.take(ind, axis=j)
            axes = m.axes if ind.ndim else [(k, a - 1 if k == 'free' and a > j else a) for k, a in m.axes[:-1]]
            out.append(_Mono(A, axes, m.factors, m.pw, m.measure))
        return out

    def grad(self, m, geom):
        nd = self.S - 1
        out = []
        if m.pw:
            # product rule over the coefficient functions of the point: their gradients are coefficient functions too (evaluated by the reference at the points of the sample)
            if any(numpy.abs(m.A[..., 1:]).sum() != 0 for _ in m.factors):
                raise Unmatched('second derivatives')
            nf = m.nfree
            for i, pnode in enumerate(m.pw):
                g = self.rf.grad(pnode, geom)
                for j in range(nd):
                    A = numpy.zeros(m.A.shape[:nf] + (nd,) + m.A.shape[nf:])
                    A[(slice(None),) * nf + (j,)] = m.A
                    out.append(_Mono(A, m.axes + [('free', nf)], m.factors, m.pw[:i] + [g[j]] + m.pw[i + 1:], m.measure))
        if not m.factors:
            return out  # (gradient of a constant: nothing)
        if len(m.factors) != 1:
            raise Unmatched('gradient of a product')
        if numpy.abs(m.A[..., 1:]).sum() != 0:
            raise Unmatched('second derivatives')
        if m.factors[0].basis is _XGEOM:
            raise Unmatched('gradient of a coordinate function')
        f = m.factors[0].copy()
        if f.geom is not None and f.geom is not geom:
            raise Unmatched('gradients with respect to different geometries')
        f.geom = geom
        nd, nf = self.S - 1, m.nfree
        A = numpy.zeros(m.A.shape[:nf] + (nd,) + m.A.shape[nf:])
        for j in range(nd):
            A[(slice(None),) * nf + (j, slice(None), 1 + j)] = m.A[..., 0]
        return out + [_Mono(A, m.axes + [('free', nf)], [f], m.pw, m.measure)]

    def mul(self, a, b):
        if len(a.axes) != len(b.axes):
            raise Unmatched('product of arrays of different rank (not broadcast)')
        if a.measure is not None and b.measure is not None:
            raise Unmatched('two measures in one product')
        # a dof axis against a constant that varies along it: per-dof coefficients (basis @ verts, basis * weights)
        for x, y in ((a, b), (b, a)):
            for i, (k, f) in enumerate(x.axes):
                if k == 'dof' and y.axes[i][0] == 'free' and not y.factors and self.varies(y.A, y.axes[i][1]):
                    return self.bind_constant(x, y, i)
                # per-dof coefficients twice (`bsplinebasis * controlweights / weightfunc @ controlpoints` evaluated at points: the weights, then the control points)
                if (k == 'cdof' and y.axes[i][0] == 'free' and not y.factors and not y.pw and y.measure is None and self.varies(y.A, y.axes[i][1])
                        and len(x.factors) == 1 and x.factors[0].ncomp == 1 and x.measure is None):
                    return self.bind_constant(x, y, i, again=True)
        fa, fb = len(a.factors), len(b.factors)
        # einsum labels (numpy accepts 0..51): a's free axes, b's free axes, then the (comp, slot) pairs of the factors
        la = list(range(a.nfree))
        lb = list(range(a.nfree, a.nfree + b.nfree))
        oa_, ob_ = a.nfree + b.nfree, a.nfree + b.nfree + 2 * fa
        if ob_ + 2 * fb > 52:
            raise Unmatched('product with too many axes')
        Ab, Bb = a.A, b.A
        out_axes, out_free = [], []
        drop_a, drop_b = [], []
        bind = []  # (factor of a | None, factor of b | None, argument name): an exposed basis meets a coefficient vector on its dof axis
        for (ka, ia), (kb, ib) in zip(a.axes, b.axes):
            if ka == 'free' and kb == 'free':
                lb[ib] = la[ia]                                     # elementwise
                out_free.append(la[ia])
                out_axes.append(('free', len(out_free) - 1))
            elif (ka == 'dof' and kb == 'avec') or (kb == 'dof' and ka == 'avec'):
                bind.append((ia, None, ib) if ka == 'dof' else (None, ib, ia))
                out_axes.append(('adof', ia if ka == 'dof' else ib + fa))
            elif ka in ('dof', 'cdof', 'adof') and kb == 'free':
                drop_b.append(ib)
                out_axes.append((ka, ia))
            elif kb in ('dof', 'cdof', 'adof') and ka == 'free':
                drop_a.append(ia)
                out_axes.append((kb, ib + fa))
            elif ka == 'avec' and kb == 'free':
                drop_b.append(ib)
                out_axes.append((ka, ia))
            elif kb == 'avec' and ka == 'free':
                drop_a.append(ia)
                out_axes.append((kb, ib))
            else:
                raise Unmatched('product of two dof axes (diagonal in the dofs)')
        for j in drop_a:
            if self.varies(Ab, j):
                raise Unmatched('dof axis against a varying axis')
        for j in drop_b:
            if self.varies(Bb, j):
                raise Unmatched('dof axis against a varying axis')
        Ab = Ab[tuple(0 if j in drop_a else slice(None) for j in range(a.nfree))]
        Bb = Bb[tuple(0 if j in drop_b else slice(None) for j in range(b.nfree))]
        la = [l for j, l in enumerate(la) if j not in drop_a] + list(range(oa_, oa_ + 2 * fa))
        lb = [l for j, l in enumerate(lb) if j not in drop_b] + list(range(ob_, ob_ + 2 * fb))
        # equal labels with different sizes cannot occur: the reference broadcasts explicitly
        A = numpy.einsum(Ab, la, Bb, lb, out_free + list(range(oa_, oa_ + 2 * fa)) + list(range(ob_, ob_ + 2 * fb)))
        factors = a.factors + b.factors
        for ia, ib, name in bind:
            i = ia if ia is not None else ib + fa
            if factors[i].name is not None or factors[i].cvals is not None:
                raise Unmatched('coefficient vector against a basis that is bound already')
            f = factors[i].copy()
            f.name = name
            factors = factors[:i] + [f] + factors[i + 1:]
        return _Mono(A, out_axes, factors, a.pw + b.pw, a.measure or b.measure)

    @staticmethod
    def varies(A, j):
        return A.shape[j] > 1 and A.strides[j] != 0 and bool(numpy.ptp(A, axis=j).any())

    def bind_constant(self, x, y, i, again=False):
        '''x: a bare basis (one factor, coefficient 1 on the value slot, other axes broadcast) with its dof axis at array position i;
        y: a constant array.  Result: the factor carries y as per-dof coefficients; the other array axes of y become its components.
        again: x carries scalar per-dof coefficients already (and possibly coefficient functions of the point): they multiply y.'''
        if len(x.factors) != 1 or (x.pw and not again) or x.measure is not None or (x.factors[0].cvals is not None) != again or x.factors[0].name is not None:
            raise Unmatched('constant coefficients on a composite expression')
        base = x.A[(0,) * x.nfree]
        if not (x.factors[0].ncomp == 1 and base[0, 0] == 1. and numpy.abs(base).sum() == 1. and not any(self.varies(x.A, j) for j in range(x.nfree))):
            raise Unmatched('constant coefficients on a derived basis expression')
        if any(k != 'free' for p, (k, _) in enumerate(x.axes) if p != i) or any(k != 'free' for k, _ in y.axes):
            raise Unmatched('constant coefficients: unexpected axes')
        order = [y.axes[i][1]] + [y.axes[p][1] for p in range(len(y.axes)) if p != i]
        C = numpy.transpose(y.A, order)                      # [ndofs, other axes in array order]
        other = C.shape[1:]
        nc = int(numpy.prod(other)) if other else 1
        cv = C.reshape(C.shape[0], nc)
        if again:
            cv = cv * numpy.asarray(x.factors[0].cvals, dtype=float).reshape(-1, 1)
        f = _Factor(x.factors[0].basis, None, nc, cvals=numpy.ascontiguousarray(cv), rational=x.factors[0].rational)
        A = numpy.zeros(other + (nc, self.S))
        for c, idx in enumerate(numpy.ndindex(*other)):
            A[idx + (c, 0)] = 1.
        axes, nfree = [], 0
        for p in range(len(y.axes)):
            if p == i:
                axes.append(('cdof', 0))
            else:
                axes.append(('free', nfree))
                nfree += 1
        return _Mono(A, axes, [f], x.pw if again else ())

    def divide(self, node, args):
        den = args[1]
        if not self.has_symbols(den):
            if not den.spaces:
                v = 1. / numpy.asarray(self.const_value(den), dtype=float)
                c = _Mono(v, [('free', j) for j in range(v.ndim)])
                return [self.mul(m, c) for m in self.conv(args[0])]
            core = self.strip_broadcast(den, 0)
            if core.ndim == 0:
                r = self.rebroadcast(_Mono(numpy.ones(()), [], pw=[1. / core]), den)
                return [self.mul(m, r) for m in self.conv(args[0])]
        ast = self.scalar_ast(den)
        if ast is not None:
            # division by an expression of scalar parameters only (`v du / dt`, dt = t - t0: examples/burgers.py:51-56): its reciprocal is a DERIVED scalar parameter,
            # computed from the caller's arguments when the plan runs (prepare_arguments) and used like a bare scalar argument
            A = numpy.zeros((1, self.S))
            A[0, 0] = 1.
            r = self.rebroadcast(_Mono(A, [], [_Factor(_SCALAR, self.derive(['divide', ['const', 1.], ast]), 1)]), den)
            return [self.mul(m, r) for m in self.conv(args[0])]
        rat = self.as_basis(node)
        if rat is not None and rat[1] is not None:  # rational basis used as an array
            A = numpy.zeros((1, self.S))
            A[0, 0] = 1.
            return [_Mono(A, [('dof', 0)], [_Factor(rat[0], rational=rat[1])])]
        raise Unmatched('division by an expression with unknowns')

    def merge(self, monos):
        '''Monomials that differ in their coefficient functions of the point only -- same coefficient tensor, factors, axes, measure -- are one monomial whose
        coefficient function is the sum of the products (a reference expression like its terms).  Transformed bases (Piola maps: every component of the field is a
        pointwise combination of the components of the basis) multiply the monomials of a product by the square of the dimension per factor; after the contraction over
        the components most of them coincide up to that function.'''
        if len(monos) < 8 or not any(m.pw for m in monos):
            return monos
        groups = {}
        for m in monos:
            key = (m.A.shape, m.A.tobytes(), tuple(m.axes), None if m.measure is None else (id(m.measure[0]), m.measure[1]),
                   tuple((id(f.basis), f.name, f.ncomp, f.part, None if f.rational is None else id(f.rational[1]), id(f.geom), id(f.cvals), id(getattr(f, 'xnode', None)), f.side) for f in m.factors))
            groups.setdefault(key, []).append(m)
        if len(groups) == len(monos):
            return monos
        out = []
        for ms in groups.values():
            if len(ms) == 1:
                out.append(ms[0])
                continue
            total = None
            for m in ms:
                prod = None
                for pnode in m.pw:
                    prod = pnode if prod is None else prod * pnode
                if prod is None:
                    prod = self.rf.Array.cast(1.)
                total = prod if total is None else total + prod
            m0 = ms[0]
            out.append(_Mono(m0.A, m0.axes, m0.factors, [total], m0.measure))
        return out

    def scalar_ast(self, node):
        '''[op, operands...] if `node` is an expression of scalar arguments and constants only (broadcast wrappers aside), else None'''
        rf = self.rf
        node = self.strip_broadcast(node, 0)
        if node.ndim or node.spaces:
            return None
        if isinstance(node, rf.Argument):
            return ['arg', self.rename.get(node.name, node.name)] if node.shape == () and node.dtype == float else None
        if not any(isinstance(c, rf.Argument) for c in self.walk(node)):
            return ['const', float(self.const_value(node))]
        t = _kind(node)
        if t == '_Replace':
            ren = {}
            for old, new in node._replacements.items():
                if not isinstance(new, rf.Argument):
                    return None
                ren[old] = new.name
            saved, self.rename = self.rename, dict(self.rename, **ren)
            try:
                return self.scalar_ast(node._arg)
            finally:
                self.rename = saved
        if t == '_Wrapper' and _name(node) in ('add', 'subtract', 'multiply', 'divide', 'negative', 'power'):
            ops = [self.scalar_ast(a) for a in node._args]
            if any(o is None for o in ops):
                return None
            if _name(node) == 'power' and ops[1][0] != 'const':
                return None
            return [_name(node)] + ops
        return None

    def derive(self, ast):
        import hashlib
        import json
        name = '_derived_' + hashlib.sha1(json.dumps(ast).encode()).hexdigest()[:10]
        self.derived[name] = ast
        return name

    # ---- integrals ---------------------------------------------------------------------------------------------------------------------

    def integral(self, node):
        '''sum of integrals / derivatives of integrals -> list of (sample, monomial, factor), list of derivative names'''
        t = _kind(node)
        if t == '_Integral':
            smp = node._sample
            tr = smp.transforms[0]
            self.sample = smp
            self.S = 1 + tr.todims
            return [(smp, m, 1.) for m in self.conv(node._integrand)], []
        if t == '_Derivative':
            terms, derivs = self.integral(node._arg)
            return terms, derivs + [node._var.name]
        if t == '_Wrapper':
            name = _name(node)
            if name in ('add', 'subtract'):
                ta, da = self.integral(node._args[0])
                tb, db = self.integral(node._args[1])
                if da != db:
                    raise Unmatched('sum of integrals differentiated differently')
                return ta + [(s, m, -f if name == 'subtract' else f) for s, m, f in tb], da
            if name == 'negative':
                ta, da = self.integral(node._args[0])
                return [(s, m, -f) for s, m, f in ta], da
            if name in ('multiply', 'divide'):
                x, y = node._args
                for i, (c, other) in enumerate(((x, y), (y, x))):
                    if (name == 'multiply' or i == 1) and not self.has_symbols(c) and not c.spaces and not any(_kind(n) == '_Integral' for n in self.walk(c)):
                        v = float(self.const_value(c))
                        ta, da = self.integral(other)
                        return [(s, m, f * (v if name == 'multiply' else 1. / v)) for s, m, f in ta], da
        raise Unmatched(f'integral-level node {t}')


# ---- geometry ----------------------------------------------------------------------------------------------------------------------------

class _Affine:
    '''symbolic value  const + sum_d a[.., d] xi_d + sum_d b[.., d] i_d  of a subtree built from element coordinates xi, the flat element index
    and constants (mesh.rectilinear with integer shape: `geom = f_coords + unravelled index`, mesh.py:45-52, and affine images of it)'''

    def __init__(self, c, a, b):
        self.c, self.a, self.b = c, a, b  # c[shape], a[shape + (nd,)], b[shape + (nd,)]


def _parse_affine(M, node, shape):
    '''-> _Affine or raises Unmatched.  `shape`: elements per axis (the flat index is ((i0 n1 + i1) n2 + i2) ...).'''
    rf = M.rf
    nd = len(shape)
    strides = [int(numpy.prod(shape[d + 1:])) for d in range(nd)]
    t = _kind(node)
    if t == '_TransformsCoords':
        return _Affine(numpy.zeros(nd), numpy.eye(nd), numpy.zeros((nd, nd)))
    if t == '_TransformsIndex':
        return _Affine(numpy.zeros(()), numpy.zeros((nd,)), numpy.array(strides, dtype=float))
    if not any(_kind(n) in ('_TransformsCoords', '_TransformsIndex') for n in M.walk(node)):
        if node.spaces or M.has_symbols(node):
            raise Unmatched('geometry: unknown leaf')
        v = numpy.asarray(M.const_value(node), dtype=float)
        return _Affine(v, numpy.zeros(v.shape + (nd,)), numpy.zeros(v.shape + (nd,)))
    if t == '_Transpose':
        x = _parse_affine(M, node._arg, shape)
        ax = tuple(node._axes)
        return _Affine(numpy.transpose(x.c, ax), numpy.transpose(x.a, ax + (len(ax),)), numpy.transpose(x.b, ax + (len(ax),)))
    if t != '_Wrapper':
        raise Unmatched(f'geometry node {t}')
    name, args = _name(node), node._args
    if name == 'astype':
        return _parse_affine(M, args[0], shape)
    if name in ('add', 'subtract'):
        x, y = _parse_affine(M, args[0], shape), _parse_affine(M, args[1], shape)
        s = 1. if name == 'add' else -1.
        return _Affine(x.c + s * y.c, x.a + s * y.a, x.b + s * y.b)
    if name == 'negative':
        x = _parse_affine(M, args[0], shape)
        return _Affine(-x.c, -x.a, -x.b)
    if name in ('multiply', 'divide'):
        x, y = _parse_affine(M, args[0], shape), _parse_affine(M, args[1], shape)
        if name == 'divide' or not (numpy.abs(y.a).sum() or numpy.abs(y.b).sum()):
            if numpy.abs(y.a).sum() or numpy.abs(y.b).sum():
                raise Unmatched('geometry: division by a non-constant')
            k = y.c if name == 'multiply' else 1. / y.c
            return _Affine(x.c * k, x.a * k[..., None], x.b * k[..., None])
        if numpy.abs(x.a).sum() or numpy.abs(x.b).sum():
            raise Unmatched('geometry: product of two non-constants')
        return _Affine(y.c * x.c, y.a * x.c[..., None], y.b * x.c[..., None])
    if name == 'InsertAxis':
        x = _parse_affine(M, args[0], shape)
        n = int(M.const_value(args[1]._arg if _kind(args[1]) == '_WithoutPoints' else args[1]))
        rep = lambda v, tail: numpy.broadcast_to(numpy.expand_dims(v, v.ndim - tail), v.shape[:v.ndim - tail] + (n,) + v.shape[v.ndim - tail:]).copy()
        return _Affine(rep(x.c, 0), rep(x.a, 1), rep(x.b, 1))
    if name in ('FloorDivide', 'Mod'):
        x, y = _parse_affine(M, args[0], shape), _parse_affine(M, args[1], shape)
        if x.c.ndim or numpy.abs(x.a).sum() or x.c != 0 or numpy.abs(y.a).sum() or numpy.abs(y.b).sum() or y.c.ndim:
            raise Unmatched('geometry: integer division of a non-index expression')
        m = int(y.c)
        b = numpy.zeros(nd)
        keep_range = 0
        for d in range(nd):
            s = int(x.b[d])
            if s != x.b[d] or s < 0:
                raise Unmatched('geometry: fractional index stride')
            if s and s % m == 0:
                b[d] = s // m if name == 'FloorDivide' else 0
            elif s:
                keep_range += s * (shape[d] - 1)
                b[d] = 0 if name == 'FloorDivide' else s
        if keep_range >= m:
            raise Unmatched('geometry: index expression does not split at this divisor')
        return _Affine(numpy.zeros(()), numpy.zeros(nd), b)
    if name == 'Inflate':
        x = _parse_affine(M, args[0], shape)
        idx = numpy.asarray(M.const_value(args[1]._arg if _kind(args[1]) == '_WithoutPoints' else args[1]))
        n = int(M.const_value(args[2]._arg if _kind(args[2]) == '_WithoutPoints' else args[2]))
        if x.c.ndim != idx.ndim:
            raise Unmatched('geometry: Inflate of an array')
        out = _Affine(numpy.zeros(n), numpy.zeros((n, nd)), numpy.zeros((n, nd)))
        if idx.ndim == 0:
            out.c[int(idx)], out.a[int(idx)], out.b[int(idx)] = x.c, x.a, x.b
        else:
            out.c[idx], out.a[idx], out.b[idx] = x.c, x.a, x.b
        return out
    raise Unmatched(f'geometry operation {name}')


# ---- plan emission -----------------------------------------------------------------------------------------------------------------------

class Emitter:
    '''Collects topologies / bases / samples / geometries / arguments of a plan, each described structurally when the reference object is of a
    known kind and by tables otherwise.'''

    def __init__(self, matcher):
        self.M = matcher
        self.plan = dict(topos=[], bases=[], samples=[], geoms=[], args=[], terms=[])
        self._topo, self._basis, self._sample, self._geom, self._arg = {}, {}, {}, {}, {}
        self._keep = []  # reference objects whose ids are used as keys

    # -- topology of a transforms sequence --
    def topo(self, transforms):
        key = id(transforms)
        if key not in self._topo:
            self._keep.append(transforms)
            spec = None
            if _kind(transforms) == 'StructuredTransforms' and all(getattr(ax, 'isdim', False) and ax.i == 0 for ax in transforms._axes):
                # (uniformly refined structured topologies included: the axes then count the refined elements)
                spec = dict(kind='structured', shape=[int(ax.j) for ax in transforms._axes], periodic=[i for i, ax in enumerate(transforms._axes) if ax.isperiodic],
                            _nrefine=int(transforms._nrefine))
            if spec is None:
                spec = dict(kind='list', nelems=len(transforms), ndims=int(transforms.fromdims))
            # the same structured topology reached through another transforms object (e.g. the basis keeps its own)
            for i, t in enumerate(self.plan['topos']):
                if spec['kind'] == 'structured' and t == spec:
                    self._topo[key] = i
                    break
            else:
                self.plan['topos'].append(spec)
                self._topo[key] = len(self.plan['topos']) - 1
        return self._topo[key]

    def basis_transforms(self, basis):
        return basis.index._transforms

    # -- basis --
    def basis(self, basis, rational=None, sample=None):
        '''sample: (reference sample, plan sample index) for rational bases (their weight function is tabulated at its points)'''
        key = (id(basis), None if rational is None else (id(rational[1]), id(sample[0]), sample[1]))
        if key in self._basis:
            return self._basis[key]
        self._keep.append(basis)
        if rational is not None:
            parent = self.basis(basis)
            w, Wnode = rational
            rf = self.M.rf
            smp, si = sample
            sspec = self.plan['samples'][si]
            nl, nq, ne = sspec['_nl'], len(sspec['weights']), sspec['_ne']
            xi = rf.transforms_coords(smp.spaces[0], self.basis_transforms(basis))  # coordinates of the parent element
            nd = int(self.basis_transforms(basis).fromdims)
            Wl = _point_values(smp, Wnode, sspec)
            dWl = _point_values(smp, rf.grad(Wnode, xi), sspec, tail=(nd,))
            if sspec.get('elist') is None:
                W, dW = Wl, dWl
            else:  # a sample on part of the topology (boundary faces): tables by element of the topology, the listed elements filled in
                W, dW = numpy.ones((ne, nq)), numpy.zeros((ne, nq, nd))
                W[sspec['elist']], dW[sspec['elist']] = Wl, dWl
            spec = dict(kind='rational', parent=parent, weights=numpy.asarray(w, dtype=float), W=W, dW=dW)
        else:
            spec = self.structured_basis(basis)
            if spec is None:
                tr = self.basis_transforms(basis)
                ne = len(tr)
                dofs = [numpy.asarray(basis.get_dofs(e), dtype=numpy.int64) for e in range(ne)]
                coeffs = [numpy.asarray(basis.get_coefficients(e), dtype=float) for e in range(ne)]
                if len({c.shape[1:] for c in coeffs}) != 1:  # (triangles beside squares: every element's polynomials written at the highest degree that occurs)
                    from . import poly as _poly
                    if any(c.ndim != 2 for c in coeffs):
                        raise Unmatched('basis with mixed polynomial degrees')
                    nv = int(tr.fromdims)
                    top = max(_poly.degree(nv, c.shape[1]) for c in coeffs)
                    coeffs = [_poly.change_degree(c, nv, top) for c in coeffs]
                spec = dict(kind='plain', topo=self.topo(tr), dofs=numpy.concatenate(dofs), coeffs=numpy.concatenate(coeffs, axis=0),
                            offsets=numpy.cumsum([0] + [len(d) for d in dofs]).astype(numpy.int64), ndofs=len(basis))
        if spec['kind'] == 'structured' and spec in self.plan['bases']:  # another reference object of the same structured basis (`gbasis` / `ns.basis`)
            self._basis[key] = self.plan['bases'].index(spec)
            return self._basis[key]
        self.plan['bases'].append(spec)
        self._basis[key] = len(self.plan['bases']) - 1
        return self._basis[key]

    def restricted_basis(self, basis, transforms):
        '''`basis` (of a coarser topology: the field of the level before, examples/adaptivity.py:62-63) written on the elements of `transforms`: element i lies in element
        `ie` of the basis' own topology behind an affine map (index_with_tail), its polynomials are composed with that map -- a plain basis of `transforms` with the same dofs.'''
        key = (id(basis), 'restricted', id(transforms))
        if key in self._basis:
            return self._basis[key]
        self._keep.append(basis)
        import nutils.transform as rtransform
        from . import poly as _poly
        own = self.basis_transforms(basis)
        nv = int(transforms.fromdims)
        if int(own.fromdims) != nv or int(own.todims) != int(transforms.todims):
            raise Unmatched('bases of different topologies in one term')
        corners = numpy.vstack([numpy.zeros((1, nv)), numpy.eye(nv)])
        dofs, coeffs = [], []
        for i in range(len(transforms)):
            try:
                ie, tail = own.index_with_tail(transforms[i])
            except (ValueError, KeyError, IndexError):
                raise Unmatched('bases of different topologies in one term (an element of the sample lies in no element of the basis)')
            c = numpy.asarray(basis.get_coefficients(int(ie)), dtype=float)
            if c.ndim != 2:
                raise Unmatched('basis with tensorial coefficients restricted to a finer topology')
            if tail:
                v = rtransform.apply(tail, corners)
                c = _poly.compose_affine(c, nv, (v[1:] - v[0]).T, v[0])
            dofs.append(numpy.asarray(basis.get_dofs(int(ie)), dtype=numpy.int64))
            coeffs.append(c)
        top = max(_poly.degree(nv, c.shape[1]) for c in coeffs)
        coeffs = [_poly.change_degree(c, nv, top) for c in coeffs]
        self.plan['bases'].append(dict(kind='plain', topo=self.topo(transforms), dofs=numpy.concatenate(dofs), coeffs=numpy.concatenate(coeffs, axis=0),
                                       offsets=numpy.cumsum([0] + [len(d) for d in dofs]).astype(numpy.int64), ndofs=len(basis)))
        self._basis[key] = len(self.plan['bases']) - 1
        return self._basis[key]

    def scalar_basis(self, transforms, topo=None):
        '''the one-function basis of a scalar argument on the topology `transforms` (see _ScalarBasis), or on plan topology `topo` (interface lists)'''
        key = ('$scalar', id(transforms)) if topo is None else ('$scalar', 'topo', topo)
        if key not in self._basis:
            ti = self.topo(transforms) if topo is None else topo
            ne = len(transforms) if topo is None else int(self.plan['topos'][topo]['nelems'])
            self.plan['bases'].append(dict(kind='plain', topo=ti, dofs=numpy.zeros(ne, dtype=numpy.int64), coeffs=numpy.ones((ne, 1)),
                                           offsets=numpy.arange(ne + 1, dtype=numpy.int64), ndofs=1))
            self._basis[key] = len(self.plan['bases']) - 1
        return self._basis[key]

    def structured_basis(self, basis):
        '''(btype, degree) of a reference StructuredBasis whose per-axis tables equal those of nutils_amd's own structured basis of that kind'''
        from . import basis as _basis
        if _kind(basis) != 'StructuredBasis':
            return None
        tr = self.basis_transforms(basis)
        ti = self.topo(tr)
        topo = self.plan['topos'][ti]
        if topo['kind'] != 'structured' or tuple(basis._transforms_shape) != tuple(topo['shape']):
            return None
        degs = {numpy.asarray(c).shape[1] - 1 for c in basis._coeffs}
        if len(degs) != 1:
            return None
        p = degs.pop()
        for btype in ('std', 'spline'):
            try:
                mine = _basis.StructuredBasis(topo['shape'], btype, p, topo['periodic'])
            except ValueError:
                continue
            if tuple(mine.dofs_shape) != tuple(int(n) for n in basis._dofs_shape):
                continue
            ok = all(numpy.array_equal(numpy.asarray(s) % n, numpy.asarray(ms) % n) for s, ms, n in zip(basis._start_dofs, mine.start_dofs, mine.dofs_shape))
            ok = ok and all(len(c) == len(mc) and all(numpy.allclose(numpy.asarray(x), y, rtol=0, atol=1e-13) for x, y in zip(c, mc)) for c, mc in zip(basis._coeffs, mine.axis_coeffs))
            if ok:
                return dict(kind='structured', topo=ti, btype=btype, degree=int(p))
        return None

    # -- sample: element indices in the numbering of `transforms` (the basis' topology), points in parent coordinates --
    def sample(self, smp, transforms, group=None):
        key = (id(smp), id(transforms), group)
        if key in self._sample:
            return self._sample[key]
        self._keep.append(smp)
        import nutils.transform as rtransform
        pts = smp.points
        tr = smp.transforms[0]
        ti = self.topo(transforms)
        spec = None
        if tr is transforms or (self.plan['topos'][ti]['kind'] == 'structured' and _kind(tr) == 'StructuredTransforms' and tr._nrefine == self.plan['topos'][ti]['_nrefine']
                                and tr.fromdims == tr.todims and [int(ax.j - ax.i) for ax in tr._axes] == self.plan['topos'][ti]['shape']
                                and all(ax.i == 0 for ax in tr._axes)):
            p0 = pts[0]
            if _kind(pts) == '_Uniform' or all(pts[i] == p0 for i in range(1, len(pts))):  # (else: groups of elements with the same table, below)
                spec = dict(topo=ti, points=numpy.asarray(p0.coords, dtype=float), weights=self.weights(p0), elist=None, bnd_axis=-1, _nl=len(tr), _ne=len(transforms))
        if spec is None:
            # generic: every element of the sample located in `transforms`; groups of elements that see the same points in parent coordinates
            groups = {}
            face = tr.fromdims < tr.todims and transforms is not tr  # (a boundary sample located in ITSELF has no parent elements: match_points, functions without a basis)
            corners = numpy.vstack([numpy.zeros((1, tr.fromdims)), numpy.eye(tr.fromdims)])
            for i in range(len(tr)):
                ie, tail = transforms.index_with_tail(tr[i])
                p = pts[i]
                c = rtransform.apply(tail, numpy.asarray(p.coords, dtype=float))
                nu = b''
                if face:  # reference normal of the face scaled by its measure (the generalised cross product of the face's edge vectors in the parent element), up to sign
                    v = rtransform.apply(tail, corners)
                    nu = _ext_normal((v[1:] - v[0]).T)
                    nu = (nu * (1 if nu[numpy.flatnonzero(numpy.abs(nu) > 1e-12)[0]] > 0 else -1)).round(12)
                k = (c.round(12).tobytes(), numpy.asarray(self.weights(p)).round(14).tobytes(), nu if not face else nu.tobytes())
                groups.setdefault(k, dict(ielems=[], pos=[], coords=c, weights=self.weights(p), nu=nu))
                groups[k]['ielems'].append(ie)
                groups[k]['pos'].append(i)
            specs = []
            for g in groups.values():
                axis, oblique = -1, {}
                if face:
                    const = [a for a in range(g['coords'].shape[1]) if numpy.ptp(g['coords'][:, a]) == 0 and g['coords'][0, a] in (0., 1.)]
                    if len(const) == 1:
                        axis = const[0]
                    elif tr.fromdims != tr.todims - 1:
                        raise Unmatched('cannot identify the face axis')
                    else:
                        # a face that is no coordinate plane of its parent (the hypotenuse of a triangle): dS = w |det J| |J^-T nu| -- the kernels take the volume
                        # measure and the terms of this sample carry |J^-T nu| as a pointwise factor (match: `_bnd_normal`)
                        oblique = dict(_bnd_normal=g['nu'])
                specs.append(dict(topo=ti, points=g['coords'], weights=g['weights'], elist=numpy.array(g['ielems'], dtype=numpy.int64), bnd_axis=axis,
                                  _nl=len(g['ielems']), _ne=len(transforms), _pos=numpy.array(g['pos'], dtype=numpy.int64), **oblique))
            idx = []
            for s in specs:
                self.plan['samples'].append(s)
                idx.append(len(self.plan['samples']) - 1)
            self._sample[key] = idx if len(idx) > 1 else idx[0]
            return self._sample[key]
        self.plan['samples'].append(spec)
        self._sample[key] = len(self.plan['samples']) - 1
        return self._sample[key]

    # -- interfaces: a sample whose terms evaluate functions on BOTH sides (function.opposite, jump, mean: function.py:1121-1133, 1500-1600) --
    def iface_sample(self, smp, transforms):
        '''The interfaces of `smp` as the elements of a LIST topology of their own, one plan sample per group of interfaces that see the same points in the parent
        coordinates of both sides.  Element k of the list is interface `_pos[k]`; `_ie[side][k]` is its parent element in `transforms` on either side, `points` are the
        coordinates in the parent of side 0 (the side the measure, the geometry tables and the normal of the sample refer to), `_c1` those in the parent of side 1.'''
        key = (id(smp), id(transforms), 'iface')
        if key in self._sample:
            return self._sample[key]
        self._keep.append(smp)
        import nutils.transform as rtransform
        pts = smp.points
        tr0, tr1 = smp.transforms[0], smp.transforms[-1]
        if tr0.fromdims != tr0.todims - 1:
            raise Unmatched('opposite sides on a sample that is no interface')
        groups = {}
        corners = numpy.vstack([numpy.zeros((1, tr0.fromdims)), numpy.eye(tr0.fromdims)])
        for i in range(len(tr0)):
            ie0, tail0 = transforms.index_with_tail(tr0[i])
            ie1, tail1 = transforms.index_with_tail(tr1[i])
            p = pts[i]
            c0 = rtransform.apply(tail0, numpy.asarray(p.coords, dtype=float))
            c1 = rtransform.apply(tail1, numpy.asarray(p.coords, dtype=float))
            v = rtransform.apply(tail0, corners)
            nu = _ext_normal((v[1:] - v[0]).T)
            nu = (nu * (1 if nu[numpy.flatnonzero(numpy.abs(nu) > 1e-12)[0]] > 0 else -1)).round(12)
            k = (c0.round(12).tobytes(), c1.round(12).tobytes(), numpy.asarray(self.weights(p)).round(14).tobytes(), nu.tobytes())
            g = groups.setdefault(k, dict(ie0=[], ie1=[], pos=[], c0=c0, c1=c1, weights=self.weights(p), nu=nu))
            g['ie0'].append(ie0), g['ie1'].append(ie1), g['pos'].append(i)
        idx = []
        for g in groups.values():
            const = [a for a in range(g['c0'].shape[1]) if numpy.ptp(g['c0'][:, a]) == 0 and g['c0'][0, a] in (0., 1.)]
            axis, oblique = (const[0], {}) if len(const) == 1 else (-1, dict(_bnd_normal=g['nu']))
            self.plan['topos'].append(dict(kind='list', nelems=len(g['pos']), ndims=int(tr0.todims)))
            self.plan['samples'].append(dict(topo=len(self.plan['topos']) - 1, points=g['c0'], weights=g['weights'], elist=None, bnd_axis=axis, _nl=len(g['pos']), _ne=len(g['pos']),
                                             _pos=numpy.array(g['pos'], dtype=numpy.int64), _ie=[numpy.array(g['ie0']), numpy.array(g['ie1'])], _c1=g['c1'], _iface=True, **oblique))
            idx.append(len(self.plan['samples']) - 1)
        self._sample[key] = idx
        return idx

    def iface_variant(self, si, key):
        '''Interface sample si, or a copy of it, for one (test basis, trial basis) pair: the front end assembles one pair of bases per sample, the four side combinations
        of a jump against a jump are four samples over the same interfaces.'''
        variants = self._sample.setdefault(('variants', si), {})
        if key not in variants:
            if not variants:
                variants[key] = si  # (the first pair takes the sample itself)
            else:
                self.plan['samples'].append(dict(self.plan['samples'][si], _root=si))
                variants[key] = len(self.plan['samples']) - 1
        return variants[key]

    def iface_affine(self, gnode, smp, si, transforms):
        '''Per interface of sample si: (A, b) with xi_1 = A xi_0 + b, the parent coordinates of side 1 as a function of those of side 0 through the physical map --
        A = (dx/dxi_1)^-1 dx/dxi_0, both Jacobians evaluated by the reference at the points of the sample and required to be constant per interface (affine elements).'''
        key = ('affine', id(gnode), self.plan['samples'][si].get('_root', si))
        if key not in self._geom:
            rf = self.M.rf
            s = self.plan['samples'][si]
            nl, nq = s['_nl'], len(s['weights'])
            nd = int(gnode.shape[-1])
            xi = rf.transforms_coords(smp.spaces[0], transforms)
            J0 = _point_values(smp, rf.grad(gnode, xi), s, tail=(nd, nd))
            J1 = _point_values(smp, rf.opposite(rf.grad(gnode, xi)), s, tail=(nd, nd))
            scale = max(numpy.abs(J0).max(), numpy.abs(J1).max())
            if numpy.abs(J0 - J0[:, :1]).max() > 1e-12 * scale or numpy.abs(J1 - J1[:, :1]).max() > 1e-12 * scale:
                raise Unmatched('opposite sides of a curved interface (the change of parent coordinates is not affine)')
            A = numpy.linalg.solve(J1[:, 0], J0[:, 0])
            c0, c1 = numpy.asarray(s['points'], dtype=float)[0], numpy.asarray(s['_c1'], dtype=float)[0]
            b = c1 - numpy.einsum('kij,j->ki', A, c0)
            self._geom[key] = A, b
        return self._geom[key]

    def iface_basis(self, basis, side, smp, si, transforms, gnode):
        '''`basis` restricted to the parents of the interfaces of sample si on one side, as a plain basis of the interface list; side 1: the polynomials composed with the
        affine change of parent coordinates, so that they are functions of the coordinates the sample's points and geometry tables are given in.'''
        root = self.plan['samples'][si].get('_root', si)  # (variants share the bases of the sample they were copied from)
        key = (id(basis), 'iface', side, root, None if side == 0 else id(gnode))
        if key in self._basis:
            return self._basis[key]
        self._keep.append(basis)
        from . import poly as _poly
        s = self.plan['samples'][si]
        nv = int(transforms.fromdims)
        ies = s['_ie'][side]
        dofs = [numpy.asarray(basis.get_dofs(int(e)), dtype=numpy.int64) for e in ies]
        coeffs = [numpy.asarray(basis.get_coefficients(int(e)), dtype=float) for e in ies]
        if any(c.ndim != 2 for c in coeffs):
            raise Unmatched('basis with tensorial coefficients on an interface')
        top = max(_poly.degree(nv, c.shape[1]) for c in coeffs)
        coeffs = [_poly.change_degree(c, nv, top) for c in coeffs]
        if side == 1:
            A, b = self.iface_affine(gnode, smp, si, transforms)
            coeffs = [_poly.compose_affine(c, nv, A[k], b[k]) for k, c in enumerate(coeffs)]
        self.plan['bases'].append(dict(kind='plain', topo=s['topo'], dofs=numpy.concatenate(dofs), coeffs=numpy.concatenate(coeffs, axis=0),
                                       offsets=numpy.cumsum([0] + [len(d) for d in dofs]).astype(numpy.int64), ndofs=len(basis)))
        self._basis[key] = len(self.plan['bases']) - 1
        return self._basis[key]

    @staticmethod
    def weights(p):
        w = getattr(p, 'weights', None)
        return numpy.full(len(p.coords), numpy.nan) if w is None else numpy.asarray(w, dtype=float)

    # -- geometry --
    def geom(self, node, smp, si):
        key = (id(node), si)
        if key in self._geom:
            return self._geom[key]
        self._keep.append(node)
        M = self.M
        spec = None
        s = self.plan['samples'][si]
        topo = self.plan['topos'][s['topo']]
        nd = int(node.shape[-1])
        # (1) gbasis @ verts
        old = M.S
        try:
            M.S = 1 + nd
            monos = M.conv(node)
            M.S = old
            if len(monos) == 1 and len(monos[0].factors) == 1 and monos[0].factors[0].cvals is not None and not monos[0].pw and monos[0].factors[0].rational is None:
                m, f = monos[0], monos[0].factors[0]
                ident = numpy.zeros((nd, nd, 1 + nd))
                ident[numpy.arange(nd), numpy.arange(nd), 0] = 1.
                if f.ncomp == nd and m.axes == [('free', 0)] and numpy.array_equal(m.A, ident):
                    bi = self.basis(f.basis)
                    spec = dict(kind='iso', basis=bi, verts=numpy.asarray(f.cvals, dtype=float))
        except Unmatched:
            M.S = old
        # (2) affine image of the root coordinates of a structured topology
        if spec is None and topo['kind'] == 'structured' and topo.get('_nrefine', 0) == 0:
            try:
                a = _parse_affine(M, node, topo['shape'])
                if a.c.shape == (nd,) and numpy.array_equal(a.a, a.b) and numpy.array_equal(a.a, numpy.diag(numpy.diag(a.a))) and numpy.diag(a.a).all():
                    spec = dict(kind='rectilinear', topo=s['topo'], offset=a.c.copy(), scale=numpy.diag(a.a).copy())
            except Unmatched:
                pass
        # (3) anything else (NURBS maps ...): tabulated at the points of the sample by geom_tab
        if spec is None:
            raise Unmatched('geometry without a structural description')
        # the same map seen through two nodes (a scalar coordinate of a line given its axis once by the gradient, once by the measure): one geometry
        for gi, g in enumerate(self.plan['geoms']):
            if g['kind'] == spec['kind'] and g.keys() == spec.keys() and all(numpy.array_equal(g[k], spec[k]) for k in spec):
                self._geom[key] = gi
                return gi
        self.plan['geoms'].append(spec)
        self._geom[key] = len(self.plan['geoms']) - 1
        return self._geom[key]

    def geom_tab(self, node, smp, si, transforms):
        '''x and dx/dxi (xi: coordinates of the parent element in `transforms`) at the points of sample si, evaluated by the reference'''
        key = (id(node), si, 'tab')
        if key in self._geom:
            return self._geom[key]
        rf = self.M.rf
        s = self.plan['samples'][si]
        nl, nq = s['_nl'], len(s['weights'])
        nd = int(node.shape[-1])
        xi = rf.transforms_coords(smp.spaces[0], transforms)
        x = numpy.asarray(smp.eval(node), dtype=float)
        jac = numpy.asarray(smp.eval(rf.grad(node, xi)), dtype=float)
        sel = _sample_rows(smp, s)
        if sel is not None:
            x, jac = x[sel], jac[sel]
        spec = dict(kind='tab', sample=si, x=x.reshape(nl, nq, nd), jac=jac.reshape(nl, nq, nd, nd))
        self.plan['geoms'].append(spec)
        self._geom[key] = len(self.plan['geoms']) - 1
        return self._geom[key]

    def geom_unit(self, si):
        '''the identity map of every element of sample si: x = the local coordinates of the points, dx/dxi = 1 (reference-space integrals)'''
        key = ('unit', si)
        if key not in self._geom:
            s = self.plan['samples'][si]
            pts = numpy.asarray(s['points'], dtype=float)
            nl, (nq, nd) = s['_nl'], pts.shape
            self.plan['geoms'].append(dict(kind='tab', sample=si, x=numpy.broadcast_to(pts, (nl, nq, nd)).copy(), jac=numpy.broadcast_to(numpy.eye(nd), (nl, nq, nd, nd)).copy()))
            self._geom[key] = len(self.plan['geoms']) - 1
        return self._geom[key]

    def arg(self, name, bi, ncomp, part=None):
        key = (name, bi, ncomp, part) if name is not None else ('$basis', bi, ncomp)
        if key not in self._arg:
            self.plan['args'].append(dict(name=name, basis=bi, ncomp=int(ncomp), **({} if part is None else dict(part=[int(x) for x in part]))))
            self._arg[key] = len(self.plan['args']) - 1
        return self._arg[key]


def _ext_normal(T):
    '''T: (n, n - 1) edge vectors of a face as columns -> the vector orthogonal to them whose length is the measure they span (2-D: the edge turned by 90 degrees, 3-D: the cross product)'''
    n = T.shape[0]
    if T.shape != (n, n - 1):
        raise Unmatched('cannot identify the face axis')
    if n == 1:
        return numpy.ones(1)
    return numpy.array([(-1) ** i * numpy.linalg.det(numpy.delete(T, i, axis=0)) for i in range(n)])


def _point_values(smp, node, s, tail=()):
    v = numpy.asarray(smp.eval(node), dtype=float)
    nq = len(s['weights'])
    sel = _sample_rows(smp, s)
    if sel is not None:
        v = v[sel]
    return v.reshape((s['_nl'], nq) + tuple(tail))


def _sample_rows(smp, s):
    '''rows of `smp.eval(...)` that belong to the points of plan sample `s`, element by element in its list order: the sample's own point indices (sample.py:129:
    consecutive for default samples, a permutation for samples with custom indices such as `topology.locate`); None: all rows in order'''
    custom = _kind(smp) != '_DefaultIndex'
    if '_pos' not in s and not custom:
        return None
    pos = s['_pos'] if '_pos' in s else range(smp.nelems)
    rows = [numpy.asarray(smp.getindex(int(i))) for i in pos]
    return numpy.concatenate(rows) if rows else numpy.zeros(0, dtype=numpy.int64)


def match(array, arguments=None):
    '''function-level array of the reference (sum of integrals, possibly differentiated with function.derivative) -> plan.

    The result kind follows the array: rank 0 -> 'scalar'; dof axes exposed by basis arrays or derivatives: one -> 'vector' (shape of the
    reference array), two -> 'matrix' (as_csr of the array flattened to (rows, cols), rows = first exposed argument).  An array of integrals without dof axes
    (`boundary.integrate('t_i dS')`, examples/elasticity.py:72) becomes a plan of kind 'stack': one scalar plan per entry.'''
    # (Pointwise functions on top of an integral -- `numpy.sqrt(abs(domain.integral('∇_k(u_k)^2 dV')))`, examples/cylinderflow.py:148,162 -- stay with the reference: the
    # matcher expands the integrand into monomials, (d0u0)^2 + 2 d0u0 d1u1 + (d1u1)^2, whose sum cancels to rounding level -- 1e-16 of the terms, inside every bar on the
    # integral itself -- and the square root of THAT is 1e-8 where the reference's test asks for < 1e-13.  Built, measured on that example, withdrawn.)
    post, inner = [], array
    M = Matcher()
    terms, derivs = M.integral(inner)
    free = [m for _, m, _ in terms if m.axes and all(k == 'free' for k, _ in m.axes) and numpy.prod(m.A.shape[:m.nfree]) > 1]
    if free and not derivs and all(all(k == 'free' for k, _ in m.axes) for _, m, _ in terms) and 0 < inner.ndim <= 2 and int(numpy.prod(inner.shape)) <= 64:
        parts = []
        for idx in numpy.ndindex(*[int(n) for n in inner.shape]):
            sub = []
            for smp, m, fac in terms:
                sel = [0] * m.nfree
                for i, (_, j) in enumerate(m.axes):
                    sel[j] = idx[i] if m.A.shape[j] > 1 else 0
                A = m.A[tuple(sel)]
                if numpy.abs(A).sum():
                    sub.append((smp, _Mono(A, [], m.factors, m.pw, m.measure), fac))
            parts.append(_match_terms(M, sub, [], (), None) if sub else None)
        plan = dict(kind='stack', shape=[int(n) for n in inner.shape], parts=parts, derivs=[], args=[a for pl in parts if pl for a in pl['args']])
        plan['_source'] = array
    else:
        if post and (derivs or inner.ndim):
            raise Unmatched('pointwise function of a differentiated / array-valued integral')
        plan = _match_terms(M, terms, derivs, inner.shape, array)
    if post:
        if plan['kind'] not in ('scalar', 'stack'):
            raise Unmatched('pointwise function of an array with dof axes')
        plan['post'] = post[::-1]  # innermost first
    return plan


def _match_terms(M, terms, derivs, shape, source):
    '''the monomials of a sum of integrals -> plan (see match)'''
    E = Emitter(M)
    nexposed = None
    # terms without any basis (constants, coefficient functions: `sigma_wall dS`) are located in the topology of the bases seen elsewhere
    anybasis = next((f.basis for _, m, _ in terms for f in m.factors if f.basis is not _SCALAR), None)
    terms = [(smp, m2, fac) for smp, m, fac in terms for m2 in M.constants_to_pointwise(m)]
    if len(terms) >= 8:  # (monomials that differ in their coefficient functions only: Matcher.merge)
        by = {}
        for smp, m, fac in terms:
            by.setdefault((id(smp), fac), (smp, fac, []))[2].append(m)
        terms = [(smp, m, fac) for smp, fac, ms in by.values() for m in M.merge(ms)]
    # samples with a term that evaluates a function on the opposite side (jumps, means): ALL their terms are written on the list of the interfaces
    iface_smps = {id(smp) for smp, m, _ in terms if any(f.side for f in m.factors)}
    for smp, m, fac in terms:
        iface = id(smp) in iface_smps
        if any(k == 'cdof' for k, _ in m.axes):
            raise Unmatched('basis weighted per dof outside a rational form')
        # constant-bound factors that are not the geometry: pointwise functions
        facs = list(m.factors)
        exposed = [i for i, f in enumerate(facs) if f.name is None and f.cvals is None]
        bound = [i for i, f in enumerate(facs) if f.name is not None]
        cf = [i for i, f in enumerate(facs) if f.cvals is not None]
        if cf:
            raise Unmatched('field with constant coefficients inside an integrand (other than the geometry)')
        order = [ax for ax in m.axes]
        dofpos = [f for k, f in order if k == 'dof']
        if sorted(dofpos) != sorted(exposed):
            raise Unmatched('exposed basis without an array axis')
        # which bound factors stay in the form (gradients, components), which become a pointwise polynomial
        S = M.S
        A = m.A
        nf = m.nfree

        def uses_gradient(i):
            sl = [slice(None)] * A.ndim
            sl[nf + 2 * i + 1] = slice(1, None)
            return bool(numpy.abs(A[tuple(sl)]).sum())
        form = list(exposed)
        poly = []
        for i in bound:
            if facs[i].basis is _SCALAR:
                if uses_gradient(i):
                    raise Unmatched('gradient of a scalar argument')
                poly.append(i)
            elif uses_gradient(i) or facs[i].ncomp > 1 or len([f for f in facs if f.basis is not _SCALAR]) <= 2 or iface:
                form.append(i)  # (interfaces: the same argument lives on two bases -- one per side --, which a coefficient polynomial cannot name: point variables below)
            else:
                poly.append(i)
        # more than two such factors (the convection u_j d_j(u_i) v_i: v, u, u): two stay in the form -- the exposed ones, then bound ones that the array is
        # differentiated to (in that order), then the first -- the others become POINT VARIABLES, one term per non-zero (component, slot) of each: the term is
        # (constant form) x U_arg[comp][slot] at the point, which the front end evaluates on the device and differentiates by the product rule (function.Integrand.pvars)
        pv = []
        if len(form) > 2:
            keep = [i for i in form if i in exposed]
            cand = [i for i in form if i not in exposed]
            cand.sort(key=lambda i: (derivs.index(facs[i].name) if facs[i].name in derivs else len(derivs), i))
            seen_names = set()
            first = [i for i in cand if facs[i].name in derivs and not (facs[i].name in seen_names or seen_names.add(facs[i].name))]
            order_c = first + [i for i in cand if i not in first]
            keep += order_c[:max(0, 2 - len(keep))]
            if len(keep) > 2:
                raise Unmatched('more than two exposed basis factors in one term')
            pv = [i for i in form if i not in keep]
            form = keep
            if len(pv) > 6:
                raise Unmatched('more than six bound field factors beside the form')
        # array axes: exposed dof axes in array order define (rows, cols); free axes are component axes tied to them
        nexp = len(exposed) + len(derivs)
        if nexposed is None:
            nexposed = nexp
        elif nexposed != nexp:
            raise Unmatched('terms of different rank')
        form.sort(key=lambda i: (dofpos.index(i) if i in dofpos else len(dofpos) + bound.index(i)))
        # contract A to the form tensor: indices [free..., (c, s) of form factors]; polynomial factors contribute their value slot
        # (point variables: every non-zero (component, slot) combination of the factors in `pv` is a term of its own)
        pv_shapes = [A.shape[nf + 2 * i: nf + 2 * i + 2] for i in pv]
        combos = []
        for combo in numpy.ndindex(*[n for sh in pv_shapes for n in sh]):
            idx = [slice(None)] * nf
            for i in range(len(facs)):
                if i in form:
                    idx += [slice(None), slice(None)]
                elif i in pv:
                    j = pv.index(i)
                    idx += [combo[2 * j], combo[2 * j + 1]]
                else:
                    idx += [0, 0]
            Tc = A[tuple(idx)]
            if nf:
                # free axes of the array are component axes: each must be tied (identity) to the component of one exposed factor
                Tc = _tie_components(Tc, nf, len(form), m, facs, form)
            if not pv or numpy.abs(Tc).sum():
                combos.append((combo, Tc))
        if form:
            home = E.basis_transforms(facs[form[0]].basis)
        elif anybasis is not None:
            home = E.basis_transforms(anybasis)
        else:
            # no basis in any term (lengths, areas, integrals of coefficient functions over a boundary): the sample's own elements are the topology; on a boundary the
            # measure is then tabulated with the coefficient functions (no parent element to take a Jacobian in)
            home = smp.transforms[0]
        selfhome = home is smp.transforms[0] and home.fromdims != home.todims
        if selfhome and (form or pv or poly and any(facs[i].basis is not _SCALAR for i in poly)):
            raise Unmatched('boundary integral without any basis: the parent topology is unknown')
        sis = E.iface_sample(smp, home) if iface else E.sample(smp, home)
        if iface and m.measure is None:
            raise Unmatched('interface term without a measure')
        if iface and any(f.rational is not None for f in facs):
            raise Unmatched('rational basis on an interface')
        for si, (combo, T) in [(si, ct) for si in (sis if isinstance(sis, list) else [sis]) for ct in combos]:
            if iface:  # one sample per pair of (basis, side) in the form: see Emitter.iface_variant
                si = E.iface_variant(si, tuple((id(facs[i].basis), facs[i].side) for i in form))
            s = E.plan['samples'][si]
            if m.measure is None:
                # an integral over the REFERENCE elements (no J(geom): the weight-function projection of examples/platewithhole.py:83): the measure of the identity map
                gnode, gi = None, E.geom_unit(si)
            elif selfhome:
                gnode, gi = None, E.geom_unit(si)  # (the measure joins the pointwise factor below)
            else:
                gnode, tip = m.measure
                gi = E.geom_tab(gnode, smp, si, home) if iface else _geom_index(E, gnode, smp, si, home)
            gg = -1
            gnodes = {id(facs[i].geom): facs[i].geom for i in form + pv if facs[i].geom is not None}
            if len(gnodes) > 1:
                raise Unmatched('gradients with respect to different geometries')
            if gnodes:
                g = next(iter(gnodes.values()))
                gg = gi if g is gnode else E.geom_tab(g, smp, si, home) if iface else _geom_index(E, g, smp, si, home)
            term = dict(sample=si, fac=float(fac), measure=gi, geom=gg, test=-1, trial=-1, rows=False, cols=False, scale=None, fpoly=None)
            def on_home(bi):  # every basis of a term must be indexed by the elements of the sample's topology (no field of a coarser level)
                b = E.plan['bases'][bi]
                while b['kind'] == 'rational':
                    b = E.plan['bases'][b['parent']]
                if b['topo'] != s['topo']:
                    raise Unmatched('bases of different topologies in one term')
                return bi
            ai = []
            def basis_of(f):
                if iface:
                    return on_home(E.iface_basis(f.basis, f.side, smp, si, home, gnode))
                if f.rational is None and f.cvals is None and E.basis_transforms(f.basis) is not home and E.topo(E.basis_transforms(f.basis)) != s['topo']:
                    return on_home(E.restricted_basis(f.basis, home))  # (a field of the coarser level in an integral over the refined one)
                return on_home(E.basis(f.basis, f.rational, (smp, si) if f.rational is not None else None))
            for i in form:
                f = facs[i]
                ai.append(E.arg(f.name, basis_of(f), f.ncomp, f.part))
            if len(form) == 2:
                Bt = numpy.ascontiguousarray(T)
                if not exposed and ai[0] > ai[1]:  # both bound: canonical order, so that B(u, w) and B(w, u) merge
                    ai = ai[::-1]
                    Bt = numpy.ascontiguousarray(numpy.moveaxis(Bt, (0, 1, 2, 3), (2, 3, 0, 1)))
                term.update(test=ai[0], trial=ai[1], rows=form[0] in exposed, cols=form[1] in exposed, B=Bt)
            elif len(form) == 1:
                term.update(test=ai[0], rows=form[0] in exposed, L=numpy.ascontiguousarray(T))
            else:
                term.update(f0=numpy.asarray(float(T)))
            if pv:
                term['pvars'] = []
                for j, i in enumerate(pv):
                    f = facs[i]
                    term['pvars'].append([E.arg(f.name, basis_of(f), f.ncomp, f.part), int(combo[2 * j]), int(combo[2 * j + 1])])
            pwn = list(m.pw) + ([M.rf.jacobian(m.measure[0], m.measure[1])] if selfhome and m.measure is not None else [])
            if pwn:
                node = pwn[0]
                for p in pwn[1:]:
                    node = node * p
                term['scale'] = _point_values(smp, node, s)
            if s.get('_bnd_normal') is not None and gnode is not None:  # (oblique face: see Emitter.sample; the reference measure of a face is that of its own parameters)
                if E.plan['geoms'][gi]['kind'] != 'tab':
                    term['measure'] = gi = E.geom_tab(gnode, smp, si, home)
                    if gg >= 0 and g is gnode:
                        term['geom'] = gi
                J = E.plan['geoms'][gi]['jac']
                r = numpy.linalg.norm(numpy.linalg.solve(numpy.swapaxes(J, -1, -2), numpy.broadcast_to(s['_bnd_normal'], J.shape[:-1])[..., None])[..., 0], axis=-1)
                term['scale'] = r if term['scale'] is None else term['scale'] * r
            if poly:
                pargs = []
                for i in poly:
                    f = facs[i]
                    if f.basis is _SCALAR:
                        a = E.arg(f.name, on_home(E.scalar_basis(home, s['topo'] if iface else None)), 1)
                        E.plan['args'][a]['scalar'] = True
                    else:
                        a = E.arg(f.name, basis_of(f), 1, f.part)
                    pargs.append(a)
                uniq = sorted(set(pargs))
                term['fpoly'] = dict(args=uniq, powers=numpy.array([[pargs.count(a) for a in uniq]]), coeffs=[1.])
            E.plan['terms'].append(term)
    plan = E.plan
    _merge_terms(plan)
    for s in plan['samples'] + plan['topos']:
        for k in [k for k in s if k.startswith('_')]:
            del s[k]
    plan['derivs'] = derivs
    used = {a['name'] for a in plan['args']}
    if any(n in used for n in M.derived):
        plan['derived'] = {n: ast for n, ast in M.derived.items() if n in used}
    plan['shape'] = [int(n) for n in shape]
    plan['kind'] = 'scalar' if nexposed == 0 else 'vector' if nexposed == 1 else 'matrix'
    if nexposed > 2:
        raise Unmatched('arrays with more than two dof axes')
    plan['_source'] = source  # (not stored: tools/hip_plan_capture.py evaluates it through the un-hooked reference to pin the plan's expected result)
    return plan


def match_points(array):
    '''`sample.bind(func)` of the reference (sample.py:217-237: `_ConcatenatePoints`, for samples with their own point indices inside `_ReorderPoints`; what
    `Sample.eval` hands to function.eval) -> plan of kind 'points'.  `func` must be a sum of (constant tensor) x (values / gradients of fields bound to arguments or to
    constant coefficient arrays, coordinates) x (coefficient function of the point): no exposed dof axis, no measure.  The result has the reference's shape
    (npoints, *func.shape) and point order.'''
    node, reorder = array, False
    if _kind(node) == '_ReorderPoints':
        node, reorder = node._func, True
    if _kind(node) != '_ConcatenatePoints':
        raise Unmatched(f'node {_kind(array)} at the top of an array')
    func, smp = node._func, node._sample
    if len(getattr(smp, 'spaces', ())) != 1 or not hasattr(smp, 'transforms') or not hasattr(smp, 'points'):
        raise Unmatched(f'points of a {_kind(smp)} sample')
    if func.dtype not in (float, int, bool) or array.dtype != float:
        raise Unmatched(f'point function of type {func.dtype}')
    M = Matcher()
    M.points_mode = True
    tr = smp.transforms[0]
    M.sample, M.S = smp, 1 + tr.todims
    # pointwise functions of one argument at the top (`sqrt(u_i u_i)`, examples/drivencavity.py:180): the plan evaluates what is inside, the host applies them to the
    # point values it returns (Sample.eval hands out host arrays)
    post = []
    while True:
        inner = func
        while _kind(inner) == '_Transpose' and tuple(inner._axes) == tuple(range(len(inner._axes))):  # (the scalar `sqrt(..)` of an expression arrives under a Transpose of no axes)
            inner = inner._arg
        if _kind(inner) == '_Wrapper' and _name(inner) in _POINTWISE and len(inner._args) == 1 and inner._args[0].shape == func.shape and inner._args[0].dtype == float:
            post.append(_name(inner))
            func = inner._args[0]
        else:
            break
    monos = M.conv(func)
    E = Emitter(M)
    anybasis = next((f.basis for m in monos for f in m.factors if f.basis is not _SCALAR and f.basis is not _XGEOM), None)

    def home_of(m):
        '''the topology a term's elements are numbered in: that of its own first basis (a geometry on the unrefined topology beside a field on the refined one: each term
        sees the sample's elements located in ITS topology), else that of any basis of the function, else the sample's own'''
        b = next((f.basis for f in m.factors if f.basis is not _SCALAR and f.basis is not _XGEOM), anybasis)
        if b is not None:
            return E.basis_transforms(b)
        # no basis anywhere in the function (coordinates, constants, scalar arguments at the points of a sample): the sample's own elements are the topology -- nothing
        # in such a term refers to a parent element (coordinates a kernel cannot evaluate from a structural description are tabulated at the points)
        return tr
    nd = M.S - 1
    pterms = []
    nconst = [0]

    def const_arg(f, bi):
        key = ('$const', id(f.cvals), bi)
        if key not in E._arg:
            E._keep.append(f.cvals)
            E.plan['args'].append(dict(name=f'_const{nconst[0]}', basis=bi, ncomp=int(f.ncomp), values=numpy.asarray(f.cvals, dtype=float).reshape(-1, f.ncomp)))
            nconst[0] += 1
            E._arg[key] = len(E.plan['args']) - 1
        return E._arg[key]

    def emit(m, si, home):
        s = E.plan['samples'][si]
        if m.measure is not None:
            raise Unmatched('measure inside a function evaluated at points')
        if any(k != 'free' for k, _ in m.axes):
            raise Unmatched('basis array evaluated at points (dense npoints x ndofs result)')
        nf = m.nfree
        A = numpy.transpose(m.A, [j for _, j in m.axes] + list(range(nf, m.A.ndim)))  # free axes in array order
        if len(m.factors) > 6:
            raise Unmatched('more than six field factors in a point function')
        facs, blocks = [], []
        for i, f in enumerate(m.factors):
            blk = [slice(None)] * A.ndim
            if f.basis is _XGEOM:
                blk[nf + 2 * i + 1] = slice(1, None)
                if numpy.abs(A[tuple(blk)]).sum():
                    raise Unmatched('gradient of a coordinate function')
                try:
                    gi = E.geom(f.xnode, smp, si)
                    if E.plan['geoms'][gi]['kind'] != 'rectilinear':
                        raise Unmatched('coordinate function without a structural description')
                except Unmatched:
                    # not a map the kernels evaluate: its components are coefficient functions of the point like any other
                    out = []
                    for c in range(nd):
                        sub = A[(slice(None),) * (nf + 2 * i) + (c, 0)]
                        if numpy.abs(sub).sum():
                            axes = [('free', j) for j in range(nf)]
                            out += emit(_Mono(sub, axes, m.factors[:i] + m.factors[i + 1:], m.pw + [f.xnode[c]], None), si, home)
                    return out
                facs.append(dict(x=gi))
                blocks.append((nd, 1))
                continue
            uses_grad = False
            blk[nf + 2 * i + 1] = slice(1, None)
            if numpy.abs(A[tuple(blk)]).sum():
                uses_grad = True
            if f.basis is _SCALAR:
                if uses_grad:
                    raise Unmatched('gradient of a scalar argument')
                ai = E.arg(f.name, E.scalar_basis(home), 1)
                E.plan['args'][ai]['scalar'] = True
            else:
                bi = E.basis(f.basis, f.rational, (smp, si) if f.rational is not None else None)
                b = E.plan['bases'][bi]
                while b['kind'] == 'rational':
                    b = E.plan['bases'][b['parent']]
                if b['topo'] != s['topo']:
                    raise Unmatched('bases of different topologies in one term')
                if f.cvals is not None:
                    ai = const_arg(f, bi)
                elif f.name is None:
                    raise Unmatched('basis array evaluated at points (dense npoints x ndofs result)')
                else:
                    ai = E.arg(f.name, bi, f.ncomp, f.part)
            gi = -1
            if uses_grad:
                if f.geom is None:
                    raise Unmatched('gradient slot without a geometry')
                gi = _geom_index(E, f.geom, smp, si, home)
            facs.append(dict(arg=ai, geom=gi))
            blocks.append((f.ncomp, M.S))
        # the x factors keep their value slot only
        idx = [slice(None)] * nf
        for (nc, sl) in blocks:
            idx += [slice(None), slice(0, sl)]
        A = numpy.array(A[tuple(idx)], dtype=float, order='C')  # (ascontiguousarray would turn a scalar into a vector)
        term = dict(sample=si, A=A.reshape(-1), Ashape=list(A.shape), factors=facs, scale=None)
        if m.pw:
            pnode = m.pw[0]
            for q in m.pw[1:]:
                pnode = pnode * q
            term['scale'] = _point_values(smp, pnode, s)
        return [term]

    sis = []
    for m in monos:
        home = home_of(m)
        these = E.sample(smp, home)
        for si in these if isinstance(these, list) else [these]:
            pterms += emit(m, si, home)
            if si not in sis:
                sis.append(si)
    plan = E.plan
    plan['pterms'] = pterms
    # rows of the result: element i of the sample owns the rows off[i] .. off[i+1] of the concatenation, which the sample's indices place (identity for default samples)
    psamples = []
    for si in sis:
        dest = _sample_rows(smp, plan['samples'][si])
        if dest is None:
            dest = numpy.arange(int(array.shape[0]))
        identity = len(sis) == 1 and len(dest) == int(array.shape[0]) and numpy.array_equal(dest, numpy.arange(len(dest)))
        psamples.append(dict(sample=si, dest=None if identity else numpy.asarray(dest, dtype=numpy.int64)))
    plan['psamples'] = psamples
    for s in plan['samples'] + plan['topos']:
        for k in [k for k in s if k.startswith('_')]:
            del s[k]
    plan['derivs'] = []
    used = {a['name'] for a in plan['args']}
    if any(n in used for n in M.derived):
        plan['derived'] = {n: ast for n, ast in M.derived.items() if n in used}
    plan['shape'] = [int(n) for n in array.shape]
    plan['kind'] = 'points'
    if post:
        plan['post'] = post[::-1]  # innermost first
    plan['_source'] = array
    return plan


def _tie_components(T, nf, nform, m, facs, form):
    '''Free array axes left at the integral are component axes of exposed vector-valued factors ((ndofs, ncomp) arguments exposed by a
    derivative are handled by the front end; here: basis arrays carry no components): not supported beyond trivial axes.'''
    if all(n == 1 for n in T.shape[:nf]):
        return T.reshape(T.shape[nf:])
    raise Unmatched('array-valued integrand (free axes left after integration)')


def _geom_index(E, gnode, smp, si, home):
    try:
        return E.geom(gnode, smp, si)
    except Unmatched:
        return E.geom_tab(gnode, smp, si, home)


def _merge_terms(plan):
    '''terms that differ only in their constant tensor / polynomial factor are added (the expansion of (phi^2 - 1)^2 into monomials gives many)'''
    out = []
    for t in plan['terms']:
        for u in out:
            same = (all(t[k] == u[k] for k in ('sample', 'measure', 'test', 'trial', 'rows', 'cols')) and t['scale'] is None and u['scale'] is None
                    and sorted(map(tuple, t.get('pvars') or [])) == sorted(map(tuple, u.get('pvars') or [])))
            if not same or (t['geom'] != u['geom'] and min(t['geom'], u['geom']) >= 0):
                continue
            u['geom'] = max(t['geom'], u['geom'])
            kt = 'B' if 'B' in t else 'L' if 'L' in t else 'f0'
            if kt not in u:
                continue
            if t['fpoly'] is None and u['fpoly'] is None:
                u[kt] = u[kt] * u['fac'] + t[kt] * t['fac']
                u['fac'] = 1.
                break
            if t['fpoly'] is not None and u['fpoly'] is not None and not numpy.array_equal(t[kt], u[kt]):
                # tensors that differ by a scalar factor only (f0 = +-1, 3, ... of the expansion of a polynomial): fold it into the polynomial
                a, b = numpy.asarray(t[kt], dtype=float).ravel(), numpy.asarray(u[kt], dtype=float).ravel()
                i = int(numpy.argmax(numpy.abs(b)))
                if b[i] != 0 and numpy.allclose(a, b * (a[i] / b[i]), rtol=1e-15, atol=0):
                    t = dict(t, fac=t['fac'] * (a[i] / b[i]))
                    t[kt] = u[kt]
            if t['fpoly'] is not None and u['fpoly'] is not None and numpy.array_equal(t[kt], u[kt]):
                args = sorted(set(t['fpoly']['args']) | set(u['fpoly']['args']))

                def lift(p, fac):
                    pw = numpy.zeros((len(p['coeffs']), len(args)), dtype=int)
                    for j, a in enumerate(p['args']):
                        pw[:, args.index(a)] = numpy.asarray(p['powers'])[:, j]
                    return pw, [c * fac for c in p['coeffs']]
                pu, cu = lift(u['fpoly'], u['fac'])
                pt, ct = lift(t['fpoly'], t['fac'])
                acc = {}
                for pw, c in zip(numpy.concatenate([pu, pt]).tolist(), cu + ct):
                    acc[tuple(pw)] = acc.get(tuple(pw), 0.) + c
                keys = [k for k, c in acc.items() if c != 0.] or [tuple([0] * len(args))]
                u['fpoly'] = dict(args=args, powers=numpy.array(keys, dtype=int).reshape(len(keys), len(args)), coeffs=[acc.get(k, 0.) for k in keys])
                u['fac'] = 1.
                break
        else:
            out.append(dict(t))
    plan['terms'] = out


# =====================================================================================================================================
# hooks at the seam of an importable reference
# =====================================================================================================================================

_STATE = None


def _empty_block(nrows, ncols):
    return numpy.zeros(0), numpy.zeros(nrows + 1, dtype=numpy.int64), numpy.zeros(0, dtype=numpy.int64), ncols


class _SystemPlans:
    '''Plans of the blocks a `solver.System` evaluates (solver.py:238-260): value, residual blocks, Jacobian blocks -- derived from the
    FUNCTION-LEVEL functional / residual the System was created with (the System itself keeps only lowered evaluables).'''

    def __init__(self, system, residual, trials, tests, executor):
        import nutils.function as rf
        self.ex = executor
        self.trials, self.tests = trials, tests
        self.sizes = [int(numpy.prod(shape)) for shape in system.trial_shapes]
        self.value = match(residual) if system.is_symmetric else None
        res_arrays = [rf.derivative(residual, t) for t in tests]
        self.res = [self._match_or_empty(a) for a in res_arrays]
        self.jac = [[self._match_or_empty(rf.derivative(a, t)) for t in trials] for a in res_arrays]

    @staticmethod
    def _match_or_empty(array):
        plan = match(array)
        build(plan)  # (the derivative may leave no term: then the block is structurally empty)
        return plan if plan['_built'].integral.terms else None

    def residual(self, arguments):
        return tuple(numpy.zeros(n) if p is None else numpy.asarray(self.ex(p, arguments), dtype=float).ravel() for p, n in zip(self.res, self.sizes))

    def jacobian(self, arguments):
        rows = []
        for i, row in enumerate(self.jac):
            blocks = []
            for j, p in enumerate(row):
                if p is None:
                    blocks.append(_empty_block(self.sizes[i], self.sizes[j]))
                else:
                    v, rp, ci = self.ex(p, arguments)
                    blocks.append((v, rp, ci, self.sizes[j]))
            rows.append(tuple(blocks))
        return tuple(rows)


def device_initialised():
    '''has this process touched the GPU (a HIP context exists)?'''
    import sys
    torch = sys.modules.get('torch')
    return bool(torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized())


def install(executor=None):
    '''Route the reference's evaluation through plans: patches nutils.function.evaluate / as_csr and nutils.solver.System.__init__.
    `executor(plan, arguments)` defaults to `execute` (the C ABI); anything unmatched takes the reference's own path.  nutils.parallel.fork is
    guarded: once the device layer is initialised the reference's element loops are not forked any more (`forks_refused` counts them).  Returns the state
    (lists `matched` / `fallback`) for inspection; `uninstall()` restores the reference.'''
    global _STATE
    if _STATE is not None:
        raise RuntimeError('seam already installed')
    import nutils.function as rf
    import nutils.solver as rs
    import nutils.matrix as rmatrix
    import nutils.parallel as rp
    import collections
    ex = executor or execute
    # plans / as_csr components are remembered per array OBJECT (id + identity check) in bounded LRU maps: a time loop that builds a fresh integral
    # every step must not grow host and device memory without bound -- an evicted plan drops its built tables ('_built') with it
    st = dict(evaluate=rf.evaluate, as_csr=rf.as_csr, system_init=rs.System.__init__, factor_class=rf._Factor, csr=collections.OrderedDict(),
              plans=collections.OrderedDict(), point_plans=collections.OrderedDict(), matched=[], fallback=[], max_plans=PLAN_CACHE_SIZE, fork=rp.fork, forks_refused=0, busy=0, declined_points=[], internal_points=[])

    def fork_guard(nprocs=None):
        '''parallel.fork (parallel.py:27-88) once the device layer is initialised in this process: NOT forked -- a child would inherit a HIP context it cannot use
        (the reference's element loop then runs in this process, as with NUTILS_NPROCS=1); before that the reference forks as always.'''
        if device_initialised():
            st['forks_refused'] += 1
            return rp._DontFork()
        return st['fork'](nprocs)

    def remember(cache, key, value, limit):
        cache[key] = value
        cache.move_to_end(key)
        while len(cache) > limit:
            _, old = cache.popitem(last=False)
            plan = old[1]
            if isinstance(plan, dict):
                plan.pop('_built', None)

    def plan_of(array):
        # (sample.bind builds a fresh array per Sample.eval call: those plans live in a small map of their own, so that a time loop that plots every step does not evict
        # the integral plans of its Systems)
        cache, limit = (st['point_plans'], 16) if _kind(array) in ('_ConcatenatePoints', '_ReorderPoints') else (st['plans'], st['max_plans'])
        hit = cache.get(id(array))
        if hit is None or hit[0] is not array:
            st['busy'] += 1  # (the matcher tabulates coefficient functions through Sample.eval: those evaluations belong to the reference)
            try:
                if _kind(array) in ('_ConcatenatePoints', '_ReorderPoints'):  # sample.bind(func): Sample.eval (sample.py:192-232)
                    plan = match_points(array)
                    build(plan)
                else:
                    plan = match(array)
                    for part in (plan['parts'] if plan['kind'] == 'stack' else [plan]):
                        if part is not None and not build(part).integral.terms:
                            raise Unmatched('empty integral')
            except Unmatched as e:
                plan = e
                if _kind(array) in ('_ConcatenatePoints', '_ReorderPoints'):
                    # (integer / boolean functions -- element indices, masks: topology.select / locate / trim evaluate those -- are not the class of Sample.eval)
                    st['internal_points' if array.dtype != float else 'declined_points'].append(str(e))
            except Exception as e:  # (an expression shape the matcher does not know must never break the user's script: reference path)
                plan = Unmatched(f'{type(e).__name__}: {e}')
                if _kind(array) in ('_ConcatenatePoints', '_ReorderPoints'):
                    st['declined_points'].append(str(plan))
            finally:
                st['busy'] -= 1
            hit = (array, plan)
            remember(cache, id(array), hit, limit)
        else:
            cache.move_to_end(id(array))
        return hit[1]

    def run(plan, arguments):
        '''executor call that never breaks the user's script: a failure turns the plan into Unmatched (recorded) and the caller takes the reference path'''
        try:
            return ex(plan, dict(arguments))
        except Exception as e:
            st['fallback'].append(f'executor failed on a {plan.get("kind")} plan: {type(e).__name__}: {e}')
            plan.pop('_built', None)
            plan['_failed'] = True
            return None

    def as_csr(array):
        out = st['as_csr'](array)  # the reference's evaluable triplet: symbolic, evaluated only if the plan is not used
        plan = plan_of(array)
        if not isinstance(plan, Unmatched):
            for i, o in enumerate(out):
                remember(st['csr'], id(o), (o, plan, i), 3 * st['max_plans'])
        return out

    def factor_hook(array):
        '''function.factor(array) (function.py:2630-2642) of an integral the matcher recognises: TRANSPARENT -- the integral itself is returned, so that
        it is re-assembled from its plan whenever it is evaluated (per Newton step, BASELINE.json configs[3]) instead of being expanded ONCE into
        sparse Taylor tensors by the reference's element loop (evaluable.py:5785-5874).  Anything else is factored by the reference.'''
        try:
            if isinstance(array, rf.Array) and not isinstance(plan_of(array), Unmatched):
                st['matched'].append('factor')
                return array
        except Exception:
            pass
        return st['factor_class'](array)

    def evaluate(*arrays, arguments={}):
        if st['busy']:  # an evaluation from inside the matcher (coefficient functions, tabulated geometries): the reference's
            return st['evaluate'](*arrays, arguments=arguments)
        results, rest = [None] * len(arrays), []
        done = {}
        for i, a in enumerate(arrays):
            hit = st['csr'].get(id(a))
            if hit is not None and hit[0] is a:
                plan, comp = hit[1], hit[2]
            elif isinstance(a, rf.Array):
                plan, comp = plan_of(a), None
            else:
                plan, comp = Unmatched('not an array'), None
            if isinstance(plan, Unmatched) or plan.get('_failed'):
                rest.append(i)
                continue
            key = id(plan)
            if key not in done:
                done[key] = run(plan, arguments)
                if done[key] is not None:
                    st['matched'].append(plan['kind'])
            out = done[key]
            if out is None:  # the executor failed: the reference evaluates this array
                rest.append(i)
                continue
            if comp is not None:
                results[i] = out[comp]
            elif plan['kind'] == 'matrix':  # a rank-2n array asked for densely: as the reference returns it
                v, rp, ci = out
                n = len(rp) - 1
                dense = numpy.zeros((n, int(numpy.prod(plan['shape'])) // n))
                dense[numpy.repeat(numpy.arange(n), numpy.diff(rp)), ci] = v
                results[i] = dense.reshape(plan['shape'])
            else:
                results[i] = numpy.asarray(out) if plan['kind'] in ('vector', 'points', 'stack') else numpy.float64(out)
        if rest:
            st['fallback'].append(len(rest))
            for i, r in zip(rest, st['evaluate'](*[arrays[i] for i in rest], arguments=arguments)):
                results[i] = r
        return tuple(results)

    def system_init(self, residual, /, trial, test=None):
        st['system_init'](self, residual, trial=trial, test=test)
        st['busy'] += 1
        try:
            if isinstance(residual, (tuple, list)):
                raise Unmatched('residual given as a list of vectors')
            tests = self.trials if test is None else tuple(test.split(',') if isinstance(test, str) else test)
            sp = _SystemPlans(self, residual, self.trials, tests, ex)
        except Unmatched as e:
            st['fallback'].append(f'System: {e}')
            return
        except Exception as e:  # (see plan_of)
            st['fallback'].append(f'System: {type(e).__name__}: {e}')
            return
        finally:
            st['busy'] -= 1
        st['matched'].append('System')
        cache = self._System__cache
        zero = lambda arguments: dict(arguments, **{t: numpy.zeros(shape) for t, shape in zip(self.trials, self.trial_shapes)})
        import nutils.evaluable as rev
        bj, br = self._System__block_jacobian, self._System__block_residual

        def zeroed_residual():  # what solver.py:364-366 compiles for a linear system
            z = dict(zip(self.trials, map(rev.zeros_like, self.trial_args)))
            return tuple(rev.replace_arguments(v, z).simplified for v in br)

        # the reference's own compiled function for each cache key (solver.py:318-420), built only if a plan fails at run time
        reference = {
            'residual': lambda: rev.compile(br),
            'value': lambda: rev.compile(self._System__value),
            'jacobian': lambda: rev.compile(bj),
            'jacobian_residual': lambda: rev.compile((bj, zeroed_residual() if self.is_linear else br)),
            'jacobian_residual_value': lambda: rev.compile((bj, zeroed_residual(), rev.replace_arguments(self._System__value, dict(zip(self.trials, map(rev.zeros_like, self.trial_args)))))
                                                          if self.is_linear else (bj, br, self._System__value)),
        }

        def guarded(key, ours):
            '''our plan-based function under the reference's cache key; an executor failure is recorded, the key is handed back to the reference's own
            compiled function and the call is answered by it: the user's script never sees the failure'''
            def f(arguments):
                try:
                    return ours(arguments)
                except Exception as e:
                    st['fallback'].append(f'System.{key}: executor failed: {type(e).__name__}: {e}')
                    ref = cache[key] = reference[key]()
                    return ref(arguments)
            return f

        cache['residual'] = guarded('residual', sp.residual)
        if self.is_symmetric:
            cache['value'] = guarded('value', lambda arguments: numpy.float64(ex(sp.value, arguments)))
        if not self.is_constant_matrix:
            cache['jacobian'] = guarded('jacobian', sp.jacobian)
        # (a constant Jacobian is kept as the assembled MATRIX under 'jacobian', solver.py:321-331: it is left to the first assemble_jacobian_residual*
        # call, which assembles it from the blocks of OUR function below -- lazily, as the reference does, and behind the guard)
        if self.is_linear:  # the reference evaluates the residual at zeroed trial arguments and adds jac @ x (solver.py:364-378)
            cache['jacobian_residual'] = guarded('jacobian_residual', lambda arguments: (sp.jacobian(arguments), sp.residual(zero(arguments))))
            if self.is_symmetric:
                cache['jacobian_residual_value'] = guarded('jacobian_residual_value',
                                                           lambda arguments: (sp.jacobian(arguments), sp.residual(zero(arguments)), numpy.float64(ex(sp.value, zero(arguments)))))
        else:
            cache['jacobian_residual'] = guarded('jacobian_residual', lambda arguments: (sp.jacobian(arguments), sp.residual(arguments)))
            if self.is_symmetric:
                cache['jacobian_residual_value'] = guarded('jacobian_residual_value',
                                                           lambda arguments: (sp.jacobian(arguments), sp.residual(arguments), numpy.float64(ex(sp.value, arguments))))

    rf.evaluate, rf.as_csr, rs.System.__init__, rf._Factor = evaluate, as_csr, system_init, factor_hook
    rp.fork = fork_guard
    _STATE = st
    return st


def uninstall():
    global _STATE
    if _STATE is None:
        return
    import nutils.function as rf
    import nutils.solver as rs
    import nutils.parallel as rp
    rf.evaluate, rf.as_csr, rs.System.__init__, rf._Factor = _STATE['evaluate'], _STATE['as_csr'], _STATE['system_init'], _STATE['factor_class']
    rp.fork = _STATE['fork']
    _STATE = None
