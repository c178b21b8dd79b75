'''ctypes binding of libnutils_hip.so (include/nutils_hip.h).

The library is loaded the way the reference loads its only native dependency
(``_util.loadlib``, /root/reference/src/nutils/_util.py:195-237; used by
matrix/_mkl.py:10): ``ctypes.CDLL`` on an explicit path.  There is NO fallback:
if the shared object is missing or a call fails, an exception is raised -- the
product path never computes on the CPU.
'''

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.environ.get('NUTILS_AMD_LIB') or os.path.join(_HERE, 'libnutils_hip.so')  # (override: A/B runs of two builds)

c_i64 = ctypes.c_int64
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f64p = ctypes.POINTER(ctypes.c_double)
vp = ctypes.c_void_p


class NutilsHipError(RuntimeError):
    '''Raised for any non-zero status of the C ABI (cf. matrix.MatrixError,
    /root/reference/src/nutils/matrix/_base.py:9-12).'''


class Geometry(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int), ('ngb', ctypes.c_int), ('gT_dev', vp), ('gdofs_dev', vp), ('verts_dev', vp),
                ('origin_dev', vp), ('size_dev', vp), ('jac_dev', vp), ('x_dev', vp), ('bnd_axis', ctypes.c_int)]


class Basis(ctypes.Structure):
    _fields_ = [('nb', ctypes.c_int), ('T_dev', vp), ('dofs_dev', vp), ('off_dev', vp), ('tab_dev', vp)]


class PatternArgs(ctypes.Structure):
    _fields_ = [('nelems', c_i64), ('nrows', c_i64), ('ncols', c_i64), ('nbt', ctypes.c_int), ('nbr', ctypes.c_int),
                ('tdofs_dev', vp), ('rdofs_dev', vp), ('toff_dev', vp), ('roff_dev', vp)]


class MatrixArgs(ctypes.Structure):
    _fields_ = [('nelems', c_i64), ('elist_dev', vp), ('ndims', ctypes.c_int), ('nq', ctypes.c_int), ('weights_dev', vp),
                ('geom', Geometry), ('test', Basis), ('trial', Basis), ('nct', ctypes.c_int), ('ncr', ctypes.c_int),
                ('C_host', vp), ('mask_host', vp), ('srowptr_dev', vp), ('emap_dev', vp), ('eoff_dev', vp), ('values_dev', vp), ('scale_dev', vp), ('flags', ctypes.c_int), ('cq_dev', vp),
                ('grid_shape', ctypes.c_int * 3), ('nodes_per_axis', ctypes.c_int), ('pattern', vp)]


class FactorArgs(ctypes.Structure):
    _fields_ = [('nelems', c_i64), ('elist_dev', vp), ('ndims', ctypes.c_int), ('nq', ctypes.c_int), ('rank', ctypes.c_int), ('weights_dev', vp),
                ('geom', Geometry), ('basis', Basis), ('scale_dev', vp), ('coeff', ctypes.c_double), ('ndofs', c_i64)]


class VectorArgs(ctypes.Structure):
    _fields_ = [('nelems', c_i64), ('elist_dev', vp), ('ndims', ctypes.c_int), ('nq', ctypes.c_int), ('weights_dev', vp),
                ('geom', Geometry), ('test', Basis), ('trial', Basis), ('nct', ctypes.c_int), ('ncr', ctypes.c_int),
                ('C_host', vp), ('f_host', vp), ('u_dev', vp), ('out_dev', vp), ('f0', ctypes.c_double), ('out_scalar_dev', vp), ('scale_dev', vp),
                ('local_dev', vp)]


class Field(ctypes.Structure):
    _fields_ = [('basis', Basis), ('u_dev', vp), ('ncomp', ctypes.c_int)]


class Block(ctypes.Structure):
    _fields_ = [('test', Basis), ('nct', ctypes.c_int), ('out_dev', vp), ('local_dev', vp)]


class PointPoly(ctypes.Structure):
    _fields_ = [('nvars', ctypes.c_int), ('nterms', ctypes.c_int), ('field', ctypes.c_int * 4), ('comp', ctypes.c_int * 4), ('coeffs_host', vp), ('powers_host', vp)]


class Term(ctypes.Structure):
    _fields_ = [('block', ctypes.c_int), ('field', ctypes.c_int), ('poly', ctypes.c_int), ('C_host', vp), ('f_host', vp), ('scale_dev', vp),
                ('qs_field_t', ctypes.c_int), ('qs_field_r', ctypes.c_int), ('qs_B_host', vp)]


class TermsArgs(ctypes.Structure):
    _fields_ = [('nelems', c_i64), ('elist_dev', vp), ('ndims', ctypes.c_int), ('nq', ctypes.c_int), ('weights_dev', vp), ('geom', Geometry),
                ('nfields', ctypes.c_int), ('fields', ctypes.POINTER(Field)), ('nblocks', ctypes.c_int), ('blocks', ctypes.POINTER(Block)),
                ('nterms', ctypes.c_int), ('terms', ctypes.POINTER(Term)), ('npolys', ctypes.c_int), ('polys', ctypes.POINTER(PointPoly))]


class MatrixTerm(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int), ('field', ctypes.c_int), ('poly', ctypes.c_int), ('C_host', vp), ('L_host', vp), ('scale_dev', vp),
                ('qs_field_t', ctypes.c_int), ('qs_field_r', ctypes.c_int), ('qs_B_host', vp)]


class MatrixTermsArgs(ctypes.Structure):
    _fields_ = [('nelems', c_i64), ('elist_dev', vp), ('ndims', ctypes.c_int), ('nq', ctypes.c_int), ('weights_dev', vp), ('geom', Geometry),
                ('test', Basis), ('trial', Basis), ('nct', ctypes.c_int), ('ncr', ctypes.c_int), ('mask_host', vp), ('srowptr_dev', vp), ('emap_dev', vp),
                ('eoff_dev', vp), ('values_dev', vp), ('flags', ctypes.c_int), ('nfields', ctypes.c_int), ('fields', ctypes.POINTER(Field)),
                ('nterms', ctypes.c_int), ('terms', ctypes.POINTER(MatrixTerm)), ('npolys', ctypes.c_int), ('polys', ctypes.POINTER(PointPoly)), ('pattern', vp)]


class EvalArgs(ctypes.Structure):
    _fields_ = [('nelems', c_i64), ('elist_dev', vp), ('ndims', ctypes.c_int), ('nq', ctypes.c_int), ('geom', Geometry), ('trial', Basis),
                ('ncr', ctypes.c_int), ('points_dev', vp), ('u_dev', vp), ('x_dev', vp), ('detj_dev', vp), ('U_dev', vp)]


class P1HexArgs(ctypes.Structure):
    _fields_ = [('shape', ctypes.c_int * 3), ('layer_begin', ctypes.c_int), ('layer_end', ctypes.c_int), ('plane_begin', ctypes.c_int),
                ('plane_end', ctypes.c_int), ('verts_dev', vp), ('origin', ctypes.c_double * 3), ('scale', ctypes.c_double * 3),
                ('gauss_x', ctypes.c_double * 2), ('gauss_w', ctypes.c_double * 2), ('kappa', ctypes.c_double), ('values_dev', vp),
                ('unit_matrix_dev', vp), ('qscale_dev', vp), ('mass', ctypes.c_double), ('qmass_dev', vp), ('max_workgroups', ctypes.c_int)]


class P2HexArgs(ctypes.Structure):
    _fields_ = [('shape', ctypes.c_int * 3), ('nq', ctypes.c_int), ('weights_dev', vp), ('geom', Geometry), ('T_dev', vp), ('ncomp', ctypes.c_int),
                ('C_host', vp), ('values_dev', vp), ('scale_dev', vp), ('layer_begin', ctypes.c_int), ('layer_end', ctypes.c_int),
                ('owner_begin', ctypes.c_int), ('owner_end', ctypes.c_int), ('max_workgroups', ctypes.c_int), ('weights_positive', ctypes.c_int)]


GEOM_ISO = 1
GEOM_BOX = 2
GEOM_TAB = 3

# name -> (restype, argtypes); kept in step with include/nutils_hip.h (tests/test_abi.py parses the header)
SIGNATURES = {
    'nh_abi_version': (ctypes.c_int, []),
    'nh_last_error': (ctypes.c_char_p, []),
    'nh_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'nh_set_device': (ctypes.c_int, [ctypes.c_int]),
    'nh_device_info': (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), c_i64p, c_i64p]),
    'nh_malloc': (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_size_t]),
    'nh_free': (ctypes.c_int, [vp]),
    'nh_memcpy_h2d': (ctypes.c_int, [vp, vp, ctypes.c_size_t, vp]),
    'nh_memcpy_d2h': (ctypes.c_int, [vp, vp, ctypes.c_size_t, vp]),
    'nh_memset': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_size_t, vp]),
    'nh_stream_sync': (ctypes.c_int, [vp]),
    'nh_release_scratch': (ctypes.c_int, []),
    'nh_poly_tabulate': (ctypes.c_int, [vp, c_i64, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    'nh_structured_dofs': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                          ctypes.POINTER(ctypes.c_int), vp, c_i64, c_i64, vp, vp]),
    'nh_pattern_build': (ctypes.c_int, [ctypes.POINTER(PatternArgs), ctypes.POINTER(vp), vp]),
    'nh_pattern_free': (ctypes.c_int, [vp]),
    'nh_pattern_info': (ctypes.c_int, [vp, c_i64p, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), c_i64p, ctypes.POINTER(vp)]),
    'nh_pattern_fused_info': (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), c_i64p, ctypes.POINTER(ctypes.c_int)]),
    'nh_pattern_owner_info': (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), c_i64p, c_i64p]),
    'nh_pattern_union_count': (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, vp, vp, vp, c_i64p, vp]),
    'nh_pattern_union_fill': (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, vp, vp, vp, vp, vp, vp]),
    'nh_pattern_expanded_nnz': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, c_i64p]),
    'nh_pattern_expand': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]),
    'nh_assemble_matrix': (ctypes.c_int, [ctypes.POINTER(MatrixArgs), vp]),
    'nh_assemble_vector': (ctypes.c_int, [ctypes.POINTER(VectorArgs), vp]),
    'nh_assemble_terms': (ctypes.c_int, [ctypes.POINTER(TermsArgs), vp]),
    'nh_assemble_terms_multi': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.POINTER(TermsArgs)), vp]),
    'nh_assemble_matrix_terms': (ctypes.c_int, [ctypes.POINTER(MatrixTermsArgs), vp]),
    'nh_sample_eval': (ctypes.c_int, [ctypes.POINTER(EvalArgs), vp]),
    'nh_monomial_csr': (ctypes.c_int, [c_i64, vp, vp, vp, vp, ctypes.c_double, vp, vp]),
    'nh_factor_tensor': (ctypes.c_int, [ctypes.POINTER(FactorArgs), ctypes.POINTER(c_i64), vp]),
    'nh_factor_fetch': (ctypes.c_int, [vp, vp, vp]),
    'nh_monomial': (ctypes.c_int, [c_i64, vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, ctypes.c_double, vp, vp]),
    'nh_index_copy': (ctypes.c_int, [c_i64, vp, vp, vp, vp, vp]),
    'nh_pointwise_poly': (ctypes.c_int, [c_i64, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_int), vp, vp]),
    'nh_scatter_plan_build': (ctypes.c_int, [c_i64, c_i64, ctypes.c_int, vp, vp, vp, c_i64, ctypes.POINTER(vp), vp]),
    'nh_scatter_plan_free': (ctypes.c_int, [vp]),
    'nh_scatter_gather': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_int, vp, ctypes.c_int, vp]),
    'nh_point_expr': (ctypes.c_int, [c_i64, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(c_i64), c_i64, vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_int, vp]),
    'nh_point_forms': (ctypes.c_int, [ctypes.c_int, c_i64, ctypes.c_int, vp, vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), vp, vp, vp]),
    'nh_rationalize': (ctypes.c_int, [vp, c_i64, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]),
    'nh_p1hex_pattern': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), c_i64, c_i64, vp, vp, vp]),
    'nh_p1hex_laplace': (ctypes.c_int, [ctypes.POINTER(P1HexArgs), vp]),
    'nh_p1hex_apply': (ctypes.c_int, [ctypes.POINTER(P1HexArgs), vp, vp, ctypes.c_int, vp]),
    'nh_p2hex_matrix': (ctypes.c_int, [ctypes.POINTER(P2HexArgs), vp]),
    'nh_p2hex_pattern': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_int, vp, vp, vp]),
    'nh_p2hex_rowptr': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), c_i64, c_i64p]),
    'nh_p2hex_rows_uniform': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, vp]),
    'nh_p1hex_unit_matrix': (ctypes.c_int, [ctypes.POINTER(P1HexArgs), vp, vp]),
}

_lib = None


def load():
    '''Return the loaded library; raise if it has not been built.'''
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise NutilsHipError(f'{LIBPATH} not found: build it with `make -C nutils_amd/csrc` (or __graft_entry__.build()); '
                                 'nutils_amd has no CPU fallback')
        # Share ONE HIP runtime with PyTorch-ROCm (our allocator / stream owner): torch bundles its own
        # libamdhip64, and device pointers or streams cannot cross runtimes.  Importing torch first makes the
        # dynamic linker resolve our libamdhip64.so.7 dependency to the copy torch already mapped.
        if os.environ.get('NUTILS_HIP_STANDALONE', '0') != '1':
            import torch  # noqa: F401
        lib = ctypes.CDLL(LIBPATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.nh_abi_version() != 1:
            raise NutilsHipError('libnutils_hip.so ABI version mismatch')
        _lib = lib
    return _lib


def check(status):
    if status != 0:
        msg = load().nh_last_error().decode(errors='replace')
        raise NutilsHipError(f'libnutils_hip status {status}: {msg}')


TRACE = None  # list of entry-point names while a `trace()` block is active


class trace:
    '''`with _lib.trace() as calls:` records the name of every C-ABI entry point called inside the block (tests assert which kernels an
    assembly went through, e.g. that a plan reached nh_p1hex_laplace and not the generic nh_assemble_matrix).'''

    def __enter__(self):
        global TRACE
        self._outer, TRACE = TRACE, []
        return TRACE

    def __exit__(self, *exc):
        global TRACE
        if self._outer is not None:
            self._outer.extend(TRACE)
        TRACE = self._outer


def call(name, *args):
    if TRACE is not None:
        TRACE.append(name)
    check(getattr(load(), name)(*args))
