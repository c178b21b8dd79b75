'''ctypes face of oracle/c/libassemble_port.so (test infrastructure; see the C file's header).'''
import ctypes
import os
import numpy

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'c', 'libassemble_port.so')


def available():
    return os.path.exists(_PATH)


def laplace3d(shape, degree, T, gT, weights, verts=None, threads=0):
    '''T (nb, nq, 4), gT (8, nq, 4): tabulated basis / trilinear geometry basis (value, d/dxi).
    Returns values, rowptr, colidx, (t_loop, t_dedup).'''
    lib = ctypes.CDLL(_PATH)
    f = lib.port_laplace3d
    f.restype = ctypes.c_int64
    vp = ctypes.c_void_p
    f.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, vp]
    shape = numpy.asarray(shape, dtype=numpy.int32)
    T = numpy.ascontiguousarray(T, dtype=float)
    gT = numpy.ascontiguousarray(gT, dtype=float)
    weights = numpy.ascontiguousarray(weights, dtype=float)
    if verts is not None:
        verts = numpy.ascontiguousarray(verts, dtype=float)
    nd = [int(n) * degree + 1 for n in shape]
    ndofs = int(numpy.prod(nd))
    cap = int(numpy.prod([(2 * degree + 1) * n for n in nd]))  # upper bound of nnz
    values = numpy.empty(cap)
    rowptr = numpy.empty(ndofs + 1, dtype=numpy.int64)
    colidx = numpy.empty(cap, dtype=numpy.int64)
    timings = numpy.zeros(2)
    p = lambda a: None if a is None else a.ctypes.data_as(vp)
    nnz = f(p(shape), degree, T.shape[1], p(T), p(gT), p(verts), p(weights), p(values), p(rowptr), p(colidx), cap, threads, p(timings))
    if nnz < 0:
        raise RuntimeError(f'port_laplace3d failed ({nnz})')
    return values[:nnz].copy(), rowptr, colidx[:nnz].copy(), tuple(timings)


def form3d(shape, degree, C, T, gT, weights, verts=None, threads=0):
    '''Vector-valued constant-coefficient form (port_form3d): C (nc, 4, nc, 4), T (nb, nq, 4), gT (8, nq, 4).
    Returns values, rowptr, colidx, (t_loop, t_dedup).'''
    lib = ctypes.CDLL(_PATH)
    f = lib.port_form3d
    f.restype = ctypes.c_int64
    vp = ctypes.c_void_p
    f.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, vp]
    shape = numpy.asarray(shape, dtype=numpy.int32)
    C = numpy.ascontiguousarray(C, dtype=float)
    nc = C.shape[0]
    T = numpy.ascontiguousarray(T, dtype=float)
    gT = numpy.ascontiguousarray(gT, dtype=float)
    weights = numpy.ascontiguousarray(weights, dtype=float)
    if verts is not None:
        verts = numpy.ascontiguousarray(verts, dtype=float)
    nd = [int(n) * degree + 1 for n in shape]
    ndofs = int(numpy.prod(nd)) * nc
    cap = int(numpy.prod([(2 * degree + 1) * n for n in nd])) * nc * nc  # upper bound of nnz
    values = numpy.empty(cap)
    rowptr = numpy.empty(ndofs + 1, dtype=numpy.int64)
    colidx = numpy.empty(cap, dtype=numpy.int64)
    timings = numpy.zeros(2)
    p = lambda a: None if a is None else a.ctypes.data_as(vp)
    nnz = f(p(shape), degree, nc, p(C), T.shape[1], p(T), p(gT), p(verts), p(weights), p(values), p(rowptr), p(colidx), cap, threads, p(timings))
    if nnz < 0:
        raise RuntimeError(f'port_form3d failed ({nnz})')
    return values[:nnz].copy(), rowptr, colidx[:nnz].copy(), tuple(timings)
