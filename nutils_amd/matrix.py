'''Matrix hand-off (boundary C of the reference): the GPU-built CSR triplet goes
back to the host solver exactly the way the reference's evaluator hands it over
(/root/reference/src/nutils/matrix/__init__.py:20-151): ``assemble_csr(values,
rowptr, colidx, ncols)`` validates the triplet and calls
``backend.current.assemble``.  The validation is vectorised here (the reference
uses Python ``all()`` over numpy arrays); the error behaviour (MatrixError, same
conditions) is the reference's.  Solvers stay on the host (scipy), out of scope.
'''

import contextlib
import numpy


class MatrixError(Exception):
    '''General error message for matrix-related failure (matrix/_base.py:9-12).'''


class ScipyMatrix:
    '''scipy.sparse.csr_matrix wrapper with the subset of the reference's Matrix
    interface (matrix/_base.py, _scipy.py:11-12) that the assembly path touches.'''

    def __init__(self, core):
        self.core = core
        self.shape = core.shape

    def export(self, form):
        if form == 'csr':
            return self.core.data, self.core.indices, self.core.indptr
        if form == 'coo':
            coo = self.core.tocoo()
            return coo.data, (coo.row, coo.col)
        if form == 'dense':
            return self.core.toarray()
        raise NotImplementedError(f'cannot export ScipyMatrix to {form!r}')

    def __matmul__(self, other):
        return self.core @ other

    def submatrix(self, rows, cols):
        rows = numpy.asarray(rows)
        cols = numpy.asarray(cols)
        if rows.dtype == bool:
            rows, = rows.nonzero()
        if cols.dtype == bool:
            cols, = cols.nonzero()
        return ScipyMatrix(self.core[rows, :][:, cols])

    def solve(self, rhs=None, *, lhs0=None, constrain=None, **solverargs):
        '''Direct solve with constraints in the reference's convention: `constrain`
        is an array with NaN for free dofs (matrix/_base.py:100-190).'''
        import scipy.sparse.linalg
        n = self.shape[0]
        x = numpy.zeros(n) if lhs0 is None else numpy.array(lhs0, dtype=float)
        rhs = numpy.zeros(n) if rhs is None else numpy.asarray(rhs, dtype=float)
        if constrain is None:
            free = numpy.ones(n, dtype=bool)
        else:
            constrain = numpy.asarray(constrain)
            if constrain.dtype == bool:
                free = ~constrain.ravel()
            else:
                free = numpy.isnan(constrain)
                x[~free] = constrain[~free]
        if free.all():
            return x + scipy.sparse.linalg.spsolve(self.core.tocsc(), rhs - self.core @ x)
        b = (rhs - self.core @ x)[free]
        A = self.core[free, :][:, free].tocsc()
        x[free] += scipy.sparse.linalg.spsolve(A, b)
        return x


class _ScipyBackend:
    @staticmethod
    def assemble(values, rowptr, colidx, ncols):
        import scipy.sparse
        return ScipyMatrix(scipy.sparse.csr_matrix((values, colidx, rowptr), (len(rowptr) - 1, ncols)))

    @staticmethod
    def assemble_trusted(values, rowptr, colidx, ncols):
        '''Same matrix for an index pair that has been through `assemble_csr` before (re-assembly of a Newton step: only the values
        are new): skips scipy's O(nnz) index scans and the copy it makes of the index arrays.'''
        import scipy.sparse
        core = scipy.sparse.csr_matrix((len(rowptr) - 1, ncols), dtype=values.dtype)
        core.data, core.indices, core.indptr = values, colidx, rowptr
        core.has_sorted_indices = True
        core.has_canonical_format = True
        return ScipyMatrix(core)


class _Backend:
    '''``matrix.backend`` selector: any object with ``.assemble(values, rowptr,
    colidx, ncols)`` is accepted (matrix/__init__.py:20-27; fake-backend precedent
    /root/reference/tests/test_matrix.py:6-23).'''

    current = _ScipyBackend

    @contextlib.contextmanager
    def __call__(self, matrix):
        if isinstance(matrix, str):
            if matrix.lower() not in ('scipy', 'auto'):
                raise ValueError(f'matrix backend {matrix!r} is not available in nutils_amd')
            matrix = _ScipyBackend
        if not hasattr(matrix, 'assemble'):
            raise ValueError('matrix backend does not have an assemble function')
        previous, _Backend.current = _Backend.current, matrix
        try:
            yield matrix
        finally:
            _Backend.current = previous


backend = _Backend()


def _check_triplet(values, rowptr, colidx, ncols):
    '''The contract a backend may rely on (the conditions of matrix/__init__.py:47-69, each raising MatrixError): values a vector with
    one entry per column index; rowptr an integer vector that starts at 0, never decreases and ends at nnz; column indices integers
    below ncols that do not decrease inside a row.  One pass of numpy per condition.'''
    nnz = values.shape[0] if values.ndim == 1 else -1
    if nnz < 0:
        raise MatrixError(f'values must be a vector, got {values.ndim} axes')
    if rowptr.ndim != 1 or rowptr.dtype.kind not in 'ui' or not rowptr.size:
        raise MatrixError('row pointers must be a non-empty integer vector')
    if rowptr[0] != 0 or rowptr[-1] != nnz or numpy.any(numpy.diff(rowptr) < 0):
        raise MatrixError(f'row pointers must rise from 0 to the number of values ({nnz})')
    if colidx.ndim != 1 or colidx.dtype.kind not in 'ui' or colidx.size != nnz:
        raise MatrixError(f'column indices must be an integer vector of length {nnz}')
    if nnz and int(colidx.max()) >= ncols:
        raise MatrixError(f'column index {int(colidx.max())} outside the {ncols} columns')
    if nnz > 1:
        drops = numpy.flatnonzero(colidx[1:] < colidx[:-1]) + 1  # allowed only where a new row starts
        if drops.size and not numpy.isin(drops, rowptr).all():
            raise MatrixError('column indices decrease inside a row')


def assemble_csr(values, rowptr, colidx, ncols):
    '''Create sparse matrix from CSR sparse data (matrix/__init__.py:30-70): validate, then hand over to the current backend.'''
    values, rowptr, colidx = numpy.asarray(values), numpy.asarray(rowptr), numpy.asarray(colidx)
    ncols = ncols.__index__()
    _check_triplet(values, rowptr, colidx, ncols)
    return backend.current.assemble(values, rowptr, colidx, ncols)


def reassemble_csr(values, rowptr, colidx, ncols):
    '''`assemble_csr` for a (rowptr, colidx) pair that a previous `assemble_csr` call has validated -- the situation of every
    re-assembly inside a Newton or time loop, where the reference repeats the O(nnz) checks (matrix/__init__.py:47-69).
    Only the values are checked; backends without a trusted entry point get the ordinary one.'''
    values = numpy.asarray(values)
    if not (values.ndim == 1 and len(values) == len(colidx)):
        raise MatrixError('assemble received invalid values')
    trusted = getattr(backend.current, 'assemble_trusted', None)
    return trusted(values, rowptr, colidx, ncols) if trusted else backend.current.assemble(values, rowptr, colidx, ncols)


def compress_indices(indices, length):
    '''rowidx -> rowptr (numeric.py:687-711), vectorised.'''
    indices = numpy.asarray(indices)
    if len(indices) and (indices[0] < 0 or indices[-1] >= length):
        raise ValueError('indices are out of bounds')
    if len(indices) > 1 and (numpy.diff(indices) < 0).any():
        raise ValueError('indices are not monotomically increasing')
    return numpy.searchsorted(indices, numpy.arange(length + 1)).astype(numpy.int64)


def assemble_coo(values, rowidx, nrows, colidx, ncols):
    '''Create sparse matrix from COO sparse data (matrix/__init__.py:73-93).'''
    return assemble_csr(values, compress_indices(rowidx, nrows), colidx, ncols)


def assemble_block_csr(blocks):
    '''Create sparse block matrix from stacked CSR sparse data
    (matrix/__init__.py:103-151).  The reference merges multi-block rows with a
    Python loop over matrix rows; here the merge is one stable sort of the
    concatenated (row, block, position) keys.'''
    ncols = sum(n for *_, n in blocks[0])
    all_values, all_rows, all_cols, nrows_total = [], [], [], 0
    for row in blocks:
        nrows = len(row[0][1]) - 1
        col_offset = 0
        for block_values, block_rowptr, block_colidx, block_ncols in row:
            block_rowptr = numpy.asarray(block_rowptr)
            if len(block_rowptr) - 1 != nrows:
                raise MatrixError('sparse blocks have inconsistent row sizes')
            all_values.append(numpy.asarray(block_values))
            all_rows.append(numpy.repeat(numpy.arange(nrows) + nrows_total, numpy.diff(block_rowptr)))
            all_cols.append(numpy.asarray(block_colidx) + col_offset)
            col_offset += block_ncols
        if col_offset != ncols:
            raise MatrixError('sparse blocks have inconsistent column sizes')
        nrows_total += nrows
    values = numpy.concatenate(all_values)
    rows = numpy.concatenate(all_rows)
    cols = numpy.concatenate(all_cols)
    order = numpy.argsort(rows, kind='stable')  # blocks were appended in column order, so columns stay sorted per row
    return assemble_csr(values[order], compress_indices(rows[order], nrows_total), cols[order], ncols)
