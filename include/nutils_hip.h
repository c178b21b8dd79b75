/* nutils_hip.h -- C ABI of libnutils_hip.so, the MI355X (gfx950) element-integration
 * and sparse-assembly backend for Nutils.
 *
 * The reference (/root/reference, pure Python) has no FFI for this path: the work is
 * done by a generated numpy-per-element Python loop.  Each entry point below names the
 * reference code it replaces (paths relative to /root/reference/src/nutils).  The
 * boundary is plain C: opaque handles, device/host pointers, sizes, an int status
 * (0 = ok, negative = error, message via nh_last_error) -- the convention the reference
 * itself uses for its only native binding (matrix/_mkl.py:29-42,76-82), loaded with
 * ctypes like _util.py:195-237 (loadlib).
 *
 * Conventions
 *   - all "dev" pointers are HIP device pointers (hipMalloc / torch.cuda allocations);
 *     "host" pointers are ordinary memory.  `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).  Calls are asynchronous on `stream` unless stated.
 *   - floating point data is IEEE f64 throughout (the reference computes in float64);
 *     index OUTPUTS are int64 (the reference's CSR index arrays are int64 and are
 *     compared bit-exactly); connectivity INPUTS are int32 on the device.
 *   - tabulated basis layout  T[fn][q][S], S = 1 + ndims: value, then d/dxi_j.
 *   - element-local ordering, dof numbering and CSR ordering follow the reference:
 *     CSR rows/cols sorted lexicographically, structural zeros retained, flat vector
 *     dof = scalar_dof * ncomp + comp (function.py:2598-2627).
 *   - thread model: one host thread per device context (the reference never threads;
 *     under NUTILS_NPROCS>1 it forks -- do not initialise before a fork).
 *   - library-owned state is per PROCESS, not per stream: the scratch of local matrices of NH_MATRIX_GATHER (grows to
 *     the largest pattern seen, 1.07 GB for the 128^3 trilinear mesh; nh_release_scratch() returns it), the staged
 *     basis tables of the thread-per-element passes and the parameter ring of nh_assemble_terms_multi are shared
 *     by all calls and carry no locks or events.  REQUIREMENT: all assembly entry points (nh_assemble_*, nh_p1hex_*,
 *     nh_p2hex_*) are issued from ONE host thread onto ONE stream at a time; only nh_index_copy, nh_memcpy_* and
 *     nh_monomial* (no library state) may run on a second stream beside them -- which is how nutils_amd/solver.py
 *     overlaps the copy of the Jacobian entries with the residual of the same step.
 */
#ifndef NUTILS_HIP_H
#define NUTILS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NH_OK 0
#define NH_EINVAL (-1)  /* invalid argument */
#define NH_EHIP (-2)    /* HIP runtime error (message has the HIP error string) */
#define NH_ELIMIT (-3)  /* problem exceeds a compiled-in limit of the kernel */
#define NH_ENOMEM (-4)

#define NH_ABI_VERSION 1

/* ---- runtime ------------------------------------------------------------------- */
int nh_abi_version(void);
const char *nh_last_error(void);
int nh_device_count(int *count);
int nh_set_device(int device);
/* name: caller buffer of namelen bytes; cus, lds_bytes, hbm_bytes may be NULL */
int nh_device_info(int device, char *name, size_t namelen, int *cus, int64_t *lds_bytes, int64_t *hbm_bytes);
int nh_malloc(void **dev, size_t bytes);
int nh_free(void *dev);
int nh_memcpy_h2d(void *dev, const void *host, size_t bytes, void *stream);
int nh_memcpy_d2h(void *host, const void *dev, size_t bytes, void *stream);
int nh_memset(void *dev, int byte, size_t bytes, void *stream);
int nh_stream_sync(void *stream);
int nh_release_scratch(void);  /* frees the library-owned scratch buffers (synchronises the device) */

/* ---- K2: basis tabulation -------------------------------------------------------
 * replaces nutils_poly eval_outer / GradPlan as called from generated code
 * (evaluable.py:4366-4374 Polyval, :4584-4653 PolyGrad).
 * coeffs[nfn][ncoeffs] in the reference coefficient order (evaluable.py:4331-4340),
 * points[nq][ndims]  ->  T[nfn][nq][1+ndims]. */
int nh_poly_tabulate(const double *coeffs_dev, int64_t nfn, int ncoeffs, const double *points_dev, int nq, int ndims,
                     double *T_dev, void *stream);

/* ---- K1: structured dof maps ----------------------------------------------------
 * replaces StructuredBasis.f_dofs_coeffs' integer part (function.py:3080-3093:
 * divmod unravel, Range + offsets, RavelIndex/Ravel).  shape[ndims] elements per axis
 * (element index: last axis fastest), start_dev[axis][shape[axis]] first dof per element
 * (concatenated over axes), per-axis dofs-per-element nloc[axis], per-axis dof counts
 * ndofs_axis[axis] (dofs wrap modulo ndofs_axis).  Output dofs[nelems][prod nloc] int32. */
int nh_structured_dofs(int ndims, const int *shape, const int *nloc, const int *ndofs_axis, const int *start_dev,
                       int64_t elem_begin, int64_t nelems, int32_t *dofs_dev, void *stream);

/* ---- K6: sparsity pattern --------------------------------------------------------
 * replaces Array.assparse's unique(flatindex) = ArgSort(stable)+UniqueMask+UniqueInverse
 * and as_csr's CompressIndices (evaluable.py:588-616, 5560-5682; numeric.py:687-711).
 * Builds the SCALAR pattern (rows = test dofs, cols = trial dofs) row-wise on the
 * device, plus the element map emap[e][m][n] = position of trial dof n of element e in
 * the sorted scalar row of test dof m.  Ragged bases pass offsets (prefix sums of dofs
 * per element, int64[nelems+1], host... see fields). */
typedef struct nh_pattern nh_pattern; /* opaque, owns device memory */

typedef struct {
  int64_t nelems;
  int64_t nrows, ncols;        /* scalar dof counts of test / trial basis */
  int nbt, nbr;                /* dofs per element if uniform; 0 if ragged (use offsets) */
  const int32_t *tdofs_dev;    /* test dofs, concatenated per element */
  const int32_t *rdofs_dev;    /* trial dofs (may alias tdofs_dev) */
  const int64_t *toff_dev;     /* ragged: int64[nelems+1] prefix sums, else NULL */
  const int64_t *roff_dev;
} nh_pattern_args;

int nh_pattern_build(const nh_pattern_args *args, nh_pattern **out, void *stream);
int nh_pattern_free(nh_pattern *p);
/* scalar nnz, and device pointers owned by the pattern (valid until nh_pattern_free);
 * eoff (ragged bases only, else NULL): int64[nelems+1] prefix sums of nbt_e*nbr_e indexing emap */
int nh_pattern_info(const nh_pattern *p, int64_t *nnz_scalar, const int64_t **srowptr_dev, const int32_t **scolidx_dev,
                    const int32_t **emap_dev, int64_t *emap_len, const int64_t **eoff_dev);
/* Expanded CSR index arrays for nct x ncr components with block mask[nct][ncr] (host,
 * nonzero = block present; NULL = all): rowptr_dev int64[nrows*nct+1], colidx_dev
 * int64[nnz]; nnz returned by nh_pattern_expanded_nnz.  Hand-back format of
 * matrix/__init__.py:30-70 (assemble_csr) -- indices int64, rows sorted, cols strictly
 * increasing within a row. */
/* owner blocks of NH_MATRIX_FUSED built for this pattern so far: number of row blocks (0: none yet, or the plan does not apply), rows of the largest
 * block, element visits over all blocks (>= nelems: elements on block borders are recomputed), and the element routine of the last fused
 * launch: -1 none yet, 0 the tabulated any-element routine, 1 / 2 the sum-factorised routine for trilinear hexahedra at the 2 x 2 x 2 Gauss
 * points (recognised from the tables of the launch; 2: with a mass term) */
int nh_pattern_fused_info(const nh_pattern *p, int *nblocks, int *rows_per_block, int64_t *nvisits, int *routine);
/* row tasks of NH_MATRIX_FUSED for vector-valued blocks built for this pattern so far (nh_owner.hip): row blocks (0: none yet, or the plan does not
 * apply), node rows per block, element visits of all blocks, chunks of 64 contributions */
int nh_pattern_owner_info(const nh_pattern *p, int *nblocks, int *rows_per_block, int64_t *nvisits, int64_t *nchunks);
int nh_pattern_expanded_nnz(const nh_pattern *p, int nct, int ncr, const unsigned char *mask, int64_t *nnz);
int nh_pattern_expand(const nh_pattern *p, int nct, int ncr, const unsigned char *mask, int64_t *rowptr_dev,
                      int64_t *colidx_dev, void *stream);

/* Union of up to 8 sorted-unique CSR patterns with the same number of rows (the per-sample matrices of one integral: volume + Nitsche / Robin
 * boundary terms; replaces the unique of evaluable.py:5560-5682 over the concatenated keys): a row of the union is the merge of the parts' sorted
 * column lists -- no sort.  nh_pattern_union_count fills rowptr_u_dev [nrows + 1] and returns the union's nnz (host, synchronises the stream);
 * nh_pattern_union_fill writes colidx_u_dev [nnz] and, for every entry k of part i, its position pos_dev[i][k] in the union. */
int nh_pattern_union_count(int nparts, int64_t nrows, const int64_t *const *rowptr_dev, const int64_t *const *colidx_dev, int64_t *rowptr_u_dev, int64_t *nnz, void *stream);
int nh_pattern_union_fill(int nparts, int64_t nrows, const int64_t *const *rowptr_dev, const int64_t *const *colidx_dev, const int64_t *rowptr_u_dev, int64_t *colidx_u_dev,
                          int64_t *const *pos_dev, void *stream);

/* ---- geometry descriptor ---------------------------------------------------------
 * replaces _TransformsCoords/_Jacobian lowering + numeric.inv + linalg.det
 * (function.py:1162-1181,1284-1295; evaluable.py:1403-1490; numeric.py:221-241). */
#define NH_GEOM_ISO 1    /* x = sum_a N_a(xi) X[gdofs[e][a]]; gT = tabulated geometry basis [ngb][nq][S] */
#define NH_GEOM_BOX 2    /* x = origin[e] + size[e] * xi  (axis aligned; rectilinear + hierarchical refinements) */
#define NH_GEOM_TAB 3    /* Jacobian (and coordinates) tabulated per element and point by the producer of the mesh: geometries
                            the kernels do not evaluate themselves (NURBS maps through transform chains, trimmed cells ...) */

typedef struct {
  int kind;
  int ngb;                   /* ISO: geometry basis functions per element (uniform) */
  const double *gT_dev;      /* ISO: [ngb][nq][S] */
  const int32_t *gdofs_dev;  /* ISO: [nelems][ngb] */
  const double *verts_dev;   /* ISO: [nverts][ndims] */
  const double *origin_dev;  /* BOX: [nelems][ndims] */
  const double *size_dev;    /* BOX: [nelems][ndims] */
  const double *jac_dev;     /* TAB: [nelems][nq][ndims][ndims]  d x_i / d xi_j */
  const double *x_dev;       /* TAB: [nelems][nq][ndims] or NULL (only needed by nh_sample_eval) */
  int bnd_axis;              /* -1: volume measure |det J|.  a >= 0: the points lie on the reference face xi_a = const and the
                                measure is the surface measure |det J| |J^-T e_a| (boundary integrals, topology.boundary) */
} nh_geometry;

/* ---- basis-on-elements descriptor ------------------------------------------------ */
typedef struct {
  int nb;                    /* functions per element if uniform, 0 if ragged */
  const double *T_dev;       /* tabulated functions [nfn][nq][S] */
  const int32_t *dofs_dev;   /* concatenated element dofs */
  const int64_t *off_dev;    /* ragged: int64[nelems+1] (dofs AND first-function offsets), else NULL */
  const int32_t *tab_dev;    /* uniform nb, several tables: table index per element (first fn = tab*nb); NULL = table 0.
                                Ragged: ignored (first function of element e = off[e]). */
} nh_basis;

/* ---- K3+K4+K5: matrix assembly ---------------------------------------------------
 * replaces the generated element loop (evaluable.py:6773-6786) for a bilinear integrand
 *   A[(m,c),(n,d)] = sum_q w_q |det J_q| sum_{a,b} Dt[q,m,a] C[c,a,d,b] Dr[q,n,b]
 * (D[.,.,0] = value, D[.,.,1+i] = d/dx_i: Basis.lower function.py:2758-2762, _Gradient
 * :1221-1231; einsum contraction evaluable.py:1885-1886, 6414-6505; sample.py:951-956)
 * and the scatter/accumulate of its sparse dedup (evaluable.py:603-605; numeric.accumulate
 * numeric.py:434-460) into `values_dev` laid out by the pattern. values are ACCUMULATED:
 * zero them first (nh_memset) for a fresh assembly. */
typedef struct {
  int64_t nelems;
  const int32_t *elist_dev;  /* optional element subset (bucket), NULL = 0..nelems-1; nelems counts the list */
  int ndims, nq;
  const double *weights_dev; /* [nq] */
  nh_geometry geom;
  nh_basis test, trial;
  int nct, ncr;              /* components of test / trial field */
  const double *C_host;      /* [nct][S][ncr][S] constant coefficient tensor (host memory) */
  const unsigned char *mask_host; /* [nct][ncr] block mask used for the pattern, NULL = all */
  const int64_t *srowptr_dev;/* scalar pattern */
  const int32_t *emap_dev;   /* element map from nh_pattern_info */
  const int64_t *eoff_dev;   /* ragged: int64[nelems+1] prefix sums of nbt_e*nbr_e, else NULL */
  double *values_dev;        /* [nnz of the expanded pattern] */
  const double *scale_dev;   /* optional pointwise factor of the integrand [nelems][nq] (a coefficient function evaluated at
                                the quadrature points), NULL = 1.  With elist_dev, scale and emap are indexed by LIST position. */
  int flags;                 /* NH_MATRIX_* bits */
  const double *cq_dev;      /* optional coefficient tensor PER QUADRATURE POINT, [nelems][nq][nct][S][ncr][S] (indexed like scale_dev),
                                replacing C_host (which then only provides the block mask): forms whose coefficients depend on a
                                field and its gradient at the point -- the product-rule term kappa'(u) phi_n grad u . grad phi_m
                                of a Newton Jacobian, advection with a computed velocity.  NULL = constant form. */
  int grid_shape[3];         /* NH_MATRIX_FIRST_TOUCH: elements per axis of the structured mesh (element id = last axis fastest) */
  int nodes_per_axis;        /* NH_MATRIX_FIRST_TOUCH: p + 1 local nodes per axis of the C0 ('std') basis, local order first axis slowest */
  const nh_pattern *pattern; /* optional: the handle the pattern pointers above come from.  With ragged bases (hierarchical / imported:
                                PlainBasis function.py:2881-2913) and no elist_dev the elements are launched per SIZE CLASS of the pattern
                                (functions per element <= 8, 16, 24, 32, 48, 64 ...), each class with the LDS footprint of its own largest
                                element; scale_dev / cq_dev stay indexed by element.  NULL: one launch sized by the largest element. */
} nh_matrix_args;

#define NH_MATRIX_EXCLUSIVE 1        /* no two elements of this launch touch the same CSR entry (one colour of an element
                                        colouring): entries are added with plain loads/stores -- deterministic -- not atomics */
#define NH_MATRIX_EMAP_BY_ELEMENT 2  /* with elist_dev: emap (and the pattern) cover ALL elements, index it by element id */
#define NH_MATRIX_NO_MFMA 4          /* force the generic one-wave-per-element VALU kernel */
#define NH_MATRIX_FIRST_TOUCH 32     /* with EXCLUSIVE | EMAP_BY_ELEMENT, colours = element index parities launched in lexicographic order
                                        (last axis fastest) on a FRESH value array of a structured C0 basis (grid_shape, nodes_per_axis):
                                        an entry whose two nodes do not both lie on a face shared with an element of an earlier colour is
                                        STORED, not added -- the caller does not zero-fill the values, and the ~70 % of the entries that
                                        receive a single contribution are not read */

#define NH_MATRIX_GATHER 64          /* deterministic owner-side reduction instead of atomics (needs `pattern`, all elements of the pattern in one
                                        call): the local matrices go to a scratch array, element by element, and every CSR entry is then
                                        summed ONCE over its contributions in ascending (element, m, n) order -- the order in which the
                                        reference's numpy.add.at / numeric.accumulate adds them (numeric.py:434-460, evaluable.py:603-605),
                                        so repeated assemblies are bit-identical.  The gather map is built on the first such call and cached
                                        in the pattern handle (one device sort of the element map). */

#define NH_MATRIX_STORE 128          /* with NH_MATRIX_GATHER or NH_MATRIX_FUSED: the sums are STORED, values_dev is not read (first term on a
                                        fresh array: no zero fill, no read-modify-write) */

#define NH_MATRIX_FUSED 256          /* owner blocks: ONE pass without scratch array or global atomics (needs `pattern`, all of its elements in
                                        one call, no elist, test and trial on one dof array).  The dofs are clustered by the Morton code of the
                                        centroid of the first element that contains them; Morton boxes of at most R rows form a block, and a
                                        block recomputes what it needs of every element touching one of its rows and writes each of its CSR rows
                                        once.  SCALAR blocks on small uniform bases (2 .. 9 functions per element): the rows of a block are ONE
                                        set of accumulators in LDS; the visiting elements add their local matrices in TURNS -- turn t holds the
                                        t-th visitor of every row in ascending (element, m) order, so no two threads meet in a row, a workgroup
                                        barrier separates the turns -- which is the order of NH_MATRIX_GATHER, i.e. of the reference's
                                        numpy.add.at: bit-identical to the gather path, 1.4 x the algorithmic bytes instead of 4.6 x.  Trilinear hexahedra at
                                        the 2 x 2 x 2 Gauss points with a form kappa grad.grad + mass phi phi (recognised from the tables passed)
                                        take the sum-factorised element routine of nh_p1hex_laplace (an exactly singular element then gives inf /
                                        NaN instead of numeric.inv's all-NaN inverse).  VECTOR-VALUED blocks (nct = ncr = 2 or 3 on trilinear
                                        hexahedra, bilinear or biquadratic quadrilaterals, one table set for test and trial): LDS holds the
                                        physical gradients of the visiting elements, a lane sums the Gram matrix of one contribution (element,
                                        m, n), a segmented sum over the lanes of a scalar entry forms it once and the form tensor is applied per
                                        entry (nh_owner.hip).  Every sum is formed in an order fixed by the plan: repeated assemblies are
                                        bit-identical.  The plan is built on the first such call and cached in the pattern handle.  Launches the
                                        flag does not apply to take the default path (atomics; with NH_MATRIX_STORE after a zero fill of the
                                        block's values). */

int nh_assemble_matrix(const nh_matrix_args *args, void *stream);

/* ---- vector / functional assembly, point evaluation --------------------------------
 * r[(m,c)] += sum_q w_q |det J_q| sum_a Dt[q,m,a] F[q,c,a],
 * F[q,c,a] = f[c][a] + sum_{d,b} C[c,a,d,b] U[q,d,b],  U[q,d,b] = sum_n Dr[q,n,b] u[rdofs[n]][d]
 * replaces Inflate/Assemble scatter (evaluable.py:3341-3495, 3552-3645; numpy.add.at /
 * numeric.accumulate) of a linear-form element loop.  With out_scalar_dev != NULL also
 * accumulates the functional  sum_q w|J| (f0 + 1/2 sum U[q,c,a] (C U)[q,c,a])  there
 * (Sample.integrate of a scalar, sample.py:160-175).  u_dev / C_host may be NULL. */
typedef struct {
  int64_t nelems;
  const int32_t *elist_dev;
  int ndims, nq;
  const double *weights_dev;
  nh_geometry geom;
  nh_basis test, trial;
  int nct, ncr;
  const double *C_host;      /* [nct][S][ncr][S] or NULL */
  const double *f_host;      /* [nct][S] constant source or NULL */
  const double *u_dev;       /* [ncols][ncr] trial coefficients or NULL */
  double *out_dev;           /* [nrows][nct] accumulated, or NULL */
  double f0;                 /* constant integrand of the functional */
  double *out_scalar_dev;    /* [1] accumulated, or NULL */
  const double *scale_dev;   /* optional pointwise factor [nelems][nq] (list position with elist_dev), NULL = 1 */
  double *local_dev;         /* NULL, or: local vectors stored element-major instead of atomics into out_dev (see nh_block.local_dev) */
} nh_vector_args;

int nh_assemble_vector(const nh_vector_args *args, void *stream);

/* ---- fused linear-form assembly: all terms of a residual in ONE element loop ----------
 * The reference evaluates the derivative of a functional as ONE generated element loop per
 * block (evaluable.py:6773-6786 with the Inflate/Assemble scatter :3341-3495; solver.py:334-386
 * assembles the residual of every trial block per Newton step): every term of the integrand
 * is evaluated at the same points inside that loop.  nh_assemble_terms is that loop:
 *   r_blk[(m,c)] += sum_q w_q |det J_q| sum_a Dt_blk[q,m,a] sum_{terms t of blk} g_t(q) F_t[q,c,a],
 *   F_t[q,c,a] = f_t[c][a] + sum_{d,b} C_t[c,a,d,b] U_{field(t)}[q,d,b],
 *   g_t(q)     = scale_t[e][q] * poly_t(values of up to 4 scalar fields at q)      (each factor optional)
 * for up to 2 output blocks (test spaces with their own out vector: the residual blocks of a
 * multi-field system), 6 fields and 32 terms sharing one sample (element list, points,
 * geometry).  Several elements are processed per workgroup (lanes over (element, point), then
 * over (element, test function)), so bases with few functions per element do not leave lanes idle
 * the way the one-wave-per-element kernel of nh_assemble_vector does.  Replaces one
 * nh_assemble_vector launch per term plus the nh_sample_eval / nh_pointwise_poly passes that
 * produced their scale arrays. */
typedef struct {
  nh_basis basis;
  const double *u_dev;       /* [ndofs][ncomp] coefficients */
  int ncomp;
} nh_field;

typedef struct {
  nh_basis test;
  int nct;                   /* components of the test space */
  double *out_dev;           /* [nrows][nct], accumulated */
  double *local_dev;         /* NULL, or: the local vectors are STORED here, element-major ([position of (e, m) in test.dofs_dev][nct]), instead of
                                being added into out_dev with atomics -- the first half of the deterministic scatter, see nh_scatter_plan_build */
} nh_block;

typedef struct {
  int nvars, nterms;         /* value = sum_t coeffs[t] prod_v x_v^powers[t][v], x_v = value of component comp[v] of fields[field[v]] at the point */
  int field[4], comp[4];
  const double *coeffs_host; /* [nterms] */
  const int *powers_host;    /* [nterms][nvars] */
} nh_point_poly;

typedef struct {
  int block;                 /* output block */
  int field;                 /* field the form is applied to, -1: source term (f only) */
  int poly;                  /* pointwise polynomial factor, -1: none */
  const double *C_host;      /* [nct][S][ncr][S] or NULL */
  const double *f_host;      /* [nct][S] or NULL */
  const double *scale_dev;   /* [nelems][nq] (list position with elist_dev) or NULL */
  int qs_field_t, qs_field_r; /* with qs_B_host: the term is also multiplied by s_q = sum_ab B[a][b] U_t[q,a] U_r[q,b] of two scalar fields */
  const double *qs_B_host;   /* [S][S] or NULL (the point factor of energy Hessians: d2/du2 of kappa(u) |grad u|^2 / 2 and the like) */
} nh_term;

typedef struct {
  int64_t nelems;
  const int32_t *elist_dev;
  int ndims, nq;
  const double *weights_dev;
  nh_geometry geom;
  int nfields;
  const nh_field *fields;
  int nblocks;
  const nh_block *blocks;
  int nterms;
  const nh_term *terms;
  int npolys;
  const nh_point_poly *polys;
} nh_terms_args;

int nh_assemble_terms(const nh_terms_args *args, void *stream);
/* The term lists of several samples (the volume sample and the sides of the boundary of one residual: the reference fuses all loops of
 * equal length under one id, evaluable.py:6841-6895, but still runs one loop per sample) in ONE launch: every list gets a range of
 * workgroups.  Same result as `count` nh_assemble_terms calls (the outputs are accumulated either way); lists of another dimension than
 * the first and lists of more than 2^20 elements are launched on their own. */
int nh_assemble_terms_multi(int count, const nh_terms_args *const *lists, void *stream);

/* ---- fused bilinear-form assembly: all matrix terms of a block in ONE element loop ----------
 * Counterpart of nh_assemble_terms for the Jacobian blocks of a Newton step (solver.py:334-386;
 * derivative of the residual, evaluable.py:5693-5751): all terms share test / trial basis and the
 * pattern, their coefficient tensors are summed PER POINT inside the kernel,
 *   A[(m,c),(n,d)] += sum_q w_q |det J_q| sum_{a,b} Dt[q,m,a] Cq[q][c,a,d,b] Dr[q,n,b],  Cq = sum_t g_t(q) C_t(q),
 * with g_t as in nh_assemble_terms (scale array, pointwise polynomial of field values) and C_t(q)
 *   kind 0: the constant tensor C_host,
 *   kind 1: scalar fields, Cq[a][0] += sum_b B[a][b] U_field[q][b]         (product-rule term on the value slot of the trial function)
 *   kind 2: scalar fields, Cq[a][b] += L[a] sum_x B[x][b] U_field[q][x]    (product-rule term on the test side)
 * (kinds 1, 2: the per-point tensors that nh_matrix_args.cq_dev takes from the caller, here evaluated in the kernel: the terms
 * kappa'(u) phi_n grad u . grad phi_m of a quasi-linear Jacobian.)  Several elements per workgroup: lanes over (element, point) for the
 * pointwise part, over (element, m, n) for the contraction.  values are ACCUMULATED with the layout of nh_assemble_matrix. */
typedef struct {
  int kind;                  /* 0, 1, 2 (above) */
  int field;                 /* kinds 1, 2: the field U; else -1 */
  int poly;                  /* pointwise polynomial factor, -1: none */
  const double *C_host;      /* [nct][S][ncr][S]: C (kind 0) or B (kinds 1, 2) */
  const double *L_host;      /* kind 2: [nct][S] */
  const double *scale_dev;   /* [nelems][nq] or NULL */
  int qs_field_t, qs_field_r; /* as in nh_term */
  const double *qs_B_host;
} nh_matrix_term;

typedef struct {
  int64_t nelems;
  const int32_t *elist_dev;
  int ndims, nq;
  const double *weights_dev;
  nh_geometry geom;
  nh_basis test, trial;
  int nct, ncr;
  const unsigned char *mask_host; /* [nct][ncr] block mask used for the pattern, NULL = all */
  const int64_t *srowptr_dev;
  const int32_t *emap_dev;
  const int64_t *eoff_dev;
  double *values_dev;
  int flags;                 /* NH_MATRIX_EMAP_BY_ELEMENT, NH_MATRIX_GATHER, NH_MATRIX_STORE */
  int nfields;
  const nh_field *fields;
  int nterms;
  const nh_matrix_term *terms;
  int npolys;
  const nh_point_poly *polys;
  const nh_pattern *pattern; /* optional pattern handle: with NH_MATRIX_GATHER (| NH_MATRIX_STORE) in flags, scalar blocks on small uniform bases
                                (2 x 2 ... 9 x 9 local matrices, at most two scalar fields on the test basis) are assembled by a thread-per-element
                                pass + the owner-side reduction of nh_assemble_matrix; other blocks ignore NH_MATRIX_GATHER */
} nh_matrix_terms_args;

int nh_assemble_matrix_terms(const nh_matrix_terms_args *args, void *stream);

/* Sample.eval / bind (sample.py:192-232, _ConcatenatePoints.lower :966-975;
 * LoopConcatenate evaluable.py:5383-5508): values at all quadrature points, element
 * major.  Any output may be NULL.  x[e][q][ndims], detj[e][q],
 * U[e][q][ncr][S] (value and physical gradient of the field u). */
typedef struct {
  int64_t nelems;
  const int32_t *elist_dev;  /* optional element subset, NULL = 0..nelems-1; outputs are indexed by list position */
  int ndims, nq;
  nh_geometry geom;
  nh_basis trial;
  int ncr;
  const double *points_dev;  /* [nq][ndims] reference coordinates (needed for BOX geometry x) */
  const double *u_dev;
  double *x_dev, *detj_dev, *U_dev;
} nh_eval_args;

int nh_sample_eval(const nh_eval_args *args, void *stream);

/* ---- fast path: structured P1 hexahedra, scalar Laplace ----------------------------
 * Same result as nh_pattern_* + nh_assemble_matrix for the 3-D trilinear 'std' basis on
 * mesh.rectilinear (mesh.py:34-60; StructuredBasis function.py:3080-3100) with 2-point
 * Gauss per axis and isoparametric P1 (or uniform) geometry, but WRITE-ONCE: each
 * workgroup owns a column tile of dof rows, recomputes the element matrices that touch it,
 * reduces them in LDS and streams the finished CSR rows to HBM -- no global atomics, no
 * zero-fill, no element map.  The pattern is the reference's (sorted unique, structural
 * zeros kept; (3n+1)^3 law) in closed form: nh_p1hex_pattern writes rowptr (re-based to
 * row_begin, row_end-row_begin+1 entries) and colidx for rows [row_begin, row_end).
 * layer_*: element layers along axis 0 that contribute values; plane_*: dof planes along
 * axis 0 whose rows are written (multi-GPU slabs, nutils_amd/partition.py); values_dev
 * is indexed with the offsets of the FULL local mesh pattern. */
typedef struct {
  int shape[3];
  int layer_begin, layer_end;
  int plane_begin, plane_end;
  const double *verts_dev;   /* [(n0+1)(n1+1)(n2+1)][3]; NULL: x = origin + scale * vertex index */
  double origin[3], scale[3];
  double gauss_x[2], gauss_w[2]; /* 1-D Gauss points / weights on [0,1] */
  double kappa;                  /* constant diffusivity */
  double *values_dev;
  const double *unit_matrix_dev; /* uniform geometry only: the 8x8 element matrix from nh_p1hex_unit_matrix (computed once per
                                    mesh, like the reference hoists it out of the loop); NULL: evaluated inside the call */
  const double *qscale_dev;      /* NULL, or [nelems][8]: coefficient at the Gauss points of every element (element index last
                                    axis fastest, points first coordinate slowest) multiplying kappa -- a scale_dev array of
                                    nh_assemble_matrix (variable or field-dependent diffusivity).  Needs verts_dev. */
  double mass;                   /* constant coefficient of an additional mass term mass * phi_m phi_n (0: stiffness only) */
  const double *qmass_dev;       /* NULL, or [nelems][8] like qscale_dev: mass coefficient at the Gauss points (multiplies `mass`,
                                    which must then be nonzero, e.g. 1) */
  int max_workgroups;            /* 0: one persistent workgroup per CU.  > 0: upper bound -- a multi-GPU caller leaves a few CUs to
                                    the RCCL send/recv kernels that run concurrently (a workgroup here takes a whole CU's LDS,
                                    so nothing else can start on a CU it occupies) */
} nh_p1hex_args;

int nh_p1hex_pattern(const int *shape, int64_t row_begin, int64_t row_end, int64_t *rowptr_dev, int64_t *colidx_dev, void *stream);
int nh_p1hex_laplace(const nh_p1hex_args *args, void *stream);
/* out (+)= K u for the same form, without the matrix: the element matrices are applied to the nodal values u_dev[ndofs] on the fly
 * and reduced row-wise (residual / matrix-free product; replaces the Inflate + numpy.add.at scatter of a vector-valued LoopSum,
 * evaluable.py:3405-3411).  Rows of the planes [plane_begin, plane_end) are written (accumulate = 0) or added to (1); needs
 * verts_dev; values_dev and unit_matrix_dev are ignored. */
int nh_p1hex_apply(const nh_p1hex_args *args, const double *u_dev, double *out_dev, int accumulate, void *stream);
/* element matrix (64 doubles, row major) of the uniform cell scale[0] x scale[1] x scale[2] (verts_dev ignored) */
int nh_p1hex_unit_matrix(const nh_p1hex_args *args, double *ke_dev, void *stream);

/* ---- fast path: structured C0 quadratic hexahedra, constant-coefficient forms, scalar or vector valued ------------------
 * Same result as nh_pattern_* + nh_assemble_matrix for the 3-D 'std' degree-2 basis (27 nodes per element, local order first axis
 * slowest: StructuredBasis function.py:3080-3100, mesh.py:34-60) of a full structured mesh, test = trial basis with ncomp
 * components, constant tensor C (BASELINE.json configs[2]: examples/elasticity.py stiffness scaled to 3-D), but WRITE-ONCE: a
 * workgroup owns K-lines of nodes, recomputes for every touching element only the rows it owns -- as (4 nodes x slots) x
 * (16 nodes) x (quadrature points) products on v_mfma_f64_16x16x4_f64, the form tensor applied to the 3x3 / 4x4 slot blocks
 * afterwards --, reduces in LDS row buffers laid out like the CSR rows and streams finished node planes to HBM.  No global
 * atomics, no zero-fill, no element map; the sorted-unique pattern of this basis is closed form (nh_p2hex_rowptr: scalar row
 * pointer of a node; per axis node X couples to [X-2, X+2] (X even) or [X-1, X+1] (X odd), clipped).  values_dev is laid out as
 * by nh_pattern_expand with all ncomp x ncomp blocks present.
 * owner_begin..owner_end (inclusive): lines of owner cells along axis 0 whose rows are written (owner io holds the node planes
 * 2 io and 2 io + 1; io = shape[0] holds the last plane); layer_begin..layer_end: element layers along axis 0 that contribute
 * (multi-GPU slabs, nutils_amd/partition.py).  Returns NH_ELIMIT when the tables do not fit the LDS (nq too large): use
 * nh_assemble_matrix. */
typedef struct {
  int shape[3];
  int nq;
  const double *weights_dev; /* [nq] */
  nh_geometry geom;
  const double *T_dev;       /* tabulated basis [27][nq][4] (one table: all elements of a 'std' basis share it) */
  int ncomp;                 /* components of the test = trial field, 1..3 */
  const double *C_host;      /* [ncomp][4][ncomp][4] (host memory) */
  double *values_dev;
  const double *scale_dev;   /* optional pointwise factor [nelems][nq], NULL = 1 */
  int layer_begin, layer_end;
  int owner_begin, owner_end;
  int max_workgroups;        /* 0: one persistent workgroup per CU */
  int weights_positive;      /* nonzero: every quadrature weight is > 0 (the caller owns the array and knows): the kernel that forms its operands in registers
                                carries sqrt(w |J|) on both of them; 0: signed weights, the table kernels */
} nh_p2hex_args;

int nh_p2hex_matrix(const nh_p2hex_args *args, void *stream);
int nh_p2hex_rowptr(const int *shape, int64_t node, int64_t *rowptr_out);
/* Meshes of UNIFORM cells (mesh.rectilinear with equidistant vertices: x = offset + scale * (multi-index + xi), mesh.py:45-52) with a constant form and no pointwise
 * factor: all element matrices are equal (the reference evaluates the same numbers for every element), so the rows of a node depend only on its class per axis (first,
 * odd, even, last) and equal the rows of the node of that class in a mesh of 2 x 2 x 2 such cells.  cell_values_dev: the values of nh_p2hex_matrix for shape {2, 2, 2}
 * on those cells (nh_p2hex_rowptr({2,2,2}, 125) * ncomp^2 doubles); this call replicates them into values_dev for the node planes [plane_begin, plane_end) of axis 0
 * (all of them: 0 .. 2 shape[0] + 1): a pure write stream.  Same layout and result as nh_p2hex_matrix on the box geometry of the same cells. */
int nh_p2hex_rows_uniform(const int *shape, int ncomp, const double *cell_values_dev, double *values_dev, int plane_begin, int plane_end, void *stream);
/* closed-form CSR index arrays of the component-expanded pattern (all ncomp x ncomp blocks): rowptr_dev int64[nnodes*ncomp+1],
 * colidx_dev int64[nh_p2hex_rowptr(shape, nnodes) * ncomp^2]; equal to nh_pattern_build + nh_pattern_expand for this basis */
int nh_p2hex_pattern(const int *shape, int ncomp, int64_t *rowptr_dev, int64_t *colidx_dev, void *stream);

/* ---- Monomial: evaluation of factored (pre-integrated) polynomial functionals -------------
 * replaces evaluable.Monomial (evaluable.py:5693-5751; `out = values.copy(); out *= arg[index]`
 * + Inflate/add.at), the per-Newton-step work after evaluable.factor (evaluable.py:5785-5874)
 * has evaluated the sparse Taylor coefficient tensors once with the element loop above.
 * nh_monomial_csr: y[r] += alpha * sum_k values[k] x[colidx[k]] over CSR row r (rank-2 tensor,
 *   deterministic, no atomics).
 * nh_monomial: out[out_index[i]] += alpha values[i] prod_k args[k][indices[k][i]], nargs <= 4;
 *   out_index NULL: scalar result accumulated in out[0].  args_dev / indices_dev are HOST arrays
 *   of device pointers. */
int nh_monomial_csr(int64_t nrows, const int64_t *rowptr_dev, const int64_t *colidx_dev, const double *values_dev, const double *x_dev,
                    double alpha, double *y_dev, void *stream);
int nh_monomial(int64_t n, const double *values_dev, int nargs, const double *const *args_dev, const int64_t *const *indices_dev,
                const int64_t *out_index_dev, double alpha, double *out_dev, void *stream);

/* ---- building the Taylor coefficient tensors of rank 3 and 4 ---------------------------------
 * replaces the one-time element loop + sparse dedup of evaluable.factor (evaluable.py:5785-5874: `zeroed.assparse`, zeros pruned) for value
 * polynomials of ONE scalar field, c int s(x) u^k dV -> C_k[i1..ik] = coeff int s N_i1 .. N_ik dV: every permutation stored, entries sorted by
 * (i1, .., ik) lexicographically, exact zeros dropped -- the arrays evaluable.Monomial holds (evaluable.py:5693-5751) and nh_monomial consumes.
 * nh_factor_tensor builds the tensor and reports its length; nh_factor_fetch copies it into caller-owned buffers (values[nnz], indices[rank][nnz])
 * and releases the library's copy.  One build at a time (the single-stream contract of the scratch buffers). */
typedef struct {
  int64_t nelems;
  const int32_t *elist_dev;  /* optional element subset; scale_dev is indexed by list position */
  int ndims, nq, rank;       /* rank 3 or 4 */
  const double *weights_dev; /* [nq] */
  nh_geometry geom;
  nh_basis basis;            /* uniform nb */
  const double *scale_dev;   /* [nelems][nq] coefficient function at the points, or NULL */
  double coeff;              /* constant factor */
  int64_t ndofs;             /* length of the field's coefficient vector; ndofs^rank must fit 63 bits */
} nh_factor_args;
int nh_factor_tensor(const nh_factor_args *args, int64_t *nnz, void *stream);
int nh_factor_fetch(double *values_dev, int64_t *indices_dev, void *stream);

/* ---- block merge / submatrix: indexed copy of CSR values -----------------------------------
 * dst[dst_index[i]] = src[src_index[i]] (an index array may be NULL: i), plain stores, no accumulation.  The value half of
 * matrix.assemble_block_csr (matrix/__init__.py:103-151: the positions of every block entry in the merged matrix are fixed once
 * per system) and of Matrix.submatrix(free, free) (solver.py:332,386).  `dst` is device memory OR page-locked host memory that is
 * mapped into the device (hipHostMalloc / a pinned torch tensor): a Newton step then sends only the entries of the field-dependent
 * blocks over PCIe, straight into their places of the host CSR value array. */
int nh_index_copy(int64_t n, const double *src_dev, const int64_t *src_index_dev, const int64_t *dst_index_dev, double *dst, void *stream);

/* ---- pointwise coefficient functions ------------------------------------------------------
 * out[i] = sum_t coeffs[t] prod_v x_v[i*strides[v]]^powers[t*nvars+v]  (nvars <= 4, nterms <= 32;
 * x_dev is a HOST array of device pointers).  Evaluates polynomial coefficient functions of field
 * values at the quadrature points (from nh_sample_eval), e.g. psi'(phi), psi''(phi) of
 * examples/cahnhilliard.py:175-176, once per Newton step; the result is a scale_dev array.  The
 * reference represents such terms as rank-3/4 sparse tensors (evaluable.factor); here the
 * integrand is re-integrated with the pointwise coefficient. */
int nh_pointwise_poly(int64_t n, int nvars, const double *const *x_dev, const int *strides, int nterms, const double *coeffs,
                      const int *powers, double *out_dev, void *stream);

/* ---- deterministic vector scatter (owner-side reduction) ------------------------------------------
 * The reference scatters local vectors with numpy.add.at(out, dofs_e, values_e) element by element (Inflate._compile_with_out,
 * evaluable.py:3405-3411; numeric.accumulate, numeric.py:434-460): every dof receives its contributions in ascending (element, local
 * index) order.  Global atomics give the same sum in an arbitrary order, i.e. results that differ in the last bits from run to run.
 * Two-pass form with the reference's order: (1) the element kernels STORE their local vectors (nh_block.local_dev /
 * nh_vector_args.local_dev); (2) nh_scatter_gather sums, for every dof, its contributions through a map built once per (basis, element
 * list): positions of the (e, m) pairs in the local array, grouped by dof, within a dof ascending in (list position, m).
 * nelems / nb / off_dev / dofs_dev: the connectivity as in nh_basis (nb = 0: ragged with off_dev); elist_dev / nlist: the elements of the
 * sample in evaluation order (NULL: all nelems).  Several (plan, local) pairs -- the samples of one residual -- are summed in the order
 * given; accumulate = 0 stores, 1 adds to out_dev[nrows][ncomp]. */
typedef struct nh_scatter_plan nh_scatter_plan;
int nh_scatter_plan_build(int64_t nelems, int64_t nrows, int nb, const int32_t *dofs_dev, const int64_t *off_dev, const int32_t *elist_dev,
                          int64_t nlist, nh_scatter_plan **plan_out, void *stream);
int nh_scatter_plan_free(nh_scatter_plan *plan);
int nh_scatter_gather(int count, const nh_scatter_plan *const *plans, const double *const *locals_dev, int ncomp, double *out_dev,
                      int accumulate, void *stream);

/* ---- per-point forms of field values ------------------------------------------------------------
 * The product-rule coefficients of quasi-linear problems at every quadrature point, from U = (value, gradient w.r.t. x) of
 * the bound scalar field(s) there (nh_sample_eval; U[npoints][S], S = 1 + ndims):
 *   kind 0: out[i]       = sc_i sum_ab B[a][b] Ut[i][a] Ur[i][b]         (point factor of an energy: a scale_dev array)
 *   kind 1: out[i][a][b] = sc_i (b == 0 ? sum_x B[a][x] Ut[i][x] : 0)    (cq_dev of nh_assemble_matrix: g'(u) phi_n B(v, u))
 *   kind 2: out[i][a][b] = sc_i L[a] sum_x B[x][b] Ut[i][x]              (cq_dev: both one-sided tensors of an energy Hessian)
 * sc_i = scale_dev[i] or 1.  B_host [S][S], L_host [S] (kind 2).  Replaces what the reference obtains by differentiating the
 * evaluable graph (evaluable.py: `_derivative` of Multiply / Einsum nodes under function.derivative, function.py:1184-1196)
 * and evaluating the product inside the generated element loop. */
int nh_point_forms(int kind, int64_t npoints, int S, const double *Ut_dev, const double *Ur_dev, const double *B_host, const double *L_host,
                   const double *scale_dev, double *out_dev, void *stream);

/* ---- array-valued expressions of field values at the points of a sample ----------------------------
 * Sample.eval / Sample.bind (sample.py:192-232; _ConcatenatePoints.lower :966-975, _ReorderPoints :978-989) of a function that is a sum of
 * (constant coefficient tensor) x (product of values / gradients of bound fields and of coordinates) x (coefficient function of the point):
 *   out[i][f] (+)= sc_i sum_{t : out_index[t] == f} coef[t] prod_v x_v[i*strides[v] + offsets[t*nvars+v]],   f < nout, nvars <= 6.
 * x_dev: HOST array of nvars device pointers (U arrays of nh_sample_eval: stride ncomp*(1+ndims), offset comp*(1+ndims)+slot; coordinate
 * arrays: stride ndims, offset = axis); strides: HOST array.  out_index / offsets / coef: DEVICE tables of the nentries non-zeros of the
 * coefficient tensor, out_index ascending (an output that is met again later is added to).  scale_dev [npoints] or NULL.
 * accumulate = 0 stores (outputs without an entry become 0), 1 adds to out_dev[npoints][nout].  Replaces the reference's per-element
 * evaluation of the lowered function inside loop_concatenate (evaluable.py:5383-5508). */
int nh_point_expr(int64_t npoints, int nvars, const double *const *x_dev, const int64_t *strides, int64_t nentries, const int32_t *out_index_dev,
                  const int32_t *offsets_dev, const double *coef_dev, const double *scale_dev, int nout, double *out_dev, int accumulate, void *stream);

/* ---- rational bases (NURBS) ------------------------------------------------------------------
 * In-place transform of per-element tabulated functions T[(e, i)][q][S] (function (e,i) = e*nb+i, or off[e]+i for ragged bases)
 * into N_i = w_i B_i / W,  dN_i = w_i (dB_i W - B_i dW) / W^2  with the dof weights w[dofs[e][i]] and the weight function W
 * either tabulated (W_dev [nelems][nq], dW_dev [nelems][nq][ndims], derivatives w.r.t. the element coordinates) or, if
 * W_dev is NULL, W = sum_j w_j B_j of the element itself.  Replaces the symbolic quotient
 * `bsplinebasis * controlweights / weightfunc` (examples/platewithhole.py:71-72,85) evaluated inside the generated loop. */
int nh_rationalize(double *T_dev, int64_t nelems, int nb, const int64_t *off_dev, const int32_t *dofs_dev, const double *weights_dev,
                   const double *W_dev, const double *dW_dev, int nq, int ndims, void *stream);

#ifdef __cplusplus
}
#endif
#endif
