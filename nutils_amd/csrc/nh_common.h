// Shared helpers for libnutils_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/nutils_hip.h"

typedef int64_t i64;

void nh_set_error(const char *fmt, ...);

#define NH_CHECK_HIP(expr)                                                                      \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      nh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
      return NH_EHIP;                                                                           \
    }                                                                                           \
  } while (0)

#define NH_REQUIRE(cond, ...)   \
  do {                          \
    if (!(cond)) {              \
      nh_set_error(__VA_ARGS__);\
      return NH_EINVAL;         \
    }                           \
  } while (0)

#define NH_LAUNCH_CHECK() NH_CHECK_HIP(hipGetLastError())

static inline hipStream_t nh_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// exclusive scan of n int32 counts into int64 offsets (out has n+1 entries; out[n] = total)
int nh_scan_exclusive(const int32_t *in_dev, i64 *out_dev, i64 n, hipStream_t stream);
int nh_scan_exclusive64(const i64 *in_dev, i64 *out_dev, i64 n, hipStream_t stream);

// geometry / basis structs are passed by value to kernels
struct GeomK {
  int kind, ngb;
  const double *gT;
  const int32_t *gdofs;
  const double *verts;
  const double *origin;
  const double *size;
  const double *jac;
  const double *x;
  int bnd_axis;
  int nograd;  // no term of the launch reads a gradient slot (mass matrices, load vectors): the inverse Jacobian is not needed and is set to zero
};

struct BasisK {
  int nb;
  const double *T;
  const int32_t *dofs;
  const i64 *off;
  const int32_t *tab;
};

static inline GeomK to_k(const nh_geometry &g) { return GeomK{g.kind, g.ngb, g.gT_dev, g.gdofs_dev, g.verts_dev, g.origin_dev, g.size_dev, g.jac_dev, g.x_dev, g.bnd_axis, 0}; }
// Does a coefficient tensor C[n0][S][n1][S] (or a source f[n0][S]: n1 = 0) have an entry in a gradient slot?  Terms without any never see the
// inverse Jacobian in the reference (a mass integrand has no gradient node): an exactly singular element, whose inverse is NaN (numeric.py:221-241),
// must leave them finite -- and 0 * NaN would not.
static inline bool uses_gradients(const double *A, int n0, int S, int n1) {
  if (!A) return false;
  if (n1 == 0) {
    for (int c = 0; c < n0; ++c)
      for (int a = 1; a < S; ++a)
        if (A[c * S + a] != 0.) return true;
    return false;
  }
  for (int c = 0; c < n0; ++c)
    for (int a = 0; a < S; ++a)
      for (int d = 0; d < n1; ++d)
        for (int b = 0; b < S; ++b)
          if ((a || b) && A[((c * S + a) * n1 + d) * S + b] != 0.) return true;
  return false;
}

static inline BasisK to_k(const nh_basis &b) { return BasisK{b.nb, b.T_dev, b.dofs_dev, b.off_dev, b.tab_dev}; }

// sparsity pattern handle (nh_pattern.hip owns it; the element kernels read the size classes of ragged bases from it)
constexpr int NH_MAX_BUCKETS = 9;
// owner blocks of NH_MATRIX_FUSED (nh_gather.hip): rows clustered into blocks whose CSR rows fit the LDS of a workgroup; a block recomputes every element
// that touches one of its rows and writes its rows once
struct nh_fused_plan {
  int nblocks, rows_per_block;  // rows_per_block: of the largest block (Morton boxes of at most that many rows)
  int max_ents;       // doubles of the fullest block's accumulator (= CSR entries of its rows)
  int nturns;         // visitors of the busiest row: rounds of the ordered adds
  i64 nvisits;
  int32_t *order;     // [nrows]: dof at rank position i
  i64 *bptr;          // [nblocks + 1]: first rank position of block b
  i64 *rstart;        // [nrows], by rank position: first entry of the row in the value array
  i64 *epos;          // [nrows + 1], by rank position: CSR entries of the rows in front (offset of the row in its block's accumulator: epos[i] - epos[bptr[b]])
  i64 *vptr;          // [nblocks + 1]: visits of block b
  int32_t *vlist;     // [nvisits]: element
  uint16_t *vrow;     // [nvisits][nbt]: row of local function m within the block | turn << 9, 0xffff: the row belongs to another block
  uint8_t *cpos;      // [nelems][nbt * nbr]: position of entry (m, n) within its CSR row
  // trilinear hexahedra at the 2 x 2 x 2 Gauss points (recognised from the tables of a launch): -1 not looked at, 0 no, 1 yes, 2 yes with a mass term
  int p1hex;
  const void *p1hex_key[4];  // table pointers and ...
  double p1hex_C[16];        // ... form the answer belongs to
  double p1hex_tab[32];      // tables of the sum-factorised routine (P1Tab of nh_gather.hip)
};

// A pointer the device code LOADS from memory (a parameter block in device memory, not the kernel argument segment) has no known address space: every access through it
// is a FLAT instruction.  Where such a pointer is hot, the access goes through a pointer cast to the global address space (nh_g: read, nh_gw: write).
#ifdef __HIPCC__
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T *nh_g(const T *p) {
  return (const __attribute__((address_space(1))) T *)p;
}
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T *nh_gw(T *p) {
  return (__attribute__((address_space(1))) T *)p;
}
#endif


// row tasks of the owner kernel for vector-valued blocks (nh_owner.hip): the contributions (visit, m, n) to the scalar entries of a block's rows as lane items, sorted by
// (row, position in the row, element) and packed into chunks of 64 lanes such that the items of one entry never straddle a chunk
struct nh_owner_plan {
  int nblocks, rows_per_block, max_visits, nsteps;  // rows_per_block: at most; nsteps: doubling steps of the segmented sum (2^nsteps >= contributions of the fullest entry)
  int qc;           // points per chunk of the D tables (= all points unless the elements have more than 9 functions)
  int rows16;       // no entry straddles a row of 16 lanes (entries of at most 16 items): the sums run on DPP row shifts
  i64 nvisits, nchunks;
  int32_t *order;   // [nrows]: dof at rank position i
  i64 *bptr;        // [nblocks + 1]: first rank position of block b (Morton boxes of at most rows_per_block rows)
  i64 *vptr;        // [nblocks + 1]
  int32_t *vlist;   // [nvisits]: element, ascending within a block
  i64 *cptr;        // [nblocks + 1]: chunks of block b
  uint32_t *isrc;   // [nchunks * 64]: bit 31 valid, bit 30 first item of its entry, bits 10-21 visit within the block, 5-9 m, 0-4 n
  uint32_t *idst;   // [nchunks * 64]: row within the block << 16 | position of the entry in its scalar row
  // the staging chain of a block cut short (a level of dependent loads less for its rows and for its vertices):
  i64 *prs;         // [nrows]: first entry of the scalar row at rank position i
  int32_t *prl;     // [nrows]: its length
  int32_t *vvert;   // [nvisits * 2^ndims] or null: vertex numbers of the visiting elements, made from the connectivity `vvert_src` (remade when a call brings another array)
  const void *vvert_src;
};
void nh_owner_free(nh_owner_plan *o);

struct nh_pattern {
  i64 nelems, nrows, ncols, nnz;
  int nbt, nbr;
  i64 *srowptr;     // [nrows+1]
  int32_t *scol;    // [nnz]
  int32_t *emap;    // [sum_e nbt_e*nbr_e]
  i64 *eoff;        // ragged only: [nelems+1] prefix sums of nbt_e*nbr_e
  i64 emap_len;
  // ragged only: elements grouped by functions per element
  int nbuckets;
  i64 bucket_n[NH_MAX_BUCKETS];
  int32_t *bucket_elist[NH_MAX_BUCKETS];  // views into bucket_store
  int bucket_nbt[NH_MAX_BUCKETS], bucket_nbr[NH_MAX_BUCKETS];
  int32_t *bucket_store;
  // gather map (nh_gather.hip; built on the first NH_MATRIX_GATHER assembly): for scalar entry k the local-matrix positions
  // gsrc[gptr[k] .. gptr[k+1]) in ascending order (element, m, n) -- the order numpy.add.at accumulates them in -- and the row of k
  int32_t *gsrc;
  int32_t *gsrc_sym;  // the same map for producers that write the node pairs m >= n only: sources of (m < n) entries point to the (n, m) block, bit 31 set = transpose it
  // triangular scratch (symmetric blocks on a symmetric pattern, nh_gram_sym.inc): an element writes the node pairs (m', n'), m' >= n', of its nodes SORTED BY DOF, packed
  // row by row -- the LOWER triangle of the global matrix is gathered from them, the upper one is its mirror.  tri_rank [sum nb_e]: position of a node among the dofs of its
  // element; tri_base [nelems + 1]: first packed pair of an element; gsrc_tri: the gather map over packed positions (sources of upper entries: 0xffffffff);
  // gmirror [nnz]: for an entry above the diagonal the entry it mirrors, else -1
  unsigned char *tri_rank;
  i64 *tri_base;
  int32_t *gsrc_tri;
  int32_t *gmirror;
  int tri_failed;
  unsigned *gptr;
  int32_t *grow;
  nh_fused_plan *fused;  // NH_MATRIX_FUSED: built on the first such assembly
  int fused_failed;      // the plan cannot be built for this pattern: do not try again
  nh_owner_plan *owner;  // NH_MATRIX_FUSED, vector-valued blocks: built on the first such assembly
  int owner_failed;
};

// component-block layout of an expanded pattern (mirrors FormK of nh_assemble_generic.hip)
struct GSlots {
  int nct, ncr, tot;
  int cnt[4], cum[4];
  signed char dpos[4][4];
  unsigned char mask[4][4];
};

// nh_gather.hip: deterministic two-pass scatter (element-major local matrices, then one sum per CSR entry)
int nh_gather_prepare(nh_pattern *p, const nh_basis &test, const int32_t *elist, hipStream_t s);  // elist: element ids of the pattern's elements, or NULL
int nh_gather_scratch(size_t doubles, double **out);
int nh_local_vector(const nh_matrix_args *a, double *local, bool *done, hipStream_t s);
int nh_gather_values(const nh_pattern *p, const double *local, i64 ld, const GSlots &slots, double *values, int store, hipStream_t s, int sym_sources = 0);  // 1: gsrc_sym, 2: triangular scratch
int nh_gather_prepare_tri(nh_pattern *p, const nh_basis &test, hipStream_t s);
int nh_gather_prepare_sym(nh_pattern *p, const nh_basis &test, const int32_t *elist, hipStream_t s);
int nh_local_scalar(const nh_matrix_args *a, double *local, bool *done, hipStream_t s);
// owner-block assembly (NH_MATRIX_FUSED); *done = false: not applicable to this launch, nothing was written
int nh_fused_scalar(const nh_matrix_args *a, bool *done, hipStream_t s);
void nh_fused_free(nh_fused_plan *f);
// nh_owner.hip: the same for vector-valued blocks on small uniform bases (row tasks: Gram sums per scalar entry from D tables in LDS)
int nh_owner_vector(const nh_matrix_args *a, const GSlots &slots, bool *done, hipStream_t s);
// nh_assemble_p1hex.hip: exchange scratch of the exact-tile kernel
int nh_p1hex_tiles_release(void);
