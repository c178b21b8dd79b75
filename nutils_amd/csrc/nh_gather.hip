// Deterministic owner-side reduction for unstructured / ragged connectivity (NH_MATRIX_GATHER, include/nutils_hip.h).
// The generic element kernels scatter their local matrices with one f64 atomic per entry (134 M for the 128^3 trilinear mesh); here
//   pass 1  writes the local matrices to a scratch array, element by element (coalesced, no atomics), and
//   pass 2  sums, for every CSR entry ONCE, its contributions through a gather map in ascending (element, m, n) order --
// the order in which the reference accumulates them (numpy.add.at / numeric.accumulate over the flattened element loop,
// numeric.py:434-460; evaluable.py:603-605), so the result does not depend on the scheduling of the element kernel.
// The map is the stable sort of the element map by CSR entry; it is built once per pattern handle.
#include "nh_common.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

#include "nh_geom.inc"

__device__ __forceinline__ i64 boff(const BasisK &b, i64 e) { return b.off ? b.off[e] : e * (i64)b.nb; }
__device__ __forceinline__ int bnb(const BasisK &b, i64 e) { return b.off ? (int)(b.off[e + 1] - b.off[e]) : b.nb; }
__device__ __forceinline__ i64 bfn(const BasisK &b, i64 e) { return b.off ? b.off[e] : (b.tab ? (i64)b.tab[e] * b.nb : 0); }

// key of local position i = (element, m, n): its scalar CSR entry
__global__ void k_gather_keys(i64 nelems, const int32_t *elist, BasisK test, int nbr_uniform, const i64 *eoff, const i64 *srowptr, const int32_t *emap,
                              unsigned *keys, unsigned *vals, int *counts) {
  for (i64 e = blockIdx.x; e < nelems; e += gridDim.x) {
    const i64 eid = elist ? elist[e] : e;  // the pattern of an element list numbers its elements by list position, the basis by element
    const int nbt = bnb(test, eid);
    if (!nbt) continue;
    const i64 t0 = boff(test, eid);
    const i64 e0 = eoff ? eoff[e] : e * (i64)nbt * nbr_uniform;
    const i64 cnt = (eoff ? eoff[e + 1] : e0 + (i64)nbt * nbr_uniform) - e0;
    const int nbr = (int)(cnt / nbt);
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
      const int m = i / nbr;
      const i64 k = srowptr[test.dofs[t0 + m]] + emap[e0 + i];
      keys[e0 + i] = (unsigned)k;
      vals[e0 + i] = (unsigned)(e0 + i);
      atomicAdd(counts + k, 1);
    }
  }
}

__global__ void k_narrow(i64 n, const i64 *in, unsigned *out) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) out[i] = (unsigned)in[i];
}

// symmetric producers (the node pairs m >= n only): canonical source of every local position -- (m, n) itself, or the (n, m) block with bit 31 = transpose
__global__ void k_gather_canon(i64 nelems, const int32_t *elist, BasisK test, int nbr_uniform, const i64 *eoff, unsigned *canon) {
  for (i64 e = blockIdx.x; e < nelems; e += gridDim.x) {
    const i64 eid = elist ? elist[e] : e;
    const int nbt = bnb(test, eid);
    if (!nbt) continue;
    const i64 e0 = eoff ? eoff[e] : e * (i64)nbt * nbr_uniform;
    const i64 cnt = (eoff ? eoff[e + 1] : e0 + (i64)nbt * nbr_uniform) - e0;
    const int nbr = (int)(cnt / nbt);
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
      const int m = i / nbr, n = i - m * nbr;
      canon[e0 + i] = m >= n ? (unsigned)(e0 + i) : ((unsigned)(e0 + (i64)n * nbr + m) | 0x80000000u);
    }
  }
}
__global__ void k_gather_remap(i64 n, const int32_t *gsrc, const unsigned *canon, int32_t *out) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) out[i] = (int32_t)canon[(unsigned)gsrc[i]];
}

__global__ void k_rowof(i64 nrows, const i64 *srowptr, int32_t *grow) {
  for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (i64)gridDim.x * blockDim.x)
    for (i64 k = srowptr[r]; k < srowptr[r + 1]; ++k) grow[k] = (int32_t)r;
}

// pass 2: one thread per scalar entry; local: element-major [position][nct * ncr]
__global__ void k_gather_values(i64 nnz, const unsigned *gptr, const int32_t *gsrc, const int32_t *grow, const i64 *srowptr, const double *local, i64 ld, int per,
                                GSlots gs, double *values, int store) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const unsigned b = gptr[k], e = gptr[k + 1];
  // (scalar blocks: the expanded pattern is the scalar one; vector-valued blocks have their own kernel below -- its 48 accumulator / staging
  // registers would take the occupancy these latency-bound 8-byte gathers live on: 0.58 -> 0.75 ms on the 128^3 trilinear mesh)
  {
    // eight contributions in flight per thread (index loads, then value loads, then the sum in the order of the map): one at a time, the
    // kernel waits out two dependent memory latencies per contribution (0.76 -> 0.58 ms on the 128^3 trilinear mesh)
    double s = 0;
    for (unsigned i0 = b; i0 < e; i0 += 8) {
      unsigned idx[8];
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) idx[u] = i0 + u < e ? (unsigned)gsrc[i0 + u] : 0xffffffffu;
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = idx[u] != 0xffffffffu ? local[idx[u]] : 0.;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < e) s += v[u];
    }
    values[k] = store ? s : values[k] + s;
  }
}

template <bool SYMSRC>  // sources with bit 31 set are the mirrored block: read transposed (nct == ncr)
__global__ void k_gather_values_v(i64 nnz, const unsigned *gptr, const int32_t *gsrc, const int32_t *grow, const i64 *srowptr, const double *local, GSlots gs,
                                  double *values, int store) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const unsigned b = gptr[k], e = gptr[k + 1];
  const int ncd = gs.nct * gs.ncr;
  // vector-valued: the nct x ncr block of a contribution is contiguous in the scratch -- read once, all components summed in one pass over the
  // sources (one pass per component re-reads the index list and touches every 72-byte block nine times: 8.0 -> 2.85 ms on 96^3 trilinear elasticity; 16-byte loads: 2.0 ms)
  const i64 r = grow[k], a0 = srowptr[r], len = srowptr[r + 1] - a0, pos = k - a0;
  double sum[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) sum[j] = 0;
  for (unsigned i0 = b; i0 < e; i0 += 2) {
    const unsigned r1 = (unsigned)gsrc[i0], r2 = i0 + 1 < e ? (unsigned)gsrc[i0 + 1] : 0u;
    const bool t1 = SYMSRC && (r1 >> 31), t2 = SYMSRC && (r2 >> 31);
    const i64 i1 = (i64)(SYMSRC ? r1 & 0x7fffffffu : r1), i2 = i0 + 1 < e ? (i64)(SYMSRC ? r2 & 0x7fffffffu : r2) : -1;
    const double *s1 = local + i1 * ncd, *s2 = local + (i2 >= 0 ? i2 : i1) * ncd;
    double v1[16], v2[16];
    if (ncd == 9) {  // 3 x 3 blocks (72 bytes, 8-byte aligned): four 16-byte loads + one instead of nine 8-byte loads per lane
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        double2 a, c;
        __builtin_memcpy(&a, s1 + j, 16);
        __builtin_memcpy(&c, s2 + j, 16);
        v1[j] = a.x, v1[j + 1] = a.y, v2[j] = c.x, v2[j + 1] = c.y;
      }
      v1[8] = s1[8], v2[8] = s2[8];
#pragma unroll
      for (int j = 9; j < 16; ++j) v1[j] = v2[j] = 0.;
      if (i2 < 0) {
#pragma unroll
        for (int j = 0; j < 9; ++j) v2[j] = 0.;
      }
    } else if (ncd == 4) {  // 2 x 2 blocks: two 16-byte loads
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        double2 a, c;
        __builtin_memcpy(&a, s1 + j, 16);
        __builtin_memcpy(&c, s2 + j, 16);
        v1[j] = a.x, v1[j + 1] = a.y, v2[j] = i2 >= 0 ? c.x : 0., v2[j + 1] = i2 >= 0 ? c.y : 0.;
      }
#pragma unroll
      for (int j = 4; j < 16; ++j) v1[j] = v2[j] = 0.;
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        v1[j] = j < ncd ? s1[j] : 0.;
        v2[j] = j < ncd && i2 >= 0 ? s2[j] : 0.;
      }
    }
    if constexpr (SYMSRC) {  // (3 x 3 and 2 x 2 blocks: the transposition is a fixed permutation of the registers)
      auto transpose = [&](double (&v)[16]) {
        if (ncd == 9) {
          double t;
          t = v[1], v[1] = v[3], v[3] = t;
          t = v[2], v[2] = v[6], v[6] = t;
          t = v[5], v[5] = v[7], v[7] = t;
        } else if (ncd == 4) {
          const double t = v[1];
          v[1] = v[2], v[2] = t;
        }
      };
      if (t1) transpose(v1);
      if (t2) transpose(v2);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) sum[j] += v1[j];
    if (i2 >= 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) sum[j] += v2[j];
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (c >= gs.nct || d >= gs.ncr || !gs.mask[c][d]) continue;
      const int j = c * gs.ncr + d;
      double v = 0;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) v = jj == j ? sum[jj] : v;
      double *dst = values + a0 * gs.tot + len * gs.cum[c] + pos * gs.cnt[c] + gs.dpos[c][d];
      *dst = store ? v : *dst + v;
    }
}

// 2 x 2 blocks, all four component blocks present (plane elasticity): the general kernel above with its run-time component count keeps 48 registers of accumulators and
// staging and two contributions in flight; here four accumulators, FOUR contributions in flight (index loads, then 32-byte value loads, then the sums in the order of the
// map) and two 16-byte stores.  The same sums in the same order: bit-identical.
__global__ void k_gather_values_2x2(i64 nnz, const unsigned *gptr, const int32_t *gsrc, const int32_t *grow, const i64 *srowptr, const double *local, GSlots gs, double *values,
                                    int store) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const unsigned b = gptr[k], e = gptr[k + 1];
  const i64 r = grow[k], a0 = srowptr[r], len = srowptr[r + 1] - a0, pos = k - a0;
  double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
  for (unsigned i0 = b; i0 < e; i0 += 4) {
    unsigned idx[4];
    double2 lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) idx[u] = i0 + u < e ? (unsigned)gsrc[i0 + u] : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double2 *src = reinterpret_cast<const double2 *>(local + (i64)(idx[u] != 0xffffffffu ? idx[u] : 0u) * 4);
      lo[u] = src[0], hi[u] = src[1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u < e) s00 += lo[u].x, s01 += lo[u].y, s10 += hi[u].x, s11 += hi[u].y;
  }
  // rows (node, c) of the expanded pattern: the entries (n, d = 0, 1) of a row are adjacent
  double *d0 = values + a0 * gs.tot + len * gs.cum[0] + pos * gs.cnt[0] + gs.dpos[0][0];
  double *d1 = values + a0 * gs.tot + len * gs.cum[1] + pos * gs.cnt[1] + gs.dpos[1][0];
  if (store) {
    d0[0] = s00, d0[1] = s01, d1[0] = s10, d1[1] = s11;
  } else {
    d0[0] += s00, d0[1] += s01, d1[0] += s10, d1[1] += s11;
  }
}

// ---- triangular scratch: symmetric two-component blocks on a symmetric pattern (the producer is k_gram_sym in its packed mode) ---------------------------------------
// The local matrix of a symmetric form is symmetric and so is the global one: an element writes only the node pairs (m', n'), m' >= n', of its nodes sorted by dof,
// packed row by row (half the scratch, and rows of a node stay contiguous: the reads of neighbouring entries stay neighbours); entries on and below the diagonal are
// gathered from them in the order of the map (ascending element: the reference's order), entries above are the transposed copies of their mirrors -- the sum of (c, r) is
// the sum of (r, c) term by term, so the copy is what a second gather would give.
__global__ void k_tri_rank(i64 nelems, BasisK test, unsigned char *rank, int *cnt) {
  for (i64 e = blockIdx.x; e < nelems; e += gridDim.x) {
    const int nb = test.off ? (int)(test.off[e + 1] - test.off[e]) : test.nb;
    const i64 t0 = test.off ? test.off[e] : e * (i64)test.nb;
    for (int m = threadIdx.x; m < nb; m += blockDim.x) {
      const int dm = test.dofs[t0 + m];
      int r = 0;
      for (int j = 0; j < nb; ++j) r += test.dofs[t0 + j] < dm;
      rank[t0 + m] = (unsigned char)r;
    }
    if (threadIdx.x == 0) cnt[e] = nb * (nb + 1) / 2;
  }
}
// canonical packed position of every local position (e, m, n): the pair of the ranks if rank(m) >= rank(n), else none
__global__ void k_tri_canon(i64 nelems, BasisK test, int nbr_uniform, const i64 *eoff, const unsigned char *rank, const i64 *tbase, unsigned *canon) {
  for (i64 e = blockIdx.x; e < nelems; e += gridDim.x) {
    const int nb = test.off ? (int)(test.off[e + 1] - test.off[e]) : test.nb;
    const i64 t0 = test.off ? test.off[e] : e * (i64)test.nb;
    const i64 e0 = eoff ? eoff[e] : e * (i64)nb * nbr_uniform;
    for (int i = threadIdx.x; i < nb * nb; i += blockDim.x) {
      const int m = i / nb, n = i - m * nb;
      const int rm = rank[t0 + m], rn = rank[t0 + n];
      canon[e0 + i] = rm >= rn ? (unsigned)(tbase[e] + rm * (rm + 1) / 2 + rn) : 0xffffffffu;
    }
  }
}
__global__ void k_tri_mirror(i64 nnz, const int32_t *grow, const i64 *srowptr, const int32_t *scol, int32_t *mirror) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const int r = grow[k], c = scol[k];
  int out = -1;
  if (c > r) {  // (r, c) above the diagonal: find (c, r)
    i64 lo = srowptr[c], hi = srowptr[c + 1];
    while (lo < hi) {
      const i64 mid = (lo + hi) >> 1;
      if (scol[mid] < r) lo = mid + 1;
      else hi = mid;
    }
    out = (lo < srowptr[c + 1] && scol[lo] == r) ? (int32_t)lo : -2;  // -2: the pattern is not symmetric (the plan is refused)
  }
  mirror[k] = out;
}
__global__ void k_tri_check(i64 nnz, const int32_t *mirror, int *bad) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nnz && mirror[k] == -2) *bad = 1;
}

__global__ void k_gather_values_2x2_tri(i64 nnz, const unsigned *gptr, const int32_t *gsrc, const int32_t *grow, const i64 *srowptr, const int32_t *mirror, const double *local,
                                        GSlots gs, double *values) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz || mirror[k] >= 0) return;
  const unsigned b = gptr[k], e = gptr[k + 1];
  const i64 r = grow[k], a0 = srowptr[r], len = srowptr[r + 1] - a0, pos = k - a0;
  double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
  for (unsigned i0 = b; i0 < e; i0 += 4) {
    unsigned idx[4];
    double2 lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) idx[u] = i0 + u < e ? (unsigned)gsrc[i0 + u] : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double2 *src = reinterpret_cast<const double2 *>(local + (i64)(idx[u] != 0xffffffffu ? idx[u] : 0u) * 4);
      lo[u] = src[0], hi[u] = src[1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u < e) s00 += lo[u].x, s01 += lo[u].y, s10 += hi[u].x, s11 += hi[u].y;
  }
  double *d0 = values + a0 * gs.tot + len * gs.cum[0] + pos * gs.cnt[0] + gs.dpos[0][0];
  double *d1 = values + a0 * gs.tot + len * gs.cum[1] + pos * gs.cnt[1] + gs.dpos[1][0];
  d0[0] = s00, d0[1] = s01, d1[0] = s10, d1[1] = s11;
}
// scalar blocks on the triangular scratch: eight contributions in flight, like k_gather_values
__global__ void k_gather_values_tri1(i64 nnz, const unsigned *gptr, const int32_t *gsrc, const int32_t *mirror, const double *local, double *values) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz || mirror[k] >= 0) return;
  const unsigned b = gptr[k], e = gptr[k + 1];
  double s = 0;
  for (unsigned i0 = b; i0 < e; i0 += 8) {
    unsigned idx[8];
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) idx[u] = i0 + u < e ? (unsigned)gsrc[i0 + u] : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = idx[u] != 0xffffffffu ? local[idx[u]] : 0.;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < e) s += v[u];
  }
  values[k] = s;
}
__global__ void k_mirror_1(i64 nnz, const int32_t *mirror, double *values) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nnz && mirror[k] >= 0) values[k] = values[mirror[k]];
}
__global__ void k_mirror_2x2(i64 nnz, const int32_t *grow, const i64 *srowptr, const int32_t *mirror, GSlots gs, double *values) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const i64 km = mirror[k];
  if (km < 0) return;
  const i64 rs = grow[km], as = srowptr[rs], ls = srowptr[rs + 1] - as, ps = km - as;
  const double *s0 = values + as * gs.tot + ls * gs.cum[0] + ps * gs.cnt[0] + gs.dpos[0][0];
  const double *s1 = values + as * gs.tot + ls * gs.cum[1] + ps * gs.cnt[1] + gs.dpos[1][0];
  const i64 r = grow[k], a0 = srowptr[r], len = srowptr[r + 1] - a0, pos = k - a0;
  double *d0 = values + a0 * gs.tot + len * gs.cum[0] + pos * gs.cnt[0] + gs.dpos[0][0];
  double *d1 = values + a0 * gs.tot + len * gs.cum[1] + pos * gs.cnt[1] + gs.dpos[1][0];
  d0[0] = s0[0], d0[1] = s1[0], d1[0] = s0[1], d1[1] = s1[1];
}

// ---- pass 1 for small scalar elements: ONE THREAD per element, the local matrix in registers --------------------------------------------
// (the one-wave-per-element kernel spends ~500 wave instructions on a trilinear element: lanes idle in the pointwise stages, LDS staging,
// barriers; a thread that keeps the NBT x NBR sums in registers needs ~90 per element and no LDS at all)
#ifdef NH_ABLATION
#define MDBGL(p) ((p).debug)
#else
#define MDBGL(p) 0
#endif
struct LocK {
  i64 nelems;
  const int32_t *elist;
  int nq;
  const double *weights;
  GeomK geom;
  BasisK test, trial;
  const double *scale;
  int by_elem;
  double C[16];  // [a][b]
  double *local;
  int debug;  // ablation builds: 1 = no stores, 2 = no vertex gather
  int ldst_doubles;  // staged tables in front of the per-wave transposition buffers
};

// local matrix of element e (list position ie) of a scalar form on small uniform bases, one thread: A[m][n] (SYMD: the upper triangle n >= m only)
template <int ND, int NBT, int NBR, bool LDST, bool SYMD>
__device__ __forceinline__ void local_scalar_matrix(const LocK &p, const double *sT, i64 e, i64 ie, double (&A)[NBT][NBR]) {
  constexpr int S = 1 + ND, NG = 1 << ND;
#pragma unroll
  for (int m = 0; m < NBT; ++m)
#pragma unroll
    for (int n = 0; n < NBR; ++n) A[m][n] = 0;
  // multilinear isoparametric geometry: the vertices of the element once, not once per point
  const bool iso = p.geom.kind == NH_GEOM_ISO && p.geom.ngb == NG;
  double X[NG][ND];
  if (iso) {
#pragma unroll
    for (int a = 0; a < NG; ++a) {
#ifdef NH_ABLATION
      if (p.debug & 2) {
#pragma unroll
        for (int i = 0; i < ND; ++i) X[a][i] = ((a >> (ND - 1 - i)) & 1) + 1e-3 * (double)((ie + i) & 7);
        continue;
      }
#endif
      const i64 v = p.geom.gdofs[e * NG + a];
#pragma unroll
      for (int i = 0; i < ND; ++i) X[a][i] = p.geom.verts[v * ND + i];
    }
  }
  const double *Tt = LDST ? sT : p.test.T + bfn(p.test, e) * p.nq * S, *Tr = LDST ? sT + NBT * p.nq * S : p.trial.T + bfn(p.trial, e) * p.nq * S;
  const double *gT = LDST ? sT + (NBT + NBR) * p.nq * S : p.geom.gT;
  for (int q = 0; q < p.nq; ++q) {
    double Ji[ND][ND], det;
    if (iso) {
      double J[ND][ND];
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int j = 0; j < ND; ++j) J[i][j] = 0;
#pragma unroll
      for (int a = 0; a < NG; ++a) {
        const double *t = gT + ((i64)a * p.nq + q) * S;
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
          for (int j = 0; j < ND; ++j) J[i][j] += X[a][i] * t[1 + j];
      }
      invert<ND>(J, Ji, det);
      if (p.geom.bnd_axis >= 0) {
        double s2 = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
          for (int i = 0; i < ND; ++i)
            if (j == p.geom.bnd_axis) s2 += Ji[j][i] * Ji[j][i];
        det *= sqrt(s2);
      }
    } else
      geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
    const double w = p.weights[q] * fabs(det) * (p.scale ? p.scale[(p.by_elem ? e : ie) * p.nq + q] : 1.);
    // trial side premultiplied by the form and the weight: W[n][a] = w sum_b C[a][b] Dr[n][b]
    double W[NBR][S], D[NBR][S], wc[S];
#pragma unroll
    for (int a = 0; a < S; ++a) wc[a] = w * p.C[a * S + a];
#pragma unroll
    for (int n = 0; n < NBR; ++n) {
      const double *T = Tr + ((size_t)n * p.nq + q) * S;
      D[n][0] = T[0];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        double s = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j) s += T[1 + j] * Ji[j][i];
        D[n][1 + i] = s;
      }
#pragma unroll
      for (int a = 0; a < S; ++a) {
        if (SYMD)
          W[n][a] = wc[a] * D[n][a];
        else {
          double s = 0;
#pragma unroll
          for (int b = 0; b < S; ++b) s += p.C[a * S + b] * D[n][b];
          W[n][a] = w * s;
        }
      }
    }
    if (SYMD) {
      // (operator slots the form does not use -- the value slot of a Laplace form, the gradient slots of a mass form -- are skipped: uniform branches)
#pragma unroll
      for (int a = 0; a < S; ++a)
        if (p.C[a * S + a] != 0.) {
#pragma unroll
          for (int m = 0; m < NBT; ++m)
#pragma unroll
            for (int n = m; n < NBR; ++n) A[m][n] += D[m][a] * W[n][a];
        }
    } else {
#pragma unroll
      for (int m = 0; m < NBT; ++m) {
        const double *T = Tt + ((size_t)m * p.nq + q) * S;
        double dt[S];
        dt[0] = T[0];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          double s = 0;
#pragma unroll
          for (int j = 0; j < ND; ++j) s += T[1 + j] * Ji[j][i];
          dt[1 + i] = s;
        }
#pragma unroll
        for (int n = 0; n < NBR; ++n)
#pragma unroll
          for (int a = 0; a < S; ++a) A[m][n] += dt[a] * W[n][a];
      }
    }
  }
}

template <int ND, int NBT, int NBR, bool LDST, bool SYMD>
__global__ __launch_bounds__(128) void k_local_scalar(LocK p) {
  // SYMD: test == trial basis and a DIAGONAL form tensor (Laplace, mass, reaction-diffusion): the local matrix is symmetric -- the upper triangle is
  // accumulated and mirrored in the store -- and the trial side costs S multiplies instead of S * S multiply-adds per function
  constexpr int S = 1 + ND, NG = 1 << ND;
  // LDST: one table for all elements (no tab / off array) -- test, trial and geometry tables are staged in LDS once per workgroup; read from
  // global memory they are L1 hits, but 96 texture-path loads per point and thread with a wait in front of their first use
  extern __shared__ __attribute__((aligned(16))) double sT[];
  if (LDST) {
    const int nt = NBT * p.nq * S, nr = NBR * p.nq * S, ng = NG * p.nq * S;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) sT[i] = p.test.T[i];
    for (int i = threadIdx.x; i < nr; i += blockDim.x) sT[nt + i] = p.trial.T[i];
    if (p.geom.kind == NH_GEOM_ISO && p.geom.ngb == NG)
      for (int i = threadIdx.x; i < ng; i += blockDim.x) sT[nt + nr + i] = p.geom.gT[i];
    __syncthreads();
  }
  const i64 ie0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 ie = min(ie0, p.nelems - 1);  // (lanes behind the last element recompute it: the store below needs whole waves, and skips them)
  const i64 e = p.elist ? p.elist[ie] : ie;
  double A[NBT][NBR];
  local_scalar_matrix<ND, NBT, NBR, LDST, SYMD>(p, sT, e, ie, A);
  // Store, element-major (the gather of a CSR row reads whole rows of the local matrices).  A thread's matrix is NBT * NBR * 8 contiguous bytes,
  // neighbouring lanes are that far apart: stored straight from the registers, every instruction touches 64 lines (0.23 of the 0.61 ms of this
  // kernel on the 128^3 trilinear mesh).  The wave transposes through LDS instead, CH values per element at a time, so that 64 / CH elements' chunks of
  // CH * 8 contiguous bytes go out per instruction.
  constexpr int NE = NBT * NBR, CH = NE % 16 == 0 ? 16 : NE % 9 == 0 ? 9 : NE % 4 == 0 ? 4 : 1, PADW = CH | 1;
  double *stg = sT + p.ldst_doubles + (threadIdx.x >> 6) * 64 * PADW;
  const int lane = threadIdx.x & 63;
  const i64 wave0 = ie0 - lane;  // first element of this wave
#pragma unroll
  for (int c0 = 0; c0 < NE; c0 += CH) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int l = c0 + j, m = l / NBR, n = l % NBR;
      stg[lane * PADW + j] = (SYMD && n < m) ? A[n < NBT ? n : 0][m < NBR ? m : 0] : A[m][n];
    }
    __builtin_amdgcn_wave_barrier();
    // lanes (el, idx): CH consecutive values of element el; 64 / CH elements per pass (CH = 9: 7 elements, one idle lane)
    constexpr int EPP = 64 / CH;
    const int sub = lane / CH, idx = lane - sub * CH;
#pragma unroll
    for (int pass = 0; pass < (64 + EPP - 1) / EPP; ++pass) {
      const int el = pass * EPP + sub;
      if (sub < EPP && el < 64 && wave0 + el < p.nelems) {
        const double v = stg[el * PADW + idx];
        if (!(MDBGL(p) & 1)) p.local[(wave0 + el) * NE + c0 + idx] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- owner blocks (NH_MATRIX_FUSED): one pass, no scratch, no global atomics ------------------------------------------------------------------
// The two-pass reduction above moves every local matrix through HBM twice and reads a 4-byte source index per contribution: 4.6 x the algorithmic bytes
// on the 128^3 trilinear mesh.  Here the ROWS are clustered (nh_blockplan.inc): dofs sorted by the Morton code of the centroid of the first element that contains
// them, Morton boxes of at most R rows form a block whose CSR rows fit the LDS of a workgroup (8 x 8 x 4 node boxes on a structured mesh: 1.6 x the element
// arithmetic), and a block recomputes every element that touches one of its rows, adds the entries of ITS rows in LDS (ds_add_f64) and writes each row once.
// Read per visit: element id, 2 bytes per local row (row within the block and turn, or "not mine"), 1 byte per local entry (position within its CSR row), the geometry.
// The sums are ORDERED: the block plan carries, for every (visit, local row), the TURN of that visit within its row (visits ascending = elements ascending: the order
// of the reference's numpy.add.at, numeric.py:434-460); the visits of a round add in rounds -- in round t every row receives the contribution of its t-th visitor and
// no other, a workgroup barrier separates the rounds -- so every CSR entry is the sum of its contributions in visit order whatever the wave scheduling: bit-reproducible.
// (Round 4 let a visit spin on a per-row LDS turn counter instead: 23 % of the kernel in waits and retries.  Measured and dropped in round 5: one LDS slot per
// CONTRIBUTION, stored without atomics and summed at the flush -- 64 slots for the 27 entries of a trilinear row shrink the blocks to 128 rows, and their fixed
// chain of dependent loads per block made the kernel slower, 0.88-1.0 against 0.55 ms.)
constexpr size_t FUSED_LDS = 66 * 1024;  // accumulator + row tables of a block (with staged element tables two workgroups fit the 160 kB of a CU)
constexpr int FUSED_NT = 256;        // threads of a block (~405 visits of a 256-row block: two rounds)
constexpr int FUSED_NT_MAX = 384;    // launch bound (ablation builds may launch more threads)

#include "nh_blockplan.inc"

// per rank position: first entry of the row in the value array, its length
__global__ void k_fs_rowinfo(i64 nrows, const unsigned *order, const i64 *srowptr, i64 *rstart, int32_t *rlen) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  const i64 r = order[i];
  rstart[i] = srowptr[r];
  rlen[i] = (int32_t)(srowptr[r + 1] - srowptr[r]);
}
// entries of the fullest block
__global__ void k_fs_maxents(int nblocks, const i64 *bptr, const i64 *epos, int *max_ents) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nblocks) atomicMax(max_ents, (int)(epos[bptr[b + 1]] - epos[bptr[b]]));
}
// (visit, local row) pairs: key = rank position of the row if it belongs to the visiting block, else ~0; value = the pair's index.  Sorted by key (stable: the
// pairs of one row stay in visit order), the position of a pair within its key group is its TURN.
__global__ void k_fs_vkeys(i64 nvisits, int nbt, const int32_t *blk, const int32_t *dofs, const int32_t *rank, const i64 *vptr, const unsigned *vlist, unsigned *pkey, unsigned *pval) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvisits * nbt) return;
  const i64 v = i / nbt;
  const int rp = rank[dofs[(i64)vlist[v] * nbt + (i - v * nbt)]];
  const int b = blk[rp];
  pkey[i] = (v >= vptr[b] && v < vptr[b + 1]) ? (unsigned)rp : 0xffffffffu;  // (visit v belongs to the block of this row?)
  pval[i] = (unsigned)i;
}
// vrow[pair] = row within the block | turn << 9 (0xffff: the row belongs to another block); flags[0]: a turn does not fit, flags[1]: turns of the busiest row
__global__ void k_fs_vrow(i64 n, const i64 *bptr, const int32_t *blk, const unsigned *pkey, const unsigned *pval, uint16_t *vrow, int *flags) {
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned key = pkey[j];
  if (key == 0xffffffffu) {
    vrow[pval[j]] = (uint16_t)0xffff;
    return;
  }
  int seq = 0;
  while (seq < 126 && j - seq - 1 >= 0 && pkey[j - seq - 1] == key) ++seq;
  if (seq >= 126) atomicOr(flags, 1);
  atomicMax(flags + 1, seq + 1);
  vrow[pval[j]] = (uint16_t)((unsigned)(key - (unsigned)bptr[blk[key]]) | (unsigned)seq << 9);
}
// the element map (position of entry (m, n) within the CSR row of its test dof) narrowed to a byte
__global__ void k_fs_cpos(i64 n, const int32_t *emap, uint8_t *cpos, int *bad) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int d = emap[i];
  if (d < 0 || d > 255) atomicOr(bad, 1);
  cpos[i] = (uint8_t)d;
}

// tables of the sum-factorised trilinear routine (nh_p1hex_math.inc; the fields it reads of the structured kernels' argument struct)
struct P1Tab {
  double n[2][2];      // n[a][q] = N_a(g_q): 1-D shape functions at the 1-D Gauss points
  double c[3][2];      // c[x+y][q] = n[x][q] n[y][q]
  double wk[2][2][2];  // kappa w_qa w_qb w_qc
  double wm[2][2][2];  // mass w_qa w_qb w_qc
};

// 1 / d for d > 0 of ordinary magnitude (as in nh_assemble_p1hex.hip)
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.), r, r);
  r = fma(fma(-d, r, 1.), r, r);
  return r;
}

struct FusK {
  LocK loc;
  P1Tab tab;
  double *values;
  int store;
  int R, max_ents, nturns;  // rows of the largest block, entries of the fullest, visitors of the busiest row
  const i64 *bptr, *vptr, *rstart, *epos;
  const int32_t *vlist;
  const uint16_t *vrow;
  const uint8_t *cpos;
};

// LDS of an owner block behind the staged element tables: [row starts R x i64][accumulator offsets R + 1 ints, padded][accumulator: the rows of the block, entry by entry]
struct FusLds {
  i64 *rstart;
  int *eoff;
  double *acc;
};
__device__ __forceinline__ FusLds fused_lds(double *base, int R) {
  FusLds l;
  l.rstart = reinterpret_cast<i64 *>(base);
  l.eoff = reinterpret_cast<int *>(l.rstart + R);
  l.acc = base + R + (R + 2) / 2;
  return l;
}
static size_t fused_lds_bytes(int R, int max_ents) { return sizeof(double) * ((size_t)max_ents + (size_t)R + ((size_t)R + 2) / 2); }

// row tables of block b, accumulator zeroed
__device__ __forceinline__ int fused_stage_rows(const FusK &p, int b, const FusLds &l) {
  const i64 r0 = p.bptr[b];
  const int nr = (int)(p.bptr[b + 1] - r0);
  const i64 e0 = p.epos[r0];
  for (int i = threadIdx.x; i <= nr; i += blockDim.x) {
    if (i < nr) l.rstart[i] = p.rstart[r0 + i];
    l.eoff[i] = (int)(p.epos[r0 + i] - e0);
  }
  const int nent = (int)(p.epos[r0 + nr] - e0);
  for (int i = threadIdx.x; i < nent; i += blockDim.x) l.acc[i] = 0.;
  return nr;
}
// the rows of the block, half a wave per row (row starts and offsets are staged: no chain of dependent global loads); a row is one contiguous piece of the value array
__device__ __forceinline__ void fused_flush_rows(const FusK &p, const FusLds &l, int nr) {
  const int hl = threadIdx.x & 31, NT = blockDim.x, step = NT >> 5;
  // four rows per half-wave and trip: their offsets, starts and first 32 entries are read from LDS together (one row per trip was a chain of two LDS latencies and a
  // store per 216 bytes: 0.13 of the 0.52 ms of 128^3 trilinear Laplace went to this loop, profiles/r05_owner_vector.md)
  for (int i0 = threadIdx.x >> 5; i0 < nr; i0 += 4 * step) {
    int lo[4], len[4];
    i64 rs[4];
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * step;
      const bool in = i < nr;
      lo[k] = in ? l.eoff[i] : 0;
      len[k] = in ? l.eoff[i + 1] - lo[k] : 0;
      rs[k] = in ? l.rstart[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = hl < len[k] ? l.acc[lo[k] + hl] : 0.;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double *dst = p.values + rs[k];
      if (hl < len[k]) dst[hl] = p.store ? v[k] : dst[hl] + v[k];
      for (int j = hl + 32; j < len[k]; j += 32) dst[j] = p.store ? l.acc[lo[k] + j] : dst[j] + l.acc[lo[k] + j];
    }
  }
}

template <int ND, int NBT, int NBR, bool LDST, bool SYMD>
__global__ __launch_bounds__(FUSED_NT_MAX) void k_fused_scalar(FusK p) {
  constexpr int S = 1 + ND, NG = 1 << ND, NE = NBT * NBR;
  extern __shared__ __attribute__((aligned(16))) double sT[];
  const FusLds l = fused_lds(sT + p.loc.ldst_doubles, p.R);
  const int b = blockIdx.x, NT = blockDim.x;
  if (LDST) {
    const int nt = NBT * p.loc.nq * S, ntr = NBR * p.loc.nq * S, ng = NG * p.loc.nq * S;
    for (int i = threadIdx.x; i < nt; i += NT) sT[i] = p.loc.test.T[i];
    for (int i = threadIdx.x; i < ntr; i += NT) sT[nt + i] = p.loc.trial.T[i];
    if (p.loc.geom.kind == NH_GEOM_ISO && p.loc.geom.ngb == NG)
      for (int i = threadIdx.x; i < ng; i += NT) sT[nt + ntr + i] = p.loc.geom.gT[i];
  }
  const int nr = fused_stage_rows(p, b, l);
  __syncthreads();
  const i64 v0 = p.vptr[b], v1 = p.vptr[b + 1];
  for (i64 base = v0; base < v1; base += NT) {  // (the same number of rounds for every thread: the turns end in workgroup barriers)
    const bool act = base + threadIdx.x < v1;
    const i64 i = act ? base + threadIdx.x : v1 - 1;  // (threads behind the last visit shadow it and add nothing)
    const i64 e = p.vlist[i];
    // rows of the element within the block with their turns, and the positions of its entries within the rows: whole words where the sizes allow
    uint16_t vr[NBT];
    uint8_t cp[NE];
    if constexpr (NBT % 8 == 0) {
#pragma unroll
      for (int k = 0; k < NBT / 8; ++k) *reinterpret_cast<uint4 *>(vr + 8 * k) = reinterpret_cast<const uint4 *>(p.vrow + i * NBT)[k];
    } else {
#pragma unroll
      for (int m = 0; m < NBT; ++m) vr[m] = p.vrow[i * NBT + m];
    }
    if constexpr (NE % 16 == 0) {
#pragma unroll
      for (int k = 0; k < NE / 16; ++k) *reinterpret_cast<uint4 *>(cp + 16 * k) = reinterpret_cast<const uint4 *>(p.cpos + e * NE)[k];
    } else {
#pragma unroll
      for (int k = 0; k < NE; ++k) cp[k] = p.cpos[e * NE + k];
    }
    double A[NBT][NBR];
    local_scalar_matrix<ND, NBT, NBR, LDST, SYMD>(p.loc, sT, e, e, A);
    // turn t: every row of the block receives the contribution of its t-th visitor -- no two threads meet in a row, the barrier orders the turns
    for (int t = 0; t < p.nturns; ++t) {
#pragma unroll
      for (int m = 0; m < NBT; ++m) {
        if (!act || vr[m] == 0xffff || (vr[m] >> 9) != t) continue;
        double *row = l.acc + l.eoff[vr[m] & 511];
#pragma unroll
        for (int n = 0; n < NBR; ++n) atomicAdd(row + cp[m * NBR + n], (SYMD && n < m) ? A[n < NBT ? n : 0][m < NBR ? m : 0] : A[m][n]);
      }
      __syncthreads();
    }
  }
  fused_flush_rows(p, l, nr);
}

// the same owner blocks for TRILINEAR HEXAHEDRA with the 2 x 2 x 2 Gauss scheme (recognised from the tables the caller passes: nh_fused_scalar) and forms
// kappa grad.grad + mass phi phi: the element routine is the sum-factorised one of the structured kernels (nh_p1hex_math.inc, ~1.1 k instead of ~3.2 k f64 instructions);
// connectivity, pattern and block plan stay those of the any-mesh entry.  (An exactly singular element gives inf / NaN here, not the all-NaN inverse of numeric.inv.)
template <bool MASS>
__global__ __launch_bounds__(FUSED_NT_MAX) void k_fused_p1hex(FusK fp) {
  constexpr int NBT = 8, NE = 64;
  extern __shared__ __attribute__((aligned(16))) double sT[];
  const FusLds l = fused_lds(sT, fp.R);
  const int b = blockIdx.x, NT = blockDim.x;
  const int nr = fused_stage_rows(fp, b, l);
  __syncthreads();
  const P1Tab &p = fp.tab;
  const i64 v0 = fp.vptr[b], v1 = fp.vptr[b + 1];
  for (i64 base = v0; base < v1; base += NT) {
    const bool act = base + threadIdx.x < v1;
    const i64 i = act ? base + threadIdx.x : v1 - 1;
    const i64 e = fp.vlist[i];
    uint16_t vr[NBT];
    uint8_t cp[NE];
    *reinterpret_cast<uint4 *>(vr) = *reinterpret_cast<const uint4 *>(fp.vrow + i * NBT);
#pragma unroll
    for (int k = 0; k < NE / 16; ++k) *reinterpret_cast<uint4 *>(cp + 16 * k) = reinterpret_cast<const uint4 *>(fp.cpos + e * NE)[k];
    double X[2][2][2][3];
    {
      int idx[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) idx[a] = fp.loc.geom.gdofs[e * 8 + a];
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int d = 0; d < 3; ++d) X[a >> 2][(a >> 1) & 1][a & 1][d] = fp.loc.geom.verts[(i64)idx[a] * 3 + d];
    }
    double qs[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) qs[q] = 1.;
    if (fp.loc.scale) {
#pragma unroll
      for (int q = 0; q < 8; ++q) qs[q] = fp.loc.scale[e * 8 + q];
    }
    constexpr bool hasm = MASS;
    double R0[3][3], R1[3][3], R2[3][3], W01[2][2][3], W02[2][2][3], W12[2][2][3], Mm[3][3][3];
#define NH_P1HEX_QS(q) qs[q]
#define NH_P1HEX_QM(q) qs[q]
#include "nh_p1hex_math.inc"
#undef NH_P1HEX_QS
#undef NH_P1HEX_QM
    auto entry = [&](int a, int bb) {  // K[a][bb], a <= bb: nine signed table values (+ the mass term), as in nh_p1hex_element.inc
      const int a0 = a >> 2, a1 = (a >> 1) & 1, a2 = a & 1;
      const int b0 = bb >> 2, b1 = (bb >> 1) & 1, b2 = bb & 1;
      const int p0 = a0 + b0, p1 = a1 + b1, p2 = a2 + b2;
      const double s00 = (a0 == b0) ? 1. : -1., s11 = (a1 == b1) ? 1. : -1., s22 = (a2 == b2) ? 1. : -1.;
      const double s01 = (a0 == b1) ? 1. : -1., s10 = (b0 == a1) ? 1. : -1.;
      const double s02 = (a0 == b2) ? 1. : -1., s20 = (b0 == a2) ? 1. : -1.;
      const double s12 = (a1 == b2) ? 1. : -1., s21 = (b1 == a2) ? 1. : -1.;
      double k = s00 * R0[p1][p2] + s11 * R1[p0][p2] + s22 * R2[p0][p1] + s01 * W01[b0][a1][p2] + s10 * W01[a0][b1][p2] + s02 * W02[b0][a2][p1] + s20 * W02[a0][b2][p1] +
                 s12 * W12[b1][a2][p0] + s21 * W12[a1][b2][p0];
      if (hasm) k += Mm[p0][p1][p2];
      return k;
    };
    double K[36];  // upper triangle, row major
    {
      int q = 0;
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int bb = a; bb < 8; ++bb) K[q++] = entry(a, bb);
    }
    for (int t = 0; t < fp.nturns; ++t) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (!act || vr[m] == 0xffff || (vr[m] >> 9) != t) continue;
        double *row = l.acc + l.eoff[vr[m] & 511];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const int lo = m < n ? m : n, hi = m < n ? n : m;
          atomicAdd(row + cp[m * 8 + n], K[lo * 8 - lo * (lo - 1) / 2 + hi - lo]);
        }
      }
      __syncthreads();
    }
  }
  fused_flush_rows(fp, l, nr);
}

double *g_scratch = nullptr;
size_t g_scratch_cap = 0;

}  // namespace

int nh_gather_scratch(size_t doubles, double **out) {
  if (doubles > g_scratch_cap) {
    if (g_scratch) {
      NH_CHECK_HIP(hipDeviceSynchronize());
      NH_CHECK_HIP(hipFree(g_scratch));
      g_scratch = nullptr, g_scratch_cap = 0;
    }
    NH_CHECK_HIP(hipMalloc((void **)&g_scratch, doubles * sizeof(double)));
    g_scratch_cap = doubles;
  }
  *out = g_scratch;
  return NH_OK;
}

extern "C" int nh_release_scratch(void) {
  if (g_scratch) {
    NH_CHECK_HIP(hipDeviceSynchronize());
    NH_CHECK_HIP(hipFree(g_scratch));
    g_scratch = nullptr, g_scratch_cap = 0;
  }
  NH_CHECK_HIP(hipDeviceSynchronize());
  return nh_p1hex_tiles_release();
}

int nh_gather_prepare(nh_pattern *p, const nh_basis &test, const int32_t *elist, hipStream_t s) {
  if (p->gsrc) return NH_OK;
  NH_REQUIRE(p->emap_len < (1ll << 32) && p->nnz < (1ll << 32), "NH_MATRIX_GATHER: pattern too large for 32-bit gather indices");
  NH_REQUIRE(test.dofs_dev, "NH_MATRIX_GATHER: test dofs missing");
  const i64 n = p->emap_len, nnz = p->nnz;
  unsigned *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr;
  int *counts = nullptr;
  i64 *gptr64 = nullptr;
  void *tmp = nullptr;
  int rc = NH_OK;
#define GP_CHECK(expr)                                                                          \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      nh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
      rc = NH_EHIP;                                                                             \
      goto done;                                                                                \
    }                                                                                           \
  } while (0)
  {
    GP_CHECK(hipMalloc((void **)&keys, n * 4));
    GP_CHECK(hipMalloc((void **)&vals, n * 4));
    GP_CHECK(hipMalloc((void **)&keys2, n * 4));
    GP_CHECK(hipMalloc((void **)&vals2, n * 4));
    GP_CHECK(hipMalloc((void **)&counts, (nnz + 1) * 4));
    GP_CHECK(hipMemsetAsync(counts, 0, (nnz + 1) * 4, s));
    hipLaunchKernelGGL(k_gather_keys, dim3((unsigned)std::min<i64>(p->nelems, 1 << 20)), dim3(64), 0, s, p->nelems, elist, to_k(test), p->nbr, p->eoff, p->srowptr, p->emap, keys,
                       vals, counts);
    GP_CHECK(hipGetLastError());
    int bits = 1;
    while ((1ll << bits) < nnz) ++bits;
    size_t tmpsz = 0;
    GP_CHECK(rocprim::radix_sort_pairs(nullptr, tmpsz, keys, keys2, vals, vals2, (size_t)n, 0, bits, s));
    GP_CHECK(hipMalloc(&tmp, tmpsz));
    GP_CHECK(rocprim::radix_sort_pairs(tmp, tmpsz, keys, keys2, vals, vals2, (size_t)n, 0, bits, s));  // stable: sources stay in (element, m, n) order
    GP_CHECK(hipMalloc((void **)&gptr64, (nnz + 1) * sizeof(i64)));
    if ((rc = nh_scan_exclusive(counts, gptr64, nnz, s)) != NH_OK) goto done;
    GP_CHECK(hipMalloc((void **)&p->gptr, (nnz + 1) * sizeof(unsigned)));  // (emap_len < 2^32: 32-bit offsets)
    hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<i64>((nnz + 256) / 256, 1 << 16)), dim3(256), 0, s, nnz + 1, gptr64, p->gptr);
    GP_CHECK(hipGetLastError());
    GP_CHECK(hipMalloc((void **)&p->grow, std::max<i64>(nnz, 1) * 4));
    hipLaunchKernelGGL(k_rowof, dim3((unsigned)std::min<i64>((p->nrows + 255) / 256, 1 << 16)), dim3(256), 0, s, p->nrows, p->srowptr, p->grow);
    GP_CHECK(hipGetLastError());
    GP_CHECK(hipStreamSynchronize(s));
    p->gsrc = reinterpret_cast<int32_t *>(vals2);
    vals2 = nullptr;
  }
done:
#undef GP_CHECK
  hipFree(keys);
  hipFree(vals);
  hipFree(keys2);
  hipFree(vals2);
  hipFree(counts);
  hipFree(gptr64);
  hipFree(tmp);
  if (rc != NH_OK) {
    hipFree(p->gptr), hipFree(p->grow);
    p->gptr = nullptr, p->grow = nullptr, p->gsrc = nullptr;
  }
  return rc;
}

// ---- pass 1 for larger scalar elements (quadratic hexahedra, cubic tensor splines): a thread owns MB ROWS of the local matrix ----------
// NBT * NBR sums do not fit one thread any more (27 x 27 doubles); the NBT / MB threads of an element sit in neighbouring lanes and repeat only
// the geometry of the point.  The form and BOTH Jacobian inverses are applied to the MB test rows (tw = w J^-1 C J^-T dt in the reference
// frame), so the inner loop over the trial functions is the bare contraction A[m][n] += sum_s tw[m][s] T_n[q][s] with the tabulated
// reference values: MB * S multiply-adds per trial function and point, nothing else.
template <int ND, int NBT, int NBR, int MB, bool LDST>
// (27 functions, tables through L1: two waves per SIMD with 19 spilled doubles beat one wave without, 3.89 -> 3.47 ms for 64^3 quadratic splines; with the tables
// in LDS the 71 kB workgroups allow one wave per SIMD either way, and the spills cost: 2.62 -> 2.83 ms)
__global__ __launch_bounds__(128, (ND == 3 && NBT == 27 && !LDST ? 2 : 1)) void k_local_rows(LocK p) {
  constexpr int S = 1 + ND, NG = 1 << ND, NMB = NBT / MB;
  static_assert(NBT % MB == 0, "row blocks");
  extern __shared__ __attribute__((aligned(16))) double sT[];
  // tables in LDS: test and trial when the basis has one element class (LDST), the geometry tables (one class by construction) in any case
  {
    const int nt = LDST ? NBT * p.nq * S : 0, nr = LDST ? NBR * p.nq * S : 0, ng = NG * p.nq * S;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) sT[i] = p.test.T[i];
    for (int i = threadIdx.x; i < nr; i += blockDim.x) sT[nt + i] = p.trial.T[i];
    if (p.geom.kind == NH_GEOM_ISO && p.geom.ngb == NG)
      for (int i = threadIdx.x; i < ng; i += blockDim.x) sT[nt + nr + i] = p.geom.gT[i];
    __syncthreads();
  }
  const i64 nthreads = p.nelems * NMB;
  const i64 t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 t = min(t0, nthreads - 1);  // (lanes behind the last row block recompute it: the store needs whole waves, and skips them)
  const i64 ie = t / NMB;
  const int mb = (int)(t - ie * NMB) * MB;
  const i64 e = p.elist ? p.elist[ie] : ie;
  double A[MB][NBR];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int n = 0; n < NBR; ++n) A[m][n] = 0;
  const bool iso = p.geom.kind == NH_GEOM_ISO && p.geom.ngb == NG;
  double X[NG][ND];
  if (iso) {
#pragma unroll
    for (int a = 0; a < NG; ++a) {
      const i64 v = p.geom.gdofs[e * NG + a];
#pragma unroll
      for (int i = 0; i < ND; ++i) X[a][i] = p.geom.verts[v * ND + i];
    }
  }
  const double *Tt = (LDST ? sT : p.test.T + bfn(p.test, e) * p.nq * S) + (size_t)mb * p.nq * S;
  const double *Tr = LDST ? sT + NBT * p.nq * S : p.trial.T + bfn(p.trial, e) * p.nq * S;
  const double *gT = sT + (LDST ? (NBT + NBR) * p.nq * S : 0);
  for (int q = 0; q < p.nq; ++q) {
    double Ji[ND][ND], det;
    if (iso) {
      double J[ND][ND];
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int j = 0; j < ND; ++j) J[i][j] = 0;
#pragma unroll
      for (int a = 0; a < NG; ++a) {
        const double *tg = gT + ((i64)a * p.nq + q) * S;
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
          for (int j = 0; j < ND; ++j) J[i][j] += X[a][i] * tg[1 + j];
      }
      invert<ND>(J, Ji, det);
      if (p.geom.bnd_axis >= 0) {
        double s2 = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
          for (int i = 0; i < ND; ++i)
            if (j == p.geom.bnd_axis) s2 += Ji[j][i] * Ji[j][i];
        det *= sqrt(s2);
      }
      if (p.geom.nograd) {
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
          for (int j = 0; j < ND; ++j) Ji[i][j] = 0.;
      }
    } else
      geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
    const double w = p.weights[q] * fabs(det) * (p.scale ? p.scale[(p.by_elem ? e : ie) * p.nq + q] : 1.);
    // test rows: physical gradients, form, weight, back to the reference frame
    double tw[MB][S];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const double *T = Tt + ((size_t)m * p.nq + q) * S;
      double dt[S], cd[S];
      dt[0] = T[0];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        double sum = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j) sum += T[1 + j] * Ji[j][i];
        dt[1 + i] = sum;
      }
#pragma unroll
      for (int b = 0; b < S; ++b) {
        double sum = 0;
#pragma unroll
        for (int a = 0; a < S; ++a) sum += dt[a] * p.C[a * S + b];
        cd[b] = w * sum;
      }
      tw[m][0] = cd[0];
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        double sum = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) sum += Ji[j][i] * cd[1 + i];
        tw[m][1 + j] = sum;
      }
    }
    // The NMB threads of an element sit in one wave (NMB divides 64) and, with test == trial tables, have between them just read all NBT table rows
    // of the point: they are exchanged through LDS instead of read again, row by row, by every thread -- the rows are nq * S doubles apart in the
    // function-major tables, 64 lines per load (32^3 tricubic splines: 11.9 -> 6.7 ms per assembly; 1024^2 bicubic splines: 3.3 -> 3.1 ms).
    constexpr bool XROW = 64 % NMB == 0 && NBT == NBR && !LDST && (64 / NMB) * NBT * S <= 64 * 17;
    if (XROW && Tt - (size_t)mb * p.nq * S == Tr) {  // (uniform: same table for test and trial)
      double *rowbuf = sT + p.ldst_doubles + (threadIdx.x >> 6) * 64 * 17 + ((threadIdx.x & 63) / NMB) * NBT * S;  // (LDS pointer: kept apart from the global one, or the loads turn into flat loads)
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const double *T = Tt + ((size_t)m * p.nq + q) * S;
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) rowbuf[(mb + m) * S + s2] = T[s2];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int n = 0; n < NBR; ++n) {
        double tn[S];
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) tn[s2] = rowbuf[n * S + s2];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int s2 = 0; s2 < S; ++s2) A[m][n] += tw[m][s2] * tn[s2];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // (the rows are overwritten at the next point)
    } else {
#pragma unroll
      for (int n = 0; n < NBR; ++n) {
        const double *T = Tr + ((size_t)n * p.nq + q) * S;
        double tn[S];
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) tn[s2] = T[s2];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int s2 = 0; s2 < S; ++s2) A[m][n] += tw[m][s2] * tn[s2];
      }
    }
  }
  // store: the thread's MB rows are MB * NBR contiguous doubles of the element-major scratch, neighbouring lanes hold neighbouring chunks -- through
  // the per-wave transposition of k_local_scalar, CH values per thread at a time
  constexpr int NE = MB * NBR, CH = NE % 16 == 0 ? 16 : NE % 9 == 0 ? 9 : NE % 4 == 0 ? 4 : 1, PADW = CH | 1;
  double *stg = sT + p.ldst_doubles + (threadIdx.x >> 6) * 64 * PADW;
  const int lane = threadIdx.x & 63;
  const i64 wave0 = t0 - lane;
#pragma unroll
  for (int c0 = 0; c0 < NE; c0 += CH) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int l = c0 + j;
      stg[lane * PADW + j] = A[l / NBR][l % NBR];
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int EPP = 64 / CH;
    const int sub = lane / CH, idx = lane - sub * CH;
#pragma unroll
    for (int pass = 0; pass < (64 + EPP - 1) / EPP; ++pass) {
      const int el = pass * EPP + sub;
      if (sub < EPP && el < 64 && wave0 + el < nthreads) p.local[(wave0 + el) * NE + c0 + idx] = stg[el * PADW + idx];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- pass 1 for vector-valued blocks on small bases (trilinear / bilinear / biquadratic elasticity): a thread owns ONE test function -----
// i.e. the NC rows (m, c) of the local matrix with all NB * NC columns, kept as A[n][c][d]: the scratch holds the NC x NC block of every scalar
// pair (m, n) contiguously (what k_gather_values expects), so the thread's rows are NB * NC * NC contiguous doubles.  Per point: the physical
// gradient of the test function, then per (c, d) the row vector w J^-1 (C[c][.][d][.]^T dt) in the reference frame (blocks outside the form's
// mask are skipped), then the bare contraction with the tabulated reference values of the NB trial functions.
struct LocVK {
  i64 nelems;
  const int32_t *elist;
  int nq;
  const double *weights;
  GeomK geom;
  BasisK test;
  const double *scale;
  int by_elem;
  double C[144];  // [c][a][d][b], NC <= 3, S <= 4
  unsigned char mask[3][3];
  double *local;
  int ldst_doubles;
  double lam, mu, mu2;  // ISOF instantiations: C[c][1+a][d][1+b] = lam d_ca d_db + mu d_cd d_ab + mu2 d_cb d_ad, nothing on the value slots
};

// ISOF: the isotropic three-parameter family (linear elasticity: lam div div + 2 mu sym grad : sym grad) applied in closed form -- per (c, d) two or three multiply-adds
// from registers instead of S * S reads of the form tensor from LDS and as many multiply-adds: the kernel was bound by those (uniform) LDS reads, 240 per point and thread.
template <int ND, int NB, int NC, bool LDST, bool ISOF>
__global__ __launch_bounds__(128) void k_local_rows_v(LocVK p) {
  constexpr int S = 1 + ND, NG = 1 << ND, NCD = NC * NC;
  // the form tensor goes to LDS: its NC * S * NC * S doubles do not fit the scalar registers (read from the kernel arguments they were spilled to
  // VGPR lanes: 600 v_readlane per point, 4.3 instead of 3.8 ms for 96^3 trilinear elasticity), a uniform ds_read is a broadcast
  extern __shared__ __attribute__((aligned(16))) double sC[];
  double *sT = sC + 144;
  for (int i = threadIdx.x; i < 144; i += blockDim.x) sC[i] = p.C[i];
  if (LDST) {
    const int nt = NB * p.nq * S, ng = NG * p.nq * S;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) sT[i] = p.test.T[i];
    if (p.geom.kind == NH_GEOM_ISO && p.geom.ngb == NG)
      for (int i = threadIdx.x; i < ng; i += blockDim.x) sT[nt + i] = p.geom.gT[i];
  }
  __syncthreads();
  const i64 nthreads = p.nelems * NB;
  const i64 t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 t = min(t0, nthreads - 1);
  const i64 ie = t / NB;
  const int m = (int)(t - ie * NB);
  const i64 e = p.elist ? p.elist[ie] : ie;
  double A[NB][NC][NC];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int d = 0; d < NC; ++d) A[n][c][d] = 0;
  const bool iso = p.geom.kind == NH_GEOM_ISO && p.geom.ngb == NG;
  // vertices of the element: in registers, or -- when the NB threads of an element never straddle two workgroups -- once per element in LDS (24
  // doubles that every one of the 8 threads of a trilinear element would hold: with them the kernel shuttles accumulators through AGPRs)
  constexpr bool XLDS = 128 % NB == 0 && NB >= NG && ND == 3;
  double X[XLDS ? 1 : NG][ND];
  double *sX = sT + p.ldst_doubles + 2 * 64 * 17 + (threadIdx.x / NB) * NG * ND;
  if (iso) {
    if (XLDS) {
      if (m < NG) {
        const i64 v = p.geom.gdofs[e * NG + m];
#pragma unroll
        for (int i = 0; i < ND; ++i) sX[m * ND + i] = p.geom.verts[v * ND + i];
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int a = 0; a < NG; ++a) {
        const i64 v = p.geom.gdofs[e * NG + a];
#pragma unroll
        for (int i = 0; i < ND; ++i) X[XLDS ? 0 : a][i] = p.geom.verts[v * ND + i];
      }
    }
  }
  const double *Tall = LDST ? sT : p.test.T + bfn(p.test, e) * p.nq * S;
  const double *Tm = Tall + (size_t)m * p.nq * S;
  const double *gT = LDST ? sT + NB * p.nq * S : p.geom.gT;
  auto geometry = [&](int q, double (&Ji)[ND][ND], double &w) {
    double det;
    if (iso) {
      double J[ND][ND];
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int j = 0; j < ND; ++j) J[i][j] = 0;
#pragma unroll
      for (int a = 0; a < NG; ++a) {
        const double *tg = gT + ((i64)a * p.nq + q) * S;
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
          for (int j = 0; j < ND; ++j) J[i][j] += (XLDS ? sX[a * ND + i] : X[XLDS ? 0 : a][i]) * tg[1 + j];
      }
      invert<ND>(J, Ji, det);
      if (p.geom.bnd_axis >= 0) {
        double s2 = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
          for (int i = 0; i < ND; ++i)
            if (j == p.geom.bnd_axis) s2 += Ji[j][i] * Ji[j][i];
        det *= sqrt(s2);
      }
      if (p.geom.nograd) {
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
          for (int j = 0; j < ND; ++j) Ji[i][j] = 0.;
      }
    } else
      geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
    w = p.weights[q] * fabs(det) * (p.scale ? p.scale[(p.by_elem ? e : ie) * p.nq + q] : 1.);
  };
  // The NB threads of an element sit in one wave (NB divides 64): thread m computes the geometry of the points m, m + NB, ... ONCE and leaves the
  // inverse Jacobian and the weight in LDS for the others -- repeated by every thread it was 2.6 of the 3.7 ms of this kernel on 96^3 trilinear
  // elasticity (ablation: form, contraction and stores switched off).
  constexpr bool SHARE = 64 % NB == 0;
  constexpr int GW = ND * ND + 1;
  double *sG = sT + p.ldst_doubles + 2 * 64 * 17 + 128 * 3 + (threadIdx.x / NB) * p.nq * GW;
  if (SHARE) {
    for (int q = m; q < p.nq; q += NB) {
      double Ji[ND][ND], w;
      geometry(q, Ji, w);
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int j = 0; j < ND; ++j) sG[q * GW + i * ND + j] = Ji[i][j];
      sG[q * GW + ND * ND] = w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  for (int q = 0; q < p.nq; ++q) {
    double Ji[ND][ND], w;
    if (SHARE) {
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int j = 0; j < ND; ++j) Ji[i][j] = sG[q * GW + i * ND + j];
      w = sG[q * GW + ND * ND];
    } else
      geometry(q, Ji, w);
    double dt[S];
    {
      const double *T = Tm + (size_t)q * S;
      dt[0] = T[0];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        double sum = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j) sum += T[1 + j] * Ji[j][i];
        dt[1 + i] = sum;
      }
    }
    double gJ[ND];  // ISOF: J^-1 applied to the physical gradient of the test function once per point
    if constexpr (ISOF) {
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        double sum = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) sum += Ji[j][i] * dt[1 + i];
        gJ[j] = p.mu * sum;
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double tw[NC][S];
#pragma unroll
      for (int d = 0; d < NC; ++d) {
        if constexpr (ISOF) {
          static_assert(!ISOF || NC == ND, "isotropic form: one component per axis");
          tw[d][0] = 0.;
          const double lc = p.lam * dt[1 + c], md = p.mu2 * dt[1 + d];
#pragma unroll
          for (int j = 0; j < ND; ++j) {
            double t = lc * Ji[j][d < ND ? d : 0] + md * Ji[j][c < ND ? c : 0];
            if (c == d) t += gJ[j];
            tw[d][1 + j] = w * t;
          }
        } else if (p.mask[c][d]) {  // (uniform)
          double cd[S];
#pragma unroll
          for (int b = 0; b < S; ++b) {
            double sum = 0;
#pragma unroll
            for (int a = 0; a < S; ++a) sum += dt[a] * sC[((c * S + a) * NC + d) * S + b];
            cd[b] = w * sum;
          }
          tw[d][0] = cd[0];
#pragma unroll
          for (int j = 0; j < ND; ++j) {
            double sum = 0;
#pragma unroll
            for (int i = 0; i < ND; ++i) sum += Ji[j][i] * cd[1 + i];
            tw[d][1 + j] = sum;
          }
        } else {
#pragma unroll
          for (int s2 = 0; s2 < S; ++s2) tw[d][s2] = 0.;
        }
      }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const double *T = Tall + ((size_t)n * p.nq + q) * S;
        double tn[S];
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) tn[s2] = T[s2];
#pragma unroll
        for (int d = 0; d < NC; ++d)
#pragma unroll
          for (int s2 = ISOF ? 1 : 0; s2 < S; ++s2) A[n][c][d] += tw[d][s2] * tn[s2];
      }
    }
  }
  constexpr int NE = NB * NCD, CH = NE % 16 == 0 ? 16 : NE % 8 == 0 ? 8 : NE % 9 == 0 ? 9 : NE % 4 == 0 ? 4 : 1, PADW = CH | 1;  // (8: 64-byte chunks, aligned)
  double *stg = sT + p.ldst_doubles + (threadIdx.x >> 6) * 64 * PADW;
  const int lane = threadIdx.x & 63;
  const i64 wave0 = t0 - lane;
#pragma unroll
  for (int c0 = 0; c0 < NE; c0 += CH) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int l = c0 + j, n = l / NCD, cd = l % NCD;
      stg[lane * PADW + j] = A[n][cd / NC][cd % NC];
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int EPP = 64 / CH;
    const int sub = lane / CH, idx = lane - sub * CH;
#pragma unroll
    for (int pass = 0; pass < (64 + EPP - 1) / EPP; ++pass) {
      const int el = pass * EPP + sub;
      if (sub < EPP && el < 64 && wave0 + el < nthreads) p.local[(wave0 + el) * NE + c0 + idx] = stg[el * PADW + idx];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

int nh_gather_values(const nh_pattern *p, const double *local, i64 ld, const GSlots &slots, double *values, int store, hipStream_t s, int sym_sources) {
  if (!p->nnz) return NH_OK;
  if (sym_sources == 2) {
    NH_REQUIRE(p->gsrc_tri && p->gmirror && store && slots.nct == slots.ncr && slots.nct <= 2, "gather: the triangular scratch needs its maps, scalar or 2 x 2 blocks and NH_MATRIX_STORE");
    const dim3 grid((unsigned)((p->nnz + 255) / 256));
    if (slots.nct == 1) {
      hipLaunchKernelGGL(k_gather_values_tri1, grid, dim3(256), 0, s, p->nnz, p->gptr, p->gsrc_tri, p->gmirror, local, values);
      hipLaunchKernelGGL(k_mirror_1, grid, dim3(256), 0, s, p->nnz, p->gmirror, values);
    } else {
      hipLaunchKernelGGL(k_gather_values_2x2_tri, grid, dim3(256), 0, s, p->nnz, p->gptr, p->gsrc_tri, p->grow, p->srowptr, p->gmirror, local, slots, values);
      hipLaunchKernelGGL(k_mirror_2x2, grid, dim3(256), 0, s, p->nnz, p->grow, p->srowptr, p->gmirror, slots, values);
    }
  } else if (sym_sources) {
    NH_REQUIRE(p->gsrc_sym && slots.nct == slots.ncr && (slots.nct == 2 || slots.nct == 3), "gather: symmetric sources need their map and 2 x 2 / 3 x 3 blocks");
    hipLaunchKernelGGL(k_gather_values_v<true>, dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, s, p->nnz, p->gptr, p->gsrc_sym, p->grow, p->srowptr, local, slots, values, store);
  } else if (slots.nct == 2 && slots.ncr == 2 && slots.mask[0][0] && slots.mask[0][1] && slots.mask[1][0] && slots.mask[1][1] && slots.dpos[0][1] == slots.dpos[0][0] + 1 &&
             slots.dpos[1][1] == slots.dpos[1][0] + 1 && !getenv("NUTILS_AMD_NO_GATHER_2X2")) {
    hipLaunchKernelGGL(k_gather_values_2x2, dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, s, p->nnz, p->gptr, p->gsrc, p->grow, p->srowptr, local, slots, values, store);
  } else if (slots.nct * slots.ncr == 1)
    hipLaunchKernelGGL(k_gather_values, dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, s, p->nnz, p->gptr, p->gsrc, p->grow, p->srowptr, local, ld, p->nbt * p->nbr, slots,
                       values, store);
  else
    hipLaunchKernelGGL(k_gather_values_v<false>, dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, s, p->nnz, p->gptr, p->gsrc, p->grow, p->srowptr, local, slots, values, store);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

// the map for symmetric producers, derived from the gather map (once per pattern)
int nh_gather_prepare_sym(nh_pattern *p, const nh_basis &test, const int32_t *elist, hipStream_t s) {
  if (p->gsrc_sym) return NH_OK;
  NH_REQUIRE(p->gsrc, "nh_gather_prepare_sym: the gather map comes first");
  NH_REQUIRE(p->emap_len < (1ll << 31), "NH_MATRIX_GATHER: pattern too large for flagged 31-bit gather indices");
  unsigned *canon = nullptr;
  NH_CHECK_HIP(hipMalloc((void **)&canon, std::max<i64>(p->emap_len, 1) * 4));
  hipError_t e = hipMalloc((void **)&p->gsrc_sym, std::max<i64>(p->emap_len, 1) * 4);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_gather_canon, dim3((unsigned)std::min<i64>(p->nelems, 1 << 20)), dim3(64), 0, s, p->nelems, elist, to_k(test), p->nbr, p->eoff, canon);
    hipLaunchKernelGGL(k_gather_remap, dim3((unsigned)std::min<i64>((p->emap_len + 255) / 256, 1 << 16)), dim3(256), 0, s, p->emap_len, p->gsrc, canon, p->gsrc_sym);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  hipFree(canon);
  if (e != hipSuccess) {
    hipFree(p->gsrc_sym);
    p->gsrc_sym = nullptr;
    nh_set_error("nh_gather_prepare_sym failed: %s", hipGetErrorString(e));
    return NH_EHIP;
  }
  return NH_OK;
}

// the maps of the triangular scratch, derived from the gather map (once per pattern); sets p->tri_failed when the pattern does not qualify
int nh_gather_prepare_tri(nh_pattern *p, const nh_basis &test, hipStream_t s) {
  if (p->gsrc_tri || p->tri_failed) return NH_OK;
  NH_REQUIRE(p->gsrc && p->gptr && p->grow, "nh_gather_prepare_tri: the gather map comes first");
  if (!(p->nrows == p->ncols && p->nbt == p->nbr && p->emap_len < (1ll << 31) && p->nnz < (1ll << 31))) {
    p->tri_failed = 1;
    return NH_OK;
  }
  unsigned *canon = nullptr;
  int *cnt = nullptr, *bad = nullptr;
  i64 ndof = 0;
  int hbad = 0, rc = NH_OK;
  hipError_t e = hipSuccess;
  const BasisK tk = to_k(test);
  if (test.off_dev) {
    e = hipMemcpyAsync(&ndof, test.off_dev + p->nelems, sizeof(i64), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  } else
    ndof = p->nelems * (i64)test.nb;
  if (e == hipSuccess) e = hipMalloc((void **)&p->tri_rank, std::max<i64>(ndof, 1));
  if (e == hipSuccess) e = hipMalloc((void **)&cnt, (p->nelems + 1) * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void **)&p->tri_base, (p->nelems + 1) * sizeof(i64));
  if (e == hipSuccess) e = hipMalloc((void **)&canon, std::max<i64>(p->emap_len, 1) * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&p->gsrc_tri, std::max<i64>(p->emap_len, 1) * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&p->gmirror, std::max<i64>(p->nnz, 1) * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&bad, sizeof(int));
  if (e == hipSuccess) e = hipMemsetAsync(bad, 0, sizeof(int), s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_tri_rank, dim3((unsigned)std::min<i64>(p->nelems, 1 << 20)), dim3(64), 0, s, p->nelems, tk, p->tri_rank, cnt);
    rc = nh_scan_exclusive(cnt, p->tri_base, p->nelems, s);
  }
  if (e == hipSuccess && rc == NH_OK) {
    hipLaunchKernelGGL(k_tri_canon, dim3((unsigned)std::min<i64>(p->nelems, 1 << 20)), dim3(64), 0, s, p->nelems, tk, p->nbr, p->eoff, p->tri_rank, p->tri_base, canon);
    hipLaunchKernelGGL(k_gather_remap, dim3((unsigned)std::min<i64>((p->emap_len + 255) / 256, 1 << 16)), dim3(256), 0, s, p->emap_len, p->gsrc, canon, p->gsrc_tri);
    const dim3 grid((unsigned)((p->nnz + 255) / 256));
    hipLaunchKernelGGL(k_tri_mirror, grid, dim3(256), 0, s, p->nnz, p->grow, p->srowptr, p->scol, p->gmirror);
    hipLaunchKernelGGL(k_tri_check, grid, dim3(256), 0, s, p->nnz, p->gmirror, bad);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  hipFree(canon), hipFree(cnt), hipFree(bad);
  if (e != hipSuccess || rc != NH_OK || hbad) {
    hipFree(p->tri_rank), hipFree(p->tri_base), hipFree(p->gsrc_tri), hipFree(p->gmirror);
    p->tri_rank = nullptr, p->tri_base = nullptr, p->gsrc_tri = nullptr, p->gmirror = nullptr;
    p->tri_failed = 1;
    if (e != hipSuccess) {
      nh_set_error("nh_gather_prepare_tri failed: %s", hipGetErrorString(e));
      return NH_EHIP;
    }
    return rc;  // (a pattern that is not symmetric: no plan, no error)
  }
  return NH_OK;
}

void nh_fused_free(nh_fused_plan *f) {
  if (!f) return;
  hipFree(f->order), hipFree(f->bptr), hipFree(f->rstart), hipFree(f->epos), hipFree(f->vptr), hipFree(f->vlist), hipFree(f->vrow), hipFree(f->cpos);
  delete f;
}

// the owner blocks of a pattern (once per pattern handle); NH_ELIMIT: the plan does not fit this pattern, the caller keeps its other paths
static int nh_fused_prepare(nh_pattern *p, const nh_matrix_args *a, hipStream_t s) {
  const i64 ne = p->nelems, nrows = p->nrows;
  const int nbt = p->nbt, nbr = p->nbr;
  if (!(ne < (1ll << 31) && nrows < (1ll << 31) && p->emap_len == ne * nbt * nbr)) return NH_ELIMIT;
  const int32_t *dofs = a->test.dofs_dev;
  BpTmp t;
  t.n = 0;
  unsigned *order = nullptr, *skeys = nullptr, *vlist = nullptr, *pkey = nullptr, *pval = nullptr, *pkey2 = nullptr, *pval2 = nullptr;
  int32_t *rank = nullptr, *blk = nullptr, *rlen = nullptr;
  i64 *rstart = nullptr, *bptr = nullptr, *vptr = nullptr, *epos = nullptr;
  uint16_t *vrow = nullptr;
  uint8_t *cpos = nullptr;
  int *flags = nullptr;  // [0] a turn / column position does not fit its field, [1] visitors of the busiest row, [2] entries of the fullest block
  int maxlen = 0, hflags[3] = {0, 0, 0}, nblocks = 0, R = 0, rc = NH_OK;
  i64 nvisits = 0;
  nh_fused_plan *f = nullptr;
#define FP_CHECK(expr)                                                                          \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      nh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
      rc = NH_EHIP;                                                                             \
      goto done;                                                                                \
    }                                                                                           \
  } while (0)
  {
    if ((rc = bp_cluster(t, p, a, &order, &rank, &maxlen, s, &skeys)) != NH_OK) goto done;
    FP_CHECK(bp_alloc(t, &flags, 3));
    FP_CHECK(hipMemsetAsync(flags, 0, 3 * sizeof(int), s));
    FP_CHECK(bp_alloc(t, &rstart, (size_t)nrows));
    FP_CHECK(bp_alloc(t, &rlen, (size_t)nrows + 1));
    FP_CHECK(bp_alloc(t, &epos, (size_t)nrows + 1));
    hipLaunchKernelGGL(k_fs_rowinfo, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s, nrows, order, p->srowptr, rstart, rlen);
    FP_CHECK(hipGetLastError());
    if ((rc = nh_scan_exclusive(rlen, epos, nrows, s)) != NH_OK) goto done;  // entries in front of a row, rows in rank order
    // rows per block: the largest candidate whose fullest block fits the LDS of a workgroup (two per CU)
    const int cand[] = {512, 256, 128, 64};
    int forced = 0;
    if (getenv("NH_FUSED_ROWS")) forced = std::min(512, std::max(16, atoi(getenv("NH_FUSED_ROWS"))));
    for (int ci = 0; ci < 4 && !R; ++ci) {
      const int Rc = forced ? forced : cand[ci];
      const int tn = t.n;
      FP_CHECK(hipMemsetAsync(flags + 2, 0, sizeof(int), s));
      if ((rc = bp_blocks(t, skeys, nrows, Rc, &nblocks, &bptr, &blk, s)) != NH_OK) goto done;
      hipLaunchKernelGGL(k_fs_maxents, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, s, nblocks, bptr, epos, flags + 2);
      FP_CHECK(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
      FP_CHECK(hipStreamSynchronize(s));
      if (fused_lds_bytes(Rc, hflags[2]) <= FUSED_LDS || forced) R = Rc;
      else {
        for (int i = tn; i < t.n; ++i) hipFree(t.ptr[i]);  // (this candidate's arrays)
        t.n = tn;
      }
    }
    if (!R) {  // (rows too long for a useful block)
      rc = NH_ELIMIT;
      goto done;
    }
    if ((rc = bp_visits(t, p, dofs, rank, blk, nblocks, &nvisits, &vptr, &vlist, s)) != NH_OK) goto done;
    const i64 np = nvisits * nbt;
    if (np >= (1ll << 32)) {  // (32-bit pair indices)
      rc = NH_ELIMIT;
      goto done;
    }
    // the turn of every (visit, local row) within its row: stable sort of the pairs by row
    FP_CHECK(bp_alloc(t, &pkey, (size_t)np));
    FP_CHECK(bp_alloc(t, &pval, (size_t)np));
    hipLaunchKernelGGL(k_fs_vkeys, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, nvisits, nbt, blk, dofs, rank, vptr, vlist, pkey, pval);
    FP_CHECK(hipGetLastError());
    if ((rc = bp_sort_pairs(t, pkey, pval, (size_t)np, 32, &pkey2, &pval2, s)) != NH_OK) goto done;
    FP_CHECK(bp_alloc(t, &vrow, (size_t)np));
    hipLaunchKernelGGL(k_fs_vrow, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, np, bptr, blk, pkey2, pval2, vrow, flags);
    FP_CHECK(bp_alloc(t, &cpos, (size_t)p->emap_len));
    hipLaunchKernelGGL(k_fs_cpos, dim3((unsigned)((p->emap_len + 255) / 256)), dim3(256), 0, s, p->emap_len, p->emap, cpos, flags);
    FP_CHECK(hipGetLastError());
    FP_CHECK(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    FP_CHECK(hipStreamSynchronize(s));
    if (hflags[0] || R > 512) {
      rc = NH_ELIMIT;
      goto done;
    }
    f = new nh_fused_plan();
    memset(f, 0, sizeof *f);
    f->p1hex = -1;
    f->nblocks = nblocks, f->rows_per_block = R, f->max_ents = hflags[2], f->nturns = hflags[1], f->nvisits = nvisits;
    f->order = reinterpret_cast<int32_t *>(bp_keep(t, order));
    f->bptr = bp_keep(t, bptr), f->rstart = bp_keep(t, rstart), f->epos = bp_keep(t, epos);
    f->vptr = bp_keep(t, vptr), f->vlist = reinterpret_cast<int32_t *>(bp_keep(t, vlist)), f->vrow = bp_keep(t, vrow), f->cpos = bp_keep(t, cpos);
  }
done:
#undef FP_CHECK
  bp_free(t);
  if (rc == NH_OK) p->fused = f;
  return rc;
}

// Are these the tables of the trilinear 'std' basis at the tensor 2 x 2 x 2 Gauss points (nodes a = 4 a0 + 2 a1 + a2, points q = 4 qa + 2 qb + qc), for test, trial and
// geometry, and is the form kappa grad.grad + mass phi phi?  Then fill the tables of the sum-factorised routine.  Host-side check of 3 x 2 kB of tables.
static bool fused_p1hex_tables(const nh_matrix_args *a, P1Tab *tab, bool *mass, hipStream_t s) {
  if (a->ndims != 3 || a->nq != 8 || a->test.nb != 8 || a->trial.nb != 8 || a->test.tab_dev || a->trial.tab_dev) return false;
  if (a->geom.kind != NH_GEOM_ISO || a->geom.ngb != 8 || a->geom.bnd_axis >= 0) return false;
  const double *C = a->C_host;  // [a][b], 4 x 4
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (i != j && C[i * 4 + j] != 0.) return false;
  if (C[5] != C[10] || C[5] != C[15]) return false;
  double T[3][8 * 8 * 4], w[8];
  const double *src[3] = {a->test.T_dev, a->trial.T_dev, a->geom.gT_dev};
  for (int t = 0; t < 3; ++t)
    if (hipMemcpyAsync(T[t], src[t], sizeof T[t], hipMemcpyDeviceToHost, s) != hipSuccess) return false;
  if (hipMemcpyAsync(w, a->weights_dev, sizeof w, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return false;
  // 1-D points and weights from the table of the first basis: x_axis(q) = sum of the functions with a_axis = 1 (partition of unity)
  double x[3][8], gx[2], gw[2];
  for (int q = 0; q < 8; ++q)
    for (int ax = 0; ax < 3; ++ax) {
      x[ax][q] = 0.;
      for (int n = 0; n < 8; ++n)
        if ((n >> (2 - ax)) & 1) x[ax][q] += T[0][(n * 8 + q) * 4];
    }
  gx[0] = x[0][0], gx[1] = x[0][4];
  gw[0] = w[0] + w[1] + w[2] + w[3], gw[1] = w[4] + w[5] + w[6] + w[7];
  const double tol = 1e-14;
  for (int q = 0; q < 8; ++q) {
    const int qi[3] = {q >> 2, (q >> 1) & 1, q & 1};
    for (int ax = 0; ax < 3; ++ax)
      if (fabs(x[ax][q] - gx[qi[ax]]) > tol) return false;
    if (fabs(w[q] - gw[qi[0]] * gw[qi[1]] * gw[qi[2]]) > tol) return false;
    for (int t = 0; t < 3; ++t)
      for (int n = 0; n < 8; ++n) {
        const int ni[3] = {n >> 2, (n >> 1) & 1, n & 1};
        double f[3], sg[3];
        for (int ax = 0; ax < 3; ++ax) {
          f[ax] = ni[ax] ? gx[qi[ax]] : 1. - gx[qi[ax]];
          sg[ax] = ni[ax] ? 1. : -1.;
        }
        const double *v = T[t] + (n * 8 + q) * 4;
        if (fabs(v[0] - f[0] * f[1] * f[2]) > tol || fabs(v[1] - sg[0] * f[1] * f[2]) > tol || fabs(v[2] - f[0] * sg[1] * f[2]) > tol || fabs(v[3] - f[0] * f[1] * sg[2]) > tol) return false;
      }
  }
  for (int q = 0; q < 2; ++q) {
    tab->n[0][q] = 1. - gx[q];
    tab->n[1][q] = gx[q];
    tab->c[0][q] = tab->n[0][q] * tab->n[0][q];
    tab->c[1][q] = tab->n[0][q] * tab->n[1][q];
    tab->c[2][q] = tab->n[1][q] * tab->n[1][q];
  }
  for (int qa = 0; qa < 2; ++qa)
    for (int qb = 0; qb < 2; ++qb)
      for (int qc = 0; qc < 2; ++qc) {
        tab->wk[qa][qb][qc] = C[5] * gw[qa] * gw[qb] * gw[qc];
        tab->wm[qa][qb][qc] = C[0] * gw[qa] * gw[qb] * gw[qc];
      }
  *mass = C[0] != 0.;
  return true;
}

int nh_fused_scalar(const nh_matrix_args *a, bool *done, hipStream_t s) {
  *done = false;
  if (a->nct != 1 || a->ncr != 1 || a->cq_dev || a->test.off_dev || a->trial.off_dev || !a->test.nb || !a->trial.nb || a->elist_dev) return NH_OK;
  if (a->test.dofs_dev != a->trial.dofs_dev) return NH_OK;  // (the pattern's column positions are taken per test row; one dof array keeps the plan simple)
  nh_pattern *pat = const_cast<nh_pattern *>(a->pattern);
  if (!pat || pat->nelems != a->nelems || pat->eoff || pat->fused_failed || pat->nbt != a->test.nb || pat->nbr != a->trial.nb) return NH_OK;
  const int key = a->ndims * 10000 + a->test.nb * 100 + a->trial.nb;
  switch (key) {
    case 10202: case 10303: case 20303: case 20404: case 20909: case 30404: case 30808: break;
    default: return NH_OK;
  }
  if (!pat->fused) {
    const int rc = nh_fused_prepare(pat, a, s);
    if (rc == NH_ELIMIT) {
      pat->fused_failed = 1;
      return NH_OK;
    }
    if (rc != NH_OK) return rc;
  }
  const nh_fused_plan *f = pat->fused;
  const int S = 1 + a->ndims;
  FusK p;
  LocK &l = p.loc;
  l.nelems = a->nelems;
  l.elist = nullptr;
  l.nq = a->nq;
  l.weights = a->weights_dev;
  l.geom = to_k(a->geom);
  l.geom.nograd = !uses_gradients(a->C_host, a->nct, 1 + a->ndims, a->ncr);
  l.test = to_k(a->test);
  l.trial = to_k(a->trial);
  l.scale = a->scale_dev;
  l.by_elem = 1;
  for (int i = 0; i < 16; ++i) l.C[i] = i < S * S ? a->C_host[i] : 0.;
  l.local = nullptr;
  l.debug = 0;
  const size_t ldsb = sizeof(double) * (size_t)a->nq * S * (a->test.nb + a->trial.nb + (1 << a->ndims));
  const bool ldst = !a->test.tab_dev && !a->trial.tab_dev && ldsb <= 32 * 1024;
  bool symd = a->test.T_dev == a->trial.T_dev && a->test.tab_dev == a->trial.tab_dev;
  for (int i = 0; i < S * S; ++i)
    if (i / S != i % S && a->C_host[i] != 0.) symd = false;
  l.ldst_doubles = ldst ? (int)(ldsb / sizeof(double)) : 0;
  p.values = a->values_dev;
  p.store = (a->flags & NH_MATRIX_STORE) != 0;
  p.R = f->rows_per_block;
  p.max_ents = f->max_ents, p.nturns = f->nturns, p.epos = f->epos;
  p.bptr = f->bptr, p.vptr = f->vptr, p.rstart = f->rstart, p.vlist = f->vlist, p.vrow = f->vrow, p.cpos = f->cpos;
  const size_t lds1 = fused_lds_bytes(f->rows_per_block, f->max_ents);  // row tables + accumulator
  const size_t ldsx = (ldst ? ldsb : 0) + lds1;
  int nt = FUSED_NT;
  if (getenv("NH_FUSED_NT")) nt = std::min(FUSED_NT_MAX, std::max(64, atoi(getenv("NH_FUSED_NT")) & ~63));
  dim3 grid((unsigned)f->nblocks), block(nt);
  // trilinear hexahedra with the 2 x 2 x 2 Gauss scheme: the sum-factorised element routine (the answer for a set of table pointers is remembered in the plan)
  {
    nh_fused_plan *fw = pat->fused;
    if (fw->p1hex_key[0] != a->test.T_dev || fw->p1hex_key[1] != a->trial.T_dev || fw->p1hex_key[2] != a->geom.gT_dev || fw->p1hex_key[3] != a->weights_dev ||
        (a->ndims == 3 && memcmp(fw->p1hex_C, a->C_host, 16 * sizeof(double)) != 0) || fw->p1hex < 0) {  // (C_host holds (1 + ndims)^2 doubles: compared for 3-D forms only)
      bool mass = false;
      P1Tab tab;
      memset(&tab, 0, sizeof tab);
      const bool ok = !getenv("NUTILS_AMD_NO_FUSED_P1HEX") && a->ndims == 3 && fused_p1hex_tables(a, &tab, &mass, s);
      fw->p1hex = ok ? (mass ? 2 : 1) : 0;
      fw->p1hex_key[0] = a->test.T_dev, fw->p1hex_key[1] = a->trial.T_dev, fw->p1hex_key[2] = a->geom.gT_dev, fw->p1hex_key[3] = a->weights_dev;
      if (a->ndims == 3) memcpy(fw->p1hex_C, a->C_host, 16 * sizeof(double));
      static_assert(sizeof(P1Tab) <= sizeof(fw->p1hex_tab), "plan storage of the sum-factorised tables");
      memcpy(fw->p1hex_tab, &tab, sizeof tab);
    }
    if (fw->p1hex > 0) {
      memcpy(&p.tab, fw->p1hex_tab, sizeof p.tab);
      l.ldst_doubles = 0;
      if (fw->p1hex == 2) {
        NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_fused_p1hex<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL((k_fused_p1hex<true>), grid, block, lds1, s, p);
      } else {
        NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_fused_p1hex<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL((k_fused_p1hex<false>), grid, block, lds1, s, p);
      }
      NH_LAUNCH_CHECK();
      *done = true;
      return NH_OK;
    }
  }
#define FUS2(ND, NBT, NBR, LD, SY)                                                                                                        \
  do {                                                                                                                                    \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_fused_scalar<ND, NBT, NBR, LD, SY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx)); \
    hipLaunchKernelGGL((k_fused_scalar<ND, NBT, NBR, LD, SY>), grid, block, ldsx, s, p);                                                  \
  } while (0)
#define FUS(ND, NBT, NBR)                                   \
  do {                                                      \
    if (NBT == NBR && symd) {                               \
      if (ldst) FUS2(ND, NBT, NBR, true, NBT == NBR);       \
      else FUS2(ND, NBT, NBR, false, NBT == NBR);           \
    } else if (ldst) FUS2(ND, NBT, NBR, true, false);       \
    else FUS2(ND, NBT, NBR, false, false);                  \
  } while (0)
  switch (key) {
    case 10202: FUS(1, 2, 2); break;
    case 10303: FUS(1, 3, 3); break;
    case 20303: FUS(2, 3, 3); break;
    case 20404: FUS(2, 4, 4); break;
    case 20909: FUS(2, 9, 9); break;
    case 30404: FUS(3, 4, 4); break;
    case 30808: FUS(3, 8, 8); break;
  }
#undef FUS
#undef FUS2
  NH_LAUNCH_CHECK();
  *done = true;
  return NH_OK;
}

// thread-per-element pass 1 for scalar forms on uniform bases of the instantiated sizes; *done = false: the caller runs the generic kernel
int nh_local_scalar(const nh_matrix_args *a, double *local, bool *done, hipStream_t s) {
  *done = false;
  if (a->nct != 1 || a->ncr != 1 || a->cq_dev || a->test.off_dev || a->trial.off_dev || !a->test.nb || !a->trial.nb) return NH_OK;
  const int S = 1 + a->ndims;
  LocK p;
  p.nelems = a->nelems;
  p.elist = a->elist_dev;
  p.nq = a->nq;
  p.weights = a->weights_dev;
  p.geom = to_k(a->geom);
  p.geom.nograd = !uses_gradients(a->C_host, a->nct, 1 + a->ndims, a->ncr);
  p.test = to_k(a->test);
  p.trial = to_k(a->trial);
  p.scale = a->scale_dev;
  p.by_elem = (a->flags & NH_MATRIX_EMAP_BY_ELEMENT) != 0;
  for (int i = 0; i < 16; ++i) p.C[i] = i < S * S ? a->C_host[i] : 0.;
  p.local = local;
  p.debug = 0;
#ifdef NH_ABLATION
  if (getenv("NH_LOCAL_DEBUG")) p.debug = atoi(getenv("NH_LOCAL_DEBUG"));
#endif
  dim3 grid((unsigned)((a->nelems + 127) / 128)), block(128);
  const int key = a->ndims * 10000 + a->test.nb * 100 + a->trial.nb;
  const size_t ldsb = sizeof(double) * (size_t)a->nq * S * (a->test.nb + a->trial.nb + (1 << a->ndims));
  const bool rows = key == 21616 || key == 32727 || key == 22525 || key == 36464;  // (row-blocked kernel: at most two workgroups per CU by registers, LDS is free)
  const bool ldst = !a->test.tab_dev && !a->trial.tab_dev && ldsb <= (rows ? 64 : 32) * 1024;
  bool symd = a->test.T_dev == a->trial.T_dev && a->test.tab_dev == a->trial.tab_dev && a->test.dofs_dev == a->trial.dofs_dev;
  for (int i = 0; i < S * S; ++i)
    if (i / S != i % S && a->C_host[i] != 0.) symd = false;
  const size_t ldsg = sizeof(double) * (size_t)a->nq * S * (1 << a->ndims);  // geometry tables: staged by the row-split kernel also when the basis tables are not
  p.ldst_doubles = ldst ? (int)(ldsb / sizeof(double)) : rows ? (int)(ldsg / sizeof(double)) : 0;
  const size_t ldsx = (ldst ? ldsb : rows ? ldsg : 0) + sizeof(double) * 2 * 64 * 17;  // + two waves' transposition buffers
#define LOC(ND, NBT, NBR)                                                                                         \
  do {                                                                                                            \
    if (NBT == NBR && symd) {                                                                                     \
      if (ldst) hipLaunchKernelGGL((k_local_scalar<ND, NBT, NBR, true, NBT == NBR>), grid, block, ldsx, s, p);    \
      else hipLaunchKernelGGL((k_local_scalar<ND, NBT, NBR, false, NBT == NBR>), grid, block, ldsx, s, p);        \
    } else if (ldst) hipLaunchKernelGGL((k_local_scalar<ND, NBT, NBR, true, false>), grid, block, ldsx, s, p);    \
    else hipLaunchKernelGGL((k_local_scalar<ND, NBT, NBR, false, false>), grid, block, ldsx, s, p);               \
  } while (0)
#define ROWS(ND, NBT, NBR, MB)                                                                                              \
  do {                                                                                                                     \
    dim3 grid2((unsigned)((a->nelems * (NBT / MB) + 127) / 128));                                                         \
    if (ldst) {                                                                                                            \
      NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_local_rows<ND, NBT, NBR, MB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx)); \
      hipLaunchKernelGGL((k_local_rows<ND, NBT, NBR, MB, true>), grid2, block, ldsx, s, p);                                \
    } else                                                                                                                 \
      hipLaunchKernelGGL((k_local_rows<ND, NBT, NBR, MB, false>), grid2, block, ldsx, s, p);                               \
  } while (0)
  switch (key) {
    case 21616: ROWS(2, 16, 16, 4); break;  // bicubic tensor splines
    case 32727: ROWS(3, 27, 27, 3); break;  // triquadratic hexahedra / splines
    case 22525: ROWS(2, 25, 25, 5); break;  // biquartic
    case 36464: ROWS(3, 64, 64, 1); break;  // tricubic: one row per thread (the one-wave-per-element kernel needs 64 kB of LDS per wave: 38 ms for 32^3)
    case 10202: LOC(1, 2, 2); break;
    case 10303: LOC(1, 3, 3); break;
    case 20303: LOC(2, 3, 3); break;
    case 20404: LOC(2, 4, 4); break;
    case 20909: LOC(2, 9, 9); break;
    case 30404: LOC(3, 4, 4); break;
    case 30808: LOC(3, 8, 8); break;
    default: return NH_OK;
  }
#undef LOC
#undef ROWS
  NH_LAUNCH_CHECK();
  *done = true;
  return NH_OK;
}

// thread pass for vector-valued blocks (nct == ncr == NC components on ONE uniform basis, test == trial tables); *done = false: not applicable
int nh_local_vector(const nh_matrix_args *a, double *local, bool *done, hipStream_t s) {
  *done = false;
  const int nc = a->nct;
  if (nc < 2 || nc > 3 || a->ncr != nc || a->cq_dev || a->test.off_dev || a->trial.off_dev || !a->test.nb || a->test.nb != a->trial.nb) return NH_OK;
  if (a->test.T_dev != a->trial.T_dev || a->test.tab_dev != a->trial.tab_dev || a->test.dofs_dev != a->trial.dofs_dev) return NH_OK;
  const int S = 1 + a->ndims;
  LocVK p;
  p.nelems = a->nelems;
  p.elist = a->elist_dev;
  p.nq = a->nq;
  p.weights = a->weights_dev;
  p.geom = to_k(a->geom);
  p.geom.nograd = !uses_gradients(a->C_host, nc, S, nc);
  p.test = to_k(a->test);
  p.scale = a->scale_dev;
  p.by_elem = (a->flags & NH_MATRIX_EMAP_BY_ELEMENT) != 0;
  for (int i = 0; i < 144; ++i) p.C[i] = i < nc * S * nc * S ? a->C_host[i] : 0.;
  for (int c = 0; c < 3; ++c)
    for (int d = 0; d < 3; ++d) p.mask[c][d] = c < nc && d < nc && (!a->mask_host || a->mask_host[c * nc + d]);
  p.local = local;
  dim3 grid((unsigned)((a->nelems * a->test.nb + 127) / 128)), block(128);
  const size_t ldsb = sizeof(double) * (size_t)a->nq * S * (a->test.nb + (1 << a->ndims));
  const bool ldst = !a->test.tab_dev && ldsb <= 64 * 1024;
  p.ldst_doubles = ldst ? (int)(ldsb / sizeof(double)) : 0;
  const size_t ldsx = sizeof(double) * 144 + (ldst ? ldsb : 0) + sizeof(double) * 2 * 64 * 17 + sizeof(double) * 128 * 3 +  // + vertices of the elements of the workgroup
                      sizeof(double) * (128 / a->test.nb + 1) * a->nq * (a->ndims * a->ndims + 1);                      // + inverse Jacobians and weights of their points
  // the isotropic three-parameter family on the gradient slots (all blocks coupled, one component per axis)?
  bool isof = nc == a->ndims && !getenv("NUTILS_AMD_NO_ISOFORM");
  {
    const double *C = a->C_host;
    auto at = [&](int c, int sa, int d, int sb) { return C[((c * S + sa) * nc + d) * S + sb]; };
    const double lam = nc > 1 ? at(0, 1, 1, 2) : 0., mu2 = nc > 1 ? at(0, 2, 1, 1) : 0., mu = nc > 1 ? at(0, 2, 0, 2) : 0.;
    for (int c = 0; c < nc && isof; ++c)
      for (int sa = 0; sa < S && isof; ++sa)
        for (int d = 0; d < nc && isof; ++d)
          for (int sb = 0; sb < S && isof; ++sb) {
            const double expect = (sa && sb) ? lam * (c == sa - 1 && d == sb - 1) + mu * (c == d && sa == sb) + mu2 * (c == sb - 1 && sa - 1 == d) : 0.;
            if (at(c, sa, d, sb) != expect || !p.mask[c][d]) isof = false;
          }
    p.lam = lam, p.mu = mu, p.mu2 = mu2;
  }
#define ROWSV(ND, NB, NC)                                                                                       \
  do {                                                                                                          \
    if (isof) {                                                                                                 \
      if (ldst) hipLaunchKernelGGL((k_local_rows_v<ND, NB, NC, true, true>), grid, block, ldsx, s, p);          \
      else hipLaunchKernelGGL((k_local_rows_v<ND, NB, NC, false, true>), grid, block, ldsx, s, p);              \
    } else if (ldst) hipLaunchKernelGGL((k_local_rows_v<ND, NB, NC, true, false>), grid, block, ldsx, s, p);    \
    else hipLaunchKernelGGL((k_local_rows_v<ND, NB, NC, false, false>), grid, block, ldsx, s, p);               \
  } while (0)
  switch (a->ndims * 1000 + a->test.nb * 10 + nc) {
    case 3083: ROWSV(3, 8, 3); break;  // trilinear hexahedra, 3 components
    case 2042: ROWSV(2, 4, 2); break;  // bilinear quadrilaterals, 2 components
    case 2092: ROWSV(2, 9, 2); break;  // biquadratic quadrilaterals / quadratic splines, 2 components
    default: return NH_OK;
  }
#undef ROWSV
  NH_LAUNCH_CHECK();
  *done = true;
  return NH_OK;
}
