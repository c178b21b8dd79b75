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
    const i64 i1 = (i64)(unsigned)gsrc[i0], i2 = i0 + 1 < e ? (i64)(unsigned)gsrc[i0 + 1] : -1;
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
  return NH_OK;
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
};

template <int ND, int NB, int NC, bool LDST>
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
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double tw[NC][S];
#pragma unroll
      for (int d = 0; d < NC; ++d) {
        if (p.mask[c][d]) {  // (uniform)
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
          for (int s2 = 0; s2 < S; ++s2) A[n][c][d] += tw[d][s2] * tn[s2];
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

int nh_gather_values(const nh_pattern *p, const double *local, i64 ld, const GSlots &slots, double *values, int store, hipStream_t s) {
  if (!p->nnz) return NH_OK;
  if (slots.nct * slots.ncr == 1)
    hipLaunchKernelGGL(k_gather_values, dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, s, p->nnz, p->gptr, p->gsrc, p->grow, p->srowptr, local, ld, p->nbt * p->nbr, slots,
                       values, store);
  else
    hipLaunchKernelGGL(k_gather_values_v, dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, s, p->nnz, p->gptr, p->gsrc, p->grow, p->srowptr, local, slots, values, store);
  NH_LAUNCH_CHECK();
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
#define ROWSV(ND, NB, NC)                                                                                       \
  do {                                                                                                          \
    if (ldst) hipLaunchKernelGGL((k_local_rows_v<ND, NB, NC, true>), grid, block, ldsx, s, p);                  \
    else hipLaunchKernelGGL((k_local_rows_v<ND, NB, NC, false>), grid, block, ldsx, s, p);                      \
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
