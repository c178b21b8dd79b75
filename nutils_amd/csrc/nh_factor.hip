// Sparse Taylor coefficient tensors of rank >= 3 built ON THE DEVICE: the one-time element loop + dedup of evaluable.factor
// (/root/reference/src/nutils/evaluable.py:5785-5874: `m_coeffs.append(zeroed.assparse)`, then values / indices pruned of zeros) for the terms
//     c int s(x) u^k dV,   k = 3, 4   ->   C_k[i1 .. ik] = c int s N_i1 ... N_ik dV
// (value polynomials of one scalar field: the double-well potential of examples/cahnhilliard.py:175, the cubic functional of
// tests/golden/factor_cubic2d_spline2_4.npz).  Layout of the result as the reference's Monomial holds it (evaluable.py:5693-5751): values[nnz] with one
// index array per tensor axis, entries sorted by the flat key (i1, .., ik) lexicographically, every permutation stored, zeros pruned.
//   1. k_fac_wdet:    w_q |det J_q| s_q per (element, point)
//   2. k_fac_moments: one thread per (element, local multi-index): the moment and its flat key
//   3. radix sort of (key, value) pairs (stable: equal keys stay in element order, the order in which the reference's bincount adds them)
//   4. run heads -> exclusive scan -> one thread per distinct key sums its run; zeros pruned with a second scan; keys decoded to index arrays
// The per-step work on these tensors is nh_monomial (nh_monomial.hip).
#include "nh_common.h"
#include <rocprim/rocprim.hpp>

namespace {

#include "nh_geom.inc"

__device__ __forceinline__ i64 bfn(const BasisK &b, i64 e) { return b.tab ? (i64)b.tab[e] * b.nb : 0; }

template <int ND>
__global__ void k_fac_wdet(i64 nelems, const int32_t *elist, int nq, const double *weights, GeomK g, const double *scale, double coeff, double *wdet) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nelems * nq) return;
  const i64 ie = t / nq;
  const int q = (int)(t - ie * nq);
  const i64 e = elist ? elist[ie] : ie;
  double Ji[ND][ND], det;
  geometry_at<ND>(g, e, q, nq, nullptr, Ji, det, nullptr);
  wdet[t] = coeff * weights[q] * fabs(det) * (scale ? scale[t] : 1.);
}

template <int RANK>
__global__ void k_fac_moments(i64 nelems, const int32_t *elist, int nq, int S, BasisK b, const double *wdet, i64 ndofs, i64 per, unsigned long long *keys, double *vals) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nelems * per) return;
  const i64 ie = t / per;
  i64 r = t - ie * per;
  const i64 e = elist ? elist[ie] : ie;
  int loc[RANK];
#pragma unroll
  for (int a = RANK - 1; a >= 0; --a) {
    loc[a] = (int)(r % b.nb);
    r /= b.nb;
  }
  const double *T = b.T + bfn(b, e) * nq * S;
  const double *w = wdet + ie * nq;
  double sum = 0;
  for (int q = 0; q < nq; ++q) {
    double prod = w[q];
#pragma unroll
    for (int a = 0; a < RANK; ++a) prod *= T[((size_t)loc[a] * nq + q) * S];
    sum += prod;
  }
  unsigned long long key = 0;
#pragma unroll
  for (int a = 0; a < RANK; ++a) key = key * (unsigned long long)ndofs + (unsigned long long)b.dofs[e * (i64)b.nb + loc[a]];
  keys[t] = key;
  vals[t] = sum;
}

__global__ void k_fac_heads(i64 n, const unsigned long long *keys, int32_t *head) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = i == 0 || keys[i] != keys[i - 1];
}

// one thread per run: the sum of its values in sorted (= element) order; nz: the sum is not zero
__global__ void k_fac_runs(i64 n, const unsigned long long *keys, const double *vals, const int32_t *head, const i64 *pos, unsigned long long *ukeys, double *uvals, int32_t *nz) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  double sum = vals[i];
  for (i64 j = i + 1; j < n && !head[j]; ++j) sum += vals[j];
  ukeys[pos[i]] = keys[i];
  uvals[pos[i]] = sum;
  nz[pos[i]] = sum != 0.;
}

template <int RANK>
__global__ void k_fac_compact(i64 n, const unsigned long long *ukeys, const double *uvals, const int32_t *nz, const i64 *pos, i64 ndofs, i64 nnz, double *values, i64 *indices) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !nz[i]) return;
  const i64 o = pos[i];
  values[o] = uvals[i];
  unsigned long long key = ukeys[i];
#pragma unroll
  for (int a = RANK - 1; a >= 0; --a) {
    indices[a * nnz + o] = (i64)(key % (unsigned long long)ndofs);
    key /= (unsigned long long)ndofs;
  }
}

// result of the last build, owned by the library until it is fetched (one build at a time: the single-stream contract of the scratch buffers)
struct FacResult {
  unsigned long long *ukeys = nullptr;
  double *uvals = nullptr;
  int32_t *nz = nullptr;
  i64 *pos = nullptr;
  i64 nuniq = 0, nnz = 0, ndofs = 0;
  int rank = 0;
} g_res;

void fac_release() {
  hipFree(g_res.ukeys), hipFree(g_res.uvals), hipFree(g_res.nz), hipFree(g_res.pos);
  g_res = FacResult();
}

}  // namespace

extern "C" int nh_factor_tensor(const nh_factor_args *a, int64_t *nnz_out, void *stream) {
  NH_REQUIRE(a && nnz_out, "nh_factor_tensor: NULL argument");
  NH_REQUIRE(a->rank == 3 || a->rank == 4, "nh_factor_tensor: rank must be 3 or 4 (got %d)", a->rank);
  NH_REQUIRE(a->ndims >= 1 && a->ndims <= 3 && a->nq >= 1 && a->weights_dev, "nh_factor_tensor: quadrature / dimension");
  NH_REQUIRE(a->basis.nb > 0 && !a->basis.off_dev && a->basis.T_dev && a->basis.dofs_dev, "nh_factor_tensor: a basis with a uniform number of functions per element is required");
  NH_REQUIRE(a->ndofs > 0, "nh_factor_tensor: ndofs");
  {
    long double cap = 1;
    for (int i = 0; i < a->rank; ++i) cap *= (long double)a->ndofs;
    NH_REQUIRE(cap < 9.0e18L, "nh_factor_tensor: %lld dofs to the power %d does not fit the 64-bit key", (long long)a->ndofs, a->rank);
  }
  hipStream_t s = nh_stream(stream);
  fac_release();
  *nnz_out = 0;
  if (a->nelems == 0) return NH_OK;
  i64 per = 1;
  for (int i = 0; i < a->rank; ++i) per *= a->basis.nb;
  const i64 n = a->nelems * per;
  NH_REQUIRE(n < (1ll << 40), "nh_factor_tensor: %lld local entries", (long long)n);
  const int S = 1 + a->ndims;
  double *wdet = nullptr, *vals = nullptr, *vals2 = nullptr;
  unsigned long long *keys = nullptr, *keys2 = nullptr;
  int32_t *head = nullptr;
  i64 *pos = nullptr;
  void *tmp = nullptr;
  int rc = NH_OK;
#define FC(expr)                                                                                 \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      nh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
      rc = NH_EHIP;                                                                              \
      goto done;                                                                                 \
    }                                                                                            \
  } while (0)
  {
    FC(hipMalloc((void **)&wdet, sizeof(double) * a->nelems * a->nq));
    const GeomK g = to_k(a->geom);
    const unsigned gw = (unsigned)((a->nelems * a->nq + 255) / 256);
    if (a->ndims == 1) hipLaunchKernelGGL(k_fac_wdet<1>, dim3(gw), dim3(256), 0, s, (i64)a->nelems, a->elist_dev, a->nq, a->weights_dev, g, a->scale_dev, a->coeff, wdet);
    if (a->ndims == 2) hipLaunchKernelGGL(k_fac_wdet<2>, dim3(gw), dim3(256), 0, s, (i64)a->nelems, a->elist_dev, a->nq, a->weights_dev, g, a->scale_dev, a->coeff, wdet);
    if (a->ndims == 3) hipLaunchKernelGGL(k_fac_wdet<3>, dim3(gw), dim3(256), 0, s, (i64)a->nelems, a->elist_dev, a->nq, a->weights_dev, g, a->scale_dev, a->coeff, wdet);
    FC(hipGetLastError());
    FC(hipMalloc((void **)&keys, sizeof(unsigned long long) * n));
    FC(hipMalloc((void **)&vals, sizeof(double) * n));
    const BasisK b{a->basis.nb, a->basis.T_dev, a->basis.dofs_dev, nullptr, a->basis.tab_dev};
    const unsigned gm = (unsigned)((n + 255) / 256);
    if (a->rank == 3) hipLaunchKernelGGL(k_fac_moments<3>, dim3(gm), dim3(256), 0, s, (i64)a->nelems, a->elist_dev, a->nq, S, b, wdet, (i64)a->ndofs, per, keys, vals);
    else hipLaunchKernelGGL(k_fac_moments<4>, dim3(gm), dim3(256), 0, s, (i64)a->nelems, a->elist_dev, a->nq, S, b, wdet, (i64)a->ndofs, per, keys, vals);
    FC(hipGetLastError());
    FC(hipMalloc((void **)&keys2, sizeof(unsigned long long) * n));
    FC(hipMalloc((void **)&vals2, sizeof(double) * n));
    int bits = 1;
    {
      long double cap = 1;
      for (int i = 0; i < a->rank; ++i) cap *= (long double)a->ndofs;
      while (bits < 64 && (long double)(1ull << bits) < cap) ++bits;
    }
    size_t tmpsz = 0;
    FC(rocprim::radix_sort_pairs(nullptr, tmpsz, keys, keys2, vals, vals2, (size_t)n, 0, bits, s));
    FC(hipMalloc(&tmp, tmpsz));
    FC(rocprim::radix_sort_pairs(tmp, tmpsz, keys, keys2, vals, vals2, (size_t)n, 0, bits, s));
    FC(hipMalloc((void **)&head, sizeof(int32_t) * n));
    FC(hipMalloc((void **)&pos, sizeof(i64) * (n + 1)));
    hipLaunchKernelGGL(k_fac_heads, dim3(gm), dim3(256), 0, s, n, keys2, head);
    FC(hipGetLastError());
    if ((rc = nh_scan_exclusive(head, pos, n, s)) != NH_OK) goto done;
    i64 nuniq = 0;
    FC(hipMemcpyAsync(&nuniq, pos + n, sizeof(i64), hipMemcpyDeviceToHost, s));
    FC(hipStreamSynchronize(s));
    FC(hipMalloc((void **)&g_res.ukeys, sizeof(unsigned long long) * nuniq));
    FC(hipMalloc((void **)&g_res.uvals, sizeof(double) * nuniq));
    FC(hipMalloc((void **)&g_res.nz, sizeof(int32_t) * nuniq));
    FC(hipMalloc((void **)&g_res.pos, sizeof(i64) * (nuniq + 1)));
    hipLaunchKernelGGL(k_fac_runs, dim3(gm), dim3(256), 0, s, n, keys2, vals2, head, pos, g_res.ukeys, g_res.uvals, g_res.nz);
    FC(hipGetLastError());
    if ((rc = nh_scan_exclusive(g_res.nz, g_res.pos, nuniq, s)) != NH_OK) goto done;
    i64 nnz = 0;
    FC(hipMemcpyAsync(&nnz, g_res.pos + nuniq, sizeof(i64), hipMemcpyDeviceToHost, s));
    FC(hipStreamSynchronize(s));
    g_res.nuniq = nuniq, g_res.nnz = nnz, g_res.ndofs = a->ndofs, g_res.rank = a->rank;
    *nnz_out = nnz;
  }
done:
#undef FC
  hipFree(wdet), hipFree(keys), hipFree(vals), hipFree(keys2), hipFree(vals2), hipFree(head), hipFree(pos), hipFree(tmp);
  if (rc != NH_OK) fac_release();
  return rc;
}

extern "C" int nh_factor_fetch(double *values_dev, int64_t *indices_dev, void *stream) {
  NH_REQUIRE(g_res.rank, "nh_factor_fetch: no tensor has been built");
  hipStream_t s = nh_stream(stream);
  if (g_res.nnz) {
    NH_REQUIRE(values_dev && indices_dev, "nh_factor_fetch: NULL buffer");
    const unsigned gu = (unsigned)((g_res.nuniq + 255) / 256);
    if (g_res.rank == 3) hipLaunchKernelGGL(k_fac_compact<3>, dim3(gu), dim3(256), 0, s, g_res.nuniq, g_res.ukeys, g_res.uvals, g_res.nz, g_res.pos, g_res.ndofs, g_res.nnz, values_dev, (i64 *)indices_dev);
    else hipLaunchKernelGGL(k_fac_compact<4>, dim3(gu), dim3(256), 0, s, g_res.nuniq, g_res.ukeys, g_res.uvals, g_res.nz, g_res.pos, g_res.ndofs, g_res.nnz, values_dev, (i64 *)indices_dev);
    NH_LAUNCH_CHECK();
    NH_CHECK_HIP(hipStreamSynchronize(s));
  }
  fac_release();
  return NH_OK;
}
