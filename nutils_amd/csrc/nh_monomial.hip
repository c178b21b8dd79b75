// Monomial (evaluable.py:5693-5751, helper of evaluable.factor :5785-5874): gather-multiply-scatter over the entries of a
// precomputed sparse coefficient tensor:   out[oidx[i]] += alpha * values[i] * prod_k arg_k[idx_k[i]]
// (the reference: `out = values.copy(); out *= arg[index] ...` followed by Inflate / add.at).  Two forms:
//   * CSR rows as output index (rank-2 tensors from as_csr): one half-wave per row, no atomics, deterministic;
//   * general COO with up to 4 gathered arguments and an optional output index (scalar result if NULL).
#include "nh_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

__global__ void k_monomial_csr(i64 nrows, const i64 *rowptr, const i64 *colidx, const double *values, const double *x, double alpha,
                               double *y) {
  const i64 row = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const int lane = threadIdx.x & 31;
  if (row >= nrows) return;
  double s = 0;
  for (i64 k = rowptr[row] + lane; k < rowptr[row + 1]; k += 32) s += values[k] * x[colidx[k]];
#pragma unroll
  for (int o = 16; o; o >>= 1) s += __shfl_xor(s, o, 32);
  if (lane == 0) y[row] += alpha * s;
}

struct MonoK {
  i64 n;
  const double *values;
  int nargs;
  const double *arg[4];
  const i64 *idx[4];
  const i64 *oidx;
  double alpha;
  double *out;
};

__global__ void k_monomial(MonoK p) {
  double acc = 0;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (i64)gridDim.x * blockDim.x) {
    double v = p.alpha * p.values[i];
    for (int k = 0; k < p.nargs; ++k) v *= p.arg[k][p.idx[k][i]];
    if (p.oidx) atomicAdd(p.out + p.oidx[i], v);
    else acc += v;
  }
  if (!p.oidx) {
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(p.out, acc);
  }
}

// dst[didx[i]] = src[sidx[i]]: the value half of the block merge / submatrix selection.  Consecutive i are consecutive entries of a CSR row
// segment, so the stores of a wave form a few contiguous runs -- also when dst is page-locked host memory and they travel as PCIe writes.
__global__ void k_index_copy(i64 n, const double *src, const i64 *sidx, const i64 *didx, double *dst) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(src[sidx ? sidx[i] : i], dst + (didx ? didx[i] : i));
}

}  // namespace

extern "C" {

int nh_index_copy(int64_t n, const double *src_dev, const int64_t *src_index_dev, const int64_t *dst_index_dev, double *dst, void *stream) {
  NH_REQUIRE(n >= 0 && src_dev && dst, "nh_index_copy: invalid argument");
  if (!n) return NH_OK;
  // host destination: the copy is PCIe bound (64-byte posted writes: ~43 GB/s) and runs beside the next assembly on another stream.  Its stores must not outrun the link: with
  // 512 workgroups in flight the posted writes fill the write queues of the L2 channels and EVERY kernel beside it waits for them (a one-element fill: 206 us; the
  // Cahn-Hilliard Newton step 2.29 ms); 64 workgroups keep the link just as busy (1.22 against 1.15 ms) and leave the queues open (step 2.07 ms; 32: 2.12, 128: 2.10,
  // profiles/r05_c4_c5_traces.md)
  hipPointerAttribute_t attr;
  const bool host = hipPointerGetAttributes(&attr, dst) == hipSuccess && attr.type == hipMemoryTypeHost;
  if (!host) (void)hipGetLastError();
  static const int hostwgs = getenv("NH_INDEX_COPY_WGS") ? atoi(getenv("NH_INDEX_COPY_WGS")) : 64;
  const unsigned grid = (unsigned)std::min<i64>((n + 255) / 256, host ? hostwgs : 256 * 32);
  hipLaunchKernelGGL(k_index_copy, dim3(grid), dim3(256), 0, nh_stream(stream), (i64)n, src_dev, (const i64 *)src_index_dev, (const i64 *)dst_index_dev, dst);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

int nh_monomial_csr(int64_t nrows, const int64_t *rowptr_dev, const int64_t *colidx_dev, const double *values_dev, const double *x_dev,
                    double alpha, double *y_dev, void *stream) {
  NH_REQUIRE(nrows >= 0 && rowptr_dev && colidx_dev && values_dev && x_dev && y_dev, "nh_monomial_csr: invalid argument");
  if (!nrows) return NH_OK;
  hipLaunchKernelGGL(k_monomial_csr, dim3((unsigned)((nrows * 32 + 255) / 256)), dim3(256), 0, nh_stream(stream), (i64)nrows, (const i64 *)rowptr_dev,
                     (const i64 *)colidx_dev, values_dev, x_dev, alpha, y_dev);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

int nh_monomial(int64_t n, const double *values_dev, int nargs, const double *const *args_dev, const int64_t *const *indices_dev,
                const int64_t *out_index_dev, double alpha, double *out_dev, void *stream) {
  NH_REQUIRE(n >= 0 && values_dev && out_dev, "nh_monomial: invalid argument");
  NH_REQUIRE(nargs >= 0 && nargs <= 4, "nh_monomial: at most 4 gathered arguments (got %d)", nargs);
  if (!n) return NH_OK;
  MonoK p;
  p.n = n;
  p.values = values_dev;
  p.nargs = nargs;
  for (int k = 0; k < 4; ++k) {
    p.arg[k] = k < nargs ? args_dev[k] : nullptr;
    p.idx[k] = k < nargs ? (const i64 *)indices_dev[k] : nullptr;
    NH_REQUIRE(k >= nargs || (p.arg[k] && p.idx[k]), "nh_monomial: NULL argument %d", k);
  }
  p.oidx = (const i64 *)out_index_dev;
  p.alpha = alpha;
  p.out = out_dev;
  const unsigned grid = (unsigned)std::min<i64>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(k_monomial, dim3(grid), dim3(256), 0, nh_stream(stream), p);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

}  // extern "C"

// ---- pointwise polynomial of field values at the quadrature points -----------------------------------------------------
// out[i] = sum_t coeff[t] prod_v x_v[i * stride_v]^power[t][v]: the coefficient functions g(phi, phi0, ...) of nonlinear
// integrands (psi'(phi), psi''(phi) of a Cahn-Hilliard free energy, ...) evaluated once per Newton step; the result enters
// the element kernels as scale_dev.  (The reference expands such terms into rank-3/4 sparse tensors with evaluable.factor;
// here the integrand is simply re-integrated with the pointwise coefficient.)
namespace {
struct PolyK {
  i64 n;
  int nvars, nterms;
  const double *x[4];
  int stride[4];
  double coeff[32];
  unsigned char power[32][4];
  double *out;
};

__global__ void k_pointwise_poly(PolyK p) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  double xv[4];
  for (int v = 0; v < p.nvars; ++v) xv[v] = p.x[v][i * p.stride[v]];
  double s = 0;
  for (int t = 0; t < p.nterms; ++t) {
    double m = p.coeff[t];
    for (int v = 0; v < p.nvars; ++v)
      for (int k = 0; k < p.power[t][v]; ++k) m *= xv[v];
    s += m;
  }
  p.out[i] = s;
}
}  // namespace

extern "C" int nh_pointwise_poly(int64_t n, int nvars, const double *const *x_dev, const int *strides, int nterms, const double *coeffs,
                                 const int *powers, double *out_dev, void *stream) {
  NH_REQUIRE(n >= 0 && out_dev, "nh_pointwise_poly: invalid argument");
  NH_REQUIRE(nvars >= 0 && nvars <= 4 && nterms >= 0 && nterms <= 32, "nh_pointwise_poly: at most 4 variables and 32 terms");
  if (!n) return NH_OK;
  PolyK p;
  p.n = n;
  p.nvars = nvars;
  p.nterms = nterms;
  for (int v = 0; v < 4; ++v) {
    p.x[v] = v < nvars ? x_dev[v] : nullptr;
    p.stride[v] = v < nvars ? strides[v] : 0;
    NH_REQUIRE(v >= nvars || p.x[v], "nh_pointwise_poly: NULL variable %d", v);
  }
  for (int t = 0; t < nterms; ++t) {
    p.coeff[t] = coeffs[t];
    for (int v = 0; v < 4; ++v) {
      const int pw = v < nvars ? powers[t * nvars + v] : 0;
      NH_REQUIRE(pw >= 0 && pw < 32, "nh_pointwise_poly: power out of range");
      p.power[t][v] = (unsigned char)pw;
    }
  }
  p.out = out_dev;
  hipLaunchKernelGGL(k_pointwise_poly, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nh_stream(stream), p);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

// ---- per-point forms of field values ------------------------------------------------------------------------------------
// The product-rule coefficients of quasi-linear problems (function.derivative of g(u) B(v, u) or of an energy g(u) B(u, u):
// the reference differentiates the evaluable graph, evaluable.py `derivative` rules of Multiply / Einsum) at every quadrature point,
// from U = (value, gradient) of the bound scalar field(s) there (nh_sample_eval):
//   kind 0:  out[i]       = sc_i * sum_ab B[a][b] Ut[i][a] Ur[i][b]           (point factor of an energy Hessian / residual)
//   kind 1:  out[i][a][b] = sc_i * (b == 0 ? sum_x B[a][x] Ut[i][x] : 0)      (C_q of qform 'trial': result on the value slot)
//   kind 2:  out[i][a][b] = sc_i * L[a] sum_x B[x][b] Ut[i][x]                (C_q of qform 'test')
// sc_i = scale[i] or 1.  S = 1 + ndims <= 4.
namespace {
struct PFormK {
  i64 n;
  int kind, S;
  const double *Ut, *Ur, *scale;
  double B[16], L[4];
  double *out;
};

__global__ void k_point_forms(PFormK p) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  const int S = p.S;
  double ut[4], ur[4];
  for (int a = 0; a < S; ++a) ut[a] = p.Ut[i * S + a];
  const double sc = p.scale ? p.scale[i] : 1.;
  if (p.kind == 0) {
    for (int a = 0; a < S; ++a) ur[a] = p.Ur[i * S + a];
    double s = 0;
    for (int a = 0; a < S; ++a) {
      double t = 0;
      for (int b = 0; b < S; ++b) t += p.B[a * 4 + b] * ur[b];
      s += ut[a] * t;
    }
    p.out[i] = sc * s;
    return;
  }
  double *o = p.out + i * S * S;
  if (p.kind == 1) {
    for (int a = 0; a < S; ++a) {
      double t = 0;
      for (int x = 0; x < S; ++x) t += p.B[a * 4 + x] * ut[x];
      o[a * S] = sc * t;
      for (int b = 1; b < S; ++b) o[a * S + b] = 0.;
    }
  } else {
    double t[4];
    for (int b = 0; b < S; ++b) {
      t[b] = 0;
      for (int x = 0; x < S; ++x) t[b] += p.B[x * 4 + b] * ut[x];
    }
    for (int a = 0; a < S; ++a)
      for (int b = 0; b < S; ++b) o[a * S + b] = sc * p.L[a] * t[b];
  }
}
}  // namespace

extern "C" int nh_point_forms(int kind, int64_t npoints, int S, const double *Ut_dev, const double *Ur_dev, const double *B_host, const double *L_host,
                              const double *scale_dev, double *out_dev, void *stream) {
  NH_REQUIRE(kind >= 0 && kind <= 2 && npoints >= 0 && S >= 2 && S <= 4 && Ut_dev && B_host && out_dev, "nh_point_forms: invalid argument");
  NH_REQUIRE(kind != 0 || Ur_dev, "nh_point_forms: kind 0 needs both fields");
  NH_REQUIRE(kind != 2 || L_host, "nh_point_forms: kind 2 needs L");
  if (!npoints) return NH_OK;
  PFormK p;
  memset(&p, 0, sizeof p);
  p.n = npoints; p.kind = kind; p.S = S;
  p.Ut = Ut_dev; p.Ur = Ur_dev; p.scale = scale_dev; p.out = out_dev;
  for (int a = 0; a < S; ++a) {
    for (int b = 0; b < S; ++b) p.B[a * 4 + b] = B_host[a * S + b];
    if (L_host) p.L[a] = L_host[a];
  }
  hipLaunchKernelGGL(k_point_forms, dim3((unsigned)((npoints + 255) / 256)), dim3(256), 0, nh_stream(stream), p);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

// ---- array-valued expressions of field values at the points of a sample -----------------------------------------------------
// Sample.eval / Sample.bind of the reference (sample.py:192-232; _ConcatenatePoints.lower :966-975: a loop_concatenate of the lowered function over the
// elements) for functions that are sums of (constant coefficient tensor) x (product of values / gradients of bound fields and coordinates) x (coefficient
// function of the point): stresses, displaced coordinates, fluxes.  The field values come from nh_sample_eval; this kernel contracts them with the
// sparse coefficient tensor,
//   out[i][f] (+)= sc_i sum_{t: oidx[t] == f} coef[t] prod_v x_v[i * stride_v + off[t][v]],
// one thread per point, the entries in ascending f (a thread keeps the running sum of one f in a register and writes each output once).
namespace {
struct PExprK {
  i64 n, nentries;
  int nvars, nout, accumulate;
  const double *x[6];
  i64 stride[6];
  const int *oidx, *off;
  const double *coef, *scale;
  double *out;
};

__global__ void k_point_expr(PExprK p) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  const double sc = p.scale ? p.scale[i] : 1.;
  double *o = p.out + i * p.nout;
  if (!p.accumulate)
    for (int f = 0; f < p.nout; ++f) o[f] = 0.;
  const double *xb[6];
  for (int v = 0; v < p.nvars; ++v) xb[v] = p.x[v] + i * p.stride[v];
  int cur = -1;
  double acc = 0.;
  for (i64 t = 0; t < p.nentries; ++t) {  // (oidx / off / coef are the same for every thread: scalar loads)
    const int f = p.oidx[t];
    if (f != cur) {
      if (cur >= 0) o[cur] += sc * acc;
      cur = f;
      acc = 0.;
    }
    double m = p.coef[t];
    for (int v = 0; v < p.nvars; ++v) m *= xb[v][p.off[t * p.nvars + v]];
    acc += m;
  }
  if (cur >= 0) o[cur] += sc * acc;
}
}  // namespace

extern "C" int nh_point_expr(int64_t npoints, int nvars, const double *const *x_dev, const int64_t *strides, int64_t nentries, const int32_t *out_index_dev,
                             const int32_t *offsets_dev, const double *coef_dev, const double *scale_dev, int nout, double *out_dev, int accumulate, void *stream) {
  NH_REQUIRE(npoints >= 0 && nentries >= 0 && nout >= 1 && out_dev, "nh_point_expr: invalid argument");
  NH_REQUIRE(nvars >= 0 && nvars <= 6, "nh_point_expr: at most 6 factors (got %d)", nvars);
  NH_REQUIRE(!nentries || (out_index_dev && coef_dev && (!nvars || offsets_dev)), "nh_point_expr: NULL entry table");
  if (!npoints) return NH_OK;
  PExprK p;
  memset(&p, 0, sizeof p);
  p.n = npoints; p.nentries = nentries; p.nvars = nvars; p.nout = nout; p.accumulate = accumulate;
  for (int v = 0; v < nvars; ++v) {
    NH_REQUIRE(x_dev[v] && strides[v] >= 0, "nh_point_expr: NULL factor %d", v);
    p.x[v] = x_dev[v];
    p.stride[v] = strides[v];
  }
  p.oidx = out_index_dev; p.off = offsets_dev; p.coef = coef_dev; p.scale = scale_dev; p.out = out_dev;
  hipLaunchKernelGGL(k_point_expr, dim3((unsigned)((npoints + 255) / 256)), dim3(256), 0, nh_stream(stream), p);
  NH_LAUNCH_CHECK();
  return NH_OK;
}
