// Runtime plumbing, device scan, basis tabulation (K2) and structured dof maps (K1).
#include "nh_common.h"

static thread_local char g_err[512] = "";

void nh_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

extern "C" {

int nh_abi_version(void) { return NH_ABI_VERSION; }
const char *nh_last_error(void) { return g_err; }

int nh_device_count(int *count) {
  NH_REQUIRE(count, "count is NULL");
  NH_CHECK_HIP(hipGetDeviceCount(count));
  return NH_OK;
}

int nh_set_device(int device) {
  NH_CHECK_HIP(hipSetDevice(device));
  return NH_OK;
}

int nh_device_info(int device, char *name, size_t namelen, int *cus, int64_t *lds_bytes, int64_t *hbm_bytes) {
  hipDeviceProp_t prop;
  NH_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  if (name && namelen) {
    snprintf(name, namelen, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cus) *cus = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int64_t)prop.maxSharedMemoryPerMultiProcessor;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return NH_OK;
}

int nh_malloc(void **dev, size_t bytes) {
  NH_REQUIRE(dev, "dev is NULL");
  NH_CHECK_HIP(hipMalloc(dev, bytes ? bytes : 8));
  return NH_OK;
}

int nh_free(void *dev) {
  if (dev) NH_CHECK_HIP(hipFree(dev));
  return NH_OK;
}

int nh_memcpy_h2d(void *dev, const void *host, size_t bytes, void *stream) {
  NH_CHECK_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, nh_stream(stream)));
  return NH_OK;
}

int nh_memcpy_d2h(void *host, const void *dev, size_t bytes, void *stream) {
  NH_CHECK_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, nh_stream(stream)));
  NH_CHECK_HIP(hipStreamSynchronize(nh_stream(stream)));
  return NH_OK;
}

int nh_memset(void *dev, int byte, size_t bytes, void *stream) {
  NH_CHECK_HIP(hipMemsetAsync(dev, byte, bytes, nh_stream(stream)));
  return NH_OK;
}

int nh_stream_sync(void *stream) {
  NH_CHECK_HIP(hipStreamSynchronize(nh_stream(stream)));
  return NH_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------
// Exclusive scan: 256 threads x 8 items per block, block totals scanned recursively.
// ---------------------------------------------------------------------------------
static constexpr int SCAN_T = 256, SCAN_I = 8, SCAN_B = SCAN_T * SCAN_I;

template <typename Tin>
__global__ __launch_bounds__(SCAN_T) void k_scan_block(const Tin *in, i64 *out, i64 *block_tot, i64 n) {
  __shared__ i64 sh[SCAN_T];
  const i64 base = (i64)blockIdx.x * SCAN_B + (i64)threadIdx.x * SCAN_I;
  i64 v[SCAN_I];
  i64 s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_I; ++i) {
    v[i] = (base + i < n) ? (i64)in[base + i] : 0;
    s += v[i];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < SCAN_T; d <<= 1) {  // Hillis-Steele inclusive scan of the thread sums
    i64 t = (threadIdx.x >= d) ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  i64 run = sh[threadIdx.x] - s;
#pragma unroll
  for (int i = 0; i < SCAN_I; ++i) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
  if (threadIdx.x == SCAN_T - 1) block_tot[blockIdx.x] = sh[SCAN_T - 1];
}

__global__ void k_scan_add(i64 *out, const i64 *block_off, i64 n, i64 *total_slot) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += block_off[i / SCAN_B];
  if (i == 0 && total_slot) *total_slot = block_off[(n + SCAN_B - 1) / SCAN_B];
}

template <typename Tin>
static int scan_rec(const Tin *in, i64 *out, i64 n, hipStream_t stream) {
  // out has n+1 entries
  if (n == 0) {
    NH_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(i64), stream));
    return NH_OK;
  }
  const i64 nblocks = (n + SCAN_B - 1) / SCAN_B;
  i64 *tot = nullptr, *off = nullptr;
  NH_CHECK_HIP(hipMalloc((void **)&tot, sizeof(i64) * (nblocks + 1) * 2));
  off = tot + nblocks + 1;
  int rc = NH_OK;
  hipError_t e = hipSuccess;
#define SCAN_CHECK(expr)                                                                     \
  do {                                                                                       \
    if (rc == NH_OK && (e = (expr)) != hipSuccess) {                                         \
      nh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e), __FILE__, __LINE__); \
      rc = NH_EHIP;                                                                          \
    }                                                                                        \
  } while (0)
  hipLaunchKernelGGL(k_scan_block<Tin>, dim3((unsigned)nblocks), dim3(SCAN_T), 0, stream, in, out, tot, n);
  SCAN_CHECK(hipGetLastError());
  if (rc == NH_OK) {
    if (nblocks == 1) {
      SCAN_CHECK(hipMemsetAsync(off, 0, sizeof(i64), stream));
      SCAN_CHECK(hipMemcpyAsync(off + 1, tot, sizeof(i64), hipMemcpyDeviceToDevice, stream));
    } else {
      rc = scan_rec<i64>(tot, off, nblocks, stream);
    }
  }
  if (rc == NH_OK) {
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, off, n, out + n);
    SCAN_CHECK(hipGetLastError());
  }
#undef SCAN_CHECK
  hipStreamSynchronize(stream);  // the scratch is freed below: every path (also the error paths) leaves through here
  hipFree(tot);
  return rc;
}

int nh_scan_exclusive(const int32_t *in_dev, i64 *out_dev, i64 n, hipStream_t stream) { return scan_rec<int32_t>(in_dev, out_dev, n, stream); }
int nh_scan_exclusive64(const i64 *in_dev, i64 *out_dev, i64 n, hipStream_t stream) { return scan_rec<i64>(in_dev, out_dev, n, stream); }

// ---------------------------------------------------------------------------------
// K2: tabulation.  One thread per (function, point); nested Horner in the reference's
// coefficient order (last variable most significant, descending powers) carried out in
// forward-mode dual numbers so value and all reference-space derivatives come out of
// one pass over the coefficients.
// ---------------------------------------------------------------------------------
template <int ND>
struct Dual {
  double v;
  double d[ND];
};

template <int ND, int L>
struct Horner {
  __device__ static Dual<ND> run(const double *c, int &idx, int r, const double *x) {
    Dual<ND> acc;
    acc.v = 0;
#pragma unroll
    for (int j = 0; j < ND; ++j) acc.d[j] = 0;
    for (int k = r; k >= 0; --k) {
      Dual<ND> in = Horner<ND, L - 1>::run(c, idx, r - k, x);
#pragma unroll
      for (int j = 0; j < ND; ++j) acc.d[j] = acc.d[j] * x[L] + in.d[j] + (j == L ? acc.v : 0.);
      acc.v = acc.v * x[L] + in.v;
    }
    return acc;
  }
};

template <int ND>
struct Horner<ND, -1> {
  __device__ static Dual<ND> run(const double *c, int &idx, int, const double *) {
    Dual<ND> out;
    out.v = c[idx++];
#pragma unroll
    for (int j = 0; j < ND; ++j) out.d[j] = 0;
    return out;
  }
};

template <int ND>
__global__ void k_tabulate(const double *coeffs, i64 nfn, int nc, int deg, const double *pts, int nq, double *T) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nfn * nq) return;
  const i64 fn = t / nq;
  const int q = (int)(t % nq);
  double x[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) x[j] = pts[q * ND + j];
  int idx = 0;
  Dual<ND> r = Horner<ND, ND - 1>::run(coeffs + fn * nc, idx, deg, x);
  double *o = T + t * (1 + ND);
  o[0] = r.v;
#pragma unroll
  for (int j = 0; j < ND; ++j) o[1 + j] = r.d[j];
}

static bool degree_of(int nd, int nc, int *deg) {
  for (int d = 0; d < 64; ++d) {
    i64 n = 1;  // C(nd + d, nd)
    for (int i = 1; i <= nd; ++i) n = n * (d + i) / i;
    if (n == nc) {
      *deg = d;
      return true;
    }
    if (n > nc) return false;
  }
  return false;
}

extern "C" int nh_poly_tabulate(const double *coeffs_dev, int64_t nfn, int ncoeffs, const double *points_dev, int nq, int ndims,
                                double *T_dev, void *stream) {
  NH_REQUIRE(ndims >= 1 && ndims <= 3, "nh_poly_tabulate: ndims must be 1..3, got %d", ndims);
  NH_REQUIRE(nfn >= 0 && nq >= 0, "nh_poly_tabulate: negative size");
  int deg;
  NH_REQUIRE(degree_of(ndims, ncoeffs, &deg), "nh_poly_tabulate: %d is not a valid coefficient count for %d variables", ncoeffs, ndims);
  const i64 n = (i64)nfn * nq;
  if (!n) return NH_OK;
  dim3 grid((unsigned)((n + 127) / 128)), block(128);
  hipStream_t s = nh_stream(stream);
  if (ndims == 1) hipLaunchKernelGGL(k_tabulate<1>, grid, block, 0, s, coeffs_dev, (i64)nfn, ncoeffs, deg, points_dev, nq, T_dev);
  if (ndims == 2) hipLaunchKernelGGL(k_tabulate<2>, grid, block, 0, s, coeffs_dev, (i64)nfn, ncoeffs, deg, points_dev, nq, T_dev);
  if (ndims == 3) hipLaunchKernelGGL(k_tabulate<3>, grid, block, 0, s, coeffs_dev, (i64)nfn, ncoeffs, deg, points_dev, nq, T_dev);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

// ---------------------------------------------------------------------------------
// K1: structured dof maps.
// ---------------------------------------------------------------------------------
struct SDofs {
  int nd;
  int shape[3], nloc[3], ndofs[3], soff[3];
};

__global__ void k_structured_dofs(SDofs p, const int *start, i64 elem_begin, i64 nelems, int nb, int32_t *dofs) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nelems * nb) return;
  i64 e = elem_begin + t / nb;
  int l = (int)(t % nb);
  i64 dof = 0;
  int idx[3], loc[3];
  for (int a = p.nd - 1; a >= 0; --a) {
    idx[a] = (int)(e % p.shape[a]);
    e /= p.shape[a];
    loc[a] = l % p.nloc[a];
    l /= p.nloc[a];
  }
  for (int a = 0; a < p.nd; ++a) {
    int d = (start[p.soff[a] + idx[a]] + loc[a]) % p.ndofs[a];
    dof = dof * p.ndofs[a] + d;
  }
  dofs[t] = (int32_t)dof;
}

extern "C" int nh_structured_dofs(int ndims, const int *shape, const int *nloc, const int *ndofs_axis, const int *start_dev,
                                  int64_t elem_begin, int64_t nelems, int32_t *dofs_dev, void *stream) {
  NH_REQUIRE(ndims >= 1 && ndims <= 3, "nh_structured_dofs: ndims must be 1..3");
  SDofs p;
  p.nd = ndims;
  int nb = 1, off = 0;
  i64 tot = 1;
  for (int a = 0; a < ndims; ++a) {
    p.shape[a] = shape[a];
    p.nloc[a] = nloc[a];
    p.ndofs[a] = ndofs_axis[a];
    p.soff[a] = off;
    off += shape[a];
    nb *= nloc[a];
    tot *= ndofs_axis[a];
  }
  NH_REQUIRE(tot < 2147483647LL, "nh_structured_dofs: %lld dofs exceed int32", (long long)tot);
  const i64 n = (i64)nelems * nb;
  if (!n) return NH_OK;
  hipLaunchKernelGGL(k_structured_dofs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nh_stream(stream), p, start_dev, (i64)elem_begin,
                     (i64)nelems, nb, dofs_dev);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

// ---------------------------------------------------------------------------------
// Rational (NURBS) bases: N_i = w_i B_i / W in place on per-element tables.
// ---------------------------------------------------------------------------------
template <int ND>
__global__ void k_rationalize(double *T, i64 nelems, int nb, const i64 *off, const int32_t *dofs, const double *w, const double *W, const double *dW,
                              int nq) {
  constexpr int S = 1 + ND;
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nelems * nq) return;
  const i64 e = t / nq;
  const int q = (int)(t % nq);
  const i64 f0 = off ? off[e] : e * (i64)nb;
  const int n = off ? (int)(off[e + 1] - off[e]) : nb;
  double Wq, dWq[ND];
  if (W) {
    Wq = W[t];
    for (int k = 0; k < ND; ++k) dWq[k] = dW[t * ND + k];
  } else {
    Wq = 0;
    for (int k = 0; k < ND; ++k) dWq[k] = 0;
    for (int i = 0; i < n; ++i) {
      const double wi = w[dofs[f0 + i]];
      const double *Ti = T + ((f0 + i) * nq + q) * S;
      Wq += wi * Ti[0];
      for (int k = 0; k < ND; ++k) dWq[k] += wi * Ti[1 + k];
    }
  }
  const double r = 1. / Wq;
  for (int i = 0; i < n; ++i) {
    const double wi = w[dofs[f0 + i]] * r;
    double *Ti = T + ((f0 + i) * nq + q) * S;
    const double B = Ti[0];
    Ti[0] = wi * B;
    for (int k = 0; k < ND; ++k) Ti[1 + k] = wi * (Ti[1 + k] - B * dWq[k] * r);
  }
}

extern "C" int nh_rationalize(double *T_dev, int64_t nelems, int nb, const int64_t *off_dev, const int32_t *dofs_dev, const double *weights_dev,
                              const double *W_dev, const double *dW_dev, int nq, int ndims, void *stream) {
  NH_REQUIRE(T_dev && dofs_dev && weights_dev, "nh_rationalize: NULL argument");
  NH_REQUIRE(ndims >= 1 && ndims <= 3, "nh_rationalize: ndims must be 1..3");
  NH_REQUIRE((nb > 0) != (off_dev != nullptr), "nh_rationalize: give either nb or off_dev");
  NH_REQUIRE(!W_dev || dW_dev, "nh_rationalize: W_dev without dW_dev");
  const i64 n = (i64)nelems * nq;
  if (!n) return NH_OK;
  dim3 grid((unsigned)((n + 127) / 128)), block(128);
  hipStream_t s = nh_stream(stream);
  if (ndims == 1) hipLaunchKernelGGL(k_rationalize<1>, grid, block, 0, s, T_dev, (i64)nelems, nb, (const i64 *)off_dev, dofs_dev, weights_dev, W_dev, dW_dev, nq);
  if (ndims == 2) hipLaunchKernelGGL(k_rationalize<2>, grid, block, 0, s, T_dev, (i64)nelems, nb, (const i64 *)off_dev, dofs_dev, weights_dev, W_dev, dW_dev, nq);
  if (ndims == 3) hipLaunchKernelGGL(k_rationalize<3>, grid, block, 0, s, T_dev, (i64)nelems, nb, (const i64 *)off_dev, dofs_dev, weights_dev, W_dev, dW_dev, nq);
  NH_LAUNCH_CHECK();
  return NH_OK;
}
