// Deterministic vector scatter: the second half of the owner-side reduction for residual vectors.
//
// The reference adds local vectors into the global one with numpy.add.at(out, dofs_e, values_e), element after element
// (Inflate._compile_with_out, evaluable.py:3405-3411; Assemble -> numeric.accumulate, numeric.py:434-460): a dof receives its
// contributions in ascending (element, local index) order.  The element kernels of this library can STORE their local vectors
// element-major instead of adding them with global f64 atomics (nh_block.local_dev, nh_vector_args.local_dev); the map built here
// lists, for every dof, the positions of its contributions in that array -- sorted by (position of the element in the sample, local
// index), the order of the reference's loop -- and k_scatter_gather sums them: bit-identical from run to run, one coalesced store per
// (dof, component).  The map is the transpose of the connectivity: a counting sort by dof (atomics only on integer cursors) followed by
// a sort of each dof's few entries by their key.
#include "nh_common.h"
#include <algorithm>

struct nh_scatter_plan {
  i64 nrows, npos;
  i64 *sptr;       // [nrows + 1]
  int32_t *ssrc;   // [npos]: position (e, m) in the local array = offset of (e, m) in dofs_dev
};

namespace {

struct SPK {
  i64 nlist, nrows;
  int nb;
  const int32_t *dofs, *elist;
  const i64 *off;
  int32_t *counts;
  const i64 *sptr;
  i64 *skey;
  int32_t *ssrc;
  int *bad;  // set when an element has >= 4096 functions (the sort key keeps 12 bits for the local index)
};

// phase 0: count the contributions per dof; phase 1: place (key, position) pairs through a per-dof cursor (counts, zeroed in between)
template <int PHASE>
__global__ void k_sp_visit(SPK p) {
  for (i64 ie = (i64)blockIdx.x * blockDim.x + threadIdx.x; ie < p.nlist; ie += (i64)gridDim.x * blockDim.x) {
    const i64 e = p.elist ? p.elist[ie] : ie;
    const i64 o = p.nb ? e * p.nb : p.off[e];
    const int nbe = p.nb ? p.nb : (int)(p.off[e + 1] - p.off[e]);
    if (PHASE == 0 && nbe >= 4096) atomicOr(p.bad, 1);
    for (int m = 0; m < nbe; ++m) {
      const int32_t dof = p.dofs[o + m];
      if (PHASE == 0) atomicAdd(p.counts + dof, 1);
      else {
        const i64 slot = p.sptr[dof] + atomicAdd(p.counts + dof, 1);
        p.skey[slot] = (ie << 12) | m;  // (a basis has fewer than 4096 functions per element)
        p.ssrc[slot] = (int32_t)(o + m);
      }
    }
  }
}

// the entries of a dof in the order of the reference's loop: ascending (list position, local index)
__global__ void k_sp_sort(i64 nrows, const i64 *sptr, i64 *skey, int32_t *ssrc) {
  for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (i64)gridDim.x * blockDim.x) {
    const i64 a = sptr[r], b = sptr[r + 1];
    for (i64 i = a + 1; i < b; ++i) {
      const i64 k = skey[i];
      const int32_t v = ssrc[i];
      i64 j = i;
      for (; j > a && skey[j - 1] > k; --j) {
        skey[j] = skey[j - 1];
        ssrc[j] = ssrc[j - 1];
      }
      skey[j] = k;
      ssrc[j] = v;
    }
  }
}

constexpr int MAXSP = 8;
struct GatherK {
  int count, ncomp, accumulate;
  i64 nrows;
  const i64 *sptr[MAXSP];
  const int32_t *ssrc[MAXSP];
  const double *local[MAXSP];
  double *out;
};

__global__ void k_scatter_gather(GatherK p) {
  const i64 n = p.nrows * p.ncomp;
  for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
    const i64 r = t / p.ncomp;
    const int c = (int)(t - r * p.ncomp);
    double acc = p.accumulate ? p.out[t] : 0.;
    for (int l = 0; l < p.count; ++l) {
      const i64 a = p.sptr[l][r], b = p.sptr[l][r + 1];
      for (i64 i = a; i < b; ++i) acc += p.local[l][(i64)p.ssrc[l][i] * p.ncomp + c];
    }
    p.out[t] = acc;
  }
}

}  // namespace

extern "C" {

int nh_scatter_plan_build(int64_t nelems, int64_t nrows, int nb, const int32_t *dofs_dev, const int64_t *off_dev, const int32_t *elist_dev,
                          int64_t nlist, nh_scatter_plan **plan_out, void *stream) {
  NH_REQUIRE(plan_out && dofs_dev && nelems >= 0 && nrows >= 0 && nb >= 0 && (nb > 0 || off_dev), "nh_scatter_plan_build: invalid argument");
  NH_REQUIRE(nb < 4096, "nh_scatter_plan_build: at most 4095 functions per element");
  hipStream_t s = nh_stream(stream);
  if (!elist_dev) nlist = nelems;
  NH_REQUIRE(nlist >= 0 && nlist < ((i64)1 << 50), "nh_scatter_plan_build: list length");
  i64 npos = 0, ntotal = nelems * (i64)nb;  // ntotal: positions of the whole basis (the plan stores 32-bit positions into its dof array)
  if (nb) npos = nlist * nb;
  else {
    NH_CHECK_HIP(hipMemcpyAsync(&ntotal, off_dev + nelems, sizeof(i64), hipMemcpyDeviceToHost, s));
    NH_CHECK_HIP(hipStreamSynchronize(s));
    if (!elist_dev) npos = ntotal;
  }
  NH_REQUIRE(ntotal < ((i64)1 << 31), "nh_scatter_plan_build: %lld (element, function) positions exceed the 32-bit positions of the plan", (long long)ntotal);
  nh_scatter_plan *P = new nh_scatter_plan{nrows, 0, nullptr, nullptr};
  int32_t *counts = nullptr;
  i64 *skey = nullptr;
  int rc = NH_OK, hbad = 0;
  SPK k{nlist, nrows, nb, dofs_dev, elist_dev, (const i64 *)off_dev, nullptr, nullptr, nullptr, nullptr, nullptr};
  const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>((nlist + 255) / 256, 256 * 16));
  const unsigned rgrid = (unsigned)std::max<i64>(1, std::min<i64>((nrows + 255) / 256, 256 * 16));
#define SP_HIP(expr)                                                                                              \
  do {                                                                                                            \
    hipError_t e_ = (expr);                                                                                       \
    if (e_ != hipSuccess) {                                                                                       \
      nh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);                    \
      rc = NH_EHIP;                                                                                               \
      goto done;                                                                                                  \
    }                                                                                                             \
  } while (0)
  SP_HIP(hipMalloc((void **)&counts, sizeof(int32_t) * (std::max<i64>(nrows, 1) + 1)));  // (+ 1: the flag of k_sp_visit)
  SP_HIP(hipMalloc((void **)&P->sptr, sizeof(i64) * (nrows + 1)));
  SP_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (std::max<i64>(nrows, 1) + 1), s));
  k.counts = counts;
  k.bad = counts + std::max<i64>(nrows, 1);
  if (nlist) hipLaunchKernelGGL(k_sp_visit<0>, dim3(grid), dim3(256), 0, s, k);
  SP_HIP(hipMemcpyAsync(&hbad, k.bad, sizeof(int), hipMemcpyDeviceToHost, s));
  SP_HIP(hipStreamSynchronize(s));
  if (hbad) {
    nh_set_error("nh_scatter_plan_build: an element with 4096 or more functions (the deterministic order keeps 12 bits for the local index)");
    rc = NH_ELIMIT;
    goto done;
  }
  if ((rc = nh_scan_exclusive(counts, P->sptr, nrows, s)) != NH_OK) goto done;
  if (!nb && elist_dev) {  // ragged basis on part of the topology: the total is the last scan entry
    SP_HIP(hipMemcpyAsync(&npos, P->sptr + nrows, sizeof(i64), hipMemcpyDeviceToHost, s));
    SP_HIP(hipStreamSynchronize(s));
  }
  P->npos = npos;
  SP_HIP(hipMalloc((void **)&P->ssrc, sizeof(int32_t) * std::max<i64>(npos, 1)));
  SP_HIP(hipMalloc((void **)&skey, sizeof(i64) * std::max<i64>(npos, 1)));
  SP_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * std::max<i64>(nrows, 1), s));
  k.sptr = P->sptr;
  k.skey = skey;
  k.ssrc = P->ssrc;
  if (nlist) hipLaunchKernelGGL(k_sp_visit<1>, dim3(grid), dim3(256), 0, s, k);
  if (nrows) hipLaunchKernelGGL(k_sp_sort, dim3(rgrid), dim3(256), 0, s, (i64)nrows, (const i64 *)P->sptr, skey, P->ssrc);
  SP_HIP(hipGetLastError());
  SP_HIP(hipStreamSynchronize(s));
done:
#undef SP_HIP
  if (counts) (void)hipFree(counts);
  if (skey) (void)hipFree(skey);
  if (rc != NH_OK) {
    nh_scatter_plan_free(P);
    return rc;
  }
  *plan_out = P;
  return NH_OK;
}

int nh_scatter_plan_free(nh_scatter_plan *P) {
  if (!P) return NH_OK;
  if (P->sptr) (void)hipFree(P->sptr);
  if (P->ssrc) (void)hipFree(P->ssrc);
  delete P;
  return NH_OK;
}

int nh_scatter_gather(int count, const nh_scatter_plan *const *plans, const double *const *locals_dev, int ncomp, double *out_dev, int accumulate,
                      void *stream) {
  NH_REQUIRE(count >= 1 && count <= MAXSP && plans && locals_dev && out_dev && ncomp >= 1, "nh_scatter_gather: invalid argument (at most %d plans)", MAXSP);
  GatherK p;
  memset(&p, 0, sizeof p);
  p.count = count;
  p.ncomp = ncomp;
  p.accumulate = accumulate != 0;
  p.nrows = plans[0]->nrows;
  for (int l = 0; l < count; ++l) {
    NH_REQUIRE(plans[l] && locals_dev[l] && plans[l]->nrows == p.nrows, "nh_scatter_gather: plan %d does not match", l);
    p.sptr[l] = plans[l]->sptr;
    p.ssrc[l] = plans[l]->ssrc;
    p.local[l] = locals_dev[l];
  }
  p.out = out_dev;
  const i64 n = p.nrows * ncomp;
  if (!n) return NH_OK;
  hipLaunchKernelGGL(k_scatter_gather, dim3((unsigned)std::min<i64>((n + 255) / 256, 256 * 32)), dim3(256), 0, nh_stream(stream), p);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

}  // extern "C"
