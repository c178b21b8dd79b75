// K6: sparsity pattern, built row-wise on the device.
//
// The reference sorts all nelems*nb^2 COO keys with a stable argsort and takes unique
// (evaluable.py:588-616, 5560-5682).  The result -- every (row, col) pair that some
// element couples, sorted lexicographically, structural zeros kept -- only depends on the
// element->dof maps, so it is produced here without a global sort:
//   1. transpose the test-dof map (dof -> elements) with a counting sort,
//   2. per row: gather the trial dofs of those elements into LDS, bitonic-sort, unique,
//   3. scan the row lengths -> rowptr, compact the columns,
//   4. element map: position of every (e, m, n) entry inside its row (binary search).
#include "nh_common.h"
#include <vector>

__device__ __forceinline__ i64 elem_off(const i64 *off, int nb, i64 e) { return off ? off[e] : e * (i64)nb; }
__device__ __forceinline__ int elem_nb(const i64 *off, int nb, i64 e) { return off ? (int)(off[e + 1] - off[e]) : nb; }

__global__ void k_count_rows(i64 nelems, int nbt, const int32_t *tdofs, const i64 *toff, i64 ntot, int32_t *cnt, int32_t *cand, int nbr,
                             const i64 *roff) {
  // one thread per (element, local test dof) entry of the concatenated list
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntot) return;
  i64 e;
  if (toff) {  // binary search the element that owns entry t
    i64 lo = 0, hi = nelems;
    while (hi - lo > 1) {
      i64 mid = (lo + hi) >> 1;
      if (toff[mid] <= t) lo = mid; else hi = mid;
    }
    e = lo;
  } else {
    e = t / nbt;
  }
  const int32_t dof = tdofs[t];
  atomicAdd(&cnt[dof], 1);
  atomicAdd(&cand[dof], elem_nb(roff, nbr, e));
}

__global__ void k_fill_rows(i64 nelems, int nbt, const int32_t *tdofs, const i64 *toff, i64 ntot, const i64 *estart, int32_t *cursor,
                            int32_t *elist) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntot) return;
  i64 e;
  if (toff) {
    i64 lo = 0, hi = nelems;
    while (hi - lo > 1) {
      i64 mid = (lo + hi) >> 1;
      if (toff[mid] <= t) lo = mid; else hi = mid;
    }
    e = lo;
  } else {
    e = t / nbt;
  }
  const int32_t dof = tdofs[t];
  const int pos = atomicAdd(&cursor[dof], 1);
  elist[estart[dof] + pos] = (int32_t)e;
}

__global__ void k_max_i32(const int32_t *v, i64 n, int32_t *out) {
  int m = 0;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) m = max(m, v[i]);
  for (int o = 32; o; o >>= 1) m = max(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// One workgroup per row: candidates -> LDS, bitonic sort, unique.  CAP = power of two >= max candidates.
template <int CAP, int THREADS>
__global__ __launch_bounds__(THREADS) void k_row_unique(i64 nrows, const i64 *estart, const int32_t *elist, int nbr, const int32_t *rdofs,
                                                        const i64 *roff, const i64 *cstart, int32_t *tmpcols, int32_t *rowcnt) {
  __shared__ int32_t key[CAP];
  __shared__ int32_t scan[THREADS];
  for (i64 row = blockIdx.x; row < nrows; row += gridDim.x) {
    const i64 e0 = estart[row], e1 = estart[row + 1];
    // gather
    int n = 0;
    if (roff == nullptr) {
      n = (int)(e1 - e0) * nbr;
      for (int i = threadIdx.x; i < CAP; i += THREADS) {
        int32_t k = 0x7fffffff;
        if (i < n) {
          const i64 e = elist[e0 + i / nbr];
          k = rdofs[e * nbr + i % nbr];
        }
        key[i] = k;
      }
    } else {
      for (int i = threadIdx.x; i < CAP; i += THREADS) key[i] = 0x7fffffff;
      __syncthreads();
      // ragged: serial offsets per element list entry (rows touch few elements)
      int base = 0;
      for (i64 j = e0; j < e1; ++j) {
        const i64 e = elist[j];
        const i64 o = roff[e];
        const int nb = (int)(roff[e + 1] - o);
        for (int i = threadIdx.x; i < nb; i += THREADS) key[base + i] = rdofs[o + i];
        base += nb;
      }
      n = base;
    }
    __syncthreads();
    // bitonic sort ascending
    for (int k = 2; k <= CAP; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < CAP; i += THREADS) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const int32_t a = key[i], b = key[ixj];
            const bool up = ((i & k) == 0);
            if ((a > b) == up) {
              key[i] = b;
              key[ixj] = a;
            }
          }
        }
        __syncthreads();
      }
    }
    // unique + compaction (block scan of head flags, processed in chunks of THREADS)
    const i64 dst = cstart[row];
    int total = 0;
    for (int c0 = 0; c0 < CAP; c0 += THREADS) {
      const int i = c0 + threadIdx.x;
      const int32_t k = key[i];
      const int flag = (k != 0x7fffffff) && (i == 0 || key[i - 1] != k);
      scan[threadIdx.x] = flag;
      __syncthreads();
      for (int d = 1; d < THREADS; d <<= 1) {
        int t = (threadIdx.x >= d) ? scan[threadIdx.x - d] : 0;
        __syncthreads();
        scan[threadIdx.x] += t;
        __syncthreads();
      }
      if (flag) tmpcols[dst + total + scan[threadIdx.x] - 1] = k;
      total += scan[THREADS - 1];
      __syncthreads();
    }
    if (threadIdx.x == 0) rowcnt[row] = total;
    __syncthreads();
  }
}

__global__ void k_compact(i64 nrows, const i64 *cstart, const int32_t *tmpcols, const i64 *srowptr, int32_t *scol) {
  // one wave per row
  const i64 row = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= nrows) return;
  const i64 a = srowptr[row], b = srowptr[row + 1], s = cstart[row];
  for (i64 i = lane; i < b - a; i += 64) scol[a + i] = tmpcols[s + i];
}

__global__ void k_emap(i64 nelems, int nbt, int nbr, const int32_t *tdofs, const i64 *toff, const int32_t *rdofs, const i64 *roff,
                       const i64 *eoff, const i64 *srowptr, const int32_t *scol, int32_t *emap, i64 total) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  i64 e;
  if (eoff) {
    i64 lo = 0, hi = nelems;
    while (hi - lo > 1) {
      i64 mid = (lo + hi) >> 1;
      if (eoff[mid] <= t) lo = mid; else hi = mid;
    }
    e = lo;
  } else {
    e = t / ((i64)nbt * nbr);
  }
  const int nr = elem_nb(roff, nbr, e);
  const i64 local = t - (eoff ? eoff[e] : e * (i64)nbt * nbr);
  const int m = (int)(local / nr), n = (int)(local % nr);
  const int32_t row = tdofs[elem_off(toff, nbt, e) + m];
  const int32_t col = rdofs[elem_off(roff, nbr, e) + n];
  i64 lo = srowptr[row], hi = srowptr[row + 1] - 1;
  const i64 base = lo;
  while (lo < hi) {
    const i64 mid = (lo + hi) >> 1;
    if (scol[mid] < col) lo = mid + 1; else hi = mid;
  }
  emap[t] = (int32_t)(lo - base);
}

struct MaskK {
  int nct, ncr, tot;
  int cnt[8], cum[8];
  unsigned char m[8][8];
};

__global__ void k_expand(i64 nrows, const i64 *srowptr, const int32_t *scol, MaskK mk, i64 *rowptr, i64 *colidx) {
  const i64 row = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= nrows) return;
  const i64 a = srowptr[row], len = srowptr[row + 1] - a;
  for (int c = 0; c < mk.nct; ++c) {
    const i64 base = a * mk.tot + len * mk.cum[c];
    if (lane == 0) rowptr[row * mk.nct + c] = base;
    if (colidx) {
      for (i64 i = lane; i < len; i += 64) {
        const i64 sc = scol[a + i];
        int k = 0;
        for (int d = 0; d < mk.ncr; ++d)
          if (mk.m[c][d]) colidx[base + i * mk.cnt[c] + (k++)] = sc * mk.ncr + d;
      }
    }
  }
  if (row == nrows - 1 && lane == 0) rowptr[nrows * mk.nct] = srowptr[nrows] * mk.tot;
}

static int make_mask(int nct, int ncr, const unsigned char *mask, MaskK *mk) {
  NH_REQUIRE(nct >= 1 && nct <= 8 && ncr >= 1 && ncr <= 8, "component counts must be 1..8 (got %d, %d)", nct, ncr);
  mk->nct = nct;
  mk->ncr = ncr;
  mk->tot = 0;
  for (int c = 0; c < nct; ++c) {
    mk->cnt[c] = 0;
    mk->cum[c] = mk->tot;
    for (int d = 0; d < ncr; ++d) {
      mk->m[c][d] = mask ? (mask[c * ncr + d] != 0) : 1;
      mk->cnt[c] += mk->m[c][d];
    }
    mk->tot += mk->cnt[c];
  }
  return NH_OK;
}

template <int CAP, int THREADS>
static void launch_row_unique(i64 nrows, const i64 *estart, const int32_t *elist, int nbr, const int32_t *rdofs, const i64 *roff,
                              const i64 *cstart, int32_t *tmpcols, int32_t *rowcnt, hipStream_t s) {
  const unsigned grid = (unsigned)(nrows < (1 << 20) ? nrows : (1 << 20));
  hipLaunchKernelGGL((k_row_unique<CAP, THREADS>), dim3(grid), dim3(THREADS), 0, s, nrows, estart, elist, nbr, rdofs, roff, cstart, tmpcols,
                     rowcnt);
}

constexpr int NH_UNION_MAX = 8;
struct UnionK {
  int np;
  const i64 *rowptr[NH_UNION_MAX], *colidx[NH_UNION_MAX];
  i64 *pos[NH_UNION_MAX];
};
// merge of the sorted column lists of row r of all parts; FILL: write the union's columns and, for every entry of every part, its position in the union
template <bool FILL>
__global__ void k_union(i64 nrows, UnionK u, int32_t *cnt, const i64 *rowptr_u, i64 *colidx_u) {
  const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  i64 at[NH_UNION_MAX], end[NH_UNION_MAX];
  for (int i = 0; i < u.np; ++i) at[i] = u.rowptr[i][r], end[i] = u.rowptr[i][r + 1];
  i64 out = FILL ? rowptr_u[r] : 0;
  int n = 0;
  for (;;) {
    i64 c = -1;
    for (int i = 0; i < u.np; ++i)
      if (at[i] < end[i] && (c < 0 || u.colidx[i][at[i]] < c)) c = u.colidx[i][at[i]];
    if (c < 0) break;
    for (int i = 0; i < u.np; ++i)
      if (at[i] < end[i] && u.colidx[i][at[i]] == c) {
        if (FILL) u.pos[i][at[i]] = out;
        ++at[i];
      }
    if (FILL) colidx_u[out] = c;
    ++out, ++n;
  }
  if (!FILL) cnt[r] = n;
}

extern "C" {

int nh_pattern_build(const nh_pattern_args *a, nh_pattern **out, void *stream) {
  NH_REQUIRE(a && out, "nh_pattern_build: NULL argument");
  NH_REQUIRE(a->nelems >= 0 && a->nrows >= 0 && a->ncols >= 0, "nh_pattern_build: negative size");
  NH_REQUIRE(a->nrows < 2147483647LL && a->ncols < 2147483647LL, "nh_pattern_build: dof counts exceed int32");
  NH_REQUIRE((a->nbt > 0) != (a->toff_dev != nullptr), "nh_pattern_build: give either nbt or toff_dev");
  NH_REQUIRE((a->nbr > 0) != (a->roff_dev != nullptr), "nh_pattern_build: give either nbr or roff_dev");
  hipStream_t s = nh_stream(stream);
  const i64 ne = a->nelems, nrows = a->nrows;
  // totals of the ragged lists
  // (copies are issued on `stream`, so that offsets a caller produced asynchronously on that stream are complete when read)
  i64 ntot_t = ne * a->nbt, ntot_r = ne * a->nbr;
  if (a->toff_dev) NH_CHECK_HIP(hipMemcpyAsync(&ntot_t, a->toff_dev + ne, sizeof(i64), hipMemcpyDeviceToHost, s));
  if (a->roff_dev) NH_CHECK_HIP(hipMemcpyAsync(&ntot_r, a->roff_dev + ne, sizeof(i64), hipMemcpyDeviceToHost, s));
  NH_CHECK_HIP(hipStreamSynchronize(s));
  (void)ntot_r;

  nh_pattern *p = new nh_pattern();
  memset(p, 0, sizeof *p);
  p->nelems = ne;
  p->nrows = nrows;
  p->ncols = a->ncols;
  p->nbt = a->nbt;
  p->nbr = a->nbr;

  int32_t *cnt = nullptr, *cand = nullptr, *cursor = nullptr, *elist = nullptr, *tmpcols = nullptr, *rowcnt = nullptr, *dmax = nullptr;
  i64 *estart = nullptr, *cstart = nullptr;
  int rc = NH_OK;
#define PB_CHECK(expr)                                                                       \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      nh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      rc = NH_EHIP;                                                                          \
      goto done;                                                                             \
    }                                                                                        \
  } while (0)
  {
    PB_CHECK(hipMalloc((void **)&cnt, sizeof(int32_t) * (nrows + 1) * 4 + 64));
    cand = cnt + (nrows + 1);
    cursor = cand + (nrows + 1);
    rowcnt = cursor + (nrows + 1);
    PB_CHECK(hipMalloc((void **)&dmax, 64));
    PB_CHECK(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (nrows + 1) * 4, s));
    PB_CHECK(hipMemsetAsync(dmax, 0, 64, s));
    PB_CHECK(hipMalloc((void **)&estart, sizeof(i64) * (nrows + 2) * 2));
    cstart = estart + (nrows + 2);
    PB_CHECK(hipMalloc((void **)&elist, sizeof(int32_t) * (ntot_t + 1)));
    if (ntot_t) {
      hipLaunchKernelGGL(k_count_rows, dim3((unsigned)((ntot_t + 255) / 256)), dim3(256), 0, s, ne, a->nbt, a->tdofs_dev, a->toff_dev, ntot_t,
                         cnt, cand, a->nbr, a->roff_dev);
      PB_CHECK(hipGetLastError());
    }
    if ((rc = nh_scan_exclusive(cnt, estart, nrows, s)) != NH_OK) goto done;
    if ((rc = nh_scan_exclusive(cand, cstart, nrows, s)) != NH_OK) goto done;
    if (ntot_t) {
      hipLaunchKernelGGL(k_fill_rows, dim3((unsigned)((ntot_t + 255) / 256)), dim3(256), 0, s, ne, a->nbt, a->tdofs_dev, a->toff_dev, ntot_t,
                         estart, cursor, elist);
      PB_CHECK(hipGetLastError());
    }
    int32_t maxcand = 0;
    i64 candtot = 0;
    if (nrows) {
      hipLaunchKernelGGL(k_max_i32, dim3(1024), dim3(256), 0, s, cand, nrows, dmax);
      PB_CHECK(hipGetLastError());
      PB_CHECK(hipMemcpyAsync(&maxcand, dmax, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      PB_CHECK(hipMemcpyAsync(&candtot, cstart + nrows, sizeof(i64), hipMemcpyDeviceToHost, s));
      PB_CHECK(hipStreamSynchronize(s));
    }
    if (maxcand > 16384) {
      nh_set_error("nh_pattern_build: a row has %d candidate columns; the row-wise kernel supports at most 16384", maxcand);
      rc = NH_ELIMIT;
      goto done;
    }
    PB_CHECK(hipMalloc((void **)&tmpcols, sizeof(int32_t) * (candtot + 1)));
    if (nrows) {
      if (maxcand <= 64) launch_row_unique<64, 64>(nrows, estart, elist, a->nbr, a->rdofs_dev, a->roff_dev, cstart, tmpcols, rowcnt, s);
      else if (maxcand <= 256) launch_row_unique<256, 64>(nrows, estart, elist, a->nbr, a->rdofs_dev, a->roff_dev, cstart, tmpcols, rowcnt, s);
      else if (maxcand <= 1024) launch_row_unique<1024, 256>(nrows, estart, elist, a->nbr, a->rdofs_dev, a->roff_dev, cstart, tmpcols, rowcnt, s);
      else if (maxcand <= 4096) launch_row_unique<4096, 256>(nrows, estart, elist, a->nbr, a->rdofs_dev, a->roff_dev, cstart, tmpcols, rowcnt, s);
      else launch_row_unique<16384, 256>(nrows, estart, elist, a->nbr, a->rdofs_dev, a->roff_dev, cstart, tmpcols, rowcnt, s);
      PB_CHECK(hipGetLastError());
    }
    PB_CHECK(hipMalloc((void **)&p->srowptr, sizeof(i64) * (nrows + 1)));
    if ((rc = nh_scan_exclusive(rowcnt, p->srowptr, nrows, s)) != NH_OK) goto done;
    PB_CHECK(hipMemcpyAsync(&p->nnz, p->srowptr + nrows, sizeof(i64), hipMemcpyDeviceToHost, s));
    PB_CHECK(hipStreamSynchronize(s));
    PB_CHECK(hipMalloc((void **)&p->scol, sizeof(int32_t) * (p->nnz + 1)));
    if (nrows) {
      hipLaunchKernelGGL(k_compact, dim3((unsigned)((nrows * 64 + 255) / 256)), dim3(256), 0, s, nrows, cstart, tmpcols, p->srowptr, p->scol);
      PB_CHECK(hipGetLastError());
    }
    // element map
    if (a->toff_dev || a->roff_dev) {
      // ragged: eoff[e] = prefix of nbt_e * nbr_e (computed on the host side of this call)
      std::vector<i64> ht(ne + 1), hr(ne + 1), he(ne + 1);
      if (a->toff_dev) PB_CHECK(hipMemcpyAsync(ht.data(), a->toff_dev, sizeof(i64) * (ne + 1), hipMemcpyDeviceToHost, s));
      else for (i64 e = 0; e <= ne; ++e) ht[e] = e * a->nbt;
      if (a->roff_dev) PB_CHECK(hipMemcpyAsync(hr.data(), a->roff_dev, sizeof(i64) * (ne + 1), hipMemcpyDeviceToHost, s));
      else for (i64 e = 0; e <= ne; ++e) hr[e] = e * a->nbr;
      PB_CHECK(hipStreamSynchronize(s));
      he[0] = 0;
      for (i64 e = 0; e < ne; ++e) he[e + 1] = he[e] + (ht[e + 1] - ht[e]) * (hr[e + 1] - hr[e]);
      // size classes of the ragged elements (functions per element, test or trial side: <= 8, 16, 24, 32, 48, 64, 96, 128, more): the
      // element kernels are launched per class with the LDS footprint of that class instead of the largest element of the mesh
      {
        static const int caps[NH_MAX_BUCKETS] = {8, 16, 24, 32, 48, 64, 96, 128, 1 << 30};
        std::vector<std::vector<int32_t>> lists(NH_MAX_BUCKETS);
        for (i64 e = 0; e < ne; ++e) {
          const int nb = (int)std::max(ht[e + 1] - ht[e], hr[e + 1] - hr[e]);
          int b = 0;
          while (nb > caps[b]) ++b;
          lists[b].push_back((int32_t)e);
          p->bucket_nbt[b] = std::max(p->bucket_nbt[b], (int)(ht[e + 1] - ht[e]));
          p->bucket_nbr[b] = std::max(p->bucket_nbr[b], (int)(hr[e + 1] - hr[e]));
        }
        i64 tot = 0;
        for (int b = 0; b < NH_MAX_BUCKETS; ++b) tot += (i64)lists[b].size();
        PB_CHECK(hipMalloc((void **)&p->bucket_store, sizeof(int32_t) * (tot + 1)));
        i64 at = 0;
        for (int b = 0; b < NH_MAX_BUCKETS; ++b) {
          p->bucket_n[b] = (i64)lists[b].size();
          p->bucket_elist[b] = p->bucket_store + at;
          if (!lists[b].empty()) PB_CHECK(hipMemcpyAsync(p->bucket_store + at, lists[b].data(), sizeof(int32_t) * lists[b].size(), hipMemcpyHostToDevice, s));
          at += (i64)lists[b].size();
        }
        PB_CHECK(hipStreamSynchronize(s));
        p->nbuckets = NH_MAX_BUCKETS;
      }
      p->emap_len = he[ne];
      PB_CHECK(hipMalloc((void **)&p->eoff, sizeof(i64) * (ne + 1)));
      PB_CHECK(hipMemcpyAsync(p->eoff, he.data(), sizeof(i64) * (ne + 1), hipMemcpyHostToDevice, s)); PB_CHECK(hipStreamSynchronize(s));
    } else {
      p->emap_len = ne * (i64)a->nbt * a->nbr;
    }
    PB_CHECK(hipMalloc((void **)&p->emap, sizeof(int32_t) * (p->emap_len + 1)));
    if (p->emap_len) {
      hipLaunchKernelGGL(k_emap, dim3((unsigned)((p->emap_len + 255) / 256)), dim3(256), 0, s, ne, a->nbt, a->nbr, a->tdofs_dev, a->toff_dev,
                         a->rdofs_dev, a->roff_dev, p->eoff, p->srowptr, p->scol, p->emap, p->emap_len);
      PB_CHECK(hipGetLastError());
    }
    PB_CHECK(hipStreamSynchronize(s));
  }
done:
  hipFree(cnt);
  hipFree(dmax);
  hipFree(estart);
  hipFree(elist);
  hipFree(tmpcols);
  if (rc != NH_OK) {
    nh_pattern_free(p);
    return rc;
  }
  *out = p;
  return NH_OK;
#undef PB_CHECK
}

int nh_pattern_free(nh_pattern *p) {
  if (!p) return NH_OK;
  hipFree(p->srowptr);
  hipFree(p->scol);
  hipFree(p->emap);
  hipFree(p->eoff);
  hipFree(p->bucket_store);
  hipFree(p->gsrc);
  hipFree(p->gsrc_sym);
  hipFree(p->tri_rank), hipFree(p->tri_base), hipFree(p->gsrc_tri), hipFree(p->gmirror);
  hipFree(p->gptr);
  hipFree(p->grow);
  nh_fused_free(p->fused);
  nh_owner_free(p->owner);
  delete p;
  return NH_OK;
}

int nh_pattern_fused_info(const nh_pattern *p, int *nblocks, int *rows_per_block, int64_t *nvisits, int *routine) {
  NH_REQUIRE(p, "nh_pattern_fused_info: NULL pattern");
  if (routine) *routine = p->fused ? p->fused->p1hex : -1;
  if (nblocks) *nblocks = p->fused ? p->fused->nblocks : 0;
  if (rows_per_block) *rows_per_block = p->fused ? p->fused->rows_per_block : 0;
  if (nvisits) *nvisits = p->fused ? p->fused->nvisits : 0;
  return NH_OK;
}

int nh_pattern_owner_info(const nh_pattern *p, int *nblocks, int *rows_per_block, int64_t *nvisits, int64_t *nchunks) {
  NH_REQUIRE(p, "nh_pattern_owner_info: NULL pattern");
  if (nblocks) *nblocks = p->owner ? p->owner->nblocks : 0;
  if (rows_per_block) *rows_per_block = p->owner ? p->owner->rows_per_block : 0;
  if (nvisits) *nvisits = p->owner ? p->owner->nvisits : 0;
  if (nchunks) *nchunks = p->owner ? p->owner->nchunks : 0;
  return NH_OK;
}

int nh_pattern_info(const nh_pattern *p, int64_t *nnz_scalar, const int64_t **srowptr_dev, const int32_t **scolidx_dev,
                    const int32_t **emap_dev, int64_t *emap_len, const int64_t **eoff_dev) {
  NH_REQUIRE(p, "nh_pattern_info: NULL pattern");
  if (nnz_scalar) *nnz_scalar = p->nnz;
  if (srowptr_dev) *srowptr_dev = (const int64_t *)p->srowptr;
  if (scolidx_dev) *scolidx_dev = p->scol;
  if (emap_dev) *emap_dev = p->emap;
  if (emap_len) *emap_len = p->emap_len;
  if (eoff_dev) *eoff_dev = (const int64_t *)p->eoff;
  return NH_OK;
}

int nh_pattern_expanded_nnz(const nh_pattern *p, int nct, int ncr, const unsigned char *mask, int64_t *nnz) {
  NH_REQUIRE(p && nnz, "nh_pattern_expanded_nnz: NULL argument");
  MaskK mk;
  int rc = make_mask(nct, ncr, mask, &mk);
  if (rc) return rc;
  *nnz = p->nnz * mk.tot;
  return NH_OK;
}

int nh_pattern_expand(const nh_pattern *p, int nct, int ncr, const unsigned char *mask, int64_t *rowptr_dev, int64_t *colidx_dev,
                      void *stream) {
  NH_REQUIRE(p && rowptr_dev, "nh_pattern_expand: NULL argument");
  MaskK mk;
  int rc = make_mask(nct, ncr, mask, &mk);
  if (rc) return rc;
  hipStream_t s = nh_stream(stream);
  if (p->nrows == 0) {
    NH_CHECK_HIP(hipMemsetAsync(rowptr_dev, 0, sizeof(i64), s));
    return NH_OK;
  }
  hipLaunchKernelGGL(k_expand, dim3((unsigned)((p->nrows * 64 + 255) / 256)), dim3(256), 0, s, p->nrows, p->srowptr, p->scol, mk,
                     (i64 *)rowptr_dev, (i64 *)colidx_dev);
  NH_LAUNCH_CHECK();
  return NH_OK;
}


// ---- union of sorted-unique CSR patterns (matrix integrals on several samples: volume + Nitsche / Robin boundary terms) ------------------------------------
// Every row of every part is a strictly increasing column list, so the union of a row is an NP-way merge: no sort, no hash.  One thread per row.
int nh_pattern_union_count(int nparts, int64_t nrows, const int64_t *const *rowptr_dev, const int64_t *const *colidx_dev, int64_t *rowptr_u_dev, int64_t *nnz, void *stream) {
  NH_REQUIRE(nparts >= 1 && nparts <= NH_UNION_MAX && rowptr_dev && colidx_dev && rowptr_u_dev && nnz, "nh_pattern_union_count: 1 .. %d parts, no NULL arguments", NH_UNION_MAX);
  hipStream_t s = nh_stream(stream);
  UnionK u;
  u.np = nparts;
  for (int i = 0; i < nparts; ++i) u.rowptr[i] = (const i64 *)rowptr_dev[i], u.colidx[i] = (const i64 *)colidx_dev[i], u.pos[i] = nullptr;
  *nnz = 0;
  if (nrows == 0) {
    NH_CHECK_HIP(hipMemsetAsync(rowptr_u_dev, 0, sizeof(i64), s));
    return NH_OK;
  }
  int32_t *cnt = nullptr;
  NH_CHECK_HIP(hipMalloc((void **)&cnt, (size_t)(nrows + 1) * 4));
  hipLaunchKernelGGL((k_union<false>), dim3((unsigned)((nrows + 127) / 128)), dim3(128), 0, s, (i64)nrows, u, cnt, (const i64 *)nullptr, (i64 *)nullptr);
  int rc = hipGetLastError() == hipSuccess ? nh_scan_exclusive(cnt, (i64 *)rowptr_u_dev, nrows, s) : NH_EHIP;
  hipError_t e = hipSuccess;
  if (rc == NH_OK) e = hipMemcpyAsync(nnz, (i64 *)rowptr_u_dev + nrows, sizeof(i64), hipMemcpyDeviceToHost, s);
  if (rc == NH_OK && e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(cnt);
  if (rc != NH_OK) return rc;
  NH_CHECK_HIP(e);
  return NH_OK;
}

int nh_pattern_union_fill(int nparts, int64_t nrows, const int64_t *const *rowptr_dev, const int64_t *const *colidx_dev, const int64_t *rowptr_u_dev, int64_t *colidx_u_dev,
                          int64_t *const *pos_dev, void *stream) {
  NH_REQUIRE(nparts >= 1 && nparts <= NH_UNION_MAX && rowptr_dev && colidx_dev && rowptr_u_dev && pos_dev, "nh_pattern_union_fill: 1 .. %d parts, no NULL arguments", NH_UNION_MAX);
  if (nrows == 0) return NH_OK;
  UnionK u;
  u.np = nparts;
  for (int i = 0; i < nparts; ++i) u.rowptr[i] = (const i64 *)rowptr_dev[i], u.colidx[i] = (const i64 *)colidx_dev[i], u.pos[i] = (i64 *)pos_dev[i];
  hipLaunchKernelGGL((k_union<true>), dim3((unsigned)((nrows + 127) / 128)), dim3(128), 0, nh_stream(stream), (i64)nrows, u, (int32_t *)nullptr, (const i64 *)rowptr_u_dev,
                     (i64 *)colidx_u_dev);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

}  // extern "C"
