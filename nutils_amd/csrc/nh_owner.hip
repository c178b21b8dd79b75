// Owner blocks for VECTOR-VALUED matrix blocks on small uniform bases (NH_MATRIX_FUSED, include/nutils_hip.h): one pass, no scratch array, no atomics, sums in a
// fixed order.  Replaces the scatter of numeric.accumulate / numpy.add.at (numeric.py:434-460, evaluable.py:3405-3411) and the einsum that feeds it
// (evaluable.py:1885-1886) for blocks with nct = ncr = 2 or 3 components per node.
//
// The scalar owner blocks of nh_gather.hip keep every CSR entry of a block's rows in LDS and let the visiting elements add to it; with NC x NC values per node pair
// 62 kB hold 32 node rows of trilinear elasticity, which 75 elements visit.  Here the roles are swapped: LDS holds what the ELEMENTS know -- the physical
// gradients D[visit][point][node][axis] and the weights w|J| of every element that touches a row of the block, computed once per visit -- and the ENTRIES are
// formed in registers.  With G_mn[a][b] = sum_q w|J| D_m[q][a] D_n[q][b] (the Gram matrix of a node pair, independent of the form and of the number of components)
//   A[(m,c),(n,d)] = sum_ab C[c,a,d,b] G_mn[a][b]:
// a lane takes one CONTRIBUTION (visit, m, n) to a scalar entry of one of the block's rows and sums its Gram matrix over the points; the contributions to an entry sit
// in adjacent lanes (the plan sorts them by (row, position in the row, element) and never lets an entry straddle a wave), a segmented sum over the lanes of the entry
// leaves the total in its first lane, which applies the form tensor ONCE per entry and stores the NC x NC values in their places of the CSR array -- neighbouring
// lanes write neighbouring positions of one row.  The order of every sum is fixed by the plan: repeated assemblies are bit-identical.
//
// Per contribution and point: 2 * ND + 1 LDS reads (the lanes of a row read a dense set of tables: no bank conflicts beyond the bandwidth) and ND + ND * ND
// multiply-adds instead of the 3 * ND * ND per (pair, component pair) of the thread pass.  Read from HBM per assembly: 8 bytes per contribution (the item words),
// the vertices of the visiting elements (L2 hits beyond the first), nothing else; written: every CSR value once.
#include "nh_common.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

#include "nh_geom.inc"
#include "nh_blockplan.inc"

typedef unsigned long long u64;

// ---- plan ---------------------------------------------------------------------------------------------------------------------------------------------
// one item per local entry (e, m, n): key = rank position of the row << 16 | position of the entry in its scalar row, value = visit within the block | m | n
__global__ void k_op_items(i64 n, int nbt, int nbr, const int32_t *blk, const int32_t *dofs, const int32_t *rank, const int32_t *emap, const i64 *vptr, const unsigned *vlist,
                           u64 *key, unsigned *val, int *bad) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const i64 e = i / (nbt * nbr);
  const int l = (int)(i - e * nbt * nbr), m = l / nbr, nn = l - m * nbr;
  const int rp = rank[dofs[e * nbt + m]];
  const int b = blk[rp];
  i64 lo = vptr[b], hi = vptr[b + 1];
  const i64 v0 = lo;
  while (lo < hi) {  // the visits of a block are sorted by element
    const i64 mid = (lo + hi) >> 1;
    if ((i64)vlist[mid] < e) lo = mid + 1;
    else hi = mid;
  }
  const int pos = emap[i];
  if (lo - v0 >= 4096 || pos < 0 || pos > 0xffff || (i64)vlist[lo] != e) atomicOr(bad, 1);
  key[i] = (u64)(unsigned)rp << 16 | (u64)(unsigned)(pos & 0xffff);
  val[i] = (unsigned)(lo - v0) << 10 | (unsigned)m << 5 | (unsigned)nn;
}

// first sorted item of every block (binary search: a block without items gets an empty range)
__global__ void k_op_bstart(int nblocks, const i64 *bptr, i64 n, const u64 *key, i64 *bstart) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nblocks) return;
  const u64 k = (u64)bptr[b] << 16;
  i64 lo = 0, hi = n;
  while (lo < hi) {
    const i64 mid = (lo + hi) >> 1;
    if (key[mid] < k) lo = mid + 1;
    else hi = mid;
  }
  bstart[b] = lo;
}

// the items of a block packed into chunks of 64 lanes, entries (runs of equal keys) never straddling a chunk; FILL = false: count the chunks
// rows16: an entry of at most 16 items does not straddle a row of 16 lanes either (the segmented sum then runs on DPP row shifts)
// The entries of a MATRIX ROW are placed longest first (stable within a length): on a hexahedral mesh they have 8, 4, 2 or 1 items, 64 per row, and fill four rows of 16 lanes
// exactly (in the order of the matrix row, 1 2 1 2 4 2 ..., 11 % of the lanes were padding); which lanes an entry sits in changes nothing about its sum -- its items stay
// adjacent, in element order.  (Longest first over the whole BLOCK packs as well but scatters the stores of a chunk over all its rows: 1.44 -> 1.48 ms at 96^3.)
template <bool FILL>
__global__ void k_op_pack(int nblocks, const i64 *bptr, const i64 *bstart, const u64 *key, const unsigned *val, int32_t *nch, const i64 *cptr, uint32_t *isrc, uint32_t *idst,
                          int *maxseg, int rows16) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const i64 ib0 = bstart[b], ib1 = bstart[b + 1];
  int longest = 0;
  i64 cur = 0;
  for (i64 i0 = ib0; i0 < ib1;) {  // one matrix row (rank position = key >> 16) at a time: its heads stay together and store neighbouring entries
    i64 i1 = i0 + 1;
    while (i1 < ib1 && (key[i1] >> 16) == (key[i0] >> 16)) ++i1;
    u64 present = 0;  // bit L - 1: the row has entries of L items (L <= 64; longer ones make the plan fail)
    for (i64 i = i0; i < i1;) {
      const u64 k = key[i];
      i64 j = i + 1;
      while (j < i1 && key[j] == k) ++j;
      const int len = (int)(j - i);
      longest = max(longest, len);
      present |= 1ull << (min(len, 64) - 1);
      i = j;
    }
    while (present) {
      const int L = 64 - __clzll(present);  // the longest length not placed yet
      present &= ~(1ull << (L - 1));
      for (i64 i = i0; i < i1;) {
        const u64 k = key[i];
        i64 j = i + 1;
        while (j < i1 && key[j] == k) ++j;
        const int len = (int)(j - i);
        if (min(len, 64) == L) {
          if (rows16 && len <= 16 && (cur & 15) + len > 16) cur = (cur + 15) & ~(i64)15;
          if ((cur & 63) + len > 64) cur = (cur + 63) & ~(i64)63;
          if (FILL && len <= 64) {
            const i64 base = cptr[b] * 64 + cur;
            const unsigned dst = (unsigned)((k >> 16) - (u64)bptr[b]) << 16 | (unsigned)(k & 0xffff);
            for (int t = 0; t < len; ++t) {
              isrc[base + t] = 0x80000000u | (t == 0 ? 0x40000000u : 0u) | val[i + t];
              idst[base + t] = dst;
            }
          }
          cur += len;
        }
        i = j;
      }
    }
    i0 = i1;
  }
  if (!FILL) {
    nch[b] = (int32_t)((cur + 63) >> 6);
    atomicMax(maxseg, longest);
  }
}

__global__ void k_op_maxvisits(int nblocks, const i64 *vptr, int *out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nblocks) atomicMax(out, (int)(vptr[b + 1] - vptr[b]));
}

// first entry and length of the scalar row at every rank position
__global__ void k_op_rowinfo(i64 n, const int32_t *order, const i64 *srowptr, i64 *prs, int32_t *prl) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const i64 r = order[i];
  prs[i] = srowptr[r];
  prl[i] = (int32_t)(srowptr[r + 1] - srowptr[r]);
}

// vertex numbers of the visiting elements
__global__ void k_op_vvert(i64 n, int ng, const int32_t *vlist, const int32_t *gdofs, int32_t *vvert) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const i64 v = i / ng;
  vvert[i] = gdofs[(i64)vlist[v] * ng + (i - v * ng)];
}

// ---- kernel -------------------------------------------------------------------------------------------------------------------------------------------
struct OwnK {
  int nq;
  const double *weights;
  GeomK geom;
  BasisK test;
  const double *scale;
  double C[144];  // [c][a][d][b], NC <= 3, S <= 4
  GSlots gs;
  double lam, mu, mu2;  // ISOF: C[c][1+a][d][1+b] = lam d_ca d_db + mu d_cd d_ab + mu2 d_cb d_ad
  const i64 *srowptr;
  double *values;
  int store;
  i64 nrows;
  int R, nsteps, vmax, ldst, rows16, qc;
  const int32_t *vlist, *prl, *vvert;
  const i64 *prs;
  const i64 *vptr, *cptr, *bptr;
  const uint32_t *isrc, *idst;
  int debug;  // ablation builds: 1 = no geometry / D tables, 2 = no Gram sums, 4 = no segmented sum, 8 = no stores, 16 = no vertex numbers (the thread's own index instead), 32 = no item words, 64 = no table copies
  unsigned long long *tdbg;  // ablation builds: cycles of wave 0 per phase, summed over the blocks
};
#ifdef NH_ABLATION
#define ODBG(p) ((p).debug)
#define OTICK(i) do { if (p.tdbg && tid == 0) { const long long t_ = __builtin_readcyclecounter(); atomicAdd(p.tdbg + i, (unsigned long long)(t_ - tlast)); tlast = t_; } } while (0)
#else
#define ODBG(p) 0
#define OTICK(i)
#endif

__device__ __forceinline__ i64 bfn(const BasisK &b, i64 e) { return b.tab ? (i64)b.tab[e] * b.nb : 0; }

// compile-time loop
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<N, F, I + 1>(static_cast<F &&>(f));
  }
}
// one double from LDS byte address a + OFF with ds_read_b64 (never paired into ds_read2_b64); the caller waits with lds_wait
template <int OFF>
__device__ __forceinline__ double lds_rd(unsigned a) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
// the same in two statements, for groups of any size: the wait, then every value of the complete group through an (empty) statement of its own -- volatile asm statements
// keep their order, and what uses a value cannot be scheduled in front of the statement that hands it on
template <int N>
__device__ __forceinline__ void lds_wait_n() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
}
__device__ __forceinline__ void lds_tie(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ unsigned lds_addr(const double *p) { return (unsigned)(size_t)p; }

// wait until at most N LDS operations are outstanding; the values of the group that is complete then pass THROUGH the statement, so that nothing that uses them is
// scheduled in front of the wait (to the compiler the result of an asm statement is there as soon as the statement has been issued)
template <int N, int SD>
__device__ __forceinline__ void lds_wait(double (&d)[2][2][SD], double (&w)[2]) {
  static_assert(SD == 2 || SD == 3, "groups of 10 or 14 values");
  if constexpr (SD == 3)
    asm volatile("s_waitcnt lgkmcnt(%14)"
                 : "+v"(d[0][0][0]), "+v"(d[0][0][1]), "+v"(d[0][0][2]), "+v"(d[0][1][0]), "+v"(d[0][1][1]), "+v"(d[0][1][2]), "+v"(d[1][0][0]), "+v"(d[1][0][1]), "+v"(d[1][0][2]),
                   "+v"(d[1][1][0]), "+v"(d[1][1][1]), "+v"(d[1][1][2]), "+v"(w[0]), "+v"(w[1])
                 : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%10)"
                 : "+v"(d[0][0][0]), "+v"(d[0][0][1]), "+v"(d[0][1][0]), "+v"(d[0][1][1]), "+v"(d[1][0][0]), "+v"(d[1][0][1]), "+v"(d[1][1][0]), "+v"(d[1][1][1]), "+v"(w[0]), "+v"(w[1])
                 : "n"(N));
}

// the value of lane + D of this lane's row of 16 lanes (0 beyond the row): DPP row_shl on the two halves of the double
template <int D>
__device__ __forceinline__ double row_shl(double x) {
  const long long u = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffll), 0x100 | D, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), 0x100 | D, 0xf, 0xf, true);
  return __longlong_as_double((long long)(unsigned)lo | ((long long)hi << 32));
}

// stride of a visit's D table for a payload of n doubles and a point stride qs: odd (the lanes of phase 2 -- arbitrary visits -- are spread by an odd multiplier) and
// not = +-qs mod 16: with vs = qs (201 and 25 for trilinear elasticity) the 64 lanes of phase 1, 8 visits x 8 points, hit bank pair 9 (v + q) mod 16 -- eight to a
// pair where four is the optimum
__host__ __device__ constexpr int own_vs(int n, int qs) {
  int v = n | 1;
  while ((v - qs) % 16 == 0 || (v + qs) % 16 == 0) v += 2;
  return v;
}
constexpr int OWN_NT_MAX = 1024;  // (launch bound; the host picks 256 .. 1024 threads by the LDS a block takes)

// ISOF: the isotropic three-parameter family on the gradient slots, applied in closed form; USE0: the form reads the value slot (then D holds S slots per node)
// XLDS: the vertices of the visiting elements are staged in LDS; else every (visit, point) lane keeps them in registers (at most 512 threads then)
// GK > 0: the points are taken in chunks of p.qc (elements of 27 functions: the D tables of all points of the 8 visits of a box do not fit) -- a wave holds the
// Gram sums of GK chunks of contributions in registers across the point chunks; at most 256 threads then
template <int ND, int NB, int NC, bool ISOF, bool USE0, bool XLDS, int GK = 0>
__global__ __launch_bounds__(GK ? 256 : XLDS ? OWN_NT_MAX : OWN_NT_MAX / 2) void k_owner_rows_v(OwnK p) {
  constexpr int S = 1 + ND, NG = 1 << ND, SD = USE0 ? S : ND, O = USE0 ? 0 : 1;  // D slot a is operator slot O + a
  static_assert(!(ISOF && USE0), "the isotropic family has no value slot");
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int nq = p.nq, b = blockIdx.x, tid = threadIdx.x, OWN_NT = blockDim.x;
  // strides of the D table, in doubles: odd per point, odd and not +- the point stride mod 16 per visit (own_vs), so that the lanes of phase 1 (consecutive points of consecutive visits) and of
  // phase 2 (the nodes of arbitrary visits at one point) spread over the banks -- with 24 / 192 doubles they all met in two bank pairs
  const int qc = p.qc;  // points per chunk (= nq unless GK)
  const int QS = (NB * SD) | 1, VS = own_vs(qc * QS, QS), XS = (NG * ND) | 1;
  const bool iso = p.geom.kind == NH_GEOM_ISO && p.geom.ngb == NG;
  // LDS: [row starts R x i64][row lengths R x int, padded][form 144][quadrature weights][test table][geometry table][vertices vmax x NG x ND][weights vmax x nq][D vmax x nq x NB x SD]
  i64 *rs = reinterpret_cast<i64 *>(sm);
  int *rl = reinterpret_cast<int *>(rs + p.R);
  double *sC = sm + p.R + (p.R + 1) / 2;
  double *sWq = sC + (ISOF ? 0 : 144);  // quadrature weights
  double *sT = sWq + nq;
  double *sgT = sT + (p.ldst ? NB * nq * S : 0);
  // staged vertices: in the LAST point slab of the visit's own D table when the nq lanes of a visit sit in one wave (64 % nq == 0) -- every lane of the wave has read
  // them (to form J) before any lane stores its gradients, so the vertices cost no LDS of their own; else behind the weights
  const bool xalias = XLDS && !GK && 64 % nq == 0 && NG * ND <= QS;
  double *sW = sgT + (p.ldst && iso ? NG * nq * S : 0);
  double *sX = sW + p.vmax * qc;
  double *sD = sX + (iso && XLDS && !xalias ? p.vmax * XS : 0);
  const int XV = xalias ? VS : XS;  // stride of the vertex sets
  if (xalias) sX = sD + (qc - 1) * QS;
#ifdef NH_ABLATION
  long long tlast = __builtin_readcyclecounter();
#endif
  const i64 v0 = p.vptr[b];
  const int nv = (int)(p.vptr[b + 1] - v0);
  constexpr int PRE = 4;
  const int lane = tid & 63, wave = tid >> 6, nw = OWN_NT >> 6;
  const i64 c0 = p.cptr[b], c1 = p.cptr[b + 1];
  uint32_t pit[PRE], pds[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    const i64 c = c0 + wave + (i64)k * nw;
    pit[k] = c < c1 && !(ODBG(p) & 32) ? p.isrc[c * 64 + lane] : 0x80000000u | (lane % 3 ? 0u : 0x40000000u);
    pds[k] = c < c1 && !(ODBG(p) & 32) ? p.idst[c * 64 + lane] : 0u;
  }
  const i64 r0 = p.bptr[b];
  const int nr = (int)(p.bptr[b + 1] - r0);
  // Staging was a chain of dependent loads (block ranges -> visits / rows -> vertex numbers / row starts -> vertices); the plan holds row starts by rank position and
  // vertex numbers by visit, which leaves block ranges -> vertex numbers -> vertices.  (Walking the chain of a LATER block beside it, to have its lines in the L2 when it is dispatched, lost: 1.58 -> 1.68 .. 1.79 ms at 96^3
  // for distances of 64 .. 2048 blocks -- the memory system is short of requests, not of latency; profiles/r06_owner.md.)
  const bool stage_x = iso && XLDS;
  // level 1
  const int xv = tid / NG, xa = tid - xv * NG;  // this thread's (visit, vertex) pair among the first OWN_NT
  const int my_vert = stage_x && tid < nv * NG && !(ODBG(p) & 16) ? p.vvert[v0 * NG + tid] : tid;
  if (tid < nr) rs[tid] = p.prs[r0 + tid], rl[tid] = p.prl[r0 + tid];
  for (int i = tid + OWN_NT; i < nr; i += OWN_NT) rs[i] = p.prs[r0 + i], rl[i] = p.prl[r0 + i];
  // level 2
  double my_x[ND];
  if (stage_x && tid < nv * NG) {
#pragma unroll
    for (int d = 0; d < ND; ++d) my_x[d] = p.geom.verts[(i64)my_vert * ND + d];
  }
  if (!ISOF)
    for (int i = tid; i < 144; i += OWN_NT) sC[i] = p.C[i];
  for (int i = tid; i < nq; i += OWN_NT) sWq[i] = p.weights[i];
  if (p.ldst && !(ODBG(p) & 64)) {
    for (int i = tid; i < NB * nq * S; i += OWN_NT) sT[i] = p.test.T[i];
    if (iso)
      for (int i = tid; i < NG * nq * S; i += OWN_NT) sgT[i] = p.geom.gT[i];
  }
  if (stage_x) {
    if (tid < nv * NG) {
#pragma unroll
      for (int d = 0; d < ND; ++d) sX[xv * XV + xa * ND + d] = my_x[d];
    }
    for (int i = tid + OWN_NT; i < nv * NG; i += OWN_NT) {
      const int v = i / NG, a = i - v * NG;
      const i64 vert = p.vvert[v0 * NG + i];
#pragma unroll
      for (int d = 0; d < ND; ++d) sX[v * XV + a * ND + d] = p.geom.verts[vert * ND + d];
    }
  }
  __syncthreads();
  OTICK(0);
  // phase 1: lanes over (visit, point of the chunk q0 .. q0 + nql) -- inverse Jacobian, weight, physical gradients of the NB functions
  // (LT: the tables are the copies in LDS -- a compile-time fact: a pointer that is `p.ldst ? LDS : global` makes every read of a table a FLAT load, which the
  // stores of D in between serialise: 19 k of the 48 k cycles a block of 96^3 trilinear elasticity took, profiles/r06_owner.md)
  auto element_phase_of = [&](auto lt_tag, const int q0, const int nql) {
  constexpr bool LT = decltype(lt_tag)::value;
  for (int i = tid; i < nv * nql && !(ODBG(p) & 1); i += OWN_NT) {
    const int v = i / nql, ql = i - v * nql, q = q0 + ql;
    const i64 e = p.vlist[v0 + v];
    double Ji[ND][ND], det;
    if (iso) {
      double Xr[XLDS ? 1 : NG][ND];
      if constexpr (!XLDS) {
        int idx[NG];
#pragma unroll
        for (int a = 0; a < NG; ++a) idx[a] = p.geom.gdofs[e * NG + a];
#pragma unroll
        for (int a = 0; a < NG; ++a)
#pragma unroll
          for (int r = 0; r < ND; ++r) Xr[XLDS ? 0 : a][r] = p.geom.verts[(i64)idx[a] * ND + r];
      }
      double J[ND][ND];
#pragma unroll
      for (int r = 0; r < ND; ++r)
#pragma unroll
        for (int c = 0; c < ND; ++c) J[r][c] = 0;
#ifndef NH_OWNER_NO_B64
      if constexpr (LT && XLDS) {
        // (ds_read_b64 per double, as in the Gram loop below: a vertex and a table row per group, the next group in flight)
        const unsigned at = lds_addr(sgT + q * S + 1), ax = lds_addr(sX + v * XV), ts = (unsigned)(nq * S * 8);
        double tb[2][ND], xb[2][ND];
        auto issue = [&](auto a) {
          constexpr int A = decltype(a)::value;
          const unsigned ta = at + A * ts;
          static_for<ND>([&](auto c) {
            tb[A & 1][c] = lds_rd<c * 8>(ta);
            xb[A & 1][c] = lds_rd<(A * ND + c) * 8>(ax);
          });
        };
        issue(std::integral_constant<int, 0>());
        static_for<NG>([&](auto a) {
          constexpr int A = decltype(a)::value;
          if constexpr (A + 1 < NG) issue(std::integral_constant<int, A + 1>());
          lds_wait_n<(A + 1 < NG ? 2 * ND : 0)>();
#pragma unroll
          for (int c = 0; c < ND; ++c) lds_tie(tb[A & 1][c]), lds_tie(xb[A & 1][c]);
#pragma unroll
          for (int r = 0; r < ND; ++r)
#pragma unroll
            for (int c = 0; c < ND; ++c) J[r][c] += xb[A & 1][r] * tb[A & 1][c];
        });
      } else
#endif
#pragma unroll
      for (int a = 0; a < NG; ++a) {
        double t[ND];
#pragma unroll
        for (int c = 0; c < ND; ++c) t[c] = LT ? sgT[(a * nq + q) * S + 1 + c] : p.geom.gT[(a * nq + q) * S + 1 + c];
#pragma unroll
        for (int r = 0; r < ND; ++r)
#pragma unroll
          for (int c = 0; c < ND; ++c) J[r][c] += (XLDS ? sX[v * XV + a * ND + r] : Xr[XLDS ? 0 : a][r]) * t[c];
      }
      invert<ND>(J, Ji, det);
      if (p.geom.bnd_axis >= 0) {
        double s2 = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
          for (int c = 0; c < ND; ++c)
            if (j == p.geom.bnd_axis) s2 += Ji[j][c] * Ji[j][c];
        det *= sqrt(s2);
      }
      if (p.geom.nograd) {
#pragma unroll
        for (int r = 0; r < ND; ++r)
#pragma unroll
          for (int c = 0; c < ND; ++c) Ji[r][c] = 0.;
      }
    } else
      geometry_at<ND>(p.geom, e, q, nq, nullptr, Ji, det, nullptr);
    sW[v * qc + ql] = sWq[q] * fabs(det) * (p.scale ? p.scale[e * nq + q] : 1.);
    const double *Tg = p.test.T + bfn(p.test, e) * nq * S;
    double *D = sD + v * VS + ql * QS;
    constexpr int NBC = NB % 4 == 0 ? 4 : 3;  // the table rows of NBC functions are read before their gradients are stored
#pragma unroll
    for (int n0 = 0; n0 < NB; n0 += NBC) {
      double t[NBC][S];
#ifndef NH_OWNER_NO_B64
      if constexpr (LT) {
        const unsigned at = lds_addr(sT + q * S), ts = (unsigned)(nq * S * 8);
        static_for<NBC>([&](auto n) {
          constexpr int N = decltype(n)::value;
          if (n0 + N < NB) {
            const unsigned ta = at + (n0 + N) * ts;
            static_for<S>([&](auto j) {
              if constexpr (USE0 || j > 0) t[N][j] = lds_rd<j * 8>(ta);
            });
          }
        });
        lds_wait_n<0>();
#pragma unroll
        for (int n = 0; n < NBC; ++n)
#pragma unroll
          for (int j = USE0 ? 0 : 1; j < S; ++j)
            if (n0 + n < NB) lds_tie(t[n][j]);
      } else
#endif
#pragma unroll
      for (int n = 0; n < NBC; ++n)
#pragma unroll
        for (int j = USE0 ? 0 : 1; j < S; ++j)
          if (n0 + n < NB) t[n][j] = LT ? sT[((n0 + n) * nq + q) * S + j] : Tg[((n0 + n) * nq + q) * S + j];
#pragma unroll
      for (int n = 0; n < NBC; ++n) {
        if (n0 + n >= NB) continue;
        if (USE0) D[(n0 + n) * SD] = t[n][0];
#pragma unroll
        for (int c = 0; c < ND; ++c) {
          double sum = 0;
#pragma unroll
          for (int j = 0; j < ND; ++j) sum += t[n][1 + j] * Ji[j][c];
          D[(n0 + n) * SD + (USE0 ? 1 : 0) + c] = sum;
        }
      }
    }
  }
  };
  auto element_phase = [&](const int q0, const int nql) {
    if (p.ldst) element_phase_of(std::true_type(), q0, nql);
    else element_phase_of(std::false_type(), q0, nql);
  };
  // phase 2: a wave per chunk of 64 contributions (the item words of the wave's first chunks were requested before phase 0: their latency is behind the element phase)
  // Gram sums of one contribution over the nql points of the tables in LDS
  auto accumulate = [&](const uint32_t item, double (&G)[SD][SD], const int nql) {
    const bool valid = item >> 31;
    const int v = (item >> 10) & 0xfff, m = (item >> 5) & 31, n = item & 31;
    if (valid && !(ODBG(p) & 2)) {
      const double *Dv = sD + v * VS, *Wv = sW + v * qc;
      int q = 0;
#ifndef NH_OWNER_NO_B64
      // The table rows are read with ds_read_b64, one per double, through inline assembly: left to itself the compiler pairs neighbouring doubles into ds_read2_b64, which
      // takes 8 LDS cycles for its 1 kB where two ds_read_b64 take 2 + 2 (MI355X_MICROARCH.md, LDS table) -- and the LDS pipe was busy 77 % of the kernel
      // (SQ_LDS_IDX_ACTIVE, profiles/r06_owner.md).  Two points per group, the reads of the next group in flight while this one is summed; the waits are explicit
      // (2 SD + 1 reads per point: a group is at most 14, inside the 4 bits of lgkmcnt; the compiler's own waits can only be longer for these).
      if constexpr (GK == 0 && (SD == 2 || SD == 3)) if (nql >= 2) {  // (not with the points in chunks: the sums of GK chunks of contributions fill the registers -- 32^3 triquadratic elasticity 2.05 -> 3.1 ms)  // (forms that read the value slot in 3-D, SD = 4: 18 reads per group -- the compiler's loop below)
        constexpr int QB = ((NB * SD) | 1) * 8, NL = 2 * (2 * SD + 1);  // bytes between the rows of consecutive points; reads per group
        const unsigned am = (unsigned)(size_t)(Dv + m * SD), an = (unsigned)(size_t)(Dv + n * SD), aw = (unsigned)(size_t)Wv;
        double dA[2][2][SD], dB[2][2][SD], wA[2], wB[2];  // [m | n][point][slot]
        auto issue = [&](const int q0, double (&d)[2][2][SD], double (&w)[2]) {
          const unsigned bm = am + q0 * QB, bn = an + q0 * QB, bw = aw + q0 * 8;
          static_for<2>([&](auto k) {
            static_for<SD>([&](auto a) {
              d[0][k][a] = lds_rd<k * QB + a * 8>(bm);
              d[1][k][a] = lds_rd<k * QB + a * 8>(bn);
            });
            w[k] = lds_rd<k * 8>(bw);
          });
        };
        auto sum = [&](const double (&d)[2][2][SD], const double (&w)[2]) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            double wm[SD];
#pragma unroll
            for (int a = 0; a < SD; ++a) wm[a] = w[k] * d[0][k][a];
#pragma unroll
            for (int a = 0; a < SD; ++a)
#pragma unroll
              for (int bb = 0; bb < SD; ++bb) G[a][bb] += wm[a] * d[1][k][bb];
          }
        };
        issue(0, dA, wA);
        q = 2;
        for (; q + 4 <= nql; q += 4) {
          issue(q, dB, wB);
          lds_wait<NL, SD>(dA, wA);
          sum(dA, wA);
          issue(q + 2, dA, wA);
          lds_wait<NL, SD>(dB, wB);
          sum(dB, wB);
        }
        if (q + 2 <= nql) {
          issue(q, dB, wB);
          lds_wait<NL, SD>(dA, wA);
          sum(dA, wA);
          lds_wait<0, SD>(dB, wB);
          sum(dB, wB);
          q += 2;
        } else {
          lds_wait<0, SD>(dA, wA);
          sum(dA, wA);
        }
      }
#endif
      for (; q < nql; ++q) {
        const double w = Wv[q];
        const double *dm = Dv + q * QS + m * SD, *dn = Dv + q * QS + n * SD;
        double wm[SD], tn[SD];
#pragma unroll
        for (int a = 0; a < SD; ++a) wm[a] = w * dm[a], tn[a] = dn[a];
#pragma unroll
        for (int a = 0; a < SD; ++a)
#pragma unroll
          for (int bb = 0; bb < SD; ++bb) G[a][bb] += wm[a] * tn[bb];
      }
    }
  };
  // sum over the contributions of every entry, form tensor, store
  auto finish = [&](const uint32_t item, const uint32_t dst, double (&G)[SD][SD]) {
    const bool valid = item >> 31, head = (item >> 30) & 1;
    // segmented sum over the lanes of an entry (ascending elements), total in the first lane: rem = lanes of my entry behind me
    const u64 hm = __ballot(head || !valid);
    const u64 behind = lane < 63 ? hm >> (lane + 1) : 0ull;
    const int rem = behind ? __builtin_ctzll(behind) : 63 - lane;
    if (p.rows16) {
      // (the plan kept every entry inside a row of 16 lanes: DPP row shifts -- VALU moves -- instead of ds_bpermute through the LDS pipe: 9 % of the kernel at 96^3)
      auto step = [&](auto dtag) {
        constexpr int D = decltype(dtag)::value;
        double o[SD][SD];  // (all shifts, then ONE masked region of adds: a test per value made a step 50-76 instructions for its 9 adds -- the kernel is bound by VALU issue)
#pragma unroll
        for (int a = 0; a < SD; ++a)
#pragma unroll
          for (int bb = 0; bb < SD; ++bb) o[a][bb] = row_shl<D>(G[a][bb]);
        if (D <= rem) {
#pragma unroll
          for (int a = 0; a < SD; ++a)
#pragma unroll
            for (int bb = 0; bb < SD; ++bb) G[a][bb] += o[a][bb];
        }
      };
      if (p.nsteps > 0 && !(ODBG(p) & 4)) step(std::integral_constant<int, 1>());
      if (p.nsteps > 1 && !(ODBG(p) & 4)) step(std::integral_constant<int, 2>());
      if (p.nsteps > 2 && !(ODBG(p) & 4)) step(std::integral_constant<int, 4>());
      if (p.nsteps > 3 && !(ODBG(p) & 4)) step(std::integral_constant<int, 8>());
    } else
      for (int s = 0, d = 1; s < p.nsteps && !(ODBG(p) & 4); ++s, d <<= 1) {
#pragma unroll
        for (int a = 0; a < SD; ++a)
#pragma unroll
          for (int bb = 0; bb < SD; ++bb) {
            const double o = __shfl_down(G[a][bb], d);
            if (d <= rem) G[a][bb] += o;
          }
      }
    if (valid && head && !(ODBG(p) & 8)) {
      const int rowl = dst >> 16, pos = dst & 0xffff;
      const i64 a0 = rs[rowl];
      const int len = rl[rowl];
      double *base = p.values + a0 * (ISOF ? NC * NC : p.gs.tot);
      double tr = 0;
      if (ISOF) {
#pragma unroll
        for (int a = 0; a < ND; ++a) tr += G[a][a];
      }
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
#pragma unroll
        for (int dd = 0; dd < NC; ++dd) {
          if (!ISOF && !p.gs.mask[cc][dd]) continue;  // (uniform)
          double val;
          if constexpr (ISOF) {
            val = p.lam * G[cc < SD ? cc : 0][dd < SD ? dd : 0] + p.mu2 * G[dd < SD ? dd : 0][cc < SD ? cc : 0];
            if (cc == dd) val += p.mu * tr;
          } else {
            val = 0;
#pragma unroll
            for (int a = 0; a < SD; ++a)
#pragma unroll
              for (int bb = 0; bb < SD; ++bb) val += sC[((cc * S + O + a) * NC + dd) * S + O + bb] * G[a][bb];
          }
          // (ISOF: every component pair is there -- the launcher checks the layout: NC values per node column, NC scalar rows of `len` node columns each)
          double *ptr = ISOF ? base + ((i64)len * cc + pos) * NC + dd : base + (i64)len * p.gs.cum[cc] + (i64)pos * p.gs.cnt[cc] + p.gs.dpos[cc][dd];
          *ptr = p.store ? val : *ptr + val;
        }
    }
  };
  if constexpr (GK == 0) {
    // (A persistent grid -- the workgroup requests the ranges, rows and vertex numbers of its next block before the element phase of this one, copies the tables once --
    // was built twice: the loop costs registers past the 128 a wave has at two workgroups per CU, 1.58 -> 1.98 ms with 24 spilled; profiles/r06_owner.md.)
    element_phase(0, nq);
    OTICK(1);
    __syncthreads();
    OTICK(2);
    auto process = [&](const uint32_t item, const uint32_t dst) {
      double G[SD][SD];
#pragma unroll
      for (int a = 0; a < SD; ++a)
#pragma unroll
        for (int bb = 0; bb < SD; ++bb) G[a][bb] = 0;
      accumulate(item, G, nq);
      finish(item, dst, G);
    };
#pragma unroll
    for (int k = 0; k < PRE; ++k)
      if (c0 + wave + (i64)k * nw < c1) process(pit[k], pds[k]);
    for (i64 c = c0 + wave + (i64)PRE * nw; c < c1; c += nw) process(p.isrc[c * 64 + lane], p.idst[c * 64 + lane]);
    OTICK(3);
  } else {
    static_assert(GK <= PRE, "the preloaded item words cover the first group");
    // groups of GK chunks per wave (one group unless a block has more chunks than the workgroup holds); per group the points in chunks of qc
    for (i64 g0 = c0; g0 < c1; g0 += (i64)GK * nw) {
      uint32_t it[GK ? GK : 1], ds[GK ? GK : 1];
      double G[GK ? GK : 1][SD][SD];
#pragma unroll
      for (int k = 0; k < GK; ++k) {
        const i64 c = g0 + wave + (i64)k * nw;
        it[k] = g0 == c0 ? pit[k] : (c < c1 ? p.isrc[c * 64 + lane] : 0u);
        ds[k] = g0 == c0 ? pds[k] : (c < c1 ? p.idst[c * 64 + lane] : 0u);
#pragma unroll
        for (int a = 0; a < SD; ++a)
#pragma unroll
          for (int bb = 0; bb < SD; ++bb) G[k][a][bb] = 0;
      }
      for (int q0 = 0; q0 < nq; q0 += qc) {
        const int nql = min(qc, nq - q0);
        if (q0 || g0 != c0) __syncthreads();  // the tables of the previous chunk have been read
        element_phase(q0, nql);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GK; ++k) accumulate(it[k], G[k], nql);
      }
#pragma unroll
      for (int k = 0; k < GK; ++k)
        if (g0 + wave + (i64)k * nw < c1) finish(it[k], ds[k], G[k]);
    }
  }
}

}  // namespace

void nh_owner_free(nh_owner_plan *o) {
  if (!o) return;
  hipFree(o->order), hipFree(o->bptr), hipFree(o->vptr), hipFree(o->vlist), hipFree(o->cptr), hipFree(o->isrc), hipFree(o->idst), hipFree(o->prs), hipFree(o->prl), hipFree(o->vvert);
  delete o;
}

// LDS bytes of a block with `vmax` visits
static size_t owner_lds(int R, int vmax, int nq, int qc, int nb, int nd, int sd, bool isof, bool ldst, bool iso, bool xlds = true) {
  const int S = 1 + nd, NG = 1 << nd;
  const size_t QS = (size_t)(nb * sd) | 1, VS = (size_t)own_vs((int)(qc * QS), (int)QS), XS = (size_t)(NG * nd) | 1;  // (the strides of the kernel)
  const bool xalias = nb <= 9 && 64 % nq == 0 && (size_t)(NG * nd) <= QS;  // (vertices inside the D tables when the lanes of a visit share a wave; not with point chunks)
  size_t d = (size_t)R + (R + 1) / 2 + (isof ? 0 : 144) + (size_t)nq + (ldst ? (size_t)nb * nq * S : 0) + (ldst && iso ? (size_t)NG * nq * S : 0) + (iso && xlds && !xalias ? (size_t)vmax * XS : 0) +
             (size_t)vmax * qc + (size_t)vmax * VS;
  return d * sizeof(double);
}

static int nh_owner_prepare(nh_pattern *p, const nh_matrix_args *a, int sd, bool isof, bool ldst, bool iso, hipStream_t s) {
  const i64 ne = p->nelems, nrows = p->nrows;
  const int nbt = p->nbt, nbr = p->nbr;
  if (!(ne < (1ll << 31) && nrows < (1ll << 31) && p->emap_len == ne * nbt * nbr && p->emap_len < (1ll << 32) && nbt <= 32 && nbr <= 32)) return NH_ELIMIT;
  const int32_t *dofs = a->test.dofs_dev;
  BpTmp t;
  t.n = 0;
  unsigned *order = nullptr, *vlist = nullptr, *val = nullptr, *val2 = nullptr;
  int32_t *rank = nullptr, *nch = nullptr;
  i64 *vptr = nullptr, *bstart = nullptr, *cptr = nullptr, *bptr = nullptr;
  int32_t *blk = nullptr;
  u64 *key = nullptr, *key2 = nullptr;
  uint32_t *isrc = nullptr, *idst = nullptr;
  int *flags = nullptr;  // [0] bad item, [1] longest entry, [2] most visits of a block
  int maxlen = 0, rc = NH_OK, hflags[3] = {0, 0, 0};
  nh_owner_plan *o = nullptr;
#define OP_CHECK(expr)                                                                          \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      nh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
      rc = NH_EHIP;                                                                             \
      goto done;                                                                                \
    }                                                                                           \
  } while (0)
  {
    unsigned *skeys = nullptr;
    if ((rc = bp_cluster(t, p, a, &order, &rank, &maxlen, s, &skeys)) != NH_OK) goto done;
    if (maxlen > 0xffff) {
      rc = NH_ELIMIT;
      goto done;
    }
    OP_CHECK(bp_alloc(t, &flags, 3));
    // rows per block: the largest candidate whose fullest block fits the LDS budget (two workgroups per CU)
    size_t budget = 80 * 1024;
    if (getenv("NH_OWNER_LDS")) budget = (size_t)std::max(16, std::min(160, atoi(getenv("NH_OWNER_LDS")))) * 1024;
    int R = 0, vmax = 0, nblocks = 0;
    i64 nvisits = 0;
    // (elements of more than 9 functions -- triquadratic: 27 -- take their points in chunks and hold the sums of a block's contributions in registers meanwhile: boxes of at
    // most 8 rows, the nodes one element owns, so that a workgroup of four waves holds them in four chunks of 64 each)
    const bool big = nbt > 9;
    // (boxes of up to 24 rows often have no more visits than those of up to 16 -- 45 at most on a hexahedral mesh -- and are fewer: 96^3 trilinear elasticity 1.39 -> 1.35 ms)
    const int cand[] = {64, 32, 24, 16, 8, 4, 2, 1};
    const size_t cand_budget[] = {0, 0, 0, 0, 0, 0, 0, 0};  // 0: the common budget (two workgroups per CU)
    int forced = getenv("NH_OWNER_ROWS") ? std::max(1, std::min(256, atoi(getenv("NH_OWNER_ROWS")))) : 0;
    int qc = a->nq;
    for (int ci = big ? 4 : 0; ci < 8; ++ci) {
      const int Rc = forced ? forced : cand[ci];
      const int tn = t.n;
      OP_CHECK(hipMemsetAsync(flags, 0, 3 * sizeof(int), s));
      if ((rc = bp_blocks(t, skeys, nrows, Rc, &nblocks, &bptr, &blk, s)) != NH_OK) goto done;
      if ((rc = bp_visits(t, p, dofs, rank, blk, nblocks, &nvisits, &vptr, &vlist, s)) != NH_OK) goto done;
      hipLaunchKernelGGL(k_op_maxvisits, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, s, nblocks, vptr, flags + 2);
      OP_CHECK(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
      OP_CHECK(hipStreamSynchronize(s));
      for (int parts = 1; parts <= (big ? 8 : 1) && !R; ++parts) {
        const int q = (a->nq + parts - 1) / parts;
        const size_t bud = (!forced && cand_budget[ci] && !getenv("NH_OWNER_LDS")) ? cand_budget[ci] : budget;
        if (hflags[2] < 4096 && owner_lds(Rc, hflags[2], a->nq, q, nbt, a->ndims, sd, isof, ldst, iso, parts > 1) <= bud) R = Rc, vmax = hflags[2], qc = q;  // (without the staged vertices if need be)
      }
      if (R) break;
      if (getenv("NH_OWNER_VERBOSE")) fprintf(stderr, "nh_owner plan: %d rows per block -> at most %d visits, %zu B of LDS: over the budget\n", Rc, hflags[2], owner_lds(Rc, hflags[2], a->nq, a->nq, nbt, a->ndims, sd, isof, ldst, iso));
      if (forced) break;
      for (int i = tn; i < t.n; ++i) hipFree(t.ptr[i]);  // (this candidate's arrays)
      t.n = tn;
    }
    if (!R) {
      rc = NH_ELIMIT;
      goto done;
    }
    const i64 ni = p->emap_len;
    OP_CHECK(bp_alloc(t, &key, (size_t)ni));
    OP_CHECK(bp_alloc(t, &val, (size_t)ni));
    hipLaunchKernelGGL(k_op_items, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, s, ni, nbt, nbr, blk, dofs, rank, p->emap, vptr, vlist, key, val, flags);
    OP_CHECK(hipGetLastError());
    int bits = 17;
    while ((1ll << (bits - 16)) < nrows) ++bits;
    if ((rc = bp_sort_pairs(t, key, val, (size_t)ni, bits, &key2, &val2, s)) != NH_OK) goto done;  // stable: the items of an entry in (element, m, n) order
    OP_CHECK(bp_alloc(t, &bstart, (size_t)nblocks + 1));
    OP_CHECK(bp_alloc(t, &nch, (size_t)nblocks + 1));
    OP_CHECK(bp_alloc(t, &cptr, (size_t)nblocks + 1));
    hipLaunchKernelGGL(k_op_bstart, dim3((unsigned)((nblocks + 256) / 256)), dim3(256), 0, s, nblocks, bptr, ni, key2, bstart);
    hipLaunchKernelGGL((k_op_pack<false>), dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, s, nblocks, bptr, bstart, key2, val2, nch, (const i64 *)nullptr, (uint32_t *)nullptr,
                       (uint32_t *)nullptr, flags + 1, 0);
    OP_CHECK(hipGetLastError());
    OP_CHECK(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    OP_CHECK(hipStreamSynchronize(s));
    // entries of at most 16 contributions (hexahedra: 8, quadrilaterals: 4): kept inside rows of 16 lanes, summed with DPP row shifts instead of ds_bpermute
    const int rows16 = hflags[1] <= 16 && !getenv("NH_OWNER_NO_DPP");
    if (rows16) {
      hipLaunchKernelGGL((k_op_pack<false>), dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, s, nblocks, bptr, bstart, key2, val2, nch, (const i64 *)nullptr, (uint32_t *)nullptr,
                         (uint32_t *)nullptr, flags + 1, 1);
      OP_CHECK(hipGetLastError());
    }
    if ((rc = nh_scan_exclusive(nch, cptr, nblocks, s)) != NH_OK) goto done;
    i64 nchunks = 0;
    OP_CHECK(hipMemcpyAsync(&nchunks, cptr + nblocks, sizeof(i64), hipMemcpyDeviceToHost, s));
    OP_CHECK(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    OP_CHECK(hipStreamSynchronize(s));
    if (hflags[0] || hflags[1] > 64 || nchunks * 64 >= (1ll << 32)) {
      rc = NH_ELIMIT;
      goto done;
    }
    OP_CHECK(bp_alloc(t, &isrc, (size_t)nchunks * 64));
    OP_CHECK(bp_alloc(t, &idst, (size_t)nchunks * 64));
    OP_CHECK(hipMemsetAsync(isrc, 0, (size_t)nchunks * 64 * 4, s));
    OP_CHECK(hipMemsetAsync(idst, 0, (size_t)nchunks * 64 * 4, s));
    hipLaunchKernelGGL((k_op_pack<true>), dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, s, nblocks, bptr, bstart, key2, val2, (int32_t *)nullptr, cptr, isrc, idst, (int *)nullptr, rows16);
    OP_CHECK(hipGetLastError());
    OP_CHECK(hipStreamSynchronize(s));
    o = new nh_owner_plan();
    memset(o, 0, sizeof *o);
    o->nblocks = nblocks, o->rows_per_block = R, o->max_visits = vmax, o->nvisits = nvisits, o->nchunks = nchunks;
    o->nsteps = 0;
    while ((1 << o->nsteps) < hflags[1]) ++o->nsteps;
    o->rows16 = rows16;
    o->qc = qc;
    o->order = reinterpret_cast<int32_t *>(bp_keep(t, order));
    o->bptr = bp_keep(t, bptr);
    o->vptr = bp_keep(t, vptr);
    o->vlist = reinterpret_cast<int32_t *>(bp_keep(t, vlist));
    o->cptr = bp_keep(t, cptr);
    o->isrc = bp_keep(t, isrc);
    o->idst = bp_keep(t, idst);
    OP_CHECK(hipMalloc((void **)&o->prs, sizeof(i64) * (size_t)std::max<i64>(nrows, 1)));
    OP_CHECK(hipMalloc((void **)&o->prl, sizeof(int32_t) * (size_t)std::max<i64>(nrows, 1)));
    hipLaunchKernelGGL(k_op_rowinfo, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s, nrows, o->order, p->srowptr, o->prs, o->prl);
    OP_CHECK(hipGetLastError());
    if (getenv("NH_OWNER_VERBOSE"))
      fprintf(stderr, "nh_owner plan: %d blocks of %d rows, %lld visits (%.2f per element, at most %d per block), %lld chunks for %lld items (%.2f lanes used), entries of up to %d items, %zu B of LDS\n",
              nblocks, R, (long long)nvisits, (double)nvisits / (double)ne, vmax, (long long)nchunks, (long long)ni, (double)ni / (64. * (double)nchunks), hflags[1],
              owner_lds(R, vmax, a->nq, qc, nbt, a->ndims, sd, isof, ldst, iso));
  }
done:
#undef OP_CHECK
  bp_free(t);
  if (rc == NH_OK) p->owner = o;
  else if (o) nh_owner_free(o);
  return rc;
}

int nh_owner_vector(const nh_matrix_args *a, const GSlots &slots, bool *done, hipStream_t s) {
  *done = false;
  const int nc = a->nct;
  if (nc < 2 || nc > 3 || a->ncr != nc || a->cq_dev || a->test.off_dev || a->trial.off_dev || !a->test.nb || a->test.nb != a->trial.nb || a->elist_dev) return NH_OK;
  if (a->test.T_dev != a->trial.T_dev || a->test.tab_dev != a->trial.tab_dev || a->test.dofs_dev != a->trial.dofs_dev) return NH_OK;
  nh_pattern *pat = const_cast<nh_pattern *>(a->pattern);
  if (!pat || pat->nelems != a->nelems || pat->eoff || pat->owner_failed || pat->nbt != a->test.nb || pat->nbr != a->trial.nb) return NH_OK;
  const int key = a->ndims * 1000 + a->test.nb * 10 + nc;
  switch (key) {
    case 3083: case 2042: case 2092: case 3273: break;
    default: return NH_OK;
  }
  const int S = 1 + a->ndims;
  OwnK p;
  memset(&p, 0, sizeof p);
  for (int i = 0; i < 144; ++i) p.C[i] = i < nc * S * nc * S ? a->C_host[i] : 0.;
  p.gs = slots;
  // the isotropic three-parameter family on the gradient slots (all blocks coupled, one component per axis)?  does the form read the value slot?
  bool isof = nc == a->ndims && !getenv("NUTILS_AMD_NO_ISOFORM"), use0 = false;
  {
    const double *C = a->C_host;
    auto at = [&](int c, int sa, int d, int sb) { return C[((c * S + sa) * nc + d) * S + sb]; };
    const double lam = at(0, 1, 1, 2), mu2 = at(0, 2, 1, 1), mu = at(0, 2, 0, 2);
    for (int c = 0; c < nc; ++c)
      for (int sa = 0; sa < S; ++sa)
        for (int d = 0; d < nc; ++d)
          for (int sb = 0; sb < S; ++sb) {
            const double expect = (sa && sb) ? lam * (c == sa - 1 && d == sb - 1) + mu * (c == d && sa == sb) + mu2 * (c == sb - 1 && sa - 1 == d) : 0.;
            if (at(c, sa, d, sb) != expect || !slots.mask[c][d]) isof = false;
            if ((!sa || !sb) && at(c, sa, d, sb) != 0.) use0 = true;
          }
    p.lam = lam, p.mu = mu, p.mu2 = mu2;
    if (slots.tot != nc * nc) isof = false;  // (the closed form also takes the full layout for granted)
    for (int c = 0; c < nc; ++c) {
      if (slots.cnt[c] != nc || slots.cum[c] != c * nc) isof = false;
      for (int d = 0; d < nc; ++d)
        if (slots.dpos[c][d] != d) isof = false;
    }
  }
  const bool iso = a->geom.kind == NH_GEOM_ISO && a->geom.ngb == (1 << a->ndims);
  const size_t ldsb = sizeof(double) * (size_t)a->nq * S * (a->test.nb + (iso ? (1 << a->ndims) : 0));
  const bool ldst = !a->test.tab_dev && ldsb <= 32 * 1024;
  const int sd = use0 ? S : a->ndims;
  if (!pat->owner) {
    const int rc = nh_owner_prepare(pat, a, sd, isof, ldst, iso, s);
    if (rc == NH_ELIMIT) {
      pat->owner_failed = 1;
      return NH_OK;
    }
    if (rc != NH_OK) return rc;
  }
  const nh_owner_plan *o = pat->owner;
  // points per chunk: the plan's choice for elements of more than 9 functions (made for the quadrature of its first launch: any other takes more or fewer chunks of that
  // size), all points otherwise
  const bool big = a->test.nb > 9;
  const int qc = big ? std::min(o->qc, a->nq) : a->nq;
  const bool chunked = big;
  size_t lds = owner_lds(o->rows_per_block, o->max_visits, a->nq, qc, a->test.nb, a->ndims, sd, isof, ldst, iso);
  // the staged vertices are given up where they cost a workgroup per CU (two of 80 kB fit, three of 53 kB)
  const size_t lds0 = owner_lds(o->rows_per_block, o->max_visits, a->nq, qc, a->test.nb, a->ndims, sd, isof, ldst, iso, false);
  bool xlds = iso && ((160 * 1024 / lds0 == 160 * 1024 / std::max<size_t>(lds, 1)) || chunked) && lds <= 160 * 1024;
  if (getenv("NH_OWNER_XLDS")) xlds = iso && atoi(getenv("NH_OWNER_XLDS")) != 0;
  if (!xlds) lds = lds0;
  if (lds > 160 * 1024) return NH_OK;  // (a plan built for other tables / forms: the caller keeps its other paths)
  p.nq = a->nq;
  p.weights = a->weights_dev;
  p.geom = to_k(a->geom);
  p.geom.nograd = !uses_gradients(a->C_host, nc, S, nc);
  p.test = to_k(a->test);
  p.scale = a->scale_dev;
  p.srowptr = pat->srowptr;
  p.values = a->values_dev;
  p.store = (a->flags & NH_MATRIX_STORE) != 0;
  p.nrows = pat->nrows;
  p.R = o->rows_per_block, p.nsteps = o->nsteps, p.vmax = o->max_visits, p.ldst = ldst, p.rows16 = o->rows16, p.qc = qc;
  if (xlds && (!o->vvert || o->vvert_src != (const void *)a->geom.gdofs_dev)) {  // (the connectivity of this call: remade when another array comes)
    nh_owner_plan *ow = pat->owner;
    const i64 n = o->nvisits * (1ll << a->ndims);
    if (!ow->vvert) NH_CHECK_HIP(hipMalloc((void **)&ow->vvert, sizeof(int32_t) * (size_t)std::max<i64>(n, 1)));
    hipLaunchKernelGGL(k_op_vvert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, 1 << a->ndims, o->vlist, a->geom.gdofs_dev, ow->vvert);
    NH_CHECK_HIP(hipGetLastError());
    ow->vvert_src = (const void *)a->geom.gdofs_dev;
  }
  p.prs = o->prs, p.prl = o->prl, p.vvert = o->vvert;
  p.vlist = o->vlist, p.vptr = o->vptr, p.cptr = o->cptr, p.bptr = o->bptr, p.isrc = o->isrc, p.idst = o->idst;
  // threads: enough waves for the chunks of a block, and for the latencies of phase 1 when one block takes most of a CU's LDS
  int nt = lds > 80 * 1024 ? 1024 : lds > 52 * 1024 ? 512 : 256;
  if (getenv("NH_OWNER_NT")) nt = std::max(64, std::min(OWN_NT_MAX, atoi(getenv("NH_OWNER_NT")) & ~63));
  if (!xlds) nt = std::min(nt, OWN_NT_MAX / 2);
  if (big) nt = 256;
#ifdef NH_ABLATION
  if (getenv("NH_OWNER_DEBUG")) p.debug = atoi(getenv("NH_OWNER_DEBUG"));
  if (getenv("NH_OWNER_TICKS")) {
    NH_CHECK_HIP(hipMalloc((void **)&p.tdbg, 4 * sizeof(unsigned long long)));
    NH_CHECK_HIP(hipMemsetAsync(p.tdbg, 0, 4 * sizeof(unsigned long long), s));
  }
#endif
  dim3 grid((unsigned)o->nblocks), block(nt);
#define OWN3(ND, NB, NC, IS, U0, XL)                                                                                                          \
  do {                                                                                                                                        \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_owner_rows_v<ND, NB, NC, IS, U0, XL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_owner_rows_v<ND, NB, NC, IS, U0, XL>), grid, block, lds, s, p);                                                      \
  } while (0)
#define OWN2(ND, NB, NC, IS, U0)             \
  do {                                       \
    if (xlds) OWN3(ND, NB, NC, IS, U0, true); \
    else OWN3(ND, NB, NC, IS, U0, false);    \
  } while (0)
#define OWN(ND, NB, NC)                              \
  do {                                               \
    if (isof && ND == NC) OWN2(ND, NB, NC, (ND == NC), false); \
    else if (use0) OWN2(ND, NB, NC, false, true);    \
    else OWN2(ND, NB, NC, false, false);             \
  } while (0)
#define OWNG(ND, NB, NC, IS, U0)                                                                                                                   \
  do {                                                                                                                                            \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_owner_rows_v<ND, NB, NC, IS, U0, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_owner_rows_v<ND, NB, NC, IS, U0, true, 4>), grid, block, lds, s, p);                                                      \
  } while (0)
  switch (key) {
    case 3273:  // triquadratic hexahedra, 3 components: points in chunks (a multilinear isoparametric geometry: with staged vertices only)
      if (iso && !xlds) return NH_OK;
      if (isof) OWNG(3, 27, 3, true, false);
      else if (use0) OWNG(3, 27, 3, false, true);
      else OWNG(3, 27, 3, false, false);
      break;
    case 3083: OWN(3, 8, 3); break;  // trilinear hexahedra, 3 components
    case 2042: OWN(2, 4, 2); break;  // bilinear quadrilaterals, 2 components
    case 2092: OWN(2, 9, 2); break;  // biquadratic quadrilaterals / quadratic splines, 2 components
  }
#undef OWN
#undef OWN2
#undef OWN3
#undef OWNG
  NH_LAUNCH_CHECK();
#ifdef NH_ABLATION
  if (p.tdbg) {
    unsigned long long h[4];
    NH_CHECK_HIP(hipStreamSynchronize(s));
    NH_CHECK_HIP(hipMemcpy(h, p.tdbg, sizeof h, hipMemcpyDeviceToHost));
    hipFree(p.tdbg);
    fprintf(stderr, "nh_owner ticks per block (wave 0): staging %.0f, element phase %.0f, barrier %.0f, entries %.0f\n", (double)h[0] / o->nblocks, (double)h[1] / o->nblocks,
            (double)h[2] / o->nblocks, (double)h[3] / o->nblocks);
  }
#endif
  *done = true;
  return NH_OK;
}
