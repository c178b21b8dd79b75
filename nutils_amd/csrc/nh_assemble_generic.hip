// Generic element kernels: one wavefront (64 lanes) per element, any basis size,
// 1-3 dimensions, constant coefficient tensor.  Correct for every configuration the
// host layer can describe (ragged bases, several tables, mixed test/trial bases,
// isoparametric or box geometry); the specialised kernels (nh_assemble_p1.hip,
// nh_assemble_mfma.hip) overtake it on the headline configurations.
//
// Stages per element (LDS-staged, separated by wave-level barriers):
//   1. lanes over q:      J_q, J_q^{-1}, w_q |det J_q|            (K3)
//   2. lanes over (q,m):  D[q][m][0] = N_m, D[q][m][1+i] = sum_j dN_m/dxi_j Jinv[j][i]   (K2 tables -> physical)
//   3. lanes over entries (m,c,n,d): A = sum_q wdet sum_ab Dt C Dr                         (K4)
//      atomicAdd into values[slot(e,m,c,n,d)] via the element map                       (K5)
#include "nh_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstddef>
#include <vector>

namespace {

constexpr int MAXC = 4;            // components
constexpr int MAXS = 4;            // 1 + ndims
constexpr int LDS_BUDGET = 60000;  // bytes of D tables per workgroup (q-chunked beyond that)

struct FormK {
  int nct, ncr;
  int formd;                            // doubles of this struct that are staged in LDS: the header and the USED part of C (the
                                        // one-wave workgroups are latency bound: every kB of LDS is occupancy)
  int hasC, hasf;
  int cnt[MAXC], cum[MAXC], tot;
  unsigned char mask[MAXC][MAXC];
  signed char dpos[MAXC][MAXC];
  double f[MAXC * MAXS];                // [c][a]
  double C[MAXC * MAXS * MAXC * MAXS];  // [c][a][d][b], nct * S * ncr * S entries used; LAST member
};

__device__ __forceinline__ const FormK &stage_form(double *lds, const FormK &arg, int lane) {
  const double *src = reinterpret_cast<const double *>(&arg);
  for (int i = lane; i < arg.formd; i += (int)blockDim.x) lds[i] = src[i];
  __syncthreads();
  return *reinterpret_cast<const FormK *>(lds);
}

#include "nh_geom.inc"

__device__ __forceinline__ i64 boff(const BasisK &b, i64 e) { return b.off ? b.off[e] : e * (i64)b.nb; }
__device__ __forceinline__ int bnb(const BasisK &b, i64 e) { return b.off ? (int)(b.off[e + 1] - b.off[e]) : b.nb; }
__device__ __forceinline__ i64 bfn(const BasisK &b, i64 e) { return b.off ? b.off[e] : (b.tab ? (i64)b.tab[e] * b.nb : 0); }

// stage 2 helper: fill D[(q - q0)][m][S] for q in [q0, q1)
template <int ND>
__device__ __forceinline__ void fill_D(double *D, const BasisK &b, i64 e, int nb, int nq, int q0, int q1, const double *Jw, int lane) {
  constexpr int S = 1 + ND, JW = ND * ND + 1;
  const i64 fn0 = bfn(b, e);
  const int n = (q1 - q0) * nb;
  for (int t = lane; t < n; t += (int)blockDim.x) {
    const int ql = t / nb, m = t % nb, q = q0 + ql;
    const double *T = b.T + ((fn0 + m) * nq + q) * S;
    const double *Ji = Jw + q * JW;
    double *o = D + (ql * nb + m) * S;
    o[0] = T[0];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      double s = 0;
#pragma unroll
      for (int j = 0; j < ND; ++j) s += T[1 + j] * Ji[j * ND + i];
      o[1 + i] = s;
    }
  }
}

// NH_MATRIX_FIRST_TOUCH: which entries of an element of a parity colouring has no element of an EARLIER colour touched?  Colours are
// launched in lexicographic order of the index parities, so of the elements that share a face, edge or corner the all-even one comes
// first: the entry (m, n) was touched before iff nodes m and n both lie on a face of this element, along an axis on which its index is odd,
// whose neighbour exists.
struct FirstTouch {
  int on, p1;
  int shape[3];
};
template <int ND>
__device__ __forceinline__ void ft_element(const FirstTouch &f, i64 e, int &lo, int &hi) {  // axes whose low / high face was touched earlier
  lo = hi = 0;
#pragma unroll
  for (int d = ND - 1; d >= 0; --d) {
    const int ed = (int)(e % f.shape[d]);
    e /= f.shape[d];
    if (ed & 1) {
      lo |= 1 << d;
      if (ed < f.shape[d] - 1) hi |= 1 << d;
    }
  }
}
template <int ND>
__device__ __forceinline__ int ft_node(const FirstTouch &f, int m) {  // bit d: node on the low face of axis d, bit 3 + d: on the high face
  int mask = 0;
#pragma unroll
  for (int d = ND - 1; d >= 0; --d) {
    const int pd = m % f.p1;
    m /= f.p1;
    if (pd == 0) mask |= 1 << d;
    if (pd == f.p1 - 1) mask |= 8 << d;
  }
  return mask;
}

struct MatK {
  FirstTouch ft;
  i64 nelems;
  const int32_t *elist;
  int nq;
  const double *weights;
  GeomK geom;
  BasisK test, trial;
  int same;  // trial tables alias test tables
  const i64 *srowptr;
  const int32_t *emap;
  const i64 *eoff;
  double *values;
  const double *scale;
  int qchunk;
  int maxnbt, maxnbr;
  int emap_by_elem;
  const double *cq;  // per-point coefficient tensors [nelems][nq][nct][S][ncr][S], or NULL
  int use_w;      // pre-multiplied trial table W in LDS (pays when it does not cost occupancy)
  int exclusive;  // NH_MATRIX_EXCLUSIVE: no two elements of this launch share a matrix entry -> plain read-modify-write (deterministic)
  double *local;  // NH_MATRIX_GATHER: element-major local matrices [emap position][nct * ncr] instead of the scatter
  int sym;        // Gram path: test == trial (tables AND dofs) and C[c][a][d][b] == C[d][b][c][a]: the node pairs m >= n only, each written to both of its places
                  // (2 with NH_MATRIX_GATHER: to its own place only -- the gather map for symmetric producers mirrors it; 3: triangular scratch, nh_gram_sym.inc)
  const unsigned char *trirank;  // sym == 3: position of a node among the dofs of its element, [sum nb_e]
  const i64 *tribase;            // sym == 3: first packed node pair of an element, [nelems + 1]
};

template <int ND>
__global__ __launch_bounds__(256) void k_matrix_generic(MatK p, FormK formarg) {  // one, two or four waves per element (the launcher: two when an element has work for them)
  constexpr int S = 1 + ND, JW = ND * ND + 1;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const FormK &form = stage_form(lds, formarg, threadIdx.x);
  double *Jw = lds + formarg.formd;                   // [nq][JW]
  double *Dt = Jw + p.nq * JW;                        // [qchunk][maxnbt][S]
  double *Dr = p.same ? Dt : Dt + p.qchunk * p.maxnbt * S;
  double *W = Dr + p.qchunk * p.maxnbr * S;            // [qchunk][maxnbr][ncr][nct][S]
  const int lane = threadIdx.x, NTG = (int)blockDim.x;
  for (i64 ie = blockIdx.x; ie < p.nelems; ie += gridDim.x) {
    const i64 e = p.elist ? p.elist[ie] : ie;
    const int nbt = bnb(p.test, e), nbr = bnb(p.trial, e);
    int ftlo = 0, fthi = 0;
    if (p.ft.on) ft_element<ND>(p.ft, e, ftlo, fthi);
    for (int q = lane; q < p.nq; q += NTG) {
      double Ji[ND][ND], det;
      geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
#pragma unroll
      for (int j = 0; j < ND; ++j)
#pragma unroll
        for (int i = 0; i < ND; ++i) Jw[q * JW + j * ND + i] = Ji[j][i];
      Jw[q * JW + ND * ND] = p.weights[q] * fabs(det) * (p.scale ? p.scale[(p.emap_by_elem ? e : ie) * p.nq + q] : 1.);
    }
    __syncthreads();
    const i64 tdof0 = boff(p.test, e);
    const i64 emap0 = p.eoff ? p.eoff[e] : (p.emap_by_elem ? e : ie) * (i64)nbt * nbr;  // list position unless NH_MATRIX_EMAP_BY_ELEMENT
    const int nentries = nbt * form.nct * nbr * form.ncr;
    for (int q0 = 0; q0 < p.nq; q0 += p.qchunk) {
      const int q1 = min(p.nq, q0 + p.qchunk);
#ifndef NH_GEN_SKIPFILL
      fill_D<ND>(Dt, p.test, e, nbt, p.nq, q0, q1, Jw, lane);
      if (!p.same) fill_D<ND>(Dr, p.trial, e, nbr, p.nq, q0, q1, Jw, lane);
#endif
      __syncthreads();
      // trial side pre-multiplied by the form and the quadrature weight, once per (q, n, d, c):
      //   W[q][n][d][c][a] = w_q |J_q| sum_b C[c][a][d][b] Dr[q][n][b]
      // so that an entry costs S multiply-adds per point instead of S*S
      if (p.use_w) {  // (never set for the blocks that take the Gram path below)
        const int ncd = form.ncr * form.nct;
        for (int t = lane; t < (q1 - q0) * nbr * ncd; t += NTG) {
          int r = t, c = 0, d = 0;
          if (ncd > 1) {  // (scalar blocks: one division per item instead of three)
            c = r % form.nct; r /= form.nct;
            d = r % form.ncr; r /= form.ncr;
          }
          const int ql = r / nbr, n = r - ql * nbr;
          const double *dr = Dr + (ql * nbr + n) * S;
          const double wq = Jw[(q0 + ql) * JW + ND * ND];
          double *w = W + (size_t)t * S;
          const int coff = ((c * S) * form.ncr + d) * S;  // C[c][a][d][b] = Cc[a*ncr*S + b]
          auto fill = [&](const double *Cc) {  // (two call sites: LDS-resident constant form / per-point tensors in global memory)
#pragma unroll
            for (int a = 0; a < S; ++a) {
              double tt = 0;
#pragma unroll
              for (int b = 0; b < S; ++b) tt += Cc[a * form.ncr * S + b] * dr[b];
              w[a] = wq * tt;
            }
          };
          if (p.cq) fill(p.cq + ((p.emap_by_elem ? e : ie) * (i64)p.nq + (q0 + ql)) * (form.nct * S * form.ncr * S) + coff);
          else fill(form.C + coff);
        }
        __syncthreads();
      }
      // Vector-valued blocks with a constant form: the quadrature sum of a node pair does not depend on the components -- G[a][b] = sum_q w_q |J_q| Dt[q][m][a] Dr[q][n][b]
      // (S x S accumulators per lane, 2 S + 1 LDS reads and S (S + 1) multiply-adds per point) and the form tensor is applied once per pair,
      // K[c][d] = sum_ab C[c][a][d][b] G[a][b], instead of S (S + 1) multiply-adds and 2 S + S S reads per point for EACH of the nct x ncr entries.
      const bool gram = !p.cq && form.nct * form.ncr > 1;
      if (gram && p.sym) {
        // symmetric blocks (test == trial, C[c][a][d][b] == C[d][b][c][a]): G_nm[a][b] = G_mn[b][a] and K_nm = K_mn^T -- the node pairs m >= n only (nb (nb + 1) / 2 of
        // nb^2), two of them per lane, each written to its place and (m != n) transposed to the mirrored one: half the quadrature sums of the kernel's longest phase
        const int np = nbt * (nbt + 1) / 2, hp = (np + 1) >> 1;
        for (int k = lane; k < hp; k += NTG) {
          int mm[2], nn[2];
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            const int kk = min(k + pp * hp, np - 1);
            int m = (int)((sqrt(8. * kk + 1.) - 1.) * .5);  // kk = m (m + 1) / 2 + n, 0 <= n <= m
            m += (m + 1) * (m + 2) / 2 <= kk;
            m -= m * (m + 1) / 2 > kk;
            mm[pp] = m, nn[pp] = kk - m * (m + 1) / 2;
          }
          const bool two = k + hp < np;
          double G[2][S][S];
#pragma unroll
          for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int a = 0; a < S; ++a)
#pragma unroll
              for (int b = 0; b < S; ++b) G[pp][a][b] = 0.;
#ifdef NH_GEN_SKIPQ  // (timing experiments)
          for (int q = q0; q < min(q1, q0 + 1); ++q) {
#else
          for (int q = q0; q < q1; ++q) {
#endif
            const double *Dq = Dt + (q - q0) * nbt * S;
            const double wq = Jw[q * JW + ND * ND];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
              const double *dt = Dq + mm[pp] * S, *dr = Dq + nn[pp] * S;
              double d0[S];
#pragma unroll
              for (int b = 0; b < S; ++b) d0[b] = dr[b];
#pragma unroll
              for (int a = 0; a < S; ++a) {
                const double wa = wq * dt[a];
#pragma unroll
                for (int b = 0; b < S; ++b) G[pp][a][b] += wa * d0[b];
              }
            }
          }
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            if (pp && !two) break;
            const int m = mm[pp], n = nn[pp];
            i64 rowm = 0, a0m = 0, lenm = 0, posm = 0, rown = 0, a0n = 0, lenn = 0, posn = 0;
            if (!p.local) {
              rowm = p.test.dofs[tdof0 + m], a0m = p.srowptr[rowm], lenm = p.srowptr[rowm + 1] - a0m, posm = p.emap[emap0 + m * nbr + n];
              rown = p.test.dofs[tdof0 + n], a0n = p.srowptr[rown], lenn = p.srowptr[rown + 1] - a0n, posn = p.emap[emap0 + n * nbr + m];
            }
            for (int c = 0; c < form.nct; ++c)
              for (int d = 0; d < form.ncr; ++d) {
                if (!form.mask[c][d]) continue;
                const double *Cc = form.C + ((c * S) * form.ncr + d) * S;
                double acc = 0;
#pragma unroll
                for (int a = 0; a < S; ++a)
#pragma unroll
                  for (int b = 0; b < S; ++b) acc += Cc[a * form.ncr * S + b] * G[pp][a][b];
#ifdef NH_GEN_SKIPST
                if (acc == 1.2345e300) p.local[0] = acc;
                continue;
#endif
                if (p.local) {
                  double *dst = p.local + (emap0 + m * nbr + n) * (form.nct * form.ncr) + c * form.ncr + d;
                  *dst = q0 ? *dst + acc : acc;
                  if (m != n && p.sym != 2) {  // (sym == 2: the gather reads the (n, m) entries from this block, transposed)
                    double *dst2 = p.local + (emap0 + n * nbr + m) * (form.nct * form.ncr) + d * form.ncr + c;
                    *dst2 = q0 ? *dst2 + acc : acc;
                  }
                  continue;
                }
                atomicAdd(p.values + a0m * form.tot + lenm * form.cum[c] + posm * form.cnt[c] + form.dpos[c][d], acc);
                if (m != n) atomicAdd(p.values + a0n * form.tot + lenn * form.cum[d] + posn * form.cnt[d] + form.dpos[d][c], acc);
              }
          }
        }
      } else if (gram) {
        // a lane takes test function m and TWO trial functions n, n + h (h = half the trial functions): the test row of a point is read once for both
        // (63 488 ragged elements 1.77 -> 1.67 ms; two test functions as well: 1.64 ms, not worth the code)
        const int h = (nbr + 1) >> 1;
        for (int k = lane; k < nbt * h; k += NTG) {
          const int m = k / h, n0 = k - m * h, n1 = n0 + h;
          const bool two = n1 < nbr;
          double G[2][S][S];
#pragma unroll
          for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int a = 0; a < S; ++a)
#pragma unroll
              for (int b = 0; b < S; ++b) G[pp][a][b] = 0.;
          for (int q = q0; q < q1; ++q) {
            const double *dt = Dt + ((q - q0) * nbt + m) * S;
            const double *dr0 = Dr + ((q - q0) * nbr + n0) * S, *dr1 = Dr + ((q - q0) * nbr + (two ? n1 : n0)) * S;
            const double wq = Jw[q * JW + ND * ND];
            double d0[S], d1[S];
#pragma unroll
            for (int b = 0; b < S; ++b) d0[b] = dr0[b], d1[b] = dr1[b];
#pragma unroll
            for (int a = 0; a < S; ++a) {
              const double wa = wq * dt[a];
#pragma unroll
              for (int b = 0; b < S; ++b) G[0][a][b] += wa * d0[b], G[1][a][b] += wa * d1[b];
            }
          }
          const i64 row = p.test.dofs[tdof0 + m];
          const i64 a0 = p.srowptr[row], len = p.srowptr[row + 1] - a0;
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            if (pp && !two) break;
            const int n = pp ? n1 : n0;
            const i64 pos = p.emap[emap0 + m * nbr + n];
            const int both = p.exclusive ? ft_node<ND>(p.ft, m) & ft_node<ND>(p.ft, n) : 0;
            const bool first = p.exclusive && p.ft.on && q0 == 0 && !((both & ftlo) | ((both >> 3) & fthi));
            for (int c = 0; c < form.nct; ++c)
              for (int d = 0; d < form.ncr; ++d) {
                if (!form.mask[c][d]) continue;
                const double *Cc = form.C + ((c * S) * form.ncr + d) * S;  // C[c][a][d][b] = Cc[a * ncr * S + b]
                double acc = 0;
#pragma unroll
                for (int a = 0; a < S; ++a)
#pragma unroll
                  for (int b = 0; b < S; ++b) acc += Cc[a * form.ncr * S + b] * G[pp][a][b];
                if (p.local) {
                  double *dst = p.local + (emap0 + m * nbr + n) * (form.nct * form.ncr) + c * form.ncr + d;
                  *dst = q0 ? *dst + acc : acc;
                  continue;
                }
                const i64 slot = a0 * form.tot + len * form.cum[c] + pos * form.cnt[c] + form.dpos[c][d];
                if (p.exclusive) p.values[slot] = first ? acc : p.values[slot] + acc;
                else atomicAdd(p.values + slot, acc);
              }
          }
        }
      } else if (form.nct * form.ncr == 1 && !p.cq && p.sym && !p.exclusive) {
        // scalar symmetric blocks (test == trial, C[a][b] == C[b][a]): the entries m >= n only, mirrored on the way out
        const int np = nbt * (nbt + 1) / 2;
        for (int k = lane; k < np; k += NTG) {
          int m = (int)((sqrt(8. * k + 1.) - 1.) * .5);  // k = m (m + 1) / 2 + n, 0 <= n <= m
          m += (m + 1) * (m + 2) / 2 <= k;
          m -= m * (m + 1) / 2 > k;
          const int n = k - m * (m + 1) / 2;
          double acc = 0;
          if (p.use_w) {
            for (int q = q0; q < q1; ++q) {
              const double *dt = Dt + ((q - q0) * nbt + m) * S, *w = W + ((size_t)(q - q0) * nbr + n) * S;
#pragma unroll
              for (int a = 0; a < S; ++a) acc += dt[a] * w[a];
            }
          } else {
            for (int q = q0; q < q1; ++q) {
              const double *dt = Dt + ((q - q0) * nbt + m) * S, *dr = Dr + ((q - q0) * nbr + n) * S;
              double sq = 0;
#pragma unroll
              for (int a = 0; a < S; ++a) {
                double t = 0;
#pragma unroll
                for (int b = 0; b < S; ++b) t += form.C[a * S + b] * dr[b];
                sq += dt[a] * t;
              }
              acc += Jw[q * JW + ND * ND] * sq;
            }
          }
          if (p.local) {
            double *dst = p.local + (emap0 + m * nbr + n), *dst2 = p.local + (emap0 + n * nbr + m);
            *dst = q0 ? *dst + acc : acc;
            if (m != n) *dst2 = q0 ? *dst2 + acc : acc;
            continue;
          }
          const i64 rowm = p.test.dofs[tdof0 + m], rown = p.test.dofs[tdof0 + n];
          atomicAdd(p.values + p.srowptr[rowm] + p.emap[emap0 + m * nbr + n], acc);
          if (m != n) atomicAdd(p.values + p.srowptr[rown] + p.emap[emap0 + n * nbr + m], acc);
        }
      } else
      for (int k = lane; k < nentries; k += NTG) {
        int r = k;
        const int d = r % form.ncr; r /= form.ncr;
        const int n = r % nbr; r /= nbr;
        const int c = r % form.nct;
        const int m = r / form.nct;
        if (!form.mask[c][d]) continue;
        double acc = 0;
        if (p.use_w) {
          for (int q = q0; q < q1; ++q) {
            const double *dt = Dt + ((q - q0) * nbt + m) * S;
            const double *w = W + ((((size_t)(q - q0) * nbr + n) * form.ncr + d) * form.nct + c) * S;
#pragma unroll
            for (int a = 0; a < S; ++a) acc += dt[a] * w[a];
          }
        } else {
          const int coff = ((c * S) * form.ncr + d) * S;  // C[c][a][d][b] = Cc[a*ncr*S + b]
          auto point = [&](const double *Cc, int q) {
            const double *dt = Dt + ((q - q0) * nbt + m) * S;
            const double *dr = Dr + ((q - q0) * nbr + n) * S;
            double sq = 0;
#pragma unroll
            for (int a = 0; a < S; ++a) {
              double t = 0;
#pragma unroll
              for (int b = 0; b < S; ++b) t += Cc[a * form.ncr * S + b] * dr[b];
              sq += dt[a] * t;
            }
            acc += Jw[q * JW + ND * ND] * sq;
          };
          if (p.cq)
            for (int q = q0; q < q1; ++q) point(p.cq + ((p.emap_by_elem ? e : ie) * (i64)p.nq + q) * (form.nct * S * form.ncr * S) + coff, q);
          else
            for (int q = q0; q < q1; ++q) point(form.C + coff, q);
        }
        if (p.local) {
          double *dst = p.local + (emap0 + m * nbr + n) * (form.nct * form.ncr) + c * form.ncr + d;
          *dst = q0 ? *dst + acc : acc;
          continue;
        }
        const i64 row = p.test.dofs[tdof0 + m];
        const i64 a0 = p.srowptr[row], len = p.srowptr[row + 1] - a0;
        const i64 slot = a0 * form.tot + len * form.cum[c] + (i64)p.emap[emap0 + m * nbr + n] * form.cnt[c] + form.dpos[c][d];
        if (p.exclusive) {
          const int both = ft_node<ND>(p.ft, m) & ft_node<ND>(p.ft, n);
          const bool first = p.ft.on && q0 == 0 && !((both & ftlo) | ((both >> 3) & fthi));
          p.values[slot] = first ? acc : p.values[slot] + acc;
        } else
          atomicAdd(p.values + slot, acc);
      }
      __syncthreads();
    }
  }
}

#include "nh_gram_sym.inc"

// ---------------------------------------------------------------------------------------------------------------------
// MFMA path for large local matrices (p >= 2, vector fields): one WORKGROUP (4 waves) per element, the local contraction as
// GEMMs on v_mfma_f64_16x16x4_f64.  For each test component c:
//     A_c[m][(n,d)] = sum_{k=(q,a)} Dt[q][m][slot_a] * H_c[k][(n,d)],   H_c[k][(n,d)] = w|J|_q sum_b C[c,slot_a,d,b] Dr[q][n][b]
// M = nbt, N = nbr*ncr, K = nq * (number of ACTIVE test slots: those a with a nonzero coefficient).  The (c, N-tile) work
// items are dealt round-robin to the 4 waves; a wave keeps MT accumulators (16 VGPRs), forms its B fragment H on the VALU from
// the LDS-resident D table (shared by the 4 waves) while the matrix pipe runs, and feeds all M tiles with it.  The small
// register footprint lets 6 workgroups (24 waves) share a CU, which hides the latency of the read-modify-write scatter.
// Operand layout of the f64 MFMA: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[row = (lane >> 4) + 4 r][col = lane & 15], r = 0..3.
// Scatter: with NH_MATRIX_EXCLUSIVE (the caller launches one colour of elements that share no dof) the result is added with
// plain loads/stores -- deterministic, at HBM rate -- otherwise with f64 atomics.
typedef double v4d __attribute__((ext_vector_type(4)));

struct MfmaX {
  int nas;         // active test slots
  int aslot[4];
  int mt, nt, kt;  // tiles: ceil(nbt/16), ceil(nbr*ncr/16), ceil(nq*nas/4)
  int flags;
};

template <int ND>
__device__ __forceinline__ void fill_D_wg(double *D, const BasisK &b, i64 e, int nb, int nq, const double *Jw, int tid, int nthreads) {
  constexpr int S = 1 + ND, JW = ND * ND + 1;
  const i64 fn0 = bfn(b, e);
  const int n = nq * nb;
#pragma unroll 4
  for (int t = tid; t < n; t += nthreads) {
    const int q = t / nb, m = t - q * nb;
    const double *T = b.T + ((fn0 + m) * nq + q) * S;
    const double *Ji = Jw + q * JW;
    double *o = D + t * S;
    o[0] = T[0];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      double sum = 0;
#pragma unroll
      for (int jj = 0; jj < ND; ++jj) sum += T[1 + jj] * Ji[jj * ND + i];
      o[1 + i] = sum;
    }
  }
}

template <int ND, int MT>
__global__ __launch_bounds__(256) void k_matrix_mfma(MatK p, FormK formarg, MfmaX x) {
  constexpr int S = 1 + ND, JW = ND * ND + 1;
  extern __shared__ __attribute__((aligned(32))) double lds[];
  const FormK &form = stage_form(lds, formarg, threadIdx.x);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nb = p.test.nb;  // test and trial share tables on this path (p.same)
  const int nqp = (p.nq + 3) & ~3;                 // q padded to the MFMA k-step: K index k = a * nqp + q (slot-major)
  const int nbp = MT * 16;                         // m padded to the M tiles
  double *Jw = lds + formarg.formd;                // [nq][JW]
  double *D = Jw + ((p.nq * JW + 3) & ~3);         // [nq][nb][S]   (32-byte aligned rows: B fragments are read as 4 doubles)
  double *At = D + (size_t)p.nq * nb * S;          // [nas][nqp][nbp] transposed copy of the active test slots, zero padded:
                                                   // the A fragment read A[m = li][k = lk] is conflict-free and branch-free
  const int li = lane & 15, lk = lane >> 4;
  const int N = p.trial.nb * form.ncr;
  const int kq = nqp >> 2;                         // k-steps per slot
  for (i64 ie = blockIdx.x; ie < p.nelems; ie += gridDim.x) {
    const i64 e = p.elist ? p.elist[ie] : ie;
    for (int q = tid; q < p.nq; q += 256) {
      double Ji[ND][ND], det;
      geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
#pragma unroll
      for (int j = 0; j < ND; ++j)
#pragma unroll
        for (int i = 0; i < ND; ++i) Jw[q * JW + j * ND + i] = Ji[j][i];
      Jw[q * JW + ND * ND] = p.weights[q] * fabs(det) * (p.scale ? p.scale[((x.flags & 2) ? e : ie) * p.nq + q] : 1.);
    }
    __syncthreads();
    fill_D_wg<ND>(D, p.test, e, nb, p.nq, Jw, tid, 256);
    __syncthreads();
    for (int a = 0; a < x.nas; ++a) {
      const int sa = x.aslot[a];
#pragma unroll 4
      for (int t = tid; t < nqp * nbp; t += 256) {
        const int q = t / nbp, m = t - q * nbp;  // nbp is a compile-time constant
        At[a * nqp * nbp + t] = (m < nb && q < p.nq) ? D[(q * nb + m) * S + sa] : 0.;
      }
    }
    __syncthreads();
    const i64 tdof0 = e * (i64)nb;
    const i64 emap0 = ((x.flags & 2) ? e : ie) * (i64)nb * p.trial.nb;
    int ftlo = 0, fthi = 0;
    if (p.ft.on) ft_element<ND>(p.ft, e, ftlo, fthi);
    // per-lane row bookkeeping: this lane owns rows (lk + 4 r) of every M tile
    i64 rbase[MT][4];
    int rlen[MT][4], rft[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mt * 16 + lk + 4 * r;
        rbase[mt][r] = -1;
        rlen[mt][r] = 0;
        rft[mt][r] = p.ft.on ? ft_node<ND>(p.ft, m) : 0;
        if (m < nb) {
          const i64 row = p.test.dofs[tdof0 + m];
          rbase[mt][r] = p.srowptr[row];
          rlen[mt][r] = (int)(p.srowptr[row + 1] - rbase[mt][r]);
        }
      }
    for (int wi = wave; wi < form.nct * x.nt; wi += 4) {
      const int c = wi / x.nt, nt = wi - c * x.nt;
      const int j = nt * 16 + li;
      const int ncol = j < N ? j / form.ncr : -1, dcol = j < N ? j - ncol * form.ncr : 0;
      v4d acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = v4d{0., 0., 0., 0.};
      for (int a = 0; a < x.nas; ++a) {
        // coefficients of this lane's column for the (wave-uniform) test slot: registers, not LDS, inside the k loop
        double Cc[S];
        const double *Cp = form.C + ((c * S + x.aslot[a]) * form.ncr + dcol) * S;
#pragma unroll
        for (int bb = 0; bb < S; ++bb) Cc[bb] = ncol >= 0 ? Cp[bb] : 0.;
        const double *Aa = At + (size_t)a * nqp * nbp;
        #ifdef NH_ABLATION
        const int kend = (x.flags & 16) ? 1 : kq;
#else
        const int kend = kq;
#endif
        for (int ks = 0; ks < kend; ++ks) {
          const int q = ks * 4 + lk;
          const int qq = q < p.nq ? q : p.nq - 1;
          const double wq = q < p.nq ? Jw[qq * JW + ND * ND] : 0.;
          const double *dr = D + ((size_t)qq * nb + (ncol >= 0 ? ncol : 0)) * S;
          double b = 0.;
#pragma unroll
          for (int bb = 0; bb < S; ++bb) b += Cc[bb] * dr[bb];
          b *= wq;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aa[q * nbp + mt * 16 + li], b, acc[mt], 0, 0, 0);
        }
      }
      // scatter this 16-column slab: all loads of the old values first (one memory latency), then add + store
      #ifdef NH_ABLATION
      const bool skip_scatter = (x.flags & 8) && acc[0][0] != 1.2345e300;
#else
      constexpr bool skip_scatter = false;
#endif
      if (ncol >= 0 && form.mask[c][dcol] && !skip_scatter) {
        i64 slot[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            slot[mt][r] = -1;
            if (rbase[mt][r] >= 0) {
              const int m = mt * 16 + lk + 4 * r;
              slot[mt][r] = rbase[mt][r] * form.tot + (i64)rlen[mt][r] * form.cum[c] + (i64)p.emap[emap0 + m * p.trial.nb + ncol] * form.cnt[c] + form.dpos[c][dcol];
            }
          }
        if (x.flags & 1) {
          double old[MT][4];
          const int cft = p.ft.on ? ft_node<ND>(p.ft, ncol) : 0;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int both = rft[mt][r] & cft;
              const bool first = p.ft.on && !((both & ftlo) | ((both >> 3) & fthi));
              old[mt][r] = slot[mt][r] >= 0 && !first ? p.values[slot[mt][r]] : 0.;
            }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (slot[mt][r] >= 0) p.values[slot[mt][r]] = old[mt][r] + acc[mt][r];
        } else {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (slot[mt][r] >= 0) atomicAdd(p.values + slot[mt][r], acc[mt][r]);
        }
      }
    }
    __syncthreads();
  }
}

struct VecK {
  i64 nelems;
  const int32_t *elist;
  int nq;
  const double *weights;
  GeomK geom;
  BasisK test, trial;
  int same;
  const double *u;
  double *out;
  double *local;  // element-major local vectors instead of atomics into out (deterministic scatter, nh_scatter.hip)
  double f0;
  double *out_scalar;
  const double *scale;
  int qchunk, maxnbt, maxnbr;
  int cs;  // component stride of the U / F tables = max(nct, ncr): the one-wave workgroups are latency bound, LDS decides occupancy
};

template <int ND>
__global__ __launch_bounds__(64) void k_vector_generic(VecK p, FormK formarg) {
  constexpr int S = 1 + ND, JW = ND * ND + 1;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const FormK &form = stage_form(lds, formarg, threadIdx.x);
  double *Jw = lds + formarg.formd;
  double *Dt = Jw + p.nq * JW;
  double *Dr = p.same ? Dt : Dt + p.qchunk * p.maxnbt * S;
  double *U = Dr + (p.same ? p.qchunk * p.maxnbt * S : p.qchunk * p.maxnbr * S);  // [qchunk][ncr][S]
  double *F = U + p.qchunk * p.cs * S;                                              // [qchunk][nct][S]
  const int lane = threadIdx.x;
  double fsum = 0;
  for (i64 ie = blockIdx.x; ie < p.nelems; ie += gridDim.x) {
    const i64 e = p.elist ? p.elist[ie] : ie;
    const int nbt = bnb(p.test, e), nbr = bnb(p.trial, e);
    for (int q = lane; q < p.nq; q += 64) {
      double Ji[ND][ND], det;
      geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
#pragma unroll
      for (int j = 0; j < ND; ++j)
#pragma unroll
        for (int i = 0; i < ND; ++i) Jw[q * JW + j * ND + i] = Ji[j][i];
      Jw[q * JW + ND * ND] = p.weights[q] * fabs(det) * (p.scale ? p.scale[ie * p.nq + q] : 1.);
    }
    __syncthreads();
    const i64 tdof0 = boff(p.test, e), rdof0 = boff(p.trial, e);
    for (int q0 = 0; q0 < p.nq; q0 += p.qchunk) {
      const int q1 = min(p.nq, q0 + p.qchunk), nql = q1 - q0;
      fill_D<ND>(Dt, p.test, e, nbt, p.nq, q0, q1, Jw, lane);
      if (!p.same) fill_D<ND>(Dr, p.trial, e, nbr, p.nq, q0, q1, Jw, lane);
      __syncthreads();
      // U[q][d][b]
      for (int t = lane; t < nql * form.ncr * S; t += 64) {
        const int b = t % S, d = (t / S) % form.ncr, ql = t / (S * form.ncr);
        double s = 0;
        if (p.u)
          for (int n = 0; n < nbr; ++n) s += Dr[(ql * nbr + n) * S + b] * p.u[(i64)p.trial.dofs[rdof0 + n] * form.ncr + d];
        U[(ql * p.cs + d) * S + b] = s;
      }
      __syncthreads();
      // F[q][c][a] = f[c][a] + sum_db C[c][a][d][b] U[q][d][b]
      for (int t = lane; t < nql * form.nct * S; t += 64) {
        const int a = t % S, c = (t / S) % form.nct, ql = t / (S * form.nct);
        double s = form.hasf ? form.f[c * S + a] : 0.;
        if (form.hasC)
          for (int d = 0; d < form.ncr; ++d)
            for (int b = 0; b < S; ++b) s += form.C[((c * S + a) * form.ncr + d) * S + b] * U[(ql * p.cs + d) * S + b];
        F[(ql * p.cs + c) * S + a] = s;
      }
      __syncthreads();
      if (p.out || p.local) {
        for (int k = lane; k < nbt * form.nct; k += 64) {
          const int c = k % form.nct, m = k / form.nct;
          double acc = 0;
          for (int ql = 0; ql < nql; ++ql) {
            double s = 0;
#pragma unroll
            for (int a = 0; a < S; ++a) s += Dt[(ql * nbt + m) * S + a] * F[(ql * p.cs + c) * S + a];
            acc += Jw[(q0 + ql) * JW + ND * ND] * s;
          }
          if (p.local) {  // (the same lane owns (m, c) in every chunk of points: plain accumulation)
            double *dst = p.local + (tdof0 + m) * form.nct + c;
            *dst = q0 ? *dst + acc : acc;
          } else
            atomicAdd(p.out + (i64)p.test.dofs[tdof0 + m] * form.nct + c, acc);
        }
      }
      if (p.out_scalar) {
        for (int ql = lane; ql < nql; ql += 64) {
          double s = p.f0;
          if (form.hasC) {
            // 1/2 U . (F - f)
            double h = 0;
            for (int c = 0; c < form.nct; ++c)
              for (int a = 0; a < S; ++a)
                h += U[(ql * p.cs + c) * S + a] * (F[(ql * p.cs + c) * S + a] - (form.hasf ? form.f[c * S + a] : 0.));
            s += .5 * h;
          }
          if (form.hasf) {
            for (int c = 0; c < form.nct; ++c)
              for (int a = 0; a < S; ++a) s += form.f[c * S + a] * U[(ql * p.cs + c) * S + a];
          }
          fsum += Jw[(q0 + ql) * JW + ND * ND] * s;
        }
      }
      __syncthreads();
    }
  }
  if (p.out_scalar) {
    for (int o = 32; o; o >>= 1) fsum += __shfl_xor(fsum, o);
    if (lane == 0) atomicAdd(p.out_scalar, fsum);
  }
}

struct EvalK {
  i64 nelems;
  const int32_t *elist;
  int nq;
  GeomK geom;
  BasisK trial;
  int ncr;
  const double *points;
  const double *u;
  double *x, *detj, *U;
};

template <int ND>
__global__ void k_sample_eval(EvalK p) {
  constexpr int S = 1 + ND;
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.nelems * p.nq) return;
  const i64 e = p.elist ? p.elist[t / p.nq] : t / p.nq;
  const int q = (int)(t % p.nq);
  double Ji[ND][ND], det, x[ND];
  geometry_at<ND>(p.geom, e, q, p.nq, p.points, Ji, det, x);
  if (p.x)
    for (int i = 0; i < ND; ++i) p.x[t * ND + i] = x[i];
  if (p.detj) p.detj[t] = fabs(det);
  if (p.U && p.u) {
    const int nb = bnb(p.trial, e);
    const i64 d0 = boff(p.trial, e), fn0 = bfn(p.trial, e);
    for (int d = 0; d < p.ncr; ++d) {
      double v = 0, g[ND];
      for (int i = 0; i < ND; ++i) g[i] = 0;
      for (int n = 0; n < nb; ++n) {
        const double *T = p.trial.T + ((fn0 + n) * p.nq + q) * S;
        const double un = p.u[(i64)p.trial.dofs[d0 + n] * p.ncr + d];
        v += T[0] * un;
        for (int j = 0; j < ND; ++j) g[j] += T[1 + j] * un;
      }
      double *o = p.U + (t * p.ncr + d) * S;
      o[0] = v;
      for (int i = 0; i < ND; ++i) {
        double s = 0;
        for (int j = 0; j < ND; ++j) s += g[j] * Ji[j][i];
        o[1 + i] = s;
      }
    }
  }
}

int make_form(int nd, int nct, int ncr, const double *C, const double *f, const unsigned char *mask, FormK *fk) {
  const int S = 1 + nd;
  NH_REQUIRE(nct >= 1 && nct <= MAXC && ncr >= 1 && ncr <= MAXC, "component counts must be 1..%d (got %d, %d)", MAXC, nct, ncr);
  memset(fk, 0, sizeof *fk);
  fk->nct = nct;
  fk->ncr = ncr;
  fk->hasC = C != nullptr;
  fk->hasf = f != nullptr;
  fk->formd = (int)(((offsetof(FormK, C) / 8 + (size_t)nct * S * ncr * S) + 3) & ~(size_t)3);  // 32-byte granules: the tables behind it are read as 4 doubles
  if (C) memcpy(fk->C, C, sizeof(double) * nct * S * ncr * S);
  if (f) memcpy(fk->f, f, sizeof(double) * nct * S);
  for (int c = 0; c < nct; ++c) {
    fk->cum[c] = fk->tot;
    for (int d = 0; d < ncr; ++d) {
      fk->mask[c][d] = mask ? (mask[c * ncr + d] != 0) : 1;
      fk->dpos[c][d] = (signed char)fk->cnt[c];
      fk->cnt[c] += fk->mask[c][d];
    }
    fk->tot += fk->cnt[c];
  }
  return NH_OK;
}

int check_geom(const nh_geometry &g) {
  if (g.kind == NH_GEOM_ISO) {
    NH_REQUIRE(g.ngb > 0 && g.gT_dev && g.gdofs_dev && g.verts_dev, "isoparametric geometry needs ngb, gT, gdofs, verts");
  } else if (g.kind == NH_GEOM_TAB) {
    NH_REQUIRE(g.jac_dev, "tabulated geometry needs jac_dev");
  } else if (g.kind == NH_GEOM_BOX) {
    NH_REQUIRE(g.origin_dev && g.size_dev, "box geometry needs origin and size");
  } else {
    nh_set_error("unknown geometry kind %d", g.kind);
    return NH_EINVAL;
  }
  return NH_OK;
}

int max_nb(const nh_basis &b, i64 nelems, int *out, hipStream_t s) {
  if (b.nb > 0 || !b.off_dev) {
    *out = b.nb;
    return NH_OK;
  }
  // ragged: scan the offsets on the host (called once per bucket; small)
  std::vector<i64> h(nelems + 1);
  // (on the caller's stream: offsets produced asynchronously on that stream are complete when read)
  NH_CHECK_HIP(hipMemcpyAsync(h.data(), b.off_dev, sizeof(i64) * (nelems + 1), hipMemcpyDeviceToHost, s));
  NH_CHECK_HIP(hipStreamSynchronize(s));
  i64 m = 0;
  for (i64 e = 0; e < nelems; ++e) m = std::max(m, h[e + 1] - h[e]);
  *out = (int)m;
  return NH_OK;
}

}  // namespace

static int grid_for(i64 nelems) { return (int)std::min<i64>(nelems, 256 * 32); }

// component-block layout of the expanded pattern for the kernels of nh_gather.hip / nh_owner.hip
static GSlots slots_of(const FormK &form) {
  GSlots gs;
  memset(&gs, 0, sizeof gs);
  gs.nct = form.nct, gs.ncr = form.ncr, gs.tot = form.tot;
  for (int c = 0; c < MAXC; ++c) {
    gs.cnt[c] = form.cnt[c], gs.cum[c] = form.cum[c];
    for (int d = 0; d < MAXC; ++d) gs.dpos[c][d] = form.dpos[c][d], gs.mask[c][d] = form.mask[c][d];
  }
  return gs;
}

extern "C" {

int nh_assemble_matrix(const nh_matrix_args *a, void *stream) {
  NH_REQUIRE(a, "nh_assemble_matrix: NULL args");
  NH_REQUIRE(a->ndims >= 1 && a->ndims <= 3, "ndims must be 1..3");
  NH_REQUIRE(a->nq >= 1 && a->weights_dev, "quadrature missing");
  constexpr int known_flags = NH_MATRIX_EXCLUSIVE | NH_MATRIX_EMAP_BY_ELEMENT | NH_MATRIX_NO_MFMA | NH_MATRIX_FIRST_TOUCH | NH_MATRIX_GATHER | NH_MATRIX_STORE | NH_MATRIX_FUSED;
  NH_REQUIRE((a->flags & ~known_flags) == 0, "nh_assemble_matrix: unknown flag bits 0x%x", a->flags & ~known_flags);
  NH_REQUIRE(a->C_host && a->srowptr_dev && a->emap_dev && a->values_dev, "nh_assemble_matrix: NULL coefficient / pattern / values");
  NH_REQUIRE(a->test.T_dev && a->test.dofs_dev && a->trial.T_dev && a->trial.dofs_dev, "basis tables missing");
  int rc = check_geom(a->geom);
  if (rc) return rc;
  if (a->nelems == 0) return NH_OK;
  FormK form;
  rc = make_form(a->ndims, a->nct, a->ncr, a->C_host, nullptr, a->mask_host, &form);
  if (rc) return rc;
  const int S = 1 + a->ndims, JW = a->ndims * a->ndims + 1;
  MatK p;
  p.nelems = a->nelems;
  p.elist = a->elist_dev;
  p.nq = a->nq;
  p.weights = a->weights_dev;
  p.geom = to_k(a->geom);
  p.geom.nograd = !a->cq_dev && !uses_gradients(a->C_host, a->nct, 1 + a->ndims, a->ncr);
  p.test = to_k(a->test);
  p.trial = to_k(a->trial);
  p.same = (a->test.T_dev == a->trial.T_dev && a->test.off_dev == a->trial.off_dev && a->test.tab_dev == a->trial.tab_dev &&
            a->test.nb == a->trial.nb);
  p.sym = 0;
  p.trirank = nullptr, p.tribase = nullptr;
  if (p.same && a->test.dofs_dev == a->trial.dofs_dev && a->nct == a->ncr && !a->cq_dev && !(a->flags & (NH_MATRIX_EXCLUSIVE | NH_MATRIX_FIRST_TOUCH)) && !getenv("NUTILS_AMD_NO_SYM_GRAM")) {
    const int S2 = 1 + a->ndims, nc2 = a->nct;
    p.sym = 1;
    for (int c = 0; c < nc2 && p.sym; ++c)
      for (int sa = 0; sa < S2 && p.sym; ++sa)
        for (int d = 0; d < nc2 && p.sym; ++d)
          for (int sb = 0; sb < S2; ++sb)
            if (a->C_host[((c * S2 + sa) * nc2 + d) * S2 + sb] != a->C_host[((d * S2 + sb) * nc2 + c) * S2 + sa] ||
                (a->mask_host && a->mask_host[c * nc2 + d] != a->mask_host[d * nc2 + c])) {
              p.sym = 0;
              break;
            }
  }
  p.srowptr = (const i64 *)a->srowptr_dev;
  p.emap = a->emap_dev;
  p.eoff = (const i64 *)a->eoff_dev;
  p.values = a->values_dev;
  p.scale = a->scale_dev;
  p.emap_by_elem = (a->flags & 2) != 0;
  p.exclusive = (a->flags & 1) != 0;
  p.ft.on = (a->flags & 32) != 0;
  p.ft.p1 = a->nodes_per_axis;
  for (int d = 0; d < 3; ++d) p.ft.shape[d] = d < a->ndims ? a->grid_shape[d] : 1;
  if (p.ft.on) {
    NH_REQUIRE((a->flags & 3) == 3 && a->elist_dev, "NH_MATRIX_FIRST_TOUCH needs NH_MATRIX_EXCLUSIVE | NH_MATRIX_EMAP_BY_ELEMENT and an element list");
    NH_REQUIRE(!a->cq_dev && !a->test.off_dev, "NH_MATRIX_FIRST_TOUCH: constant forms on uniform bases only");
    i64 ncell = 1, nloc = 1;
    for (int d = 0; d < a->ndims; ++d) {
      NH_REQUIRE(a->grid_shape[d] >= 1, "NH_MATRIX_FIRST_TOUCH: grid_shape");
      ncell *= a->grid_shape[d];
      nloc *= a->nodes_per_axis;
    }
    NH_REQUIRE(a->nodes_per_axis >= 2 && nloc == a->test.nb && nloc == a->trial.nb, "NH_MATRIX_FIRST_TOUCH: nodes_per_axis^ndims must equal the local basis size");
    (void)ncell;
  }
  p.cq = a->cq_dev;
  NH_REQUIRE(!a->elist_dev || (a->test.nb && a->trial.nb), "nh_assemble_matrix: elist with ragged bases is not supported");
  const bool bucketed = (a->test.off_dev || a->trial.off_dev) && a->pattern && a->pattern->nbuckets && !a->elist_dev && a->pattern->nelems == a->nelems;
  if (bucketed) {  // (the size classes of the pattern carry their own maxima: no scan of the offsets per call)
    p.maxnbt = p.maxnbr = 0;
    for (int b = 0; b < a->pattern->nbuckets; ++b) {
      p.maxnbt = std::max(p.maxnbt, a->pattern->bucket_nbt[b]);
      p.maxnbr = std::max(p.maxnbr, a->pattern->bucket_nbr[b]);
    }
  } else {
    if ((rc = max_nb(a->test, a->nelems, &p.maxnbt, nh_stream(stream))) != NH_OK) return rc;
    if ((rc = max_nb(a->trial, a->nelems, &p.maxnbr, nh_stream(stream))) != NH_OK) return rc;
  }
  // NH_MATRIX_GATHER: local matrices to scratch, then one deterministic sum per CSR entry
  p.local = nullptr;
  const bool gather = (a->flags & NH_MATRIX_GATHER) != 0;
  i64 local_ld = 0;  // > 0: the scratch is entry-major, local[(m * nbr + n) * ld + list position] (thread-per-element kernel)
  const bool fused = (a->flags & NH_MATRIX_FUSED) != 0;
  NH_REQUIRE(gather || fused || !(a->flags & NH_MATRIX_STORE), "NH_MATRIX_STORE is an option of NH_MATRIX_GATHER / NH_MATRIX_FUSED");
  if (fused) {  // owner blocks: one pass, no scratch, no global atomics
    NH_REQUIRE(!gather && !(a->flags & (NH_MATRIX_EXCLUSIVE | NH_MATRIX_FIRST_TOUCH)), "NH_MATRIX_FUSED excludes NH_MATRIX_GATHER / EXCLUSIVE / FIRST_TOUCH");
    bool done = false;
    if ((rc = nh_fused_scalar(a, &done, nh_stream(stream))) != NH_OK) return rc;
    if (done) return NH_OK;
    if ((rc = nh_owner_vector(a, slots_of(form), &done, nh_stream(stream))) != NH_OK) return rc;  // vector-valued blocks: row tasks (nh_owner.hip)
    if (done) return NH_OK;
    if (a->flags & NH_MATRIX_STORE) {  // not applicable to this launch: the default path adds, so the block starts from zero
      NH_REQUIRE(a->pattern, "NH_MATRIX_FUSED | NH_MATRIX_STORE needs the pattern handle");
      NH_CHECK_HIP(hipMemsetAsync(a->values_dev, 0, sizeof(double) * (size_t)a->pattern->nnz * form.tot, nh_stream(stream)));
    }
  }
  if (gather) {
    NH_REQUIRE(a->pattern && a->pattern->nelems == a->nelems && !(a->flags & (NH_MATRIX_EXCLUSIVE | NH_MATRIX_FIRST_TOUCH)) && !(a->elist_dev && (a->flags & NH_MATRIX_EMAP_BY_ELEMENT)),
               "NH_MATRIX_GATHER needs the pattern handle and all of its elements in one call");
    nh_pattern *pat = const_cast<nh_pattern *>(a->pattern);
    if ((rc = nh_gather_prepare(pat, a->test, a->elist_dev, nh_stream(stream))) != NH_OK) return rc;
    double *scratch = nullptr;
    // (the thread-per-element kernel writes blocks of 64 elements: room for the last, partly filled block)
    const size_t padded = (size_t)((pat->nelems + 63) / 64 * 64) * std::max(pat->nbt * pat->nbr, 1);
    if ((rc = nh_gather_scratch(std::max((size_t)pat->emap_len, padded) * a->nct * a->ncr, &scratch)) != NH_OK) return rc;
    bool done = false;
    if ((rc = nh_local_scalar(a, scratch, &done, nh_stream(stream))) != NH_OK) return rc;
    if (!done && (rc = nh_local_vector(a, scratch, &done, nh_stream(stream))) != NH_OK) return rc;
    if (done) {
      return nh_gather_values(pat, scratch, a->nelems, slots_of(form), a->values_dev, (a->flags & NH_MATRIX_STORE) != 0, nh_stream(stream));
    }
    p.local = scratch;
    // NUTILS_AMD_SYM_SCRATCH=1 (measured, NOT the default): symmetric Gram blocks of 2 or 3 components send the node pairs m >= n only to the scratch and the gather mirrors
    // them -- half the scratch bytes, but the mirrored sources of a CSR row lie nb blocks apart: the gather of the ragged rational workload 0.39 -> ~1.0 ms (1.27 -> 1.88 ms in all)
    if (p.sym && a->nct * a->ncr > 1 && (a->nct == 2 || a->nct == 3) && getenv("NUTILS_AMD_SYM_SCRATCH") && atoi(getenv("NUTILS_AMD_SYM_SCRATCH"))) {
      if ((rc = nh_gather_prepare_sym(pat, a->test, a->elist_dev, nh_stream(stream))) != NH_OK) return rc;
      p.sym = 2;
    }
    // Triangular scratch (default where it applies: k_gram_sym takes EVERY launch of this call, the sums are stored, the pattern is symmetric): node pairs of the
    // dof-sorted nodes, m' >= n', packed -- half the scratch written and read; the upper triangle of the matrix is mirrored from the lower (nh_gather.hip)
    const bool notri = getenv("NUTILS_AMD_NO_TRI_SCRATCH") && atoi(getenv("NUTILS_AMD_NO_TRI_SCRATCH"));
    bool allmask = true;
    for (int c = 0; c < a->nct; ++c)
      for (int d = 0; d < a->ncr; ++d) allmask = allmask && form.mask[c][d];
    if (p.sym == 1 && !notri && allmask && (a->flags & NH_MATRIX_STORE) && !a->elist_dev && !pat->tri_failed) {
      bool all = true;
      const bool rag = (a->test.off_dev || a->trial.off_dev) && pat->nbuckets && pat->nelems == a->nelems;
      for (int b = 0; b < (rag ? pat->nbuckets : 1) && all; ++b) {
        MatK q = p;
        if (rag) {
          if (!pat->bucket_n[b]) continue;
          q.maxnbt = pat->bucket_nbt[b], q.maxnbr = pat->bucket_nbr[b];
        }
        all = nh_gram_sym_applies(q, form, a->ndims);
      }
      if (all) {
        if ((rc = nh_gather_prepare_tri(pat, a->test, nh_stream(stream))) != NH_OK) return rc;
        if (pat->gsrc_tri) p.sym = 3, p.trirank = pat->tri_rank, p.tribase = pat->tri_base;
      }
    }
  }
  // MFMA path: uniform shared tables, >= 16 local rows and columns, tile counts we instantiate
  const int Nloc = a->trial.nb * a->ncr;
  if (!gather && !(a->flags & 4) && !a->cq_dev && p.same && a->test.nb >= 16 && Nloc >= 16 && !a->test.off_dev && a->ndims >= 2 && a->geom.kind != 0) {
    MfmaX x;
    x.nas = 0;
    for (int sa = 0; sa < S; ++sa) {
      bool any = false;
      for (int c = 0; c < a->nct && !any; ++c)
        for (int d = 0; d < a->ncr && !any; ++d)
          for (int b = 0; b < S && !any; ++b) any = a->C_host[((c * S + sa) * a->ncr + d) * S + b] != 0.;
      if (any) x.aslot[x.nas++] = sa;
    }
    x.mt = (a->test.nb + 15) / 16;
    x.nt = (Nloc + 15) / 16;
    x.kt = ((a->nq + 3) / 4) * x.nas;
    x.flags = a->flags;
#ifdef NH_ABLATION  // ablation hooks (skip the scatter: 8, one k-step: 16) exist only in -DNH_ABLATION builds
    if (getenv("NH_MFMA_DEBUG")) x.flags |= atoi(getenv("NH_MFMA_DEBUG")) & (8 | 16);
#endif
    const size_t ldsm = sizeof(double) * ((size_t)form.formd + (((size_t)a->nq * JW + 3) & ~(size_t)3) + (size_t)a->nq * a->test.nb * S +
                                          (size_t)x.nas * ((a->nq + 3) & ~3) * x.mt * 16);
    if (x.nas > 0 && ldsm <= 160 * 1024 && x.mt <= 4) {
      hipStream_t sm = nh_stream(stream);
      int dev = 0, cus = 256;
      NH_CHECK_HIP(hipGetDevice(&dev));
      NH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(6, (160 * 1024) / ldsm));
      dim3 gridm((unsigned)std::min<i64>(a->nelems, (i64)cus * wg_per_cu)), blockm(256);
#define LAUNCHM(ND, MT)                                                                                                          \
  do {                                                                                                                           \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_matrix_mfma<ND, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm)); \
    hipLaunchKernelGGL((k_matrix_mfma<ND, MT>), gridm, blockm, ldsm, sm, p, form, x);                                            \
  } while (0)
      const int key = a->ndims * 10 + x.mt;
      bool launched = true;
      switch (key) {
        case 21: LAUNCHM(2, 1); break;
        case 22: LAUNCHM(2, 2); break;
        case 31: LAUNCHM(3, 1); break;
        case 32: LAUNCHM(3, 2); break;   // 3-D p2: 27 local scalar dofs
        case 34: LAUNCHM(3, 4); break;   // 3-D p3: 64
        default: launched = false;
      }
#undef LAUNCHM
      if (launched) {
        NH_LAUNCH_CHECK();
        return NH_OK;
      }
    }
  }
  hipStream_t s = nh_stream(stream);
  auto launch_generic = [&](MatK q) -> int {  // q.nelems / elist / maxnb* select the elements, LDS sized for them
    if (q.local) {  // symmetric two-component blocks on the way to the gather scratch: the instruction-lean kernel of nh_gram_sym.inc
      bool taken = false;
      const int rg = nh_gram_sym_launch(q, form, a->ndims, &taken, s);
      if (rg != NH_OK || taken) return rg;
    }
    const int per_q0 = (q.same ? q.maxnbt : q.maxnbt + q.maxnbr) * S * (int)sizeof(double);
    const int per_qw = per_q0 + q.maxnbr * a->nct * a->ncr * S * (int)sizeof(double);
    const size_t fixed = sizeof(double) * ((size_t)form.formd + (size_t)a->nq * JW);
    // the W table trades S*S for S multiply-adds per entry and point; these one-wave workgroups are latency bound, so it is only
    // used while the workgroup stays small enough for >= 16 of them per CU (measured: 3-D P1 4.8 -> 3.8 ms, 2-D p2 0.96 -> 1.1 ms)
    q.use_w = fixed + (size_t)a->nq * per_qw <= 10 * 1024 && !(!a->cq_dev && a->nct * a->ncr > 1);  // (vector-valued constant forms: the Gram path needs no W table)
    const int per_q = q.use_w ? per_qw : per_q0;
    q.qchunk = std::max(1, std::min(a->nq, LDS_BUDGET / std::max(per_q, 1)));
    const size_t lds = fixed + (size_t)q.qchunk * per_q;
    NH_REQUIRE(lds <= 160 * 1024, "element too large for LDS (%zu bytes)", lds);
    // two waves per element when its (node pairs of the Gram path | entries) keep both busy: the one-wave workgroups of the ragged rational workload ran at 1.9 waves
    // per SIMD (LDS bound) and waited 61 % of their cycles
    const i64 work = (!a->cq_dev && a->nct * a->ncr > 1) ? (q.sym ? (i64)q.maxnbt * (q.maxnbt + 1) / 4 : (i64)q.maxnbt * ((q.maxnbr + 1) / 2)) : (i64)q.maxnbt * a->nct * q.maxnbr * a->ncr;
    int nw = work > 96 ? 2 : 1;
    if (getenv("NUTILS_AMD_GENERIC_WAVES")) nw = std::max(1, std::min(4, atoi(getenv("NUTILS_AMD_GENERIC_WAVES"))));
    dim3 grid(grid_for(q.nelems)), block(64 * nw);
#define LAUNCH(ND)                                                                                                          \
  do {                                                                                                                      \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_matrix_generic<ND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k_matrix_generic<ND>, grid, block, lds, s, q, form);                                                 \
  } while (0)
    if (a->ndims == 1) LAUNCH(1);
    if (a->ndims == 2) LAUNCH(2);
    if (a->ndims == 3) LAUNCH(3);
#undef LAUNCH
    NH_LAUNCH_CHECK();
    return NH_OK;
  };
  const bool ragged = a->test.off_dev || a->trial.off_dev;
  if (ragged && a->pattern && a->pattern->nbuckets && !a->elist_dev && a->pattern->nelems == a->nelems) {
    // one launch per size class (LDS-bucketed): element lists of the pattern, everything indexed by element id
    for (int b = 0; b < a->pattern->nbuckets; ++b) {
      if (!a->pattern->bucket_n[b]) continue;
      MatK q = p;
      q.nelems = a->pattern->bucket_n[b];
      q.elist = a->pattern->bucket_elist[b];
      q.emap_by_elem = 1;
      q.maxnbt = a->pattern->bucket_nbt[b];
      q.maxnbr = a->pattern->bucket_nbr[b];
      if ((rc = launch_generic(q)) != NH_OK) return rc;
    }
  } else if ((rc = launch_generic(p)) != NH_OK)
    return rc;
  if (gather) {
    return nh_gather_values(a->pattern, p.local, local_ld, slots_of(form), a->values_dev, (a->flags & NH_MATRIX_STORE) != 0, nh_stream(stream), p.sym == 3 ? 2 : p.sym == 2);
  }
  return NH_OK;
}

int nh_assemble_vector(const nh_vector_args *a, void *stream) {
  NH_REQUIRE(a, "nh_assemble_vector: NULL args");
  NH_REQUIRE(a->ndims >= 1 && a->ndims <= 3, "ndims must be 1..3");
  NH_REQUIRE(a->nq >= 1 && a->weights_dev, "quadrature missing");
  NH_REQUIRE(a->out_dev || a->out_scalar_dev || a->local_dev, "nh_assemble_vector: no output");
  NH_REQUIRE((a->test.nb == 0 && !a->test.off_dev) || (a->test.T_dev && a->test.dofs_dev), "test basis tables missing");
  NH_REQUIRE((a->trial.nb == 0 && !a->trial.off_dev) || (a->trial.T_dev && a->trial.dofs_dev), "trial basis tables missing");
  NH_REQUIRE(!(a->C_host && !a->u_dev), "nh_assemble_vector: coefficient tensor given without field u");
  NH_REQUIRE(!a->elist_dev || (!a->test.off_dev && !a->trial.off_dev), "elist with ragged bases is not supported");
  int rc = check_geom(a->geom);
  if (rc) return rc;
  if (a->nelems == 0) return NH_OK;
  FormK form;
  rc = make_form(a->ndims, a->nct, a->ncr, a->C_host, a->f_host, nullptr, &form);
  if (rc) return rc;
  const int S = 1 + a->ndims, JW = a->ndims * a->ndims + 1;
  VecK p;
  p.nelems = a->nelems;
  p.elist = a->elist_dev;
  p.nq = a->nq;
  p.weights = a->weights_dev;
  p.geom = to_k(a->geom);
  p.geom.nograd = !uses_gradients(a->C_host, a->nct, S, a->ncr) && !uses_gradients(a->f_host, a->nct, S, 0);
  p.test = to_k(a->test);
  p.trial = to_k(a->trial);
  p.same = (a->test.T_dev == a->trial.T_dev && a->test.off_dev == a->trial.off_dev && a->test.tab_dev == a->trial.tab_dev &&
            a->test.nb == a->trial.nb);
  p.u = a->u_dev;
  p.out = a->local_dev ? nullptr : a->out_dev;
  p.local = a->local_dev;
  p.f0 = a->f0;
  p.out_scalar = a->out_scalar_dev;
  p.scale = a->scale_dev;
  if ((rc = max_nb(a->test, a->nelems, &p.maxnbt, nh_stream(stream))) != NH_OK) return rc;
  if ((rc = max_nb(a->trial, a->nelems, &p.maxnbr, nh_stream(stream))) != NH_OK) return rc;
  p.cs = std::max(a->nct, a->ncr);
  const int per_q = ((p.same ? p.maxnbt : p.maxnbt + p.maxnbr) * S + 2 * p.cs * S) * (int)sizeof(double);
  p.qchunk = std::max(1, std::min(a->nq, LDS_BUDGET / per_q));
  const size_t lds = sizeof(double) * ((size_t)form.formd + (size_t)a->nq * JW) + (size_t)p.qchunk * per_q;
  NH_REQUIRE(lds <= 160 * 1024, "element too large for LDS (%zu bytes)", lds);
  hipStream_t s = nh_stream(stream);
  dim3 grid(grid_for(a->nelems)), block(64);
#define LAUNCH(ND)                                                                                                          \
  do {                                                                                                                      \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_vector_generic<ND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k_vector_generic<ND>, grid, block, lds, s, p, form);                                                 \
  } while (0)
  if (a->ndims == 1) LAUNCH(1);
  if (a->ndims == 2) LAUNCH(2);
  if (a->ndims == 3) LAUNCH(3);
#undef LAUNCH
  NH_LAUNCH_CHECK();
  return NH_OK;
}

int nh_sample_eval(const nh_eval_args *a, void *stream) {
  NH_REQUIRE(a, "nh_sample_eval: NULL args");
  NH_REQUIRE(a->ndims >= 1 && a->ndims <= 3, "ndims must be 1..3");
  int rc = check_geom(a->geom);
  if (rc) return rc;
  NH_REQUIRE(a->geom.kind != NH_GEOM_BOX || !a->x_dev || a->points_dev, "box geometry coordinates need points_dev");
  NH_REQUIRE(!a->U_dev || (a->u_dev && a->trial.T_dev && a->trial.dofs_dev && a->ncr >= 1), "field evaluation needs u, tables and dofs");
  const i64 n = a->nelems * a->nq;
  if (!n) return NH_OK;
  EvalK p;
  p.nelems = a->nelems;
  p.elist = a->elist_dev;
  p.nq = a->nq;
  p.geom = to_k(a->geom);
  p.trial = to_k(a->trial);
  p.ncr = a->ncr;
  p.points = a->points_dev;
  p.u = a->u_dev;
  p.x = a->x_dev;
  p.detj = a->detj_dev;
  p.U = a->U_dev;
  dim3 grid((unsigned)((n + 127) / 128)), block(128);
  hipStream_t s = nh_stream(stream);
  if (a->ndims == 1) hipLaunchKernelGGL(k_sample_eval<1>, grid, block, 0, s, p);
  if (a->ndims == 2) hipLaunchKernelGGL(k_sample_eval<2>, grid, block, 0, s, p);
  if (a->ndims == 3) hipLaunchKernelGGL(k_sample_eval<3>, grid, block, 0, s, p);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

}  // extern "C"
