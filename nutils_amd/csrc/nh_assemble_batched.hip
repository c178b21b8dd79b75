// Batched element kernels: several elements per workgroup, for bases with few functions per element (splines, low-order
// Lagrange).  The one-wave-per-element kernels of nh_assemble_generic.hip spend a wavefront, five barriers and a physical-gradient
// table in LDS on an element with 9 functions and 25 points and are latency bound (2.7e8 elements/s); here the lanes of a workgroup run
// over (element, point) pairs of a batch of elements for everything that is pointwise -- geometry, field values, the integrand
// coefficients of ALL terms -- and then over (element, test function) pairs for the one contraction with the test tables.
//
// nh_assemble_terms: fused linear forms (include/nutils_hip.h).  Replaces the generated residual loop of the reference
// (evaluable.py:6773-6786, Inflate/Assemble scatter :3341-3495) for a list of terms, instead of one launch per term.
#include "nh_common.h"
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

#include "nh_geom.inc"

constexpr int MAXF = 6;    // fields
constexpr int MAXFC = 8;   // sum of field components
constexpr int MAXB = 2;    // output blocks
constexpr int MAXCT = 4;   // sum of block components
constexpr int MAXT = 32;   // terms
constexpr int MAXP = 4;    // pointwise polynomials
constexpr int NTB = 256;   // threads per workgroup
constexpr int NBLK = 3;     // trial functions per lane in the contraction of the matrix kernel
constexpr int MAXL = 8;    // term lists of one launch (nh_assemble_terms_multi)
constexpr int TABARG = 256;  // doubles of the term table that fit the kernel arguments
constexpr int TABPAD = 24;   // readable doubles behind them (records read whole, at the length of the longest kind)

__device__ __forceinline__ i64 boff(const BasisK &b, i64 e) { return b.off ? nh_g(b.off)[e] : e * (i64)b.nb; }
__device__ __forceinline__ int bnb(const BasisK &b, i64 e) { return b.off ? (int)(nh_g(b.off)[e + 1] - nh_g(b.off)[e]) : b.nb; }
__device__ __forceinline__ i64 bfn(const BasisK &b, i64 e) { return b.off ? nh_g(b.off)[e] : (b.tab ? (i64)nh_g(b.tab)[e] * b.nb : 0); }

struct FieldK {
  BasisK b;
  const double *u;
  int ncomp, c0;  // c0: first slot of this field in the per-point value table
  int maxnb, ue0; // coefficients of an element staged in LDS: [maxnb][ncomp] at ue0 of the element's record
  int tsame;      // matrix kernel: the tables are the test tables (staged in LDS)
};
struct BlockK {
  BasisK test;
  double *out;
  double *local;       // element-major local vectors instead of atomics into out (deterministic scatter, nh_scatter.hip)
  int nct, c0, maxnb;  // c0: first slot in the per-point integrand table
};
// term table (doubles, staged in LDS): per term [block, field, poly, hasC, hasf, f[nct][S], C[nct][S][ncr][S]], per polynomial
// [nvars, nterms, slot[4], (coeff, power[4]) x nterms]
struct TermsK {
  i64 nelems;
  const int32_t *elist;
  int nq, eb;  // eb: elements per batch
  const double *weights;
  GeomK geom;
  int nfields, nblocks, nterms, npolys;
  int fct, ct;  // value slots per point, integrand slots per point
  FieldK fields[MAXF];
  BlockK blocks[MAXB];
  const double *scale[MAXT];
  int toff[MAXT], poff[MAXP], qoff[MAXT];  // qoff: 0 or the record [slot_t, slot_r, B[S][S]] of the point factor U_t . B . U_r of the term
  const double *table;  // NULL: the table is tabarg (small tables travel with the kernel arguments: no copy, no buffer to keep alive)
  int tlen;
  double tabarg[TABARG];
  int rowsper;  // sum over blocks of maxnb * nct: output lanes per element
  int uesz;     // staged field coefficients per element
  int tstage;   // all blocks and fields live on ONE uniform-size basis: the table of the batch's first element class is staged in LDS ([maxnb][nq][S] behind UE)
};

// coefficients of all fields on the elements of a batch -> LDS, ue[el][uesz]: the gathers u[dofs[...]] are two dependent global loads; here
// every lane has its own in flight at once, instead of one pair per basis function in the loop over the functions of a point
__device__ __forceinline__ void stage_coeffs(const FieldK *fields, int nfields, int uesz, double *ue, int eb, i64 b0, i64 nelems, const int32_t *elist,
                                              int tid) {
  for (int i = tid; i < eb * uesz; i += NTB) {
    const int el = i / uesz;
    int r = i - el * uesz;
    if (b0 + el >= nelems) continue;
    const i64 e = elist ? nh_g(elist)[b0 + el] : b0 + el;
    int f = 0;
    while (f + 1 < nfields && r >= fields[f + 1].ue0) ++f;
    const FieldK &F = fields[f];
    r -= F.ue0;
    const int n = r / F.ncomp, d = r - n * F.ncomp;
    if (n < bnb(F.b, e)) ue[i] = nh_g(F.u)[(i64)nh_g(F.b.dofs)[boff(F.b, e) + n] * F.ncomp + d];
  }
}

// field values and physical gradients of all fields at point q of element e -> u[fct][S]; ue: staged coefficients of the element,
// TT: staged test tables of the element (fields with tsame) or NULL
template <int ND>
__device__ __forceinline__ void eval_fields(const FieldK *fields, int nfields, i64 e, int q, int nq, const double (&Ji)[ND][ND], const double *ue,
                                             const double *TT, double *u) {
  constexpr int S = 1 + ND;
  // several scalar fields on ONE basis (the phase field and the chemical potential of configs[3], ...): a table row is loaded once for all of them
  bool shared = nfields >= 2 && nfields <= 4;
  for (int f = 0; f < nfields && shared; ++f)
    shared = fields[f].ncomp == 1 && fields[f].b.T == fields[0].b.T && fields[f].b.tab == fields[0].b.tab && fields[f].b.off == fields[0].b.off && fields[f].b.nb == fields[0].b.nb &&
             fields[f].tsame == fields[0].tsame;
  if (shared) {
    const FieldK &F0 = fields[0];
    const int nb = bnb(F0.b, e);
    // (the staged copy in LDS or the table in global memory: two loops, each over a pointer of ONE address space -- a pointer selected at run time makes every read a
    // flat load)
    double r[4][S];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int s = 0; s < S; ++s) r[f][s] = 0;
    auto rows = [&](auto T) __attribute__((always_inline)) {
#pragma unroll 4
      for (int n = 0; n < nb; ++n) {
        const auto Tn = T + (size_t)n * nq * S;
        double tn[S];
#pragma unroll
        for (int s = 0; s < S; ++s) tn[s] = Tn[s];
#pragma unroll
        for (int f = 0; f < 4; ++f)
          if (f < nfields) {
            const double un = ue[fields[f].ue0 + n];
#pragma unroll
            for (int s = 0; s < S; ++s) r[f][s] += tn[s] * un;
          }
      }
    };
    if (TT && F0.tsame) rows(TT + (size_t)q * S);
    else {
      rows(nh_g(F0.b.T) + (bfn(F0.b, e) * nq + q) * S);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
      if (f < nfields) {
        double *o = u + fields[f].c0 * S;
        o[0] = r[f][0];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          double sum = 0;
#pragma unroll
          for (int j = 0; j < ND; ++j) sum += r[f][1 + j] * Ji[j][i];
          o[1 + i] = sum;
        }
      }
    return;
  }
  for (int f = 0; f < nfields; ++f) {
    const FieldK &F = fields[f];
    const int nb = bnb(F.b, e);
    const double *c = ue + F.ue0;
    for (int d = 0; d < F.ncomp; ++d) {
      double r[S];
#pragma unroll
      for (int s = 0; s < S; ++s) r[s] = 0;
      auto rows = [&](auto T) __attribute__((always_inline)) {
#pragma unroll 4
        for (int n = 0; n < nb; ++n) {
          const double un = c[n * F.ncomp + d];
          const auto Tn = T + (size_t)n * nq * S;
#pragma unroll
          for (int s = 0; s < S; ++s) r[s] += Tn[s] * un;
        }
      };
      if (TT && F.tsame) rows(TT + (size_t)q * S);
      else {
        rows(nh_g(F.b.T) + (bfn(F.b, e) * nq + q) * S);
      }
      double *o = u + (F.c0 + d) * S;
      o[0] = r[0];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        double s = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j) s += r[1 + j] * Ji[j][i];
        o[1 + i] = s;
      }
    }
  }
}

// pointwise polynomials of field values: table entry [nvars, nterms, slot[4], (coeff, power[4]) x nterms]
template <int S>
__device__ __forceinline__ void eval_polys(const double *tab, const int *poff, int npolys, const double *u, double (&pv)[MAXP]) {
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    pv[k] = 1.;
    if (k < npolys) {
      const double *P = tab + poff[k];
      const int nv = (int)P[0], nt = (int)P[1];
      double x[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) x[v] = v < nv ? u[(int)P[2 + v] * S] : 1.;
      double s = 0;
      for (int t2 = 0; t2 < nt; ++t2) {
        const double *M = P + 6 + 5 * t2;
        double m = M[0];
#pragma unroll
        for (int v = 0; v < 4; ++v)
          for (int k2 = (int)M[1 + v]; k2 > 0; --k2) m *= x[v];
        s += m;
      }
      pv[k] = s;
    }
  }
}
// point factor s = sum_ab B[a][b] U_t[a] U_r[b] of a term (scalar fields; the energy Hessians of quasi-linear problems), record [slot_t, slot_r, B]
template <int S>
__device__ __forceinline__ double point_factor(const double *Q, const double *u) {
  const double *ua = u + (int)Q[0] * S, *ub = u + (int)Q[1] * S;
  double s = 0;
#pragma unroll
  for (int a = 0; a < S; ++a)
#pragma unroll
    for (int b = 0; b < S; ++b) s += Q[2 + a * S + b] * ua[a] * ub[b];
  return s;
}

__device__ __forceinline__ double pick(const double (&pv)[MAXP], int k) { return k == 0 ? pv[0] : k == 1 ? pv[1] : k == 2 ? pv[2] : pv[3]; }

template <int ND>
__device__ __forceinline__ void terms_body(const TermsK &p, const unsigned bid, const unsigned nbid) {
  constexpr int S = 1 + ND;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *tab = lds;                                // term table
  double *G = lds + p.tlen;                         // [eb * nq][ct][S]: integrand, then its reference form times w |J|
  double *U = G + (size_t)p.eb * p.nq * p.ct * S;   // [NTB][fct][S]: field values of this thread's point
  double *UE = U + (size_t)NTB * p.fct * S;         // [eb][uesz]: field coefficients of the elements of the batch
  double *TS = UE + (size_t)p.eb * p.uesz;          // tstage: [maxnb][nq][S] table of one element class, then its tag
  i64 *tag = reinterpret_cast<i64 *>(TS + (size_t)p.blocks[0].maxnb * p.nq * S);
  const int tid = threadIdx.x;
  for (int i = tid; i < p.tlen; i += NTB) tab[i] = p.table ? p.table[i] : p.tabarg[i];
  if (p.tstage && tid == 0) *tag = -1;
  const int npts = p.eb * p.nq;
  for (i64 b0 = (i64)bid * p.eb; b0 < p.nelems; b0 += (i64)nbid * p.eb) {
    __syncthreads();  // table staged; G of the previous batch consumed
    i64 cls0 = -1;
    if (p.tstage) {  // the class of the batch's first element (it stays staged while consecutive batches keep it)
      cls0 = bfn(p.blocks[0].test, p.elist ? nh_g(p.elist)[b0] : b0);
      if (*tag != cls0) {
        const int n = p.blocks[0].maxnb * p.nq * S;
        for (int i = tid; i < n; i += NTB) TS[i] = nh_g(p.blocks[0].test.T)[cls0 * p.nq * S + i];
      }
    }
    stage_coeffs(p.fields, p.nfields, p.uesz, UE, p.eb, b0, p.nelems, p.elist, tid);
    __syncthreads();
    if (p.tstage && tid == 0) *tag = cls0;
    for (int t = tid; t < npts; t += NTB) {
      const int el = t / p.nq, q = t - el * p.nq;
      const i64 ie = b0 + el;
      double *g = G + (size_t)t * p.ct * S;
      if (ie >= p.nelems) continue;
      const i64 e = p.elist ? nh_g(p.elist)[ie] : ie;
      double Ji[ND][ND], det;
      geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
      const double wdet = nh_g(p.weights)[q] * fabs(det);
      double *u = U + (size_t)tid * p.fct * S;
      // (wave-uniform choice: a pointer that is LDS for some lanes and global for others would be a flat access)
      if (p.tstage && __all(bfn(p.blocks[0].test, e) == cls0)) eval_fields<ND>(p.fields, p.nfields, e, q, p.nq, Ji, UE + (size_t)el * p.uesz, TS, u);
      else eval_fields<ND>(p.fields, p.nfields, e, q, p.nq, Ji, UE + (size_t)el * p.uesz, nullptr, u);
      double pv[MAXP];
      eval_polys<S>(tab, p.poff, p.npolys, u, pv);
      // integrand of every block: sum of the terms
      for (int i = 0; i < p.ct * S; ++i) g[i] = 0.;
      for (int t2 = 0; t2 < p.nterms; ++t2) {
        const double *H = tab + p.toff[t2];
        const int blk = (int)H[0], fld = (int)H[1], pol = (int)H[2], hasC = (int)H[3], hasf = (int)H[4];
        const BlockK &B = p.blocks[blk];
        double coef = p.scale[t2] ? nh_g(p.scale[t2])[ie * p.nq + q] : 1.;
        if (pol >= 0) coef *= pick(pv, pol);
        if (p.qoff[t2]) coef *= point_factor<S>(tab + p.qoff[t2], u);
        const double *fv = H + 5, *C = fv + B.nct * S;
        const int ncr = fld >= 0 ? p.fields[fld].ncomp : 0;
        const double *uf = fld >= 0 ? u + p.fields[fld].c0 * S : u;
        for (int c = 0; c < B.nct; ++c)
#pragma unroll
          for (int a = 0; a < S; ++a) {
            double s = hasf ? fv[c * S + a] : 0.;
            if (hasC)
              for (int d = 0; d < ncr; ++d)
#pragma unroll
                for (int b = 0; b < S; ++b) s += C[((c * S + a) * ncr + d) * S + b] * uf[d * S + b];
            g[(B.c0 + c) * S + a] += coef * s;
          }
      }
      // reference form: r[m] = sum_q sum_s T[m][q][s] G[q][s], G[0] = w|J| F[0], G[1+j] = w|J| sum_i Jinv[j][i] F[1+i]
      for (int c = 0; c < p.ct; ++c) {
        double F[S];
#pragma unroll
        for (int a = 0; a < S; ++a) F[a] = g[c * S + a];
        g[c * S] = wdet * F[0];
#pragma unroll
        for (int j = 0; j < ND; ++j) {
          double s = 0;
#pragma unroll
          for (int i = 0; i < ND; ++i) s += Ji[j][i] * F[1 + i];
          g[c * S + 1 + j] = wdet * s;
        }
      }
    }
    __syncthreads();
    // lanes over (element, block, test function, component)
    for (int k = tid; k < p.eb * p.rowsper; k += NTB) {
      const int el = k / p.rowsper;
      int r = k - el * p.rowsper;
      const i64 ie = b0 + el;
      if (ie >= p.nelems) continue;
      const i64 e = p.elist ? nh_g(p.elist)[ie] : ie;
      int blk = 0;
      if (p.nblocks > 1 && r >= p.blocks[0].maxnb * p.blocks[0].nct) r -= p.blocks[0].maxnb * p.blocks[0].nct, blk = 1;
      const BlockK &B = p.blocks[blk];
      const int m = r / B.nct, c = r - m * B.nct;
      if (m >= bnb(B.test, e)) continue;
      const double *g = G + ((size_t)el * p.nq * p.ct + (B.c0 + c)) * S;
      double acc = 0;
      auto row = [&](auto T) __attribute__((always_inline)) {
        for (int q = 0; q < p.nq; ++q) {
#pragma unroll
          for (int s = 0; s < S; ++s) acc += T[q * S + s] * g[(size_t)q * p.ct * S + s];
        }
      };
      if (p.tstage && bfn(B.test, e) == cls0) row(TS + (size_t)m * p.nq * S);
      else row(nh_g(B.test.T) + (bfn(B.test, e) + m) * p.nq * S);
      if (B.local) nh_gw(B.local)[(boff(B.test, e) + m) * B.nct + c] = acc;
      else __hip_atomic_fetch_add(nh_gw(B.out) + (i64)nh_g(B.test.dofs)[boff(B.test, e) + m] * B.nct + c, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int ND>
__global__ __launch_bounds__(NTB) void k_terms(TermsK p) {
  terms_body<ND>(p, blockIdx.x, gridDim.x);
}

// several term lists (samples: the volume and the sides of the boundary) in ONE launch: list i owns the workgroups first[i] .. first[i+1];
// the parameter blocks live in device memory (they do not fit the kernel argument segment together), uniform per workgroup
struct MultiK {
  int count;
  unsigned first[MAXL + 1];
};
template <int ND>
__global__ __launch_bounds__(NTB) void k_terms_multi(const TermsK *__restrict__ lists, MultiK m) {
  int i = 0;
  while (i + 1 < m.count && blockIdx.x >= m.first[i + 1]) ++i;
  terms_body<ND>(lists[i], blockIdx.x - m.first[i], m.first[i + 1] - m.first[i]);
}

// ---- fused bilinear forms ---------------------------------------------------------------------------------------------------------
// term table: per term [kind, field, poly, B[nct][S][ncr][S], L[nct][S]]
#ifdef NH_ABLATION
#define MDBG(p, bit) ((p).debug & (bit))
#define MTICK(i) do { if (tid == 0) { const long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define MDBG(p, bit) 0
#define MTICK(i) do {} while (0)
#endif
struct MTermsK {
  long long *tdbg;  // ablation builds: phase timers
  int debug;  // ablation builds: 1 = no atomics, 2 = no contraction
  i64 nelems;
  const int32_t *elist;
  int nq, eb;
  const double *weights;
  GeomK geom;
  int nfields, nterms, npolys, fct, uesz;
  FieldK fields[MAXF];
  BasisK test, trial;
  int nct, ncr, maxnbt, maxnbr, same;
  const i64 *srowptr;
  const int32_t *emap;
  const i64 *eoff;
  double *values;
  int emap_by_elem;
  unsigned char mask[3][3];
  signed char dpos[3][3];
  int cnt[3], cum[3], tot;
  const double *scale[MAXT];
  int toff[MAXT], poff[MAXP], qoff[MAXT];
  const double *table;
  int tlen;
  double tabarg[TABARG];
};

template <int ND>
__global__ __launch_bounds__(NTB) void k_mterms(MTermsK p) {
  constexpr int S = 1 + ND;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int CS = p.nct * S * p.ncr * S;
  double *tab = lds;
  double *G = lds + p.tlen;                        // [eb * nq][nct][S][ncr][S]: coefficient tensor in reference form times w |J|
  double *U = G + (size_t)p.eb * p.nq * CS;        // [NTB][fct][S]
  i64 *meta = reinterpret_cast<i64 *>(U + (size_t)NTB * p.fct * S);  // [eb][8]: element, first test / trial function, first test dof, element map offset, sizes, table slots
  double *TT = reinterpret_cast<double *>(meta + 8 * p.eb);  // [eb][maxnbt][nq][S]: the test tables of the batch (global loads in the contraction are L1
  double *TR = p.same ? TT : TT + (size_t)p.eb * p.maxnbt * p.nq * S;  // hits, but 16 cycles of the texture path each; LDS reads take 4)
  double *UE = TR + (size_t)p.eb * p.maxnbr * p.nq * S;  // [eb][uesz]: field coefficients
  const int tid = threadIdx.x;
  for (int i = tid; i < p.tlen; i += NTB) tab[i] = p.table ? p.table[i] : p.tabarg[i];
  const int npts = p.eb * p.nq;
  // The tables of the batch live in eb LDS slots that persist across batches (tag = first function of the table): on a structured mesh
  // nearly every element uses the table of its predecessor, so most batches stage nothing.  (Test and trial tables that differ get slot sets
  // of their own.)
  i64 *tagT = reinterpret_cast<i64 *>(UE + (size_t)p.eb * p.uesz), *tagR = tagT + p.eb;
  if (tid < 2 * p.eb) tagT[tid] = -1;
#ifdef NH_ABLATION
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  for (i64 b0 = (i64)blockIdx.x * p.eb; b0 < p.nelems; b0 += (i64)gridDim.x * p.eb) {
    __syncthreads();
    if (tid < p.eb && b0 + tid < p.nelems) {
      const i64 ie = b0 + tid, e = p.elist ? p.elist[ie] : ie;
      const int nbt = bnb(p.test, e), nbr = bnb(p.trial, e);
      i64 *mt = meta + tid * 8;
      mt[0] = e, mt[1] = bfn(p.test, e), mt[2] = bfn(p.trial, e), mt[3] = boff(p.test, e);
      mt[4] = p.eoff ? p.eoff[e] : (p.emap_by_elem ? e : ie) * (i64)nbt * nbr;
      mt[5] = nbt | ((i64)nbr << 16);
    }
    __syncthreads();
    MTICK(0);
    // the table of element el of a batch lives in slot el and stays there: on a structured mesh element el of the NEXT batch of this workgroup
    // nearly always uses the same table (tag = first function), and nothing is staged
    int miss = 0;
    if (tid < p.eb && b0 + tid < p.nelems) {
      i64 *mt = meta + tid * 8;
      const int needT = tagT[tid] != mt[1], needR = !p.same && tagR[tid] != mt[2];
      tagT[tid] = mt[1];
      if (!p.same) tagR[tid] = mt[2];
      mt[6] = tid | (needT ? 0x100 : 0), mt[7] = tid | (needR ? 0x100 : 0);
      miss = needT | needR;
    }
    stage_coeffs(p.fields, p.nfields, p.uesz, UE, p.eb, b0, p.nelems, p.elist, tid);
    double Ji[ND][ND], det = 0;
    if (tid < npts && b0 + tid / p.nq < p.nelems) {
      const int el = tid / p.nq;
      geometry_at<ND>(p.geom, meta[el * 8], tid - el * p.nq, p.nq, nullptr, Ji, det, nullptr);
    }
    miss = __syncthreads_or(miss);
    MTICK(1);
    if (miss) {
      // missing tables -> LDS, eight loads in flight per thread (one load per loop trip is one memory latency per trip)
      auto stage = [&](double *dst, const double *T, int maxnb, int fs, int ms, int shift) {
        const int tsz = maxnb * p.nq * S, total = p.eb * tsz;
        for (int i0 = tid; i0 < total; i0 += 8 * NTB) {
          double v[8];
          int at[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * NTB, el = i / tsz, r = i - el * tsz;
            const bool ok = i < total && b0 + el < p.nelems && (meta[el * 8 + ms] & 0x100) && r < (int)((meta[el * 8 + 5] >> shift) & 0xffff) * p.nq * S;
            at[u] = ok ? (int)(meta[el * 8 + ms] & 0xff) * tsz + r : -1;
            v[u] = ok ? T[meta[el * 8 + fs] * p.nq * S + r] : 0.;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (at[u] >= 0) dst[at[u]] = v[u];
        }
      };
      stage(TT, p.test.T, p.maxnbt, 1, 6, 0);
      if (!p.same) stage(TR, p.trial.T, p.maxnbr, 2, 7, 16);
    }
    __syncthreads();
    MTICK(2);
    for (int t = tid; t < npts; t += NTB) {
      const int el = t / p.nq, q = t - el * p.nq;
      const i64 ie = b0 + el;
      if (ie >= p.nelems) continue;
      const i64 e = meta[el * 8];
      const i64 ip = (p.emap_by_elem ? e : ie) * p.nq + q;  // index of the point in the scale arrays
      if (t != tid) geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);  // (more points than threads: batches of one element)
      const double wdet = nh_g(p.weights)[q] * fabs(det);
      double *u = U + (size_t)tid * p.fct * S;
      eval_fields<ND>(p.fields, p.nfields, e, q, p.nq, Ji, UE + (size_t)el * p.uesz, TT + (size_t)(meta[el * 8 + 6] & 0xff) * p.maxnbt * p.nq * S, u);
      double pv[MAXP];
      eval_polys<S>(tab, p.poff, p.npolys, u, pv);
      double *g = G + (size_t)t * CS;
      if (CS <= 16) {
        // scalar-sized tensors: the sum over the terms stays in registers (an LDS read-modify-write per term and entry is a latency chain)
        double gr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) gr[i] = 0.;
        for (int t2 = 0; t2 < p.nterms; ++t2) {
          const double *H = tab + p.toff[t2];
          const int kind = (int)H[0], fld = (int)H[1], pol = (int)H[2];
          double coef = p.scale[t2] ? p.scale[t2][ip] : 1.;
          if (pol >= 0) coef *= pick(pv, pol);
          if (p.qoff[t2]) coef *= point_factor<S>(tab + p.qoff[t2], u);
          const double *B = H + 3;
          if (kind == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
              if (i < CS) gr[i] += coef * B[i];
          } else {
            const double *uf = u + p.fields[fld].c0 * S;
            double ul[S];
#pragma unroll
            for (int b = 0; b < S; ++b) ul[b] = uf[b];
            if (kind == 1) {
#pragma unroll
              for (int a = 0; a < S; ++a) {
                double sum = 0;
#pragma unroll
                for (int b = 0; b < S; ++b) sum += B[a * S + b] * ul[b];
                gr[a * S] += coef * sum;
              }
            } else {
              const double *L = B + CS;
#pragma unroll
              for (int b = 0; b < S; ++b) {
                double sum = 0;
#pragma unroll
                for (int x = 0; x < S; ++x) sum += B[x * S + b] * ul[x];
#pragma unroll
                for (int a = 0; a < S; ++a) gr[a * S + b] += coef * L[a] * sum;
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i < CS) g[i] = gr[i];
      } else {
        for (int i = 0; i < CS; ++i) g[i] = 0.;
        for (int t2 = 0; t2 < p.nterms; ++t2) {
          const double *H = tab + p.toff[t2];
          const int pol = (int)H[2];
          double coef = p.scale[t2] ? p.scale[t2][ip] : 1.;
          if (pol >= 0) coef *= pick(pv, pol);
          if (p.qoff[t2]) coef *= point_factor<S>(tab + p.qoff[t2], u);
          const double *B = H + 3;
          for (int i = 0; i < CS; ++i) g[i] += coef * B[i];  // (point-dependent kinds are scalar: CS = S * S <= 16)
        }
      }
      // reference form: D[a] = sum_s T[s] X[s][a], X = diag(1, Jinv)  ->  G[c][s][d][t] = w|J| sum_ab X[s][a] Cq[c][a][d][b] X[t][b]
      for (int c = 0; c < p.nct; ++c)
        for (int d = 0; d < p.ncr; ++d) {
          double M[S][S], R[S][S];
#pragma unroll
          for (int a = 0; a < S; ++a)
#pragma unroll
            for (int b = 0; b < S; ++b) M[a][b] = g[((c * S + a) * p.ncr + d) * S + b];
#pragma unroll
          for (int b = 0; b < S; ++b) {
            R[0][b] = M[0][b];
#pragma unroll
            for (int j = 0; j < ND; ++j) {
              double s = 0;
#pragma unroll
              for (int i = 0; i < ND; ++i) s += Ji[j][i] * M[1 + i][b];
              R[1 + j][b] = s;
            }
          }
#pragma unroll
          for (int s2 = 0; s2 < S; ++s2) {
            g[((c * S + s2) * p.ncr + d) * S] = wdet * R[s2][0];
#pragma unroll
            for (int j = 0; j < ND; ++j) {
              double s = 0;
#pragma unroll
              for (int i = 0; i < ND; ++i) s += R[s2][1 + i] * Ji[j][i];
              g[((c * S + s2) * p.ncr + d) * S + 1 + j] = wdet * s;
            }
          }
        }
    }
    __syncthreads();
    MTICK(3);
    // Contraction A[m][n] = sum_q sum_t (sum_s Tt[m][q][s] G[q][s][t]) Tr[n][q][t]: every lane owns NBLK consecutive n of one (element, m)
    // and keeps their sums in registers; per point it forms the S values of the bracket once (S loads of the test table, S * S broadcast reads
    // of G) and spends S loads + S multiply-adds per entry.  Nothing but G lives in LDS, so several workgroups share a CU.
    const int nblk = (p.maxnbr + NBLK - 1) / NBLK;
    for (int k = tid; k < p.eb * p.maxnbt * nblk; k += NTB) {
      const int el = k / (p.maxnbt * nblk), r = k - el * (p.maxnbt * nblk);
      if (b0 + el >= p.nelems) continue;
      const i64 *mt = meta + el * 8;
      const int m = r / nblk, n0 = (r - m * nblk) * NBLK;
      const int nbt = (int)(mt[5] & 0xffff), nbr = (int)(mt[5] >> 16);
      if (m >= nbt || n0 >= nbr) continue;
      const double *Tt = TT + ((size_t)(mt[6] & 0xff) * p.maxnbt + m) * p.nq * S;
      const double *Tr[NBLK];
#pragma unroll
      for (int j = 0; j < NBLK; ++j) Tr[j] = TR + ((size_t)(mt[p.same ? 6 : 7] & 0xff) * p.maxnbr + min(n0 + j, nbr - 1)) * p.nq * S;
      const i64 row = p.test.dofs[mt[3] + m];
      const i64 a0 = p.srowptr[row], len = p.srowptr[row + 1] - a0;
      const int32_t *em = p.emap + mt[4] + m * nbr + n0;
      const double *g0 = G + (size_t)el * p.nq * CS;
      for (int c = 0; c < p.nct; ++c)
        for (int d = 0; d < p.ncr; ++d) {
          if (!p.mask[c][d]) continue;
          double acc[NBLK];
#pragma unroll
          for (int j = 0; j < NBLK; ++j) acc[j] = 0;
          // software pipeline: the LDS reads of point q + 1 are issued before the arithmetic of point q (two waves per SIMD do not hide the
          // ~100 cycles of an LDS read that is waited for right behind its issue)
          const double *gb = g0 + (c * S * p.ncr + d) * S;
          double tt[2][S], gg[2][S][S], tr[2][NBLK][S];
          auto fetch = [&](int q, int b2) {
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) {
              tt[b2][s2] = Tt[q * S + s2];
#pragma unroll
              for (int t2 = 0; t2 < S; ++t2) gg[b2][s2][t2] = gb[(size_t)q * CS + s2 * p.ncr * S + t2];
            }
#pragma unroll
            for (int j = 0; j < NBLK; ++j)
#pragma unroll
              for (int t2 = 0; t2 < S; ++t2) tr[b2][j][t2] = Tr[j][q * S + t2];
          };
          auto point = [&](int b2) {
            double w[S];
#pragma unroll
            for (int t2 = 0; t2 < S; ++t2) w[t2] = 0;
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2)
#pragma unroll
              for (int t2 = 0; t2 < S; ++t2) w[t2] += tt[b2][s2] * gg[b2][s2][t2];
#pragma unroll
            for (int j = 0; j < NBLK; ++j)
#pragma unroll
              for (int t2 = 0; t2 < S; ++t2) acc[j] += w[t2] * tr[b2][j][t2];
          };
          const int nqc = MDBG(p, 2) ? 1 : p.nq;
          fetch(0, 0);
          int q = 0;
          for (; q + 2 <= nqc; q += 2) {
            fetch(q + 1, 1);
            point(0);
            if (q + 2 < nqc) fetch(q + 2, 0);
            point(1);
          }
          if (q < nqc) point(0);
          double *dst = p.values + a0 * p.tot + len * p.cum[c] + p.dpos[c][d];
#pragma unroll
          for (int j = 0; j < NBLK; ++j)
            if (n0 + j < nbr && !MDBG(p, 1)) atomicAdd(dst + (i64)em[j] * p.cnt[c], acc[j]);
#ifdef NH_ABLATION
          if (MDBG(p, 1) && acc[0] == 1.2345e300) dst[0] = 1.;
#endif
        }
    }
    MTICK(4);
  }
#ifdef NH_ABLATION
  if (p.tdbg && tid == 0)
    for (int i = 0; i < 6; ++i) atomicAdd((unsigned long long *)p.tdbg + i, (unsigned long long)tacc[i]);
#endif
}

// ---- fused bilinear forms, scalar, small uniform bases: rows of the local matrix per THREAD, owner-side reduction -------------------------------
// (nh_gather.hip: the thread-per-element pass is ~5x cheaper than a workgroup pipeline for local matrices of this size; here with the term
// list of nh_assemble_matrix_terms evaluated at the point in registers.)  Thread = (element, block of MB test functions); up to NF scalar
// fields on the test basis, their element coefficients in registers; the local matrix goes to the element-major scratch of NH_MATRIX_GATHER.
struct LTermsK {
  i64 nelems;
  const int32_t *elist;
  int nq;
  const double *weights;
  GeomK geom;
  BasisK test, trial;
  int by_elem;
  int nterms, npolys;
  const double *u[2];
  const double *scale[MAXT];
  int toff[MAXT], poff[MAXP], qoff[MAXT];
  int tlen;
  double tabarg[TABARG + TABPAD];  // (+ pad: the two-phase kernel reads every record at its full length)
  double *local;
  int ncls;  // k_local_terms2: element classes whose tables a workgroup stages
};

template <int ND, int NBT, int NBR, int MB, int NF>
__global__ __launch_bounds__(128) void k_local_terms(LTermsK p) {
  constexpr int S = 1 + ND, NG = 1 << ND, NMB = NBT / MB;
  static_assert(NBT % MB == 0, "row blocks");
  __shared__ double tab[TABARG];
  extern __shared__ __attribute__((aligned(16))) double stab[];  // the tables of the element class of the block's first element: [NBT][nq][S] (+ [NBR][nq][S])
  for (int i = threadIdx.x; i < p.tlen; i += blockDim.x) tab[i] = p.tabarg[i];
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 ie_raw = t / NMB;
  const i64 ie = ie_raw < p.nelems ? ie_raw : p.nelems - 1;  // (threads behind the last element shadow it and do not store)
  const int mb = (int)(t - ie_raw * NMB) * MB;
  const i64 e = p.elist ? p.elist[ie] : ie;
  // On a structured mesh nearly all elements of a block share one table (class of the knot span): staged in LDS once per block, the reads of the point loop
  // then take 4 cycles instead of the 16 of the texture path (63 loads per point and thread for configs[3]).  Waves with an element of another class read global
  // memory as before.
  const bool same_tables = p.test.T == p.trial.T && p.test.tab == p.trial.tab;
  i64 fn0t, fn0r;
  {
    const i64 ie0 = min((i64)blockIdx.x * blockDim.x / NMB, p.nelems - 1), e0 = p.elist ? p.elist[ie0] : ie0;
    fn0t = bfn(p.test, e0), fn0r = bfn(p.trial, e0);
    const int nt = NBT * p.nq * S, nr = same_tables ? 0 : NBR * p.nq * S;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) stab[i] = p.test.T[fn0t * p.nq * S + i];
    for (int i = threadIdx.x; i < nr; i += blockDim.x) stab[nt + i] = p.trial.T[fn0r * p.nq * S + i];
  }
  __syncthreads();
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
  double ue[NF > 0 ? NF : 1][NBT];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int n = 0; n < NBT; ++n) ue[f][n] = p.u[f][p.test.dofs[e * (i64)NBT + n]];
  const bool staged = __all(bfn(p.test, e) == fn0t && bfn(p.trial, e) == fn0r);  // (wave-uniform)
  auto points = [&](const double *Tt, const double *Tr) {
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
        const double *tg = p.geom.gT + ((i64)a * p.nq + q) * S;
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
    } else
      geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
    const double w = p.weights[q] * fabs(det);
    const i64 ip = (p.by_elem ? e : ie) * p.nq + q;
    // field values and physical gradients
    double U[NF > 0 ? NF : 1][S];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      double r[S];
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) r[s2] = 0;
#pragma unroll
      for (int n = 0; n < NBT; ++n) {
        const double *T = Tt + ((size_t)n * p.nq + q) * S;
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) r[s2] += T[s2] * ue[f][n];
      }
      U[f][0] = r[0];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        double sum = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j) sum += r[1 + j] * Ji[j][i];
        U[f][1 + i] = sum;
      }
    }
    auto field = [&](int f, int s2) { return NF > 1 && f == 1 ? U[NF > 1 ? 1 : 0][s2] : U[0][s2]; };
    // pointwise polynomials (variables: field values; slot = field index since the fields are scalar)
    double pv[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      pv[k] = 1.;
      if (k < p.npolys) {
        const double *P = tab + p.poff[k];
        const int nv = (int)P[0], nt = (int)P[1];
        double x[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) x[v] = v < nv ? field((int)P[2 + v], 0) : 1.;
        double sum = 0;
        for (int t2 = 0; t2 < nt; ++t2) {
          const double *M = P + 6 + 5 * t2;
          double mm = M[0];
#pragma unroll
          for (int v = 0; v < 4; ++v)
            for (int k2 = (int)M[1 + v]; k2 > 0; --k2) mm *= x[v];
          sum += mm;
        }
        pv[k] = sum;
      }
    }
    // coefficient tensor of the point: sum of the terms
    double Cq[S][S];
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
      for (int b = 0; b < S; ++b) Cq[a][b] = 0;
    for (int t2 = 0; t2 < p.nterms; ++t2) {
      const double *H = tab + p.toff[t2];
      const int kind = (int)H[0], fld = (int)H[1], pol = (int)H[2];
      double coef = p.scale[t2] ? p.scale[t2][ip] : 1.;
      if (pol >= 0) coef *= pick(pv, pol);
      if (p.qoff[t2]) {
        const double *Q = tab + p.qoff[t2];
        double sum = 0;
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
          for (int b = 0; b < S; ++b) sum += Q[2 + a * S + b] * field((int)Q[0], a) * field((int)Q[1], b);
        coef *= sum;
      }
      const double *B = H + 3;
      if (kind == 0) {
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
          for (int b = 0; b < S; ++b) Cq[a][b] += coef * B[a * S + b];
      } else if (kind == 1) {
#pragma unroll
        for (int a = 0; a < S; ++a) {
          double sum = 0;
#pragma unroll
          for (int b = 0; b < S; ++b) sum += B[a * S + b] * field(fld, b);
          Cq[a][0] += coef * sum;
        }
      } else {
        const double *L = B + S * S;
#pragma unroll
        for (int b = 0; b < S; ++b) {
          double sum = 0;
#pragma unroll
          for (int x2 = 0; x2 < S; ++x2) sum += B[x2 * S + b] * field(fld, x2);
#pragma unroll
          for (int a = 0; a < S; ++a) Cq[a][b] += coef * L[a] * sum;
        }
      }
    }
    // trial side premultiplied: W[n][a] = w sum_b Cq[a][b] Dr[n][b]
    double W[NBR][S];
#pragma unroll
    for (int n = 0; n < NBR; ++n) {
      const double *T = Tr + ((size_t)n * p.nq + q) * S;
      double dr[S];
      dr[0] = T[0];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        double sum = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j) sum += T[1 + j] * Ji[j][i];
        dr[1 + i] = sum;
      }
#pragma unroll
      for (int a = 0; a < S; ++a) {
        double sum = 0;
#pragma unroll
        for (int b = 0; b < S; ++b) sum += Cq[a][b] * dr[b];
        W[n][a] = w * sum;
      }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const double *T = Tt + ((size_t)(mb + m) * p.nq + q) * S;
      double dt[S];
      dt[0] = T[0];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        double sum = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j) sum += T[1 + j] * Ji[j][i];
        dt[1 + i] = sum;
      }
#pragma unroll
      for (int n = 0; n < NBR; ++n)
#pragma unroll
        for (int a = 0; a < S; ++a) A[m][n] += dt[a] * W[n][a];
    }
  }
  };
  if (staged) points(stab, same_tables ? stab : stab + NBT * p.nq * S);
  else points(p.test.T + bfn(p.test, e) * p.nq * S, p.trial.T + bfn(p.trial, e) * p.nq * S);
  if (ie_raw >= p.nelems) return;
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int n = 0; n < NBR; ++n) p.local[ie * (NBT * NBR) + (mb + m) * NBR + n] = A[m][n];
}

// k_local_terms re-arranged so that nothing of a point is computed twice (round 4).  The thread-per-row-block kernel above evaluates the geometry, the fields, the
// polynomials and the term list of a point in EVERY row-block thread of the element (three times for configs[3]) and forms the pre-multiplied trial side W[n][a] for all
// trial functions in each of them: ~14 k instructions per thread where the contraction needs ~3 k.  Here a workgroup of TPE * 64 threads owns 64 elements and walks over the
// points in chunks of TPE:
//   phase A: wave w takes point q0 + w of the chunk, lane l element l -- geometry, fields, polynomials, term list ONCE per (element, point); the coefficient tensor is
//            folded with the weight and the inverse Jacobian, C'[a][b] = w sum P[a][a'] Cq[a'][b'] P[b][b'] (P = diag(1, J^-1)), so that the contraction below needs the
//            REFERENCE tables only; C' goes to LDS (double buffered: one barrier per chunk).  The point is wave-uniform: every table read of the phase is a broadcast.
//   phase B: thread (element, block of NBK trial functions) adds the chunk's points to its NBT x NBK entries: W[j][a] = sum_b C'[a][b] T_r[n_j][q][b] (its own trial
//            functions only), A[m][j] += sum_a T_t[m][q][a] W[j][a]; the point is uniform over the workgroup.
#ifndef NH_LT2_WPE
#define NH_LT2_WPE 3
#endif
#ifndef NH_LT2_DB
#define NH_LT2_DB 0  // 1: two C' buffers (one barrier per chunk, 14 kB more LDS)
#endif
#ifndef NH_LT2_CLS
#define NH_LT2_CLS 6
#endif
#ifndef NH_LT2_GRP
#define NH_LT2_GRP 3  // table rows read ahead of their arithmetic
#endif
constexpr int LT2_EPB = 64;
constexpr int LT2_CLS = NH_LT2_CLS;  // most element classes whose tables a workgroup stages (a run of 64 consecutive elements of a wide tensor spline mesh meets two)
template <int ND, int NBT, int NBR, int NBK, int NF, bool ISO>
__global__ __launch_bounds__(LT2_EPB *(NBR / NBK)) __attribute__((amdgpu_waves_per_eu(NH_LT2_WPE, NH_LT2_WPE))) void k_local_terms2(LTermsK p) {
  constexpr int S = 1 + ND, NG = 1 << ND, TPE = NBR / NBK, QC = TPE, NT = LT2_EPB * TPE, SS = S * S;
  static_assert(NBR % NBK == 0, "trial blocks");
  // (the term list is read from the kernel arguments with scalar loads, uniform over the launch.  Measured and not kept: the list in two registers per lane read with
  // v_readlane -- no memory round trips, but 0.48 instead of 0.345 ms: four readlanes and a select per entry, 134 spilled scalar registers)
  const double *tab = p.tabarg;
  extern __shared__ __attribute__((aligned(16))) double stab[];  // tables of up to LT2_CLS element classes of the workgroup, then the C' buffers [2][QC][64][S * S]
  __shared__ i64 clsT[LT2_CLS], clsR[LT2_CLS];
  __shared__ int ncls, slotS[LT2_EPB];
  const bool same_tables = p.test.T == p.trial.T && p.test.tab == p.trial.tab;
  const int ntab = ((NBT * p.nq * S + (same_tables ? 0 : NBR * p.nq * S)) + 1) & ~1;
  double *cbuf = stab + p.ncls * ntab;
  constexpr int NCB = NH_LT2_DB ? 2 : 1, UES = (NF * NBT) | 1;  // C' buffers; odd pitch of an element's field coefficients
  double *ueS = cbuf + NCB * QC * LT2_EPB * SS;                 // [64][UES]: the phase-A lane reads them per point (in registers they cost a third wave per SIMD)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const i64 ie0 = (i64)blockIdx.x * LT2_EPB;
  // phase-A element of this lane / phase-B element of this thread
  const i64 ieA = min(ie0 + lane, p.nelems - 1), eA = p.elist ? p.elist[ieA] : ieA;
  const int elB = threadIdx.x / TPE, nb0 = (threadIdx.x - elB * TPE) * NBK;
  const i64 ieB_raw = ie0 + elB, ieB = min(ieB_raw, p.nelems - 1), eB = p.elist ? p.elist[ieB] : ieB;
  // The element classes (table pairs) of the 64 elements: on a tensor spline mesh a run of consecutive elements meets two or three (the first / last knot span of a row
  // differs from the interior ones); with ONE staged class every fourth workgroup of a 512-element row fell back to global table reads and took ten times as long.
  // Wave 0 numbers the distinct classes (leader election over the lanes not yet numbered); more than p.ncls (the launcher sizes the LDS for 2 on large meshes, LT2_CLS on small ones): global reads for the whole workgroup.
  if (wave == 0) {
    const i64 ft = bfn(p.test, eA), fr = bfn(p.trial, eA);
    int slot = -1, k = 0;
    unsigned long long todo = __ballot(1);
    while (todo && k < p.ncls) {
      const int leader = __ffsll((long long)todo) - 1;
      const i64 lt = __shfl(ft, leader), lr = __shfl(fr, leader);
      if (lane == leader) clsT[k] = lt, clsR[k] = lr;
      if (ft == lt && fr == lr) slot = k;
      todo = __ballot(slot < 0);
      ++k;
    }
    slotS[lane] = slot;
    if (lane == 0) ncls = todo ? -1 : k;
  }
  __syncthreads();
  const int staged = ncls > 0;
  if (staged) {
    const int nt = NBT * p.nq * S, nr = same_tables ? 0 : NBR * p.nq * S;
    for (int c = 0; c < ncls; ++c) {
      for (int i = threadIdx.x; i < nt; i += NT) stab[c * ntab + i] = p.test.T[clsT[c] * p.nq * S + i];
      for (int i = threadIdx.x; i < nr; i += NT) stab[c * ntab + nt + i] = p.trial.T[clsR[c] * p.nq * S + i];
    }
  }
  const int slotA = staged ? slotS[lane] : 0, slotB = staged ? slotS[elB] : 0;
  __syncthreads();
  constexpr bool iso = ISO;  // (p.geom.kind == NH_GEOM_ISO && p.geom.ngb == NG: the launcher's choice)
  double X[ISO ? NG : 1][ND];
  if (iso) {
#pragma unroll
    for (int a = 0; a < NG; ++a) {
      const i64 v = p.geom.gdofs[eA * NG + a];
#pragma unroll
      for (int i = 0; i < ND; ++i) X[ISO ? a : 0][i] = p.geom.verts[v * ND + i];
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int n = 0; n < NBT; ++n) ueS[lane * UES + f * NBT + n] = p.u[f][p.test.dofs[eA * (i64)NBT + n]];
  }
  const double *ue = ueS + lane * UES;
  __syncthreads();
  double A[NBT][NBK];
#pragma unroll
  for (int m = 0; m < NBT; ++m)
#pragma unroll
    for (int j = 0; j < NBK; ++j) A[m][j] = 0;

  auto body = [&](const double *TtA, const double *TtB, const double *TrB) {
    for (int q0 = 0, buf = 0; q0 < p.nq; q0 += QC, buf ^= NCB - 1) {
      double *cq = cbuf + buf * (QC * LT2_EPB * SS);
      if (NCB == 1 && q0) __syncthreads();  // (single buffer: phase B of the previous chunk has read it)
      const int q = q0 + wave;
#ifdef NH_LT2_SKIPA  // (timing experiments only)
      if (q < p.nq && p.nq < 0) {
#else
      if (q < p.nq) {  // ---- phase A: (element = lane, point = q0 + wave)
#endif
        double Ji[ND][ND], det;
        if (iso) {
          double J[ND][ND];
#pragma unroll
          for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int j = 0; j < ND; ++j) J[i][j] = 0;
#pragma unroll
          for (int a = 0; a < NG; ++a) {
            const double *tg = p.geom.gT + ((i64)a * p.nq + q) * S;
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
              for (int j = 0; j < ND; ++j) J[i][j] += X[ISO ? a : 0][i] * tg[1 + j];
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
          if (p.geom.nograd) {  // (no term reads a gradient slot: the inverse -- NaN on an exactly singular element -- must not reach the folded tensor; as nh_geom.inc)
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
              for (int j = 0; j < ND; ++j) Ji[i][j] = 0.;
          }
        } else
          geometry_at<ND>(p.geom, eA, q, p.nq, nullptr, Ji, det, nullptr);
        const double w = p.weights[q] * fabs(det);
        const i64 ip = (p.by_elem ? eA : ieA) * p.nq + q;
        double U[NF > 0 ? NF : 1][S];
        {
          double r[NF > 0 ? NF : 1][S];
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) r[f][s2] = 0;
#pragma unroll
          for (int n = 0; n < NBT; ++n) {  // (one pass over the table rows of the point for all fields)
            const double *T = TtA + ((size_t)n * p.nq + q) * S;
            double tv[S];
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) tv[s2] = T[s2];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
              for (int s2 = 0; s2 < S; ++s2) r[f][s2] += tv[s2] * ue[f * NBT + n];
            if (n % NH_LT2_GRP == NH_LT2_GRP - 1) __builtin_amdgcn_sched_barrier(0);  // (bounds the reads in flight: hoisted all at once they cost a wave of occupancy)
          }
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            U[f][0] = r[f][0];
#pragma unroll
            for (int i = 0; i < ND; ++i) {
              double sum = 0;
#pragma unroll
              for (int j = 0; j < ND; ++j) sum += r[f][1 + j] * Ji[j][i];
              U[f][1 + i] = sum;
            }
          }
        }
        auto field = [&](int f, int s2) { return NF > 1 && f == 1 ? U[NF > 1 ? 1 : 0][s2] : U[0][s2]; };
        double pv[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
          pv[k] = 1.;
          if (k < p.npolys) {
            // (every record is read whole before it is used: one scalar round trip per record -- written entry by entry inside the loops, each power and each coefficient
            // was a load + wait of its own and the phase was bound by them)
            const double *P = tab + p.poff[k];
            const double h0 = P[0], h1 = P[1], s0 = P[2], s1 = P[3], s2v = P[4], s3 = P[5];
            const int nv = (int)h0, nt = (int)h1;
            const int sl[4] = {(int)s0, (int)s1, (int)s2v, (int)s3};
            double x[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) x[v] = v < nv ? field(sl[v], 0) : 1.;
            double sum = 0;
            for (int t2 = 0; t2 < nt; ++t2) {
              const double *M = P + 6 + 5 * t2;
              const double c0 = M[0], e0 = M[1], e1 = M[2], e2 = M[3], e3 = M[4];
              const int pw[4] = {(int)e0, (int)e1, (int)e2, (int)e3};
              double mm = c0;
#pragma unroll
              for (int v = 0; v < 4; ++v) {  // x^pw by squaring, pw < 64 (branch-free over the bits that occur: pw is uniform)
                double xp = x[v];
                for (int e = pw[v]; e > 0; e >>= 1) {
                  if (e & 1) mm *= xp;
                  xp *= xp;
                }
              }
              sum += mm;
            }
            pv[k] = sum;
          }
        }
        double Cq[S][S];
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
          for (int b = 0; b < S; ++b) Cq[a][b] = 0;
        for (int t2 = 0; t2 < p.nterms; ++t2) {
          const double *H = tab + p.toff[t2];
          double rec[3 + SS + S];  // [kind, field, poly, B[S][S], L[S] (kind 2)]: read whole (the tail of a shorter record is the head of the next entry of the table)
#pragma unroll
          for (int i2 = 0; i2 < 3 + SS + S; ++i2) rec[i2] = H[i2];
          const int kind = (int)rec[0], fld = (int)rec[1], pol = (int)rec[2];
          double coef = p.scale[t2] ? p.scale[t2][ip] : 1.;
          if (pol >= 0) coef *= pick(pv, pol);
          if (p.qoff[t2]) {
            const double *Q = tab + p.qoff[t2];
            double qr[2 + SS];
#pragma unroll
            for (int i2 = 0; i2 < 2 + SS; ++i2) qr[i2] = Q[i2];
            double sum = 0;
            const int fa = (int)qr[0], fb = (int)qr[1];
#pragma unroll
            for (int a = 0; a < S; ++a)
#pragma unroll
              for (int b = 0; b < S; ++b) sum += qr[2 + a * S + b] * field(fa, a) * field(fb, b);
            coef *= sum;
          }
          const double *B = rec + 3;
          if (kind == 0) {
#pragma unroll
            for (int a = 0; a < S; ++a)
#pragma unroll
              for (int b = 0; b < S; ++b) Cq[a][b] += coef * B[a * S + b];
          } else if (kind == 1) {
#pragma unroll
            for (int a = 0; a < S; ++a) {
              double sum = 0;
#pragma unroll
              for (int b = 0; b < S; ++b) sum += B[a * S + b] * field(fld, b);
              Cq[a][0] += coef * sum;
            }
          } else {
            const double *L = B + S * S;
#pragma unroll
            for (int b = 0; b < S; ++b) {
              double sum = 0;
#pragma unroll
              for (int x2 = 0; x2 < S; ++x2) sum += B[x2 * S + b] * field(fld, x2);
#pragma unroll
              for (int a = 0; a < S; ++a) Cq[a][b] += coef * L[a] * sum;
            }
          }
        }
        // fold: C'[a][b] = w sum_{a' b'} P[a][a'] Cq[a'][b'] P[b][b'], P = diag(1, J^-1) (dt[1 + i] = sum_j T[1 + j] Ji[j][i])
        double H1[S][S];  // H1[a][b'] = sum_a' P[a][a'] Cq[a'][b']
#pragma unroll
        for (int b = 0; b < S; ++b) {
          H1[0][b] = Cq[0][b];
#pragma unroll
          for (int j = 0; j < ND; ++j) {
            double sum = 0;
#pragma unroll
            for (int i = 0; i < ND; ++i) sum += Ji[j][i] * Cq[1 + i][b];
            H1[1 + j][b] = sum;
          }
        }
        double *o = cq + ((size_t)wave * LT2_EPB + lane) * SS;
#pragma unroll
        for (int a = 0; a < S; ++a) {
          o[a * S] = w * H1[a][0];
#pragma unroll
          for (int j = 0; j < ND; ++j) {
            double sum = 0;
#pragma unroll
            for (int i = 0; i < ND; ++i) sum += H1[a][1 + i] * Ji[j][i];
            o[a * S + 1 + j] = w * sum;
          }
        }
      }
      __syncthreads();
      // ---- phase B: (element, trial block) x the points of the chunk
#ifdef NH_LT2_SKIPB
      const int nqc = p.nq < 0 ? 1 : 0;
#else
      const int nqc = min(QC, p.nq - q0);
#endif
      for (int ql = 0; ql < nqc; ++ql) {
        const int qb = q0 + ql;
        const double *c = cq + ((size_t)ql * LT2_EPB + elB) * SS;
        double Cp[S][S];
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
          for (int b = 0; b < S; ++b) Cp[a][b] = c[a * S + b];
        double W[NBK][S];
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
          const double *T = TrB + ((size_t)(nb0 + j) * p.nq + qb) * S;
          double tr[S];
#pragma unroll
          for (int b = 0; b < S; ++b) tr[b] = T[b];
#pragma unroll
          for (int a = 0; a < S; ++a) {
            double sum = 0;
#pragma unroll
            for (int b = 0; b < S; ++b) sum += Cp[a][b] * tr[b];
            W[j][a] = sum;
          }
        }
#pragma unroll
        for (int m = 0; m < NBT; ++m) {
          const double *T = TtB + ((size_t)m * p.nq + qb) * S;
          double tt[S];
#pragma unroll
          for (int a = 0; a < S; ++a) tt[a] = T[a];
#pragma unroll
          for (int j = 0; j < NBK; ++j)
#pragma unroll
            for (int a = 0; a < S; ++a) A[m][j] += tt[a] * W[j][a];
          if (m % NH_LT2_GRP == NH_LT2_GRP - 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  if (staged) body(stab + slotA * ntab, stab + slotB * ntab, stab + slotB * ntab + (same_tables ? 0 : NBT * p.nq * S));
  else body(p.test.T + bfn(p.test, eA) * p.nq * S, p.test.T + bfn(p.test, eB) * p.nq * S, p.trial.T + bfn(p.trial, eB) * p.nq * S);
  if (ieB_raw >= p.nelems) return;
#pragma unroll
  for (int m = 0; m < NBT; ++m)
#pragma unroll
    for (int j = 0; j < NBK; ++j) p.local[ieB * (NBT * NBR) + m * NBR + nb0 + j] = A[m][j];
}

// ---- fused linear forms, scalar blocks on small uniform bases: ONE THREAD per element ------------------------------------------------------------
// The counterpart of k_local_terms for residuals: up to two scalar output blocks on ONE test basis (the residual blocks of a two-field system), up to
// three scalar fields on the same basis with their element coefficients in registers; everything of a point -- field values, polynomial factors, the
// integrand of every block -- stays in registers, and the NB sums per block go out with one atomic each.
struct VTermsK {
  i64 nelems;
  const int32_t *elist;
  int nq;
  const double *weights;
  GeomK geom;
  BasisK test;
  int nterms, npolys, nblocks;
  const double *u[3];
  double *out[2];
  double *local[2];  // element-major local vectors instead of atomics (deterministic scatter)
  const double *scale[MAXT];
  int toff[MAXT], poff[MAXP], qoff[MAXT];
  int tlen;
  double tabarg[TABARG + TABPAD];
  int ncls;  // k_local_vterms2: element classes whose tables a workgroup stages
};

template <int ND, int NB, int NF>
__global__ __launch_bounds__(128) void k_local_vterms(VTermsK p) {
  constexpr int S = 1 + ND, NG = 1 << ND;
  __shared__ double tab[TABARG];
  for (int i = threadIdx.x; i < p.tlen; i += blockDim.x) tab[i] = p.tabarg[i];
  __syncthreads();
  const i64 ie = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (ie >= p.nelems) return;
  const i64 e = p.elist ? p.elist[ie] : ie;
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
  int dofs[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) dofs[n] = p.test.dofs[e * (i64)NB + n];
  double ue[NF > 0 ? NF : 1][NB];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int n = 0; n < NB; ++n) ue[f][n] = p.u[f][dofs[n]];
  double r[2][NB];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int m = 0; m < NB; ++m) r[b][m] = 0;
  const double *Tt = p.test.T + bfn(p.test, e) * p.nq * S;
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
        const double *tg = p.geom.gT + ((i64)a * p.nq + q) * S;
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
    } else
      geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
    const double w = p.weights[q] * fabs(det);
    const i64 ip = ie * p.nq + q;
    // reference tables of the point (shared by the fields and the test side), field values and physical gradients
    double T[NB][S];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) T[n][s2] = Tt[((size_t)n * p.nq + q) * S + s2];
    double U[NF > 0 ? NF : 1][S];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      double rr[S];
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) rr[s2] = 0;
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) rr[s2] += T[n][s2] * ue[f][n];
      U[f][0] = rr[0];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        double sum = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j) sum += rr[1 + j] * Ji[j][i];
        U[f][1 + i] = sum;
      }
    }
    auto field = [&](int f, int s2) { return NF > 2 && f == 2 ? U[NF > 2 ? 2 : 0][s2] : NF > 1 && f == 1 ? U[NF > 1 ? 1 : 0][s2] : U[0][s2]; };
    double pv[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      pv[k] = 1.;
      if (k < p.npolys) {
        const double *P = tab + p.poff[k];
        const int nv = (int)P[0], nt = (int)P[1];
        double x[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) x[v] = v < nv ? field((int)P[2 + v], 0) : 1.;
        double sum = 0;
        for (int t2 = 0; t2 < nt; ++t2) {
          const double *M = P + 6 + 5 * t2;
          double mm = M[0];
#pragma unroll
          for (int v = 0; v < 4; ++v)
            for (int k2 = (int)M[1 + v]; k2 > 0; --k2) mm *= x[v];
          sum += mm;
        }
        pv[k] = sum;
      }
    }
    // integrand of both blocks: sum of the terms
    double G[2][S];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int a = 0; a < S; ++a) G[b][a] = 0;
    for (int t2 = 0; t2 < p.nterms; ++t2) {
      const double *H = tab + p.toff[t2];
      const int blk = (int)H[0], fld = (int)H[1], pol = (int)H[2], hasC = (int)H[3], hasf = (int)H[4];
      double coef = p.scale[t2] ? p.scale[t2][ip] : 1.;
      if (pol >= 0) coef *= pick(pv, pol);
      if (p.qoff[t2]) {
        const double *Q = tab + p.qoff[t2];
        double sum = 0;
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
          for (int b = 0; b < S; ++b) sum += Q[2 + a * S + b] * field((int)Q[0], a) * field((int)Q[1], b);
        coef *= sum;
      }
      const double *fv = H + 5, *C = fv + S;
#pragma unroll
      for (int a = 0; a < S; ++a) {
        double sum = hasf ? fv[a] : 0.;
        if (hasC) {
#pragma unroll
          for (int b = 0; b < S; ++b) sum += C[a * S + b] * field(fld, b);
        }
        if (blk == 0) G[0][a] += coef * sum;
        else G[1][a] += coef * sum;
      }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (b >= p.nblocks) break;
      double g[S];
      g[0] = w * G[b][0];
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        double sum = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) sum += Ji[j][i] * G[b][1 + i];
        g[1 + j] = w * sum;
      }
#pragma unroll
      for (int m = 0; m < NB; ++m)
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) r[b][m] += T[m][s2] * g[s2];
    }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (b >= p.nblocks) break;
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      if (p.local[b]) p.local[b][e * NB + m] = r[b][m];
      else atomicAdd(p.out[b] + dofs[m], r[b][m]);
    }
  }
}

// k_local_vterms for meshes of 2^14 elements and more, with the parallelism of the batched kernel and none of its run-time loops (round 4): a workgroup of LV2_NW waves
// owns 64 elements -- lane l of every wave is element l, wave w takes the points w, w + LV2_NW, ... -- so that the point is uniform over a wave (every table read a broadcast
// from the staged tables of the element's class), each thread keeps partial sums r[block][m] of ITS points, and the waves' partial sums are added in wave order through
// LDS at the end (deterministic).  configs[3] residual (262 144 elements, 25 points, 3 fields, 2 blocks): k_terms_multi 0.56 ms with 257 M wave instructions, 38 % of them
// scalar bookkeeping of its generic loops over blocks / components / slots.
constexpr int LV2_NW = 4;
constexpr i64 LV2_MIN_ELEMS = 1 << 14;  // lists of this size get a launch of their own (nh_assemble_terms_multi)
#ifndef NH_LV2_WPE
#define NH_LV2_WPE 2
#endif
template <int ND, int NB, int NF>
__global__ __launch_bounds__(64 * LV2_NW) __attribute__((amdgpu_waves_per_eu(NH_LV2_WPE, NH_LV2_WPE))) void k_local_vterms2(VTermsK p) {
  constexpr int S = 1 + ND, NG = 1 << ND, NT = 64 * LV2_NW, UES = (NF * NB) | 1, RS = (2 * NB) | 1;
  const double *tab = p.tabarg;
  extern __shared__ __attribute__((aligned(16))) double stab[];  // [ncls][NB][nq][S] tables | ue [64][UES], later the partial sums [LV2_NW - 1][64][RS]
  __shared__ i64 clsT[LT2_CLS];
  __shared__ int ncls, slotS[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntab = (NB * p.nq * S + 1) & ~1;
  double *ueS = stab + p.ncls * ntab;
  const i64 ie_raw = (i64)blockIdx.x * 64 + lane, ie = min(ie_raw, p.nelems - 1), e = p.elist ? p.elist[ie] : ie;
  if (wave == 0) {  // element classes of the 64 elements (k_local_terms2)
    const i64 ft = bfn(p.test, e);
    int slot = -1, k = 0;
    unsigned long long todo = __ballot(1);
    while (todo && k < p.ncls) {
      const int leader = __ffsll((long long)todo) - 1;
      const i64 lt = __shfl(ft, leader);
      if (lane == leader) clsT[k] = lt;
      if (ft == lt) slot = k;
      todo = __ballot(slot < 0);
      ++k;
    }
    slotS[lane] = slot;
    if (lane == 0) ncls = todo ? -1 : k;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int n = 0; n < NB; ++n) ueS[lane * UES + f * NB + n] = p.u[f][p.test.dofs[e * (i64)NB + n]];
  }
  __syncthreads();
  const int staged = ncls > 0;
  if (staged)
    for (int c = 0; c < ncls; ++c)
      for (int i = threadIdx.x; i < NB * p.nq * S; i += NT) stab[c * ntab + i] = p.test.T[clsT[c] * p.nq * S + i];
  const int slot = staged ? slotS[lane] : 0;
  __syncthreads();
  const bool iso = p.geom.kind == NH_GEOM_ISO && p.geom.ngb == NG;
  const bool perpoint = iso || p.geom.kind == NH_GEOM_TAB;
  double Jc[ND][ND], detc = 0;  // geometry that does not depend on the point (boxes): once
  if (!perpoint) geometry_at<ND>(p.geom, e, 0, p.nq, nullptr, Jc, detc, nullptr);
  const double *ue = ueS + lane * UES;
  double r[2][NB];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int m = 0; m < NB; ++m) r[b][m] = 0;
  auto body = [&](const double *Tt) {
    for (int q = wave; q < p.nq; q += LV2_NW) {
      double Ji[ND][ND], det;
      if (perpoint) geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
      else {
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
          for (int j = 0; j < ND; ++j) Ji[i][j] = Jc[i][j];
        det = detc;
      }
      const double w = p.weights[q] * fabs(det);
      const i64 ip = ie * p.nq + q;
      const double *Tq = Tt + (size_t)q * S;  // row n of the point: Tq[n * nq * S + s] (read for the fields and again for the test side: broadcasts, not registers)
      double U[NF > 0 ? NF : 1][S];
      {
        double rr[NF > 0 ? NF : 1][S];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
          for (int s2 = 0; s2 < S; ++s2) rr[f][s2] = 0;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          double tn[S];
#pragma unroll
          for (int s2 = 0; s2 < S; ++s2) tn[s2] = Tq[(size_t)n * p.nq * S + s2];
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const double un = ue[f * NB + n];
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) rr[f][s2] += tn[s2] * un;
          }
          if (n % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          U[f][0] = rr[f][0];
#pragma unroll
          for (int i = 0; i < ND; ++i) {
            double sum = 0;
#pragma unroll
            for (int j = 0; j < ND; ++j) sum += rr[f][1 + j] * Ji[j][i];
            U[f][1 + i] = sum;
          }
        }
      }
      // (a select of VALUES: written as a select of array elements it becomes one load through a selected address, and U lives in scratch memory)
      auto field = [&](int f, int s2) {
        const double u0 = U[0][s2], u1 = U[NF > 1 ? 1 : 0][s2], u2 = U[NF > 2 ? 2 : 0][s2];
        double v = u0;
        if (NF > 1) v = f == 1 ? u1 : v;
        if (NF > 2) v = f == 2 ? u2 : v;
        return v;
      };
      double pv[MAXP];
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        pv[k] = 1.;
        if (k < p.npolys) {  // (records read whole before use: one scalar round trip each)
          const double *P = tab + p.poff[k];
          const double h0 = P[0], h1 = P[1], s0 = P[2], s1 = P[3], s2v = P[4], s3 = P[5];
          const int nv = (int)h0, nt = (int)h1;
          const int sl[4] = {(int)s0, (int)s1, (int)s2v, (int)s3};
          double x[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) x[v] = v < nv ? field(sl[v], 0) : 1.;
          double sum = 0;
          for (int t2 = 0; t2 < nt; ++t2) {
            const double *M = P + 6 + 5 * t2;
            const double c0 = M[0], e0 = M[1], e1 = M[2], e2 = M[3], e3 = M[4];
            const int pw[4] = {(int)e0, (int)e1, (int)e2, (int)e3};
            double mm = c0;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              double xp = x[v];
              for (int ex = pw[v]; ex > 0; ex >>= 1) {
                if (ex & 1) mm *= xp;
                xp *= xp;
              }
            }
            sum += mm;
          }
          pv[k] = sum;
        }
      }
      double G[2][S];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < S; ++a) G[b][a] = 0;
      for (int t2 = 0; t2 < p.nterms; ++t2) {
        const double *H = tab + p.toff[t2];
        double rec[5 + S + S * S];  // [block, field, poly, hasC, hasf, f[S], C[S][S]]
#pragma unroll
        for (int i2 = 0; i2 < 5 + S + S * S; ++i2) rec[i2] = H[i2];
        const int blk = (int)rec[0], fld = (int)rec[1], pol = (int)rec[2], hasC = (int)rec[3], hasf = (int)rec[4];
        double coef = p.scale[t2] ? p.scale[t2][ip] : 1.;
        if (pol >= 0) coef *= pick(pv, pol);
        if (p.qoff[t2]) {
          const double *Q = tab + p.qoff[t2];
          double qr[2 + S * S];
#pragma unroll
          for (int i2 = 0; i2 < 2 + S * S; ++i2) qr[i2] = Q[i2];
          const int fa = (int)qr[0], fb = (int)qr[1];
          double sum = 0;
#pragma unroll
          for (int a = 0; a < S; ++a)
#pragma unroll
            for (int b = 0; b < S; ++b) sum += qr[2 + a * S + b] * field(fa, a) * field(fb, b);
          coef *= sum;
        }
        const double *fv = rec + 5, *C = fv + S;
#pragma unroll
        for (int a = 0; a < S; ++a) {
          double sum = hasf ? fv[a] : 0.;
          if (hasC) {
#pragma unroll
            for (int b = 0; b < S; ++b) sum += C[a * S + b] * field(fld, b);
          }
          if (blk == 0) G[0][a] += coef * sum;
          else G[1][a] += coef * sum;
        }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (b >= p.nblocks) break;
        double g[S];
        g[0] = w * G[b][0];
#pragma unroll
        for (int j = 0; j < ND; ++j) {
          double sum = 0;
#pragma unroll
          for (int i = 0; i < ND; ++i) sum += Ji[j][i] * G[b][1 + i];
          g[1 + j] = w * sum;
        }
#pragma unroll
        for (int m = 0; m < NB; ++m) {
#pragma unroll
          for (int s2 = 0; s2 < S; ++s2) r[b][m] += Tq[(size_t)m * p.nq * S + s2] * g[s2];
          if (m % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  if (staged) body(stab + slot * ntab);
  else body(p.test.T + bfn(p.test, e) * p.nq * S);
  // partial sums of waves 1 .. LV2_NW - 1 -> LDS (over the field coefficients, which nobody reads any more), added by wave 0 in wave order
  __syncthreads();
  double *part = ueS;
  if (wave > 0) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int m = 0; m < NB; ++m) part[((wave - 1) * 64 + lane) * RS + b * NB + m] = r[b][m];
  }
  __syncthreads();
  if (wave > 0 || ie_raw >= p.nelems) return;
  for (int w2 = 0; w2 < LV2_NW - 1; ++w2)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int m = 0; m < NB; ++m) r[b][m] += part[(w2 * 64 + lane) * RS + b * NB + m];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (b >= p.nblocks) break;
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      if (p.local[b]) p.local[b][e * NB + m] = r[b][m];
      else atomicAdd(p.out[b] + p.test.dofs[e * (i64)NB + m], r[b][m]);
    }
  }
}

int check_geom2(const nh_geometry &g) {
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

int max_nb2(const nh_basis &b, i64 nelems, int *out, hipStream_t s) {
  if (b.nb > 0 || !b.off_dev) {
    *out = b.nb;
    return NH_OK;
  }
  std::vector<i64> h(nelems + 1);
  // (on the caller's stream: offsets produced asynchronously on that stream are complete when read)
  NH_CHECK_HIP(hipMemcpyAsync(h.data(), b.off_dev, sizeof(i64) * (nelems + 1), hipMemcpyDeviceToHost, s));
  NH_CHECK_HIP(hipStreamSynchronize(s));
  i64 m = 0;
  for (i64 e = 0; e < nelems; ++e) m = std::max(m, h[e + 1] - h[e]);
  *out = (int)m;
  return NH_OK;
}

// polynomial k of the argument list -> term table
int push_poly(const nh_point_poly &P, const nh_field *fields, int nfields, const FieldK *fk, int k, std::vector<double> &tab, int *off) {
  NH_REQUIRE(P.nvars >= 0 && P.nvars <= 4 && P.nterms >= 0 && P.nterms <= 64 && (P.nterms == 0 || (P.coeffs_host && (P.nvars == 0 || P.powers_host))), "polynomial %d: at most 4 variables and 64 terms", k);
  *off = (int)tab.size();
  tab.push_back(P.nvars), tab.push_back(P.nterms);
  for (int v = 0; v < 4; ++v) {
    int slot = 0;
    if (v < P.nvars) {
      NH_REQUIRE(P.field[v] >= 0 && P.field[v] < nfields && P.comp[v] >= 0 && P.comp[v] < fields[P.field[v]].ncomp, "polynomial %d: variable %d refers to a missing field component", k, v);
      slot = fk[P.field[v]].c0 + P.comp[v];
    }
    tab.push_back(slot);
  }
  for (int t = 0; t < P.nterms; ++t) {
    tab.push_back(P.coeffs_host[t]);
    for (int v = 0; v < 4; ++v) {
      const int pw = v < P.nvars ? P.powers_host[t * P.nvars + v] : 0;
      NH_REQUIRE(pw >= 0 && pw < 64, "polynomial %d: power out of range", k);
      tab.push_back(pw);
    }
  }
  return NH_OK;
}

// point factor record of a term -> table; *off = 0: none
int push_qs(int ft, int fr, const double *B, int S, const nh_field *fields, int nfields, const FieldK *fk, int t, std::vector<double> &tab, int *off) {
  *off = 0;
  if (!B) return NH_OK;
  NH_REQUIRE(ft >= 0 && ft < nfields && fr >= 0 && fr < nfields && fields[ft].ncomp == 1 && fields[fr].ncomp == 1, "term %d: the point factor U_t . B . U_r needs two scalar fields", t);
  *off = (int)tab.size();
  tab.push_back(fk[ft].c0), tab.push_back(fk[fr].c0);
  for (int i = 0; i < S * S; ++i) tab.push_back(B[i]);
  return NH_OK;
}

// small tables travel with the kernel arguments; large ones go through a device buffer owned by the library (the pageable host vector
// must outlive the copy, and earlier launches on the stream may still read the buffer: synchronous)
int place_table(const std::vector<double> &tab, double *tabarg, const double **table, hipStream_t s) {
  if (tab.size() <= (size_t)TABARG) {
    *table = nullptr;
    std::copy(tab.begin(), tab.end(), tabarg);
    return NH_OK;
  }
  static double *dtab = nullptr;
  static size_t dcap = 0;
  if (tab.size() > dcap) {
    if (dtab) NH_CHECK_HIP(hipFree(dtab));
    dcap = 2 * tab.size();
    NH_CHECK_HIP(hipMalloc((void **)&dtab, dcap * sizeof(double)));
  }
  NH_CHECK_HIP(hipStreamSynchronize(s));
  NH_CHECK_HIP(hipMemcpy(dtab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
  *table = dtab;
  return NH_OK;
}

// owner-side reduction for scalar blocks on small uniform bases (NH_MATRIX_GATHER): thread-per-element pass + gather; *done = false: not applicable
int local_terms(const nh_matrix_terms_args *a, const MTermsK &m, const std::vector<double> &tab, bool *done, hipStream_t s) {
  *done = false;
  if (!a->pattern || a->pattern->nelems != a->nelems || (a->elist_dev && (a->flags & NH_MATRIX_EMAP_BY_ELEMENT))) return NH_OK;
  if (a->nct != 1 || a->ncr != 1 || a->test.off_dev || a->trial.off_dev || !a->test.nb || !a->trial.nb || a->nfields > 2 || tab.size() > (size_t)TABARG) return NH_OK;
  for (int f = 0; f < a->nfields; ++f)
    if (!m.fields[f].tsame || a->fields[f].ncomp != 1) return NH_OK;
  LTermsK p;
  p.nelems = a->nelems, p.elist = a->elist_dev, p.nq = a->nq, p.weights = a->weights_dev;
  p.geom = m.geom, p.test = m.test, p.trial = m.trial;
  p.by_elem = m.emap_by_elem;
  p.nterms = m.nterms, p.npolys = m.npolys;
  p.u[0] = a->nfields > 0 ? a->fields[0].u_dev : nullptr;
  p.u[1] = a->nfields > 1 ? a->fields[1].u_dev : nullptr;
  for (int t = 0; t < MAXT; ++t) p.scale[t] = m.scale[t], p.toff[t] = m.toff[t], p.qoff[t] = m.qoff[t];
  for (int k = 0; k < MAXP; ++k) p.poff[k] = m.poff[k];
  p.tlen = (int)tab.size();
  std::copy(tab.begin(), tab.end(), p.tabarg);
  int nmb = 0;
  const int key = (a->ndims * 100 + a->test.nb) * 100 + a->trial.nb;
  switch (key) {
    case 10202: nmb = 1; break;
    case 10303: nmb = 1; break;
    case 20404: nmb = 1; break;
    case 20909: nmb = 3; break;  // (one thread per element, 9 x 9 sums in 256 VGPRs: 0.85 against 0.90 ms on C4 -- not worth the occupancy)
    case 30808: nmb = 2; break;
    default: return NH_OK;
  }
  nh_pattern *pat = const_cast<nh_pattern *>(a->pattern);
  int rc;
  if ((rc = nh_gather_prepare(pat, a->test, a->elist_dev, s)) != NH_OK) return rc;
  double *scratch = nullptr;
  if ((rc = nh_gather_scratch((size_t)pat->emap_len, &scratch)) != NH_OK) return rc;
  p.local = scratch;
  const i64 nthreads = a->nelems * nmb;
  dim3 grid((unsigned)((nthreads + 127) / 128)), block(128);
  const bool same_tables = a->test.T_dev == a->trial.T_dev && a->test.tab_dev == a->trial.tab_dev;
  const size_t ldst = sizeof(double) * (size_t)a->nq * (1 + a->ndims) * (a->test.nb + (same_tables ? 0 : a->trial.nb));  // staged tables of one element class
  if (ldst > 48 * 1024) return NH_OK;  // (tables too large to stage: the batched kernel keeps the block)
  if (getenv("NH_DEBUG_TERMS")) {
    fprintf(stderr, "local_terms: key %d nterms %d npolys %d nfields %d tlen %zu:", key, m.nterms, m.npolys, a->nfields, tab.size());
    for (int t = 0; t < m.nterms; ++t) fprintf(stderr, " [kind %d fld %d pol %d scale %d q %d]", (int)tab[m.toff[t]], (int)tab[m.toff[t] + 1], (int)tab[m.toff[t] + 2], m.scale[t] != nullptr, m.qoff[t] != 0);
    for (int k = 0; k < m.npolys; ++k) fprintf(stderr, " {poly nv %d nt %d}", (int)tab[m.poff[k]], (int)tab[m.poff[k] + 1]);
    fprintf(stderr, "\n");
  }
  // the two-phase arrangement (k_local_terms2) for the blocks whose row-block threads repeat the point work; NUTILS_AMD_LOCAL_TERMS=1 keeps the kernel above
  const bool two_phase = !(getenv("NUTILS_AMD_LOCAL_TERMS") && atoi(getenv("NUTILS_AMD_LOCAL_TERMS")) == 1);
  bool launched2 = false;
  if (two_phase && (key == 20909 || key == 30808)) {
    const int S2 = 1 + a->ndims, tpe = key == 20909 ? 3 : 4;
    p.ncls = a->nelems >= (1 << 16) ? 2 : LT2_CLS;  // (large meshes: occupancy; small ones, short rows: every class of the run staged)
    const size_t lds2_rest = sizeof(double) * ((NH_LT2_DB ? 2 : 1) * tpe * LT2_EPB * S2 * S2 + LT2_EPB * ((a->nfields * a->test.nb) | 1));
    while (p.ncls > 1 && p.ncls * ((ldst + 15) & ~(size_t)15) + lds2_rest > 64 * 1024) --p.ncls;
    const size_t lds2 = p.ncls * ((ldst + 15) & ~(size_t)15) + lds2_rest;
    dim3 grid2((unsigned)((a->nelems + LT2_EPB - 1) / LT2_EPB)), block2(LT2_EPB * tpe);
    const bool iso2 = m.geom.kind == NH_GEOM_ISO && m.geom.ngb == (1 << a->ndims);
#define LT2I(ND, NBT, NBR, NBK, NF, ISO)                                                                                                                  \
  do {                                                                                                                                                   \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_local_terms2<ND, NBT, NBR, NBK, NF, ISO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));   \
    hipLaunchKernelGGL((k_local_terms2<ND, NBT, NBR, NBK, NF, ISO>), grid2, block2, lds2, s, p);                                                          \
  } while (0)
#define LT2(ND, NBT, NBR, NBK, NF)             \
  do {                                         \
    if (iso2) LT2I(ND, NBT, NBR, NBK, NF, true); \
    else LT2I(ND, NBT, NBR, NBK, NF, false);   \
  } while (0)
#define LT2F(ND, NBT, NBR, NBK)                        \
  do {                                                 \
    if (a->nfields == 0) LT2(ND, NBT, NBR, NBK, 0);    \
    else if (a->nfields == 1) LT2(ND, NBT, NBR, NBK, 1); \
    else LT2(ND, NBT, NBR, NBK, 2);                    \
  } while (0)
    if (key == 20909) LT2F(2, 9, 9, 3);
    else LT2F(3, 8, 8, 2);
#undef LT2F
#undef LT2
#undef LT2I
    launched2 = true;
  }
#define LT(ND, NBT, NBR, MB)                                                                                  \
  do {                                                                                                        \
    if (a->nfields == 0) hipLaunchKernelGGL((k_local_terms<ND, NBT, NBR, MB, 0>), grid, block, ldst, s, p);      \
    else if (a->nfields == 1) hipLaunchKernelGGL((k_local_terms<ND, NBT, NBR, MB, 1>), grid, block, ldst, s, p); \
    else hipLaunchKernelGGL((k_local_terms<ND, NBT, NBR, MB, 2>), grid, block, ldst, s, p);                      \
  } while (0)
  if (!launched2) switch (key) {
    case 10202: LT(1, 2, 2, 2); break;
    case 10303: LT(1, 3, 3, 3); break;
    case 20404: LT(2, 4, 4, 4); break;
    case 20909: LT(2, 9, 9, 3); break;
    case 30808: LT(3, 8, 8, 4); break;
  }
#undef LT
  NH_LAUNCH_CHECK();
  GSlots gs;
  memset(&gs, 0, sizeof gs);
  gs.nct = gs.ncr = gs.tot = 1;
  gs.cnt[0] = 1, gs.mask[0][0] = 1;
  if ((rc = nh_gather_values(pat, scratch, 0, gs, a->values_dev, (a->flags & NH_MATRIX_STORE) != 0, s)) != NH_OK) return rc;
  *done = true;
  return NH_OK;
}

// thread-per-element residual for scalar blocks on ONE small uniform test basis with all fields on it; *done = false: not applicable
int local_vterms(const nh_terms_args *a, const TermsK &m, const std::vector<double> &tab, bool *done, hipStream_t s) {
  *done = false;
  const nh_basis &tb = a->blocks[0].test;
  // (one thread per element pays when there are enough elements to hide its serial pass over the points: 2 M trilinear elements 0.82 ms against
  // 1.16 ms of the batched kernel, but 262 144 elements of 25 points 0.78 against 0.45 ms -- there the lanes-over-points batches win)
  if (a->nelems < LV2_MIN_ELEMS || a->nfields > 3 || tab.size() > (size_t)TABARG || tb.off_dev || !tb.nb) return NH_OK;
  auto same = [&](const nh_basis &b) { return b.T_dev == tb.T_dev && b.dofs_dev == tb.dofs_dev && b.tab_dev == tb.tab_dev && b.off_dev == tb.off_dev && b.nb == tb.nb; };
  for (int b = 0; b < a->nblocks; ++b)
    if (a->blocks[b].nct != 1 || !same(a->blocks[b].test)) return NH_OK;
  for (int f = 0; f < a->nfields; ++f)
    if (a->fields[f].ncomp != 1 || !same(a->fields[f].basis)) return NH_OK;
  if (getenv("NH_DEBUG_TERMS")) {
    fprintf(stderr, "local_vterms: nelems %lld nterms %d npolys %d nfields %d nblocks %d tlen %zu:", (long long)a->nelems, m.nterms, m.npolys, a->nfields, a->nblocks, tab.size());
    for (int t = 0; t < m.nterms; ++t) fprintf(stderr, " [blk %d fld %d pol %d C %d f %d scale %d q %d]", (int)tab[m.toff[t]], (int)tab[m.toff[t] + 1], (int)tab[m.toff[t] + 2], (int)tab[m.toff[t] + 3], (int)tab[m.toff[t] + 4], m.scale[t] != nullptr, m.qoff[t] != 0);
    for (int k = 0; k < m.npolys; ++k) fprintf(stderr, " {poly nv %d nt %d}", (int)tab[m.poff[k]], (int)tab[m.poff[k] + 1]);
    fprintf(stderr, "\n");
  }
  VTermsK p;
  p.nelems = a->nelems, p.elist = a->elist_dev, p.nq = a->nq, p.weights = a->weights_dev;
  p.geom = m.geom, p.test = to_k(tb);
  p.nterms = m.nterms, p.npolys = m.npolys, p.nblocks = a->nblocks;
  for (int f = 0; f < 3; ++f) p.u[f] = f < a->nfields ? a->fields[f].u_dev : nullptr;
  for (int b = 0; b < 2; ++b) p.out[b] = b < a->nblocks ? a->blocks[b].out_dev : nullptr;
  for (int b = 0; b < 2; ++b) p.local[b] = b < a->nblocks ? a->blocks[b].local_dev : nullptr;
  for (int t = 0; t < MAXT; ++t) p.scale[t] = m.scale[t], p.toff[t] = m.toff[t], p.qoff[t] = m.qoff[t];
  for (int k = 0; k < MAXP; ++k) p.poff[k] = m.poff[k];
  p.tlen = (int)tab.size();
  std::copy(tab.begin(), tab.end(), p.tabarg);
  // the waves-over-points arrangement (k_local_vterms2); NUTILS_AMD_LOCAL_VTERMS=1 keeps one thread per element (from 2^20 elements on, as before round 4)
  const bool waves_over_points = !(getenv("NUTILS_AMD_LOCAL_VTERMS") && atoi(getenv("NUTILS_AMD_LOCAL_VTERMS")) == 1);
  if (waves_over_points) {
    const int S2 = 1 + a->ndims, nb = tb.nb;
    const size_t tabb = sizeof(double) * (((size_t)nb * a->nq * S2 + 1) & ~(size_t)1);
    const size_t rest = sizeof(double) * std::max<size_t>((size_t)64 * ((a->nfields * nb) | 1), (size_t)(LV2_NW - 1) * 64 * ((2 * nb) | 1));
    p.ncls = a->nelems >= (1 << 16) ? 2 : LT2_CLS;
    while (p.ncls > 1 && p.ncls * tabb + rest > 64 * 1024) --p.ncls;
    const size_t lds2 = p.ncls * tabb + rest;
    if (lds2 <= 96 * 1024) {
      dim3 grid2((unsigned)((a->nelems + 63) / 64)), block2(64 * LV2_NW);
#define LV2I(ND, NB, NF)                                                                                                                  \
  do {                                                                                                                                   \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_local_vterms2<ND, NB, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));  \
    hipLaunchKernelGGL((k_local_vterms2<ND, NB, NF>), grid2, block2, lds2, s, p);                                                         \
  } while (0)
#define LV2(ND, NB)                             \
  do {                                          \
    if (a->nfields == 0) LV2I(ND, NB, 0);       \
    else if (a->nfields == 1) LV2I(ND, NB, 1);  \
    else if (a->nfields == 2) LV2I(ND, NB, 2);  \
    else LV2I(ND, NB, 3);                       \
  } while (0)
      bool ok = true;
      switch (a->ndims * 100 + tb.nb) {
        case 102: LV2(1, 2); break;
        case 103: LV2(1, 3); break;
        case 204: LV2(2, 4); break;
        case 209: LV2(2, 9); break;
        case 308: LV2(3, 8); break;
        default: ok = false;
      }
#undef LV2
#undef LV2I
      if (ok) {
        NH_LAUNCH_CHECK();
        *done = true;
        return NH_OK;
      }
    }
  }
  if (a->nelems < (1 << 20)) return NH_OK;
  dim3 grid((unsigned)((a->nelems + 127) / 128)), block(128);
#define LV(ND, NB)                                                                                     \
  do {                                                                                                 \
    if (a->nfields == 0) hipLaunchKernelGGL((k_local_vterms<ND, NB, 0>), grid, block, 0, s, p);        \
    else if (a->nfields == 1) hipLaunchKernelGGL((k_local_vterms<ND, NB, 1>), grid, block, 0, s, p);   \
    else if (a->nfields == 2) hipLaunchKernelGGL((k_local_vterms<ND, NB, 2>), grid, block, 0, s, p);   \
    else hipLaunchKernelGGL((k_local_vterms<ND, NB, 3>), grid, block, 0, s, p);                        \
  } while (0)
  switch (a->ndims * 100 + tb.nb) {
    case 102: LV(1, 2); break;
    case 103: LV(1, 3); break;
    case 204: LV(2, 4); break;
    case 209: LV(2, 9); break;
    case 308: LV(3, 8); break;
    default: return NH_OK;
  }
#undef LV
  NH_LAUNCH_CHECK();
  *done = true;
  return NH_OK;
}

// argument checks, parameter block and term table of one term list
int build_terms(const nh_terms_args *a, hipStream_t stream, TermsK &p, std::vector<double> &tab, size_t *ldsbytes) {
  NH_REQUIRE(a, "nh_assemble_terms: NULL args");
  NH_REQUIRE(a->ndims >= 1 && a->ndims <= 3, "ndims must be 1..3");
  NH_REQUIRE(a->nq >= 1 && a->weights_dev, "quadrature missing");
  NH_REQUIRE(a->nfields >= 0 && a->nfields <= MAXF && (a->nfields == 0 || a->fields), "nh_assemble_terms: 0..%d fields", MAXF);
  NH_REQUIRE(a->nblocks >= 1 && a->nblocks <= MAXB && a->blocks, "nh_assemble_terms: 1..%d output blocks", MAXB);
  NH_REQUIRE(a->nterms >= 1 && a->nterms <= MAXT && a->terms, "nh_assemble_terms: 1..%d terms", MAXT);
  NH_REQUIRE(a->npolys >= 0 && a->npolys <= MAXP && (a->npolys == 0 || a->polys), "nh_assemble_terms: 0..%d pointwise polynomials", MAXP);
  int rc = check_geom2(a->geom);
  if (rc) return rc;
  *ldsbytes = 0;
  if (a->nelems == 0) return NH_OK;
  const int S = 1 + a->ndims;
  p.nelems = a->nelems;
  p.elist = a->elist_dev;
  p.nq = a->nq;
  p.weights = a->weights_dev;
  p.geom = to_k(a->geom);
  p.geom.nograd = 1;
  for (int t = 0; t < a->nterms; ++t) {
    const nh_term &T = a->terms[t];
    if (T.block < 0 || T.block >= a->nblocks) continue;  // (reported below)
    const int nct = a->blocks[T.block].nct, ncr = T.field >= 0 && T.field < a->nfields ? a->fields[T.field].ncomp : 0;
    if ((ncr && uses_gradients(T.C_host, nct, S, ncr)) || uses_gradients(T.f_host, nct, S, 0) || uses_gradients(T.qs_B_host, 1, S, 1)) p.geom.nograd = 0;
  }
  p.nfields = a->nfields, p.nblocks = a->nblocks, p.nterms = a->nterms, p.npolys = a->npolys;
  p.fct = 0, p.uesz = 0;
  for (int f = 0; f < a->nfields; ++f) {
    const nh_field &F = a->fields[f];
    NH_REQUIRE(F.basis.T_dev && F.basis.dofs_dev && F.u_dev && F.ncomp >= 1 && F.ncomp <= 3, "nh_assemble_terms: field %d incomplete", f);
    NH_REQUIRE(!a->elist_dev || !F.basis.off_dev, "elist with ragged bases is not supported");
    p.fields[f].b = to_k(F.basis);
    p.fields[f].u = F.u_dev;
    p.fields[f].ncomp = F.ncomp;
    p.fields[f].c0 = p.fct;
    p.fct += F.ncomp;
    if ((rc = max_nb2(F.basis, a->nelems, &p.fields[f].maxnb, stream)) != NH_OK) return rc;
    p.fields[f].ue0 = p.uesz;
    p.fields[f].tsame = 0;
    p.uesz += p.fields[f].maxnb * F.ncomp;
  }
  NH_REQUIRE(p.fct <= MAXFC, "nh_assemble_terms: more than %d field components", MAXFC);
  p.ct = 0, p.rowsper = 0;
  for (int b = 0; b < a->nblocks; ++b) {
    const nh_block &B = a->blocks[b];
    NH_REQUIRE(B.test.T_dev && B.test.dofs_dev && (B.out_dev || B.local_dev) && B.nct >= 1 && B.nct <= 3, "nh_assemble_terms: block %d incomplete", b);
    NH_REQUIRE(!a->elist_dev || !B.test.off_dev, "elist with ragged bases is not supported");
    p.blocks[b].test = to_k(B.test);
    p.blocks[b].out = B.out_dev;
    p.blocks[b].local = B.local_dev;
    p.blocks[b].nct = B.nct;
    p.blocks[b].c0 = p.ct;
    if ((rc = max_nb2(B.test, a->nelems, &p.blocks[b].maxnb, stream)) != NH_OK) return rc;
    p.ct += B.nct;
    p.rowsper += p.blocks[b].maxnb * B.nct;
  }
  NH_REQUIRE(p.ct <= MAXCT, "nh_assemble_terms: more than %d test components", MAXCT);
  tab.clear();
  for (int t = 0; t < a->nterms; ++t) {
    const nh_term &T = a->terms[t];
    NH_REQUIRE(T.block >= 0 && T.block < a->nblocks && T.field >= -1 && T.field < a->nfields && T.poly >= -1 && T.poly < a->npolys, "nh_assemble_terms: term %d refers to a missing block / field / polynomial", t);
    NH_REQUIRE(T.C_host || T.f_host, "nh_assemble_terms: term %d has neither a form nor a source", t);
    NH_REQUIRE(!T.C_host || T.field >= 0, "nh_assemble_terms: term %d: coefficient tensor given without field", t);
    const int nct = a->blocks[T.block].nct, ncr = T.field >= 0 ? a->fields[T.field].ncomp : 0;
    p.toff[t] = (int)tab.size();
    p.scale[t] = T.scale_dev;
    tab.push_back(T.block), tab.push_back(T.field), tab.push_back(T.poly), tab.push_back(T.C_host ? 1 : 0), tab.push_back(T.f_host ? 1 : 0);
    for (int i = 0; i < nct * S; ++i) tab.push_back(T.f_host ? T.f_host[i] : 0.);
    if (T.C_host)
      for (int i = 0; i < nct * S * ncr * S; ++i) tab.push_back(T.C_host[i]);
    if ((rc = push_qs(T.qs_field_t, T.qs_field_r, T.qs_B_host, S, a->fields, a->nfields, p.fields, t, tab, &p.qoff[t])) != NH_OK) return rc;
  }
  for (int t = a->nterms; t < MAXT; ++t) p.scale[t] = nullptr, p.toff[t] = 0, p.qoff[t] = 0;
  for (int k = 0; k < MAXP; ++k) p.poff[k] = 0;
  for (int k = 0; k < a->npolys; ++k)
    if ((rc = push_poly(a->polys[k], a->fields, a->nfields, p.fields, k, tab, &p.poff[k])) != NH_OK) return rc;
  p.tlen = (int)tab.size();
  // elements per batch: as many as fill the workgroup in the pointwise phase
  p.eb = std::max(1, NTB / a->nq);
  // one basis for every block and field (uniform size, no ragged offsets): its table of the batch's element class goes to LDS (a structured mesh has one class
  // nearly everywhere; waves with another class read global memory as before)
  p.tstage = a->nblocks >= 1 && a->blocks[0].test.nb > 0 && !a->blocks[0].test.off_dev;
  for (int b = 0; b < a->nblocks && p.tstage; ++b)
    p.tstage = a->blocks[b].test.T_dev == a->blocks[0].test.T_dev && a->blocks[b].test.tab_dev == a->blocks[0].test.tab_dev && a->blocks[b].test.nb == a->blocks[0].test.nb && !a->blocks[b].test.off_dev;
  for (int f = 0; f < a->nfields && p.tstage; ++f)
    p.tstage = a->fields[f].basis.T_dev == a->blocks[0].test.T_dev && a->fields[f].basis.tab_dev == a->blocks[0].test.tab_dev && a->fields[f].basis.nb == a->blocks[0].test.nb && !a->fields[f].basis.off_dev;
  const size_t tsz = p.tstage ? (size_t)a->blocks[0].test.nb * a->nq * S + 2 : 0;
  if (tsz * sizeof(double) > 16 * 1024) p.tstage = 0;
  for (int f = 0; f < a->nfields; ++f) p.fields[f].tsame = p.tstage;
  {
    // ... within an LDS budget: with few points per element (one-point rules, boundary sides) NTB / nq elements of a wide vector-valued basis do not fit
    // (27 nodes x 3 components x 256 elements: 217 kB); half of the LDS keeps two workgroups per CU, one element per batch is the floor
    const size_t fixed = (size_t)p.tlen + (size_t)NTB * p.fct * S + (p.tstage ? tsz : 0), per_elem = (size_t)a->nq * p.ct * S + (size_t)p.uesz;
    const size_t limit = 160 * 1024 / sizeof(double), budget = limit / 2;
    if (per_elem && fixed + (size_t)p.eb * per_elem > limit) p.eb = (int)std::max<size_t>(1, budget > fixed ? (budget - fixed) / per_elem : 1);
  }
  *ldsbytes = sizeof(double) * ((size_t)p.tlen + (size_t)p.eb * a->nq * p.ct * S + (size_t)NTB * p.fct * S + (size_t)p.eb * p.uesz + (p.tstage ? tsz : 0));
  p.table = nullptr;
  return NH_OK;
}

// parameter blocks (+ the term tables that do not fit them) of a multi-list launch: page-locked host and device buffer pairs in a ring, so
// that neither the stream nor the host has to wait for the previous launch.  A launch whose parameter image equals that of a slot (the same
// lists on the same arrays: a Newton loop, where the allocator hands out the same blocks step after step) reuses the device copy: no
// host-to-device transfer in front of the kernel
struct ListSlot {
  char *host = nullptr, *dev = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr, uploaded = nullptr;
  bool used = false;
  std::vector<char> image;  // what the device copy holds (device addresses of the tables not filled in)
};
constexpr int NSLOT = 8;
int list_slot(const std::vector<char> &image, ListSlot **out, bool *hit) {
  static ListSlot ring[NSLOT];
  static int next = 0;
  for (ListSlot &sl : ring)
    if (sl.used && sl.image.size() == image.size() && !std::memcmp(sl.image.data(), image.data(), image.size())) {
      *out = &sl, *hit = true;
      return NH_OK;
    }
  ListSlot &sl = ring[next];
  next = (next + 1) % NSLOT;
  if (!sl.done) {
    NH_CHECK_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    NH_CHECK_HIP(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
  }
  if (sl.used) NH_CHECK_HIP(hipEventSynchronize(sl.done));
  sl.used = false;
  const size_t bytes = image.size();
  if (bytes > sl.cap) {
    if (sl.host) NH_CHECK_HIP(hipHostFree(sl.host));
    if (sl.dev) NH_CHECK_HIP(hipFree(sl.dev));
    sl.host = sl.dev = nullptr, sl.cap = 0;
    NH_CHECK_HIP(hipHostMalloc((void **)&sl.host, 2 * bytes, hipHostMallocDefault));
    NH_CHECK_HIP(hipMalloc((void **)&sl.dev, 2 * bytes));
    sl.cap = 2 * bytes;
  }
  sl.image = image;
  *out = &sl, *hit = false;
  return NH_OK;
}

}  // namespace

extern "C" int nh_assemble_terms(const nh_terms_args *a, void *stream) {
  TermsK p;
  std::vector<double> tab;
  size_t lds;
  hipStream_t s = nh_stream(stream);
  int rc = build_terms(a, s, p, tab, &lds);
  if (rc) return rc;
  if (a->nelems == 0) return NH_OK;
  {
    bool done = false;
    if ((rc = local_vterms(a, p, tab, &done, s)) != NH_OK) return rc;
    if (done) return NH_OK;
  }
  NH_REQUIRE(lds <= 160 * 1024, "nh_assemble_terms: batch too large for LDS (%zu bytes)", lds);
  if ((rc = place_table(tab, p.tabarg, &p.table, s)) != NH_OK) return rc;
  const i64 nbatch = (a->nelems + p.eb - 1) / p.eb;
  dim3 grid((unsigned)std::min<i64>(nbatch, 256 * 8)), block(NTB);
#define LAUNCH(ND)                                                                                                          \
  do {                                                                                                                      \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_terms<ND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
    hipLaunchKernelGGL(k_terms<ND>, grid, block, lds, s, p);                                                                \
  } while (0)
  if (a->ndims == 1) LAUNCH(1);
  if (a->ndims == 2) LAUNCH(2);
  if (a->ndims == 3) LAUNCH(3);
#undef LAUNCH
  NH_LAUNCH_CHECK();
  return NH_OK;
}

extern "C" int nh_assemble_terms_multi(int count, const nh_terms_args *const *lists, void *stream) {
  NH_REQUIRE(count >= 0 && (count == 0 || lists), "nh_assemble_terms_multi: invalid argument");
  hipStream_t s = nh_stream(stream);
  std::vector<TermsK> ps;
  std::vector<std::vector<double>> tabs;
  std::vector<i64> nbatch;
  size_t lds = 0;
  int ndims = 0, rc, last = -1;
  for (int i = 0; i < count; ++i) {
    const nh_terms_args *a = lists[i];
    // lists that get their own kind of kernel (thread per element on large meshes), other dimensions or more lists than a launch takes: on their own
    if (a && (a->nelems >= LV2_MIN_ELEMS || (ndims && a->ndims != ndims) || (int)ps.size() == MAXL)) {
      if ((rc = nh_assemble_terms(a, stream)) != NH_OK) return rc;
      continue;
    }
    TermsK p;
    std::memset((void *)&p, 0, sizeof p);  // (padding and unused entries: the parameter images are compared byte by byte)
    std::vector<double> tab;
    size_t l;
    if ((rc = build_terms(a, s, p, tab, &l)) != NH_OK) return rc;
    if (a->nelems == 0) continue;
    NH_REQUIRE(l <= 160 * 1024, "nh_assemble_terms: batch too large for LDS (%zu bytes)", l);
    ndims = a->ndims, last = i;
    lds = std::max(lds, l);
    nbatch.push_back((a->nelems + p.eb - 1) / p.eb);
    ps.push_back(p);
    tabs.push_back(std::move(tab));
  }
  if (ps.empty()) return NH_OK;
  if (ps.size() == 1) return nh_assemble_terms(lists[last], stream);  // nothing to merge: parameters in the kernel arguments
  // workgroups per list: in proportion to the batches, at least one, 256 * 8 in total when there is that much work
  i64 total = 0;
  for (i64 n : nbatch) total += n;
  MultiK m;
  m.count = (int)ps.size();
  m.first[0] = 0;
  for (int i = 0; i < m.count; ++i) {
    const i64 share = total <= 256 * 8 ? nbatch[i] : std::max<i64>(1, nbatch[i] * (256 * 8) / total);
    m.first[i + 1] = m.first[i] + (unsigned)share;
  }
  for (int i = m.count; i < MAXL; ++i) m.first[i + 1] = m.first[m.count];
  size_t bytes = ps.size() * sizeof(TermsK);
  std::vector<size_t> toff(ps.size(), 0);
  for (size_t i = 0; i < ps.size(); ++i)
    if (tabs[i].size() > (size_t)TABARG) toff[i] = bytes, bytes += tabs[i].size() * sizeof(double);
  std::vector<char> image(bytes, 0);
  for (size_t i = 0; i < ps.size(); ++i) {
    if (toff[i])
      std::memcpy(image.data() + toff[i], tabs[i].data(), tabs[i].size() * sizeof(double));
    else
      std::copy(tabs[i].begin(), tabs[i].end(), ps[i].tabarg);
    ps[i].table = nullptr;
    std::memcpy(image.data() + i * sizeof(TermsK), &ps[i], sizeof(TermsK));
  }
  ListSlot *sl;
  bool hit;
  if ((rc = list_slot(image, &sl, &hit)) != NH_OK) return rc;
  if (hit) {
    NH_CHECK_HIP(hipStreamWaitEvent(s, sl->uploaded, 0));  // (the copy may still be queued on the stream of the launch that made it)
  } else {
    std::memcpy(sl->host, image.data(), bytes);
    for (size_t i = 0; i < ps.size(); ++i)
      if (toff[i]) {
        const double *table = (const double *)(sl->dev + toff[i]);
        std::memcpy(sl->host + i * sizeof(TermsK) + offsetof(TermsK, table), &table, sizeof table);
      }
    NH_CHECK_HIP(hipMemcpyAsync(sl->dev, sl->host, bytes, hipMemcpyHostToDevice, s));
    NH_CHECK_HIP(hipEventRecord(sl->uploaded, s));
  }
  dim3 grid(m.first[m.count]), block(NTB);
#define LAUNCH(ND)                                                                                                                \
  do {                                                                                                                            \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_terms_multi<ND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
    hipLaunchKernelGGL(k_terms_multi<ND>, grid, block, lds, s, (const TermsK *)sl->dev, m);                                       \
  } while (0)
  if (ndims == 1) LAUNCH(1);
  if (ndims == 2) LAUNCH(2);
  if (ndims == 3) LAUNCH(3);
#undef LAUNCH
  NH_LAUNCH_CHECK();
  NH_CHECK_HIP(hipEventRecord(sl->done, s));
  sl->used = true;
  return NH_OK;
}

extern "C" int nh_assemble_matrix_terms(const nh_matrix_terms_args *a, void *stream) {
  NH_REQUIRE(a, "nh_assemble_matrix_terms: NULL args");
  NH_REQUIRE(a->ndims >= 1 && a->ndims <= 3, "ndims must be 1..3");
  NH_REQUIRE(a->nq >= 1 && a->weights_dev, "quadrature missing");
  NH_REQUIRE((a->flags & ~(NH_MATRIX_EMAP_BY_ELEMENT | NH_MATRIX_GATHER | NH_MATRIX_STORE)) == 0, "nh_assemble_matrix_terms: unknown flag bits 0x%x", a->flags & ~(NH_MATRIX_EMAP_BY_ELEMENT | NH_MATRIX_GATHER | NH_MATRIX_STORE));
  NH_REQUIRE(!(a->flags & NH_MATRIX_STORE) || (a->flags & NH_MATRIX_GATHER), "NH_MATRIX_STORE is an option of NH_MATRIX_GATHER");
  NH_REQUIRE(a->srowptr_dev && a->emap_dev && a->values_dev, "nh_assemble_matrix_terms: NULL pattern / values");
  NH_REQUIRE(a->test.T_dev && a->test.dofs_dev && a->trial.T_dev && a->trial.dofs_dev, "basis tables missing");
  NH_REQUIRE(a->nct >= 1 && a->nct <= 3 && a->ncr >= 1 && a->ncr <= 3, "component counts must be 1..3 (got %d, %d)", a->nct, a->ncr);
  NH_REQUIRE(a->nfields >= 0 && a->nfields <= MAXF && (a->nfields == 0 || a->fields), "nh_assemble_matrix_terms: 0..%d fields", MAXF);
  NH_REQUIRE(a->nterms >= 1 && a->nterms <= MAXT && a->terms, "nh_assemble_matrix_terms: 1..%d terms", MAXT);
  NH_REQUIRE(a->npolys >= 0 && a->npolys <= MAXP && (a->npolys == 0 || a->polys), "nh_assemble_matrix_terms: 0..%d pointwise polynomials", MAXP);
  NH_REQUIRE(!a->elist_dev || (a->test.nb && a->trial.nb), "elist with ragged bases is not supported");
  int rc = check_geom2(a->geom);
  if (rc) return rc;
  if (a->nelems == 0) return NH_OK;
  const int S = 1 + a->ndims, CS = a->nct * S * a->ncr * S;
  MTermsK p;
  p.debug = 0;
  p.tdbg = nullptr;
#ifdef NH_ABLATION
  if (getenv("NH_BATCHED_DEBUG")) p.debug = atoi(getenv("NH_BATCHED_DEBUG"));
  static long long *tdbg = nullptr;
  if (!tdbg) NH_CHECK_HIP(hipMalloc((void **)&tdbg, 8 * sizeof(long long)));
  NH_CHECK_HIP(hipMemsetAsync(tdbg, 0, 8 * sizeof(long long), nh_stream(stream)));
  if (getenv("NH_BATCHED_TIMERS")) p.tdbg = tdbg;
#endif
  p.nelems = a->nelems;
  p.elist = a->elist_dev;
  p.nq = a->nq;
  p.weights = a->weights_dev;
  p.geom = to_k(a->geom);
  p.geom.nograd = 1;
  for (int t = 0; t < a->nterms; ++t) {
    const nh_matrix_term &T = a->terms[t];
    if (T.kind != 0 || uses_gradients(T.C_host, a->nct, 1 + a->ndims, a->ncr) || uses_gradients(T.qs_B_host, 1, 1 + a->ndims, 1)) p.geom.nograd = 0;
  }
  p.nfields = a->nfields, p.nterms = a->nterms, p.npolys = a->npolys;
  p.fct = 0, p.uesz = 0;
  for (int f = 0; f < a->nfields; ++f) {
    const nh_field &F = a->fields[f];
    NH_REQUIRE(F.basis.T_dev && F.basis.dofs_dev && F.u_dev && F.ncomp >= 1 && F.ncomp <= 3, "nh_assemble_matrix_terms: field %d incomplete", f);
    NH_REQUIRE(!a->elist_dev || !F.basis.off_dev, "elist with ragged bases is not supported");
    p.fields[f].b = to_k(F.basis);
    p.fields[f].u = F.u_dev;
    p.fields[f].ncomp = F.ncomp;
    p.fields[f].c0 = p.fct;
    p.fct += F.ncomp;
    if ((rc = max_nb2(F.basis, a->nelems, &p.fields[f].maxnb, nh_stream(stream))) != NH_OK) return rc;
    p.fields[f].ue0 = p.uesz;
    p.fields[f].tsame = (F.basis.T_dev == a->test.T_dev && F.basis.off_dev == a->test.off_dev && F.basis.tab_dev == a->test.tab_dev && F.basis.nb == a->test.nb);
    p.uesz += p.fields[f].maxnb * F.ncomp;
  }
  NH_REQUIRE(p.fct <= MAXFC, "nh_assemble_matrix_terms: more than %d field components", MAXFC);
  p.test = to_k(a->test), p.trial = to_k(a->trial);
  p.nct = a->nct, p.ncr = a->ncr;
  if ((rc = max_nb2(a->test, a->nelems, &p.maxnbt, nh_stream(stream))) != NH_OK) return rc;
  if ((rc = max_nb2(a->trial, a->nelems, &p.maxnbr, nh_stream(stream))) != NH_OK) return rc;
  p.srowptr = (const i64 *)a->srowptr_dev;
  p.emap = a->emap_dev;
  p.eoff = (const i64 *)a->eoff_dev;
  p.values = a->values_dev;
  p.emap_by_elem = (a->flags & NH_MATRIX_EMAP_BY_ELEMENT) != 0;
  p.tot = 0;
  for (int c = 0; c < 3; ++c) {
    p.cum[c] = p.tot, p.cnt[c] = 0;
    for (int d = 0; d < 3; ++d) {
      p.mask[c][d] = c < a->nct && d < a->ncr && (a->mask_host ? a->mask_host[c * a->ncr + d] != 0 : 1);
      p.dpos[c][d] = (signed char)p.cnt[c];
      p.cnt[c] += p.mask[c][d];
    }
    p.tot += p.cnt[c];
  }
  std::vector<double> tab;
  for (int t = 0; t < a->nterms; ++t) {
    const nh_matrix_term &T = a->terms[t];
    NH_REQUIRE(T.kind >= 0 && T.kind <= 2 && T.C_host, "nh_assemble_matrix_terms: term %d: kind 0..2 with a coefficient tensor", t);
    NH_REQUIRE(T.poly >= -1 && T.poly < a->npolys, "nh_assemble_matrix_terms: term %d refers to a missing polynomial", t);
    if (T.kind) {
      NH_REQUIRE(a->nct == 1 && a->ncr == 1 && T.field >= 0 && T.field < a->nfields && a->fields[T.field].ncomp == 1, "nh_assemble_matrix_terms: term %d: point-dependent forms are for scalar fields", t);
      NH_REQUIRE(T.kind != 2 || T.L_host, "nh_assemble_matrix_terms: term %d: kind 2 needs L", t);
    }
    p.toff[t] = (int)tab.size();
    p.scale[t] = T.scale_dev;
    tab.push_back(T.kind), tab.push_back(T.kind ? T.field : -1), tab.push_back(T.poly);
    for (int i = 0; i < CS; ++i) tab.push_back(T.C_host[i]);
    if (T.kind == 2)
      for (int i = 0; i < a->nct * S; ++i) tab.push_back(T.L_host[i]);
    if ((rc = push_qs(T.qs_field_t, T.qs_field_r, T.qs_B_host, S, a->fields, a->nfields, p.fields, t, tab, &p.qoff[t])) != NH_OK) return rc;
  }
  for (int t = a->nterms; t < MAXT; ++t) p.scale[t] = nullptr, p.toff[t] = 0, p.qoff[t] = 0;
  for (int k = 0; k < MAXP; ++k) p.poff[k] = 0;
  for (int k = 0; k < a->npolys; ++k)
    if ((rc = push_poly(a->polys[k], a->fields, a->nfields, p.fields, k, tab, &p.poff[k])) != NH_OK) return rc;
  p.tlen = (int)tab.size();
  if (a->flags & NH_MATRIX_GATHER) {
    bool done = false;
    if ((rc = local_terms(a, p, tab, &done, nh_stream(stream))) != NH_OK) return rc;
    if (done) return NH_OK;
    NH_REQUIRE(!(a->flags & NH_MATRIX_STORE), "nh_assemble_matrix_terms: NH_MATRIX_STORE, but this block does not qualify for the owner-side reduction");
  }
  p.same = (a->test.T_dev == a->trial.T_dev && a->test.off_dev == a->trial.off_dev && a->test.tab_dev == a->trial.tab_dev && a->test.nb == a->trial.nb);
  // elements per batch: fill the workgroup in the pointwise phase, within an LDS budget that keeps two workgroups per CU
  const size_t per_elem = sizeof(double) * (a->nq * ((size_t)CS + (size_t)(p.same ? p.maxnbt : p.maxnbt + p.maxnbr) * S) + p.uesz);
  p.eb = std::max(1, std::min(NTB / a->nq, (int)(60 * 1024 / per_elem)));
  p.eb = std::min(p.eb, 32);  // (slot bookkeeping in a 32-bit mask)
  const size_t lds = sizeof(double) * ((size_t)p.tlen + (size_t)NTB * p.fct * S + 10 * (size_t)p.eb) + p.eb * per_elem;
  NH_REQUIRE(lds <= 160 * 1024, "nh_assemble_matrix_terms: batch too large for LDS (%zu bytes)", lds);
  hipStream_t s = nh_stream(stream);
  if ((rc = place_table(tab, p.tabarg, &p.table, s)) != NH_OK) return rc;
  const i64 nbatch = (a->nelems + p.eb - 1) / p.eb;
  dim3 grid((unsigned)std::min<i64>(nbatch, 256 * 8)), block(NTB);
#define LAUNCH(ND)                                                                                                          \
  do {                                                                                                                      \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_mterms<ND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
    hipLaunchKernelGGL(k_mterms<ND>, grid, block, lds, s, p);                                                               \
  } while (0)
  if (a->ndims == 1) LAUNCH(1);
  if (a->ndims == 2) LAUNCH(2);
  if (a->ndims == 3) LAUNCH(3);
#undef LAUNCH
  NH_LAUNCH_CHECK();
#ifdef NH_ABLATION
  if (p.tdbg) {
    long long h[8];
    NH_CHECK_HIP(hipMemcpy(h, p.tdbg, sizeof h, hipMemcpyDeviceToHost));
    const double g = grid.x;
    fprintf(stderr, "k_mterms cycles per workgroup: meta %.0f | slots+coeffs+geometry %.0f | stage %.0f | pointwise %.0f | contraction %.0f (eb %d, %u workgroups)\n", h[0] / g, h[1] / g, h[2] / g, h[3] / g, h[4] / g, p.eb, grid.x);
  }
#endif
  return NH_OK;
}
