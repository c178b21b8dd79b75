// Write-once assembly for the structured C0 quadratic ('std', degree 2) basis on hexahedra, scalar or vector valued,
// constant-coefficient bilinear forms (BASELINE.json configs[2]: 3-D linear elasticity, P2 vector basis).
//
// Replaces, for this basis, the generated element loop + einsum (evaluable.py:6773-6786, 1885-1886, 6414-6505) and the
// sparse dedup/accumulate of its result (evaluable.py:588-616, 5560-5682; numeric.py:434-460): same CSR layout as
// nh_pattern_expand (rows/cols lexicographic, structural zeros kept, flat dof = node * ncomp + comp), each value written ONCE.
//
// Formulation.  A[(m,c),(n,d)] = sum_{a,b} C[c,a,d,b] G_mn[a,b] with the generalised Gram matrices
//     G_mn[a,b] = sum_q w_q |J_q| D_m[q,a] D_n[q,b]            (D[.,.,0] = value, D[.,.,1+i] = d/dx_i)
// so the quadrature sum is ONE rank-nq update per element, independent of the form and of the number of components:
// (4 nodes x NS slots) x (16 nodes) tiles on v_mfma_f64_16x16x4_f64 with K = quadrature points; the 9 (or 16) slot pairs of a node
// pair end up in the SAME lane (row = 4 a + node, register index = a; one tile per trial slot b), where the constant tensor C
// is applied on the VALU.  For elasticity this is 3.0x fewer MFMAs than contracting (m) x (n,d) per test component.
//
// Ownership (no global atomics, no zero-fill, no colours).  Owner cell (io,jo,ko) owns the 8 nodes (2io+ai, 2jo+aj, 2ko+ak),
// a in {0,1}; a persistent workgroup takes whole K-lines of owner cells and marches along K.  In step k it visits the (up to) four
// elements (io-di, jo-dj, k) and computes of each only the rows of nodes the line owns, in units of 4 nodes; contributions are
// reduced in LDS (ds_add_f64) into row buffers laid out exactly like the CSR rows -- one buffer per node plane K, a ring of 3 even +
// 2 odd planes -- and a finished plane (all of its rows complete, each node's rows contiguous in the value array) is streamed to HBM
// with 16-byte stores.  Every element is visited by 4 lines (halo recomputation of the cheap part: geometry + D table); the MFMA
// work is 8 units of 4 rows per element instead of 6.75.
//
// Three kernels.  k_p2hex_inreg (default, round 3): 4 MFMA waves form their operands in registers from one record per quadrature point (row broadcast
// by v_fmac_f64_dpp row_newbcast), 4 service waves evaluate the geometry one slice ahead and stream the planes finished in the previous slice -- one
// barrier per SLICE.  k_p2hex_pipe (rounds 1-2; kept for forms with a scale array and the one instantiation the in-register variant would spill):
// the service waves also build a D table per element in LDS, one element ahead -- one barrier per element.  k_p2hex (all waves in lockstep through the
// same phases, less LDS) takes the configurations whose tables do not fit twice (four operator slots).
//
// The closed-form pattern of this basis: along an axis with n elements node X couples to [X-2, X+2] (X even, clipped) or
// [X-1, X+1] (X odd); rows are tensor products of these ranges, so row pointers and column positions are arithmetic.
#include "nh_common.h"
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {
#include "nh_geom.inc"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int NB = 27;   // local nodes, order first axis slowest
constexpr int NT = 512;  // lockstep kernel
constexpr int NQMAX = 28;  // quadrature points per element (7 k-steps): beyond that the tables exceed the LDS anyway

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// per-axis pattern of the quadratic C0 basis on n elements, node X in [0, 2n]
__host__ __device__ __forceinline__ int ax_lo(int X) { return (X & 1) ? X - 1 : (X >= 2 ? X - 2 : 0); }
__host__ __device__ __forceinline__ int ax_hi(int X, int n) { return (X & 1) ? X + 1 : (X + 2 <= 2 * n ? X + 2 : 2 * n); }
__host__ __device__ __forceinline__ int ax_cnt(int X, int n) { return ax_hi(X, n) - ax_lo(X) + 1; }
// sum of ax_cnt over nodes < X  (X in [0, 2n+1])
__host__ __device__ __forceinline__ int ax_cum(int X, int n) {
  if (X <= 0) return 0;
  int s = 5 * ((X + 1) >> 1) + 3 * (X >> 1) - 2;
  if (X > 2 * n) s -= 2;
  return s;
}

struct P2K {
  int n0, n1, n2;
  int nq, ks;        // quadrature points, k-steps = ceil(nq / 4)
  int slot[4];       // active operator slots (ascending), NS of them
  int io0, io1;      // owner lines io in [io0, io1] along axis 0
  int l0, l1;        // element layers [l0, l1) along axis 0 contribute
  int esz, osz, dsz; // doubles per even / odd plane buffer, per D table
  int prio;          // pipelined kernel: raise the priority of the table / geometry waves
  int ndb;           // lockstep kernel: D tables in LDS (2: the table of the next element is built while the tasks of this one run)
  const double *weights;
  GeomK geom;
  const double *T;   // [27][nq][4]
  double *values;
  const double *scale;
  double lam, mu, mu2;
  double C[144];     // dense: [c][a][d][b] over the active slots
#ifdef NH_ABLATION
  long long *tdbg;   // phase timers (cycles summed over waves): see nh_p2hex_matrix
  int debug;         // 1: no global stores, 2: no LDS atomics, 4: no MFMA loop, 8: no D build, 16: no geometry
#endif
};

#if defined(NH_ABLATION) && !defined(NH_NOTICKS)
#define TICK(i) do { const long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define TICK(i) do {} while (0)
#endif
#ifdef NH_ABLATION
#define DBG(p, bit) ((p).debug & (bit))
#else
#define DBG(p, bit) 0
#endif

struct Line {  // constants of an owner line (io, jo)
  int io, jo, vmask;  // vmask bit v: element (io - (v >> 1), jo - (v & 1)) exists and contributes
  int loI[2], cntI[2], cumI[2];
  int loJ[2], cntJ[2], cumJ[2];
  int SJ, SK;
};

struct Lds {
  double *E, *O, *Dt, *Jv;
  int *meta;   // [8 planes][4 nodes][2]: offset of the node's rows in the plane buffer, scalar row length (0: node absent)
  i64 *gmeta;  // [8][4]: offset of the node's first row in the value array
  int esz, osz;
  __device__ __forceinline__ double *pbuf(int K) const { return (K & 1) ? O + ((K >> 1) & 1) * osz : E + ((K >> 1) % 3) * esz; }
};

__device__ __forceinline__ Line make_line(const P2K &p, int line) {
  Line L;
  L.io = p.io0 + line / (p.n1 + 1);
  L.jo = line % (p.n1 + 1);
  L.SJ = 8 * p.n1 + 1;
  L.SK = 8 * p.n2 + 1;
  L.vmask = 0;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int ei = L.io - (v >> 1), ej = L.jo - (v & 1);
    if (ei >= p.l0 && ei < p.l1 && ej >= 0 && ej < p.n1) L.vmask |= 1 << v;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int I = 2 * L.io + a, J = 2 * L.jo + a;
    L.loI[a] = ax_lo(I); L.cntI[a] = ax_cnt(I, p.n0); L.cumI[a] = ax_cum(I, p.n0);
    L.loJ[a] = ax_lo(J); L.cntJ[a] = ax_cnt(J, p.n1); L.cumJ[a] = ax_cum(J, p.n1);
  }
  return L;
}

// one thread: where the rows of the 4 nodes (ai, aj) of node plane K live -- in the plane buffer and in the value array
template <int NC>
__device__ __forceinline__ void plane_meta(const P2K &p, const Line &L, const Lds &S, int K) {
  const int cK = ax_cnt(K, p.n2), cumK = ax_cum(K, p.n2);
  int cur = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ai = j >> 1, aj = j & 1;
    bool ex = false;  // the node has contributions iff an element that contains it is visited
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (((L.vmask >> v) & 1) && (!(v >> 1) || ai == 0) && (!(v & 1) || aj == 0)) ex = true;
    const int len = ex ? L.cntI[ai] * L.cntJ[aj] * cK : 0;
    const i64 row0 = (i64)L.cumI[ai] * L.SJ * L.SK + (i64)L.cntI[ai] * ((i64)L.cumJ[aj] * L.SK + (i64)L.cntJ[aj] * cumK);
    const i64 goff = row0 * (NC * NC);
    cur = ((cur + 1) & ~1) + (int)(goff & 1);  // blocks never share a 16-byte pair: the flush copies and zeroes whole pairs
    S.meta[((K & 7) * 4 + j) * 2] = cur;
    S.meta[((K & 7) * 4 + j) * 2 + 1] = len;
    S.gmeta[(K & 7) * 4 + j] = goff;
    cur += len * NC * NC;
  }
}

// 16-byte pairs the rows of node j of plane K occupy in the value array
template <int NC>
__device__ __forceinline__ int node_pairs(const Lds &S, int K, int j) {
  const int len = S.meta[((K & 7) * 4 + j) * 2 + 1];
  if (!len) return 0;
  const int par = (int)(S.gmeta[(K & 7) * 4 + j] & 1);
  return (par + len * NC * NC + 1) >> 1;
}
// stream pairs [pair0, pair1) of the finished rows of node j of plane K to the value array and zero them in the buffer (thread t of nt)
template <int NC>
__device__ __forceinline__ void flush_node(const P2K &p, const Lds &S, int K, int j, int pair0, int pair1, int t, int nt) {
  const int off = S.meta[((K & 7) * 4 + j) * 2], len = S.meta[((K & 7) * 4 + j) * 2 + 1];
  if (!len) return;
  const i64 goff = S.gmeta[(K & 7) * 4 + j];
  const int par = (int)(goff & 1), nd = len * NC * NC;
  double *lb = S.pbuf(K) + off - par;
  double *gb = p.values + (goff - par);
  for (int pi = pair0 + t; pi < pair1; pi += nt) {
    const v2d v = *reinterpret_cast<const v2d *>(lb + 2 * pi);
    *reinterpret_cast<v2d *>(lb + 2 * pi) = v2d{0., 0.};
    const int e0 = 2 * pi;
    const bool v0 = e0 >= par, v1 = e0 + 1 < par + nd;
    if (DBG(p, 1)) continue;
    if (v0 && v1) *reinterpret_cast<v2d *>(gb + e0) = v;
    else if (v0) gb[e0] = v[0];
    else if (v1) gb[e0 + 1] = v[1];
  }
}
template <int NC>
__device__ __forceinline__ void flush_plane(const P2K &p, const Lds &S, int K, int t, int nt) {
#pragma unroll 1
  for (int j = 0; j < 4; ++j) flush_node<NC>(p, S, K, j, 0, node_pairs<NC>(S, K, j), t, nt);
}

// geometry of point q of element (visit v, slice k): Jinv (row major [j][i]) and w |det J| scale
__device__ __forceinline__ void geometry_point(const P2K &p, const Line &L, int v, int k, int q, double *o) {
  const i64 e = ((i64)(L.io - (v >> 1)) * p.n1 + (L.jo - (v & 1))) * p.n2 + k;
  double Ji[3][3], det;
  geometry_at<3>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) o[j * 3 + i] = Ji[j][i];
  o[9] = p.weights[q] * fabs(det) * (p.scale ? p.scale[e * p.nq + q] : 1.);
}

// D table entry (q, n): physical derivatives in MFMA operand order [slot][kstep][ntile][lk][li], q = 4 kstep + lk, n = 16 ntile + li
template <int NS>
__device__ __forceinline__ void d_entry(const P2K &p, const double (&T4)[4], const double *Ji, double *D, int q, int n) {
  double d[4];
  d[0] = T4[0];
#pragma unroll
  for (int i = 0; i < 3; ++i) d[1 + i] = T4[1] * Ji[i] + T4[2] * Ji[3 + i] + T4[3] * Ji[6 + i];
  double *o = D + (((q >> 2) * 2 + (n >> 4)) * 64 + (q & 3) * 16 + (n & 15));
#pragma unroll
  for (int a = 0; a < NS; ++a) {
    const int sl = p.slot[a];
    o[a * p.ks * 128] = sl == 0 ? d[0] : sl == 1 ? d[1] : sl == 2 ? d[2] : d[3];
  }
}

// MFMA tasks of one element (visit v of slice k): (unit of 4 row nodes, tile of 16 column nodes, half of the k-steps), dealt to
// the nw waves; results go to the plane buffers with ds_add_f64
template <int NC, int NS, int MODE>
__device__ __forceinline__ void mfma_tasks(const P2K &p, const Line &L, const Lds &S, const double *D, const double *Jvs, int v, int k, int wave,
                                           int nw, int lane) {
  const int lk = lane >> 4, li = lane & 15, KS = p.ks;
  const int di = v >> 1, dj = v & 1;
  const int nunits = v == 0 ? 3 : v == 3 ? 1 : 2;
  const int ei = L.io - di, ej = L.jo - dj;
#pragma unroll 1
  for (int t = wave; t < nunits * 4; t += nw) {
    const int u = t >> 2, nt = (t >> 1) & 1, h = t & 1;
    // local node (ai, aj, ak) of row slot mu of this unit, or invalid
    auto unit_node = [&](int mu, int &ai, int &aj, int &ak) -> bool {
      if (v == 0) { ai = mu >> 1; aj = mu & 1; ak = u; return true; }
      if (v == 1) { aj = 2; if (u == 0) { ai = mu >> 1; ak = mu & 1; return true; } ai = mu & 1; ak = 2; return mu < 2; }
      if (v == 2) { ai = 2; if (u == 0) { aj = mu >> 1; ak = mu & 1; return true; } aj = mu & 1; ak = 2; return mu < 2; }
      ai = 2; aj = 2; ak = mu < 3 ? mu : 0; return mu < 3;
    };
    int Aoff;
    {
      int ai, aj, ak;
      const int aslot = li >> 2;
      const bool ok = unit_node(li & 3, ai, aj, ak) && aslot < NS;
      const int node = ok ? (ai * 3 + aj) * 3 + ak : 31;  // node 31: a pad column of the table, always zero
      Aoff = (ok ? aslot : 0) * KS * 128 + (node >> 4) * 64 + lk * 16 + (node & 15);
    }
    const int Boff = nt * 64 + lane;
    const int kh = (KS + 1) >> 1;
    const int ks0 = h ? kh : 0, ks1 = h ? KS : kh;
    v4d acc[NS];
#pragma unroll
    for (int b = 0; b < NS; ++b) acc[b] = v4d{0., 0., 0., 0.};
#pragma unroll 1
    for (int ks = ks0; ks < (DBG(p, 4) ? ks0 : ks1); ++ks) {
      const double a = D[Aoff + ks * 128];
      const double wq = Jvs[(v * p.nq + min(4 * ks + lk, p.nq - 1)) * 10 + 9];  // (pad points: the table entries are zero)
      double bv[NS];
#pragma unroll
      for (int b = 0; b < NS; ++b) bv[b] = wq * D[Boff + (b * KS + ks) * 128];
#pragma unroll
      for (int b = 0; b < NS; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[b], acc[b], 0, 0, 0);
    }
    // this lane: row node mu = lk, column node n = 16 nt + li, G[a][b] = acc[b][a]
    int ai, aj, ak;
    const bool okr = unit_node(lk, ai, aj, ak);
    const int n = nt * 16 + li;
    if (okr && n < NB && !DBG(p, 2)) {
      double Kcd[NC][NC];
      if constexpr (MODE == 1) {  // C[c,a,d,b] = lam d_ca d_db + mu d_cd d_ab + mu2 d_cb d_ad over the three gradient slots
        const double tr = p.mu * (acc[0][0] + acc[1][1] + acc[2][2]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int d = 0; d < NC; ++d) Kcd[c][d] = p.lam * acc[d][c] + p.mu2 * acc[c][d] + (c == d ? tr : 0.);
      } else {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int d = 0; d < NC; ++d) {
            double s = 0;
#pragma unroll
            for (int a = 0; a < NS; ++a)
#pragma unroll
              for (int b = 0; b < NS; ++b) s += p.C[((c * NS + a) * NC + d) * NS + b] * acc[b][a];
            Kcd[c][d] = s;
          }
      }
      const int alI = di ? 0 : ai, alJ = dj ? 0 : aj;
      const int K = 2 * k + ak;
      const int j = alI * 2 + alJ;
      const int off = S.meta[((K & 7) * 4 + j) * 2], len = S.meta[((K & 7) * 4 + j) * 2 + 1];
      const int cK = ax_cnt(K, p.n2), lK = ax_lo(K);
      const int ni = n / 9, nj = (n / 3) % 3, nk = n % 3;
      const int loI = alI ? L.loI[1] : L.loI[0], loJ = alJ ? L.loJ[1] : L.loJ[0], cJ = alJ ? L.cntJ[1] : L.cntJ[0];
      const int pos = ((2 * ei + ni - loI) * cJ + (2 * ej + nj - loJ)) * cK + (2 * k + nk - lK);
      double *row = S.pbuf(K) + off + pos * NC;
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int d = 0; d < NC; ++d) atomicAdd(row + c * len * NC + d, Kcd[c][d]);
    }
  }
}

__device__ __forceinline__ int nth_visit(int vmask, int i) {  // index of the i-th set bit
  int v = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if ((vmask >> b) & 1) {
      if (i == 0) v = b;
      --i;
    }
  }
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// lockstep kernel: every wave walks through geometry -> D table -> MFMA tasks -> flush
template <int NC, int NS, int MODE>
__global__ __launch_bounds__(NT) void k_p2hex(P2K p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Lds S;
  S.esz = p.esz; S.osz = p.osz;
  S.E = lds;                      // 3 even-plane buffers
  S.O = S.E + 3 * p.esz;          // 2 odd-plane buffers
  S.Dt = S.O + 2 * p.osz;         // ndb D tables
  S.Jv = S.Dt + p.ndb * p.dsz;    // [4 visits][nq][10]
  S.meta = reinterpret_cast<int *>(S.Jv + 4 * p.nq * 10);
  S.gmeta = reinterpret_cast<i64 *>(S.meta + 64);
  // zero everything once: row buffers are re-zeroed by the flush, the pad entries of the D tables (q >= nq, n >= 27) stay zero
  for (int i = tid; i < 3 * p.esz + 2 * p.osz + p.ndb * p.dsz; i += NT) lds[i] = 0.;
  const int nlines = (p.io1 - p.io0 + 1) * (p.n1 + 1);
  for (int line = blockIdx.x; line < nlines; line += gridDim.x) {
    const Line L = make_line(p, line);
    if (!L.vmask) continue;
    lds_barrier();  // previous line done (meta, buffers)
    if (tid == 0) plane_meta<NC>(p, L, S, 0);
    for (int k = 0; k < p.n2; ++k) {
      // ---- meta of the planes entering the ring, geometry of the 4 elements of this slice
      if (tid == 64) plane_meta<NC>(p, L, S, 2 * k + 1);
      if (tid == 128) plane_meta<NC>(p, L, S, 2 * k + 2);
      if (tid < 4 * p.nq) {
        const int v = tid / p.nq, q = tid - v * p.nq;
        if ((L.vmask >> v) & 1) geometry_point(p, L, v, k, q, S.Jv + (v * p.nq + q) * 10);
      }
      lds_barrier();
      int nbuilt = 0;
#pragma unroll 1
      for (int v = 0; v < 4; ++v) {
        if (!((L.vmask >> v) & 1)) continue;
        double *D = S.Dt + (p.ndb == 2 ? (nbuilt & 1) * p.dsz : 0);
        if (p.ndb == 1 && nbuilt) lds_barrier();  // single table: the tasks of the previous element have read it
        ++nbuilt;
        for (int idx = tid; idx < p.nq * NB; idx += NT) {
          const int q = idx / NB, n = idx - q * NB;
          const double *Tp = p.T + ((i64)n * p.nq + q) * 4;
          const double T4[4] = {Tp[0], Tp[1], Tp[2], Tp[3]};
          d_entry<NS>(p, T4, S.Jv + (v * p.nq + q) * 10, D, q, n);
        }
        lds_barrier();
        mfma_tasks<NC, NS, MODE>(p, L, S, D, S.Jv, v, k, wave, NT / 64, lane);
      }
      lds_barrier();  // all contributions of slice k are in LDS
      flush_plane<NC>(p, S, 2 * k, tid, NT);
      flush_plane<NC>(p, S, 2 * k + 1, tid, NT);
    }
    lds_barrier();
    flush_plane<NC>(p, S, 2 * p.n2, tid, NT);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// pipelined kernel (NS = 3 slots, 25..28 quadrature points = 7 k-steps): waves 0..3 -- one per SIMD -- run the MFMA tasks of
// element r while waves 4..6 build the D table of element r + 1 and stream the planes finished in the previous slice and wave 7
// evaluates the geometry of the next slice; one barrier per element.  On gfx950 an f64 MFMA occupies its SIMD for all 64 cycles: VALU
// work of a co-resident wave does NOT overlap it (tools/ubench/mfma_valu_overlap.hip), so every VALU instruction of any role adds to
// the matrix time of its SIMD; the roles exist to overlap LATENCIES (LDS, vertex loads, store issue), and all are written for the
// lowest instruction count: compile-time LDS layout, k-steps fully unrolled with immediate offsets, per-lane task constants computed
// once per line, the flush in node blocks with wave-uniform bases from records prepared by the geometry wave.
constexpr int PKS = 7, PNQ = 4 * PKS;  // k-steps, padded quadrature points
constexpr int NTP4 = 512, NTW = 3;     // threads; table / flush waves (waves 4 .. 4 + NTW - 1), wave 7: geometry
constexpr int FLW = (NTW + 1) * 64;    // threads that stream finished rows in the phases in which the geometry wave has nothing else to do

template <int NC>
struct PL {  // LDS layout in doubles from the start of the dynamic LDS
  static constexpr int ESZ = 320 * NC * NC + 8, OSZ = 192 * NC * NC + 8, DSZ = 3 * PKS * 128, JSZ = 4 * PNQ * 10;
  static constexpr int O = 3 * ESZ, DT = O + 2 * OSZ, JV = DT + 2 * DSZ, META = JV + 2 * JSZ, XV = META + 32 * 3, END = XV + 4 * 8 * 4;
  // meta, per (plane slot K & 7, node j): int2 row {offset of the node's rows, scalar row length}, int2 fl {first whole 16-byte pair,
  // pairs | head << 30 | tail << 31}, i64 element offset of the first row in the value array
  __device__ static __forceinline__ int plane(int K) { return (K & 1) ? O + ((K >> 1) & 1) * OSZ : ((K >> 1) % 3) * ESZ; }
};

// local node (ai, aj, ak) of row slot mu of unit u of visit V, or invalid
template <int V>
__device__ __forceinline__ bool unit_node(int u, int mu, int &ai, int &aj, int &ak) {
  if constexpr (V == 0) { ai = mu >> 1; aj = mu & 1; ak = u; return true; }
  if constexpr (V == 1) { aj = 2; if (u == 0) { ai = mu >> 1; ak = mu & 1; return true; } ai = mu & 1; ak = 2; return mu < 2; }
  if constexpr (V == 2) { ai = 2; if (u == 0) { aj = mu >> 1; ak = mu & 1; return true; } aj = mu & 1; ak = 2; return mu < 2; }
  ai = 2; aj = 2; ak = mu < 3 ? mu : 0; return mu < 3;
}

// what a lane needs to know about one of its wave's tasks (constant along a line)
struct TaskD {
  int Aoff;   // A operand: offset (doubles) of this lane's entry in the D table at k-step 0
  int Boff;   // B operand
  int posIJ;  // column position of this lane's (row node, column node) pair in the row, without the K part
  int info;   // bit 0: valid pair; bits 1-2: ak (0, 1: planes 2k, 2k+1 of this slice, 2: plane 2k+2); bits 3-4: node index in the plane; bits 5-6: nk
};

template <int V>
__device__ __forceinline__ TaskD make_task(const Line &L, int u, int nt, int lane) {
  const int lk = lane >> 4, li = lane & 15;
  constexpr int di = V >> 1, dj = V & 1;
  TaskD d;
  {
    int ai, aj, ak;
    const int aslot = li >> 2;
    const bool ok = unit_node<V>(u, li & 3, ai, aj, ak) && aslot < 3;
    const int node = ok ? (ai * 3 + aj) * 3 + ak : 31;  // node 31: a pad column of the table, always zero
    d.Aoff = (ok ? aslot : 0) * PKS * 128 + (node >> 4) * 64 + lk * 16 + (node & 15);
  }
  d.Boff = nt * 64 + lane;
  int ai, aj, ak;
  const bool okr = unit_node<V>(u, lk, ai, aj, ak);
  const int n = nt * 16 + li;
  const int alI = di ? 0 : ai, alJ = dj ? 0 : aj;
  const int ni = n / 9, nj = (n / 3) % 3, nk = n % 3;
  const int loI = alI ? L.loI[1] : L.loI[0], loJ = alJ ? L.loJ[1] : L.loJ[0], cJ = alJ ? L.cntJ[1] : L.cntJ[0];
  d.posIJ = (2 * (L.io - di) + ni - loI) * cJ + (2 * (L.jo - dj) + nj - loJ);
  d.info = (okr && n < NB ? 1 : 0) | ak << 1 | (alI * 2 + alJ) << 3 | nk << 5;
  return d;
}

// form tensor + LDS reduction of one lane's (row node, column node) pair: G[a][b] = acc[b][a]; m = row record of the node (offset of its rows in the plane
// buffer, scalar row length)
template <int NC, int MODE>
__device__ __forceinline__ void reduce_pair(const P2K &p, double *lds, const v4d (&acc)[3], const TaskD &d, int2 m, int cK0, int cK2, int dK0) {
  const int ak = (d.info >> 1) & 3;
  if ((d.info & 1) && !DBG(p, 2)) {
    double Kcd[NC][NC];
    if constexpr (MODE == 1) {  // C[c,a,d,b] = lam d_ca d_db + mu d_cd d_ab + mu2 d_cb d_ad over the three gradient slots
      const double tr = p.mu * (acc[0][0] + acc[1][1] + acc[2][2]);
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int dd = 0; dd < NC; ++dd) {
          Kcd[c][dd] = p.lam * acc[dd][c] + p.mu2 * acc[c][dd];
          if (c == dd) Kcd[c][dd] += tr;  // (not "+ (c == dd ? tr : 0.)": x + 0. is an instruction, -0. + 0. = +0.)
        }
    } else {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int dd = 0; dd < NC; ++dd) {
          double sum = 0;
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) sum += p.C[((c * 3 + a) * NC + dd) * 3 + b] * acc[b][a];
          Kcd[c][dd] = sum;
        }
    }
    // per-lane constants of the plane by masks (a select chain on ak is turned into a table in scratch by the compiler)
    const int m1 = -(ak & 1), m2 = -(ak >> 1), m0 = ~(m1 | m2);
    const int cK = cK0 + (m1 & (3 - cK0)) + (m2 & (cK2 - cK0));
    const int pos = d.posIJ * cK + ((d.info >> 5) & 3) + (m0 & dK0);
    double *row = lds + (m.x + pos * NC);
    const int rs = m.y * NC;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int dd = 0; dd < NC; ++dd) atomicAdd(row + c * rs + dd, Kcd[c][dd]);
  }
}

// k-steps [K0, K1) of one (unit, column tile) product + the form tensor + the LDS reduction.
// cK0 / cK2: columns along K of a row of plane 2k / 2k+2 (3 in a boundary plane, else 5; plane 2k+1: 3); dK0 = 2k - first column of
// a row of plane 2k; m0 = meta slot of plane 2k
template <int NC, int MODE, int K0, int K1>
__device__ __forceinline__ void run_task(const P2K &p, double *lds, int cK0, int cK2, int dK0, int m0s, const TaskD &d, const double *D, const double *Jw, int lane) {
  const int lk = lane >> 4;
  const int ak = (d.info >> 1) & 3, j = (d.info >> 3) & 3;
  // where the row lives: read ahead of the MFMA chain
  const int2 m = *reinterpret_cast<const int2 *>(reinterpret_cast<const int *>(lds + PL<NC>::META) + ((((m0s + ak) & 7) * 4 + j) * 2));
  const double *pa = D + d.Aoff, *pb = D + d.Boff, *pw = Jw + lk * 10;
  v4d acc[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) acc[b] = v4d{0., 0., 0., 0.};
  if (!DBG(p, 4)) {
    double an = pa[K0 * 128], wn = pw[K0 * 40], bn[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) bn[b] = pb[(b * PKS + K0) * 128];
#pragma unroll
    for (int ks = K0; ks < K1; ++ks) {
      const double a = an * wn;  // the weight w |J| enters through the A operand
      double bv[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) bv[b] = bn[b];
      if (ks + 1 < K1) {  // operands of the next k-step are in flight while the matrix pipe works
        an = pa[(ks + 1) * 128];
        wn = pw[(ks + 1) * 40];
#pragma unroll
        for (int b = 0; b < 3; ++b) bn[b] = pb[(b * PKS + ks + 1) * 128];
      }
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[b], acc[b], 0, 0, 0);
    }
  }
  reduce_pair<NC, MODE>(p, lds, acc, d, m, cK0, cK2, dK0);
}

// one thread: the records of the 4 nodes (ai, aj) of node plane K
template <int NC>
__device__ __forceinline__ void pipe_meta(const P2K &p, const Line &L, double *lds, int K) {
  int *meta = reinterpret_cast<int *>(lds + PL<NC>::META);
  i64 *gm = reinterpret_cast<i64 *>(lds + PL<NC>::META + 64);
  const int cK = ax_cnt(K, p.n2), cumK = ax_cum(K, p.n2);
  int cur = PL<NC>::plane(K);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ai = j >> 1, aj = j & 1;
    bool ex = false;  // the node has contributions iff an element that contains it is visited
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (((L.vmask >> v) & 1) && (!(v >> 1) || ai == 0) && (!(v & 1) || aj == 0)) ex = true;
    const int len = ex ? L.cntI[ai] * L.cntJ[aj] * cK : 0, nd = len * NC * NC;
    const i64 row0 = (i64)L.cumI[ai] * L.SJ * L.SK + (i64)L.cntI[ai] * ((i64)L.cumJ[aj] * L.SK + (i64)L.cntJ[aj] * cumK);
    const i64 goff = row0 * (NC * NC);
    const int head = nd ? (int)(goff & 1) : 0;
    cur = ((cur + 1) & ~1) + head;  // blocks never share a 16-byte pair: the flush copies and zeroes whole pairs
    const int s = (K & 7) * 4 + j;
    meta[s * 2] = cur;
    meta[s * 2 + 1] = len;
    meta[64 + s * 2] = cur + head;
    meta[64 + s * 2 + 1] = ((nd - head) >> 1) | head << 30 | ((nd - head) & 1) << 31;
    gm[s] = goff;
    cur += nd;
  }
}

// stream the finished rows of NBLK node blocks (block c = plane K0 + (c >> 2), node c & 3) to the value array and zero them in the
// buffers: 16-byte pairs with wave-uniform bases (a block is contiguous in both memories and starts with the same parity), the odd head
// / tail element by one lane.  All records are read together, then all data, then the stores: the flushing waves have nothing else
// to hide an LDS round trip behind.
template <int NC, int NBLK, int UB = 3>  // UB pairs per thread and block: 125 * 9 / 2 = 563 <= UB * nt
__device__ __forceinline__ void flush_blocks(const P2K &p, double *lds, int K0, int c0, int t, int nt, long long *ft = nullptr) {
#if defined(NH_ABLATION) && !defined(NH_NOTICKS)
  long long f0 = __builtin_readcyclecounter();
#define FT(i) do { if (ft) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long t_ = __builtin_readcyclecounter(); ft[i] += t_ - f0; f0 = t_; } } while (0)
#else
#define FT(i) do {} while (0)
#endif
  const int *meta = reinterpret_cast<const int *>(lds + PL<NC>::META);
  const i64 *gm = reinterpret_cast<const i64 *>(lds + PL<NC>::META + 64);
  int2 f[NBLK];
  i64 g[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    const int s = ((K0 + ((c0 + b) >> 2)) & 7) * 4 + ((c0 + b) & 3);
    f[b] = *reinterpret_cast<const int2 *>(meta + 64 + s * 2);
    g[b] = gm[s];
  }
  FT(0);
  // read-and-clear in ONE LDS round trip per value (ds_wrxchg_rtn_b64): the separate zeroing pass was a second trip through an LDS queue that the
  // ds_add_f64 stream of the MFMA waves keeps ~40 % busy (1.6 k + 0.9 k cycles per call of 4.0 k)
  v2d v[NBLK][UB];
  double hv[NBLK], tv[NBLK];
  auto grab = [](double *q) { return __hip_atomic_exchange(q, 0., __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    const int first = __builtin_amdgcn_readfirstlane(f[b].x), w = __builtin_amdgcn_readfirstlane(f[b].y);
    const int np = w & 0x3fffffff;
    double *lp = lds + first;
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (t + u * nt < np) {
        if (DBG(p, 32)) {  // A/B: plain 16-byte read + 16-byte zero store instead of two returning exchanges
          v[b][u] = *reinterpret_cast<const v2d *>(lp + 2 * (t + u * nt));
          *reinterpret_cast<v2d *>(lp + 2 * (t + u * nt)) = v2d{0., 0.};
        } else
          v[b][u] = v2d{grab(lp + 2 * (t + u * nt)), grab(lp + 2 * (t + u * nt) + 1)};
      }
    if (t == 0 && (w & (1 << 30))) hv[b] = grab(lds + first - 1);
    if (t == 1 && (w < 0)) tv[b] = grab(lds + first + 2 * np);
  }
  FT(1);
  FT(2);
  if (DBG(p, 1)) return;
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    const int w = __builtin_amdgcn_readfirstlane(f[b].y);
    const int np = w & 0x3fffffff, head = (w >> 30) & 1;
    const i64 goff = ((i64)__builtin_amdgcn_readfirstlane((int)(g[b] >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)g[b]);
    double *gb = p.values + goff + head;
    v2d *gp = reinterpret_cast<v2d *>(gb);
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (t + u * nt < np) gp[t + u * nt] = v[b][u];
    if (t == 0 && head) gb[-1] = hv[b];
    if (t == 1 && (w < 0)) gb[2 * np] = tv[b];
  }
  FT(3);
#undef FT
}

template <int NC, int MODE>
__device__ __forceinline__ void mfma_role(const P2K &p, double *lds, int wave, int lane) {
#ifdef NH_ABLATION
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  const int nlines = (p.io1 - p.io0 + 1) * (p.n1 + 1);
  for (int line = blockIdx.x; line < nlines; line += gridDim.x) {
    const Line L = make_line(p, line);
    if (!L.vmask) continue;
    const int nv = __builtin_popcount(L.vmask), nph = nv < 2 ? 2 : nv;
    // the tasks of this wave, one matrix pipe each: element 0 (6 products): a whole one + half the k-steps of another;
    // elements 1, 2 (4 products): one each; element 3 (2 products): half of one
    const TaskD d0 = make_task<0>(L, wave >> 1, wave & 1, lane), d0h = make_task<0>(L, 2, wave >> 1, lane);
    const TaskD d1 = make_task<1>(L, wave >> 1, wave & 1, lane), d2 = make_task<2>(L, wave >> 1, wave & 1, lane), d3h = make_task<3>(L, 0, wave >> 1, lane);
    lds_barrier();
    lds_barrier();
    lds_barrier();
    int r = 0;
#pragma unroll 1
    for (int k = 0; k < p.n2; ++k) {
      const int cK0 = ax_cnt(2 * k, p.n2), cK2 = ax_cnt(2 * k + 2, p.n2), dK0 = k > 0 ? 2 : 0, m0s = (2 * k) & 7;
      const double *Jvs = lds + PL<NC>::JV + (k & 1) * PL<NC>::JSZ + 9;
#pragma unroll 1
      for (int i = 0; i < nph; ++i) {
        TICK(7);
        if (i < nv) {
          const int v = nth_visit(L.vmask, i);
          const double *D = lds + PL<NC>::DT + (r & 1) * PL<NC>::DSZ, *Jw = Jvs + v * PNQ * 10;
          if (v == 0) {
            run_task<NC, MODE, 0, PKS>(p, lds, cK0, cK2, dK0, m0s, d0, D, Jw, lane);
            if (wave & 1) run_task<NC, MODE, 4, PKS>(p, lds, cK0, cK2, dK0, m0s, d0h, D, Jw, lane);
            else run_task<NC, MODE, 0, 4>(p, lds, cK0, cK2, dK0, m0s, d0h, D, Jw, lane);
          } else if (v == 1) {
            run_task<NC, MODE, 0, PKS>(p, lds, cK0, cK2, dK0, m0s, d1, D, Jw, lane);
          } else if (v == 2) {
            run_task<NC, MODE, 0, PKS>(p, lds, cK0, cK2, dK0, m0s, d2, D, Jw, lane);
          } else {
            if (wave & 1) run_task<NC, MODE, 4, PKS>(p, lds, cK0, cK2, dK0, m0s, d3h, D, Jw, lane);
            else run_task<NC, MODE, 0, 4>(p, lds, cK0, cK2, dK0, m0s, d3h, D, Jw, lane);
          }
          ++r;
        }
        TICK(0);
        lds_barrier();
        TICK(1);
      }
    }
    lds_barrier();  // the table waves stream the planes of the last slice
  }
#ifdef NH_ABLATION
  if (p.tdbg && lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)p.tdbg + i, (unsigned long long)tacc[i]);
#endif
}

// waves 4 .. 6: D tables one element ahead, flush of the previous slice
template <int NC, int S0>
__device__ __forceinline__ void table_role(const P2K &p, double *lds, int lane, int st) {
  // this thread's share of every D table: point q = st / 7, nodes 4 g .. 4 g + 3 (g = st % 7); its slice of the basis table stays in registers
  const bool tok = st < p.nq * 7;
  const int tq = tok ? st / 7 : 0, tg = tok ? st % 7 : 0;
  double T4[4][4];
  int to[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = min(4 * tg + r, NB - 1);
    to[r] = ((tq >> 2) * 2 + (n >> 4)) * 64 + (tq & 3) * 16 + (n & 15);
#pragma unroll
    for (int s = 0; s < 4; ++s) T4[r][s] = p.T[((i64)n * p.nq + tq) * 4 + s];
  }
  // pin the table values HERE: without a use the loads stay "pending" for the compiler's wait-count bookkeeping, and it puts
  // s_waitcnt vmcnt(0) in front of their first use -- inside the D-table build of every element, where that waits for all the
  // stores of the flush issued before it (microseconds under load)
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(T4[r][s]));
#ifdef NH_ABLATION
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
  long long ftacc[4] = {0, 0, 0, 0};
#define FTP ftacc
#else
#define FTP nullptr
#endif
  const int nlines = (p.io1 - p.io0 + 1) * (p.n1 + 1);
  for (int line = blockIdx.x; line < nlines; line += gridDim.x) {
    const int io = p.io0 + line / (p.n1 + 1), jo = line % (p.n1 + 1);
    int vmask = 0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int ei = io - (v >> 1), ej = jo - (v & 1);
      if (ei >= p.l0 && ei < p.l1 && ej >= 0 && ej < p.n1) vmask |= 1 << v;
    }
    if (!vmask) continue;
    const int nv = __builtin_popcount(vmask), nph = nv < 2 ? 2 : nv;
    auto build_D = [&](int k, int v, double *D) {  // D table of element (v, k): physical derivatives in MFMA operand order
      if (DBG(p, 8) || !tok) return;
      const double *Ji = lds + PL<NC>::JV + (k & 1) * PL<NC>::JSZ + (v * PNQ + tq) * 10;
      double J9[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) J9[i] = Ji[i];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double d[4];
        d[0] = T4[r][0];
#pragma unroll
        for (int i = 0; i < 3; ++i) d[1 + i] = T4[r][1] * J9[i] + T4[r][2] * J9[3 + i] + T4[r][3] * J9[6 + i];
#pragma unroll
        for (int a = 0; a < 3; ++a) D[to[r] + a * PKS * 128] = d[S0 + a];  // active slots S0 .. S0 + 2 (group 6 writes node 26 twice)
      }
    };
    lds_barrier();  // previous line done
    lds_barrier();  // geometry of slice 0
    build_D(0, nth_visit(vmask, 0), lds + PL<NC>::DT);
    lds_barrier();
    int r = 0;  // elements done
#pragma unroll 1
    for (int k = 0; k < p.n2; ++k) {
#pragma unroll 1
      for (int i = 0; i < nph; ++i) {
        TICK(7);
        // D table of the next element: within the slice right away, across slices in the last phase (its geometry is due in the first)
        if (i + 1 < nv) build_D(k, nth_visit(vmask, i + 1), lds + PL<NC>::DT + ((r + 1) & 1) * PL<NC>::DSZ);
        else if (i == nph - 1 && k + 1 < p.n2) build_D(k + 1, nth_visit(vmask, 0), lds + PL<NC>::DT + (((k + 1) * nv) & 1) * PL<NC>::DSZ);
        TICK(2);
        if (k > 0 && !DBG(p, 128)) {  // the planes 2k-2, 2k-1 finished in the previous slice: 8 node blocks, nph phases
          if (nph == 4) {
            // 3 + 2 + 2 + 1 node blocks over the four phases: the matrix waves have 31.5 / 21 / 21 / 10.5 MFMA per phase, and a phase lasts as long as the
            // slower of the two roles -- an even 2 + 2 + 2 + 2 left the table waves waiting in the first phase and the matrix waves in the last.
            // From the second phase on the geometry wave (done with the next slice by then) streams a quarter of every block: FLW threads.
            if (i == 0) flush_blocks<NC, 3>(p, lds, 2 * k - 2, 0, st, NTW * 64, FTP);
            else if (i == 3) flush_blocks<NC, 1>(p, lds, 2 * k - 2, 7, st, FLW, FTP);
            else flush_blocks<NC, 2>(p, lds, 2 * k - 2, 2 * i + 1, st, FLW, FTP);
          }
          else flush_blocks<NC, 4>(p, lds, 2 * k - 2, 4 * i, st, NTW * 64);
        }
        TICK(4);
        if (i < nv) ++r;
        lds_barrier();
        TICK(5);
      }
    }
    // the planes of the last slice
#pragma unroll 1
    for (int c = 0; c < 12; c += 2) flush_blocks<NC, 2>(p, lds, 2 * p.n2 - 2, c, st, NTW * 64);
    lds_barrier();
  }
#ifdef NH_ABLATION
  if (p.tdbg && lane == 0)
    for (int i = 0; i < 4; ++i) atomicAdd((unsigned long long *)p.tdbg + 8 + i, (unsigned long long)ftacc[i]);
#endif
#ifdef NH_ABLATION
  if (p.tdbg && lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)p.tdbg + i, (unsigned long long)tacc[i]);
#endif
}

// row records of node plane K by FOUR lanes (node j each): the constants of the line are in `LM`, a record costs one 64-bit multiply-add and a few
// integer operations (the serial per-plane version, one lane doing four nodes with 64-bit products from scratch, was ~2 k cycles of the critical path)
struct LineMeta {
  int lenIJ[4];  // scalar row length of node j without its K factor (0: node absent)
  i64 G0[4];     // element offset of the first row of node j in the value array: G0 + G1 * ax_cum(K)
  int G1[4];
};

template <int NC>
__device__ __forceinline__ LineMeta make_line_meta(const Line &L) {
  LineMeta M;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ai = j >> 1, aj = j & 1;
    bool ex = false;  // the node has contributions iff an element that contains it is visited
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (((L.vmask >> v) & 1) && (!(v >> 1) || ai == 0) && (!(v & 1) || aj == 0)) ex = true;
    M.lenIJ[j] = ex ? L.cntI[ai] * L.cntJ[aj] : 0;
    M.G0[j] = ((i64)L.cumI[ai] * L.SJ * L.SK + (i64)L.cntI[ai] * ((i64)L.cumJ[aj] * L.SK)) * (NC * NC);
    M.G1[j] = L.cntI[ai] * L.cntJ[aj] * (NC * NC);
  }
  return M;
}

template <int NC>
__device__ __forceinline__ void pipe_meta_node(const P2K &p, const LineMeta &M, double *lds, int K, int j) {
  int *meta = reinterpret_cast<int *>(lds + PL<NC>::META);
  i64 *gm = reinterpret_cast<i64 *>(lds + PL<NC>::META + 64);
  const int cK = ax_cnt(K, p.n2), cumK = ax_cum(K, p.n2);
  int cur = PL<NC>::plane(K), nd = 0, head = 0;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    if (jj <= j) {
      cur += nd;  // (rows of the previous node)
      nd = M.lenIJ[jj] * cK * NC * NC;
      head = nd ? (int)((M.G0[jj] + (i64)M.G1[jj] * cumK) & 1) : 0;
      cur = ((cur + 1) & ~1) + head;  // blocks never share a 16-byte pair: the flush copies and zeroes whole pairs
    }
  }
  // (the constants of node j by compare-and-select: a runtime index into the register arrays would put them into scratch memory)
  int lenj = 0, g1 = 0;
  i64 g0 = 0;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
    if (jj == j) {
      lenj = M.lenIJ[jj];
      g0 = M.G0[jj];
      g1 = M.G1[jj];
    }
  const int s = (K & 7) * 4 + j;
  meta[s * 2] = cur;
  meta[s * 2 + 1] = lenj * cK;
  meta[64 + s * 2] = cur + head;
  meta[64 + s * 2 + 1] = ((nd - head) >> 1) | head << 30 | ((nd - head) & 1) << 31;
  gm[s] = g0 + (i64)g1 * cumK;
}

// wave 7: geometry of the next slice, row records of the planes that enter the ring with it.
// Trilinear isoparametric geometry (8 geometry functions, the case of configs[2]) takes the cheap route: the 8 vertices of each of the four elements of a slice
// are fetched ONCE (one lane per vertex, one slice ahead: the index -> coordinate chain of two memory latencies stays off the phase), staged in LDS, and a lane per
// (element, point) forms J from them with the reference gradients of ITS point held in registers -- the generic geometry_point (per point: 8 index loads, 24
// coordinate loads, 32 table loads) made this one wave the critical path of every phase (profiles/r03_c3_geometry_role.md).
template <int NC>
__device__ __forceinline__ void geometry_role(const P2K &p, double *lds, int lane) {
#ifdef NH_ABLATION
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  const int nq = p.nq;
  const bool fast = p.geom.kind == NH_GEOM_ISO && p.geom.ngb == 8 && !p.geom.nograd && p.geom.bnd_axis < 0;
  const int gq = lane < nq ? lane : lane - nq;  // the point of this lane in both rounds: items (v, q) = (2 r + (lane >= nq), gq), r = 0, 1
  const bool gok = lane < 2 * nq;
  double dN[8][3], wq = 0.;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int j = 0; j < 3; ++j) dN[a][j] = 0.;
  if (fast && gok) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int j = 0; j < 3; ++j) dN[a][j] = p.geom.gT[((i64)a * nq + gq) * 4 + 1 + j];
    wq = p.weights[gq];
  }
  // (pinned here: loads without a use stay pending for the wait-count bookkeeping and would be waited for inside the first slice)
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(dN[a][j]));
  asm volatile("" : "+v"(wq));
  double *XV = lds + PL<NC>::XV;  // [4 visits][8 vertices][4]: staged vertex coordinates of one slice
  const int nlines = (p.io1 - p.io0 + 1) * (p.n1 + 1);
  for (int line = blockIdx.x; line < nlines; line += gridDim.x) {
    const Line L = make_line(p, line);
    if (!L.vmask) continue;
    const LineMeta LM = make_line_meta<NC>(L);
    const int nv = __builtin_popcount(L.vmask), nph = nv < 2 ? 2 : nv;
    auto geometry_slice = [&](int k) {  // generic route: all valid (v, q) of slice k
      double *Jvs = lds + PL<NC>::JV + (k & 1) * PL<NC>::JSZ;
      for (int i = lane; i < 4 * p.nq; i += 64) {
        const int v = i / p.nq, q = i - v * p.nq;
        if (((L.vmask >> v) & 1) && !DBG(p, 16)) geometry_point(p, L, v, k, q, Jvs + (v * PNQ + q) * 10);
      }
    };
    // cheap route, lanes 0 .. 31 = (visit, vertex): coordinates of the vertices of slice k into registers
    const int vv = lane >> 3, va = lane & 7;
    const bool vok = fast && lane < 32 && ((L.vmask >> vv) & 1);
    const i64 ecol = ((i64)(L.io - (vv >> 1)) * p.n1 + (L.jo - (vv & 1))) * p.n2;
    double X0 = 0., X1 = 0., X2 = 0.;
    auto fetch_vertices = [&](int k) {
      if (vok && k < p.n2) {
        const i64 idx = p.geom.gdofs[(ecol + k) * 8 + va];
        X0 = p.geom.verts[idx * 3];
        X1 = p.geom.verts[idx * 3 + 1];
        X2 = p.geom.verts[idx * 3 + 2];
      }
    };
    auto fast_slice = [&](int k) {  // registers -> LDS -> J, J^-1, w |det J| of all valid (v, q) of slice k
      if (vok) {
        double *o = XV + lane * 4;
        *reinterpret_cast<v2d *>(o) = v2d{X0, X1};
        o[2] = X2;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (same wave writes and reads: no barrier)
      double *Jvs = lds + PL<NC>::JV + (k & 1) * PL<NC>::JSZ;
#pragma unroll 1
      for (int r = 0; r < 2; ++r) {
        const int v = 2 * r + (lane >= nq ? 1 : 0);
        if (!gok || !((L.vmask >> v) & 1) || DBG(p, 16)) continue;
        double J[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) J[i][j] = 0.;
        const double *Xe = XV + v * 32;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const v2d x01 = *reinterpret_cast<const v2d *>(Xe + a * 4);
          const double x[3] = {x01[0], x01[1], Xe[a * 4 + 2]};
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) J[i][j] += x[i] * dN[a][j];
        }
        double Ji[3][3], det;
        invert<3>(J, Ji, det);
        double *o = Jvs + (v * PNQ + gq) * 10;
        const i64 e = ((i64)(L.io - (v >> 1)) * p.n1 + (L.jo - (v & 1))) * p.n2 + k;
        const double w = wq * fabs(det) * (p.scale ? p.scale[e * nq + gq] : 1.);
        // o[j * 3 + i] = Ji[j][i], o[9] = w: five 16-byte stores (a point's record is 80 bytes)
        *reinterpret_cast<v2d *>(o) = v2d{Ji[0][0], Ji[0][1]};
        *reinterpret_cast<v2d *>(o + 2) = v2d{Ji[0][2], Ji[1][0]};
        *reinterpret_cast<v2d *>(o + 4) = v2d{Ji[1][1], Ji[1][2]};
        *reinterpret_cast<v2d *>(o + 6) = v2d{Ji[2][0], Ji[2][1]};
        *reinterpret_cast<v2d *>(o + 8) = v2d{Ji[2][2], w};
      }
    };
    lds_barrier();  // previous line done
    if (lane < 12) pipe_meta_node<NC>(p, LM, lds, lane >> 2, lane & 3);
    if (fast) {
      fetch_vertices(0);
      fast_slice(0);
      fetch_vertices(1);  // (in flight until the first phase of slice 0)
    } else
      geometry_slice(0);
    lds_barrier();
    lds_barrier();
#pragma unroll 1
    for (int k = 0; k < p.n2; ++k) {
#pragma unroll 1
      for (int i = 0; i < nph; ++i) {
        TICK(7);
        if (k + 1 < p.n2) {
          if (i == 0 && !DBG(p, 64) && lane < 8) pipe_meta_node<NC>(p, LM, lds, 2 * k + 3 + (lane >> 2), lane & 3);
          double *Jvs = lds + PL<NC>::JV + ((k + 1) & 1) * PL<NC>::JSZ;
          if (fast) {
            // everything in the first phase (the one with the most matrix work): the records of slice k + 1 are first read by the table build of the LAST phase
            if (i == 0) {
              fast_slice(k + 1);
              fetch_vertices(k + 2);
            }
          } else if (nph == 4) {
            // generic route: one element per phase (its table is due one element ahead of it); lines with two or three phases keep everything in the first
            if (!DBG(p, 16))
              for (int q = lane; q < p.nq; q += 64) geometry_point(p, L, i, k + 1, q, Jvs + (i * PNQ + q) * 10);
          } else if (i == 0)
            geometry_slice(k + 1);
        }
        TICK(3);
        if (nph == 4 && i > 0 && k > 0 && !DBG(p, 128)) {  // the fourth streaming wave of these phases (table_role: the same blocks, FLW threads)
          if (i == 3) flush_blocks<NC, 1>(p, lds, 2 * k - 2, 7, NTW * 64 + lane, FLW);
          else flush_blocks<NC, 2>(p, lds, 2 * k - 2, 2 * i + 1, NTW * 64 + lane, FLW);
        }
        lds_barrier();
        TICK(6);
      }
    }
    lds_barrier();
  }
#ifdef NH_ABLATION
  if (p.tdbg && lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)p.tdbg + i, (unsigned long long)tacc[i]);
#endif
}

template <int NC, int S0, int MODE>
__global__ __launch_bounds__(NTP4) void k_p2hex_pipe(P2K p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // row buffers are re-zeroed by the flush; the pad entries of the D tables (q >= nq, n >= 27) and the weights of the pad points stay zero
  for (int i = tid; i < PL<NC>::META; i += NTP4) lds[i] = 0.;
  if (wave < 4) {
    mfma_role<NC, MODE>(p, lds, wave, lane);
    return;
  }
  // an f64 MFMA holds its SIMD for 64 cycles and the MFMA waves issue them back to back: without priority the co-resident wave gets
  // about one issue slot per MFMA, and the element-ahead work of these roles falls behind the matrix pipe
  if (p.prio) __builtin_amdgcn_s_setprio(3);
  if (wave < 4 + NTW) table_role<NC, S0>(p, lds, lane, tid - 256);
  else {
#ifdef NH_ABLATION
    const long long t0 = __builtin_readcyclecounter();
#endif
    geometry_role<NC>(p, lds, lane);
#ifdef NH_ABLATION
    if (p.tdbg && lane == 0) p.tdbg[16 + blockIdx.x] = __builtin_readcyclecounter() - t0;
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_p2hex_inreg: the MFMA operands are formed IN REGISTERS, no D tables.  Waves 0..3 (one per SIMD) own a (visit group, column tile) each: wave w takes
// column tile w >> 1 of the elements of visits {0, 3} (w even: 3 + 1 units of 4 row nodes) or {1, 2} (w odd: 2 + 2 units) -- 84 MFMA per wave and slice, the
// B operand of a k-step formed once and used by all units of the element.  Per k-step a lane reads ONE double of the point's record ([J^-1 sqrt(w |J|)] row
// major, 12 doubles per point, lane li of the 16 lanes of a point reads entry li) and the products with the reference gradients of ITS column node (28 doubles
// in registers for the whole kernel) take the multiplicand from lane N of the row: v_fmac_f64_dpp row_newbcast:N (tools/ubench/dpp_bcast.hip: the rate of a plain
// v_fmac_f64).  The A operand (row node mu, slot a) reads column a of the record and the reference gradients of its node from a 25 kB table in LDS.  What the
// D tables cost is gone: their build (2187 entries per element, four times per element), ~45 kB of LDS, and the barrier per ELEMENT that handed them over -- the
// roles meet once per SLICE.  Waves 4..7: wave 4 + v evaluates the geometry of visit v one slice ahead (27 lanes: one per point; 8 lanes fetch the vertices
// another slice ahead), all four stream the two planes finished in the previous slice (256 threads, 13 pair slots per thread for NC = 3).
// sqrt(w |J|) is folded into J^-1 (both operands carry it): forms with a signed scale array keep k_p2hex_pipe.
constexpr int RJ = 12;                       // doubles per point record: [0..8] J^-1[j][i] sqrt(w |J|) at 3 j + i, [9] sqrt(w |J|), [10], [11] zero
constexpr int RJSZ = 4 * PNQ * RJ;           // one slice: [visit][point]
constexpr int RTSZ = 28 * PNQ * 4;           // reference table of the A side: [node (27: zero row)][point][dN/dxi_0, dN/dxi_1, dN/dxi_2, N]
template <int NC>
struct PR {  // the D-table and J regions of PL hold the two record buffers and the A-side table
  static constexpr int JV = PL<NC>::DT, TT = JV + 2 * RJSZ;
  static_assert(TT + RTSZ <= PL<NC>::META, "LDS layout");
};

template <int N>
__device__ __forceinline__ void fmac_bc(double &acc, double x, double t) {  // acc += (x of lane N of this lane's row of 16) * t
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(t), "n"(N));
}

// B operand of the three active slots S0 .. S0 + 2 of this lane's column node at its point: tb = {N, dN/dxi_0..2}, xr = this lane's entry of the point record
template <int S0>
__device__ __forceinline__ void form_B(double xr, const double (&tb)[4], double (&B)[3]) {
  B[0] = B[1] = B[2] = 0.;
  if constexpr (S0 == 1) {
    fmac_bc<0>(B[0], xr, tb[1]); fmac_bc<1>(B[1], xr, tb[1]); fmac_bc<2>(B[2], xr, tb[1]);
    fmac_bc<3>(B[0], xr, tb[2]); fmac_bc<4>(B[1], xr, tb[2]); fmac_bc<5>(B[2], xr, tb[2]);
    fmac_bc<6>(B[0], xr, tb[3]); fmac_bc<7>(B[1], xr, tb[3]); fmac_bc<8>(B[2], xr, tb[3]);
  } else {
    fmac_bc<9>(B[0], xr, tb[0]);
    fmac_bc<0>(B[1], xr, tb[1]); fmac_bc<1>(B[2], xr, tb[1]);
    fmac_bc<3>(B[1], xr, tb[2]); fmac_bc<4>(B[2], xr, tb[2]);
    fmac_bc<6>(B[1], xr, tb[3]); fmac_bc<7>(B[2], xr, tb[3]);
  }
  // a VALU f64 write needs two wait states before a v_mfma_f64 reads the register (tools/ubench/dpp_mfma_hazard.hip); the compiler inserts them for the
  // instructions it knows, not for the inline assembly above
  asm("s_nop 1" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]));
}

struct LaneK {   // per-lane constants of the operand fetch
  int xoff;      // entry of the point record this lane holds for the broadcasts
  int aoff[3];   // entries J^-1[j][a] of the A side (value-slot lanes: a zero entry)
  int awoff;     // S0 = 0: sqrt(w |J|) for the value-slot lanes, a zero entry for the others
};

// all units of one element for one column tile: 7 k-steps, then form tensor + LDS reduction of every unit
template <int NC, int S0, int MODE, int NU, int KS0 = 0, int KS1 = PKS>
__device__ __forceinline__ void element_task(const P2K &p, double *lds, const double *jv, const double (&TB)[PKS][4], const LaneK &lc, const int (&taoff)[NU],
                                             const TaskD (&d)[NU], int cK0, int cK2, int dK0, int m0s) {
  int2 m[NU];  // where the rows live: read ahead of the MFMA chain
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int ak = (d[u].info >> 1) & 3, j = (d[u].info >> 3) & 3;
    m[u] = *reinterpret_cast<const int2 *>(reinterpret_cast<const int *>(lds + PL<NC>::META) + ((((m0s + ak) & 7) * 4 + j) * 2));
  }
  v4d acc[NU][3];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[u][b] = v4d{0., 0., 0., 0.};
  const double *TT = lds + PR<NC>::TT;
  struct Ops {
    double xr, ac[3], aw;
    v2d ta[NU][2];
  };
  auto fetch = [&](int ks, Ops &o) {
    const double *r = jv + ks * (4 * RJ);
    o.xr = r[lc.xoff];
#pragma unroll
    for (int j = 0; j < 3; ++j) o.ac[j] = r[lc.aoff[j]];
    o.aw = S0 == 0 ? r[lc.awoff] : 0.;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      o.ta[u][0] = *reinterpret_cast<const v2d *>(TT + taoff[u] + ks * 16);
      o.ta[u][1] = *reinterpret_cast<const v2d *>(TT + taoff[u] + ks * 16 + 2);
    }
  };
  if (!DBG(p, 4)) {
    constexpr bool PREF = KS1 - KS0 == PKS;  // (the split variant: the other matrix wave of the SIMD covers the operand latency, and the second operand set does not fit its registers)
    Ops cur, nxt;
    fetch(KS0, cur);
    asm volatile("s_nop 4");  // (exec written by the branch above -> first DPP read)
#pragma unroll
    for (int ks = KS0; ks < KS1; ++ks) {
      if (PREF && ks + 1 < KS1) fetch(ks + 1, nxt);  // operands of the next k-step are in flight while the matrix pipe works
      double B[3], A[NU];
      form_B<S0>(cur.xr, TB[ks], B);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        A[u] = cur.ta[u][0][0] * cur.ac[0] + cur.ta[u][0][1] * cur.ac[1] + cur.ta[u][1][0] * cur.ac[2];
        if constexpr (S0 == 0) A[u] += cur.ta[u][1][1] * cur.aw;
      }
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[u][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[u], B[b], acc[u][b], 0, 0, 0);
      if (ks + 1 < KS1) {
        if constexpr (PREF) cur = nxt;
        else fetch(ks + 1, cur);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) reduce_pair<NC, MODE>(p, lds, acc[u], d[u], m[u], cK0, cK2, dK0);
}

// offset (doubles) in the A-side table of the entry of this lane's A row (node mu = li & 3 of unit u of visit V, slot li >> 2) at the point of k-step 0
template <int V>
__device__ __forceinline__ int ta_offset(int u, int lane) {
  const int lk = lane >> 4, li = lane & 15;
  int ai, aj, ak;
  const bool ok = unit_node<V>(u, li & 3, ai, aj, ak) && (li >> 2) < 3;
  const int node = ok ? (ai * 3 + aj) * 3 + ak : 27;
  return (node * PNQ + lk) * 4;
}

template <int NC, int S0, int MODE, int VA, int NUA, int VB, int NUB, int KS0 = 0, int KS1 = PKS>
__device__ __forceinline__ void inreg_mfma_lines(const P2K &p, double *lds, const double (&TB)[PKS][4], const LaneK &lc, int nt, int lane) {
  const int lk = lane >> 4;
#ifdef NH_ABLATION
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  int taA[NUA], taB[NUB];
#pragma unroll
  for (int u = 0; u < NUA; ++u) taA[u] = ta_offset<VA>(u, lane);
#pragma unroll
  for (int u = 0; u < NUB; ++u) taB[u] = ta_offset<VB>(u, lane);
  const int nlines = (p.io1 - p.io0 + 1) * (p.n1 + 1);
  for (int line = blockIdx.x; line < nlines; line += gridDim.x) {
    const Line L = make_line(p, line);
    if (!L.vmask) continue;
    TaskD dA[NUA], dB[NUB];
#pragma unroll
    for (int u = 0; u < NUA; ++u) dA[u] = make_task<VA>(L, u, nt, lane);
#pragma unroll
    for (int u = 0; u < NUB; ++u) dB[u] = make_task<VB>(L, u, nt, lane);
    const bool hasA = (L.vmask >> VA) & 1, hasB = (L.vmask >> VB) & 1;
    lds_barrier();  // the last planes of the previous line are out
    lds_barrier();  // records and geometry of slice 0
#pragma unroll 1
    for (int k = 0; k < p.n2; ++k) {
      const int cK0 = ax_cnt(2 * k, p.n2), cK2 = ax_cnt(2 * k + 2, p.n2), dK0 = k > 0 ? 2 : 0, m0s = (2 * k) & 7;
      const double *jvs = lds + PR<NC>::JV + (k & 1) * RJSZ + lk * RJ;
      if (hasA) element_task<NC, S0, MODE, NUA, KS0, KS1>(p, lds, jvs + VA * PNQ * RJ, TB, lc, taA, dA, cK0, cK2, dK0, m0s);
      if (hasB) element_task<NC, S0, MODE, NUB, KS0, KS1>(p, lds, jvs + VB * PNQ * RJ, TB, lc, taB, dB, cK0, cK2, dK0, m0s);
      TICK(0);
      lds_barrier();
      TICK(1);
    }
    TICK(7);
  }
#ifdef NH_ABLATION
  if (p.tdbg && lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)p.tdbg + i, (unsigned long long)tacc[i]);
#endif
}

template <int NC, int S0, int MODE, int KS0 = 0, int KS1 = PKS>
__device__ __forceinline__ void inreg_mfma_role(const P2K &p, double *lds, int wave, int lane) {
  const int lk = lane >> 4, li = lane & 15, nt = wave >> 1;
  double TB[PKS][4];  // reference values of this lane's column node at its point of every k-step
  {
    const int n = nt * 16 + li;
#pragma unroll
    for (int ks = 0; ks < PKS; ++ks) {
      const int q = 4 * ks + lk;
      const bool ok = n < NB && q < p.nq && ks >= KS0 && ks < KS1;  // (a wave of the split variant keeps the k-steps of its half only)
#pragma unroll
      for (int i = 0; i < 4; ++i) TB[ks][i] = ok ? p.T[((i64)n * p.nq + q) * 4 + i] : 0.;
    }
    // (pinned: loads without a use stay pending for the wait-count bookkeeping)
#pragma unroll
    for (int ks = KS0; ks < KS1; ++ks)
#pragma unroll
      for (int i = (S0 == 1 ? 1 : 0); i < 4; ++i) asm volatile("" : "+v"(TB[ks][i]));
  }
  LaneK lc;
  {
    const int aslot = li >> 2;
    lc.xoff = li < RJ ? li : RJ - 1;
    if constexpr (S0 == 1) {
      const int col = aslot < 3 ? aslot : 0;  // (rows of slot 3: the table row of node 27 is zero)
#pragma unroll
      for (int j = 0; j < 3; ++j) lc.aoff[j] = 3 * j + col;
      lc.awoff = 10;
    } else {
      const bool val = aslot == 0;
      const int col = aslot == 2 ? 1 : 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) lc.aoff[j] = val ? 10 : 3 * j + col;
      lc.awoff = val ? 9 : 10;
    }
  }
  if (wave & 1) inreg_mfma_lines<NC, S0, MODE, 1, 2, 2, 2, KS0, KS1>(p, lds, TB, lc, nt, lane);
  else inreg_mfma_lines<NC, S0, MODE, 0, 3, 3, 1, KS0, KS1>(p, lds, TB, lc, nt, lane);
}

// pair slots per thread (256 threads) of node block b of a pair of planes: blocks 0..3 the even plane, 4..7 the odd one; interior row lengths are the maxima
__host__ __device__ constexpr int inreg_ub(int b, int nc) {
  constexpr int LEN[8] = {125, 75, 75, 45, 75, 45, 45, 27};
  return ((LEN[b] * nc * nc + 1) / 2 + 1 + 255) / 256;
}

// stream the finished rows of node blocks 0 .. NBLK - 1 of planes K0 (blocks 0..3) and K0 + 1 (4..7) to the value array and zero them in the buffers; thread t of 256.
// As flush_blocks, with the slot count of every block fitted to its size.
template <int NC, int NBLK, int B0 = 0>
__device__ __forceinline__ void inreg_flush(const P2K &p, double *lds, int K0, int t) {
  if (DBG(p, 128)) return;
  const int *meta = reinterpret_cast<const int *>(lds + PL<NC>::META);
  const i64 *gm = reinterpret_cast<const i64 *>(lds + PL<NC>::META + 64);
  int2 f[NBLK];
  i64 g[NBLK];
#pragma unroll
  for (int b = B0; b < NBLK; ++b) {
    const int s = ((K0 + (b >> 2)) & 7) * 4 + (b & 3);
    f[b] = *reinterpret_cast<const int2 *>(meta + 64 + s * 2);
    g[b] = gm[s];
  }
  constexpr int UBM = inreg_ub(B0, NC);
  v2d v[NBLK][UBM];
  double hv[NBLK], tv[NBLK];
  auto grab = [](double *q) { return __hip_atomic_exchange(q, 0., __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
#pragma unroll
  for (int b = B0; b < NBLK; ++b) {
    const int first = __builtin_amdgcn_readfirstlane(f[b].x), w = __builtin_amdgcn_readfirstlane(f[b].y);
    const int np = w & 0x3fffffff;
    double *lp = lds + first;
#pragma unroll
    for (int u = 0; u < UBM; ++u)
      if (u < inreg_ub(b, NC) && t + u * 256 < np) v[b][u] = v2d{grab(lp + 2 * (t + u * 256)), grab(lp + 2 * (t + u * 256) + 1)};
    if (t == 0 && (w & (1 << 30))) hv[b] = grab(lds + first - 1);
    if (t == 1 && (w < 0)) tv[b] = grab(lds + first + 2 * np);
  }
  if (DBG(p, 1)) return;
#pragma unroll
  for (int b = B0; b < NBLK; ++b) {
    const int w = __builtin_amdgcn_readfirstlane(f[b].y);
    const int np = w & 0x3fffffff, head = (w >> 30) & 1;
    const i64 goff = ((i64)__builtin_amdgcn_readfirstlane((int)(g[b] >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)g[b]);
    double *gb = p.values + goff + head;
    v2d *gp = reinterpret_cast<v2d *>(gb);
#pragma unroll
    for (int u = 0; u < UBM; ++u)
      if (u < inreg_ub(b, NC) && t + u * 256 < np) gp[t + u * 256] = v[b][u];
    if (t == 0 && head) gb[-1] = hv[b];
    if (t == 1 && (w < 0)) gb[2 * np] = tv[b];
  }
}

// waves 4..7: wave 4 + V evaluates the geometry of visit V one slice ahead; all stream the planes finished in the previous slice
template <int NC, bool SPLIT = false>
__device__ __forceinline__ void inreg_service_role(const P2K &p, double *lds, int V, int lane) {
  const int st = V * 64 + lane, nq = p.nq;
  const bool fast = p.geom.kind == NH_GEOM_ISO && p.geom.ngb == 8 && !p.geom.nograd && p.geom.bnd_axis < 0;
  const bool gok = lane < nq;
  double dN[8][3], wq = 0.;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int j = 0; j < 3; ++j) dN[a][j] = 0.;
  if (gok) {
    if (fast) {
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) dN[a][j] = p.geom.gT[((i64)a * nq + lane) * 4 + 1 + j];
    }
    wq = p.weights[lane];
  }
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(dN[a][j]));
  asm volatile("" : "+v"(wq));
  double *XV = lds + PL<NC>::XV + V * 32;  // [8 vertices][4]: staged vertex coordinates of this wave's element
#ifdef NH_ABLATION
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  const int nlines = (p.io1 - p.io0 + 1) * (p.n1 + 1);
  for (int line = blockIdx.x; line < nlines; line += gridDim.x) {
    const Line L = make_line(p, line);
    if (!L.vmask) continue;
    const LineMeta LM = make_line_meta<NC>(L);
    const bool has = (L.vmask >> V) & 1;
    const i64 ecol = ((i64)(L.io - (V >> 1)) * p.n1 + (L.jo - (V & 1))) * p.n2;
    const bool vlane = fast && has && lane >= 32 && lane < 40;
    double X0 = 0., X1 = 0., X2 = 0.;
    auto fetch_vertices = [&](int k) {
      if (vlane && k < p.n2) {
        const i64 idx = p.geom.gdofs[(ecol + k) * 8 + (lane - 32)];
        X0 = p.geom.verts[idx * 3];
        X1 = p.geom.verts[idx * 3 + 1];
        X2 = p.geom.verts[idx * 3 + 2];
      }
    };
    auto geometry = [&](int k) {  // records of the points of element (V, k)
      if (!has || DBG(p, 16)) return;
      double Ji[3][3], det;
      if (fast) {
        if (vlane) {
          double *o = XV + (lane - 32) * 4;
          *reinterpret_cast<v2d *>(o) = v2d{X0, X1};
          o[2] = X2;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (same wave writes and reads: no barrier)
        if (!gok) return;
        double J[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) J[i][j] = 0.;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const v2d x01 = *reinterpret_cast<const v2d *>(XV + a * 4);
          const double x[3] = {x01[0], x01[1], XV[a * 4 + 2]};
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) J[i][j] += x[i] * dN[a][j];
        }
        invert<3>(J, Ji, det);
      } else {
        if (!gok) return;
        geometry_at<3>(p.geom, ecol + k, lane, nq, nullptr, Ji, det, nullptr);
      }
      const double sq = sqrt(wq * fabs(det));
      double *o = lds + PR<NC>::JV + (k & 1) * RJSZ + (V * PNQ + lane) * RJ;
      *reinterpret_cast<v2d *>(o) = v2d{Ji[0][0] * sq, Ji[0][1] * sq};
      *reinterpret_cast<v2d *>(o + 2) = v2d{Ji[0][2] * sq, Ji[1][0] * sq};
      *reinterpret_cast<v2d *>(o + 4) = v2d{Ji[1][1] * sq, Ji[1][2] * sq};
      *reinterpret_cast<v2d *>(o + 6) = v2d{Ji[2][0] * sq, Ji[2][1] * sq};
      *reinterpret_cast<v2d *>(o + 8) = v2d{Ji[2][2] * sq, sq};
    };
    lds_barrier();  // (the last planes of the previous line are out: their records may go)
    if (V == 0 && lane < 12) pipe_meta_node<NC>(p, LM, lds, lane >> 2, lane & 3);
    fetch_vertices(0);
    geometry(0);
    fetch_vertices(1);  // (in flight during slice 0)
    lds_barrier();
#pragma unroll 1
    for (int k = 0; k < p.n2; ++k) {
      if (k + 1 < p.n2) {
        geometry(k + 1);
        fetch_vertices(k + 2);
        if (V == 0 && lane < 8 && !DBG(p, 64)) pipe_meta_node<NC>(p, LM, lds, 2 * k + 3 + (lane >> 2), lane & 3);
      }
      TICK(3);
      if (k > 0) {
        if constexpr (SPLIT) {  // (the twelve-wave variant: the even and the odd plane one after the other -- half the registers)
          inreg_flush<NC, 4>(p, lds, 2 * k - 2, st);
          inreg_flush<NC, 8, 4>(p, lds, 2 * k - 2, st);
        } else
          inreg_flush<NC, 8>(p, lds, 2 * k - 2, st);
      }
      TICK(4);
      lds_barrier();
      TICK(5);
    }
    if constexpr (SPLIT) {
      inreg_flush<NC, 4>(p, lds, 2 * p.n2 - 2, st);
      inreg_flush<NC, 8, 4>(p, lds, 2 * p.n2 - 2, st);
    } else
      inreg_flush<NC, 8>(p, lds, 2 * p.n2 - 2, st);
    inreg_flush<NC, 4>(p, lds, 2 * p.n2, st);
    TICK(2);
  }
#ifdef NH_ABLATION
  if (p.tdbg && lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)p.tdbg + 8 + i, (unsigned long long)tacc[i]);
#endif
}

// SPLIT: twelve waves -- the k-steps of every (unit, column tile) product are shared by TWO matrix waves per SIMD (waves w and 8 + w: k-steps [0, 4) and [4, 7)), each of
// which adds its partial Gram matrices to the row buffers: the stalls of one wave (operand reads, the LDS reduction) are covered by the other.
constexpr int KSPLIT = 4;
template <int NC, int S0, int MODE, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? NTP4 + 256 : NTP4) void k_p2hex_inreg(P2K p) {
  constexpr int NTK = SPLIT ? NTP4 + 256 : NTP4;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // row buffers are re-zeroed by the flush; the records of the pad points stay zero
  for (int i = tid; i < PR<NC>::TT; i += NTK) lds[i] = 0.;
  for (int i = tid; i < 28 * PNQ; i += NTK) {
    const int node = i / PNQ, q = i - node * PNQ;
    const bool ok = node < NB && q < p.nq;
    const double *Tp = p.T + ((i64)(ok ? node : 0) * p.nq + (ok ? q : 0)) * 4;
    double *o = lds + PR<NC>::TT + i * 4;
    *reinterpret_cast<v2d *>(o) = ok ? v2d{Tp[1], Tp[2]} : v2d{0., 0.};
    *reinterpret_cast<v2d *>(o + 2) = ok ? v2d{Tp[3], Tp[0]} : v2d{0., 0.};
  }
  for (int i = PL<NC>::META + tid; i < PL<NC>::END; i += NTK) lds[i] = 0.;
  if constexpr (SPLIT) {
    if (wave < 4) {
      inreg_mfma_role<NC, S0, MODE, 0, KSPLIT>(p, lds, wave, lane);
      return;
    }
    if (wave >= 8) {
      inreg_mfma_role<NC, S0, MODE, KSPLIT, PKS>(p, lds, wave - 8, lane);
      return;
    }
  } else if (wave < 4) {
    inreg_mfma_role<NC, S0, MODE>(p, lds, wave, lane);
    return;
  }
  if (p.prio) __builtin_amdgcn_s_setprio(3);
  inreg_service_role<NC, SPLIT>(p, lds, wave - 4, lane);
}

// closed-form CSR index arrays: one wave per node, rows (node, c) of length len * NC, columns (colnode, d) lexicographic
__global__ __launch_bounds__(256) void k_p2hex_pattern(int n0, int n1, int n2, int nc, i64 *rowptr, i64 *colidx) {
  const i64 N1 = 2 * n1 + 1, N2 = 2 * n2 + 1, nnodes = (2 * (i64)n0 + 1) * N1 * N2;
  const i64 SJ = 8 * n1 + 1, SK = 8 * n2 + 1;
  const int lane = threadIdx.x & 63;
  for (i64 node = (i64)blockIdx.x * 4 + (threadIdx.x >> 6); node < nnodes; node += (i64)gridDim.x * 4) {
    const int K = (int)(node % N2), J = (int)((node / N2) % N1), I = (int)(node / (N2 * N1));
    const int cI = ax_cnt(I, n0), cJ = ax_cnt(J, n1), cK = ax_cnt(K, n2), lI = ax_lo(I), lJ = ax_lo(J), lK = ax_lo(K);
    const int len = cI * cJ * cK;
    const i64 row0 = (i64)ax_cum(I, n0) * SJ * SK + (i64)cI * ((i64)ax_cum(J, n1) * SK + (i64)cJ * ax_cum(K, n2));
    const i64 base = row0 * nc * nc;
    if (lane < nc) rowptr[node * nc + lane] = base + (i64)lane * len * nc;
    if (node == nnodes - 1 && lane == 0) rowptr[nnodes * nc] = base + (i64)len * nc * nc;
    for (int t = lane; t < len * nc; t += 64) {
      const int cn = t / nc, d = t - cn * nc;
      const int pk = cn % cK, pj = (cn / cK) % cJ, pi = cn / (cK * cJ);
      const i64 col = (((i64)(lI + pi) * N1 + (lJ + pj)) * N2 + (lK + pk)) * nc + d;
      for (int c = 0; c < nc; ++c) colidx[base + (i64)c * len * nc + t] = col;
    }
  }
}

// UNIFORM CELLS (x = offset + scale * (multi-index + xi): mesh.rectilinear with equidistant vertices, mesh.py:45-52) and a constant form: every element matrix is the same, so
// the CSR rows of a node depend only on its CLASS per axis -- first node, odd (interior of an element), even (shared by two elements), last -- and equal the rows of the
// node of that class in a mesh of 2 x 2 x 2 such cells (5 nodes per axis: 0 first, 1 odd, 2 even, 4 last; the clipped column ranges correspond one to one).  The caller
// assembles that small mesh once with nh_p2hex_matrix; this kernel replicates its rows: a pure write stream, one workgroup per line of nodes (I, J) -- a contiguous piece of
// the value array -- with the four row blocks the line can meet staged in LDS.
constexpr int RU_NT = 256, RU_BLOCK = 125 * 9;  // doubles of the longest row block (125 coupled nodes x 3 x 3 components)
__host__ __device__ __forceinline__ int ax_cls(int X, int n) { return X == 0 ? 0 : X == 2 * n ? 4 : (X & 1) ? 1 : 2; }
__global__ __launch_bounds__(RU_NT) void k_p2hex_rows_uniform(int n0, int n1, int n2, int nc, int plane_begin, int plane_end, const double *__restrict__ cell, double *__restrict__ values) {
  __shared__ double blk[4][RU_BLOCK];
  const int N1 = 2 * n1 + 1, N2 = 2 * n2 + 1, nc2 = nc * nc;
  const i64 SJ = 8 * n1 + 1, SK = 8 * n2 + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nlines = (plane_end - plane_begin) * N1;
  for (int line = blockIdx.x; line < nlines; line += gridDim.x) {
    const int I = plane_begin + line / N1, J = line % N1;
    const int cI = ax_cnt(I, n0), cJ = ax_cnt(J, n1);
    const int sI = ax_cls(I, n0), sJ = ax_cls(J, n1);
    const i64 rowIJ = (i64)ax_cum(I, n0) * SJ * SK + (i64)cI * ((i64)ax_cum(J, n1) * SK);
    const int srowIJ = ax_cum(sI, 2) * 17 * 17 + cI * (ax_cum(sJ, 2) * 17);
    // the row blocks of the classes first / odd / even / last along the last axis (wave w stages class w)
    {
      const int sK = wave == 3 ? 4 : wave;
      const int cnt = cI * cJ * ax_cnt(sK, 2) * nc2;
      const double *src = cell + (i64)(srowIJ + cI * cJ * ax_cum(sK, 2)) * nc2;
      for (int t = lane; t < cnt; t += 64) blk[wave][t] = src[t];
    }
    __syncthreads();
    for (int K = wave; K < N2; K += RU_NT / 64) {
      const int sK = ax_cls(K, n2);
      const int cnt = cI * cJ * ax_cnt(K, n2) * nc2;
      double *dst = values + (rowIJ + (i64)cI * cJ * ax_cum(K, n2)) * nc2;
      const double *src = blk[sK == 4 ? 3 : sK];
      // 16-byte stores from the first even double on (one scalar entry in front of an odd start, one behind an odd end).  64^3 cells, 9.72 GB: 1.955 ms = 4.97 TB/s;
      // 8-byte stores 2.01 ms, nontemporal 16-byte stores 2.16 ms, nontemporal 8-byte stores 2.68 ms, 512 threads 1.99 ms
      const int head = (int)((reinterpret_cast<size_t>(dst) >> 3) & 1), npairs = (cnt - head) >> 1;
      v2d *const d2 = reinterpret_cast<v2d *>(dst + head);
      const double *const s2 = src + head;
      for (int t = lane; t < npairs; t += 64) d2[t] = v2d{s2[2 * t], s2[2 * t + 1]};
      if (lane == 0 && head) dst[0] = src[0];
      if (lane == 1 && ((cnt - head) & 1)) dst[cnt - 1] = src[cnt - 1];
    }
    __syncthreads();
  }
}

template <int NC, int S0, int MODE>
hipError_t launch_pipe(unsigned grid, size_t ldsb, hipStream_t s, const P2K &p, bool inreg) {
  auto kern = k_p2hex_pipe<NC, S0, MODE>;
  // (three components with the value slot and a dense form tensor: the in-register variant spills, the table kernel keeps it)
  int nthreads = NTP4;
  if constexpr (!(NC == 3 && S0 == 0))
    if (inreg) {
      static const bool split = getenv("NH_P2HEX_SPLIT") && atoi(getenv("NH_P2HEX_SPLIT"));
      kern = k_p2hex_inreg<NC, S0, MODE>;
      if (split) kern = k_p2hex_inreg<NC, S0, MODE, true>, nthreads = NTP4 + 256;
    }
  hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), ldsb, s, p);
  return hipSuccess;
}

}  // namespace

extern "C" {

int nh_p2hex_pattern(const int *shape, int ncomp, int64_t *rowptr_dev, int64_t *colidx_dev, void *stream) {
  NH_REQUIRE(shape && rowptr_dev && colidx_dev, "nh_p2hex_pattern: NULL argument");
  NH_REQUIRE(shape[0] >= 1 && shape[1] >= 1 && shape[2] >= 1 && ncomp >= 1 && ncomp <= 3, "nh_p2hex_pattern: shape / ncomp");
  const i64 nnodes = (2 * (i64)shape[0] + 1) * (2 * shape[1] + 1) * (2 * shape[2] + 1);
  const unsigned grid = (unsigned)std::min<i64>((nnodes + 3) / 4, 256 * 64);
  hipLaunchKernelGGL(k_p2hex_pattern, dim3(grid), dim3(256), 0, nh_stream(stream), shape[0], shape[1], shape[2], ncomp, (i64 *)rowptr_dev, (i64 *)colidx_dev);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

int nh_p2hex_matrix(const nh_p2hex_args *a, void *stream) {
  NH_REQUIRE(a, "nh_p2hex_matrix: NULL args");
  NH_REQUIRE(a->shape[0] >= 1 && a->shape[1] >= 1 && a->shape[2] >= 1, "nh_p2hex_matrix: shape");
  NH_REQUIRE(a->nq >= 1 && a->weights_dev && a->T_dev && a->values_dev && a->C_host, "nh_p2hex_matrix: NULL table / weights / values / C");
  NH_REQUIRE(a->ncomp >= 1 && a->ncomp <= 3, "nh_p2hex_matrix: ncomp must be 1..3");
  NH_REQUIRE((i64)(8 * a->shape[0] + 1) * (8 * a->shape[1] + 1) * (8 * a->shape[2] + 1) * a->ncomp * a->ncomp < ((i64)1 << 62), "nh_p2hex_matrix: size");
  if (a->nq > NQMAX) {
    nh_set_error("nh_p2hex_matrix: %d quadrature points per element: the tables exceed the LDS (at most %d)", a->nq, NQMAX);
    return NH_ELIMIT;
  }
  const nh_geometry &g = a->geom;
  if (g.kind == NH_GEOM_ISO) NH_REQUIRE(g.ngb > 0 && g.gT_dev && g.gdofs_dev && g.verts_dev, "isoparametric geometry needs ngb, gT, gdofs, verts");
  else if (g.kind == NH_GEOM_TAB) NH_REQUIRE(g.jac_dev, "tabulated geometry needs jac_dev");
  else if (g.kind == NH_GEOM_BOX) NH_REQUIRE(g.origin_dev && g.size_dev, "box geometry needs origin and size");
  else NH_REQUIRE(false, "unknown geometry kind %d", g.kind);
  NH_REQUIRE(g.bnd_axis < 0, "nh_p2hex_matrix: volume integrals only");
  const int nc = a->ncomp, S = 4;
  P2K p;
  memset(&p, 0, sizeof p);
  p.n0 = a->shape[0]; p.n1 = a->shape[1]; p.n2 = a->shape[2];
  p.nq = a->nq;
  p.ks = (a->nq + 3) / 4;
  p.l0 = a->layer_begin; p.l1 = a->layer_end;
  p.io0 = a->owner_begin; p.io1 = a->owner_end;
  NH_REQUIRE(0 <= p.l0 && p.l0 <= p.l1 && p.l1 <= p.n0, "nh_p2hex_matrix: element layers [%d, %d) outside the mesh", p.l0, p.l1);
  NH_REQUIRE(0 <= p.io0 && p.io0 <= p.io1 && p.io1 <= p.n0, "nh_p2hex_matrix: owner lines [%d, %d] outside the mesh", p.io0, p.io1);
  // active operator slots (those with a nonzero coefficient on either side) -> a contiguous window S0 .. S0 + NS - 1
  bool act[4];
  int ns = 0;
  for (int s = 0; s < S; ++s) {
    bool any = false;
    for (int c = 0; c < nc && !any; ++c)
      for (int d = 0; d < nc && !any; ++d)
        for (int b = 0; b < S && !any; ++b)
          any = a->C_host[((c * S + s) * nc + d) * S + b] != 0. || a->C_host[((c * S + b) * nc + d) * S + s] != 0.;
    act[s] = any;
    ns += any;
  }
  NH_REQUIRE(ns > 0, "nh_p2hex_matrix: zero coefficient tensor");
  const int NS = (!act[0] || !act[3]) ? 3 : 4;  // instantiated windows: {1,2,3} (gradient forms), {0,1,2}, {0,1,2,3}
  const int S0 = NS == 3 && !act[0] ? 1 : 0;
  for (int i = 0; i < NS; ++i) p.slot[i] = S0 + i;
  for (int c = 0; c < nc; ++c)
    for (int sa = 0; sa < NS; ++sa)
      for (int d = 0; d < nc; ++d)
        for (int sb = 0; sb < NS; ++sb) p.C[((c * NS + sa) * nc + d) * NS + sb] = a->C_host[((c * S + p.slot[sa]) * nc + d) * S + p.slot[sb]];
  // isotropic three-parameter family over the gradient slots?
  int mode = 0;
  if (nc == 3 && NS == 3 && S0 == 1) {
    const double lam = p.C[((0 * 3 + 0) * 3 + 1) * 3 + 1], mu2 = p.C[((0 * 3 + 1) * 3 + 1) * 3 + 0], mu = p.C[((0 * 3 + 1) * 3 + 0) * 3 + 1];
    bool iso = true;
    for (int c = 0; c < 3 && iso; ++c)
      for (int sa = 0; sa < 3 && iso; ++sa)
        for (int d = 0; d < 3 && iso; ++d)
          for (int sb = 0; sb < 3 && iso; ++sb) {
            const double expect = lam * (c == sa && d == sb) + mu * (c == d && sa == sb) + mu2 * (c == sb && sa == d);
            iso = p.C[((c * 3 + sa) * 3 + d) * 3 + sb] == expect;
          }
    if (iso) {
      mode = 1;
      p.lam = lam; p.mu = mu; p.mu2 = mu2;
    }
  }
  p.esz = 320 * nc * nc + 8;
  p.osz = 192 * nc * nc + 8;
  p.dsz = NS * p.ks * 128;
  p.weights = a->weights_dev;
  p.geom = to_k(a->geom);
  p.T = a->T_dev;
  p.values = a->values_dev;
  p.scale = a->scale_dev;
  const size_t fixed = sizeof(double) * ((size_t)3 * p.esz + 2 * p.osz) + 64 * sizeof(int) + 32 * sizeof(i64);
  const size_t lds_pipe = sizeof(double) * (nc == 1 ? PL<1>::END : nc == 2 ? PL<2>::END : PL<3>::END);
  auto lds_lock = [&](int ndb) { return fixed + sizeof(double) * ((size_t)ndb * p.dsz + 4 * p.nq * 10); };
  bool pipe = NS == 3 && p.ks == PKS && lds_pipe <= 160 * 1024;
  p.prio = 1;
  // sqrt(w |J|) on both operands of the in-register kernel: not with a signed scale array, nor with a quadrature weight <= 0 (sqrt of it: NaN in every entry of the
  // touched elements; the table kernels carry signed weights).  The caller, who owns the weight array, states the sign (nh_p2hex_args.weights_positive).
  bool inreg = pipe && !a->scale_dev && a->weights_positive;
#ifdef NH_ABLATION  // A/B switches of the ablation build only
  if (getenv("NH_P2HEX_PIPE") && atoi(getenv("NH_P2HEX_PIPE"))) inreg = false;
  if (getenv("NH_P2HEX_LOCKSTEP") && atoi(getenv("NH_P2HEX_LOCKSTEP"))) pipe = false;
  if (getenv("NH_P2HEX_PRIO")) p.prio = atoi(getenv("NH_P2HEX_PRIO"));
#endif
  p.ndb = lds_lock(2) <= 160 * 1024 ? 2 : 1;
  const size_t ldsb = pipe ? lds_pipe : lds_lock(p.ndb);
  if (ldsb > 160 * 1024) {
    nh_set_error("nh_p2hex_matrix: %zu bytes of LDS needed (nq = %d)", ldsb, a->nq);
    return NH_ELIMIT;
  }
  int dev = 0, cus = 256;
  NH_CHECK_HIP(hipGetDevice(&dev));
  NH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int nlines = (p.io1 - p.io0 + 1) * (p.n1 + 1);
  NH_REQUIRE(a->max_workgroups >= 0, "nh_p2hex_matrix: negative max_workgroups");
  const unsigned grid = (unsigned)std::min(nlines, a->max_workgroups ? std::min(cus, a->max_workgroups) : cus);
  hipStream_t s = nh_stream(stream);
#ifdef NH_ABLATION
  static long long *tdbg = nullptr;
  if (!tdbg) NH_CHECK_HIP(hipMalloc((void **)&tdbg, (16 + 1024) * sizeof(long long)));
  NH_CHECK_HIP(hipMemsetAsync(tdbg, 0, (16 + 1024) * sizeof(long long), s));
  p.tdbg = getenv("NH_P2HEX_TIMERS") ? tdbg : nullptr;
  p.debug = getenv("NH_P2HEX_DEBUG") ? atoi(getenv("NH_P2HEX_DEBUG")) : 0;
#endif
#define LAUNCH(NC, NS_, MODE)                                                                                                         \
  do {                                                                                                                                \
    if (pipe) {                                                                                                                       \
      if (S0)                                                                                                                         \
        NH_CHECK_HIP((launch_pipe<NC, 1, MODE>(grid, ldsb, s, p, inreg)));                                                                   \
      else                                                                                                                            \
        NH_CHECK_HIP((launch_pipe<NC, 0, 0>(grid, ldsb, s, p, inreg)));                                                                      \
    } else {                                                                                                                          \
      NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_p2hex<NC, NS_, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));  \
      hipLaunchKernelGGL((k_p2hex<NC, NS_, MODE>), dim3(grid), dim3(NT), ldsb, s, p);                                                 \
    }                                                                                                                                 \
  } while (0)
  const int key = nc * 100 + NS * 10 + mode;
  switch (key) {
    case 130: LAUNCH(1, 3, 0); break;
    case 140: LAUNCH(1, 4, 0); break;
    case 230: LAUNCH(2, 3, 0); break;
    case 240: LAUNCH(2, 4, 0); break;
    case 330: LAUNCH(3, 3, 0); break;
    case 331: LAUNCH(3, 3, 1); break;
    case 340: LAUNCH(3, 4, 0); break;
    default: nh_set_error("nh_p2hex_matrix: no kernel for ncomp %d, %d slots", nc, NS); return NH_ELIMIT;
  }
#undef LAUNCH
  NH_LAUNCH_CHECK();
#ifdef NH_ABLATION
  if (p.tdbg && inreg) {
    static long long h[16 + 1024];
    NH_CHECK_HIP(hipMemcpy(h, tdbg, sizeof h, hipMemcpyDeviceToHost));
    const double g = grid * 4.;
    fprintf(stderr, "p2hex_inreg cycles per wave: MFMA waves: tasks %.0f + barrier %.0f + line setup %.0f | service waves: geometry %.0f, flush %.0f + barrier %.0f, line end / start %.0f\n",
            h[0] / g, h[1] / g, h[7] / g, h[8 + 3] / g, h[8 + 4] / g, h[8 + 5] / g, h[8 + 2] / g);
  } else if (p.tdbg && pipe) {
    static long long h[16 + 1024];
    NH_CHECK_HIP(hipMemcpy(h, tdbg, sizeof h, hipMemcpyDeviceToHost));
    const double g = grid;
    if (getenv("NH_P2HEX_WGTIMES"))
      for (unsigned w = 0; w < grid && w < 1024; ++w) fprintf(stderr, "wg %u cycles %lld\n", w, h[16 + w]);
    fprintf(stderr, "p2hex flush cycles per table wave: records %.0f | lds reads %.0f | zero %.0f | stores %.0f\n", h[8] / (g * NTW), h[9] / (g * NTW), h[10] / (g * NTW), h[11] / (g * NTW));
    fprintf(stderr, "p2hex_pipe cycles per wave: MFMA waves: tasks %.0f + barrier %.0f | table waves: D %.0f, flush %.0f + barrier %.0f | geometry wave: geometry %.0f + barrier %.0f | loop head %.0f\n",
              h[0] / (g * 4), h[1] / (g * 4), h[2] / (g * NTW), h[4] / (g * NTW), h[5] / (g * NTW), h[3] / g, h[6] / g, h[7] / (g * 8));
  }
#endif
  return NH_OK;
}

int nh_p2hex_rows_uniform(const int *shape, int ncomp, const double *cell_values_dev, double *values_dev, int plane_begin, int plane_end, void *stream) {
  NH_REQUIRE(shape && cell_values_dev && values_dev, "nh_p2hex_rows_uniform: NULL argument");
  NH_REQUIRE(shape[0] >= 1 && shape[1] >= 1 && shape[2] >= 1 && ncomp >= 1 && ncomp <= 3, "nh_p2hex_rows_uniform: shape / ncomp");
  NH_REQUIRE(plane_begin >= 0 && plane_begin <= plane_end && plane_end <= 2 * shape[0] + 1, "nh_p2hex_rows_uniform: node planes %d..%d of %d", plane_begin, plane_end, 2 * shape[0] + 1);
  const i64 nlines = (i64)(plane_end - plane_begin) * (2 * shape[1] + 1);
  if (!nlines) return NH_OK;
  hipLaunchKernelGGL(k_p2hex_rows_uniform, dim3((unsigned)std::min<i64>(nlines, 256 * 16)), dim3(RU_NT), 0, nh_stream(stream), shape[0], shape[1], shape[2], ncomp, plane_begin, plane_end,
                     cell_values_dev, values_dev);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

int nh_p2hex_rowptr(const int *shape, int64_t node, int64_t *rowptr_out) {
  NH_REQUIRE(shape && rowptr_out, "nh_p2hex_rowptr: NULL argument");
  const int n0 = shape[0], n1 = shape[1], n2 = shape[2];
  const i64 N1 = 2 * n1 + 1, N2 = 2 * n2 + 1, N0 = 2 * n0 + 1;
  NH_REQUIRE(node >= 0 && node <= N0 * N1 * N2, "nh_p2hex_rowptr: node out of range");
  const i64 SJ = 8 * n1 + 1, SK = 8 * n2 + 1;
  if (node == N0 * N1 * N2) {
    *rowptr_out = (i64)(8 * n0 + 1) * SJ * SK;
    return NH_OK;
  }
  const int K = (int)(node % N2), J = (int)((node / N2) % N1), I = (int)(node / (N2 * N1));
  *rowptr_out = (i64)ax_cum(I, n0) * SJ * SK + (i64)ax_cnt(I, n0) * ((i64)ax_cum(J, n1) * SK + (i64)ax_cnt(J, n1) * ax_cum(K, n2));
  return NH_OK;
}

}  // extern "C"
