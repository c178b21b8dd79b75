// Fast path for the headline configuration: scalar Laplace stiffness, trilinear ('std'
// degree 1) basis on a structured hex mesh (mesh.rectilinear, mesh.py:34-60), 2-point
// Gauss per axis, isoparametric P1 geometry (or the uniform box geometry).
//
// WRITE-ONCE design (no global atomics, no zero-fill, CSR values stored exactly once):
//   * a persistent workgroup (one per CU) OWNS a column tile of dof rows and marches through it along axis 0 (k_p1hex_march);
//   * its threads (one per element) recompute the local matrices of all elements that touch the owned rows, i.e. a one-element
//     lateral halo is recomputed instead of communicated; the vertex tile of the NEXT step is prefetched and staged in LDS;
//   * contributions are reduced in LDS (ds_add_f64) into per-row slot arrays (slot = column offset relative to the row dof);
//   * the finished rows are streamed to HBM, coalesced, at closed-form CSR offsets: for the reference's structured dof
//     numbering the sorted-unique pattern (evaluable.py:588-616) is the tensor product of per-axis ranges
//     [max(X-1,0), min(X+1,N-1)], so rowptr(I,J,K) and the position of a column inside its row are pure arithmetic.
// HBM traffic = vertex coordinates (each tile re-reads its lateral halo) + values.  Uniform meshes take k_p1hex_uniform instead.
#include "nh_common.h"
#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace {

// ablation switches for profiling (build with -DNH_ABLATION, select with NH_P1HEX_DEBUG=bits): compiled out otherwise
#ifdef NH_ABLATION
#define DEBUG(p) ((p).debug)
#else
#define DEBUG(p) 0
#endif

struct P1Args {
  int n0, n1, n2;          // elements per axis
  int lay0, lay1;          // element layers [lay0, lay1) along axis 0 that contribute values
  int pl0, pl1;            // dof planes [pl0, pl1) along axis 0 whose rows are written
  const double *verts;     // [(n0+1)(n1+1)(n2+1)][3] or NULL (uniform: x = origin + scale*index)
  double origin[3], scale[3];
  double n[2][2];          // n[a][q] = N_a(g_q): 1-D shape functions at the 1-D Gauss points
  double c[3][2];          // c[x+y][q] = n[x][q] n[y][q]
  double wk[2][2][2];      // kappa w_qa w_qb w_qc
  double wm[2][2][2];      // mass w_qa w_qb w_qc
  int hasm;                // mass term requested (mass != 0 or a mass coefficient array)
  const double *qmass;     // NULL or [nelems][8] mass coefficient at the Gauss points
  double *values;
  const double *qscale;    // NULL or [nelems][8] coefficient at the Gauss points
  const double *u;         // vector variant: nodal values of the field the form is applied to
  double *out;             // vector variant: out[dof] (+)= sum_n K[dof][n] u[n]
  int accumulate;
  int nbj, nbk;            // column tiles per axis (j, k)
  long long *tdbg;         // phase timers (ablation builds)
  int wbnd;                // cost weight (x16) of a plane of a J-boundary column in the work split of the skewed kernel
  int debug;               // ablation switches (NH_P1HEX_DEBUG env): 1 = no LDS reduction, 2 = no HBM stores, 4 = no element math; skewed kernel: 8 = no vertex loads / staging, 16 = every second line stored only
};

__device__ __forceinline__ int len_of(int X, int N) { return (X > 0) + 1 + (X < N - 1); }       // columns coupled along one axis
__device__ __forceinline__ i64 cum_of(int X, int N) { return X == 0 ? 0 : 3 * (i64)X - 1; }     // sum_{X'<X} len_of (N >= 2)

// 1/d for d > 0 of ordinary magnitude (|det J|): hardware reciprocal estimate + two Newton steps (full f64 accuracy without the
// scaling / fix-up sequence of an IEEE division; det J is never denormal, zero or huge for a valid mesh)
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.), r, r);
  r = fma(fma(-d, r, 1.), r, r);
  return r;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release/acquire fence over ALL address
// spaces, i.e. s_waitcnt vmcnt(0): every wave would sit at the barrier until its prefetch loads have returned and its CSR stores
// have been acknowledged by L2/HBM.  The kernels below never read back what they store and consume prefetched registers behind
// the compiler's own vmcnt wait, so only the LDS accumulators need to be ordered.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct ElemTables {
  double R0[3][3], R1[3][3], R2[3][3];             // diagonal (j == k) terms, indexed by the pair classes p = a_d + b_d
  double W01[2][2][3], W02[2][2][3], W12[2][2][3];  // off-diagonal terms
  double Mm[3][3][3];                               // mass term
};

// Everything per element except the final signed sums: Jacobian columns, metric tensors at the 8 Gauss points, axis-by-axis
// contractions (see the derivation in DESIGN.md, "P1-hex local matrix by sum factorisation").
// K[a][b] for local vertices a = (a0,a1,a2), b = (b0,b1,b2): nine signed table entries.
__device__ __forceinline__ double element_entry(const ElemTables &T, int a, int bb) {
  const int a0 = a >> 2, a1 = (a >> 1) & 1, a2 = a & 1;
  const int b0 = bb >> 2, b1 = (bb >> 1) & 1, b2 = bb & 1;
  const int p0 = a0 + b0, p1 = a1 + b1, p2 = a2 + b2;
  const double s00 = (a0 == b0) ? 1. : -1., s11 = (a1 == b1) ? 1. : -1., s22 = (a2 == b2) ? 1. : -1.;
  const double s01 = (a0 == b1) ? 1. : -1., s10 = (b0 == a1) ? 1. : -1.;
  const double s02 = (a0 == b2) ? 1. : -1., s20 = (b0 == a2) ? 1. : -1.;
  const double s12 = (a1 == b2) ? 1. : -1., s21 = (b1 == a2) ? 1. : -1.;
  return s00 * T.R0[p1][p2] + s11 * T.R1[p0][p2] + s22 * T.R2[p0][p1] + s01 * T.W01[b0][a1][p2] + s10 * T.W01[a0][b1][p2] +
         s02 * T.W02[b0][a2][p1] + s20 * T.W02[a0][b2][p1] + s12 * T.W12[b1][a2][p0] + s21 * T.W12[a1][b2][p0] + T.Mm[p0][p1][p2];
}

// ---- marching kernel: a workgroup owns a COLUMN tile of (TJ-1) x (TK-1) dofs and marches along axis 0, L element layers per
// step, carrying the partially summed dof plane in LDS, so that elements are recomputed only across the two lateral tile faces
// (TJ TK / (TJ-1)(TK-1), 1.14 for 16 x 16) instead of across all six box faces (1.49 for 7^3 boxes).  The local matrix is
// symmetric and local vertex order equals global dof order, so only the 36 entries a <= b are reduced, into [rows][14] slots
// (column offset >= 0, lexicographically); a row's 13 "lower" entries are read at flush time from the upper slots of the
// neighbouring rows -- which is why the accumulated rows of a plane are ALL (TJ+1)(TK+1) tile vertices (owned dofs + a halo line
// on either side) and one extra plane is kept.
// The circular plane buffer holds L + 2 planes: the previous plane (source of the dI = -1 entries), the L planes completed by this
// step, and the partially summed plane carried to the next step.  Work is split over workgroups by (column, plane) units; every
// contiguous run of planes inside a column costs one extra element layer at its start.
template <int TJ, int TK, int L, int VPT, int VW>
__device__ __forceinline__ void load_vertex_tile(const P1Args &p, bool valid, int J0, int K0, int L0, int tid, double (&V)[VPT][VW]) {
  constexpr int NT = L * TJ * TK, VJ = TJ + 1, VK = TK + 1, NV = (L + 1) * VJ * VK;
  const int N0 = p.n0 + 1, N1 = p.n1 + 1, N2 = p.n2 + 1;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = tid + k * NT;
    const int c = v % VK, bb = (v / VK) % VJ, a = v / (VK * VJ);
    const int I = L0 + a, J = J0 - 1 + bb, K = K0 - 1 + c;
#pragma unroll
    for (int w = 0; w < VW; ++w) V[k][w] = 0.;
    if (valid && v < NV && I >= 0 && I < N0 && J >= 0 && J < N1 && K >= 0 && K < N2) {
      const i64 node = ((i64)I * N1 + J) * N2 + K;
      const double *src = p.verts + node * 3;
      V[k][0] = src[0];
      V[k][1] = src[1];
      V[k][2] = src[2];
      if constexpr (VW == 4) V[k][3] = p.u[node];
    }
  }
}

// VEC: instead of the matrix, out (+)= K u is assembled -- the same element matrices applied to the nodal values u on the fly and
// reduced into ONE slot per row (residual of the same form: evaluable.py:3405-3411 Inflate + add.at in the reference).
template <int TJ, int TK, int L, bool VEC, bool MASS, bool COEF>
__global__ __launch_bounds__(L * TJ * TK) void k_p1hex_march(P1Args p) {
#ifdef NH_ABLATION
  long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
#define NH_TICK(i) { const long long tnow = clock64(); tacc[i] += tnow - tprev; tprev = tnow; }
#else
#define NH_TICK(i)
#endif
  constexpr int NT = L * TJ * TK, NP = L + 2, OJ = TJ - 1, OK = TK - 1;
  constexpr int NS = VEC ? 1 : 15;  // 14 slots + 1 pad: an odd row stride (in doubles) spreads the 64 lanes of a ds_add_f64 over all LDS banks
  constexpr int VW = VEC ? 4 : 3;   // doubles per staged vertex: coordinates (+ nodal value)
  constexpr int VJ = TJ + 1, VK = TK + 1, RP = VJ * VK, PS = (RP * NS + 1) & ~1, NV = (L + 1) * VJ * VK, VPT = (NV + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *acc = lds;             // [NP][RP][NS], plane stride PS
  double *vbuf = lds + NP * PS;  // [L+1][VJ][VK][VW]
  const int tid = threadIdx.x;
  const int N0 = p.n0 + 1, N1 = p.n1 + 1, N2 = p.n2 + 1;
  const int NPL = p.pl1 - p.pl0;
  const i64 U = (i64)p.nbj * p.nbk * NPL;
  // XCD-aware work order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so workgroup b runs on XCD b % 8;
  // give the workgroups of one XCD a CONTIGUOUS range of (column, plane) units -- neighbours in the mesh share vertex planes and the
  // partially written cache lines at the tile edges in the same L2
  const unsigned wg = gridDim.x % 8 == 0 ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
  i64 u = U * wg / gridDim.x;
  const i64 u1 = U * (wg + 1) / gridDim.x;
  if (u >= u1) return;
  auto slot_of = [](int P) { return (int)((unsigned)(P + 2 * NP) % NP) * PS; };  // P >= -2
  // current run: planes [A, B) of column (cj, ck); current step: element layers [L0, L0 + L)
  int col = (int)(u / NPL), A = p.pl0 + (int)(u % NPL), B = (int)min((i64)p.pl1, A + (u1 - u)), L0 = A - 1;
  double V[VPT][VW];
  load_vertex_tile<TJ, TK, L, VPT, VW>(p, true, (col / p.nbk) * OJ, (col % p.nbk) * OK, L0, tid, V);
  for (int t = tid; t < NP * PS / 2; t += NT) reinterpret_cast<double2 *>(acc)[t] = make_double2(0., 0.);
  auto stage_vertex_tile = [&]() {
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const int v = tid + k * NT;
      if (v < NV) {
#pragma unroll
        for (int w = 0; w < VW; ++w) vbuf[v * VW + w] = V[k][w];
      }
    }
  };
  stage_vertex_tile();

  for (;;) {
    const int J0 = (col / p.nbk) * OJ, K0 = (col % p.nbk) * OK;
    // next step (or next run): its vertex loads are issued at the head of the element phase and land in LDS right after it
    int ncol = col, nA = A, nB = B, nL0 = L0 + L;
    bool last = false, fresh = false;
    if (nL0 >= B) {
      u += B - A;
      if (u >= u1) last = true;
      else {
        ncol = (int)(u / NPL), nA = p.pl0 + (int)(u % NPL), nB = (int)min((i64)p.pl1, nA + (u1 - u)), nL0 = nA - 1;
        fresh = true;
      }
    }
    NH_TICK(0)
    lds_barrier();  // vertex tile staged, plane slots zeroed
    NH_TICK(1)
    load_vertex_tile<TJ, TK, L, VPT, VW>(p, !last, (ncol / p.nbk) * OJ, (ncol % p.nbk) * OK, nL0, tid, V);

    {
      const int lay = tid / (TJ * TK), el = tid % (TJ * TK), ej = el / TK, ek = el % TK;
      const int gi = L0 + lay, gj = J0 - 1 + ej, gk = K0 - 1 + ek;
      if (gi >= p.lay0 && gi < p.lay1 && gj >= 0 && gj < p.n1 && gk >= 0 && gk < p.n2 && !(DEBUG(p) & 4)) {
#define NH_P1HEX_VERT(a, bb, c) (vbuf + (((lay + (a)) * VJ + (ej + (bb))) * VK + (ek + (c))) * VW)
#include "nh_p1hex_element.inc"
#undef NH_P1HEX_VERT
      }
    }
    NH_TICK(2)
    lds_barrier();  // all contributions of this step are in LDS, all vertex reads done
    NH_TICK(3)
    if (!last) stage_vertex_tile();

    // ---- stream the completed planes to HBM ---------------------------------------------------------------------------------
    const i64 T1 = 3 * (i64)N1 - 2, T2 = 3 * (i64)N2 - 2;
    const int Pb = max(L0, A), Pe = (DEBUG(p) & 2) ? Pb : min(L0 + L, B);
    if constexpr (VEC) {
      // one value per owned row: thread t -> (plane, oj, ok)
      for (int t = tid; t < L * OJ * OK; t += NT) {
        const int ok = t % OK, oj = (t / OK) % OJ, P = Pb + t / (OK * OJ);
        const int J = J0 + oj, Kk = K0 + ok;
        if (P < Pe && J < N1 && Kk < N2) {
          double *dst = p.out + ((i64)P * N1 + J) * N2 + Kk;
          const double r = acc[slot_of(P) + (oj + 1) * VK + (ok + 1)];
          *dst = p.accumulate ? *dst + r : r;
        }
      }
    } else {
      // 32 lanes per row (27 slots), one K line (or part of it) per pass; fully interior steps take the 16-byte path further down
      constexpr int RPP = NT / 32, KP = (OK + RPP - 1) / RPP;  // rows per flush pass, passes per K line
      const int sl = tid & 31, rsub = tid >> 5;
      const int dI = sl / 9 - 1, dJ = (sl / 3) % 3 - 1, dK = sl % 3 - 1;
      const bool upper = sl >= 13;
      const int need = (dI < 0 ? 1 : dI > 0 ? 8 : 0) | (dJ < 0 ? 2 : dJ > 0 ? 16 : 0) | (dK < 0 ? 4 : dK > 0 ? 32 : 0);
      constexpr int C = OJ % 5 == 0 ? 5 : OJ % 7 == 0 ? 7 : OJ % 3 == 0 ? 3 : 1;  // LDS reads in flight per lane
      const int cumJ0 = J0 == 0 ? 0 : 3 * J0 - 1;
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
        const int ok = kp * RPP + rsub, Kk = K0 + ok;
        const bool kact = sl < 27 && ok < OK && Kk < N2;
        const int loK = Kk > 0, hiK = Kk < N2 - 1, lenK = loK + 1 + hiK, cumK = Kk == 0 ? 0 : 3 * Kk - 1;
        // LDS source relative to (plane, line vj = 1): own upper slot, or the mirrored upper slot of the neighbouring row
        const int srow = (VK + ok + 1) * NS + (upper ? sl - 13 : (dJ * VK + dK) * NS + 13 - sl);
        const bool lowJ = J0 == 0, highJ = J0 + OJ >= N1;
        // one plane loop per tile kind (the kind is decided OUTSIDE the loop: otherwise the loop invariants of the rarer kinds are
        // hoisted in front of the common one and executed on every step)
        auto planes = [&](auto kind) {
        for (int P = Pb; P < Pe; ++P) {
          const int loI = P > 0, hiI = P < N0 - 1, lenI = loI + 1 + hiI;
          const double *src = acc + ((!upper && dI < 0) ? slot_of(P - 1) : slot_of(P)) + srow;
          // CSR offset of row (P, J, 0): uniform, advanced line by line
          double *line = p.values + ((P == 0 ? 0 : 3 * (i64)P - 1) * T1 * T2 + lenI * (cumJ0 * T2));
          if constexpr (decltype(kind)::value == 0) {
            // all OJ lines exist and are interior along J (7 of 9 tiles at 128^3): lane offset and mask are loop invariants and the
            // whole pass is ONE exec region of straight-line code -- per store one global_store (uniform line pointer + 32-bit lane
            // byte offset) and one scalar pointer bump.  In this phase nothing else runs on the CU, so every instruction around
            // a store is exposed: the select-per-line variant below costs 0.05 ms more when used for all tiles.
            const int flag = loI | 2 | loK << 2 | hiI << 3 | 16 | hiK << 5;
            const unsigned voff8 = 8u * (unsigned)(lenI * 3 * cumK + ((dI + loI) * 3 + (dJ + 1)) * lenK + (dK + loK));
            const i64 stride8 = 8 * (i64)(lenI * 3 * (int)T2);
            if (kact && (flag & need) == need) {
              double v[OJ];  // all LDS reads of the pass in flight before the first store
#pragma unroll
              for (int oj = 0; oj < OJ; ++oj) v[oj] = src[oj * (VK * NS)];
              char *lp = reinterpret_cast<char *>(line);
#pragma unroll
              for (int oj = 0; oj < OJ; ++oj) {
                *reinterpret_cast<double *>(lp + voff8) = v[oj];
                lp += stride8;
              }
            }
          } else if constexpr (decltype(kind)::value == 1) {
            // straight-line: lane offset and mask are loop invariants, with a second set for the ONE line of a tile that can
            // touch the J boundary (J = 0 in the first tile, J = N1 - 1 in the last; lines beyond it do not exist)
            const int jb = lowJ ? 0 : highJ ? N1 - 1 - J0 : -1, nlines = highJ ? N1 - J0 : OJ;
            const int flag3 = loI | 2 | loK << 2 | hiI << 3 | 16 | hiK << 5;
            const unsigned voff3 = (unsigned)(lenI * 3 * cumK + ((dI + loI) * 3 + (dJ + 1)) * lenK + (dK + loK));
            const bool act3 = kact && (flag3 & need) == need;
            const int loJb = lowJ ? 0 : 1, flagb = loI | loJb << 1 | loK << 2 | hiI << 3 | (lowJ ? 16 : 0) | hiK << 5;
            const unsigned voffb = (unsigned)(lenI * 2 * cumK + ((dI + loI) * 2 + (dJ + loJb)) * lenK + (dK + loK));
            const bool actb = kact && (flagb & need) == need;
            double v[OJ];  // all LDS reads of the pass in flight before the first store
#pragma unroll
            for (int oj = 0; oj < OJ; ++oj) v[oj] = src[oj * (VK * NS)];
#pragma unroll
            for (int oj = 0; oj < OJ; ++oj) {
              if (oj < nlines) {  // uniform
                const bool bnd = oj == jb;
                if (bnd ? actb : act3) line[bnd ? voffb : voff3] = v[oj];
                line += lenI * (bnd ? 2 : 3) * (int)T2;
              }
            }
          } else {
            // meshes with a single tile along J: per-line flags recomputed on the scalar unit
#pragma unroll 1
            for (int oj = 0; oj < OJ; oj += C) {
              double v[C];
#pragma unroll
              for (int q = 0; q < C; ++q) v[q] = src[(oj + q) * (VK * NS)];
#pragma unroll
              for (int q = 0; q < C; ++q) {
                const int J = J0 + oj + q;
                const int loJ = J > 0, hiJ = J < N1 - 1, lenJ = loJ + 1 + hiJ;
                const int flag = loI | loJ << 1 | loK << 2 | hiI << 3 | hiJ << 4 | hiK << 5;
                const unsigned voff = (unsigned)(lenI * lenJ * cumK + ((dI + loI) * lenJ + (dJ + loJ)) * lenK + (dK + loK));
                if (kact && J < N1 && (flag & need) == need) line[voff] = v[q];
                line += lenI * lenJ * (int)T2;
              }
            }
          }
        }
        };
        if (L == 2 && NT == 512 && KP == 1 && !lowJ && !highJ && Pe - Pb == 2 && Pb > 0 && Pe < N0 && K0 > 0 && K0 + OK < N2) {
          // fully interior step (59 % of the steps at 128^3): all rows have 27 entries, a K line is 405 contiguous doubles.  Half a
          // workgroup takes one line, TWO CONSECUTIVE ENTRIES PER LANE: the vector-memory issue path of a CU carries address + data of
          // one 64-lane store per ~16 cycles whatever its width, so 16-byte stores halve the instruction count of this phase.
          const int l = tid & 255, half = __builtin_amdgcn_readfirstlane(tid >> 8), e = 2 * l;
          auto source = [&](int ee, bool &below) {  // LDS position of entry ee of the line, relative to (plane, first owned line)
            const int ok_ = ee / 27, sl_ = ee - ok_ * 27;
            const int dJ_ = (sl_ / 3) % 3 - 1, dK_ = sl_ % 3 - 1;
            below = sl_ < 9;
            return (VK + ok_ + 1) * NS + (sl_ >= 13 ? sl_ - 13 : (dJ_ * VK + dK_) * NS + 13 - sl_);
          };
          bool belowA, belowB;
          const int offA = source(e < 405 ? e : 404, belowA) + half * (VK * NS), offB = source(e + 1 < 405 ? e + 1 : 404, belowB) + half * (VK * NS);
          double a0[2][8], a1[2][8];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const double *sA = acc + (belowA ? slot_of(Pb + pl - 1) : slot_of(Pb + pl)) + offA;
            const double *sB = acc + (belowB ? slot_of(Pb + pl - 1) : slot_of(Pb + pl)) + offB;
#pragma unroll
            for (int i = 0; i < 8; ++i) {  // (line 15 of the odd half lies in the halo: read, not stored)
              a0[pl][i] = sA[i * (2 * VK * NS)];
              a1[pl][i] = sB[i * (2 * VK * NS)];
            }
          }
          const i64 stride8 = 8 * (i64)(9 * (int)T2);
          char *const base = reinterpret_cast<char *>(p.values + ((3 * (i64)Pb - 1) * T1 * T2 + 3 * (cumJ0 * T2) + 9 * (3 * (i64)K0 - 1))) + half * stride8;
          const int nl = half ? 7 : 8;
          if (l < 202) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
              char *lp = base + pl * (8 * (3 * T1 * T2)) + 16 * l;
#pragma unroll
              for (int i = 0; i < 8; ++i)
                if (i < nl) {
                  const double2 v = make_double2(a0[pl][i], a1[pl][i]);
                  __builtin_memcpy(lp + i * (2 * stride8), &v, 16);  // 8-byte aligned 16-byte store
                }
            }
          } else if (l == 202) {  // 405 is odd: the last entry of a line goes alone
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
              char *lp = base + pl * (8 * (3 * T1 * T2)) + 16 * l;
#pragma unroll
              for (int i = 0; i < 8; ++i)
                if (i < nl) *reinterpret_cast<double *>(lp + i * (2 * stride8)) = a0[pl][i];
            }
          }
        } else if (L == 2 && !lowJ && !highJ && Pe - Pb == 2 && Pb > 0 && Pe < N0) {
          // the common step: two planes, nothing on a boundary except possibly K -- one set of lane constants, ONE exec region, the 30
          // LDS reads of both planes in flight before the first of the 30 stores
          const int flag = 1 | 2 | loK << 2 | 8 | 16 | hiK << 5;
          const unsigned voff8 = 8u * (unsigned)(9 * cumK + ((dI + 1) * 3 + (dJ + 1)) * lenK + (dK + loK));
          const i64 stride8 = 8 * (i64)(9 * (int)T2);
          if (kact && (flag & need) == need) {
            const bool below = !upper && dI < 0;
            const double *src0 = acc + (below ? slot_of(Pb - 1) : slot_of(Pb)) + srow, *src1 = acc + (below ? slot_of(Pb) : slot_of(Pb + 1)) + srow;
            double v0[OJ], v1[OJ];
#pragma unroll
            for (int oj = 0; oj < OJ; ++oj) v0[oj] = src0[oj * (VK * NS)];
#pragma unroll
            for (int oj = 0; oj < OJ; ++oj) v1[oj] = src1[oj * (VK * NS)];
            char *lp0 = reinterpret_cast<char *>(p.values + ((3 * (i64)Pb - 1) * T1 * T2 + 3 * (cumJ0 * T2)));
            char *lp1 = lp0 + 8 * (3 * T1 * T2);
#pragma unroll
            for (int oj = 0; oj < OJ; ++oj) {
              *reinterpret_cast<double *>(lp0 + voff8) = v0[oj];
              lp0 += stride8;
            }
#pragma unroll
            for (int oj = 0; oj < OJ; ++oj) {
              *reinterpret_cast<double *>(lp1 + voff8) = v1[oj];
              lp1 += stride8;
            }
          }
        } else if (!lowJ && !highJ) planes(std::integral_constant<int, 0>{});
        else if (!(lowJ && highJ)) planes(std::integral_constant<int, 1>{});
        else planes(std::integral_constant<int, 2>{});
      }
    }
    NH_TICK(4)
    if (last) break;
    lds_barrier();  // flush reads done: recycle plane slots
    NH_TICK(5)
    if (fresh) {
      for (int t = tid; t < NP * PS / 2; t += NT) reinterpret_cast<double2 *>(acc)[t] = make_double2(0., 0.);
    } else {
#pragma unroll
      for (int P = L0 - 1; P < L0 + L - 1; ++P) {
        double2 *z = reinterpret_cast<double2 *>(acc + slot_of(P));
        for (int t = tid; t < PS / 2; t += NT) z[t] = make_double2(0., 0.);
      }
    }
    col = ncol, A = nA, B = nB, L0 = nL0;
    NH_TICK(6)
  }
#ifdef NH_ABLATION
  if (p.tdbg && (tid & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)p.tdbg + i, (unsigned long long)tacc[i]);
#endif
}
#undef NH_TICK

// ---- skewed marching kernel (matrix): the two halves of the workgroup alternate roles every slot ----------------------------------
// In the kernel above all waves compute (f64 VALU saturated, memory idle) and then all waves flush (memory busy, VALU idle), and the
// kernel time is the sum of the two phases.  Here a slot is ONE element layer: half A (4 waves, one per SIMD) computes layer s while
// half B streams the plane completed one slot earlier to HBM, recycles a plane slot and stages the vertex plane of the next layer;
// in the next slot the roles are swapped.  Plane P is complete after layers P-1 and P and is flushed in slot P+1 by the half that
// computed layer P; the flush reads planes P-1 and P, the concurrent arithmetic writes planes P+1 and P+2: four plane slots, no
// conflict.  One workgroup barrier per slot; the flushing half needs one barrier of its own (all its reads of plane P-2 before the
// slot is zeroed): an LDS counter, since s_barrier cannot address half a workgroup.
// The gain is modest (-3.3 % in steady-state A/B runs at 128^3): a lone wave per SIMD sustains only 65 % of the f64 rate of two, so the
// arithmetic of a layer takes 7.9 k cycles instead of 5; the stores of a plane take ~7.6 k at the chip-wide write rate, plus ~2.7 k
// until they are acknowledged (the wait for the vertex loads of the same waves is a wait for everything before it: one in-order
// counter).  profiles/r01_e_skewed_marching_kernel.md lists the variants that were measured slower (fixed roles with LDS-counter hand-offs, a fourth wave
// that only recycles and stages, vertex loads from the computing half, ...).
__device__ __forceinline__ void half_arrive(unsigned *cnt) {  // this wave's LDS reads have returned
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) atomicAdd(cnt, 1u);
}
__device__ __forceinline__ void half_wait(unsigned *cnt, unsigned target) {
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

#ifdef NH_P1HEX_WPE  // experiment: tell the scheduler that two waves per SIMD is all there will ever be
#define NH_WPE __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define NH_WPE
#endif
template <int TJ, int TK, bool MASS, bool COEF>
__global__ __launch_bounds__(2 * TJ * TK) NH_WPE void k_p1hex_skew(P1Args p) {
  constexpr bool VEC = false;
#ifdef NH_ABLATION
  long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
  const long long tstart = tprev;
  int nslow = 0, nfast = 0;
#ifdef NH_NOTICKS
#define NH_TICK(i)
#else
#define NH_TICK(i) { const long long tnow = clock64(); tacc[i] += tnow - tprev; tprev = tnow; }
#endif
#else
#define NH_TICK(i)
#endif
  constexpr int G = TJ * TK, NT = 2 * G, NP = 4, OJ = TJ - 1, OK = TK - 1, NS = 15, VW = 3;
  constexpr int VJ = TJ + 1, VK = TK + 1, RP = VJ * VK, PS = (RP * NS + 1) & ~1, VPG = (RP + G - 1) / G;
  const bool getenv_stage_late = p.wbnd < 0;  // (A/B switch of the launcher: negative weight = stage behind the stores as before)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *acc = lds;                                               // [NP][RP][NS], plane stride PS
  double *vbuf = lds + NP * PS;                                    // [3][VJ][VK][VW]
  unsigned *hcnt = reinterpret_cast<unsigned *>(vbuf + 3 * RP * VW);  // arrivals at the half barrier
  const int tid = threadIdx.x, lt = tid & (G - 1), grp = __builtin_amdgcn_readfirstlane(tid / G);
  const int N0 = p.n0 + 1, N1 = p.n1 + 1, N2 = p.n2 + 1;
  const int NPL = p.pl1 - p.pl0;
  const unsigned wg = gridDim.x % 8 == 0 ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;  // XCD-aware, as above
  // Work split in SLOTS, not planes: a run of n planes costs n + 2 slots (the first has nothing to flush, the last nothing to compute), so a
  // workgroup whose share crosses a column boundary gets two planes less -- every column is given NPL + 2 cost units, one in front of
  // its first plane and one behind its last, and the workgroups share the cost axis evenly
  // Columns on the J boundary (first and last tile row) flush through the general path (rows of different lengths, 8-byte stores) and take
  // ~1.3x as long per plane: their cost units are weighted p.wbnd / 16 so that all workgroups finish together
  const i64 CP = NPL + 2, WI = 16, WB = p.nbj > 1 ? (p.wbnd < 0 ? -p.wbnd : p.wbnd) : 16;
  const i64 rowB = p.nbk * CP * WB, rowI = p.nbk * CP * WI, nmid = max(p.nbj - 2, 0);
  const i64 ctot = (p.nbj > 1 ? 2 : 1) * rowB + nmid * rowI;
  auto unit_at = [&](i64 c) {
    i64 jt, rem, w;
    if (c < rowB) jt = 0, rem = c, w = WB;
    else if (c < rowB + nmid * rowI) jt = 1 + (c - rowB) / rowI, rem = (c - rowB) % rowI, w = WI;
    else jt = p.nbj > 1 ? p.nbj - 1 : 1, rem = c - rowB - nmid * rowI, w = WB;  // (c == ctot: one past the last column)
    const i64 kt = rem / (CP * w), pos = (rem - kt * CP * w) / w;
    return (jt * p.nbk + kt) * NPL + min(max(pos - 1, (i64)0), (i64)NPL);
  };
  i64 u = unit_at(ctot * wg / gridDim.x);
  const i64 u1 = unit_at(ctot * (wg + 1) / gridDim.x);
  if (u >= u1) return;
  auto slot_of = [](int P) { return (int)((unsigned)(P + 2 * NP) % NP) * PS; };  // P >= -2 * NP
  auto vslot_of = [](int P) { return (int)((unsigned)(P + 3) % 3) * (RP * VW); };  // P >= -3
  auto load_vertex = [&](int I, int r, int J0, int K0, double (&x)[3]) {
    const int bb = r / VK, c = r % VK, J = J0 - 1 + bb, K = K0 - 1 + c;
    x[0] = x[1] = x[2] = 0.;
    if (r < RP && I >= 0 && I < N0 && J >= 0 && J < N1 && K >= 0 && K < N2) {
      const double *src = p.verts + (((i64)I * N1 + J) * N2 + K) * 3;
      x[0] = src[0], x[1] = src[1], x[2] = src[2];
    }
  };
  if (tid == 0) *hcnt = 0;
  unsigned htarget = 0;
  const i64 T1 = 3 * (i64)N1 - 2, T2 = 3 * (i64)N2 - 2;
  while (u < u1) {
    // run: planes [A, B) of column col; element layers A-1 .. B-1 in slots A-1 .. B-1, plane P flushed in slot P+1
    const int col = (int)(u / NPL), A = p.pl0 + (int)(u % NPL), B = (int)min((i64)p.pl1, A + (u1 - u));
    const int J0 = (col / p.nbk) * OJ, K0 = (col % p.nbk) * OK;
    {
      // vertex planes A-1 and A: unconditional loads from clamped addresses (predicated ones cost one memory round trip EACH: the compiler
      // waits at every join), issued in front of the zeroing of the plane slots
      constexpr int NVL = (2 * RP + NT - 1) / NT;
      double x[NVL][3];
      bool valid[NVL];
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        const int v = min(tid + k * NT, 2 * RP - 1), pl = v / RP, r = v % RP, I = A - 1 + pl;
        const int J = J0 - 1 + r / VK, K = K0 - 1 + r % VK;
        valid[k] = I >= 0 && I < N0 && J >= 0 && J < N1 && K >= 0 && K < N2;
        const double *src = p.verts + (((i64)min(max(I, 0), N0 - 1) * N1 + min(max(J, 0), N1 - 1)) * N2 + min(max(K, 0), N2 - 1)) * 3;
        x[k][0] = src[0], x[k][1] = src[1], x[k][2] = src[2];
      }
      for (int t = tid; t < NP * PS / 2; t += NT) reinterpret_cast<double2 *>(acc)[t] = make_double2(0., 0.);
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        const int v = tid + k * NT, pl = v / RP, r = v % RP;
        if (v < 2 * RP) {
          double *dst = vbuf + vslot_of(A - 1 + pl) + r * VW;
          dst[0] = valid[k] ? x[k][0] : 0., dst[1] = valid[k] ? x[k][1] : 0., dst[2] = valid[k] ? x[k][2] : 0.;
        }
      }
    }
    lds_barrier();

    for (int s = A - 1; s <= B; ++s) {
      htarget += G / 64;
      NH_TICK(0)
      const bool mathrole = ((s - (A - 1)) & 1) == grp;
#ifdef NH_P1HEX_PRIO  // experiment: issue priority by role (1 = arithmetic wave first, 2 = memory wave first)
      if (mathrole == (NH_P1HEX_PRIO == 1)) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
#endif
      if (mathrole) {
        // ---- arithmetic role: element layer s -------------------------------------------------------------------------------
        const int ej = lt / TK, ek = lt % TK;
        const int gi = s, gj = J0 - 1 + ej, gk = K0 - 1 + ek;
        if (s < B && gi >= p.lay0 && gi < p.lay1 && gj >= 0 && gj < p.n1 && gk >= 0 && gk < p.n2 && !(DEBUG(p) & 4)) {
#define NH_P1HEX_VERT(a, bb, c) (vbuf + vslot_of(gi + (a)) + ((ej + (bb)) * VK + (ek + (c))) * VW)
#include "nh_p1hex_element.inc"
#undef NH_P1HEX_VERT
        }
        NH_TICK(3)
      } else {
        // ---- memory role: vertex plane s+2 (for layer s+1, next slot), flush of plane s-1, recycling of the slot of plane s-2 ------
        const bool needv = s + 1 < B && !(DEBUG(p) & 8);  // vertex plane s+2 for layer s+1 (next slot): loaded in front of the stores, staged behind them
        double Vn[VPG][3];
#pragma unroll
        for (int k = 0; k < VPG; ++k) load_vertex(needv ? s + 2 : -1, lt + k * G, J0, K0, Vn[k]);
        const int P = s - 1;
        bool arrived = false, staged = false, recycled = false;
        auto stage_vertices = [&]() {
#pragma unroll
          for (int k = 0; k < VPG; ++k) {
            const int r = lt + k * G;
            if (r < RP) {
              double *dst = vbuf + vslot_of(s + 2) + r * VW;
              dst[0] = Vn[k][0], dst[1] = Vn[k][1], dst[2] = Vn[k][2];
            }
          }
        };
        if (P >= A && !(DEBUG(p) & 2)) {
          const bool lowJ = J0 == 0, highJ = J0 + OJ >= N1;
          const int cumJ0 = J0 == 0 ? 0 : 3 * J0 - 1;
#ifdef NH_ABLATION
          if (!lowJ && !highJ && P > 0 && P < N0 - 1 && K0 > 0 && K0 + OK < N2) ++nfast; else ++nslow;
#endif
          if (!lowJ && !highJ && P > 0 && P < N0 - 1 && K0 > 0 && K0 + OK < N2) {
            // fully interior plane: every row has 27 entries, a K line is 405 contiguous doubles; two consecutive entries per lane
            // (16-byte stores), one line per pass
            const int e = 2 * lt;
            auto source = [&](int ee, bool &below) {  // LDS position of entry ee of a line, relative to (plane, first owned line)
              const int ok_ = ee / 27, sl_ = ee - ok_ * 27;
              const int dJ_ = (sl_ / 3) % 3 - 1, dK_ = sl_ % 3 - 1;
              below = sl_ < 9;
              return (VK + ok_ + 1) * NS + (sl_ >= 13 ? sl_ - 13 : (dJ_ * VK + dK_) * NS + 13 - sl_);
            };
#ifdef NH_P1HEX_PIPEFLUSH
            // Per-LINE software pipeline: the LDS reads of line i + D are in flight while line i is stored, so the first store leaves ~1.5 k cycles
            // earlier and the reads of the plane hide behind the (throughput-bound) stores instead of preceding them.  No exec region around the
            // stores: the lanes beyond the 202 pairs of a line repeat lane 201 (same address, same data) -- a branch here would make the
            // compiler's wait for the vertex loads BEHIND the stores a vmcnt(0), i.e. a wait for the whole flush to drain (the join of the skipped
            // path has the loads as its youngest operations); straight-line, it is vmcnt(stores issued since), which only needs the loads.
            constexpr int D = NH_P1HEX_PIPEFLUSH;
            const int ec = 2 * min(lt, 201);
            bool belowA, belowB;
            const int offA = source(ec, belowA), offB = source(ec + 1, belowB);
            const double *sA = acc + (belowA ? slot_of(P - 1) : slot_of(P)) + offA;
            const double *sB = acc + (belowB ? slot_of(P - 1) : slot_of(P)) + offB;
            const int li = lt - 202;
            const bool lone = li >= 0 && li < OJ;
            bool belowL;
            const int offL = source(404, belowL);
            const double aL = acc[(belowL ? slot_of(P - 1) : slot_of(P)) + offL + (lone ? li : 0) * (VK * NS)];
            const i64 stride8 = 8 * (i64)(9 * (int)T2);
            char *l0 = reinterpret_cast<char *>(p.values + ((3 * (i64)P - 1) * T1 * T2 + 3 * (cumJ0 * T2) + 9 * (3 * (i64)K0 - 1)));
            char *lp = l0 + 8 * ec;
            double a0[OJ], a1[OJ];
#pragma unroll
            for (int i = 0; i < D && i < OJ; ++i) a0[i] = sA[i * (VK * NS)], a1[i] = sB[i * (VK * NS)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < OJ; ++i) {
              if (i + D < OJ) a0[i + D] = sA[(i + D) * (VK * NS)], a1[i + D] = sB[(i + D) * (VK * NS)];
              __builtin_amdgcn_sched_barrier(0);
              const double2 v = make_double2(a0[i], a1[i]);
              __builtin_memcpy(lp + i * stride8, &v, 16);
              __builtin_amdgcn_sched_barrier(0);
            }
            half_arrive(hcnt);
            arrived = true;
            if (needv) {
              stage_vertices();
              staged = true;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (lone) *reinterpret_cast<double *>(l0 + li * stride8 + 8 * 404) = aL;
#else
            bool belowA, belowB;
            const int offA = source(e < 405 ? e : 404, belowA), offB = source(e + 1 < 405 ? e + 1 : 404, belowB);
            const double *sA = acc + (belowA ? slot_of(P - 1) : slot_of(P)) + offA;
            const double *sB = acc + (belowB ? slot_of(P - 1) : slot_of(P)) + offB;
            double a0[OJ], a1[OJ];
#pragma unroll
            for (int i = 0; i < OJ; ++i) {
              a0[i] = sA[i * (VK * NS)];
              a1[i] = sB[i * (VK * NS)];
            }
            // 405 is odd: the last entries of the OJ lines go out in ONE more store of the last wave (lane 202 + i: line i)
            const int li = lt - 202;
            const bool lone = li >= 0 && li < OJ;
            bool belowL;
            const int offL = source(404, belowL);
            const double aL = acc[(belowL ? slot_of(P - 1) : slot_of(P)) + offL + (lone ? li : 0) * (VK * NS)];
            half_arrive(hcnt);
            arrived = true;
            NH_TICK(8)
            // the vertex plane of the next layer is staged HERE, in front of the stores: its loads were issued in front of the LDS reads and have
            // returned by now, while behind the stores the (in-order) wait for them would be a wait for the whole flush to drain
            if (needv && !getenv_stage_late) {
              __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
              stage_vertices();
              staged = true;
            }
            const i64 stride8 = 8 * (i64)(9 * (int)T2);
            char *l0 = reinterpret_cast<char *>(p.values + ((3 * (i64)P - 1) * T1 * T2 + 3 * (cumJ0 * T2) + 9 * (3 * (i64)K0 - 1)));
            char *lp = l0 + 16 * lt;
#ifdef NH_P1HEX_INTERLEAVE
            // the recycling of the slot of plane s-2 BETWEEN the stores: a store that finds the queue of the memory pipeline full stalls the wave, and the
            // LDS writes placed behind all of them waited for the whole flush; one write behind each store goes out while the queue drains
            half_wait(hcnt, htarget);
            recycled = true;
            {
              double2 *z = reinterpret_cast<double2 *>(acc + slot_of(s - 2));
              constexpr int NZ = (PS / 2 + G - 1) / G;
              static_assert(NZ <= OJ, "one zeroing write per store");
#pragma unroll
              for (int i = 0; i < OJ; ++i) {
                if (lt < 202) {
                  const double2 v = make_double2(a0[i], a1[i]);
                  __builtin_memcpy(lp + i * stride8, &v, 16);
                }
                if (i < NZ && lt + i * G < PS / 2) z[lt + i * G] = make_double2(0., 0.);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
#else
            if (lt < 202) {
#pragma unroll
              for (int i = 0; i < OJ; ++i) {
                if ((DEBUG(p) & 16) && (i & 1)) continue;  // (ablation: half of the store bytes)
                const double2 v = make_double2(a0[i], a1[i]);
                __builtin_memcpy(lp + i * stride8, &v, 16);  // 8-byte aligned 16-byte store
              }
            }
#endif
            if (lone) *reinterpret_cast<double *>(l0 + li * stride8 + 8 * 404) = aL;
#endif
          } else {
            // 32 lanes per row (27 slots), 8 rows per pass, two passes per K line
            constexpr int RPP = G / 32, KP = (OK + RPP - 1) / RPP;
            const int sl = lt & 31, rsub = lt >> 5;
            const int dI = sl / 9 - 1, dJ = (sl / 3) % 3 - 1, dK = sl % 3 - 1;
            const bool upper = sl >= 13;
            const int need = (dI < 0 ? 1 : dI > 0 ? 8 : 0) | (dJ < 0 ? 2 : dJ > 0 ? 16 : 0) | (dK < 0 ? 4 : dK > 0 ? 32 : 0);
            const int loI = P > 0, hiI = P < N0 - 1, lenI = loI + 1 + hiI;
            double v[KP][OJ];  // ALL LDS reads of the plane first: the other half recycles the slot of plane s-2 as soon as they have returned
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) {
              const int ok = kp * RPP + rsub;
              const int srow = (VK + ok + 1) * NS + (upper ? sl - 13 : (dJ * VK + dK) * NS + 13 - sl);
              const double *src = acc + ((!upper && dI < 0) ? slot_of(P - 1) : slot_of(P)) + (ok < OK ? srow : 0);
#pragma unroll
              for (int oj = 0; oj < OJ; ++oj) v[kp][oj] = src[oj * (VK * NS)];
            }
            half_arrive(hcnt);
            arrived = true;
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) {
              const int ok = kp * RPP + rsub, Kk = K0 + ok;
              const bool kact = sl < 27 && ok < OK && Kk < N2;
              const int loK = Kk > 0, hiK = Kk < N2 - 1, lenK = loK + 1 + hiK, cumK = Kk == 0 ? 0 : 3 * Kk - 1;
              double *line = p.values + ((P == 0 ? 0 : 3 * (i64)P - 1) * T1 * T2 + lenI * (cumJ0 * T2));
              if (!lowJ && !highJ) {
                const int flag = loI | 2 | loK << 2 | hiI << 3 | 16 | hiK << 5;
                const unsigned voff8 = 8u * (unsigned)(lenI * 3 * cumK + ((dI + loI) * 3 + (dJ + 1)) * lenK + (dK + loK));
                const i64 stride8 = 8 * (i64)(lenI * 3 * (int)T2);
                if (kact && (flag & need) == need) {
                  char *lp = reinterpret_cast<char *>(line);
#pragma unroll
                  for (int oj = 0; oj < OJ; ++oj) {
                    if (!((DEBUG(p) & 16) && (oj & 1))) *reinterpret_cast<double *>(lp + voff8) = v[kp][oj];
                    lp += stride8;
                  }
                }
              } else {
#pragma unroll
                for (int oj = 0; oj < OJ; ++oj) {
                  const int J = J0 + oj;
                  const int loJ = J > 0, hiJ = J < N1 - 1, lenJ = loJ + 1 + hiJ;
                  const int flag = loI | loJ << 1 | loK << 2 | hiI << 3 | hiJ << 4 | hiK << 5;
                  const unsigned voff = (unsigned)(lenI * lenJ * cumK + ((dI + loI) * lenJ + (dJ + loJ)) * lenK + (dK + loK));
                  if (kact && J < N1 && (flag & need) == need && !((DEBUG(p) & 16) && (oj & 1))) line[voff] = v[kp][oj];
                  line += lenI * lenJ * (int)T2;
                }
              }
            }
          }
        }
        NH_TICK(6)
        if (!arrived) half_arrive(hcnt);
        if (!recycled) half_wait(hcnt, htarget);  // every wave of this half has read what it needs of plane s-2: recycle its slot
        NH_TICK(7)
        if (!recycled) {
          double2 *z = reinterpret_cast<double2 *>(acc + slot_of(s - 2));
          for (int t = lt; t < PS / 2; t += G) z[t] = make_double2(0., 0.);
        }
        NH_TICK(9)
        if (needv && !staged) stage_vertices();
      }
      if (mathrole) { NH_TICK(1) } else { NH_TICK(4) }
      lds_barrier();
      if (mathrole) { NH_TICK(2) } else { NH_TICK(5) }
    }
    u += B - A;
  }
#ifdef NH_ABLATION
  if (p.tdbg && tid == 0) {
    p.tdbg[16 + 4 * blockIdx.x] = clock64() - tstart;
    p.tdbg[16 + 4 * blockIdx.x + 1] = nfast;
    p.tdbg[16 + 4 * blockIdx.x + 2] = nslow;
  }
  if (p.tdbg && (tid & 63) == 0)
    for (int i = 0; i < 10; ++i) atomicAdd((unsigned long long *)p.tdbg + i, (unsigned long long)tacc[i]);
#endif
}
#undef NH_TICK

#include "nh_p1hex_tiles.inc"
#include "nh_p1hex_tri.inc"

// ---- uniform geometry: all element matrices are equal (the reference hoists them out of the loop too) -------------
// One thread evaluates the element matrix of the unit cell; the assembly is then a pure streaming kernel: every CSR entry is the
// sum of the <= 8 element-matrix entries of the elements that contain both its row and its column dof.
__global__ void k_p1hex_unit_matrix(P1Args p, double *Ke) {
  double X[2][2][2][3];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int c = 0; c < 2; ++c) {
        X[a][b][c][0] = p.origin[0] + p.scale[0] * a;
        X[a][b][c][1] = p.origin[1] + p.scale[1] * b;
        X[a][b][c][2] = p.origin[2] + p.scale[2] * c;
      }
  ElemTables T;
  {
    double (&R0)[3][3] = T.R0, (&R1)[3][3] = T.R1, (&R2)[3][3] = T.R2;
    double (&W01)[2][2][3] = T.W01, (&W02)[2][2][3] = T.W02, (&W12)[2][2][3] = T.W12;
    double (&Mm)[3][3][3] = T.Mm;
    for (int i = 0; i < 27; ++i) (&Mm[0][0][0])[i] = 0.;
    const bool hasm = p.hasm;
#define NH_P1HEX_QS(q) 1.
#define NH_P1HEX_QM(q) 1.
#include "nh_p1hex_math.inc"
#undef NH_P1HEX_QS
#undef NH_P1HEX_QM
  }
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) Ke[a * 8 + b] = element_entry(T, a < b ? a : b, a < b ? b : a);
}

__global__ __launch_bounds__(256) void k_p1hex_uniform(P1Args p, const double *KeG) {
  // One workgroup per (I, J) dof line.  Along K the rows of a line repeat: [first row: R2 entries][N2 - 2 interior rows with the
  // SAME R3 = lenI lenJ 3 values each][last row: R2 = lenI lenJ 2 entries], contiguous in the CSR array.  27 threads build the three
  // small tables in LDS; then all 256 threads stream the line as a periodic fill with 16-byte stores (two entries per lane).
  __shared__ double Ke[64], tabI[2 * 27], tabF[18], tabL[18];
  if (threadIdx.x < 64) Ke[threadIdx.x] = KeG[threadIdx.x];
  __syncthreads();
  const int N0 = p.n0 + 1, N1 = p.n1 + 1, N2 = p.n2 + 1;
  const int I = p.pl0 + blockIdx.x / N1, J = blockIdx.x % N1;
  const i64 T1 = 3 * (i64)N1 - 2, T2 = 3 * (i64)N2 - 2;
  const int lenI = len_of(I, N0), lenJ = len_of(J, N1), R3 = lenI * lenJ * 3, R2 = lenI * lenJ * 2;
  double *const line = p.values + (cum_of(I, N0) * T1 * T2 + lenI * (cum_of(J, N1) * T2));  // CSR position of row (I, J, 0)
  const int sl = threadIdx.x;
  if (sl < 27) {
    const int dI = sl / 9 - 1, dJ = (sl / 3) % 3 - 1, dK = sl % 3 - 1;
    const int cI = I + dI, cJ = J + dJ;
    if (cI >= 0 && cI < N0 && cJ >= 0 && cJ < N1) {
      // w[a2][b2]: sum over the (<= 4) element columns (i, j) around the dof line of Ke[(a0,a1,a2)][(b0,b1,b2)]
      double w[2][2];
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
          double v = 0.;
#pragma unroll
          for (int oi = -1; oi <= 0; ++oi) {
            const int i = I + oi, a0 = -oi, b0 = a0 + dI;
            if (b0 < 0 || b0 > 1 || i < p.lay0 || i >= p.lay1) continue;
#pragma unroll
            for (int oj = -1; oj <= 0; ++oj) {
              const int j = J + oj, a1 = -oj, b1 = a1 + dJ;
              if (b1 < 0 || b1 > 1 || j < 0 || j >= p.n1) continue;
              v += Ke[((a0 * 2 + a1) * 2 + a2) * 8 + (b0 * 2 + b1) * 2 + b2];
            }
          }
          w[a2][b2] = v;
        }
      const int prefix = (dI + (I > 0)) * lenJ + (dJ + (J > 0));
      // a row K gets element k = K - 1 (local a2 = 1) and element k = K (a2 = 0); b2 = a2 + dK
      double v = 0.;
      if (dK <= 0) v += w[1][1 + dK];
      if (dK >= 0) v += w[0][dK];
      tabI[prefix * 3 + (dK + 1)] = v;
      tabI[R3 + prefix * 3 + (dK + 1)] = v;                      // second period: pairs may wrap around
      if (dK >= 0) tabF[prefix * 2 + dK] = w[0][dK];            // K = 0: element 0 only, columns K and K + 1
      if (dK <= 0) tabL[prefix * 2 + (dK + 1)] = w[1][1 + dK];  // K = N2 - 1: element N2 - 2 only, columns K - 1 and K
    }
  }
  __syncthreads();
  const int ninner = (N2 - 2) * R3;
  double *const inner = line + R2;
  // 16-byte stores need 16-byte alignment to be worth it: one scalar entry first if the run starts on an odd double
  const int head = (int)((reinterpret_cast<size_t>(inner) >> 3) & 1);
  const int npairs = (ninner - head) / 2;
  int r = (head + 2 * (int)threadIdx.x) % R3;
  const int step = 512 % R3;
  double2 *const dst = reinterpret_cast<double2 *>(inner + head);
  for (int e = threadIdx.x; e < npairs; e += 256) {
    dst[e] = make_double2(tabI[r], tabI[r + 1]);
    r += step;
    r -= r >= R3 ? R3 : 0;
  }
  if (threadIdx.x == 0 && head) inner[0] = tabI[0];
  if (threadIdx.x == 1 && ((ninner - head) & 1)) inner[ninner - 1] = tabI[(ninner - 1) % R3];
  if (threadIdx.x < R2) {
    line[threadIdx.x] = tabF[threadIdx.x];
    line[R2 + ninner + threadIdx.x] = tabL[threadIdx.x];
  }
}

__global__ void k_p1hex_pattern(int n0, int n1, int n2, i64 row0, i64 row1, i64 *rowptr, i64 *colidx) {
  // one thread per (row, slot); rows in [row0, row1); output re-based to row0
  const int N0 = n0 + 1, N1 = n1 + 1, N2 = n2 + 1;
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 row = row0 + t / 27;
  const int slot = (int)(t % 27);
  if (row > row1 || (row == row1 && slot != 0)) return;
  const i64 T1 = 3 * (i64)N1 - 2, T2 = 3 * (i64)N2 - 2;
  auto rp = [&](i64 r) -> i64 {
    if (r >= (i64)N0 * N1 * N2) return (3 * (i64)N0 - 2) * T1 * T2;
    const int Kk = (int)(r % N2), J = (int)((r / N2) % N1), I = (int)(r / ((i64)N2 * N1));
    return cum_of(I, N0) * T1 * T2 + len_of(I, N0) * (cum_of(J, N1) * T2 + (i64)len_of(J, N1) * cum_of(Kk, N2));
  };
  const i64 base = rp(row0);
  if (slot == 0) rowptr[row - row0] = rp(row) - base;
  if (row >= row1) return;
  const int Kk = (int)(row % N2), J = (int)((row / N2) % N1), I = (int)(row / ((i64)N2 * N1));
  const int dI = slot / 9 - 1, dJ = (slot / 3) % 3 - 1, dK = slot % 3 - 1;
  const int cI = I + dI, cJ = J + dJ, cK = Kk + dK;
  if (cI < 0 || cI >= N0 || cJ < 0 || cJ >= N1 || cK < 0 || cK >= N2) return;
  const int lenJ = len_of(J, N1), lenK = len_of(Kk, N2);
  const int pos = ((dI + (I > 0)) * lenJ + (dJ + (J > 0))) * lenK + (dK + (Kk > 0));
  colidx[rp(row) - base + pos] = ((i64)cI * N1 + cJ) * N2 + cK;
}

}  // namespace

// launch of the marching kernel (matrix: VEC = false; K u: VEC = true)
template <bool VEC, bool MASS, bool COEF>
static int launch_march_inst(const nh_p1hex_args *a, P1Args &p, void *stream) {
#ifdef NH_ABLATION
  p.debug = getenv("NH_P1HEX_DEBUG") ? atoi(getenv("NH_P1HEX_DEBUG")) : 0;
#endif
  // measured optimum at 64^3 .. 256^3 (16 = unweighted: +7 .. +18 %); the tuning override is read once per process and clamped to a sane range
  static const int wbnd_env = getenv("NH_P1HEX_WBND") ? std::min(64, std::max(8, atoi(getenv("NH_P1HEX_WBND")))) : 20;
  p.wbnd = getenv("NH_P1HEX_STAGE_LATE") ? -wbnd_env : wbnd_env;
  constexpr int TJ = 16, TK = 16, L = 2, NTM = L * TJ * TK, NS = VEC ? 1 : 15, VW = VEC ? 4 : 3;
  // (per-step path of a Newton loop / of a multi-GPU slab whose kernel lasts ~20 us: device queries and the LDS attribute once per process)
  // (function attributes and the CU count belong to a DEVICE: remembered per device index; the library is driven by one host thread, include/nutils_hip.h)
  int dev = 0;
  NH_CHECK_HIP(hipGetDevice(&dev));
  NH_REQUIRE(dev >= 0 && dev < 16, "nh_p1hex: device index %d not supported", dev);
  static int cus_of[16] = {0};
  if (!cus_of[dev]) {
    int n = 256;
    NH_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus_of[dev] = n;
  }
  const int cus = cus_of[dev];
  p.nbj = (p.n1 + 1 + TJ - 2) / (TJ - 1);
  p.nbk = (p.n2 + 1 + TK - 2) / (TK - 1);
  const size_t ldsm = sizeof(double) * ((L + 2) * (((TJ + 1) * (TK + 1) * NS + 1) & ~1) + (L + 1) * (TJ + 1) * (TK + 1) * VW + 2);
  const i64 units = (i64)p.nbj * p.nbk * (p.pl1 - p.pl0);
  NH_REQUIRE(a->max_workgroups >= 0, "nh_p1hex: negative max_workgroups");
  unsigned grid = (unsigned)std::min<i64>(units, a->max_workgroups ? std::min(cus, a->max_workgroups) : cus);
#ifdef NH_ABLATION
  if (getenv("NH_P1HEX_MAXWG")) grid = std::min<unsigned>(grid, (unsigned)atoi(getenv("NH_P1HEX_MAXWG")));
#endif
  auto kern = k_p1hex_march<TJ, TK, L, VEC, MASS, COEF>;
  if constexpr (!VEC && L == 2) {  // the matrix goes through the skewed kernel (same tile, same LDS, same launch) unless NH_P1HEX_MARCH=1
    const char *env = getenv("NH_P1HEX_MARCH"), *kenv = getenv("NH_P1HEX_KERNEL");  // (read per launch: the tests compare the kernels within one process)
    const bool march = (env && atoi(env)) || (kenv && !strcmp(kenv, "march"));
    if (!march) kern = k_p1hex_skew<TJ, TK, MASS, COEF>;
  }
  static const void *attr_set[16][2] = {{nullptr, nullptr}};  // (this instantiation: the marching and the skewed kernel, per device)
  if (attr_set[dev][0] != (const void *)kern && attr_set[dev][1] != (const void *)kern) {
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm));
    attr_set[dev][attr_set[dev][0] ? 1 : 0] = (const void *)kern;
  }
#ifdef NH_ABLATION
  static long long *tdbg = nullptr;
  if (!tdbg) NH_CHECK_HIP(hipMalloc((void **)&tdbg, (16 + 4 * 1024) * sizeof(long long)));
  NH_CHECK_HIP(hipMemsetAsync(tdbg, 0, (16 + 4 * 1024) * sizeof(long long), nh_stream(stream)));
  p.tdbg = getenv("NH_P1HEX_TIMERS") ? tdbg : nullptr;
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTM), ldsm, nh_stream(stream), p);
  NH_LAUNCH_CHECK();
#ifdef NH_ABLATION
  if (p.tdbg) {
    static long long h[16 + 4 * 1024];
    NH_CHECK_HIP(hipMemcpy(h, tdbg, sizeof h, hipMemcpyDeviceToHost));
    if (getenv("NH_P1HEX_WGTIMES"))
      for (unsigned g = 0; g < grid; ++g) fprintf(stderr, "wg %u cycles %lld fast %lld slow %lld\n", g, h[16 + 4 * g], h[16 + 4 * g + 1], h[16 + 4 * g + 2]);
    const double nw = (double)grid * (NTM / 64);
    if (kern != (void (*)(P1Args))k_p1hex_march<TJ, TK, L, VEC, MASS, COEF>)
      fprintf(stderr, "p1hex_skew cycles per wave: prologue %.0f | math %.0f | - %.0f | math wait %.0f | lds reads %.0f | store issue %.0f | half wait %.0f | zero %.0f | stage %.0f | mem wait %.0f\n", h[0] / nw, h[3] / nw, h[1] / nw,
              h[2] / nw, h[8] / nw, h[6] / nw, h[7] / nw, h[9] / nw, h[4] / nw, h[5] / nw);
    else
    fprintf(stderr, "p1hex_march cycles per wave: stage+next %.0f | B1 %.0f | load+math %.0f | B2 %.0f | stage+flush %.0f | B3 %.0f | zero %.0f\n", h[0] / nw, h[1] / nw,
            h[2] / nw, h[3] / nw, h[4] / nw, h[5] / nw, h[6] / nw);
  }
#endif
  return NH_OK;
}


// ---- launch of the exact-tile kernel (matrix) ------------------------------------------------------------------------------------
namespace {
struct TileScratch {
  void *base = nullptr;  // one allocation: ctl (64 B) | flags | flags_pro | exp | exp_pro
  size_t cap = 0;
  unsigned *err_host = nullptr, *err_dev = nullptr;
};
TileScratch g_tiles[16];  // per device
}  // namespace

int nh_p1hex_tiles_release(void) {
  for (auto &t : g_tiles) {
    if (t.base) NH_CHECK_HIP(hipFree(t.base));
    t.base = nullptr, t.cap = 0;
  }
  return NH_OK;
}

template <bool MASS, bool COEF>
static int launch_tiles_inst(const nh_p1hex_args *a, P1Args &p, void *stream) {
  using namespace p1t;
  int dev = 0;
  NH_CHECK_HIP(hipGetDevice(&dev));
  NH_REQUIRE(dev >= 0 && dev < 16, "nh_p1hex: device index %d not supported", dev);
  static int cus_of[16] = {0};
  static bool attr_set[16] = {false};
  if (!cus_of[dev]) {
    int n = 256;
    NH_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus_of[dev] = n;
  }
  auto kern = k_p1hex_tiles<MASS, COEF>;
  if (!attr_set[dev]) {
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr_set[dev] = true;
  }
  TileScratch &ts = g_tiles[dev];
  if (!ts.err_host) {
    NH_CHECK_HIP(hipHostMalloc((void **)&ts.err_host, sizeof(unsigned), hipHostMallocMapped));
    *ts.err_host = 0;
    NH_CHECK_HIP(hipHostGetDevicePointer((void **)&ts.err_dev, ts.err_host, 0));
  }
  static size_t layout[16][3] = {{0}};
  if (*ts.err_host) {  // raised by an EARLIER launch (the word is read without synchronising)
    *ts.err_host = 0;
    layout[dev][0] = 0;  // the flag counts of that launch are incomplete: start over on zeroed flags
    nh_set_error("nh_p1hex_laplace: a workgroup of an earlier launch timed out waiting for the tile faces of its neighbour -- are other kernels holding CUs? "
                 "(NH_P1HEX_KERNEL=skew selects the kernel without inter-workgroup exchange)");
    return NH_EHIP;
  }
  p.nbj = (p.n1 + T - 1) / T;
  p.nbk = (p.n2 + T - 1) / T;
  const int ncols = p.nbj * p.nbk, NPL = p.pl1 - p.pl0;
  NH_REQUIRE(a->max_workgroups >= 0, "nh_p1hex: negative max_workgroups");
  const int wgmax = a->max_workgroups ? std::min(cus_of[dev], a->max_workgroups) : cus_of[dev];
  // plane segments per column: the same in every column (producer and consumer of a tile face work on the same plane at the same time);
  // a run of n planes costs n + 3 slots
  int nseg = 1;
  {
    static const int nseg_env = getenv("NH_P1HEX_NSEG") ? atoi(getenv("NH_P1HEX_NSEG")) : 0;
    double best = 1e300;
    for (int c = 1; c <= NPL; ++c) {
      const i64 rounds = ((i64)ncols * c + wgmax - 1) / wgmax;
      const double cost = (double)rounds * ((NPL + c - 1) / c + 3);
      if (cost < best) best = cost, nseg = c;
    }
    if (nseg_env > 0) nseg = std::min(nseg_env, NPL);
  }
  const unsigned grid = (unsigned)std::min<i64>((i64)ncols * nseg, wgmax);
  const size_t nflags = (size_t)ncols * NPL, nfpro = (size_t)ncols * nseg;
  const size_t off_exp = tile_exp_offset(ncols, NPL, nseg), need = off_exp + (nflags + nfpro) * NEXP * sizeof(double);
  // the layout depends on (ncols, NPL, nseg): a change of any of them restarts the epoch on zeroed flags
  if (need > ts.cap || layout[dev][0] != nflags || layout[dev][1] != nfpro || layout[dev][2] != (size_t)NPL) {
    NH_CHECK_HIP(hipStreamSynchronize(nh_stream(stream)));
    if (need > ts.cap) {
      if (ts.base) NH_CHECK_HIP(hipFree(ts.base));
      ts.base = nullptr, ts.cap = 0;
      NH_CHECK_HIP(hipMalloc(&ts.base, need));
      ts.cap = need;
    }
    NH_CHECK_HIP(hipMemsetAsync(ts.base, 0, off_exp, nh_stream(stream)));
    NH_CHECK_HIP(hipMemcpyAsync((char *)ts.base + 8, &ts.err_dev, sizeof(unsigned *), hipMemcpyHostToDevice, nh_stream(stream)));
    static const unsigned always_raised = 0x7fffffffu;  // ctl[4]: the flag of a producer that does not exist
    NH_CHECK_HIP(hipMemcpyAsync((char *)ts.base + 16, &always_raised, sizeof(unsigned), hipMemcpyHostToDevice, nh_stream(stream)));
    NH_CHECK_HIP(hipStreamSynchronize(nh_stream(stream)));
    layout[dev][0] = nflags, layout[dev][1] = nfpro, layout[dev][2] = (size_t)NPL;
  }
  TileArgs ta;
  ta.base = (char *)ts.base;
  ta.nseg = nseg;
  ta.xcd = grid % 8 == 0 && !a->max_workgroups;
#ifdef NH_ABLATION
  p.debug = getenv("NH_P1HEX_DEBUG") ? atoi(getenv("NH_P1HEX_DEBUG")) : 0;
  if (getenv("NH_P1HEX_NOXCD")) ta.xcd = 0;
  static long long *tdbg = nullptr;
  if (!tdbg) NH_CHECK_HIP(hipMalloc((void **)&tdbg, 16 * sizeof(long long)));
  NH_CHECK_HIP(hipMemsetAsync(tdbg, 0, 16 * sizeof(long long), nh_stream(stream)));
  p.tdbg = getenv("NH_P1HEX_TIMERS") ? tdbg : nullptr;
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, nh_stream(stream), p, ta);
  NH_LAUNCH_CHECK();
#ifdef NH_ABLATION
  if (p.tdbg) {
    long long h[16];
    NH_CHECK_HIP(hipMemcpy(h, tdbg, sizeof h, hipMemcpyDeviceToHost));
    const double nw = (double)grid * 4;  // waves per role
    static int nprint = 0;
    if (nprint++ < 3)
      fprintf(stderr, "p1hex_tiles cycles per wave and slot (grid %u nseg %d, %.1f slots per role): MATH verts+routine %.0f | flags+imports %.0f | barrier %.0f || MEM import add+vertex loads+export+waits %.0f | "
              "flush reads+flag+stage+stores %.0f | half wait+zero %.0f | barrier %.0f\n", grid, nseg, ((double)(p.pl1 - p.pl0) / nseg + 3) / 2,
              h[2] / nw / (((double)(p.pl1 - p.pl0) / nseg + 3) / 2), h[3] / nw / (((double)(p.pl1 - p.pl0) / nseg + 3) / 2), h[11] / nw / (((double)(p.pl1 - p.pl0) / nseg + 3) / 2),
              h[6] / nw / (((double)(p.pl1 - p.pl0) / nseg + 3) / 2), h[8] / nw / (((double)(p.pl1 - p.pl0) / nseg + 3) / 2), h[10] / nw / (((double)(p.pl1 - p.pl0) / nseg + 3) / 2),
              h[12] / nw / (((double)(p.pl1 - p.pl0) / nseg + 3) / 2));
  }
#endif
  return NH_OK;
}

// ---- launch of the three-role kernel (matrix) ---------------------------------------------------------------------------------------
template <bool MASS, bool COEF>
static int launch_tri_inst(const nh_p1hex_args *a, P1Args &p, void *stream) {
  int dev = 0;
  NH_CHECK_HIP(hipGetDevice(&dev));
  NH_REQUIRE(dev >= 0 && dev < 16, "nh_p1hex: device index %d not supported", dev);
  static int cus_of[16] = {0};
  static bool attr_set[16] = {false};
  if (!cus_of[dev]) {
    int n = 256;
    NH_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus_of[dev] = n;
  }
  auto kern = k_p1hex_tri<MASS, COEF>;
  if (!attr_set[dev]) {
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p1r::LDS_BYTES));
    attr_set[dev] = true;
  }
  p.nbj = (p.n1 + 1 + p1r::OW - 1) / p1r::OW;
  p.nbk = (p.n2 + 1 + p1r::OW - 1) / p1r::OW;
  NH_REQUIRE(a->max_workgroups >= 0, "nh_p1hex: negative max_workgroups");
  const i64 units = (i64)p.nbj * p.nbk * (p.pl1 - p.pl0);
  const unsigned grid = (unsigned)std::min<i64>(units, a->max_workgroups ? std::min(cus_of[dev], a->max_workgroups) : cus_of[dev]);
  static const int delay = getenv("NH_P1HEX_TRI_DELAY") ? std::min(64, std::max(0, atoi(getenv("NH_P1HEX_TRI_DELAY")))) : 1;
  p.wbnd = delay;  // (start delay of every second workgroup, in units of 4096 cycles)
  static const int noprio = getenv("NH_P1HEX_TRI_NOPRIO") ? 1 : 0;
  p.debug = noprio;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(768), p1r::LDS_BYTES, nh_stream(stream), p);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

static int launch_tri(const nh_p1hex_args *a, P1Args &p, void *stream) {
  const bool coef = p.qscale || p.qmass;
  if (p.hasm) return coef ? launch_tri_inst<true, true>(a, p, stream) : launch_tri_inst<true, false>(a, p, stream);
  return coef ? launch_tri_inst<false, true>(a, p, stream) : launch_tri_inst<false, false>(a, p, stream);
}

static int launch_tiles(const nh_p1hex_args *a, P1Args &p, void *stream) {
  const bool coef = p.qscale || p.qmass;
  if (p.hasm) return coef ? launch_tiles_inst<true, true>(a, p, stream) : launch_tiles_inst<true, false>(a, p, stream);
  return coef ? launch_tiles_inst<false, true>(a, p, stream) : launch_tiles_inst<false, false>(a, p, stream);
}

template <bool VEC>
static int launch_march(const nh_p1hex_args *a, P1Args &p, void *stream) {
  const bool coef = p.qscale || p.qmass;
  if (p.hasm) return coef ? launch_march_inst<VEC, true, true>(a, p, stream) : launch_march_inst<VEC, true, false>(a, p, stream);
  return coef ? launch_march_inst<VEC, false, true>(a, p, stream) : launch_march_inst<VEC, false, false>(a, p, stream);
}

extern "C" {

int nh_p1hex_pattern(const int *shape, int64_t row_begin, int64_t row_end, int64_t *rowptr_dev, int64_t *colidx_dev, void *stream) {
  NH_REQUIRE(shape && rowptr_dev && colidx_dev, "nh_p1hex_pattern: NULL argument");
  NH_REQUIRE(shape[0] >= 1 && shape[1] >= 1 && shape[2] >= 1, "nh_p1hex_pattern: empty mesh");
  const i64 nrows_total = (i64)(shape[0] + 1) * (shape[1] + 1) * (shape[2] + 1);
  NH_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= nrows_total, "nh_p1hex_pattern: row range out of bounds");
  const i64 n = (row_end - row_begin + 1) * 27;
  hipLaunchKernelGGL(k_p1hex_pattern, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nh_stream(stream), shape[0], shape[1], shape[2],
                     (i64)row_begin, (i64)row_end, (i64 *)rowptr_dev, (i64 *)colidx_dev);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

static int fill_p1args(const nh_p1hex_args *a, P1Args &p) {
  NH_REQUIRE(a, "nh_p1hex: NULL argument");
  NH_REQUIRE(a->shape[0] >= 1 && a->shape[1] >= 1 && a->shape[2] >= 1, "nh_p1hex: empty mesh");
  NH_REQUIRE(0 <= a->layer_begin && a->layer_begin <= a->layer_end && a->layer_end <= a->shape[0], "nh_p1hex: layer range");
  NH_REQUIRE(0 <= a->plane_begin && a->plane_begin <= a->plane_end && a->plane_end <= a->shape[0] + 1, "nh_p1hex: plane range");
  p.n0 = a->shape[0];
  p.n1 = a->shape[1];
  p.n2 = a->shape[2];
  p.lay0 = a->layer_begin;
  p.lay1 = a->layer_end;
  p.pl0 = a->plane_begin;
  p.pl1 = a->plane_end;
  p.verts = a->verts_dev;
  for (int d = 0; d < 3; ++d) {
    p.origin[d] = a->origin[d];
    p.scale[d] = a->scale[d];
  }
  for (int q = 0; q < 2; ++q) {
    p.n[0][q] = 1. - a->gauss_x[q];
    p.n[1][q] = a->gauss_x[q];
    p.c[0][q] = p.n[0][q] * p.n[0][q];
    p.c[1][q] = p.n[0][q] * p.n[1][q];
    p.c[2][q] = p.n[1][q] * p.n[1][q];
  }
  for (int qa = 0; qa < 2; ++qa)
    for (int qb = 0; qb < 2; ++qb)
      for (int qc = 0; qc < 2; ++qc) {
        p.wk[qa][qb][qc] = a->kappa * a->gauss_w[qa] * a->gauss_w[qb] * a->gauss_w[qc];
        p.wm[qa][qb][qc] = a->mass * a->gauss_w[qa] * a->gauss_w[qb] * a->gauss_w[qc];
      }
  p.hasm = a->mass != 0. || a->qmass_dev != nullptr;
  p.qmass = a->qmass_dev;
  p.values = a->values_dev;
  p.qscale = a->qscale_dev;
  p.u = nullptr;
  p.out = nullptr;
  p.accumulate = 0;
  p.nbj = p.nbk = 0;
  p.debug = 0;
  p.wbnd = 16;
  p.tdbg = nullptr;
  return NH_OK;
}

int nh_p1hex_unit_matrix(const nh_p1hex_args *a, double *ke_dev, void *stream) {
  NH_REQUIRE(ke_dev, "nh_p1hex_unit_matrix: NULL output");
  P1Args p;
  int rc = fill_p1args(a, p);
  if (rc) return rc;
  hipLaunchKernelGGL(k_p1hex_unit_matrix, dim3(1), dim3(1), 0, nh_stream(stream), p, ke_dev);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

int nh_p1hex_laplace(const nh_p1hex_args *a, void *stream) {
  NH_REQUIRE(a && a->values_dev, "nh_p1hex_laplace: NULL argument");
  P1Args p;
  int rc = fill_p1args(a, p);
  if (rc) return rc;
  if (a->plane_begin == a->plane_end) return NH_OK;
  NH_REQUIRE(!(a->qscale_dev || a->qmass_dev) || a->verts_dev, "nh_p1hex_laplace: a coefficient array needs explicit vertices");
  if (!a->verts_dev) {  // uniform geometry: unit element matrix + streaming kernel
    double *Ke = nullptr;
    if (!a->unit_matrix_dev) {
      NH_CHECK_HIP(hipMallocAsync((void **)&Ke, 64 * sizeof(double), nh_stream(stream)));
      hipLaunchKernelGGL(k_p1hex_unit_matrix, dim3(1), dim3(1), 0, nh_stream(stream), p, Ke);
    }
    hipLaunchKernelGGL(k_p1hex_uniform, dim3((unsigned)((p.pl1 - p.pl0) * (p.n1 + 1))), dim3(256), 0, nh_stream(stream), p,
                       a->unit_matrix_dev ? a->unit_matrix_dev : (const double *)Ke);
    NH_LAUNCH_CHECK();
    if (Ke) NH_CHECK_HIP(hipFreeAsync(Ke, nh_stream(stream)));
    return NH_OK;
  }
  // matrix kernel (read per launch: the tests compare the kernels within one process): skew (default) | tiles | march.  The exact-tile kernel
  // (no lateral halo, tile faces exchanged between workgroups) is correct on every mesh of the test suite but measured SLOWER at 128^3 (0.21 ms
  // against 0.167 ms: profiles/r04_c2_exact_tiles.md) and stays opt-in.
  const char *env = getenv("NH_P1HEX_KERNEL");
  if (env && !strcmp(env, "tiles")) return launch_tiles(a, p, stream);
  if (env && !strcmp(env, "tri")) return launch_tri(a, p, stream);
  return launch_march<false>(a, p, stream);
}

int nh_p1hex_apply(const nh_p1hex_args *a, const double *u_dev, double *out_dev, int accumulate, void *stream) {
  NH_REQUIRE(a && u_dev && out_dev, "nh_p1hex_apply: NULL argument");
  NH_REQUIRE(a->verts_dev, "nh_p1hex_apply: explicit vertices required");
  P1Args p;
  int rc = fill_p1args(a, p);
  if (rc) return rc;
  if (a->plane_begin == a->plane_end) return NH_OK;
  p.u = u_dev;
  p.out = out_dev;
  p.accumulate = accumulate != 0;
  return launch_march<true>(a, p, stream);
}

}  // extern "C"
