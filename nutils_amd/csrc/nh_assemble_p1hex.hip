// Fast path for the headline configuration: scalar Laplace stiffness, trilinear ('std'
// degree 1) basis on a structured hex mesh (mesh.rectilinear, mesh.py:34-60), 2-point
// Gauss per axis, isoparametric P1 geometry (or the uniform box geometry).
//
// WRITE-ONCE design (no global atomics, no zero-fill, CSR values stored exactly once):
//   * a persistent workgroup (one per CU) OWNS one box of BI x BJ x BK dof rows at a time;
//   * its threads (one per element) recompute the local matrices of all elements that touch the box --
//     (BI+1)(BJ+1)(BK+1), i.e. a one-element halo is recomputed instead of communicated; the vertex block of the NEXT box
//     is prefetched during barrier + flush and staged in LDS;
//   * contributions are reduced in LDS (ds_add_f64) into a [rows][27] slot array
//     (slot = 9(dI+1) + 3(dJ+1) + (dK+1): column offset relative to the row dof);
//   * the finished rows are streamed to HBM, coalesced, at closed-form CSR offsets: for the reference's structured dof
//     numbering the sorted-unique pattern (evaluable.py:588-616) is the tensor product of per-axis ranges
//     [max(X-1,0), min(X+1,N-1)], so rowptr(I,J,K) and the position of a column inside its row are pure arithmetic.
// HBM traffic = vertex coordinates (each box re-reads its halo) + values.  Uniform meshes take k_p1hex_uniform instead.
#include "nh_common.h"
#include <algorithm>
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
  double *values;
  int nbj, nbk;            // boxes per axis (j, k)
  int nboxes;
  int debug;               // ablation switches (NH_P1HEX_DEBUG env): 1 = no LDS reduction, 2 = no HBM stores, 4 = no element math
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

struct ElemTables {
  double R0[3][3], R1[3][3], R2[3][3];             // diagonal (j == k) terms, indexed by the pair classes p = a_d + b_d
  double W01[2][2][3], W02[2][2][3], W12[2][2][3];  // off-diagonal terms
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
         s02 * T.W02[b0][a2][p1] + s20 * T.W02[a0][b2][p1] + s12 * T.W12[b1][a2][p0] + s21 * T.W12[a1][b2][p0];
}

// Cooperative load of the (BI+2)(BJ+2)(BK+2) vertex block of box `box` into registers: thread t owns block vertices t + k*NT.
// Issued one box ahead (after the scatter, before the barrier) so that the HBM/L2 latency hides behind barrier + flush; the
// block is then staged in LDS and every element thread reads its 8 vertices from there.
template <int BI, int BJ, int BK, int NT, int VPT>
__device__ __forceinline__ void load_vertex_block(const P1Args &p, int box, int tid, double (&V)[VPT][3]) {
  constexpr int VI = BI + 2, VJ = BJ + 2, VK = BK + 2;
  const int N0 = p.n0 + 1, N1 = p.n1 + 1, N2 = p.n2 + 1;
  int b = box;
  const int bk = b % p.nbk; b /= p.nbk;
  const int bj = b % p.nbj;
  const int bi = b / p.nbj;
  const int I0 = p.pl0 + bi * BI - 1, J0 = bj * BJ - 1, K0 = bk * BK - 1;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = tid + k * NT;
    const int c = v % VK, bb = (v / VK) % VJ, a = v / (VK * VJ);
    const int I = I0 + a, J = J0 + bb, K = K0 + c;
    const bool ok = box < p.nboxes && v < VI * VJ * VK && I >= 0 && I < N0 && J >= 0 && J < N1 && K >= 0 && K < N2;
    V[k][0] = V[k][1] = V[k][2] = 0.;
    if (ok) {  // the box kernel is only launched with a vertex array (uniform meshes take k_p1hex_uniform)
      const double *src = p.verts + (((i64)I * N1 + J) * N2 + K) * 3;
      V[k][0] = src[0];
      V[k][1] = src[1];
      V[k][2] = src[2];
    }
  }
}

template <int BI, int BJ, int BK, int NT, int NBUF, int EPT>
__global__ __launch_bounds__(NT) void k_p1hex_laplace(P1Args p) {
  constexpr int ROWS = BI * BJ * BK;
  constexpr int EI = BI + 1, EJ = BJ + 1, EK = BK + 1;
  static_assert(EI * EJ * EK <= NT * EPT, "EPT elements per thread");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // two accumulator sets (double buffering across consecutive boxes of this persistent workgroup):
  //   acc [ROWS][27] f64, rowbase [ROWS] i64 (CSR offset of the row, -1 = not written), rowflag [ROWS] i32
  constexpr int SETD = ROWS * 27 + ROWS + (ROWS + 1) / 2 + 1;  // doubles per set (kept even for 16-byte alignment)
  constexpr int SET = SETD + (SETD & 1);
  constexpr int VI = BI + 2, VJ = BJ + 2, VK = BK + 2, NV = VI * VJ * VK, VPT = (NV + NT - 1) / NT;
  double *vbuf = lds + NBUF * SET;  // [NV][3] vertex block of the current box
  const int tid = threadIdx.x;
  const int N0 = p.n0 + 1, N1 = p.n1 + 1, N2 = p.n2 + 1;
  for (int t = tid; t < NBUF * SET; t += NT) lds[t] = 0.;
  double V[VPT][3];
  load_vertex_block<BI, BJ, BK, NT, VPT>(p, blockIdx.x, tid, V);
  __syncthreads();

  for (int box = blockIdx.x, it = 0; box < p.nboxes; box += gridDim.x, ++it) {
  double *acc = lds + (NBUF == 2 ? (it & 1) : 0) * SET;
  // the row table alternates between two copies even with a single accumulator set: fast waves write the table of box k+1
  // while slow waves still flush box k
  i64 *rowbase = reinterpret_cast<i64 *>(vbuf + NV * 3 + (NV & 1)) + (it & 1) * ROWS;
  int *rowflag = reinterpret_cast<int *>(reinterpret_cast<i64 *>(vbuf + NV * 3 + (NV & 1)) + 2 * ROWS) + (it & 1) * ROWS;
  int b = box;
  const int bk = b % p.nbk; b /= p.nbk;
  const int bj = b % p.nbj;
  const int bi = b / p.nbj;
  const int I0 = p.pl0 + bi * BI, J0 = bj * BJ, K0 = bk * BK;

#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = tid + k * NT;
    if (v < NV) {
      vbuf[v * 3 + 0] = V[k][0];
      vbuf[v * 3 + 1] = V[k][1];
      vbuf[v * 3 + 2] = V[k][2];
    }
  }
  for (int rr = tid; rr < ROWS; rr += NT) {  // closed-form CSR row offsets, once per row (not once per entry)
    const int lk = rr % BK, lj = (rr / BK) % BJ, li = rr / (BK * BJ);
    const int I = I0 + li, J = J0 + lj, Kk = K0 + lk;
    i64 base = -1;
    int flag = 0;
    if (I < p.pl1 && J < N1 && Kk < N2) {
      const i64 T1 = 3 * (i64)N1 - 2, T2 = 3 * (i64)N2 - 2;  // sum of len over an axis
      base = cum_of(I, N0) * T1 * T2 + len_of(I, N0) * (cum_of(J, N1) * T2 + (i64)len_of(J, N1) * cum_of(Kk, N2));
      flag = (I > 0) | (J > 0) << 1 | (Kk > 0) << 2 | (I < N0 - 1) << 3 | (J < N1 - 1) << 4 | (Kk < N2 - 1) << 5;
    }
    rowbase[rr] = base;
    rowflag[rr] = flag;
  }
  __syncthreads();  // vertex block + row table visible; the previous flush (which zeroed acc) is complete

#pragma unroll 1
  for (int ke = 0; ke < EPT; ++ke) {
    const int el = tid + ke * NT;
    const int ek = el % EK, ej = (el / EK) % EJ, ei = el / (EK * EJ);
    double X[2][2][2][3];
    const int gi = I0 - 1 + ei, gj = J0 - 1 + ej, gk = K0 - 1 + ek;
    if (el < EI * EJ * EK && gi >= p.lay0 && gi < p.lay1 && gj >= 0 && gj < p.n1 && gk >= 0 && gk < p.n2 && !(DEBUG(p) & 4)) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const double *src = vbuf + (((ei + a) * VJ + (ej + bb)) * VK + (ek + c)) * 3;
            X[a][bb][c][0] = src[0];
            X[a][bb][c][1] = src[1];
            X[a][bb][c][2] = src[2];
          }
      double R0[3][3], R1[3][3], R2[3][3], W01[2][2][3], W02[2][2][3], W12[2][2][3];
#include "nh_p1hex_math.inc"
      // ---- form the 36 upper-triangle entries, then reduce row by row: ONE exec-mask region per row vertex (8 per element)
      // instead of one per (row, column) pair (64) -- the scalar mask bookkeeping was ~8 % of the instruction stream
      double Kt[36];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int a0 = a >> 2, a1 = (a >> 1) & 1, a2 = a & 1;
#pragma unroll
        for (int bb = a; bb < 8; ++bb) {
          const int b0 = bb >> 2, b1 = (bb >> 1) & 1, b2 = bb & 1;
          const int p0 = a0 + b0, p1 = a1 + b1, p2 = a2 + b2;
          const double s00 = (a0 == b0) ? 1. : -1., s11 = (a1 == b1) ? 1. : -1., s22 = (a2 == b2) ? 1. : -1.;
          const double s01 = (a0 == b1) ? 1. : -1., s10 = (b0 == a1) ? 1. : -1.;
          const double s02 = (a0 == b2) ? 1. : -1., s20 = (b0 == a2) ? 1. : -1.;
          const double s12 = (a1 == b2) ? 1. : -1., s21 = (b1 == a2) ? 1. : -1.;
          Kt[a * 8 - a * (a - 1) / 2 + (bb - a)] = s00 * R0[p1][p2] + s11 * R1[p0][p2] + s22 * R2[p0][p1]
                                                 + s01 * W01[b0][a1][p2] + s10 * W01[a0][b1][p2]
                                                 + s02 * W02[b0][a2][p1] + s20 * W02[a0][b2][p1]
                                                 + s12 * W12[b1][a2][p0] + s21 * W12[a1][b2][p0];
        }
      }
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int a0 = a >> 2, a1 = (a >> 1) & 1, a2 = a & 1;
        const int ra0 = ei - 1 + a0, ra1 = ej - 1 + a1, ra2 = ek - 1 + a2;  // row of vertex a relative to the box
        if (ra0 >= 0 && ra0 < BI && ra1 >= 0 && ra1 < BJ && ra2 >= 0 && ra2 < BK && !(DEBUG(p) & 1)) {
          double *row = acc + ((ra0 * BJ + ra1) * BK + ra2) * 27;
#pragma unroll
          for (int bb = 0; bb < 8; ++bb) {
            const int b0 = bb >> 2, b1 = (bb >> 1) & 1, b2 = bb & 1;
            const int lo = a < bb ? a : bb, hi = a < bb ? bb : a;
            atomicAdd(&row[(b0 - a0 + 1) * 9 + (b1 - a1 + 1) * 3 + (b2 - a2 + 1)], Kt[lo * 8 - lo * (lo - 1) / 2 + (hi - lo)]);
          }
        }
      }
      if ((DEBUG(p) & 1) && Kt[0] == 1.2345e300) acc[0] = Kt[0];
    }
  }
  load_vertex_block<BI, BJ, BK, NT, VPT>(p, box + gridDim.x, tid, V);  // next box: in flight during barrier + flush
  __syncthreads();

  // ---- stream the finished rows to HBM: 32 lanes per row (27 slots), NT/32 rows per pass ------------------
  {
    const int sl = tid & 31, rsub = tid >> 5;
    const int dI = sl / 9 - 1, dJ = (sl / 3) % 3 - 1, dK = sl % 3 - 1;
    // slot is stored iff the column exists: a -1 offset needs the lo flag, a +1 offset the hi flag
    const int need = (dI < 0 ? 1 : dI > 0 ? 8 : 0) | (dJ < 0 ? 2 : dJ > 0 ? 16 : 0) | (dK < 0 ? 4 : dK > 0 ? 32 : 0);
    if (sl < 27) {
      for (int r = rsub; r < ROWS; r += NT / 32) {
        const i64 base = rowbase[r];
        const int flag = rowflag[r];
        if (base >= 0 && (flag & need) == need && !(DEBUG(p) & 2)) {
          const int loI = flag & 1, loJ = (flag >> 1) & 1, loK = (flag >> 2) & 1;
          const int lenJ = loJ + 1 + ((flag >> 4) & 1), lenK = loK + 1 + ((flag >> 5) & 1);
          const int pos = ((dI + loI) * lenJ + (dJ + loJ)) * lenK + (dK + loK);
          p.values[base + pos] = acc[r * 27 + sl];
        }
        acc[r * 27 + sl] = 0.;  // ready for the box after next; ordered by the barrier of the next box
      }
    }
  }
  }  // persistent loop over boxes: with NBUF == 2, ONE barrier per box; the stores of this box drain while the next box computes
}

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
#include "nh_p1hex_math.inc"
  }
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) Ke[a * 8 + b] = element_entry(T, a < b ? a : b, a < b ? b : a);
}

__global__ __launch_bounds__(256) void k_p1hex_uniform(P1Args p, const double *KeG) {
  // one workgroup per (I, J) dof line; 32 lanes per row (27 slots), 8 rows per pass along K
  __shared__ double Ke[64];
  if (threadIdx.x < 64) Ke[threadIdx.x] = KeG[threadIdx.x];
  __syncthreads();
  const int N0 = p.n0 + 1, N1 = p.n1 + 1, N2 = p.n2 + 1;
  const int I = p.pl0 + blockIdx.x / N1, J = blockIdx.x % N1;
  const i64 T1 = 3 * (i64)N1 - 2, T2 = 3 * (i64)N2 - 2;
  const int lenI = len_of(I, N0), lenJ = len_of(J, N1);
  const i64 line = cum_of(I, N0) * T1 * T2 + lenI * (cum_of(J, N1) * T2);  // CSR offset of row (I, J, 0)
  const int sl = threadIdx.x & 31, rsub = threadIdx.x >> 5;
  if (sl >= 27) return;
  const int dI = sl / 9 - 1, dJ = (sl / 3) % 3 - 1, dK = sl % 3 - 1;
  const int cI = I + dI, cJ = J + dJ;
  if (cI < 0 || cI >= N0 || cJ < 0 || cJ >= N1) return;
  // per-thread constants: which of the (<= 4) element columns (i, j) around the dof line contribute, with their local vertices
  double w[2][2][2];  // [ok][a2][b2] summed over the valid (oi, oj): entry of sum_{oi,oj} Ke[(a0,a1,a2)][(b0,b1,b2)]
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
      w[0][a2][b2] = v;
    }
  const int prefix = ((dI + (I > 0)) * lenJ + (dJ + (J > 0)));
  for (int Kk = rsub; Kk < N2; Kk += 8) {
    const int cK = Kk + dK;
    if (cK < 0 || cK >= N2) continue;
    double v = 0.;
    // elements k = Kk - 1 (a2 = 1) and k = Kk (a2 = 0); b2 = a2 + dK
    if (Kk >= 1 && dK <= 0) v += w[0][1][1 + dK];
    if (Kk < p.n2 && dK >= 0) v += w[0][0][dK];
    const int lenK = len_of(Kk, N2);
    p.values[line + (i64)lenI * lenJ * cum_of(Kk, N2) + (prefix * lenK + (dK + (Kk > 0)))] = v;
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
      for (int qc = 0; qc < 2; ++qc) p.wk[qa][qb][qc] = a->kappa * a->gauss_w[qa] * a->gauss_w[qb] * a->gauss_w[qc];
  p.values = a->values_dev;
  p.nbj = p.nbk = p.nboxes = 0;
  p.debug = 0;
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
  constexpr int BI = 7, BJ = 7, BK = 7, NT = 512, NBUF = 1, EPT = 1;
  const int nbi = (p.pl1 - p.pl0 + BI - 1) / BI;
  p.nbj = (p.n1 + 1 + BJ - 1) / BJ;
  p.nbk = (p.n2 + 1 + BK - 1) / BK;
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
  constexpr int ROWS = BI * BJ * BK, SETD = ROWS * 27 + ROWS + (ROWS + 1) / 2 + 1, SET = SETD + (SETD & 1);
  const size_t lds = sizeof(double) * (NBUF * SET + (BI + 2) * (BJ + 2) * (BK + 2) * 3 + 1 + 2 * ROWS + ROWS + 2);
  p.nboxes = nbi * p.nbj * p.nbk;
  p.debug = getenv("NH_P1HEX_DEBUG") ? atoi(getenv("NH_P1HEX_DEBUG")) : 0;
  int dev = 0, cus = 256;
  NH_CHECK_HIP(hipGetDevice(&dev));
  NH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  auto kern = k_p1hex_laplace<BI, BJ, BK, NT, NBUF, EPT>;
  NH_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min(p.nboxes, cus * (NT == 256 ? 2 : 1))), dim3(NT), lds, nh_stream(stream), p);
  NH_LAUNCH_CHECK();
  return NH_OK;
}

}  // extern "C"
