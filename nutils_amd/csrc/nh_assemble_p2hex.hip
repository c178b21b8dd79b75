// Write-once assembly for the structured C0 quadratic ('std', degree 2) basis on hexahedra, scalar or vector valued,
// constant-coefficient bilinear forms (BASELINE.json configs[2]: 3-D linear elasticity, P2 vector basis).
//
// Replaces, for this basis, the generated element loop + einsum (evaluable.py:6773-6786, 1885-1886, 6414-6505) and the
// sparse dedup/accumulate of its result (evaluable.py:588-616, 5560-5682; numeric.py:434-460): same CSR layout as
// nh_pattern_expand (rows/cols lexicographic, structural zeros kept, flat dof = node * ncomp + comp), each value written ONCE.
//
// Formulation.  A[(m,c),(n,d)] = sum_{a,b} C[c,a,d,b] G_mn[a,b] with the generalised Gram matrices
//     G_mn[a,b] = sum_q w_q |J_q| D_m[q,a] D_n[q,b]            (D[.,.,0] = value, D[.,.,1+i] = d/dx_i)
// so the quadrature sum is ONE symmetric rank-nq update per element, independent of the form and of the number of components:
// (4 nodes x NS slots) x (16 nodes) tiles on v_mfma_f64_16x16x4_f64 with K = quadrature points, the 9 (or 16) slot pairs of a node
// pair end up in the SAME lane (row = 4 a + node, register index = a; one tile per trial slot b), where the constant tensor C
// is applied on the VALU.  For elasticity this is 3.0x fewer MFMAs than contracting (m) x (n,d) per test component.
//
// Ownership (no global atomics, no zero-fill, no colours).  Owner cell (io,jo,ko) owns the 8 nodes (2io+ai, 2jo+aj, 2ko+ak),
// a in {0,1}; a persistent workgroup takes whole K-lines of owner cells and marches along K.  In step k it visits the (up to) four
// elements (io-di, jo-dj, k) and computes of each only the rows of nodes the line owns, in units of 4 nodes; contributions are
// reduced in LDS (ds_add_f64) into row buffers laid out exactly like the CSR rows -- one buffer per node plane K, a ring of 3 even +
// 2 odd planes -- and a finished plane (all of its rows complete, each node's 3 rows contiguous in the value array) is streamed to HBM
// with 16-byte stores.  Every element is visited by 4 lines (halo recomputation of the cheap part: geometry + D table); the MFMA
// work is 8 units of 4 rows per element instead of 6.75.
//
// The closed-form pattern of this basis: along an axis with n elements node X couples to [X-2, X+2] (X even, clipped) or
// [X-1, X+1] (X odd); rows are tensor products of these ranges, so row pointers and column positions are arithmetic.
#include "nh_common.h"
#include <algorithm>
#include <cmath>

namespace {
#include "nh_geom.inc"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int NB = 27;  // local nodes, order first axis slowest
constexpr int NT = 512;

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
  int ndb;           // D tables in LDS (2: the table of the next element is built while the MFMA tasks of this one run)
  const double *weights;
  GeomK geom;
  const double *T;   // [27][nq][4]
  double *values;
  const double *scale;
  double lam, mu, mu2;
  double C[144];     // dense: [c][a][d][b] over the active slots
};

struct Line {  // constants of an owner line (io, jo)
  int io, jo;
  int loI[2], cntI[2], cumI[2];
  int loJ[2], cntJ[2], cumJ[2];
  int SJ, SK;
};

template <int NC, int NS, int MODE>
__global__ __launch_bounds__(NT) void k_p2hex(P2K p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4, li = lane & 15;
  constexpr int NC2 = NC * NC;
  double *E = lds;                     // 3 even-plane buffers
  double *O = E + 3 * p.esz;           // 2 odd-plane buffers
  double *Dt = O + 2 * p.osz;          // 2 D tables, layout [slot][kstep][ntile][lk][li] = D_n[q = 4 kstep + lk][slot], n = 16 ntile + li
  double *Jv = Dt + p.ndb * p.dsz;         // [4 visits][nq][10]: Jinv (row major [j][i]), w |det J| scale
  int *meta = reinterpret_cast<int *>(Jv + 4 * p.nq * 10);  // [8 planes][4 nodes][2]: offset in the plane buffer, scalar row length
  i64 *gmeta = reinterpret_cast<i64 *>(meta + 64);           // [8][4]: offset of the node's first row in the value array

  // zero everything once: row buffers are re-zeroed by the flush, the pad entries of the D tables (q >= nq, n >= 27) stay zero
  {
    const int tot = 3 * p.esz + 2 * p.osz + p.ndb * p.dsz;
    for (int i = tid; i < tot; i += NT) lds[i] = 0.;
  }
  const int KS = p.ks;
  const int nlines = (p.io1 - p.io0 + 1) * (p.n1 + 1);
  auto pbuf = [&](int K) -> double * { return (K & 1) ? O + ((K >> 1) & 1) * p.osz : E + ((K >> 1) % 3) * p.esz; };

  for (int line = blockIdx.x; line < nlines; line += gridDim.x) {
    Line L;
    L.io = p.io0 + line / (p.n1 + 1);
    L.jo = line % (p.n1 + 1);
    L.SJ = 8 * p.n1 + 1;
    L.SK = 8 * p.n2 + 1;
    int vmask = 0;  // valid visits (bit v: element (io - (v >> 1), jo - (v & 1)) exists and contributes)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int ei = L.io - (v >> 1), ej = L.jo - (v & 1);
      if (ei >= p.l0 && ei < p.l1 && ej >= 0 && ej < p.n1) vmask |= 1 << v;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int I = 2 * L.io + a, J = 2 * L.jo + a;
      L.loI[a] = ax_lo(I); L.cntI[a] = ax_cnt(I, p.n0); L.cumI[a] = ax_cum(I, p.n0);
      L.loJ[a] = ax_lo(J); L.cntJ[a] = ax_cnt(J, p.n1); L.cumJ[a] = ax_cum(J, p.n1);
    }
    if (!vmask) continue;
    // node (ai, aj) of this line has contributions iff an element containing it is visited
    auto node_exists = [&](int ai, int aj) -> bool {
      bool ex = false;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int di = v >> 1, dj = v & 1;
        // element (io - di, jo - dj) contains node (2io + ai, 2jo + aj) iff (di ? ai == 0 : true) and (dj ? aj == 0 : true)
        if (((vmask >> v) & 1) && (!di || ai == 0) && (!dj || aj == 0)) ex = true;
      }
      return ex;
    };
    auto plane_meta = [&](int K) {  // one thread: offsets of the 4 nodes of plane K inside its buffer, global offsets
      const int cK = ax_cnt(K, p.n2), cumK = ax_cum(K, p.n2);
      int cur = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ai = j >> 1, aj = j & 1;
        const bool ex = node_exists(ai, aj);
        const int len = ex ? L.cntI[ai] * L.cntJ[aj] * cK : 0;
        const i64 row0 = (i64)L.cumI[ai] * L.SJ * L.SK + (i64)L.cntI[ai] * ((i64)L.cumJ[aj] * L.SK + (i64)L.cntJ[aj] * cumK);
        const i64 goff = row0 * NC2;
        cur = ((cur + 1) & ~1) + (int)(goff & 1);  // blocks never share a 16-byte pair: the flush copies and zeroes whole pairs
        meta[((K & 7) * 4 + j) * 2] = cur;
        meta[((K & 7) * 4 + j) * 2 + 1] = len;
        gmeta[(K & 7) * 4 + j] = goff;
        cur += len * NC2;
      }
    };
    auto flush_plane = [&](int K) {
      double *buf = pbuf(K);
#pragma unroll 1
      for (int j = 0; j < 4; ++j) {
        const int off = meta[((K & 7) * 4 + j) * 2], len = meta[((K & 7) * 4 + j) * 2 + 1];
        if (!len) continue;
        const i64 goff = gmeta[(K & 7) * 4 + j];
        const int par = (int)(goff & 1), nd = len * NC2;
        double *lb = buf + off - par;
        double *gb = p.values + (goff - par);
        const int npairs = (par + nd + 1) >> 1;
        for (int pi = tid; pi < npairs; pi += NT) {
          const v2d v = *reinterpret_cast<const v2d *>(lb + 2 * pi);
          *reinterpret_cast<v2d *>(lb + 2 * pi) = v2d{0., 0.};
          const int e0 = 2 * pi;
          const bool v0 = e0 >= par, v1 = e0 + 1 < par + nd;
          if (v0 && v1) *reinterpret_cast<v2d *>(gb + e0) = v;
          else if (v0) gb[e0] = v[0];
          else if (v1) gb[e0 + 1] = v[1];
        }
      }
    };

    lds_barrier();  // previous line done (meta, buffers)
    if (tid == 0) plane_meta(0);
    for (int k = 0; k < p.n2; ++k) {
      // ---- phase G: meta of the planes entering the ring, geometry of the 4 elements of this slice
      if (tid == 64) plane_meta(2 * k + 1);
      if (tid == 128) plane_meta(2 * k + 2);
      if (tid < 4 * p.nq) {
        const int v = tid / p.nq, q = tid - v * p.nq;
        if ((vmask >> v) & 1) {
          const i64 e = ((i64)(L.io - (v >> 1)) * p.n1 + (L.jo - (v & 1))) * p.n2 + k;
          double Ji[3][3], det;
          geometry_at<3>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
          double *o = Jv + (v * p.nq + q) * 10;
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) o[j * 3 + i] = Ji[j][i];
          o[9] = p.weights[q] * fabs(det) * (p.scale ? p.scale[e * p.nq + q] : 1.);
        }
      }
      lds_barrier();
      int nbuilt = 0;
#pragma unroll 1
      for (int v = 0; v < 4; ++v) {
        if (!((vmask >> v) & 1)) continue;
        double *D = Dt + (p.ndb == 2 ? (nbuilt & 1) * p.dsz : 0);
        if (p.ndb == 1 && nbuilt) lds_barrier();  // single table: the tasks of the previous element have read it
        ++nbuilt;
        // ---- D table of element v: physical derivatives in MFMA operand order (the weight w |J| multiplies the B operand when it is read)
        for (int idx = tid; idx < p.nq * NB; idx += NT) {
          const int q = idx / NB, n = idx - q * NB;
          const double *T4 = p.T + ((i64)n * p.nq + q) * 4;
          const double *Ji = Jv + (v * p.nq + q) * 10;
          double d[4];
          d[0] = T4[0];
#pragma unroll
          for (int i = 0; i < 3; ++i) d[1 + i] = T4[1] * Ji[i] + T4[2] * Ji[3 + i] + T4[3] * Ji[6 + i];
          double *o = D + (((q >> 2) * 2 + (n >> 4)) * 64 + (q & 3) * 16 + (n & 15));
#pragma unroll
          for (int a = 0; a < NS; ++a) {
            const int sl = p.slot[a];
            o[a * KS * 128] = sl == 0 ? d[0] : sl == 1 ? d[1] : sl == 2 ? d[2] : d[3];
          }
        }
        lds_barrier();
        // ---- MFMA tasks: (unit of 4 row nodes, tile of 16 column nodes, half of the k-steps)
        const int di = v >> 1, dj = v & 1;
        const int nunits = v == 0 ? 3 : v == 3 ? 1 : 2;
        const int ei = L.io - di, ej = L.jo - dj;
#pragma unroll 1
        for (int t = wave; t < nunits * 4; t += NT / 64) {
          const int u = t >> 2, nt = (t >> 1) & 1, h = t & 1;
          // local node of row slot mu of this unit: (ai, aj, ak), or invalid
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
          for (int ks = ks0; ks < ks1; ++ks) {
            const double a = D[Aoff + ks * 128];
            const double wq = Jv[(v * p.nq + min(4 * ks + lk, p.nq - 1)) * 10 + 9];  // (pad points: the table entries are zero)
            double bv[NS];
#pragma unroll
            for (int b = 0; b < NS; ++b) bv[b] = wq * D[Boff + (b * KS + ks) * 128];
#pragma unroll
            for (int b = 0; b < NS; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[b], acc[b], 0, 0, 0);
          }
          // ---- this lane: row node mu = lk, column node n = 16 nt + li, G[a][b] = acc[b][a]
          int ai, aj, ak;
          const bool okr = unit_node(lk, ai, aj, ak);
          const int n = nt * 16 + li;
          if (okr && n < NB) {
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
            const int off = meta[((K & 7) * 4 + j) * 2], len = meta[((K & 7) * 4 + j) * 2 + 1];
            const int cK = ax_cnt(K, p.n2), lK = ax_lo(K);
            const int ni = n / 9, nj = (n / 3) % 3, nk = n % 3;
            const int loI = alI ? L.loI[1] : L.loI[0], loJ = alJ ? L.loJ[1] : L.loJ[0], cJ = alJ ? L.cntJ[1] : L.cntJ[0];
            const int pos = ((2 * ei + ni - loI) * cJ + (2 * ej + nj - loJ)) * cK + (2 * k + nk - lK);
            double *row = pbuf(K) + off + pos * NC;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
              for (int d = 0; d < NC; ++d) atomicAdd(row + c * len * NC + d, Kcd[c][d]);
          }
        }
      }
      lds_barrier();  // all contributions of slice k are in LDS
      flush_plane(2 * k);
      flush_plane(2 * k + 1);
    }
    lds_barrier();
    flush_plane(2 * p.n2);
  }
}

}  // namespace

extern "C" {

int nh_p2hex_matrix(const nh_p2hex_args *a, void *stream) {
  NH_REQUIRE(a, "nh_p2hex_matrix: NULL args");
  NH_REQUIRE(a->shape[0] >= 1 && a->shape[1] >= 1 && a->shape[2] >= 1, "nh_p2hex_matrix: shape");
  NH_REQUIRE(a->nq >= 1 && a->nq <= 128 && a->weights_dev && a->T_dev && a->values_dev && a->C_host, "nh_p2hex_matrix: NULL table / weights / values / C");
  NH_REQUIRE(a->ncomp >= 1 && a->ncomp <= 3, "nh_p2hex_matrix: ncomp must be 1..3");
  NH_REQUIRE((i64)(8 * a->shape[0] + 1) * (8 * a->shape[1] + 1) * (8 * a->shape[2] + 1) * a->ncomp * a->ncomp < ((i64)1 << 62), "nh_p2hex_matrix: size");
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
  // active slots: those with a nonzero coefficient on either side
  int ns = 0;
  for (int s = 0; s < S; ++s) {
    bool any = false;
    for (int c = 0; c < nc && !any; ++c)
      for (int d = 0; d < nc && !any; ++d)
        for (int b = 0; b < S && !any; ++b)
          any = a->C_host[((c * S + s) * nc + d) * S + b] != 0. || a->C_host[((c * S + b) * nc + d) * S + s] != 0.;
    if (any) p.slot[ns++] = s;
  }
  NH_REQUIRE(ns > 0, "nh_p2hex_matrix: zero coefficient tensor");
  const int NS = ns <= 3 ? 3 : 4;  // instantiated slot counts (unused slots carry zero coefficients)
  if (ns < NS) {  // pad the slot list with unused slot ids
    for (int s = 0; s < S && ns < NS; ++s) {
      bool used = false;
      for (int i = 0; i < ns; ++i) used |= p.slot[i] == s;
      if (!used) p.slot[ns++] = s;
    }
    std::sort(p.slot, p.slot + NS);
  }
  for (int c = 0; c < nc; ++c)
    for (int sa = 0; sa < NS; ++sa)
      for (int d = 0; d < nc; ++d)
        for (int sb = 0; sb < NS; ++sb) p.C[((c * NS + sa) * nc + d) * NS + sb] = a->C_host[((c * S + p.slot[sa]) * nc + d) * S + p.slot[sb]];
  // isotropic three-parameter family over the gradient slots?
  int mode = 0;
  if (nc == 3 && NS == 3 && p.slot[0] == 1) {
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
  auto lds_bytes = [&](int ndb) { return sizeof(double) * ((size_t)3 * p.esz + 2 * p.osz + ndb * p.dsz + 4 * p.nq * 10) + 64 * sizeof(int) + 32 * sizeof(i64); };
  p.ndb = lds_bytes(2) <= 160 * 1024 ? 2 : 1;
  const size_t ldsb = lds_bytes(p.ndb);
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
#define LAUNCH(NC, NS_, MODE)                                                                                                  \
  do {                                                                                                                         \
    NH_CHECK_HIP(hipFuncSetAttribute((const void *)k_p2hex<NC, NS_, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); \
    hipLaunchKernelGGL((k_p2hex<NC, NS_, MODE>), dim3(grid), dim3(NT), ldsb, s, p);                                            \
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
