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
constexpr int TABARG = 256;  // doubles of the term table that fit the kernel arguments

__device__ __forceinline__ i64 boff(const BasisK &b, i64 e) { return b.off ? b.off[e] : e * (i64)b.nb; }
__device__ __forceinline__ int bnb(const BasisK &b, i64 e) { return b.off ? (int)(b.off[e + 1] - b.off[e]) : b.nb; }
__device__ __forceinline__ i64 bfn(const BasisK &b, i64 e) { return b.off ? b.off[e] : (b.tab ? (i64)b.tab[e] * b.nb : 0); }

struct FieldK {
  BasisK b;
  const double *u;
  int ncomp, c0;  // c0: first slot of this field in the per-point value table
};
struct BlockK {
  BasisK test;
  double *out;
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
  int toff[MAXT], poff[MAXP];
  const double *table;  // NULL: the table is tabarg (small tables travel with the kernel arguments: no copy, no buffer to keep alive)
  int tlen;
  double tabarg[TABARG];
  int rowsper;  // sum over blocks of maxnb * nct: output lanes per element
};

template <int ND>
__global__ __launch_bounds__(NTB) void k_terms(TermsK p) {
  constexpr int S = 1 + ND;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *tab = lds;                                // term table
  double *G = lds + p.tlen;                         // [eb * nq][ct][S]: integrand, then its reference form times w |J|
  double *U = G + (size_t)p.eb * p.nq * p.ct * S;   // [NTB][fct][S]: field values of this thread's point
  const int tid = threadIdx.x;
  for (int i = tid; i < p.tlen; i += NTB) tab[i] = p.table ? p.table[i] : p.tabarg[i];
  const int npts = p.eb * p.nq;
  for (i64 b0 = (i64)blockIdx.x * p.eb; b0 < p.nelems; b0 += (i64)gridDim.x * p.eb) {
    __syncthreads();  // table staged; G of the previous batch consumed
    for (int t = tid; t < npts; t += NTB) {
      const int el = t / p.nq, q = t - el * p.nq;
      const i64 ie = b0 + el;
      double *g = G + (size_t)t * p.ct * S;
      if (ie >= p.nelems) continue;
      const i64 e = p.elist ? p.elist[ie] : ie;
      double Ji[ND][ND], det;
      geometry_at<ND>(p.geom, e, q, p.nq, nullptr, Ji, det, nullptr);
      const double wdet = p.weights[q] * fabs(det);
      // field values and physical gradients at the point
      double *u = U + (size_t)tid * p.fct * S;
      for (int f = 0; f < p.nfields; ++f) {
        const FieldK &F = p.fields[f];
        const int nb = bnb(F.b, e);
        const i64 d0 = boff(F.b, e);
        const double *T = F.b.T + (bfn(F.b, e) * p.nq + q) * S;
        for (int d = 0; d < F.ncomp; ++d) {
          double r[S];
#pragma unroll
          for (int s = 0; s < S; ++s) r[s] = 0;
          for (int n = 0; n < nb; ++n) {
            const double un = F.u[(i64)F.b.dofs[d0 + n] * F.ncomp + d];
            const double *Tn = T + (size_t)n * p.nq * S;
#pragma unroll
            for (int s = 0; s < S; ++s) r[s] += Tn[s] * un;
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
      // pointwise polynomial factors
      double pv[MAXP];
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        pv[k] = 1.;
        if (k < p.npolys) {
          const double *P = tab + p.poff[k];
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
      // integrand of every block: sum of the terms
      for (int i = 0; i < p.ct * S; ++i) g[i] = 0.;
      for (int t2 = 0; t2 < p.nterms; ++t2) {
        const double *H = tab + p.toff[t2];
        const int blk = (int)H[0], fld = (int)H[1], pol = (int)H[2], hasC = (int)H[3], hasf = (int)H[4];
        const BlockK &B = p.blocks[blk];
        double coef = p.scale[t2] ? p.scale[t2][ie * p.nq + q] : 1.;
        if (pol >= 0) coef *= pol == 0 ? pv[0] : pol == 1 ? pv[1] : pol == 2 ? pv[2] : pv[3];
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
      const i64 e = p.elist ? p.elist[ie] : ie;
      int blk = 0;
      if (p.nblocks > 1 && r >= p.blocks[0].maxnb * p.blocks[0].nct) r -= p.blocks[0].maxnb * p.blocks[0].nct, blk = 1;
      const BlockK &B = p.blocks[blk];
      const int m = r / B.nct, c = r - m * B.nct;
      if (m >= bnb(B.test, e)) continue;
      const double *T = B.test.T + (bfn(B.test, e) + m) * p.nq * S;
      const double *g = G + ((size_t)el * p.nq * p.ct + (B.c0 + c)) * S;
      double acc = 0;
      for (int q = 0; q < p.nq; ++q) {
#pragma unroll
        for (int s = 0; s < S; ++s) acc += T[q * S + s] * g[(size_t)q * p.ct * S + s];
      }
      atomicAdd(B.out + (i64)B.test.dofs[boff(B.test, e) + m] * B.nct + c, acc);
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

int max_nb2(const nh_basis &b, i64 nelems, int *out) {
  if (b.nb > 0 || !b.off_dev) {
    *out = b.nb;
    return NH_OK;
  }
  std::vector<i64> h(nelems + 1);
  NH_CHECK_HIP(hipMemcpy(h.data(), b.off_dev, sizeof(i64) * (nelems + 1), hipMemcpyDeviceToHost));
  i64 m = 0;
  for (i64 e = 0; e < nelems; ++e) m = std::max(m, h[e + 1] - h[e]);
  *out = (int)m;
  return NH_OK;
}

}  // namespace

extern "C" int nh_assemble_terms(const nh_terms_args *a, void *stream) {
  NH_REQUIRE(a, "nh_assemble_terms: NULL args");
  NH_REQUIRE(a->ndims >= 1 && a->ndims <= 3, "ndims must be 1..3");
  NH_REQUIRE(a->nq >= 1 && a->weights_dev, "quadrature missing");
  NH_REQUIRE(a->nfields >= 0 && a->nfields <= MAXF && (a->nfields == 0 || a->fields), "nh_assemble_terms: 0..%d fields", MAXF);
  NH_REQUIRE(a->nblocks >= 1 && a->nblocks <= MAXB && a->blocks, "nh_assemble_terms: 1..%d output blocks", MAXB);
  NH_REQUIRE(a->nterms >= 1 && a->nterms <= MAXT && a->terms, "nh_assemble_terms: 1..%d terms", MAXT);
  NH_REQUIRE(a->npolys >= 0 && a->npolys <= MAXP && (a->npolys == 0 || a->polys), "nh_assemble_terms: 0..%d pointwise polynomials", MAXP);
  int rc = check_geom2(a->geom);
  if (rc) return rc;
  if (a->nelems == 0) return NH_OK;
  const int S = 1 + a->ndims;
  TermsK p;
  p.nelems = a->nelems;
  p.elist = a->elist_dev;
  p.nq = a->nq;
  p.weights = a->weights_dev;
  p.geom = to_k(a->geom);
  p.nfields = a->nfields, p.nblocks = a->nblocks, p.nterms = a->nterms, p.npolys = a->npolys;
  p.fct = 0;
  for (int f = 0; f < a->nfields; ++f) {
    const nh_field &F = a->fields[f];
    NH_REQUIRE(F.basis.T_dev && F.basis.dofs_dev && F.u_dev && F.ncomp >= 1 && F.ncomp <= 3, "nh_assemble_terms: field %d incomplete", f);
    NH_REQUIRE(!a->elist_dev || !F.basis.off_dev, "elist with ragged bases is not supported");
    p.fields[f].b = to_k(F.basis);
    p.fields[f].u = F.u_dev;
    p.fields[f].ncomp = F.ncomp;
    p.fields[f].c0 = p.fct;
    p.fct += F.ncomp;
  }
  NH_REQUIRE(p.fct <= MAXFC, "nh_assemble_terms: more than %d field components", MAXFC);
  p.ct = 0, p.rowsper = 0;
  for (int b = 0; b < a->nblocks; ++b) {
    const nh_block &B = a->blocks[b];
    NH_REQUIRE(B.test.T_dev && B.test.dofs_dev && B.out_dev && B.nct >= 1 && B.nct <= 3, "nh_assemble_terms: block %d incomplete", b);
    NH_REQUIRE(!a->elist_dev || !B.test.off_dev, "elist with ragged bases is not supported");
    p.blocks[b].test = to_k(B.test);
    p.blocks[b].out = B.out_dev;
    p.blocks[b].nct = B.nct;
    p.blocks[b].c0 = p.ct;
    if ((rc = max_nb2(B.test, a->nelems, &p.blocks[b].maxnb)) != NH_OK) return rc;
    p.ct += B.nct;
    p.rowsper += p.blocks[b].maxnb * B.nct;
  }
  NH_REQUIRE(p.ct <= MAXCT, "nh_assemble_terms: more than %d test components", MAXCT);
  std::vector<double> tab;
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
  }
  for (int t = a->nterms; t < MAXT; ++t) p.scale[t] = nullptr, p.toff[t] = 0;
  for (int k = 0; k < MAXP; ++k) p.poff[k] = 0;
  for (int k = 0; k < a->npolys; ++k) {
    const nh_point_poly &P = a->polys[k];
    NH_REQUIRE(P.nvars >= 0 && P.nvars <= 4 && P.nterms >= 0 && P.nterms <= 64 && (P.nterms == 0 || (P.coeffs_host && (P.nvars == 0 || P.powers_host))), "nh_assemble_terms: polynomial %d: at most 4 variables and 64 terms", k);
    p.poff[k] = (int)tab.size();
    tab.push_back(P.nvars), tab.push_back(P.nterms);
    for (int v = 0; v < 4; ++v) {
      int slot = 0;
      if (v < P.nvars) {
        NH_REQUIRE(P.field[v] >= 0 && P.field[v] < a->nfields && P.comp[v] >= 0 && P.comp[v] < a->fields[P.field[v]].ncomp, "nh_assemble_terms: polynomial %d: variable %d refers to a missing field component", k, v);
        slot = p.fields[P.field[v]].c0 + P.comp[v];
      }
      tab.push_back(slot);
    }
    for (int t = 0; t < P.nterms; ++t) {
      tab.push_back(P.coeffs_host[t]);
      for (int v = 0; v < 4; ++v) {
        const int pw = v < P.nvars ? P.powers_host[t * P.nvars + v] : 0;
        NH_REQUIRE(pw >= 0 && pw < 64, "nh_assemble_terms: power out of range");
        tab.push_back(pw);
      }
    }
  }
  p.tlen = (int)tab.size();
  // elements per batch: as many as fill the workgroup in the pointwise phase
  p.eb = std::max(1, NTB / a->nq);
  const size_t lds = sizeof(double) * ((size_t)p.tlen + (size_t)p.eb * a->nq * p.ct * S + (size_t)NTB * std::max(p.fct, 1) * S);
  NH_REQUIRE(lds <= 160 * 1024, "nh_assemble_terms: batch too large for LDS (%zu bytes)", lds);
  hipStream_t s = nh_stream(stream);
  if (tab.size() <= (size_t)TABARG) {
    p.table = nullptr;
    std::copy(tab.begin(), tab.end(), p.tabarg);
  } else {
    // large tables go through a device buffer owned by the library; the pageable host vector must outlive the copy: wait for it
    static double *dtab = nullptr;
    static size_t dcap = 0;
    if (tab.size() > dcap) {
      if (dtab) NH_CHECK_HIP(hipFree(dtab));
      dcap = 2 * tab.size();
      NH_CHECK_HIP(hipMalloc((void **)&dtab, dcap * sizeof(double)));
    }
    NH_CHECK_HIP(hipStreamSynchronize(s));  // earlier launches on this stream may still read the buffer
    NH_CHECK_HIP(hipMemcpy(dtab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    p.table = dtab;
  }
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
