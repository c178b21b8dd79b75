/* ORACLE / CPU BASELINE (test infrastructure, NOT product code).
 *
 * Plain-C port of the reference's algorithm for the headline workload (3-D scalar
 * Laplace stiffness on a structured hex mesh, tensor-product basis, isoparametric or
 * uniform geometry), used (a) as bench.py's cpu_baseline (kind "port") and (b) as a
 * second, independent check of oracle/assemble.py.  It follows the reference's data
 * flow, not the GPU design:
 *   1. per-element loop (evaluable.py:6773-6786; fork-parallel over elements,
 *      parallel.py:128-154 -> OpenMP here): gather vertex coords, J = sum_a X_a (x) dN_a,
 *      inverse + determinant (numeric.py:221-241, evaluable.py:1463-1490), physical
 *      gradients, nb x nb local matrix by contraction over points and dims
 *      (evaluable.py:1885-1886), written to element-major COO (evaluable.py:5322-5343);
 *   2. sparse dedup in the parent, SERIAL like the reference: flat key = row*ncols+col,
 *      STABLE sort (evaluable.py:5560-5652; LSD radix sort here, numpy argsort there),
 *      unique, accumulate in sorted order (numeric.py:434-460), compress_indices
 *      (numeric.py:687-711).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;

static void radix_sort_pairs(uint64_t *key, uint32_t *idx, uint64_t *key2, uint32_t *idx2, i64 n, int bits) {
  /* stable LSD radix sort, 11 bits per pass */
  const int R = 11, B = 1 << R;
  i64 *cnt = (i64 *)malloc(sizeof(i64) * B);
  for (int shift = 0; shift < bits; shift += R) {
    memset(cnt, 0, sizeof(i64) * B);
    for (i64 i = 0; i < n; ++i) cnt[(key[i] >> shift) & (B - 1)]++;
    i64 s = 0;
    for (int b = 0; b < B; ++b) { i64 c = cnt[b]; cnt[b] = s; s += c; }
    for (i64 i = 0; i < n; ++i) {
      i64 p = cnt[(key[i] >> shift) & (B - 1)]++;
      key2[p] = key[i];
      idx2[p] = idx[i];
    }
    uint64_t *tk = key; key = key2; key2 = tk;
    uint32_t *ti = idx; idx = idx2; idx2 = ti;
  }
  free(cnt);
  /* result may live in either buffer: copy back if needed */
  int passes = (bits + R - 1) / R;
  if (passes & 1) { memcpy(key2, key, sizeof(uint64_t) * n); memcpy(idx2, idx, sizeof(uint32_t) * n); }
}

/* shape[3] elements per axis, p = basis degree per axis for the 'std' basis with nloc = p+1 dofs per axis and
 * start dof = i*p; T[nb][nq][4] tabulated basis (value, d/dxi); gT[8][nq][4] tabulated trilinear geometry basis;
 * verts[(n0+1)(n1+1)(n2+1)][3] or NULL for the uniform unit-cell geometry; weights[nq].
 * Outputs: values[cap], rowptr[ndofs+1], colidx[cap]; returns nnz (or -1 if cap is too small).
 * timings[0] = element loop seconds, timings[1] = dedup seconds. */
i64 port_laplace3d(const int *shape, int p, int nq, const double *T, const double *gT, const double *verts, const double *weights,
                   double *values, i64 *rowptr, i64 *colidx, i64 cap, int threads, double *timings) {
  const int nl = p + 1, nb = nl * nl * nl;
  const i64 n0 = shape[0], n1 = shape[1], n2 = shape[2], ne = n0 * n1 * n2;
  const i64 N1 = n1 * p + 1, N2 = n2 * p + 1, ndofs = (n0 * p + 1) * N1 * N2;
  const i64 V1 = n1 + 1, V2 = n2 + 1;
  const i64 ncoo = ne * nb * nb;
  double *cv = (double *)malloc(sizeof(double) * ncoo);
  uint64_t *key = (uint64_t *)malloc(sizeof(uint64_t) * ncoo), *key2 = (uint64_t *)malloc(sizeof(uint64_t) * ncoo);
  uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * ncoo), *idx2 = (uint32_t *)malloc(sizeof(uint32_t) * ncoo);
  if (!cv || !key || !key2 || !idx || !idx2 || ncoo >= 4294967296LL) return -2;
#ifdef _OPENMP
  double t0 = omp_get_wtime();
  if (threads > 0) omp_set_num_threads(threads);
#else
  double t0 = 0;
#endif
#pragma omp parallel
  {
    double *G = (double *)malloc(sizeof(double) * nq * nb * 3);
    double *wd = (double *)malloc(sizeof(double) * nq);
    i64 *dofs = (i64 *)malloc(sizeof(i64) * nb);
#pragma omp for schedule(dynamic, 256)
    for (i64 e = 0; e < ne; ++e) {
      const i64 i = e / (n1 * n2), j = (e / n2) % n1, k = e % n2;
      for (int a = 0; a < nl; ++a)
        for (int b = 0; b < nl; ++b)
          for (int c = 0; c < nl; ++c) dofs[(a * nl + b) * nl + c] = ((i * p + a) * N1 + (j * p + b)) * N2 + (k * p + c);
      double X[8][3];
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          for (int c = 0; c < 2; ++c) {
            const i64 v = ((i + a) * V1 + (j + b)) * V2 + (k + c);
            for (int d = 0; d < 3; ++d) X[(a * 2 + b) * 2 + c][d] = verts ? verts[v * 3 + d] : (double)((d == 0 ? i + a : d == 1 ? j + b : k + c));
          }
      for (int q = 0; q < nq; ++q) {
        double J[3][3] = {{0}};
        for (int a = 0; a < 8; ++a)
          for (int r = 0; r < 3; ++r)
            for (int s = 0; s < 3; ++s) J[r][s] += X[a][r] * gT[(a * nq + q) * 4 + 1 + s];
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2],
                     c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, r = 1. / det;
        double Ji[3][3];
        Ji[0][0] = c00 * r; Ji[1][0] = c01 * r; Ji[2][0] = c02 * r;
        Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
        Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
        Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
        Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
        Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
        Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
        wd[q] = weights[q] * fabs(det);
        for (int m = 0; m < nb; ++m)
          for (int s = 0; s < 3; ++s) {
            double g = 0;
            for (int t = 0; t < 3; ++t) g += T[(m * nq + q) * 4 + 1 + t] * Ji[t][s];
            G[(q * nb + m) * 3 + s] = g;
          }
      }
      const i64 base = e * nb * nb;
      for (int m = 0; m < nb; ++m)
        for (int n = 0; n < nb; ++n) {
          double acc = 0;
          for (int q = 0; q < nq; ++q) {
            const double *gm = G + (q * nb + m) * 3, *gn = G + (q * nb + n) * 3;
            acc += wd[q] * (gm[0] * gn[0] + gm[1] * gn[1] + gm[2] * gn[2]);
          }
          cv[base + m * nb + n] = acc;
          key[base + m * nb + n] = (uint64_t)(dofs[m] * ndofs + dofs[n]);
          idx[base + m * nb + n] = (uint32_t)(base + m * nb + n);
        }
    }
    free(G); free(wd); free(dofs);
  }
#ifdef _OPENMP
  double t1 = omp_get_wtime();
#else
  double t1 = 0;
#endif
  int bits = 1;
  while (bits < 64 && ((uint64_t)1 << bits) < (uint64_t)(ndofs * ndofs)) ++bits;
  radix_sort_pairs(key, idx, key2, idx2, ncoo, bits);
  /* after the call the sorted data is in (key, idx) when passes is even, else copied into key2/idx2 */
  const int passes = (bits + 10) / 11;
  const uint64_t *sk = (passes & 1) ? key2 : key;
  const uint32_t *si = (passes & 1) ? idx2 : idx;
  i64 nnz = 0;
  i64 row_prev = -1;
  for (i64 t = 0; t < ncoo; ++t) {
    if (t == 0 || sk[t] != sk[t - 1]) {
      if (nnz >= cap) { nnz = -1; break; }
      const i64 row = (i64)(sk[t] / (uint64_t)ndofs), col = (i64)(sk[t] % (uint64_t)ndofs);
      while (row_prev < row) rowptr[++row_prev] = nnz;
      colidx[nnz] = col;
      values[nnz] = 0;
      ++nnz;
    }
    values[nnz - 1] += cv[si[t]];
  }
  if (nnz >= 0)
    while (row_prev < ndofs) rowptr[++row_prev] = nnz;
#ifdef _OPENMP
  double t2 = omp_get_wtime();
#else
  double t2 = 0;
#endif
  if (timings) { timings[0] = t1 - t0; timings[1] = t2 - t1; }
  free(cv); free(key); free(key2); free(idx); free(idx2);
  return nnz;
}


/* Vector-valued constant-coefficient form on the same kind of mesh (BASELINE.json configs[2]: 3-D elasticity, p = 2, nc = 3):
 *   A[(m,c),(n,d)] = sum_q w_q |J_q| sum_ab D[q,m,a] C[c,a,d,b] D[q,n,b],  D[.,.,0] = value, D[.,.,1+i] = d/dx_i,
 * flat dof = node * nc + comp (function.py:2598-2627), COO emitted per element with m slowest, then c, n, d
 * (evaluable.py:5322-5343), then the same serial stable-sort dedup.  C[nc][4][nc][4]; all nc x nc blocks are kept. */
i64 port_form3d(const int *shape, int p, int nc, const double *C, int nq, const double *T, const double *gT, const double *verts,
                const double *weights, double *values, i64 *rowptr, i64 *colidx, i64 cap, int threads, double *timings) {
  const int nl = p + 1, nb = nl * nl * nl, nloc = nb * nc;
  const i64 n0 = shape[0], n1 = shape[1], n2 = shape[2], ne = n0 * n1 * n2;
  const i64 N1 = n1 * p + 1, N2 = n2 * p + 1, ndofs = (n0 * p + 1) * N1 * N2 * nc;
  const i64 V1 = n1 + 1, V2 = n2 + 1;
  const i64 ncoo = ne * nloc * nloc;
  if (ncoo >= 4294967296LL) return -2;
  double *cv = (double *)malloc(sizeof(double) * ncoo);
  uint64_t *key = (uint64_t *)malloc(sizeof(uint64_t) * ncoo), *key2 = (uint64_t *)malloc(sizeof(uint64_t) * ncoo);
  uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * ncoo), *idx2 = (uint32_t *)malloc(sizeof(uint32_t) * ncoo);
  if (!cv || !key || !key2 || !idx || !idx2) return -2;
#ifdef _OPENMP
  double t0 = omp_get_wtime();
  if (threads > 0) omp_set_num_threads(threads);
#else
  double t0 = 0;
#endif
#pragma omp parallel
  {
    double *D = (double *)malloc(sizeof(double) * nq * nb * 4);
    double *W = (double *)malloc(sizeof(double) * nq * nb * nc * nc * 4); /* W[q][n][d][c][a] = w|J| sum_b C[c,a,d,b] D[q,n,b] */
    i64 *dofs = (i64 *)malloc(sizeof(i64) * nb);
#pragma omp for schedule(dynamic, 16)
    for (i64 e = 0; e < ne; ++e) {
      const i64 i = e / (n1 * n2), j = (e / n2) % n1, k = e % n2;
      for (int a = 0; a < nl; ++a)
        for (int b = 0; b < nl; ++b)
          for (int c = 0; c < nl; ++c) dofs[(a * nl + b) * nl + c] = ((i * p + a) * N1 + (j * p + b)) * N2 + (k * p + c);
      double X[8][3];
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          for (int c = 0; c < 2; ++c) {
            const i64 v = ((i + a) * V1 + (j + b)) * V2 + (k + c);
            for (int d = 0; d < 3; ++d) X[(a * 2 + b) * 2 + c][d] = verts ? verts[v * 3 + d] : (double)((d == 0 ? i + a : d == 1 ? j + b : k + c));
          }
      for (int q = 0; q < nq; ++q) {
        double J[3][3] = {{0}};
        for (int a = 0; a < 8; ++a)
          for (int r = 0; r < 3; ++r)
            for (int s = 0; s < 3; ++s) J[r][s] += X[a][r] * gT[(a * nq + q) * 4 + 1 + s];
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2],
                     c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, r = 1. / det;
        double Ji[3][3];
        Ji[0][0] = c00 * r; Ji[1][0] = c01 * r; Ji[2][0] = c02 * r;
        Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
        Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
        Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
        Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
        Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
        Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
        const double wd = weights[q] * fabs(det);
        for (int m = 0; m < nb; ++m) {
          double *d = D + (q * nb + m) * 4;
          d[0] = T[(m * nq + q) * 4];
          for (int s = 0; s < 3; ++s) {
            double g = 0;
            for (int t = 0; t < 3; ++t) g += T[(m * nq + q) * 4 + 1 + t] * Ji[t][s];
            d[1 + s] = g;
          }
          for (int dd = 0; dd < nc; ++dd)
            for (int c = 0; c < nc; ++c)
              for (int a = 0; a < 4; ++a) {
                double w = 0;
                for (int b = 0; b < 4; ++b) w += C[((c * 4 + a) * nc + dd) * 4 + b] * d[b];
                W[(((q * nb + m) * nc + dd) * nc + c) * 4 + a] = wd * w;
              }
        }
      }
      const i64 base = e * nloc * nloc;
      for (int m = 0; m < nb; ++m)
        for (int c = 0; c < nc; ++c)
          for (int n = 0; n < nb; ++n)
            for (int dd = 0; dd < nc; ++dd) {
              double acc = 0;
              for (int q = 0; q < nq; ++q) {
                const double *dm = D + (q * nb + m) * 4, *w = W + (((q * nb + n) * nc + dd) * nc + c) * 4;
                acc += dm[0] * w[0] + dm[1] * w[1] + dm[2] * w[2] + dm[3] * w[3];
              }
              const i64 o = base + ((i64)(m * nc + c) * nb + n) * nc + dd;
              cv[o] = acc;
              key[o] = (uint64_t)((dofs[m] * nc + c) * ndofs + dofs[n] * nc + dd);
              idx[o] = (uint32_t)o;
            }
    }
    free(D); free(W); free(dofs);
  }
#ifdef _OPENMP
  double t1 = omp_get_wtime();
#else
  double t1 = 0;
#endif
  int bits = 1;
  while (bits < 64 && ((uint64_t)1 << bits) < (uint64_t)(ndofs * ndofs)) ++bits;
  radix_sort_pairs(key, idx, key2, idx2, ncoo, bits);
  const int passes = (bits + 10) / 11;
  const uint64_t *sk = (passes & 1) ? key2 : key;
  const uint32_t *si = (passes & 1) ? idx2 : idx;
  i64 nnz = 0, row_prev = -1;
  for (i64 t = 0; t < ncoo; ++t) {
    if (t == 0 || sk[t] != sk[t - 1]) {
      if (nnz >= cap) { nnz = -1; break; }
      const i64 row = (i64)(sk[t] / (uint64_t)ndofs), col = (i64)(sk[t] % (uint64_t)ndofs);
      while (row_prev < row) rowptr[++row_prev] = nnz;
      colidx[nnz] = col;
      values[nnz] = 0;
      ++nnz;
    }
    values[nnz - 1] += cv[si[t]];
  }
  if (nnz >= 0)
    while (row_prev < ndofs) rowptr[++row_prev] = nnz;
#ifdef _OPENMP
  double t2 = omp_get_wtime();
#else
  double t2 = 0;
#endif
  if (timings) { timings[0] = t1 - t0; timings[1] = t2 - t1; }
  free(cv); free(key); free(key2); free(idx); free(idx2);
  return nnz;
}
