// Does f64 VALU / int VALU / LDS work of one wave overlap with v_mfma_f64_16x16x4_f64 of another wave on the same SIMD (gfx950)?
// 8 waves per CU: waves 0-3 run MFMAs (mode & 1), waves 4-7 run op X (mode & 2); kernel time for MFMA only, X only, both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int X>
__global__ __launch_bounds__(512) void k(double *out, int iters, int mode) {
  __shared__ double lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double r = 0;
  if (wave < 4) {
    if (mode & 1) {
      v4d acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      double a = lane * 1e-3, b = lane * 2e-3;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
      r = acc[0][0] + acc[1][1] + acc[2][2];
    }
  } else if (mode & 2) {
    if (X == 0) {  // f64 FMA, 8 independent chains
      double x[8];
      for (int i = 0; i < 8; ++i) x[i] = lane + i;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = fma(x[i], 1.0000001, 1e-9);
      for (int i = 0; i < 8; ++i) r += x[i];
    } else if (X == 1) {  // int32 VALU
      int x[8];
      for (int i = 0; i < 8; ++i) x[i] = lane + i;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = x[i] * 3 + it;
      for (int i = 0; i < 8; ++i) r += x[i];
    } else if (X == 2) {  // LDS reads
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) r += lds[(lane + u * 64 + it) & 8191];
      }
    } else if (X == 3) {  // f32 FMA
      float x[8];
      for (int i = 0; i < 8; ++i) x[i] = lane + i;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 1.0000001f, 1e-9f);
      for (int i = 0; i < 8; ++i) r += x[i];
    }
  }
  if (r == 1.2345) out[0] = r;
}
template <int X>
float run(int mode, int iters) {
  double *out; (void)hipMalloc(&out, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<X><<<256, 512>>>(out, 10, mode);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<X><<<256, 512>>>(out, iters, mode);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(out);
  return ms;
}
int main() {
  const int iters = 20000;
  const char *names[] = {"f64 fma", "int mad", "lds read", "f32 fma"};
  float m = run<0>(1, iters);
  printf("MFMA only (12 per iter): %.3f ms -> %.1f cycles per MFMA @2.4GHz\n", m, m * 1e-3 * 2.4e9 / (iters * 12.));
  float x0 = run<0>(2, iters), b0 = run<0>(3, iters);
  printf("%-9s: alone %.3f ms (%.1f cycles/instr), with MFMA %.3f ms\n", names[0], x0, x0 * 1e-3 * 2.4e9 / (iters * 32.), b0);
  float x1 = run<1>(2, iters), b1 = run<1>(3, iters);
  printf("%-9s: alone %.3f ms (%.1f cycles/instr), with MFMA %.3f ms\n", names[1], x1, x1 * 1e-3 * 2.4e9 / (iters * 32.), b1);
  float x2 = run<2>(2, iters), b2 = run<2>(3, iters);
  printf("%-9s: alone %.3f ms (%.1f cycles/instr), with MFMA %.3f ms\n", names[2], x2, x2 * 1e-3 * 2.4e9 / (iters * 32.), b2);
  float x3 = run<3>(2, iters), b3 = run<3>(3, iters);
  printf("%-9s: alone %.3f ms (%.1f cycles/instr), with MFMA %.3f ms\n", names[3], x3, x3 * 1e-3 * 2.4e9 / (iters * 32.), b3);
  return 0;
}
