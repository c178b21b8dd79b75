// v_mfma_f64_4x4x4_4b_f64 on gfx950: issue rate, overlap with VALU of another wave on the same SIMD, operand layout probe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int X>
__global__ __launch_bounds__(512) void k(double *out, int iters, int mode) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double r = 0;
  if (wave < 4) {
    if (mode & 1) {
      double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      double a = lane * 1e-3, b = lane * 2e-3;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[j], 0, 0, 0);
      for (int j = 0; j < 9; ++j) r += acc[j];
    }
  } else if (mode & 2) {
    if (X == 0) {
      double x[8];
      for (int i = 0; i < 8; ++i) x[i] = lane + i;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = fma(x[i], 1.0000001, 1e-9);
      for (int i = 0; i < 8; ++i) r += x[i];
    } else {
      int x[8];
      for (int i = 0; i < 8; ++i) x[i] = lane + i;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = x[i] * 3 + it;
      for (int i = 0; i < 8; ++i) r += x[i];
    }
  }
  if (r == 1.2345) out[0] = r;
}
__global__ void probe(const double *A, const double *B, double *D) {  // one wave: D = mfma(A[lane], B[lane], 0)
  const int lane = threadIdx.x;
  D[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[lane], B[lane], 0., 0, 0, 0);
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
  float m = run<0>(1, iters);
  printf("4x4x4 MFMA only (9 per iter): %.3f ms -> %.1f cycles per MFMA @2.4GHz\n", m, m * 1e-3 * 2.4e9 / (iters * 9.));
  float x0 = run<0>(2, iters), b0 = run<0>(3, iters);
  printf("f64 fma: alone %.3f ms, with MFMA %.3f ms\n", x0, b0);
  float x1 = run<1>(2, iters), b1 = run<1>(3, iters);
  printf("int mad: alone %.3f ms, with MFMA %.3f ms\n", x1, b1);
  // layout probe: A one-hot at lane la, B one-hot at lane lb -> which D lanes light up
  double *dA, *dB, *dD; (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 512);
  std::vector<double> hA(64), hB(64), hD(64);
  for (int la : {0, 1, 4, 5, 16, 17, 21}) {
    printf("A one-hot lane %2d: ", la);
    for (int lb = 0; lb < 64; ++lb) {
      std::fill(hA.begin(), hA.end(), 0.); std::fill(hB.begin(), hB.end(), 0.);
      hA[la] = 1.; hB[lb] = 1.;
      (void)hipMemcpy(dA, hA.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB.data(), 512, hipMemcpyHostToDevice);
      probe<<<1, 64>>>(dA, dB, dD);
      (void)hipMemcpy(hD.data(), dD, 512, hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l) if (hD[l] != 0.) printf("B%d->D%d ", lb, l);
    }
    printf("\n");
  }
  return 0;
}
