// v_fmac_f64_dpp row_newbcast:N on gfx950: semantics (lane N of each row of 16 lanes is the multiplicand of all lanes of the row) and issue rate
// against a plain v_fma_f64.   hipcc -O3 --offload-arch=gfx950 dpp_bcast.hip -o dpp_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int N> __device__ __forceinline__ void fmac_bc(double &acc, double x, double t) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(t), "n"(N));
}
__global__ void k_sem(const double *in, const double *t, double *out) {
  const int l = threadIdx.x;
  const double x = in[l], tv = t[l];
  double a3 = 1., a9 = 2., a15 = 3.;
  fmac_bc<3>(a3, x, tv);
  fmac_bc<9>(a9, x, tv);
  fmac_bc<15>(a15, x, tv);
  out[l] = a3; out[64 + l] = a9; out[128 + l] = a15;
}
template <int DPP>
__global__ __launch_bounds__(256) void k_rate(double *out, int iters) {
  double x = threadIdx.x * 1e-3, t = 1.0000001;
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (DPP) {
        fmac_bc<0>(a[0], x, t); fmac_bc<1>(a[1], x, t); fmac_bc<2>(a[2], x, t); fmac_bc<3>(a[3], x, t);
        fmac_bc<4>(a[4], x, t); fmac_bc<5>(a[5], x, t); fmac_bc<6>(a[6], x, t); fmac_bc<7>(a[7], x, t);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(t));
      }
    }
  }
  double r = 0;
  for (int i = 0; i < 8; ++i) r += a[i];
  if (r == 1.2345) out[0] = r;
}
int main() {
  double h[64], ht[64], ho[192], *d, *dt, *dout;
  for (int i = 0; i < 64; ++i) { h[i] = 100 + i; ht[i] = 1 + i * 0.5; }
  (void)hipMalloc(&d, 512); (void)hipMalloc(&dt, 512); (void)hipMalloc(&dout, 1536);
  (void)hipMemcpy(d, h, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dt, ht, 512, hipMemcpyHostToDevice);
  k_sem<<<1, 64>>>(d, dt, dout);
  (void)hipMemcpy(ho, dout, 1536, hipMemcpyDeviceToHost);
  int bad = 0;
  const int ns[3] = {3, 9, 15};
  for (int s = 0; s < 3; ++s)
    for (int l = 0; l < 64; ++l) {
      const double expect = (s + 1) + h[(l & ~15) + ns[s]] * ht[l];
      if (ho[s * 64 + l] != expect) { if (bad < 5) printf("lane %d bcast %d: got %g expect %g\n", l, ns[s], ho[s * 64 + l], expect); ++bad; }
    }
  printf("semantics: %s (%d mismatches)\n", bad ? "DIFFERENT" : "row_newbcast:N = lane N of the own row of 16", bad);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int dpp = 0; dpp < 2; ++dpp) {
    const int iters = 20000;
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (dpp) k_rate<1><<<256, 256>>>(dout, iters); else k_rate<0><<<256, 256>>>(dout, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%s: %.3f ms -> %.2f cycles per wave instruction @2.4 GHz (one wave per SIMD)\n", dpp ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64", ms, ms * 1e-3 * 2.4e9 / (iters * 32.));
  }
  return bad != 0;
}
