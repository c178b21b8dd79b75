// per-CU store throughput vs store width and alignment (one 512-thread workgroup per XCD: G = 8, and all CUs: G = 256)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W>  // doubles per lane per store
__global__ __launch_bounds__(512) void k_store(double *out, long n_per_block, long shift) {
  double *dst = out + (long)blockIdx.x * n_per_block + shift;
  for (long i = (long)threadIdx.x * W; i + W <= n_per_block - 2; i += 512 * W) {
    if (W == 1) dst[i] = 1.0;
    if (W == 2) { double2 v = make_double2(1.0, 2.0); __builtin_memcpy(dst + i, &v, 16); }
    if (W == 4) { double4 v = make_double4(1.0, 2.0, 3.0, 4.0); __builtin_memcpy(dst + i, &v, 32); }
  }
}
int main() {
  const long total = 1L << 30;
  double *buf;
  (void)hipMalloc(&buf, total);
  (void)hipMemset(buf, 0, total);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int G : {8, 256}) {
    const long n = (G <= 32 ? (8L << 20) : (total / G / 2)) / 8;
    for (int w : {1, 2, 4})
      for (long shift : {0L, 1L}) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          (void)hipEventRecord(a);
          if (w == 1) k_store<1><<<G, 512>>>(buf, n, shift);
          if (w == 2) k_store<2><<<G, 512>>>(buf, n, shift);
          if (w == 4) k_store<4><<<G, 512>>>(buf, n, shift);
          (void)hipEventRecord(b);
          (void)hipEventSynchronize(b);
          float ms;
          (void)hipEventElapsedTime(&ms, a, b);
          best = ms < best ? ms : best;
        }
        printf("G=%3d  %d doubles/lane, shift %ld: %.1f GB/s per WG\n", G, w, shift, n * 8 / best / 1e6);
      }
  }
  return 0;
}
