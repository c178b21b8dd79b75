// How fast can ONE CU store when the rest of the chip is quiet?  G workgroups (one per CU, G = 1 .. 256) of W waves; every wave streams
// 16-byte-per-lane stores (1 KB per instruction, contiguous) into its own region.  If the per-CU rate at small G is far above 1/256 of the
// chip-wide rate, a kernel whose CUs store in bursts is limited by how many CUs burst at once, not by the CU's own path.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2), aligned(8)));
__global__ __launch_bounds__(1024) void k(char *out, int iters, size_t per_wave) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  char *base = out + ((size_t)(blockIdx.x * nw + wave)) * per_wave + 16 * lane;
  const d2v v = {1.0, 2.0};
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<d2v *>(base + (size_t)(i + q) * 1024) = v;
  }
}
int main() {
  const size_t total = 256ull * (16u << 20);
  char *buf;
  (void)hipMalloc(&buf, total + 4096);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int W : {1, 4})
    for (int G : {1, 2, 8, 16, 32, 64, 128, 256}) {
      const size_t per_wave = (16u << 20) / W;  // 16 MB per CU
      const int iters = (int)(per_wave / 1024);
      for (int w = 0; w < 3; ++w) k<<<G, 64 * W>>>(buf, iters, per_wave);
      (void)hipEventRecord(a);
      for (int w = 0; w < 5; ++w) k<<<G, 64 * W>>>(buf, iters, per_wave);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      const double bytes = 5.0 * G * W * (double)iters * 1024;
      printf("waves/CU %d  CUs %3d : %.3f TB/s  %.1f GB/s per CU\n", W, G, bytes / ms * 1e-9, bytes / ms * 1e-6 / G);
    }
  return 0;
}
