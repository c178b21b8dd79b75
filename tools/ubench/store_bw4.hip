// Does the write rate of a CU depend on HOW MANY WAVES store?  256 workgroups (one per CU) of W waves; every wave streams 16-byte-per-lane
// stores (1 KB per instruction, contiguous) into its own region, `align` bytes off a 16-byte boundary.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2), aligned(8)));
__global__ __launch_bounds__(1024) void k(char *out, int iters, int off, size_t per_wave) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  char *base = out + ((size_t)(blockIdx.x * nw + wave)) * per_wave + off + 16 * lane;
  const d2v v = {1.0, 2.0};
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<d2v *>(base + (size_t)(i + q) * 1024) = v;
  }
}
int main() {
  const size_t total = 1ull << 30;  // 1 GiB written per launch
  char *buf;
  (void)hipMalloc(&buf, total + 4096);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int off : {0, 8})
    for (int W : {1, 2, 3, 4, 6, 8, 16}) {
      const size_t per_wave = total / (256 * W) / 8192 * 8192;
      const int iters = (int)(per_wave / 1024);
      for (int w = 0; w < 3; ++w) k<<<256, 64 * W>>>(buf, iters, off, per_wave);
      (void)hipEventRecord(a);
      for (int w = 0; w < 5; ++w) k<<<256, 64 * W>>>(buf, iters, off, per_wave);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      const double bytes = 5.0 * 256 * W * (double)iters * 1024;
      printf("offset %d  waves/CU %2d : %.2f TB/s  (%.2f GB/s per wave)\n", off, W, bytes / ms * 1e-9, bytes / ms * 1e-6 / (256 * W));
    }
  return 0;
}
