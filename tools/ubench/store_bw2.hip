// per-CU store throughput for the CSR flush pattern: rows of 27 doubles written by 32-lane groups (5 lanes idle), rows contiguous
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void k_rows27(double *out, long rows_per_block, double v) {
  double *dst = out + (long)blockIdx.x * rows_per_block * 27;
  const int sl = threadIdx.x & 31, rsub = threadIdx.x >> 5;
  if (sl < 27)
    for (long r = rsub; r < rows_per_block; r += 16) dst[r * 27 + sl] = v;
}
__global__ __launch_bounds__(512) void k_flat(double *out, long rows_per_block, double v) {
  double *dst = out + (long)blockIdx.x * rows_per_block * 27;
  for (long i = threadIdx.x; i < rows_per_block * 27; i += 512) dst[i] = v;
}
// 15-row lines (405 doubles) separated by a large stride, like one K line per pass
__global__ __launch_bounds__(512) void k_lines(double *out, long lines_per_block, long stride, double v) {
  double *dst = out + (long)blockIdx.x * lines_per_block * stride;
  const int sl = threadIdx.x & 31, rsub = threadIdx.x >> 5;
  if (sl < 27 && rsub < 15)
    for (long l = 0; l < lines_per_block; ++l) dst[l * stride + rsub * 27 + sl] = v;
}
int main() {
  const long total = 1L << 30;
  double *buf;
  (void)hipMalloc(&buf, total);
  (void)hipMemset(buf, 0, total);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int G : {8, 64, 256}) {
    const long rows = (G <= 32 ? (8L << 20) : (total / G / 2)) / 216;
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e30f;
      long bytes = rows * 216;
      for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(a);
        if (mode == 0) k_rows27<<<G, 512>>>(buf, rows, 1.0);
        else if (mode == 1) k_flat<<<G, 512>>>(buf, rows, 1.0);
        else { k_lines<<<G, 512>>>(buf, rows / 15 / 8, 405 * 8, 1.0); bytes = rows / 15 / 8 * 405 * 8; }
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
      }
      printf("G=%4d %s: %.1f GB/s total, %.1f GB/s per WG\n", G, mode == 0 ? "rows27" : mode == 1 ? "flat  " : "lines ", G * bytes / best / 1e6, bytes / best / 1e6);
    }
  }
  return 0;
}
