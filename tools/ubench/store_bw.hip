// per-CU store throughput: G workgroups of 512 threads stream-write `bytes` each (coalesced 8-byte or 16-byte lanes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <typename T>
__global__ __launch_bounds__(512) void k_store(T *out, long n_per_block, T v) {
  T *dst = out + (long)blockIdx.x * n_per_block;
  for (long i = threadIdx.x; i < n_per_block; i += 512) dst[i] = v;
}
template <typename T>
__global__ __launch_bounds__(512) void k_load(const T *in, long n_per_block, T *sink) {
  const T *src = in + (long)blockIdx.x * n_per_block;
  T acc = T();
  for (long i = threadIdx.x; i < n_per_block; i += 512) acc += src[i];
  if (acc == (T)1.2345e300) *sink = acc;
}
int main() {
  const long total = 1L << 30;
  double *buf;
  hipMalloc(&buf, total);
  hipMemset(buf, 0, total);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int G : {8, 16, 32, 64, 128, 256, 512, 1024}) {
    const long bytes_per_block = G <= 32 ? (8L << 20) : (total / G / 2);
    const long n = bytes_per_block / 8;
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        if (mode == 0) k_store<double><<<G, 512>>>(buf, n, 1.0);
        else k_load<double><<<G, 512>>>(buf, n, buf);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
      }
      printf("G=%4d %s: %.1f GB/s total, %.1f GB/s per WG\n", G, mode ? "load " : "store", G * bytes_per_block / best / 1e6, bytes_per_block / best / 1e6);
    }
  }
  return 0;
}
