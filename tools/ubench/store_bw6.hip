// Does arithmetic on the other waves of the CU slow the flush stores down?  As store_bw5 mode 0 (4 waves per workgroup stream plane after
// plane of a 15 x 15 dof column: 15 lines of 3240 B, 16 B per lane), plus 4 more waves per workgroup that (mode 1) run dependent-free f64 FMAs,
// (mode 2) FMAs + ds_add_f64 into LDS, until the storing waves are done.  Reported: time of the launch = time the stores take.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2), aligned(8)));
__global__ __launch_bounds__(512) void k(char *out, int mode, int nplanes, double *sink) {
  __shared__ double acc[4096];
  __shared__ int done;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) done = 0;
  for (int i = threadIdx.x; i < 4096; i += 512) acc[i] = 0;
  __syncthreads();
  if (wave >= 4) {
    if (mode == 0) return;
    double x0 = lane, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5;
    while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        x0 = fma(x0, 1.0000001, 1e-9); x1 = fma(x1, 1.0000001, 1e-9); x2 = fma(x2, 1.0000001, 1e-9);
        x3 = fma(x3, 1.0000001, 1e-9); x4 = fma(x4, 1.0000001, 1e-9); x5 = fma(x5, 1.0000001, 1e-9);
        if (mode == 2 && (i & 3) == 0) atomicAdd(&acc[(threadIdx.x * 15 + i) & 4095], x0);
      }
    }
    if (x0 + x1 + x2 + x3 + x4 + x5 == 1.2345) sink[0] = x0;
    return;
  }
  const long T2 = 385, T1 = 385, line = 9 * T2 * 8, plane = 3 * T1 * T2 * 8;
  const int col = blockIdx.x % 49, J0 = 15 * (1 + col / 7), K0 = 15 * (1 + col % 7);
  const int p0 = 1 + (blockIdx.x / 49) * nplanes;
  char *base = out + (3L * (3 * J0 - 1) * T2 + 9L * (3 * K0 - 1)) * 8;
  const d2v v = {1.0, 2.0};
  for (int p = 0; p < nplanes; ++p) {
    char *pb = base + (3L * (p0 + p) - 1) * plane / 3;
    if (wave * 64 + lane < 202)
#pragma unroll
      for (int l = 0; l < 15; ++l) *reinterpret_cast<d2v *>(pb + l * line + wave * 1024 + 16 * lane) = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) atomicAdd(&done, 1);
}
int main() {
  char *buf;
  double *sink;
  (void)hipMalloc(&buf, 57066625L * 8 + (1 << 20));
  (void)hipMalloc(&sink, 64);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int nplanes = 24;
  for (int mode : {0, 1, 2, 0, 1, 2}) {
    for (int w = 0; w < 5; ++w) k<<<245, 512>>>(buf, mode, nplanes, sink);
    (void)hipEventRecord(a);
    for (int w = 0; w < 10; ++w) k<<<245, 512>>>(buf, mode, nplanes, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = 10.0 * 245 * nplanes * 15 * 3232;
    printf("mode %d (%s): %.3f ms per launch, %.2f TB/s\n", mode, mode == 0 ? "stores only" : mode == 1 ? "+ f64 FMA waves" : "+ FMA and ds_add_f64 waves", ms / 10, bytes / ms * 1e-9);
  }
  return 0;
}
