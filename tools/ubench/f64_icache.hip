// Does a long straight-line f64 instruction stream (like the unrolled element routine) issue slower than a short loop?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int UNROLL>
__global__ void k(double *out, const double *in, int iters) {
  double x[8], y[8], z[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = in[threadIdx.x + i * 64]; y[i] = in[threadIdx.x + i * 64 + 1]; z[i] = in[threadIdx.x + i * 64 + 2]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < UNROLL; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = fma(y[(i + r) % 8], z[(i + 3 * r) % 8], x[i]);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 1.2345) out[0] = s;
}
template <int UNROLL>
void run(int threads, int blocks_per_cu) {
  double *out, *in; hipMalloc(&out, 8); hipMalloc(&in, 8 * 4096); hipMemset(in, 0, 8 * 4096);
  const int iters = 32000 / UNROLL, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<UNROLL><<<blocks, threads>>>(out, in, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<UNROLL><<<blocks, threads>>>(out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts = (double)blocks * threads / 64 * iters * UNROLL * 8;
  printf("body %5d instrs (%6.1f KB)  waves/SIMD=%d : %.2f cycles/instr @2.4GHz\n", UNROLL * 8, UNROLL * 8 * 8 / 1024., threads * blocks_per_cu / 256,
         2.4e9 / (insts / (ms * 1e-3) / 1024));
  hipFree(out); hipFree(in);
}
int main() {
  run<16>(512, 1); run<64>(512, 1); run<200>(512, 1); run<400>(512, 1); run<800>(512, 1); run<2000>(512, 1);
  run<200>(256, 2); run<800>(256, 2);
  return 0;
}
