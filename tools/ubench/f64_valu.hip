// Microbenchmark: f64 VALU issue rate on gfx950 for independent / dependent chains, 1-8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, int OP>
__global__ void k(double *out, double a, double b, int iters) {
  double x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if (OP == 0) x[i] = fma(x[i], a, b);
        if (OP == 1) x[i] = x[i] * a;
        if (OP == 2) x[i] = x[i] + b;
      }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  if (s == 1.2345) out[0] = s;
}
template <int CHAINS, int OP>
void run(const char *name, int threads, int blocks_per_cu) {
  double *out; hipMalloc(&out, 8);
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CHAINS, OP><<<blocks, threads>>>(out, 1.0000001, 1e-9, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS, OP><<<blocks, threads>>>(out, 1.0000001, 1e-9, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts = (double)blocks * threads / 64 * iters * 16 * CHAINS;  // wave instructions
  double per_simd_per_s = insts / (ms * 1e-3) / 1024;
  printf("%-6s chains=%2d waves/SIMD=%d : %.3f ms  %.2f G wave-instr/s/SIMD  -> %.2f cycles/instr @2.4GHz, %.1f TFLOP/s-equiv(FMA)\n", name, CHAINS,
         threads * blocks_per_cu / 256, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, insts * 64 * 2 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  run<1, 0>("fma", 256, 1); run<2, 0>("fma", 256, 1); run<4, 0>("fma", 256, 1); run<8, 0>("fma", 256, 1);
  run<1, 0>("fma", 512, 1); run<2, 0>("fma", 512, 1); run<4, 0>("fma", 512, 1); run<8, 0>("fma", 512, 1);
  run<8, 0>("fma", 1024, 1); run<8, 0>("fma", 1024, 2);
  run<8, 1>("mul", 512, 1); run<8, 2>("add", 512, 1); run<8, 1>("mul", 1024, 2); run<8, 2>("add", 1024, 2);
  return 0;
}
