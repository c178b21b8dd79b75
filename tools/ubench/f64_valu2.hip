// f64 VALU issue rate with 2-3 VGPR source operands (vs SGPR/constant operands in f64_valu.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, int OP>
__global__ void k(double *out, const double *in, int iters) {
  double x[CHAINS], y[CHAINS], z[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) { x[i] = in[threadIdx.x + i * 64]; y[i] = in[threadIdx.x + i * 64 + 1]; z[i] = in[threadIdx.x + i * 64 + 2]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if (OP == 0) x[i] = fma(x[i], y[i], z[i]);                       // 3 VGPR operands
        if (OP == 1) x[i] = x[i] * y[i];                                 // 2 VGPR
        if (OP == 2) x[i] = x[i] + y[i];                                 // 2 VGPR
        if (OP == 3) x[i] = fma(y[(i + 1) % CHAINS], z[i], x[i]);        // fmac form: acc += a*b, 3 VGPR
        if (OP == 4) x[i] = fma(x[i], 1.0000001, z[i]);                  // 2 VGPR + literal
      }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  if (s == 1.2345) out[0] = s;
}
template <int CHAINS, int OP>
void run(const char *name, int threads, int blocks_per_cu) {
  double *out, *in; hipMalloc(&out, 8); hipMalloc(&in, 8 * 4096); hipMemset(in, 0, 8 * 4096);
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CHAINS, OP><<<blocks, threads>>>(out, in, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS, OP><<<blocks, threads>>>(out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts = (double)blocks * threads / 64 * iters * 16 * CHAINS;
  double per_simd_per_s = insts / (ms * 1e-3) / 1024;
  printf("%-14s chains=%2d waves/SIMD=%d : %.2f cycles/instr @2.4GHz\n", name, CHAINS, threads * blocks_per_cu / 256, 2.4e9 / per_simd_per_s);
  hipFree(out); hipFree(in);
}
int main() {
  run<8, 0>("fma 3vgpr", 512, 1); run<8, 3>("fmac 3vgpr", 512, 1); run<8, 1>("mul 2vgpr", 512, 1); run<8, 2>("add 2vgpr", 512, 1); run<8, 4>("fma 2vgpr+lit", 512, 1);
  run<8, 0>("fma 3vgpr", 256, 1); run<8, 3>("fmac 3vgpr", 256, 1);
  run<8, 0>("fma 3vgpr", 1024, 2); run<8, 3>("fmac 3vgpr", 1024, 2); run<8, 1>("mul 2vgpr", 1024, 2);
  return 0;
}
