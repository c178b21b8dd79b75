// Issue-rate of the P1-hex element routine alone (registers only): how many cycles per VALU instruction does the instruction
// stream of nh_p1hex_math.inc sustain at 2 waves/SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int64_t i64;
struct P1Args { double n[2][2], c[3][2], wk[2][2][2], wm[2][2][2]; };
__device__ __forceinline__ double fast_rcp(double d) { double r = __builtin_amdgcn_rcp(d); r = fma(fma(-d, r, 1.), r, r); r = fma(fma(-d, r, 1.), r, r); return r; }
__global__ __launch_bounds__(512) void k(P1Args p, const double *in, double *out, int iters) {
  double X[2][2][2][3];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int c = 0; c < 2; ++c) for (int d = 0; d < 3; ++d)
    X[a][b][c][d] = (d == 0 ? a : d == 1 ? b : c) + in[(threadIdx.x * 24 + ((a * 2 + b) * 2 + c) * 3 + d) % 4096];
  double acc = 0;
  for (int it = 0; it < iters; ++it) {
    double R0[3][3], R1[3][3], R2[3][3], W01[2][2][3], W02[2][2][3], W12[2][2][3], Mm[3][3][3];
    constexpr bool hasm = false;
#define NH_P1HEX_QS(q) 1.
#define NH_P1HEX_QM(q) 1.
#include "../../nutils_amd/csrc/nh_p1hex_math.inc"
    double s = 0;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) s += R0[a][b] + 2 * R1[a][b] + 3 * R2[a][b];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int c = 0; c < 3; ++c) s += W01[a][b][c] + 2 * W02[a][b][c] + 3 * W12[a][b][c];
    acc += s;
    X[0][0][0][0] += 1e-9 * s;  // loop-carried dependence so that iterations are not hoisted
  }
  if (acc == 1.2345) out[0] = acc;
}
int main() {
  P1Args p;
  const double g[2] = {.5 - .5 / sqrt(3.), .5 + .5 / sqrt(3.)};
  for (int q = 0; q < 2; ++q) { p.n[0][q] = 1 - g[q]; p.n[1][q] = g[q]; p.c[0][q] = p.n[0][q] * p.n[0][q]; p.c[1][q] = p.n[0][q] * p.n[1][q]; p.c[2][q] = p.n[1][q] * p.n[1][q]; }
  for (int i = 0; i < 8; ++i) (&p.wk[0][0][0])[i] = .125;
  double *in, *out; hipMalloc(&in, 8 * 4096); hipMalloc(&out, 8); hipMemset(in, 0, 8 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<256, 512>>>(p, in, out, 2); hipDeviceSynchronize();
  const int iters = 200;
  for (int nt : {256, 512, 256, 512}) {  // 1 and 2 waves per SIMD: can a lone wave keep the f64 pipe busy?
    for (int w = 0; w < 3; ++w) k<<<256, nt>>>(p, in, out, iters);
    hipEventRecord(e0); k<<<256, nt>>>(p, in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms1; hipEventElapsedTime(&ms1, e0, e1);
    printf("block %d: %.3f ms for %d iterations -> %.0f cycles @2.4GHz per element routine per SIMD-wave slot\n", nt, ms1, iters, ms1 * 1e-3 / iters * 2.4e9 / (nt / 256));
  }
  hipEventRecord(e0); k<<<256, 512>>>(p, in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("element routine: %.3f us per element-thread-iteration per wave; %d iters: %.3f ms -> %.1f ns per wave-element\n", 0., iters, ms, ms * 1e6 / iters);
  printf("  = %.0f cycles @2.4GHz per wave-element (2 waves/SIMD interleaved: per SIMD %.0f cycles per 2 elements)\n", ms * 1e-3 / iters * 2.4e9, ms * 1e-3 / iters * 2.4e9);
  return 0;
}
