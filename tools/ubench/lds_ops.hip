// Microbenchmark: LDS operation rates on gfx950 as the p2hex kernel uses them: ds_add_f64 (no return), ds_read_b64 + ds_write_b64
// read-modify-write, ds_write_b64, ds_read_b64; lane address = base + (lane>>4)*ROWSTRIDE + (lane&15)*STRIDE doubles, 9 ops per round
// at offsets {0,1,2} + c*RS.  Reports cycles per wave-instruction per CU (all waves of the CU together).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(double *out, int iters, int stride, int rowstride, int rs) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *p = lds + wave * 1600 + (lane >> 4) * rowstride + (lane & 15) * stride;
  double acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        double *q = p + c * rs + d;
        if (OP == 0) atomicAdd(q, 1.0);
        if (OP == 1) *q = *q + 1.0;
        if (OP == 2) *q = (double)it;
        if (OP == 3) acc += *q;
      }
    if (OP == 3) asm volatile("" ::: "memory");
  }
  __syncthreads();
  if (acc == 1.2345 || lds[threadIdx.x] == -1.) out[0] = acc;
}
template <int OP>
void run(const char *name, int threads, int stride, int rowstride, int rs) {
  double *out; hipMalloc(&out, 8);
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, threads, 131072>>>(out, 10, stride, rowstride, rs);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, threads, 131072>>>(out, iters, stride, rowstride, rs);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts = (double)threads / 64 * iters * 9;  // wave instructions per CU
  printf("%-10s waves/CU=%2d stride=%d rowstride=%d rs=%d : %.3f ms -> %.1f cycles per wave-op per CU @2.4GHz\n", name, threads / 64, stride, rowstride, rs, ms,
         ms * 1e-3 * 2.4e9 / insts);
  hipFree(out);
}
int main() {
  hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int threads : {64, 256, 512}) {
    run<0>("ds_add_f64", threads, 3, 375, 125);
    run<0>("ds_add_f64", threads, 1, 16, 64);
    run<1>("rmw", threads, 3, 375, 125);
    run<2>("write_b64", threads, 3, 375, 125);
    run<3>("read_b64", threads, 3, 375, 125);
  }
  return 0;
}
