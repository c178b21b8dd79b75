// The marching kernel's flush in isolation: every workgroup alternates an idle phase (stand-in for the element arithmetic) with a
// burst of 30 K-line writes (15 rows x 27 doubles each, lines 27.8 kB apart), 21 times.  Questions: how long does the burst take
// when all 256 CUs burst together, and does spreading the workgroups' phases (stagger) shorten it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void k(double *out, int steps, int idle_cycles, int stagger, long long *tburst, int busy) {
  const int sl = threadIdx.x & 31, rsub = threadIdx.x >> 5;
  if (stagger) {
    const long long t0 = clock64(), dt = (long long)stagger * ((blockIdx.x * 37) & 63) / 64;
    while (clock64() - t0 < dt) __builtin_amdgcn_s_sleep(8);
  }
  long long acc = 0;
  const long T2 = 385, T1T2 = 385L * 385;  // 128^3 mesh
  for (int s = 0; s < steps; ++s) {
    long long t0 = clock64();
    if (busy) {  // stand-in that keeps the f64 pipes busy (and the chip at power) instead of sleeping
      double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
      while (clock64() - t0 < idle_cycles) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
          x0 = fma(x0, 1.0000001, 1e-9); x1 = fma(x1, 1.0000001, 1e-9); x2 = fma(x2, 1.0000001, 1e-9); x3 = fma(x3, 1.0000001, 1e-9);
        }
      }
      if (x0 + x1 + x2 + x3 == 1.2345e300) out[0] = x0;
    } else
      while (clock64() - t0 < idle_cycles) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    t0 = clock64();
    const int col = (blockIdx.x * steps + s) % 81, P = 1 + (blockIdx.x * steps + s) / 81 * 2 % 120;
    const int J0 = (col / 9) * 15, K0 = (col % 9) * 15;
    if (sl < 27 && rsub < 15)
      for (int pl = 0; pl < 2; ++pl)
        for (int oj = 0; oj < 15; ++oj) {
          const long base = (3L * (P + pl) - 1) * T1T2 + 3 * ((3L * (J0 + oj) - 1) * T2) + 9 * (3L * (K0 + rsub) - 1);
          if (base + 27 < 57066625L) out[base + sl] = 1.0;
        }
    __syncthreads();
    acc += clock64() - t0;
  }
  if (threadIdx.x == 0) atomicAdd((unsigned long long *)tburst, (unsigned long long)acc);
}
int main() {
  double *buf;
  long long *tb;
  (void)hipMalloc(&buf, 57066625L * 8);
  (void)hipMalloc(&tb, 8);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int G : {256, 64})
    for (int idle : {0, 10000})
      for (int stagger : {0, 1}) {  // second loop variable reused: 0 = sleeping stand-in, 1 = f64-busy stand-in
        const int busy = stagger;
        (void)hipMemset(tb, 0, 8);
        for (int w = 0; w < 20; ++w) k<<<G, 512>>>(buf, 21, idle, 0, tb, busy);  // long enough for power management to react
        (void)hipMemset(tb, 0, 8);
        (void)hipEventRecord(a);
        k<<<G, 512>>>(buf, 21, idle, 0, tb, busy);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        long long h;
        (void)hipMemcpy(&h, tb, 8, hipMemcpyDeviceToHost);
        printf("G=%3d idle=%5d busy=%d: kernel %.3f ms, burst %.0f cycles per step\n", G, idle, stagger, ms, (double)h / G / 21);
      }
  return 0;
}
