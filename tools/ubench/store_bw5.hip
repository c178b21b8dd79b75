// The flush pattern of the marching kernels in isolation, at the store rate only (no LDS, no arithmetic): every workgroup owns a 15 x 15
// dof column of the 128^3 matrix and streams plane after plane: 15 K lines of 405 doubles (3240 B), 27 720 B apart, planes 3.56 MB apart.
// mode 0: 4 waves, wave w stores chunk w (1 kB, 16 B per lane) of every line       (the kernels' order)
// mode 1: 4 waves, wave w stores lines w, w+4, ... completely (4 consecutive stores)
// mode 2: as 0 but 8 waves on two planes at once (the marching kernel's flush)
// mode 3: contiguous: the same number of bytes per workgroup as one long stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2), aligned(8)));
__global__ __launch_bounds__(512) void k(char *out, int mode, int nplanes) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long T2 = 385, T1 = 385, line = 9 * T2 * 8, plane = 3 * T1 * T2 * 8;
  const int col = blockIdx.x % 49, J0 = 15 * (1 + col / 7), K0 = 15 * (1 + col % 7);  // interior columns only
  const int p0 = 1 + (blockIdx.x / 49) * nplanes;
  char *base = out + (3L * (3 * J0 - 1) * T2 + 9L * (3 * K0 - 1)) * 8;
  const d2v v = {1.0, 2.0};
  if (mode == 3) {
    char *b = out + (size_t)blockIdx.x * nplanes * 15 * 3240 + wave * 1024 + 16 * lane;
    for (int i = 0; i < nplanes * 15; ++i)
      if (wave * 64 + lane < 202) *reinterpret_cast<d2v *>(b + (size_t)i * 3240 / 8 * 8) = v;
    return;
  }
  for (int p = 0; p < nplanes; p += (mode == 2 ? 2 : 1)) {
    char *pb = base + (3L * (p0 + p + (mode == 2 ? wave / 4 : 0)) - 1) * plane / 3;
    const int w = wave & 3;
    if (mode == 1) {
      for (int l = w; l < 15; l += 4)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c * 64 + lane < 202) *reinterpret_cast<d2v *>(pb + l * line + c * 1024 + 16 * lane) = v;
    } else {
      if (w * 64 + lane < 202)
#pragma unroll
        for (int l = 0; l < 15; ++l) *reinterpret_cast<d2v *>(pb + l * line + w * 1024 + 16 * lane) = v;
    }
  }
}
int main() {
  char *buf;
  (void)hipMalloc(&buf, 57066625L * 8 + (1 << 20));
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int nplanes = 24;  // 245 workgroups x 24 planes = 5 x 49 columns x 120 planes
  for (int mode : {0, 1, 2, 3, 0}) {
    const int threads = mode == 2 ? 512 : 256;
    for (int w = 0; w < 5; ++w) k<<<245, threads>>>(buf, mode, nplanes);
    (void)hipEventRecord(a);
    for (int w = 0; w < 10; ++w) k<<<245, threads>>>(buf, mode, nplanes);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = 10.0 * 245 * nplanes * 15 * 3232;
    printf("mode %d: %.3f ms per launch, %.2f TB/s\n", mode, ms / 10, bytes / ms * 1e-9);
  }
  return 0;
}
