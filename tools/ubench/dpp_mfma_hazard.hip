// How many wait states does gfx950 need between a VALU f64 write (v_fmac_f64 / v_fmac_f64_dpp, inline asm: invisible to the compiler's hazard recogniser)
// and a v_mfma_f64_16x16x4_f64 that reads the register as its B (or A) operand?   hipcc -O3 --offload-arch=gfx950 dpp_mfma_hazard.hip -o dpp_mfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NOPS, int DPP, int ASIDE>
__global__ void k(const double *in, double *out) {
  const int l = threadIdx.x;
  double x = in[l], t = in[64 + l], other = in[128 + l];
  v4d acc = {0., 0., 0., 0.};
  double b = 123.;  // stale value the MFMA sees if it reads too early
  asm volatile(
      "v_mov_b64 %1, 0\n\t"
      "s_nop 7\n\t"
      ".if %6\n\t"
      "v_fmac_f64_dpp %1, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      ".else\n\t"
      "v_fmac_f64 %1, %2, %3\n\t"
      ".endif\n\t"
      ".if %5 >= 0\n\t"
      "s_nop %5\n\t"
      ".endif\n\t"
      ".if %7\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %1, %4, 0\n\t"
      ".else\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %4, %1, 0\n\t"
      ".endif\n\t"
      "s_nop 7\n\ts_nop 7\n\ts_nop 7"
      : "+v"(acc), "+v"(b)
      : "v"(x), "v"(t), "v"(other), "n"(NOPS), "n"(DPP), "n"(ASIDE));
  for (int i = 0; i < 4; ++i) out[i * 64 + l] = acc[i];
  out[256 + l] = b;
}
template <int NOPS, int DPP, int ASIDE>
int run(const double *d, double *dout, const double *h) {
  double ho[320];
  k<NOPS, DPP, ASIDE><<<1, 64>>>(d, dout);
  (void)hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  // reference on the host: B (or A) operand value per lane
  double opv[64];
  for (int l = 0; l < 64; ++l) opv[l] = (DPP ? h[(l & ~15)] : h[l]) * h[64 + l];
  int bad = 0;
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) {
      double s = 0;
      for (int kk = 0; kk < 4; ++kk) {
        const double a = ASIDE ? opv[kk * 16 + r] : h[128 + kk * 16 + r];
        const double bb = ASIDE ? h[128 + kk * 16 + c] : opv[kk * 16 + c];
        s += a * bb;
      }
      const double got = ho[(r >> 2) * 64 + (r & 3) * 16 + c];  // row r: register r / 4, lane group r % 4
      if (fabs(got - s) > 1e-9 * fabs(s)) ++bad;
    }
  return bad;
}
int main() {
  double h[192], *d, *dout;
  for (int i = 0; i < 192; ++i) h[i] = 1 + 0.01 * i + (i % 7) * 0.1;
  (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&dout, 320 * 8);
  (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
#define ROW(DPP, AS) printf("%-16s -> MFMA %s operand: mismatching entries with no nop / s_nop 0..4: %d | %d %d %d %d %d\n", DPP ? "v_fmac_f64_dpp" : "v_fmac_f64", AS ? "A" : "B", \
    run<-1, DPP, AS>(d, dout, h), run<0, DPP, AS>(d, dout, h), run<1, DPP, AS>(d, dout, h), run<2, DPP, AS>(d, dout, h), run<3, DPP, AS>(d, dout, h), run<4, DPP, AS>(d, dout, h))
  ROW(0, 0); ROW(1, 0); ROW(0, 1); ROW(1, 1);
  return 0;
}
