// What the fp32-input matrix instruction sustains on THIS chip with random operands (the practical roof of
// csrc/f32path.hip's GEMM): every wave issues v_mfma_f32_32x32x2_f32 back to back on NACC independent accumulators, operands
// from registers (re-randomised per outer iteration so that the data toggles like a real GEMM's), 1 / 2 waves per SIMD.
// The datasheet peak (157.3 TFLOP/s) assumes 2.4 GHz; the chip clocks to its power budget.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak tools/micro/mfma_f32_peak.hip && ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, int zero) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = zero ? 0.f : in[(threadIdx.x * 8 + i) & 4095];
    b[i] = zero ? 0.f : in[(threadIdx.x * 8 + i + 2048) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float *in, *out;
  hipMalloc(&in, 4096 * 4);
  hipMalloc(&out, 4);
  float h[4096];
  unsigned x = 12345;
  for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xffff) / 32768.f - 1.f; }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
  const int iters = 4000;
  for (int zero = 0; zero < 2; ++zero)
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<4>, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, in, iters, zero);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const double flop = (double)cus * wg_per_cu * 4 * iters * 8 * 4 * 4096.0;  // waves x iters x 8 x NACC MFMAs x 4096 FLOP
      printf("%s operands, %d waves per SIMD: %.2f ms, %.1f TFLOP/s (%.3f of 157.3)\n", zero ? "zero  " : "random", wg_per_cu, best,
             flop / best / 1e9, flop / best / 1e9 / 157.3);
    }
  return 0;
}
