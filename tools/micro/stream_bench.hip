// Streaming-access micro-benchmark (GPU box): does a memory-bound elementwise kernel (LayerNorm backward: fp32 + bf16 streams) lose
// bandwidth to 8-byte bf16 accesses, and what does the alternative cost -- a lane owning 8 consecutive columns (16-byte bf16
// accesses, the fp32 streams as TWO 16-byte accesses 32 bytes apart between lanes)?
//   mode 0: quad mapping  -- per lane: fp32 16 B (lanes 16 B apart), bf16 8 B            (what norm.hip does)
//   mode 1: octet mapping -- per lane: fp32 2 x 16 B (lanes 32 B apart), bf16 16 B
// Each mode reads two fp32 streams + two bf16 streams and writes one fp32 + one bf16 stream (the 18 B/element of
// ln_bwd_gate_split).  hipcc --offload-arch=gfx950 -O3 -o stream_bench tools/micro/stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned short h4;
typedef __attribute__((ext_vector_type(8))) unsigned short h8;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* a, const float* b, const unsigned short* c, const unsigned short* d, float* o,
                                         unsigned short* p, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  if (MODE == 0) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += stride) {
      f4 x = ((const f4*)a)[i], y = ((const f4*)b)[i];
      h4 u = ((const h4*)c)[i], v = ((const h4*)d)[i];
      f4 r = x * 1.5f + y;
      h4 s;
      for (int e = 0; e < 4; ++e) { r[e] += (float)u[e]; s[e] = (unsigned short)(v[e] + u[e]); }
      ((f4*)o)[i] = r;
      ((h4*)p)[i] = s;
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 8; i += stride) {
      f4 x0 = ((const f4*)a)[2 * i], x1 = ((const f4*)a)[2 * i + 1], y0 = ((const f4*)b)[2 * i], y1 = ((const f4*)b)[2 * i + 1];
      h8 u = ((const h8*)c)[i], v = ((const h8*)d)[i];
      f4 r0 = x0 * 1.5f + y0, r1 = x1 * 1.5f + y1;
      h8 s;
      for (int e = 0; e < 4; ++e) { r0[e] += (float)u[e]; r1[e] += (float)u[4 + e]; }
      for (int e = 0; e < 8; ++e) s[e] = (unsigned short)(v[e] + u[e]);
      ((f4*)o)[2 * i] = r0;
      ((f4*)o)[2 * i + 1] = r1;
      ((h8*)p)[i] = s;
    }
  }
}

int main() {
  const long n = 131072L * 1152;
  float *a, *b, *o;
  unsigned short *c, *d, *p;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, n * 4);
  hipMalloc(&c, n * 2); hipMalloc(&d, n * 2); hipMalloc(&p, n * 2);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(c, 0, n * 2); hipMemset(d, 0, n * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {2048, 8192}) {
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, a, b, c, d, o, p, n);
        else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, a, b, c, d, o, p, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("grid %5d mode %d (%s): %7.1f us  %5.2f TB/s\n", grid, mode, mode ? "octets: bf16 16 B, fp32 2 x 16 B" : "quads: bf16 8 B, fp32 16 B", best * 1e3,
             18.0 * n / best / 1e9);
    }
  }
  return 0;
}
