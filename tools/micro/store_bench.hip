// Store-pattern micro-benchmark (run on the GPU box): how fast can one workgroup per CU write 256 x 192 bf16 output tiles,
// depending on how a wave's 16-byte-per-lane store instruction is laid out over rows?
//   pattern 0: 16 rows x 64 B per instruction  (the gemm_nt8 epilogue: lane = (row & 15, 16-byte piece of a 64-byte segment))
//   pattern 1:  8 rows x 128 B per instruction (full cache lines)
//   pattern 2:  4 rows x 256 B per instruction
//   pattern 3:  2 rows x 512 B  (only with 384-byte rows: 1.33 rows...) -> skipped; 3 = one row-contiguous KiB per wave (linear)
// hipcc --offload-arch=gfx950 -O3 -o store_bench tools/micro/store_bench.hip && ./store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(char* out, int ld_bytes, int tiles_m, int tiles_n, int tiles_per_wg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int tile = blockIdx.x + t * gridDim.x;
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    if (tn >= tiles_n) break;
    char* base = out + (long)tm * 256 * ld_bytes + tn * 384;
    uint4 v = make_uint4(tile, lane, wave, t);
    if (PAT == 0) {
      // wave (wr = wave >> 2, wc = wave & 3): 128 rows x 48 columns (96 B); per 16-row band: one 16-B store covering
      // 64 B per row (2 fragments) + one 8-byte store (third fragment), as store_band_bf16<3>
      const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fg = lane >> 4;
      char* wb = base + (long)(wr * 128) * ld_bytes + wc * 96;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        char* p = wb + (long)(16 * i + fr) * ld_bytes;
        *(uint4*)(p + ((fg & 1) ? 32 + 8 * (fg - 1) : 8 * fg)) = v;
        *(uint2*)(p + 64 + 8 * fg) = make_uint2(v.x, v.y);
      }
    } else if (PAT == 1) {
      // wave: rows 32 w .. +31, all 384 B; instruction: 8 rows x 128 B
      const int r8 = lane >> 3, c = lane & 7;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int cg = 0; cg < 3; ++cg)
          *(uint4*)(base + (long)(32 * wave + 8 * rg + r8) * ld_bytes + cg * 128 + c * 16) = v;
    } else if (PAT == 2) {
      // instruction: 4 rows x 256 B (+ a 128-B remainder column group with 8 rows)
      const int r4 = lane >> 4, c = lane & 15;
#pragma unroll
      for (int rg = 0; rg < 8; ++rg)
        *(uint4*)(base + (long)(32 * wave + 4 * rg + r4) * ld_bytes + c * 16) = v;
      const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        *(uint4*)(base + (long)(32 * wave + 8 * rg + r8) * ld_bytes + 256 + c8 * 16) = v;
    } else {
      // fp32 GATE_RES-like: 16 rows x 64 B per instruction, 768-byte rows (192 fp32): 3 x 4 instructions per band
      const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fg = lane >> 4;
      char* wb = base + (long)(wr * 128) * ld_bytes + wc * 96;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        char* p = wb + (long)(16 * i + fr) * ld_bytes;
        *(uint4*)(p + 16 * fg) = v;
        *(uint2*)(p + 64 + 8 * fg) = make_uint2(v.x, v.y);
      }
    }
  }
}

int main() {
  const int M = 131072, N = 3456;
  const int ld = N * 2;
  char* out;
  hipMalloc(&out, (size_t)M * ld);
  hipMemset(out, 0, (size_t)M * ld);
  const int tiles_m = M / 256, tiles_n = N / 192, ntiles = tiles_m * tiles_n;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int grid : {256, 128, 512}) {
    const int per = (ntiles + grid - 1) / grid;
    for (int pat = 0; pat < 3; ++pat) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(512), 0, 0, out, ld, tiles_m, tiles_n, per);
        if (pat == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(512), 0, 0, out, ld, tiles_m, tiles_n, per);
        if (pat == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(512), 0, 0, out, ld, tiles_m, tiles_n, per);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double bytes = (double)M * ld;
      printf("grid %3d pattern %d: %8.1f us  %6.2f TB/s  %6.1f GB/s per workgroup  (%.2f us per 96-KB tile)\n", grid, pat, best * 1e3,
             bytes / best / 1e9, bytes / best / 1e6 / grid, best * 1e3 / per);
    }
  }
  return 0;
}
