// How fast can ONE CU pull L2-resident data into LDS with global_load_lds_dwordx4 (the operand path of gemm_nt8 / gemm_tn8 /
// gemm_nt8o)?  DESIGN.md section 0 reads the GEMM tables as "the K loops run the CU's vector-memory path at ~ 23-28 B/clk";
// this measures the path with NOTHING else on the CU: no MFMA, no fragment reads, no barrier.
//   one 512-thread workgroup per CU; NW of its 8 waves issue 1 KiB pieces (16 B / lane) back to back into their own LDS
//   slice, `vmcnt(16)` after every 8 pieces (16 in flight per wave, like the GEMMs' ring); source = the workgroup's own
//   region, walked repeatedly (L2-resident after the first pass): pattern 0 = 8 rows x 128 B per piece with a row pitch of
//   2304 B (the A operand at K = 1152), pattern 1 = 1 KiB contiguous per piece.
// Output per (pattern, region size, NW): bytes / shader clock / CU (s_memtime inside the kernel, slowest workgroup) and
// aggregate TB/s (HIP events).      hipcc --offload-arch=gfx950 -O3 -o lds_dma_bench tools/micro/lds_dma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int PATTERN>
__global__ __launch_bounds__(512, 2) void k(const char* src, long region_bytes, int passes, int nw, unsigned long long* clocks) {
  __shared__ __attribute__((aligned(16))) char smem[8 * 16 * 1024];  // 16 KiB per wave = 16 pieces in flight
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (long)blockIdx.x * region_bytes;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < nw) {
    const long pieces = region_bytes / 1024;            // pieces per pass over the region
    const long per_wave = pieces / nw;
    // lane -> source offset inside a piece
    // pattern 0 / 2: 8 rows x 128 B, row pitch 2304 / 9216 B (A or B operand at K = 1152 / 4608); 1: 1 KiB contiguous, lane-linear;
    // 3: 1 KiB contiguous read in the GEMMs' swizzled lane order (row lane / 8 at 128-byte pitch, chunk (lane % 8) ^ (lane / 8))
    constexpr long PITCH = PATTERN == 2 ? 9216 : 2304;
    const long lane_off = (PATTERN == 0 || PATTERN == 2) ? (long)(lane >> 3) * PITCH + (lane & 7) * 16
                          : PATTERN == 1 ? (long)lane * 16 : (long)(lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);
    for (int pass = 0; pass < passes; ++pass) {
      for (long q = 0; q < per_wave; q += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const long piece = (long)wave * per_wave + q + u;
          // pattern 0: piece p = rows 8 (p / 18) .. + 7 of a [rows][2304 B] matrix, 128-byte column block p % 18
          constexpr long CB = PITCH / 128;  // 128-byte column blocks per row
          const long off = (PATTERN == 0 || PATTERN == 2) ? (piece / CB) * 8 * PITCH + (piece % CB) * 128 : piece * 1024;
          __builtin_amdgcn_global_load_lds(GLOBAL_PTR(base + off + lane_off), LDS_PTR(smem + wave * 16384 + ((q + u) & 15) * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt((16 & 15) | (7 << 4) | (15 << 8) | ((16 >> 4) << 14));  // vmcnt(16): the older 8 have landed
      }
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
  }
  __syncthreads();
  if (threadIdx.x == 0) clocks[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

int main() {
  int dev = 0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  const int cus = prop.multiProcessorCount;
  const long max_region = 1152L * 1024;  // multiple of 18 KiB (pattern 0's row blocks) and of 1 KiB x 8 x 8 waves
  char* src;
  hipMalloc(&src, (size_t)cus * max_region + (1 << 20));
  hipMemset(src, 1, (size_t)cus * max_region + (1 << 20));
  unsigned long long* clk;
  hipMalloc(&clk, cus * sizeof(unsigned long long));
  std::vector<unsigned long long> h(cus);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%d CUs; one workgroup per CU; L2-resident source (each workgroup re-reads its own region)\n", cus);
  printf("%-28s %10s %6s %14s %12s\n", "pattern", "region KiB", "waves", "B/clk/CU", "aggregate TB/s");
  const char* names[4] = {"8 rows x 128 B, pitch 2304", "1 KiB contiguous", "8 rows x 128 B, pitch 9216", "1 KiB contiguous, swizzled"};
  for (int pattern = 0; pattern < 4; ++pattern)
    for (long region : {72L * 1024, 144L * 1024, 1152L * 1024})  // x 32 CUs per XCD: 2.25 MiB (L2-resident), 4.5 MiB (L2 4 MiB: a mix), 36 MiB (Infinity Cache)
      for (int nw : {2, 8}) {
        const int passes = region >= 1152L * 1024 ? 8 : 64;
        float best_ms = 1e9f;
        unsigned long long worst = 0;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          if (pattern == 0) hipLaunchKernelGGL(k<0>, dim3(cus), dim3(512), 0, 0, src, region, passes, nw, clk);
          else if (pattern == 1) hipLaunchKernelGGL(k<1>, dim3(cus), dim3(512), 0, 0, src, region, passes, nw, clk);
          else if (pattern == 2) hipLaunchKernelGGL(k<2>, dim3(cus), dim3(512), 0, 0, src, region, passes, nw, clk);
          else hipLaunchKernelGGL(k<3>, dim3(cus), dim3(512), 0, 0, src, region, passes, nw, clk);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best_ms) {
            best_ms = ms;
            hipMemcpy(h.data(), clk, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            worst = 0;
            for (auto c : h) worst = c > worst ? c : worst;
          }
        }
        const double bytes = (double)region * passes;
        printf("%-28s %10ld %6d %14.1f %12.2f\n", names[pattern], region / 1024, nw, bytes / (double)worst,
               bytes * cus / (best_ms * 1e-3) / 1e12);
      }
  return 0;
}
