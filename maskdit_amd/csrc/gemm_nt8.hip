// gemm_nt8: the large-problem bf16 NT GEMM (C[M,N] = A[M,K] * B[N,K]^T + fused epilogue).
//
// 256 x (64*NF) output tile per 512-thread workgroup (8 waves as 2(M) x 4(N); each wave owns
// 128 x 16*NF = 8 x NF MFMA 16x16x32 fragments), K-step 64, ONE workgroup per CU.
// Pipeline (per K-tile 4 phases, one raw s_barrier each, no vmcnt(0) in steady state):
//
//   * operands go HBM -> LDS by 16-byte LDS-DMA (global_load_lds) into a 2-stage ring; the ring
//     is managed at SLOT granularity: A slot p = the 2 x 32 tile rows the two wave-rows consume
//     in phase p (exactly one LDS-DMA instruction per wave), B = NF instructions per wave.
//     A slot is refilled with K-tile t+2 in the phase right after its last ds_read retired, so
//     every load has ~6 phases (1.5 K-tiles of MFMA work) to land;
//   * phase g: [ds_read the fragments of phase g+1 into the alternate register set]
//              [LDS-DMA refill of the slot read during phase g-1]  [4*NF MFMAs of phase g]
//              s_waitcnt vmcnt(W_p) lgkmcnt(0) ; s_barrier
//     W_p = number of loads issued after the one that the NEXT phase's reads depend on (loads
//     retire in order), computed at compile time -- never 0 until the last two K-tiles;
//   * XOR-swizzled LDS image through the *source* address (LDS-DMA destinations are lane-linear),
//     conflict-free ds_read_b128 fragment reads; XCD-aware tile order;
//   * PERSISTENT: one workgroup per CU walks the tile list; when a tile's K loop ends, the first
//     two K-tiles of the workgroup's NEXT output tile are put in flight before the epilogue runs
//     (the epilogue stages accumulators through a dedicated LDS region), so the operand fetch
//     latency and the drain of the epilogue's global stores overlap instead of adding up.
//
// Requirements (checked by the dispatcher in gemm.hip): M % 256 == 0, N % (64*NF) == 0,
// K % 128 == 0.  Everything else runs the 128x128 kernel in gemm.hip.
#include "common.h"
#include "../../include/maskdit_hip.h"
#include "gemm_common.h"

namespace {

// loads issued per wave in phase p: one A slot + RPP B rounds while p < NF (RPP = 1 with 8 waves,
// 2 with 4 waves: half as many waves share the same B tile)
constexpr int c_issue(int p, int NF, int RPP) { return 1 + (p < NF ? RPP : 0); }

// steady-state vmcnt operand at the end of phase p (see header)
constexpr int wait_count(int p, int NF, int RPP) {
  // next phase (g+1) prefetches A slot (p+2)&3 [of the current or the next K-tile], issued at
  // phase g-6 whose phase index is (p+2)&3; the B instruction of that phase was issued after it.
  int w = (((p + 2) & 3) < NF) ? RPP : 0;
  for (int d = 5; d >= 0; --d) w += c_issue(((p - d) % 4 + 4) % 4, NF, RPP);
  if (p == 2) {
    // phase 3 also reads the whole next-tile B: its last instruction was issued at phase NF-1 of
    // the previous K-tile; after it: one A load per phase NF..3, then phases 0..2 of this tile
    int wb = (4 - NF) + c_issue(0, NF, RPP) + c_issue(1, NF, RPP) + c_issue(2, NF, RPP);
    if (wb < w) w = wb;
  }
  return w;
}

// s_waitcnt vmcnt(N) lgkmcnt(0) as the BUILTIN (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8]
// | vmcnt_hi[15:14]) so that hipcc's own waitcnt bookkeeping sees the LDS reads as retired and
// does not re-wait (lgkmcnt(0)) in front of the next phase's MFMAs; the empty asm statements pin
// the memory-operation order around it.
// vmcnt operand at the end of phase d (0..7) of the LAST pair of K-tiles, where nothing is issued any
// more: the steady-state count minus the loads those phases would have issued
constexpr int drain_count(int d, int NF, int RPP) {
  if (d >= 6) return 0;  // nothing left to fetch: only LDS reads remain
  int w = (((d + 2) & 3) < NF) ? RPP : 0;                        // B issued right after the awaited A load (phase -6+d)
  for (int e = d - 5; e < 0; ++e) w += c_issue(((e % 4) + 4) % 4, NF, RPP);  // steady phases after it
  if (d == 2) {
    int wb = 4 - NF;  // A loads issued after the last B instruction of the final K-tile
    if (wb < w) w = wb;
  }
  return w;
}

template <int N> __device__ __forceinline__ void wait_vm_lgkm() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

}  // namespace

// WR = wave rows: 2 -> 256-row tile, 8 waves, one workgroup per CU (next-tile prefetch under the
// epilogue); 1 -> 128-row tile, 4 waves, TWO independent workgroups per CU, so one workgroup's
// epilogue (an HBM-write burst with idle matrix cores) runs under the other's K loop.
template <int NF, int WR>
__global__ __launch_bounds__(256 * WR, 2) void gemm_nt8_kernel(NTParams p) {
  constexpr int BN8 = 64 * NF;
  constexpr int BM8 = 128 * WR;
  constexpr int RPP = 2 / WR;            // B LDS-DMA rounds per phase
  constexpr int BROWS = 32 * WR;         // B rows covered by one round (8 rows per wave)
  constexpr int A_BYTES = BM8 * 128;
  constexpr int STAGE = A_BYTES + BN8 * 128;
  // epilogue staging (wave-private 16-row bands, fp32): padded pitch where LDS allows it; the
  // 256x256 tile uses the last 32 KiB of the 160 KiB LDS unpadded
  constexpr int WN = 16 * NF;
  constexpr int SP = (NF == 4) ? WN : WN + 4;
  constexpr int STG_BYTES = 4 * WR * 16 * SP * 4;
  // with one workgroup per CU the staging region is separate (next-tile prefetch overlaps the
  // epilogue); with two per CU each gets 80 KiB and the staging reuses stage 0 after the K loop
  constexpr bool PREFETCH = (WR == 2);
  constexpr int LDS_BYTES = PREFETCH ? 2 * STAGE + STG_BYTES : 2 * STAGE;
  static_assert(LDS_BYTES * (WR == 2 ? 1 : 2) <= 160 * 1024, "LDS budget");
  static_assert(PREFETCH || STG_BYTES <= STAGE, "staging must fit in a stage");
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int tiles_m = p.M / BM8, tiles_n = p.N / BN8;
  const int ntiles = tiles_m * tiles_n;
  int vt = blockIdx.x;  // virtual tile id of this workgroup's current tile (stride gridDim.x)
  int tm, tn;
  const int group_m = p.group_m > 0 ? p.group_m : GROUP_M;
  tile_coords(xcd_remap(vt, ntiles), tiles_m, tiles_n, tm, tn, group_m);
  int m0 = tm * BM8, n0 = tn * BN8;

  // ---- LDS-DMA addressing.  One wave-instruction = 8 tile rows x 128 B; lane -> (row lane/8,
  // LDS chunk lane%8); the global chunk is XOR-swizzled with (row & 7) = lane/8.
  const int lr = lane >> 3, gch = (lane & 7) ^ lr;
  // A slot q: wave w covers tile rows (w>>2)*128 + 32q + 8(w&3) .. +7
  const int a_row0 = (wave >> 2) * 128 + 8 * (wave & 3);
  const bf16* a_src = p.A + (long)(m0 + a_row0 + lr) * p.lda + gch * 8;  // re-pointed per tile
  const long a_qstride = 32L * p.lda;
  // B round j: wave w covers tile rows BROWS*j + 8w .. +7
  const bf16* b_src = p.B + (long)(n0 + 8 * wave + lr) * p.ldb + gch * 8;
  const long b_jstride = (long)BROWS * p.ldb;
  const int a_lds0 = a_row0 * 128;           // + 32q*128 + stage*STAGE
  const int b_lds0 = A_BYTES + wave * 1024;  // + j*8192 + stage*STAGE

  auto issue = [&](int stage, int kt, int ph) {
    char* base = smem + stage * STAGE;
    glds16(a_src + ph * a_qstride + (long)kt * 64, base + a_lds0 + ph * 4096);
    if (ph < NF) {
#pragma unroll
      for (int r = 0; r < RPP; ++r)
        glds16(b_src + (ph * RPP + r) * b_jstride + (long)kt * 64, base + b_lds0 + (ph * RPP + r) * (BROWS * 128));
    }
  };

  // ---- fragment read offsets (bytes inside a stage); row & 7 == fr & 7 for every fragment
  const int fr = lane & 15, fg = lane >> 4;
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int sw = ((ks * 4 + fg) ^ (fr & 7)) << 4;
    a_off[ks] = (wr * 128 + fr) * 128 + sw;
    b_off[ks] = A_BYTES + (wc * 16 * NF + fr) * 128 + sw;
  }

  f32x4 acc[8][NF];

  // B fragments are double-buffered across K-tiles while the register file allows it (NF <= 3);
  // for NF = 4 the next tile's B replaces the current one inside phase 3, ks by ks.
  constexpr bool BDB = NF < 4;
  bf16x8 Ar[2][2][2];                // [set][frag in phase][ks]
  bf16x8 Br[BDB ? 2 : 1][NF][2];     // [set][frag][ks]

  const int nk = p.K >> 6;  // even, >= 2

  // ---- optional stagger (p.epi bit 9): every other workgroup starts half a tile period late so that
  // the epilogues (HBM-write bursts with idle matrix cores) of one half of the chip fall under the K
  // loops of the other half instead of all 256 CUs bursting in lock-step.
  if ((p.epi & 0x200) && (WR == 1 ? (blockIdx.x >= (gridDim.x >> 1)) : ((blockIdx.x & 8) != 0))) {
    // 4-wave form: the second workgroup of each CU (dispatched in the second half of the grid)
    const int naps = ((p.K >> 6) * 1700 + 6000) >> 13;  // ~0.7 us per K-tile + half an epilogue, in 8192-cycle naps
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // ---- prologue of the first tile: K-tiles 0 and 1 in steady-state issue order
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) issue(0, 0, ph);
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) issue(1, 1, ph);

  for (;;) {  // persistent tile loop
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the two K-tiles of this output tile were put in flight before the previous tile's epilogue
  // (or just above): everything older -- including that epilogue's stores -- must have retired
  wait_vm_lgkm<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) Ar[0][i][ks] = *(const bf16x8*)(smem + a_off[ks] + i * 2048);
#pragma unroll
    for (int j = 0; j < NF; ++j) Br[0][j][ks] = *(const bf16x8*)(smem + b_off[ks] + j * 2048);
  }

#define PAIR_BODY(DRAIN)                                                                              \
  _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                            \
    const char* cur = smem + half * STAGE;                                                            \
    const char* nxt = smem + (half ^ 1) * STAGE;                                                      \
    _Pragma("unroll") for (int ph = 0; ph < 4; ++ph) {                                                \
      /* (1) prefetch the fragments of the next phase */                                              \
      if (ph < 3) {                                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
          Ar[(ph + 1) & 1][i][ks] = *(const bf16x8*)(cur + a_off[ks] + (2 * (ph + 1) + i) * 2048);    \
      } else if (!(DRAIN && half == 1)) {                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                            \
          _Pragma("unroll") for (int i = 0; i < 2; ++i)                                               \
            Ar[0][i][ks] = *(const bf16x8*)(nxt + a_off[ks] + i * 2048);                              \
          if (BDB) {                                                                                  \
            _Pragma("unroll") for (int j = 0; j < NF; ++j)                                            \
              Br[BDB ? (half ^ 1) : 0][j][ks] = *(const bf16x8*)(nxt + b_off[ks] + j * 2048);         \
          }                                                                                           \
        }                                                                                             \
      }                                                                                               \
      /* (2) refill the slot whose reads retired before the previous barrier */                       \
      if (!DRAIN) issue(half, kt + half + 2, ph);                                                     \
      /* (3) this phase's MFMAs */                                                                    \
      __builtin_amdgcn_s_setprio(1);                                                                  \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                \
          acc[2 * ph + i][j] = mfma16(Ar[ph & 1][i][ks], Br[BDB ? half : 0][j][ks], acc[2 * ph + i][j]); \
        if (!BDB && ph == 3 && !(DRAIN && half == 1)) {                                               \
          _Pragma("unroll") for (int j = 0; j < NF; ++j)                                              \
            Br[0][j][ks] = *(const bf16x8*)(nxt + b_off[ks] + j * 2048);                              \
        }                                                                                             \
      }                                                                                               \
      __builtin_amdgcn_s_setprio(0);                                                                  \
      /* (4) publish: my share of the next phase's data has landed, my LDS reads have retired */      \
      if (DRAIN) {                                                                                    \
        if (half == 0 && ph == 0) wait_vm_lgkm<drain_count(0, NF, RPP)>();                                 \
        else if (half == 0 && ph == 1) wait_vm_lgkm<drain_count(1, NF, RPP)>();                            \
        else if (half == 0 && ph == 2) wait_vm_lgkm<drain_count(2, NF, RPP)>();                            \
        else if (half == 0 && ph == 3) wait_vm_lgkm<drain_count(3, NF, RPP)>();                            \
        else if (half == 1 && ph == 0) wait_vm_lgkm<drain_count(4, NF, RPP)>();                            \
        else if (half == 1 && ph == 1) wait_vm_lgkm<drain_count(5, NF, RPP)>();                            \
        else wait_vm_lgkm<0>();                                                                       \
      }                                                                                               \
      else if (ph == 0) wait_vm_lgkm<wait_count(0, NF, RPP)>();                                            \
      else if (ph == 1) wait_vm_lgkm<wait_count(1, NF, RPP)>();                                            \
      else if (ph == 2) wait_vm_lgkm<wait_count(2, NF, RPP)>();                                            \
      else wait_vm_lgkm<wait_count(3, NF, RPP)>();                                                         \
      __builtin_amdgcn_s_barrier();                                                                   \
      asm volatile("" ::: "memory");                                                                  \
    }                                                                                                 \
  }

  int kt = 0;
  for (; kt + 2 < nk; kt += 2) { PAIR_BODY(false) }
  { PAIR_BODY(true) }
#undef PAIR_BODY

  // ---- next tile: put its first two K-tiles in flight (every LDS read of this tile retired
  // before the last barrier), then run this tile's epilogue underneath them
  const int em0 = m0, en0 = n0;
  vt += gridDim.x;
  const bool more = vt < ntiles;
  auto next_tile = [&]() {
    tile_coords(xcd_remap(vt, ntiles), tiles_m, tiles_n, tm, tn, group_m);
    m0 = tm * BM8;
    n0 = tn * BN8;
    a_src = p.A + (long)(m0 + a_row0 + lr) * p.lda + gch * 8;
    b_src = p.B + (long)(n0 + 8 * wave + lr) * p.ldb + gch * 8;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) issue(0, 0, ph);
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) issue(1, 1, ph);
  };
  if (PREFETCH && more) next_tile();

  // ---- epilogue: restage each 16-row fragment band through the wave-private staging region so
  // that a lane owns 4*NF consecutive columns of one row, then run the shared fused epilogue.
  float* stg = (float*)(smem + (PREFETCH ? 2 * STAGE : 0) + wave * (16 * SP * 4));
  const int er = lane >> 2, ec = (lane & 3) * (4 * NF);
  const int n = en0 + wc * WN + ec;
  if (p.epi & 0x100) {  // benchmarking aid (mdt_set_tuning "nt8_skip_epilogue"): main loop only
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) asm volatile("" ::"v"(acc[i][j]));
    if (!more) break;
    if (!PREFETCH) next_tile();
    continue;
  }
  // software pipeline over the 8 bands: the LDS round trip and the row-dependent global loads of
  // band i+1 are issued before the arithmetic + stores of band i
  float bias[4 * NF], csum[4 * NF];
#pragma unroll
  for (int q = 0; q < 4 * NF; ++q) csum[q] = 0.f;
  nt_load_bias<4 * NF>(p, n, bias);
  NtPre<4 * NF> pre[2];
  nt_epilogue_prefetch<4 * NF>(p, em0 + wr * 128 + er, n, pre[0]);
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) stg[(fg * 4 + r) * SP + j * 16 + fr] = acc[0][j][r];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v[4 * NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      f32x4 t = *(const f32x4*)(stg + er * SP + ec + q * 4);
      v[q * 4 + 0] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
    }
    if (i < 7) {
      // LDS executes a wave's operations in order: these writes cannot overtake the reads above
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(fg * 4 + r) * SP + j * 16 + fr] = acc[i + 1][j][r];
      nt_epilogue_prefetch<4 * NF>(p, em0 + wr * 128 + (i + 1) * 16 + er, n, pre[(i + 1) & 1]);
    }
    const int m = em0 + wr * 128 + i * 16 + er;
    nt_epilogue_finish<4 * NF>(p, m, n, v, bias, pre[i & 1], csum);
  }
  if (p.colsum) nt_colsum_flush<4 * NF>(p, n, csum, lane);
  if (!more) break;
  if (!PREFETCH) {  // the staging region aliases stage 0: every wave must be done with it first
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    next_tile();
  }
  }  // persistent tile loop
}

template __global__ void gemm_nt8_kernel<2, 2>(NTParams);
template __global__ void gemm_nt8_kernel<3, 2>(NTParams);
template __global__ void gemm_nt8_kernel<4, 2>(NTParams);
template __global__ void gemm_nt8_kernel<2, 1>(NTParams);
template __global__ void gemm_nt8_kernel<3, 1>(NTParams);

static int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

int launch_gemm_nt8(const NTParams& p, int nf, int wr, hipStream_t stream) {
  const int bm = 128 * wr;
  const int ntiles = (p.M / bm) * (p.N / (64 * nf));
  const int slots = num_cus() * (wr == 1 ? 2 : 1);  // persistent workgroups: 1 (8 waves) or 2 (4 waves) per CU
  const int grid = ntiles < slots ? ntiles : slots;
  const dim3 blk(256 * wr);
  if (wr == 2) {
    switch (nf) {
      case 2: hipLaunchKernelGGL((gemm_nt8_kernel<2, 2>), dim3(grid), blk, 0, stream, p); break;
      case 3: hipLaunchKernelGGL((gemm_nt8_kernel<3, 2>), dim3(grid), blk, 0, stream, p); break;
      default: hipLaunchKernelGGL((gemm_nt8_kernel<4, 2>), dim3(grid), blk, 0, stream, p); break;
    }
  } else {
    if (nf == 3) hipLaunchKernelGGL((gemm_nt8_kernel<3, 1>), dim3(grid), blk, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt8_kernel<2, 1>), dim3(grid), blk, 0, stream, p);
  }
  return mdt_check_launch("gemm_nt8");
}
