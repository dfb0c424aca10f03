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
//   * phase g: the 4*NF MFMAs of phase g with ONE memory instruction pinned behind each of the first ones
//              (sched_barrier): the ds_reads of phase g+1's fragments into the alternate register set, then the
//              LDS-DMA refill of the slot read during phase g-1 (round 3; round 2 clustered reads / DMA / MFMAs: both
//              waves of a SIMD then queue memory instructions in front of an idle matrix pipe, -13 % on the K loop);
//              s_waitcnt vmcnt(W_p) lgkmcnt(0) ; s_barrier
//     W_p = number of loads issued after the one that the NEXT phase's reads depend on (loads
//     retire in order), computed at compile time -- never 0 until the last two K-tiles;
//   * XOR-swizzled LDS image through the *source* address (LDS-DMA destinations are lane-linear),
//     conflict-free ds_read_b128 fragment reads; XCD-aware tile order;
//   * PERSISTENT: one workgroup per CU walks the tile list; when a tile's K loop ends, the first
//     two K-tiles of the workgroup's NEXT output tile are put in flight before the epilogue runs;
//   * EPILOGUE (round 2): the MFMAs are issued with the operands SWAPPED (D^T = B A^T), so a lane's four
//     accumulator registers of a fragment are four CONSECUTIVE COLUMNS of one output row (row = lane & 15,
//     columns 4*(lane>>4)..+3): the fused epilogue works straight out of the accumulators with 16-byte fp32 /
//     8-byte bf16 global accesses (bf16 pairs of fragments are widened to 16 bytes with v_permlane16_swap) --
//     no LDS restaging, no barrier, no LDS region.  The epilogue class is a TEMPLATE parameter and every
//     global load of the tile (residual / saved pre-activation, gates, bias) is issued up front (a
//     DEPTH-band look-ahead bounded by the register file), so the tile pays ONE memory round trip and its
//     stores stream out back to back instead of one load->store round trip per 16-row band.
//
// Requirements (checked by the dispatcher in gemm.hip): M % (128*WR) == 0, N % (64*NF) == 0,
// K % 128 == 0, 16-byte aligned rows of every output.  Everything else runs the 128x128 kernel in gemm.hip.
#pragma once
#ifndef NT8_DEFAULT_SCHED
#define NT8_DEFAULT_SCHED 5  // one memory instruction behind each MFMA: +8..11 % on the K loop over the clustered form (round 3)
#endif
#include "common.h"
#include "../../include/maskdit_hip.h"
#include "gemm_common.h"
#include <type_traits>

// an all-zero row standing in for a NULL bias (the epilogue's loads are unconditional)
#define NT8_ZERO_ROW 1024
static __device__ float nt8_zero_row[NT8_ZERO_ROW + 64];  // zero-initialised; one copy per translation unit

// timing stamps of the experiment kernels (SCHED bit 9; tools/nt8_stamps.py): [tile][event] shader-clock values of wave 0
// of workgroup 0 -- 0 tile start (after the tile-top wait + barrier), 1 K loop done, 2 next tile's LDS-DMA issued,
// 3 epilogue's last store issued
static __device__ unsigned long long nt8_stamps[2 * 64 * 4];  // [wave 0 | last wave]

namespace nt8 {

enum { E_PLAIN = 0, E_F32 = 1, E_ACT = 2, E_GATE = 3, E_DACT = 4,
       E_TRK = 5 };  // E_TRK: overlap EXPERIMENT (tools/nt8_bench.py): GATE-sized epilogue traffic issued one 16-byte op per
                     // phase inside the K loop instead of after it (results are garbage; timing only)

// loads issued per wave in phase p: one A slot + RPP B rounds while p < NF (RPP = 1 with 8 waves,
// 2 with 4 waves: half as many waves share the same B tile)
constexpr int c_issue(int p, int NF, int RPP) { return 1 + (p < NF ? RPP : 0); }

// steady-state vmcnt operand at the end of phase p (see header)
constexpr int wait_count(int p, int NF, int RPP, int trk = 0) {
  // next phase (g+1) prefetches A slot (p+2)&3 [of the current or the next K-tile], issued at
  // phase g-6 whose phase index is (p+2)&3; the B instruction of that phase was issued after it.
  int w = (((p + 2) & 3) < NF) ? RPP : 0;
  for (int d = 5; d >= 0; --d) w += c_issue(((p - d) % 4 + 4) % 4, NF, RPP) + trk;  // trk extra ops per phase (E_TRK)
  if (p == 2) {
    // phase 3 also reads the whole next-tile B: its last instruction was issued at phase NF-1 of
    // the previous K-tile; after it: one A load per phase NF..3, then phases 0..2 of this tile
    int wb = (4 - NF) + c_issue(0, NF, RPP) + c_issue(1, NF, RPP) + c_issue(2, NF, RPP) + trk * ((4 - NF) + 3);
    if (wb < w) w = wb;
  }
  return w;
}

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

// s_waitcnt vmcnt(N) lgkmcnt(0) as the BUILTIN (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8]
// | vmcnt_hi[15:14]) so that hipcc's own waitcnt bookkeeping sees the LDS reads as retired and
// does not re-wait (lgkmcnt(0)) in front of the next phase's MFMAs; the empty asm statements pin
// the memory-operation order around it.
template <int N> __device__ __forceinline__ void wait_vm_lgkm() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  asm volatile("; MDT_CHK hand_wait" ::: "memory");  // (a comment in the ISA: tools/check_waits.py audits the immediate that follows)
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  bf16x2 t;
  t[0] = f2bf(lo);
  t[1] = f2bf(hi);
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ f32x4 round_bf16(f32x4 v) {
  return (f32x4){bf2f(f2bf(v[0])), bf2f(f2bf(v[1])), bf2f(f2bf(v[2])), bf2f(f2bf(v[3]))};
}
__device__ __forceinline__ f32x4 unpack_bf16x4(uint2 u) {
  bf16x2 a = __builtin_bit_cast(bf16x2, u.x), b = __builtin_bit_cast(bf16x2, u.y);
  return (f32x4){bf2f(a[0]), bf2f(a[1]), bf2f(b[0]), bf2f(b[1])};
}

// Store one 16-row band of a wave's tile as bf16.  y[j] = this lane's 4 consecutive columns of fragment j
// (columns 16j + 4g .. +3 of row `fr`, g = lane >> 4).  `ub` = wave-uniform address of the band's first row at
// the wave tile's column 0; `lo_pair` / `lo_tail` = this lane's byte offsets (bf16_lane_offsets).  Fragment pairs
// (j, j+1) are exchanged between the odd and even 16-lane rows with v_permlane16_swap so that every lane stores
// 8 consecutive columns (16 bytes): even g -> fragment j columns 4g..4g+7, odd g -> fragment j+1 columns
// 4(g-1)..4(g-1)+7.
// For an ODD fragment count the last fragment of band i (8 bytes per lane) is not stored on its own: it is carried to
// band i + 1 and exchanged with that band's last fragment the same way, so that ONE 16-byte store per lane writes both
// (even 16-lane rows: 8 columns of band i's row, odd rows: 8 columns of band i + 1's row; `lo_tail2` = bf16_tail2_offset
// relative to the EVEN band).  8-byte accesses run at 0.54-0.70 of the 16-byte rate: 12 instead of 16 store instructions
// per wave and 256 x 192 tile.
struct TailCarry { unsigned lo, hi; };
template <int NF, bool ODD_BAND>
__device__ __forceinline__ void store_band_bf16(char* ub, char* ub_even, unsigned lo_pair, unsigned lo_tail2, const f32x4* y, TailCarry& tc) {
  unsigned lo[NF], hi[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    lo[j] = pack_bf16x2(y[j][0], y[j][1]);
    hi[j] = pack_bf16x2(y[j][2], y[j][3]);
  }
#pragma unroll
  for (int j = 0; j + 1 < NF; j += 2) {
    auto a = __builtin_amdgcn_permlane16_swap(lo[j], lo[j + 1], false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi[j], hi[j + 1], false, false);
    *(uint4*)(ub + opaque(lo_pair) + 32 * j) = make_uint4(a[0], b[0], a[1], b[1]);
  }
  if (NF & 1) {
    if (!ODD_BAND) {
      tc.lo = lo[NF - 1];
      tc.hi = hi[NF - 1];
    } else {
      auto a = __builtin_amdgcn_permlane16_swap(tc.lo, lo[NF - 1], false, false);
      auto b = __builtin_amdgcn_permlane16_swap(tc.hi, hi[NF - 1], false, false);
      *(uint4*)(ub_even + opaque(lo_tail2) + 32 * (NF - 1)) = make_uint4(a[0], b[0], a[1], b[1]);
    }
  }
}
// band i of a [rows, ld] bf16 array at wave-uniform base `base0` (band 0): the constant-index wrapper the epilogues use
#define NT8_STORE_BAND(NFV, base0, ld, i, lp, lt2, y, tc)                                                     \
  do {                                                                                                       \
    if ((i) & 1) store_band_bf16<NFV, true>(band(base0, (i), (ld), 2), band(base0, (i) - 1, (ld), 2), lp, lt2, y, tc); \
    else store_band_bf16<NFV, false>(band(base0, (i), (ld), 2), band(base0, (i), (ld), 2), lp, lt2, y, tc);   \
  } while (0)
// (measured and dropped, round 3: whole-128-byte-line stores for NF = 4 -- the second 64-byte piece of a row rotated by 8
// lanes with DPP row_ror:8 so that one instruction writes 8 rows x 128 B.  tools/micro/store_bench.hip: a lone CU writes
// 64-byte segments at <= 24 GB/s and whole lines at >= 51 GB/s, but inside this epilogue nothing moved (1156 vs 1164 us at
// 256 CUs, 4.5 vs 4.75 us per tile at 128 CUs, with or without a half-period workgroup stagger: gpurun_out/r3/nt8_lines2.log)
// -- with every CU in its epilogue at once the stores run at the chip's HBM write rate, 5-6 TB/s.)
// The inverse of store_band_bf16: load one 16-row band of a bf16 array as 16 bytes per lane (8 consecutive columns of a
// fragment pair; 8 bytes for an odd last fragment) and hand every lane the 4 columns per fragment the accumulators use
// (v_permlane16_swap is its own inverse on a pair).  Half as many load instructions as one 8-byte load per fragment --
// 8-byte accesses run at 0.54-0.70 of the 16-byte rate (MI355X_MICROARCH.md).  raw[] is kept in the packed form so that
// the look-ahead costs the same registers as before.
template <int NF> struct BandRaw {
  uint4 pr[NF / 2 > 0 ? NF / 2 : 1];
  uint2 tail;
};
template <int NF> __device__ __forceinline__ void load_band_bf16(BandRaw<NF>& r, const char* ub, unsigned lo_pair, unsigned lo_tail) {
#pragma unroll
  for (int j = 0; j + 1 < NF; j += 2) r.pr[j / 2] = *(const uint4*)(ub + opaque(lo_pair) + 32 * j);
  if (NF & 1) r.tail = *(const uint2*)(ub + opaque(lo_tail) + 32 * (NF - 1));
}
template <int NF> __device__ __forceinline__ void unpack_band_bf16(const BandRaw<NF>& r, f32x4* h) {
#pragma unroll
  for (int j = 0; j + 1 < NF; j += 2) {
    const uint4 x = r.pr[j / 2];  // (a0, b0, a1, b1) of store_band_bf16
    auto l = __builtin_amdgcn_permlane16_swap(x.x, x.z, false, false);
    auto hh = __builtin_amdgcn_permlane16_swap(x.y, x.w, false, false);
    h[j] = unpack_bf16x4(make_uint2(l[0], hh[0]));
    h[j + 1] = unpack_bf16x4(make_uint2(l[1], hh[1]));
  }
  if (NF & 1) h[NF - 1] = unpack_bf16x4(r.tail);
}

// lane byte offsets into a bf16 [rows, ld] array for store_band_bf16 (fr = lane & 15, fg = lane >> 4)
__device__ __forceinline__ unsigned bf16_pair_offset(int fr, int fg, int ld) {
  return (unsigned)(fr * ld + ((fg & 1) ? 16 + 4 * (fg - 1) : 4 * fg)) * 2u;
}
__device__ __forceinline__ unsigned bf16_tail_offset(int fr, int fg, int ld) { return (unsigned)(fr * ld + 4 * fg) * 2u; }
// paired tail store (store_band_bf16): even 16-lane rows write band i's row fr, odd rows band i + 1's row fr (+ 16 rows)
__device__ __forceinline__ unsigned bf16_tail2_offset(int fr, int fg, int ld) {
  return (unsigned)((fr + ((fg & 1) ? 16 : 0)) * ld + ((fg & 1) ? 4 * (fg - 1) : 4 * fg)) * 2u;
}

// look-ahead (in 16-row bands) of the epilogue's row-dependent loads: as deep as the register file allows
constexpr int epi_depth(int E, int NF) {
  return E == E_DACT ? (NF >= 4 ? 4 : 8) : 0;
}

}  // namespace nt8

// WR = wave rows: 2 -> 256-row tile, 8 waves, one workgroup per CU (next-tile prefetch under the
// epilogue); 1 -> 128-row tile, 4 waves, TWO independent workgroups per CU, so one workgroup's
// epilogue (an HBM-write burst with idle matrix cores) runs under the other's K loop.
// SCHED = placement of the LDS-DMA refills (and fragment reads) inside a phase -- see PAIR_BODY; bit 4 = the
// round-2 addressing (per-lane 64-bit address arithmetic in the K loop) for A/B runs.
template <int NF, int WR, int E, int SCHED = NT8_DEFAULT_SCHED>
__global__ __launch_bounds__(256 * WR, 2) void gemm_nt8_kernel(NTParams p) {
  using namespace nt8;
  constexpr int SP = SCHED & 15;            // placement variant
  constexpr bool SADDR = !(SCHED & 16);     // scalar K-tile base + 32-bit lane offset (saddr-form LDS-DMA)
  // timing decomposition of the K loop (garbage results; tools/nt8_sched.py): skip the steady-state LDS-DMA refills /
  // the fragment reads / the phase barriers
  // fine-interleave (NT8_FINE) parameters: fragment-read stride, MFMA index behind which the A / B pieces of the
  // LDS-DMA refill are issued, j-major last half-phase for the single-buffered B of NF = 4
  constexpr int RS = SP == 9 ? 2 : 1;
  constexpr int QLAST = 4 * NF - 1;
  constexpr int QA = SP == 8 ? QLAST : SP == 9 ? (9 < QLAST ? 9 : QLAST) : 4;
  constexpr int QB = SP == 8 ? QLAST : SP == 9 ? QLAST : (8 < QLAST ? 8 : QLAST);
  constexpr bool FJ = NF == 4 && (SP == 5 || SP == 11);
  constexpr bool CONV = (SCHED & 4096) != 0;  // A operand gathered from an NHWC activation (implicit 3x3 convolution)
  constexpr bool CONV_FUSE = (SCHED & 8192) != 0;  // ... with the fused skip-connection / GroupNorm-statistics epilogue (NF = 2 only: the
                                                   // 256-wide tile sits at 256 VGPRs and spilled 38-48 registers with it)
  constexpr bool XPF = !(SCHED & 1024) && E != E_TRK;  // cross-tile prefetch inside the last K-tile pair (bit 10 = the round-2 burst, A/B runs)
  // (measured and dropped: non-temporal epilogue stores -- the plain-bf16 epilogue gets 8-17 % SLOWER, gpurun_out/r3/sched4.log)
  constexpr bool X_NODMA = (SCHED & 32) != 0, X_NOREAD = (SCHED & 64) != 0, X_NOBAR = (SCHED & 128) != 0;
  constexpr int BN8 = 64 * NF;
  constexpr int BM8 = 128 * WR;
  constexpr int RPP = 2 / WR;            // B LDS-DMA rounds per phase
  constexpr int BROWS = 32 * WR;         // B rows covered by one round (8 rows per wave)
  constexpr int A_BYTES = BM8 * 128;
  constexpr int STAGE = A_BYTES + BN8 * 128;
  constexpr int WN = 16 * NF;
  constexpr int LDS_BYTES = 2 * STAGE + (E == E_TRK ? 8192 : 0);  // E_TRK: 1 KiB per wave of DMA scratch
  static_assert(LDS_BYTES * (WR == 2 ? 1 : 2) <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  constexpr bool STAMPS = (SCHED & 512) != 0;
  int stamp_tile = 0;
  auto stamp = [&](int ev) {
    if (STAMPS && blockIdx.x == 0 && (tid == 0 || tid == 256 * WR - 64) && stamp_tile < 64)
      nt8_stamps[(tid ? 256 : 0) + stamp_tile * 4 + ev] = __builtin_readcyclecounter();
  };
  // the two waves of a SIMD are w and w + WAVES/2 (a workgroup's waves go round the four SIMDs): "first half"
  // = the first wave of each SIMD
  const bool first_half = wave < 2 * WR;

  const int tiles_m = p.M / BM8, tiles_n = p.N / BN8;
  const int ntiles = tiles_m * tiles_n;
  int vt = blockIdx.x;  // virtual tile id of this workgroup's current tile (stride gridDim.x)
  int tm, tn;
  // tile order: groups of 8 row tiles, 4 where the problem has few column tiles (N = 1152: 6) -- the 32 concurrent
  // workgroups of an XCD then share fewer distinct A rows (fc1 dgrad 1077 -> 1019 us, proj dgrad 310 -> 301; wide
  // problems prefer 8: gpurun_out/r3/nt8_group.log)
  // ... and 3 where N is a power of two (the 512- / 2048-wide decoder): with power-of-two row pitches the row tiles an
  // XCD works on at the same time sit at multiples of 512 KB - 2 MB, i.e. on the same HBM channels; an odd group spreads
  // them (decoder proj + GATE_RES 400 -> 337 us, fc1 + GELU 801 -> 740: gpurun_out/r3/nt8_groups_all.log)
  const int group_m = p.group_m > 0 ? p.group_m : ((p.N & (p.N - 1)) == 0 ? 3 : tiles_n <= 8 ? 4 : GROUP_M);
  tile_coords(xcd_remap(vt, ntiles), tiles_m, tiles_n, tm, tn, group_m);
  int m0 = tm * BM8, n0 = tn * BN8;

  // ---- LDS-DMA addressing.  One wave-instruction = 8 tile rows x 128 B; lane -> (row lane/8,
  // LDS chunk lane%8); the global chunk is XOR-swizzled with (row & 7) = lane/8.
  const int lr = lane >> 3, gch = (lane & 7) ^ lr;
  // A slot q: wave w covers tile rows (w>>2)*128 + 32q + 8(w&3) .. +7.  Addresses are kept as a wave-UNIFORM
  // 64-bit base (scalar registers, re-pointed per tile, advanced with scalar adds) plus one 32-bit per-lane byte
  // offset, so the LDS-DMA instructions take the (saddr + voffset) form and the K loop carries no 64-bit
  // vector address arithmetic.
  const int a_row0 = (wave >> 2) * 128 + 8 * (wave & 3);
  const char* a_u = (const char*)(p.A + (long)(m0 + a_row0) * p.lda);  // re-pointed per tile
  // B round j: wave w covers tile rows BROWS*j + 8w .. +7
  const char* b_u = (const char*)(p.B + (long)(n0 + 8 * wave) * p.ldb);
  // per-lane byte offsets of every slot / round (32-bit): the only vector address state of the K loop
  unsigned a_lo[4], b_lo[NF * RPP];
#pragma unroll
  for (int q = 0; q < 4; ++q) a_lo[q] = (unsigned)((lr + 32 * q) * p.lda + gch * 8) * 2u;
  // CONV: a_lo[q] instead holds the output pixel of this lane's row of slot q, packed x | y << 12 | b << 24 (re-derived
  // per tile); the source offset of a K-tile = the tap-shifted (and, with up-sampling, halved) pixel x C channels, or
  // the zero line in front of the activation for taps that fall outside the image
  auto conv_rows = [&](int m_first) {
    const int hl = p.conv_ho_log2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m_first + a_row0 + 32 * q + lr;
      const int pix = m & ((1 << (2 * hl)) - 1);
      a_lo[q] = (unsigned)((pix & ((1 << hl) - 1)) | ((pix >> hl) << 12) | ((m >> (2 * hl)) << 24));
    }
  };
  if constexpr (CONV) conv_rows(m0);
#pragma unroll
  for (int j = 0; j < NF * RPP; ++j) b_lo[j] = (unsigned)((lr + BROWS * j) * p.ldb + gch * 8) * 2u;
  const int a_lds0 = a_row0 * 128;           // + 32q*128 + stage*STAGE
  const int b_lds0 = A_BYTES + wave * 1024;  // + j*8192 + stage*STAGE

  auto issue = [&](int stage, int kt, int ph, int part = 3) {  // part: 1 = the A slot, 2 = the B rounds
    char* base = smem + stage * STAGE;
    // the empty asm makes a lane offset opaque at every use: hipcc would otherwise fold it into a per-lane
    // 64-bit base once and carry vector addresses (and their 64-bit adds) through the K loop; the K-tile base is
    // pinned in scalar registers the same way (otherwise the loop-invariant part of kt is re-associated to the
    // vector side: two v_lshl_add_u64 per LDS-DMA in the round-2 ISA)
    const char* ak = a_u + (long)kt * 128;
    const char* bk = b_u + (long)kt * 128;
    if (SADDR) { ak = sopaque(ak); bk = sopaque(bk); }
    if constexpr (CONV) {
      if (part & 1) {
        const int kk = kt * 64;                       // K index of this K-tile: tap * C + channel (C % 64 == 0)
        const int tap = kk / p.conv_c, c0 = kk - tap * p.conv_c;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int hl = p.conv_ho_log2, ho = 1 << hl, hi_l = hl - p.conv_up;
        const unsigned pk = opaque(a_lo[ph]);
        const int xx = (int)(pk & 0xfff) + dx, yy = (int)((pk >> 12) & 0xfff) + dy, bb = (int)(pk >> 24);
        const bool ok = (unsigned)xx < (unsigned)ho && (unsigned)yy < (unsigned)ho;
        const unsigned pix = (unsigned)((((bb << hi_l) + (yy >> p.conv_up)) << hi_l) + (xx >> p.conv_up));
        const unsigned off = ok ? 256u + (pix * (unsigned)p.conv_c + (unsigned)c0 + (unsigned)gch * 8u) * 2u : (unsigned)(lane & 7) * 16u;
        glds16((const char*)p.A + off, base + a_lds0 + ph * 4096);
      }
    } else if (part & 1) glds16(ak + opaque(a_lo[ph]), base + a_lds0 + ph * 4096);
    if ((part & 2) && ph < NF) {
#pragma unroll
      for (int r = 0; r < RPP; ++r) {
        glds16(bk + opaque(b_lo[ph * RPP + r]), base + b_lds0 + (ph * RPP + r) * (BROWS * 128));
      }
    }
  };

  // E_TRK: one 16-byte-per-lane memory operation per phase, alternating an fp32 store into this tile's outf rows
  // and an LDS-DMA load from its res rows (together ~ the bytes of a GATE_RES epilogue spread over a K = 1152 loop)
  auto trickle = [&](int slot, f32x4 v) {
    const int u = slot >> 1, band = u & 7, j = (u >> 3) % NF;  // the epilogue's (band, fragment) walk
    const long off = (long)(m0 + (wave >> 2) * 128 + band * 16 + (lane & 15)) * p.ldof + n0 + (wave & 3) * (16 * NF) + 16 * j +
                     4 * (lane >> 4);
    if (slot & 1) glds16(p.res + off, smem + 2 * STAGE + wave * 1024);
    else *(f32x4*)(p.outf + off) = v;
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

  // bias of the wave's columns (4 consecutive per fragment).  Loaded unconditionally (a conditional load is
  // waited for with vmcnt(0) on the spot; no bias = a zero row) at the START of a tile where the register file
  // allows (NF <= 3), so the epilogue finds it in registers; for NF = 4 at the start of the epilogue.
  constexpr bool EARLY_BIAS = NF < 4;
  f32x4 bias[NF];
  auto load_bias = [&](int c0) {
    const char* bp = p.bias ? (const char*)(p.bias + c0) : (const char*)(nt8_zero_row + (c0 & (NT8_ZERO_ROW - 1)));
    const unsigned lo_b = 16u * fg;
#pragma unroll
    for (int j = 0; j < NF; ++j) bias[j] = *(const f32x4*)(bp + opaque(lo_b) + 64 * j);
  };

  // ---- optional stagger (p.epi bit 9): every other workgroup starts half a tile period late so that
  // the epilogues (HBM bursts with idle matrix cores) of one half of the chip fall under the K
  // loops of the other half instead of all 256 CUs bursting in lock-step.
  if ((p.epi & 0x200) && (WR == 1 ? (blockIdx.x >= (gridDim.x >> 1)) : ((blockIdx.x & 8) != 0))) {
    // 4-wave form: the second workgroup of each CU (dispatched in the second half of the grid)
    // delay in 8192-cycle naps: (epi >> 16) & 0xff when given (mdt_set_tuning nt8_stagger = naps), else about half a tile period
    const int naps = ((p.epi >> 16) & 0xff) > 1 ? ((p.epi >> 16) & 0xff) : (((p.K >> 6) * 1500 + 8000) >> 14);
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // ---- prologue of the first tile: K-tiles 0 and 1 in steady-state issue order
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) issue(0, 0, ph);
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) issue(1, 1, ph);

  static_assert(SP == 5 || (SP >= 8 && SP <= 10), "only the fine-interleaved phase forms are compiled");
  if (SP == 10 && !first_half) __builtin_amdgcn_s_setprio(1);  // static priority for the second wave of each SIMD
  for (;;) {  // persistent tile loop
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the two K-tiles of this output tile were put in flight before the previous tile's epilogue
  // (or just above): everything older -- including that epilogue's stores -- must have retired
  wait_vm_lgkm<0>();
  asm volatile("; MDT_CHK vm_empty" ::: "memory");  // tools/check_waits.py: nothing in flight at the tile hand-over, on any path
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  stamp(0);
  // (returns in order right behind the two prefetched K-tiles; the counted waits of the K loop stay valid --
  // they only become marginally stricter for the first phases)
  if constexpr (EARLY_BIAS) load_bias(n0 + wc * WN);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) Ar[0][i][ks] = *(const bf16x8*)(smem + a_off[ks] + i * 2048);
#pragma unroll
    for (int j = 0; j < NF; ++j) Br[0][j][ks] = *(const bf16x8*)(smem + b_off[ks] + j * 2048);
  }

  // NOTE the operand order of the MFMA: (B fragment, A fragment) -> the accumulator holds the TRANSPOSED
  // 16x16 block: register r of lane l = C[row = l & 15][col = 4 * (l >> 4) + r].
  // Placement variants of a phase (SP; measured with tools/nt8_sched.py at M = 131072, K loop only, qkv forward /
  // 4608 x 1152 (NF 4), TFLOP/s, one box: gpurun_out/r3/sched{1,2,3}.log):
  //   0  reads + LDS-DMA at the phase start, then the MFMA cluster (round 2; 16 = with round 2's 64-bit vector
  //      address arithmetic: 1242 / 1350)                                                                1262 / 1364
  //   1  waves 0-3 as 0; waves 4-7 (the second wave of every SIMD) issue their LDS-DMA AFTER the cluster   (-1 %)
  //   2  every wave issues its LDS-DMA after the cluster                                                  1362 / 1450
  //   3  every wave issues its LDS-DMA between the ks = 0 and ks = 1 halves of the cluster                (+1..4 %)
  //   4  waves 0-3 as 0; waves 4-7 LDS-DMA in the middle                                                  (-5 %)
  //   5  ONE MEMORY INSTRUCTION PINNED BEHIND EACH MFMA (NT8_FINE), no setprio: THE PRODUCT FORM         1412 / 1512
  //   6  as 1, and waves 4-7 also issue their fragment reads in the middle of the cluster                 (-3 %)
  //   7  as 0 without s_setprio                                                                           1316 / 1423
  //   8 / 9 / 10  variations of 5 (LDS-DMA behind the last MFMA; reads behind every second MFMA; static s_setprio 1
  //      for waves 4-7): all within +-1.5 % of 5
  // In the clustered forms both waves of a SIMD queue their reads and LDS-DMA in front of an idle matrix pipe at every
  // phase start; interleaved, each memory instruction issues in the shadow of the partner wave's MFMA.  Decomposition
  // of 5 (SCHED bits 32 / 64 / 128, garbage results): no LDS-DMA 1615, no fragment reads 1700, no barriers 1460,
  // neither DMA nor reads 1836, MFMAs alone 1955-2054 (= the clock-limited matrix rate): what is left is the cost of
  // the memory instructions themselves (~14 matrix-pipe cycles per ds_read_b128, ~35 per LDS-DMA), not the barriers.
  // The counted waits are the same for every variant: per wave the ORDER of (issue, wait) events is unchanged.
  // Only form 5 (and its 8 / 9 / 10 parameterisations) is compiled since the cross-tile prefetch went in; the clustered
  // forms live in the round-3 history (git: 38727ef).
  // One phase (NT8_FINE): MFMA q is followed by memory instruction q -- the ds_reads of the next phase's fragments
  // first, then the A piece / the B piece of the LDS-DMA refill (QA / QB).  MODE: 0 = steady state (refill with K-tile
  // kt + 2 of THIS output tile); 1 = the last K-tile pair of a tile that has a successor: the refills fetch K-tiles 0 / 1
  // of the workgroup's NEXT tile, so the K loop's issue pattern -- and its counted waits -- simply continue across the
  // tile boundary (round 3; rounds 1-2 issued those 14-16 LDS-DMAs per wave as ONE burst after the K loop: the CU's
  // vector-memory path takes ~48 clocks per wave-instruction there, tools/nt8_stamps.py measured 7000 clocks = 15 % of
  // a K = 1152 tile for the last wave to get its burst out before it could start its epilogue); 2 = the last pair of
  // the workgroup's last tile (nothing to fetch, draining waits).
#define NT8_FINE(MODE)                                                                                \
      _Pragma("unroll") for (int q = 0; q < 4 * NF; ++q) {                                            \
        /* MFMA order: (ks, i, j); FJ (NF = 4, last phase): the ks = 1 half runs (j, i) so that B[j][1] dies early */ \
        const bool jm = FJ && ph == 3 && q >= 2 * NF;                                                 \
        const int ks = q / (2 * NF), i = jm ? (q & 1) : (q / NF) & 1, j = jm ? (q - 2 * NF) >> 1 : q % NF; \
        acc[2 * ph + i][j] = mfma16(Br[BDB ? half : 0][j][ks], Ar[ph & 1][i][ks], acc[2 * ph + i][j]); \
        if (X_NOREAD) {                                                                               \
        } else if (ph < 3) {                                                                          \
          if (q % RS == 0 && q / RS < 4) {                                                            \
            const int r = q / RS;                                                                     \
            Ar[(ph + 1) & 1][r & 1][r >> 1] = *(const bf16x8*)(cur + a_off[r >> 1] + (2 * (ph + 1) + (r & 1)) * 2048); \
          }                                                                                           \
        } else if (!(MODE != 0 && half == 1)) {                                                       \
          if (q < 4) Ar[0][q & 1][q >> 1] = *(const bf16x8*)(nxt + a_off[q >> 1] + (q & 1) * 2048);   \
          else if (BDB && q - 4 < 2 * NF)                                                             \
            Br[BDB ? (half ^ 1) : 0][(q - 4) % NF][(q - 4) / NF] = *(const bf16x8*)(nxt + b_off[(q - 4) / NF] + ((q - 4) % NF) * 2048); \
          else if (!BDB && !FJ && q >= 2 * NF && q - 2 * NF < NF)                                     \
            Br[0][q - 2 * NF][0] = *(const bf16x8*)(nxt + b_off[0] + (q - 2 * NF) * 2048);            \
          else if (!BDB && FJ && q >= NF + 1 && q <= 2 * NF)           /* B[j][0] dies at q = NF + j */ \
            Br[0][q - NF - 1][0] = *(const bf16x8*)(nxt + b_off[0] + (q - NF - 1) * 2048);            \
          else if (!BDB && FJ && q > 2 * NF && (q & 1))                /* B[j][1] dies at q = 2 NF + 2 j + 1 */ \
            Br[0][(q - 2 * NF) >> 1][1] = *(const bf16x8*)(nxt + b_off[1] + ((q - 2 * NF) >> 1) * 2048); \
        }                                                                                             \
        if (MODE != 2 && !X_NODMA && q == (ph < 3 ? QA : 3 * NF)) {                                   \
          if (MODE == 0) issue(half, kt + half + 2, ph, 1); else issue_next(half, ph, 1);             \
        }                                                                                             \
        if (MODE != 2 && !X_NODMA && q == (ph < 3 ? QB : 4 * NF - 1)) {                               \
          if (MODE == 0) issue(half, kt + half + 2, ph, 2); else issue_next(half, ph, 2);             \
        }                                                                                             \
        if (E == E_TRK && MODE == 0 && q == 2) trickle((kt + half) * 4 + ph, acc[2 * ph][0]);         \
        __builtin_amdgcn_sched_barrier(0);                                                            \
      }                                                                                               \
      if (!BDB && !FJ && ph == 3 && !(MODE != 0 && half == 1) && !X_NOREAD) {                         \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                \
          Br[0][j][1] = *(const bf16x8*)(nxt + b_off[1] + j * 2048);                                  \
      }
#define PAIR_BODY(MODE)                                                                               \
  _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                            \
    const char* cur = smem + half * STAGE;                                                            \
    const char* nxt = smem + (half ^ 1) * STAGE;                                                      \
    _Pragma("unroll") for (int ph = 0; ph < 4; ++ph) {                                                \
      NT8_FINE(MODE)                                                                                  \
      /* publish: my share of the next phase's data has landed, my LDS reads have retired */          \
      if (MODE == 2) {                                                                                \
        if (half == 0 && ph == 0) wait_vm_lgkm<drain_count(0, NF, RPP)>();                            \
        else if (half == 0 && ph == 1) wait_vm_lgkm<drain_count(1, NF, RPP)>();                       \
        else if (half == 0 && ph == 2) wait_vm_lgkm<drain_count(2, NF, RPP)>();                       \
        else if (half == 0 && ph == 3) wait_vm_lgkm<drain_count(3, NF, RPP)>();                       \
        else if (half == 1 && ph == 0) wait_vm_lgkm<drain_count(4, NF, RPP)>();                       \
        else if (half == 1 && ph == 1) wait_vm_lgkm<drain_count(5, NF, RPP)>();                       \
        else wait_vm_lgkm<0>();                                                                       \
      }                                                                                               \
      else if (ph == 0) wait_vm_lgkm<wait_count(0, NF, RPP, E == E_TRK)>();                           \
      else if (ph == 1) wait_vm_lgkm<wait_count(1, NF, RPP, E == E_TRK)>();                           \
      else if (ph == 2) wait_vm_lgkm<wait_count(2, NF, RPP, E == E_TRK)>();                           \
      else wait_vm_lgkm<wait_count(3, NF, RPP, E == E_TRK)>();                                        \
      if (!X_NOBAR) __builtin_amdgcn_s_barrier();                                                     \
      asm volatile("" ::: "memory");                                                                  \
    }                                                                                                 \
  }

  // (measured and dropped, round 3: leaving the epilogue's stores in flight across the tile boundary -- tile-top
  // vmcnt(#stores) and the first six phases' counts raised by the same number.  With an exact store count it changes
  // nothing (889 vs 894 us qkv forward, gpurun_out/r3/sched7.log): what the tile-top wait waits for is the next tile's
  // second K-tile, fetched during the last phases of the K loop, not the store acknowledgements.)
  // K-tile `stage` (0 / 1) of the NEXT tile (a_u / b_u already point at it when the last pair runs)
  auto issue_next = [&](int stage, int ph, int part) { issue(stage, stage, ph, part); };
  int kt = 0;
  for (; kt + 2 < nk; kt += 2) { PAIR_BODY(0) }
  // the workgroup's next output tile (if any): its first two K-tiles are fetched by the refills of the last pair
  const int em0 = m0 + wr * 128, en0 = n0 + wc * WN;  // origin of this wave's 128 x WN block of the CURRENT tile
  vt += gridDim.x;
  const bool more = vt < ntiles;
  if (more) {
    tile_coords(xcd_remap(vt, ntiles), tiles_m, tiles_n, tm, tn, group_m);
    m0 = tm * BM8;
    n0 = tn * BN8;
    a_u = (const char*)(p.A + (long)(m0 + a_row0) * p.lda);
    b_u = (const char*)(p.B + (long)(n0 + 8 * wave) * p.ldb);
    if constexpr (CONV) conv_rows(m0);
  }
  // (a workgroup's LAST tile re-fetches its own first two K-tiles into the freed slots instead of branching to a
  // draining variant: the steady-state waits stay valid, nothing reads those slots again, and a second copy of the
  // unrolled pair behind a run-time branch made hipcc spill 130-390 registers at the join)
  if constexpr (XPF) { PAIR_BODY(1) } else { PAIR_BODY(2) }
#undef PAIR_BODY
#undef NT8_FINE
  stamp(1);
  if (!XPF && more) {  // (rounds 1-2: the next tile's first two K-tiles as one burst after the K loop)
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) issue(0, 0, ph);
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) issue(1, 1, ph);
  }
  stamp(2);


  if (MDT_EXP(p.epi & 0x100) || E == E_TRK) {  // benchmarking aid (mdt_set_tuning "nt8_skip_epilogue"): main loop only
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) asm volatile("" ::"v"(acc[i][j]));
    if (!more) {
      wait_vm_lgkm<0>();
      break;
    }
    continue;
  }

  // ---- epilogue, straight out of the accumulators: this lane owns rows em0 + 16 i + fr (i = 0..7) and
  // columns en0 + 16 j + 4 fg .. +3 (j < NF).  Band i = the 16 rows of accumulator row-fragment i.
  // * Addresses = wave-uniform 64-bit base (scalar; + 16 i rows per band) + ONE 32-bit lane offset per array
  //   + an immediate: the epilogue holds a handful of address registers instead of one pair per access.
  // * Run-time options (optional outputs, column sums, which activation) select between straight-line bodies
  //   (generic lambdas over compile-time tags): no band contains a branch, hipcc counts vmcnt exactly and the
  //   stores of a tile are issued back to back.
  // * Every load of the tile is issued before the first use (sched_barrier): ONE memory round trip per tile.
  auto ubase = [&](const void* base, int ld, int es) { return (char*)base + ((long)em0 * ld + en0) * es; };
  auto band = [&](char* ub, int i, int ld, int es) { return ub + (long)i * 16 * ld * es; };
  if constexpr (!EARLY_BIAS) load_bias(en0);
  if constexpr (E == E_GATE) {
    // the bias goes into the accumulators before anything is loaded, so its registers are free for the
    // residual look-ahead
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] += bias[j];
#pragma unroll
    for (int j = 0; j < NF; ++j) bias[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int act = p.epi & 0xff;
  using T = std::true_type;
  using F = std::false_type;

  auto flush_colsum = [&](const f32x4* csum) {
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float s = csum[j][c];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
        if (fr == 0) atomic_add_f32(p.colsum + en0 + 4 * fg + 16 * j + c, s);
      }
  };

  if constexpr (E == E_PLAIN) {
    auto body = [&](auto cs) {
      constexpr bool CS = decltype(cs)::value;
      f32x4 csum[NF];
#pragma unroll
      for (int j = 0; j < NF; ++j) csum[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      char* ob = ubase(p.out, p.ldo, 2);
      const unsigned lp = bf16_pair_offset(fr, fg, p.ldo), lt2 = bf16_tail2_offset(fr, fg, p.ldo);
      TailCarry tc = {0u, 0u};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4 y[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) y[j] = acc[i][j] + bias[j];
        NT8_STORE_BAND(NF, ob, p.ldo, i, lp, lt2, y, tc);
        if (CS) {
#pragma unroll
          for (int j = 0; j < NF; ++j) csum[j] += round_bf16(y[j]);
        }
      }
      if (CS) flush_colsum(csum);
    };
    if (p.colsum) body(T{}); else body(F{});
  } else if constexpr (E == E_F32 && CONV && CONV_FUSE) {
    // implicit-GEMM convolution (mdt_conv3x3_nhwc): outf = acc + bias (+ res: the ResnetBlock skip connection,
    // autoencoder.py:129) and, optionally, the GroupNorm statistics of what is stored (the next layer's Normalize,
    // autoencoder.py:35-36) -- both used to be separate HBM passes (mdt_add_f32: 12 B / element, mdt_gn_stats: 4)
    auto body = [&](auto has_res, auto has_gn) {
      constexpr bool R = decltype(has_res)::value, GN = decltype(has_gn)::value;
      constexpr int D = 4;  // residual look-ahead in bands (4 NF registers per band)
      char* fb = ubase(p.outf, p.ldof, 4);
      const unsigned lf = (unsigned)(fr * p.ldof + 4 * fg) * 4u;
      char* rb = R ? ubase(p.res, p.ldres, 4) : nullptr;
      const unsigned lr_ = (unsigned)(fr * p.ldres + 4 * fg) * 4u;
      f32x4 pre[8][NF];
      if (R) {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int j = 0; j < NF; ++j) pre[i][j] = *(const f32x4*)(band(rb, i, p.ldres, 4) + opaque(lr_) + 64 * j);
        __builtin_amdgcn_sched_barrier(0);
      }
      float s1[NF], s2[NF];
#pragma unroll
      for (int j = 0; j < NF; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (R && i + D < 8) {
#pragma unroll
          for (int j = 0; j < NF; ++j) pre[i + D][j] = *(const f32x4*)(band(rb, i + D, p.ldres, 4) + opaque(lr_) + 64 * j);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          f32x4 y = acc[i][j] + bias[j];
          if (R) y += pre[i][j];
          *(f32x4*)(band(fb, i, p.ldof, 4) + opaque(lf) + 64 * j) = y;
          if (GN) {
            s1[j] += (y[0] + y[1]) + (y[2] + y[3]);
            s2[j] += (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
          }
        }
      }
      if (GN) {  // the wave's 128 rows lie in ONE sample (the entry point checks Ho * Ho % 128 == 0)
        float* gs = p.gn_sums + (long)(em0 >> (2 * p.conv_ho_log2)) * 64;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          float a = s1[j], b = s2[j];
          a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 8, 64);
          b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64); b += __shfl_xor(b, 4, 64); b += __shfl_xor(b, 8, 64);
          if (fr == 0) {
            const int grp = (en0 + 16 * j + 4 * fg) >> p.gn_cpg_log2;
            atomic_add_f32(gs + 2 * grp, a);
            atomic_add_f32(gs + 2 * grp + 1, b);
          }
        }
      }
    };
    if (p.res) { if (p.gn_sums) body(T{}, T{}); else body(T{}, F{}); }
    else { if (p.gn_sums) body(F{}, T{}); else body(F{}, F{}); }
  } else if constexpr (E == E_F32) {  // outf = acc + bias (the dispatcher sends "also bf16" requests elsewhere)
    char* fb = ubase(p.outf, p.ldof, 4);
    const unsigned lf = (unsigned)(fr * p.ldof + 4 * fg) * 4u;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) *(f32x4*)(band(fb, i, p.ldof, 4) + opaque(lf) + 64 * j) = acc[i][j] + bias[j];
  } else if constexpr (E == E_ACT) {
    // out = h = bf16(acc + bias) (optional), out2 = bf16(act(h))
    auto body = [&](auto keep_h, auto is_gelu) {
      constexpr bool KH = decltype(keep_h)::value, GELU = decltype(is_gelu)::value;
      char* hb = KH ? ubase(p.out, p.ldo, 2) : nullptr;
      char* ab = ubase(p.out2, p.ldo2, 2);
      const unsigned hp = bf16_pair_offset(fr, fg, p.ldo), ht2 = bf16_tail2_offset(fr, fg, p.ldo);
      const unsigned ap = bf16_pair_offset(fr, fg, p.ldo2), at2 = bf16_tail2_offset(fr, fg, p.ldo2);
      TailCarry tch = {0u, 0u}, tca = {0u, 0u};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4 y[NF], a[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          y[j] = round_bf16(acc[i][j] + bias[j]);
#pragma unroll
          for (int c = 0; c < 4; ++c) a[j][c] = GELU ? gelu_tanh(y[j][c]) : silu(y[j][c]);
        }
        if (KH) NT8_STORE_BAND(NF, hb, p.ldo, i, hp, ht2, y, tch);
        NT8_STORE_BAND(NF, ab, p.ldo2, i, ap, at2, a, tca);
      }
    };
    if (act == MDT_EPI_GELU) { if (p.out) body(T{}, T{}); else body(F{}, T{}); }
    else { if (p.out) body(T{}, F{}); else body(F{}, F{}); }
  } else if constexpr (E == E_GATE) {
    // y = bf16(acc + bias) (stored when out != NULL); outf = res + gate[sample] * y.  rows_per_sample % 64 == 0
    // (dispatcher), so each 64-row half of the wave's block lies in one sample.
    auto body = [&](auto keep_y, auto two_gates) {
      constexpr bool KY = decltype(keep_y)::value, G2 = decltype(two_gates)::value;
      constexpr int D = (NF >= 3) ? (G2 ? 6 : 8) : 8;  // residual look-ahead in bands (register budget)
      char* rb = ubase(p.res, p.ldres, 4);
      char* fb = ubase(p.outf, p.ldof, 4);
      char* yb = KY ? ubase(p.out, p.ldo, 2) : nullptr;
      const unsigned lr_ = (unsigned)(fr * p.ldres + 4 * fg) * 4u, lf = (unsigned)(fr * p.ldof + 4 * fg) * 4u;
      const unsigned yp = bf16_pair_offset(fr, fg, p.ldo), yt2 = bf16_tail2_offset(fr, fg, p.ldo);
      TailCarry tcy = {0u, 0u};
      const char* g0 = (const char*)(p.gate + (long)(em0 / p.rows_per_sample) * p.gate_ld + en0);
      const char* g1 = (const char*)(p.gate + (long)((em0 + 64) / p.rows_per_sample) * p.gate_ld + en0);
      const unsigned lg = 16u * fg;
      f32x4 pre[8][NF], gate[G2 ? 2 : 1][NF];
#pragma unroll
      for (int i = 0; i < (D < 8 ? D : 8); ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) pre[i][j] = *(const f32x4*)(band(rb, i, p.ldres, 4) + opaque(lr_) + 64 * j);
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        gate[0][j] = *(const f32x4*)(g0 + opaque(lg) + 64 * j);
        if (G2) gate[G2 ? 1 : 0][j] = *(const f32x4*)(g1 + opaque(lg) + 64 * j);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4 y[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) y[j] = round_bf16(acc[i][j]);
        if (i + D < 8) {
#pragma unroll
          for (int j = 0; j < NF; ++j) pre[i + D][j] = *(const f32x4*)(band(rb, i + D, p.ldres, 4) + opaque(lr_) + 64 * j);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) *(f32x4*)(band(fb, i, p.ldof, 4) + opaque(lf) + 64 * j) = pre[i][j] + gate[G2 ? (i >> 2) : 0][j] * y[j];
        if (KY) NT8_STORE_BAND(NF, yb, p.ldo, i, yp, yt2, y, tcy);
      }
    };
    // rows_per_sample % 128 == 0 (every shipped shape): the wave's 128-row block lies in ONE sample
    if (p.rows_per_sample % 128 == 0) { if (p.out) body(T{}, F{}); else body(F{}, F{}); }
    else { if (p.out) body(T{}, T{}); else body(F{}, T{}); }
  } else {  // E_DACT: out = bf16((acc + bias) * act'(aux)), optional column sums of the stored values
    auto body = [&](auto cs, auto is_gelu) {
      constexpr bool CS = decltype(cs)::value, GELU = decltype(is_gelu)::value;
      constexpr int D = epi_depth(E, NF);
      char* xb = ubase(p.aux, p.ldaux, 2);
      char* ob = ubase(p.out, p.ldo, 2);
      const unsigned lxp = bf16_pair_offset(fr, fg, p.ldaux), lxt = bf16_tail_offset(fr, fg, p.ldaux);
      const unsigned lp = bf16_pair_offset(fr, fg, p.ldo), lt2 = bf16_tail2_offset(fr, fg, p.ldo);
      TailCarry tc = {0u, 0u};
      BandRaw<NF> pre[8];  // saved pre-activation, 16 bytes per lane and fragment pair (round 2: one 8-byte load per fragment)
#pragma unroll
      for (int i = 0; i < (D < 8 ? D : 8); ++i) load_band_bf16<NF>(pre[i], band(xb, i, p.ldaux, 2), lxp, lxt);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 csum[NF];
#pragma unroll
      for (int j = 0; j < NF; ++j) csum[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4 y[NF];
#pragma unroll
        for (int j0 = 0; j0 < NF; j0 += 2) {  // one fragment pair at a time: short live ranges for the unpacked values
          f32x4 hb[2];
          if (j0 + 1 < NF) {
            const uint4 x = pre[i].pr[j0 / 2];
            auto l = __builtin_amdgcn_permlane16_swap(x.x, x.z, false, false);
            auto hh = __builtin_amdgcn_permlane16_swap(x.y, x.w, false, false);
            hb[0] = unpack_bf16x4(make_uint2(l[0], hh[0]));
            hb[1] = unpack_bf16x4(make_uint2(l[1], hh[1]));
          } else {
            hb[0] = unpack_bf16x4(pre[i].tail);
          }
#pragma unroll
          for (int jj = 0; jj < 2 && j0 + jj < NF; ++jj) {
            const int j = j0 + jj;
            const f32x4 v = acc[i][j] + bias[j];
#pragma unroll
            for (int c = 0; c < 4; ++c) y[j][c] = v[c] * (GELU ? gelu_tanh_grad(hb[jj][c]) : silu_grad(hb[jj][c]));
          }
        }
        if (i + D < 8) load_band_bf16<NF>(pre[i + D], band(xb, i + D, p.ldaux, 2), lxp, lxt);
        NT8_STORE_BAND(NF, ob, p.ldo, i, lp, lt2, y, tc);
        if (CS) {
#pragma unroll
          for (int j = 0; j < NF; ++j) csum[j] += round_bf16(y[j]);
        }
      }
      if (CS) flush_colsum(csum);
    };
    if (act == MDT_EPI_DGELU) { if (p.colsum) body(T{}, T{}); else body(F{}, T{}); }
    else { if (p.colsum) body(T{}, F{}); else body(F{}, F{}); }
  }
  stamp(3);
  ++stamp_tile;
  if (!more) {
    wait_vm_lgkm<0>();  // the dummy refills of the last pair must have landed before the workgroup's LDS is released
    break;
  }
  }  // persistent tile loop
}

int nt8_num_cus();

// one translation unit per epilogue class (gemm_nt8_c<class>.hip: #define NT8_CLASS <0..4>, then include this
// header and expand NT8_INSTANTIATE_CLASS) -- the five classes compile in parallel
#define NT8_CAT_(a, b) a##b
#define NT8_CAT(a, b) NT8_CAT_(a, b)
#define NT8_INST(NF, WR) template __global__ void gemm_nt8_kernel<NF, WR, NT8_CLASS>(NTParams);
#define NT8_LAUNCH(NF, WR) hipLaunchKernelGGL((gemm_nt8_kernel<NF, WR, NT8_CLASS>), dim3(grid), blk, 0, stream, p)
// NT8_WIDE = 1: the class has a 256 x 256 (NF = 4) instantiation (E_GATE keeps a 5-band residual look-ahead in
// registers next to the accumulators and stops at NF = 3)
#define NT8_INSTANTIATE_CLASS(NT8_WIDE, NT8_WIDE_INST)                                                            \
  NT8_INST(2, 2) NT8_INST(3, 2) NT8_INST(2, 1) NT8_INST(3, 1)                                       \
  NT8_WIDE_INST                                                                                     \
  int NT8_CAT(launch_gemm_nt8_class, NT8_CLASS)(const NTParams& p, int nf, int wr, hipStream_t stream) { \
    const int bm = 128 * wr;                                                                        \
    const int ntiles = (p.M / bm) * (p.N / (64 * nf));                                              \
    const int slots = nt8_num_cus() * (wr == 1 ? 2 : 1);                                            \
    const int grid = ntiles < slots ? ntiles : slots;                                               \
    const dim3 blk(256 * wr);                                                                       \
    if (nf == 4 && !(NT8_WIDE && wr == 2)) {                                                        \
      mdt_set_error("gemm_nt8: no 256-column instantiation for this epilogue class");               \
      return MDT_ERR_ARG;                                                                           \
    }                                                                                               \
    if (wr == 2) {                                                                                  \
      switch (nf) {                                                                                 \
        case 2: NT8_LAUNCH(2, 2); break;                                                            \
        case 3: NT8_LAUNCH(3, 2); break;                                                            \
        default: NT8_LAUNCH((NT8_WIDE ? 4 : 2), 2); break;                                          \
      }                                                                                             \
    } else {                                                                                        \
      if (nf == 3) NT8_LAUNCH(3, 1);                                                                \
      else NT8_LAUNCH(2, 1);                                                                        \
    }                                                                                               \
    return mdt_check_launch("gemm_nt8");                                                            \
  }
