// gemm_nt8o: the WAVE-SPECIALISED form of the large bf16 NT GEMM -- the fused epilogue of tile i runs UNDER the K loop
// of tile i + 1 ("waves that wait on operand LDS-DMAs never issue a global store").
//
// STATUS: bit-identical to gemm_nt8, MEASURED SLOWER on every XL/2 shape (profiles/r5_nt8o_*.txt):
// an A/B form behind mdt_set_tuning("nt8_overlap"), not on the product path.  What it established is in "Findings" below.
//
// gemm_nt8 (gemm_nt8_impl.h) serialises, per CU, a matrix-bound K loop (HBM idle) and an HBM-bound epilogue (matrix
// pipes idle): every wave does everything, and gfx950 retires a wave's loads AND stores through ONE in-order vmcnt
// counter, so epilogue traffic issued by a wave that also waits for operand LDS-DMAs turns L2-hit waits into HBM round
// trips (the round-2 E_TRK experiment).  Here the 8 waves of the one workgroup per CU have three ROLES:
//
//   waves 0-3      MMA       one per SIMD, 128 x 64 of the 256 x 128 output tile each (8 x 4 MFMA 16x16x32 accumulators):
//                            ds_read_b128 fragment reads + MFMAs, no vector-memory instruction inside the K loop.  At the
//                            tile's end y = bf16(acc + bias) goes to `out` as 16 plain 1 KiB stores per wave that nobody
//                            waits for (this wave has no loads to wait for either), then straight on to the next tile.
//   waves 4..3+NL  LOADERS   every operand LDS-DMA (global_load_lds, 16 B / lane; 48 pieces of 8 rows x 128 B per K-tile,
//                            48 / NL per loader) into a 3-stage ring of whole K-tiles (3 x 48 KiB); their vmcnt sees
//                            nothing but those pieces (tools/check_waits.py rule_nt8o).  NL = 2 or 3 (4 for plain bf16).
//   the other 4-NL EPILOGUE  own every global load / store of the fused epilogue beyond y itself: the tile's y comes
//                            back from `out` (L2: this CU's MMA waves wrote it a few K-tiles ago), the residual from HBM,
//                            fp32 residual / activation pair go out in whole 128-byte lines, all of it streamed with a
//                            14-group (GATE) / 24-group (activation) register look-ahead while the MMA waves are K-tiles
//                            into the next tile.  The residual rows of tile T + 1 are requested BEFORE tile T + 1 exists.
//
// The product epilogues round y = acc + bias to bf16 FIRST and compute everything else from the rounded value
// (gemm_nt8_impl.h E_GATE / E_ACT / E_PLAIN), so handing the tile over as bf16 is bit-identical.  (First form of this file:
// the hand-over went through LDS into 128 registers of the epilogue waves -- the 3 x 48 KiB ring leaves no staging room, so
// the halves overlaid the ring stage the tile's last K-tile had vacated; with the tile in registers the epilogue waves had
// 4 groups of look-ahead left before spilling.  y is an OUTPUT of every training launch anyway; read back through L2 it
// costs no HBM traffic and no registers, and the whole lent-stage protocol went away.  `out == NULL` -> gemm_nt8.)
//
// Synchronisation is by monotonic LDS counters, NOT s_barrier (one barrier at kernel start only): with one MMA wave per
// SIMD every workgroup barrier is a matrix-pipe bubble nobody fills (round 3's 4-wave experiment), and a counter wait
// that is already satisfied costs one broadcast ds_read issued a phase earlier.
//   full[s]   += 1 by each loader when its share of a fill of stage s has landed   (fill n ready   <=> full[s]  == NL n)
//   empty[s]  += 1 by each MMA wave after its last read of a fill of stage s       (fill n drained <=> empty[s] == 4 n)
//   ydone[w]  += 1 by MMA wave w once its y stores of a tile are acknowledged      (tile T in L2   <=> every ydone[w] >= T + 1)
// (ONE word per MMA wave: with a single summed counter, two waves that have finished the workgroup's last tile and
// published it could lift the sum over the threshold of the tile BEFORE while the other two waves -- up to three K-tiles
// behind, the ring allows that -- had not acknowledged theirs: possible for K < 512, found by the exhaustive interleaving
// check tools/nt8o_protocol_model.py, never observed on hardware.  full / empty are exact by construction: nobody can add
// for fill n + 1 of a stage before everybody has added for fill n.)
// Every spin is bounded; a wave that gives up poisons every counter (all later waits fall through, results garbage, the
// launch terminates) and raises nt8o_abort, which mdt_nt8o_report returns -- a protocol bug must not hang the GPU.
//
// Findings (MI355X, M = 131072, us per launch; tools/nt8o_bench.py, profiles/r5_nt8o_bench.txt):
//   * correct on the first GPU run and bit-identical on every shape / class / loader count; the guard has never fired;
//   * proj + GATE_RES (N = K = 1152): product 514-534, this kernel 592-602.  Decomposition of this kernel: MFMAs + fragment
//     reads + counters 244-259 (1.34-1.42 PF: ONE wave per SIMD cannot hide its own ds_read latency the way gemm_nt8's two
//     do -- its no-DMA figure is 1.6 PF), + the ring's LDS-DMAs 294-310, + the MMA waves' y stores 349-364, + the epilogue
//     waves' y read-back 366-382, + their residual loads 441-451 or their stores 493-501, everything 592-602.  With the
//     epilogue waves running, the MMA waves wait for operands 53 % of their lifetime while the loaders wait for a free stage
//     only 7 %: the LOADERS are held up -- their L2-hit LDS-DMAs queue behind the epilogue waves' HBM-latency loads and
//     the store back-pressure in the CU's one in-order vector-memory pipeline.  So the serialisation the round-2 E_TRK
//     experiment saw is NOT an artefact of one wave's in-order vmcnt: it is a property of the CU's TA / L1 path, which no
//     assignment of instructions to waves removes;
//   * fc2 + GATE_RES (K = 4608) 1275-1329 -> 1526-1556, fc1 + GELU 1317-1379 -> 1763-1810, plain bf16 (4 loaders, stores
//     only) 294-1102 -> 362-1292: slower everywhere; three loader waves instead of two change nothing (the LDS-DMA rate is
//     a CU limit, ~ 40 clocks per 1 KiB piece); decoder shapes likewise (347 -> 375, 698 -> 737, 732 -> 975).
//
// Requirements (nt8o_eligible): M % 256 == 0, N % 128 == 0, K % 64 == 0, K >= 256, out != NULL, epilogue BF16 (no colsum)
// / GELU / SILU / GATE_RES with rows_per_sample % 64 == 0; row alignments as gemm_nt8.  Reference arithmetic replaced: the
// nn.Linear + GELU / `x + gate * f(x)` sites of models/maskdit.py:178-191 (see include/maskdit_hip.h mdt_gemm_nt).
#include "common.h"
#include "../../include/maskdit_hip.h"
#include "gemm_common.h"
#include <type_traits>

namespace nt8o {
constexpr int BM = 256, BN = 128;
constexpr int A_BYTES = BM * 128;         // 256 rows x 64 bf16
constexpr int B_BYTES = BN * 128;
constexpr int STAGE = A_BYTES + B_BYTES;  // 48 KiB
constexpr int NS = 3;
enum { F_FULL = 0, F_EMPTY = 4, F_YDONE = 8 /* .. 11: one per MMA wave */, F_COUNT = 16 };
constexpr unsigned POISON = 0x40000000u;
constexpr int SPIN_LIMIT = 1 << 15;       // x (~64-clock nap + LDS round trip) ~ 3 ms
constexpr int DEPTH = 14;                 // GATE epilogue look-ahead in 4-row groups (12 registers each: 2 x 16 B residual + 16 B y per lane)
enum { E_PLAIN = 0, E_ACT = 2, E_GATE = 3 };
// debug / timing-decomposition bits (experiments build only; results garbage except bit 0)
enum { DBG_STATS = 1, DBG_NO_EPI_MEM = 2, DBG_NO_DMA = 4, DBG_NO_RES_LOAD = 8, DBG_NO_STORE = 16, DBG_NO_YSTORE = 32 };
// stall-clock sums written by STATS launches (one wave of each role per workgroup, summed over workgroups)
enum { S_MMA_FULL = 0, S_LD_EMPTY = 3, S_EP_DUMP = 5, S_MMA_TOTAL = 6, S_EP_TOTAL = 7, S_LD_TOTAL = 8, S_WGS = 9, S_COUNT = 16 };
}  // namespace nt8o

static __device__ float nt8o_zero_row[1024 + 64];                        // stands in for a NULL bias (zero-initialised)
static __device__ unsigned nt8o_abort;                                   // != 0: some wave gave up a bounded spin
static __device__ unsigned long long nt8o_stats[nt8o::S_COUNT];
// STATS launches: shader-clock stamps of workgroup 0, per role (0 MMA wave 0, 1 loader 0, 2 epilogue wave 0) and tile (first 32):
// [0] = the role starts the tile (MMA: first MFMA phase; loader: first piece issued; epilogue: tile published and picked up),
// [1] = it is done with it (MMA: K loop finished, y stores about to issue; loader: last piece issued; epilogue: last store issued)
static __device__ unsigned long long nt8o_stamps[3 * 32 * 2];

namespace nt8o {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;  // (HIP's uint4 is a struct: an array of them went through scratch)
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// The counters are touched with INLINE-ASM LDS instructions on purpose: hipcc's waitcnt pass makes every LDS instruction
// it can see wait (vmcnt(0)) for the LDS-DMAs the issuing wave has in flight -- it cannot tell a counter from a ring slot
// -- which drained the loaders' queue at every poll and right behind the counted vmcnt(12) (first build of this file).
// Untracked LDS operations only make hipcc's own counted lgkmcnt waits more conservative (LDS returns in order).
__device__ __forceinline__ unsigned flag_ld_async(unsigned addr) {  // the value is valid after flag_settle()
  unsigned v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ unsigned flag_settle(unsigned v) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
  return v;
}
__device__ __forceinline__ unsigned flag_ld(unsigned addr) { return flag_settle(flag_ld_async(addr)); }
__device__ __forceinline__ void flag_add(unsigned addr) {  // += 1 by ONE lane, no branch
  unsigned long long save;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0"
               : "=&s"(save)
               : "v"(addr), "v"(1u)
               : "memory");
}
// s_waitcnt with only the named counter constrained (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  // (comments in the ISA for tools/check_waits.py: rule_nt8o audits the loop around a counted loader wait -- nothing but
  // LDS-DMAs, exactly N of them between the loop head and the wait and N after it)
  if (N > 0) asm volatile("; MDT_CHK nt8o_loader_wait" ::: "memory");
  asm volatile("; MDT_CHK hand_wait" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_lgkm0() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));
  asm volatile("" ::: "memory");
}

// Wait until the counter at LDS byte address `addr` is >= need.  `seen` = a value of the counter read earlier (the fast
// path costs nothing but the compare).  flags0 = address of counter 0 (for the give-up path).
template <bool STATS>
__device__ __forceinline__ void wait_ge(unsigned flags0, int idx, unsigned need, unsigned seen, int lane, unsigned long long& stall, unsigned& gave_up) {
  if (__builtin_amdgcn_readfirstlane(seen) >= need) return;
  unsigned long long t0 = 0;
  if (STATS) t0 = __builtin_readcyclecounter();
  int it = 0;
  for (; it < SPIN_LIMIT; ++it) {
    const unsigned v = __builtin_amdgcn_readfirstlane(flag_ld(flags0 + 4u * idx));
    if (v >= need) break;
    __builtin_amdgcn_s_sleep(1);
  }
  if (it == SPIN_LIMIT) {  // give up: let every other wait of the workgroup fall through and report
    if (lane < F_COUNT) asm volatile("ds_write_b32 %0, %1" ::"v"(flags0 + 4u * lane), "v"(POISON) : "memory");
    gave_up = 1u + (unsigned)idx;  // reported when the wave leaves (a store HERE would sit in the loaders' counted vmcnt window)
  }
  if (STATS) stall += __builtin_readcyclecounter() - t0;
}
template <bool STATS>
__device__ __forceinline__ void wait_ge(unsigned flags0, int idx, unsigned need, int lane, unsigned long long& stall, unsigned& gave_up) {
  wait_ge<STATS>(flags0, idx, need, flag_ld(flags0 + 4u * idx), lane, stall, gave_up);
}

// ... until FOUR consecutive counters (idx .. idx + 3, 16-byte aligned) are all >= need: the per-MMA-wave ydone words
template <bool STATS>
__device__ __forceinline__ void wait_ge4(unsigned flags0, int idx, unsigned need, int lane, unsigned long long& stall, unsigned& gave_up) {
  unsigned long long t0 = 0;
  bool timed = false;
  int it = 0;
  for (; it < SPIN_LIMIT; ++it) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flags0 + 4u * idx) : "memory");
    const unsigned m = min(min(v[0], v[1]), min(v[2], v[3]));
    if (__builtin_amdgcn_readfirstlane(m) >= need) break;
    if (STATS && !timed) {
      t0 = __builtin_readcyclecounter();
      timed = true;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (it == SPIN_LIMIT) {
    if (lane < F_COUNT) asm volatile("ds_write_b32 %0, %1" ::"v"(flags0 + 4u * lane), "v"(POISON) : "memory");
    gave_up = 1u + (unsigned)idx;
  }
  if (STATS && timed) stall += __builtin_readcyclecounter() - t0;
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  bf16x2 t;
  t[0] = f2bf(lo);
  t[1] = f2bf(hi);
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ f32x4 unpack_lo4(u32x4 u) {  // columns 0..3 of a lane's 8
  return (f32x4){__builtin_bit_cast(float, u[0] << 16), __builtin_bit_cast(float, u[0] & 0xffff0000u),
                 __builtin_bit_cast(float, u[1] << 16), __builtin_bit_cast(float, u[1] & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 unpack_hi4(u32x4 u) {  // columns 4..7
  return (f32x4){__builtin_bit_cast(float, u[2] << 16), __builtin_bit_cast(float, u[2] & 0xffff0000u),
                 __builtin_bit_cast(float, u[3] << 16), __builtin_bit_cast(float, u[3] & 0xffff0000u)};
}

}  // namespace nt8o

// the ring and its counters are SEPARATE LDS objects: hipcc's waitcnt pass makes a ds_read wait (vmcnt) for every
// in-flight LDS-DMA whose destination it may alias -- a counter poll inside the ring's array would drain the loaders'
// queue at every poll
// NL = loader waves (2..4), the other 4 - NL helper waves run the epilogue (E_PLAIN has none: NL = 4)
template <int E, int NL, int DBG>
__global__ __launch_bounds__(512, 2) void gemm_nt8o_kernel(NTParams p) {
  using namespace nt8o;
  constexpr int NE = 4 - NL;            // epilogue waves
  constexpr int NP = 48 / NL;           // LDS-DMA pieces per loader and K-tile (32 A + 16 B pieces of 8 rows x 128 B)
  static_assert(NL >= 2 && NL <= 4 && (E == E_PLAIN) == (NL == 4), "role split");
  constexpr bool STATS = (DBG & DBG_STATS) != 0;
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
  __shared__ __attribute__((aligned(16))) unsigned flags_mem[F_COUNT];  // (ydone[0..3] are read with one ds_read_b128)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < F_COUNT) flags_mem[tid] = 0;
  __syncthreads();  // the only workgroup barrier of the kernel
  const unsigned flags = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)flags_mem;  // LDS byte address

  const int tiles_m = p.M / BM, tiles_n = p.N / BN;
  const int ntiles = tiles_m * tiles_n;
  const int nk = p.K >> 6;  // >= 4
  const int group_m = p.group_m > 0 ? p.group_m : ((p.N & (p.N - 1)) == 0 ? 3 : tiles_n <= 8 ? 4 : GROUP_M);
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;  // >= 1: grid <= ntiles
  unsigned long long st_a = 0, t_begin = 0;
  unsigned gave_up = 0;
  if (STATS) t_begin = __builtin_readcyclecounter();

  if (wave < 4) {
    // ================================================================ MMA waves
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    unsigned a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned sw = (unsigned)(((ks * 4 + fg) ^ (fr & 7)) << 4);  // the loaders' XOR swizzle: chunk ^ (row & 7)
      a_off[ks] = (unsigned)((wr * 128 + fr) * 128) + sw;
      b_off[ks] = (unsigned)(A_BYTES + (wc * 64 + fr) * 128) + sw;
    }
    // bf16 store of a fragment PAIR (j, j + 1): after v_permlane16_swap every lane holds 8 consecutive columns (16 bytes)
    // -- even 16-lane rows columns 16 j + 4 fg .. + 7, odd rows 16 (j + 1) + 4 (fg - 1) .. + 7 (gemm_nt8's store_band_bf16)
    const unsigned y_lo = (unsigned)(fr * p.ldo + ((fg & 1) ? 16 + 4 * (fg - 1) : 4 * fg)) * 2u;

    f32x4 acc[8][4];
    bf16x8 Ar[2][2][2];  // [set][fragment of the phase][ks]
    bf16x8 Br[4][2];     // [fragment][ks]: replaced by the next K-tile's inside phase 3, as each one dies
    f32x4 bias[4];

    int st = 0;         // stage of the current K-tile
    unsigned use = 1;   // ... which is that stage's use-th fill
    wait_ge<STATS>(flags, F_FULL + 0, (unsigned)NL, lane, st_a, gave_up);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) Ar[0][i][ks] = *(const bf16x8*)(smem + a_off[ks] + i * 2048);
#pragma unroll
      for (int j = 0; j < 4; ++j) Br[j][ks] = *(const bf16x8*)(smem + b_off[ks] + j * 2048);
    }
    // the y stores of tile T - 1 are acknowledged (vmcnt(0)) and published to the epilogue waves a few K-tiles INTO tile T:
    // this wave issues no other vector-memory instruction in between, so nothing ever waits for them earlier
    const int sig_kt = nk > 4 ? 4 : nk - 1;
    int vt = blockIdx.x;
    for (int T = 0; T < my_tiles; ++T, vt += gridDim.x) {
      int tm, tn;
      tile_coords(xcd_remap(vt, ntiles), tiles_m, tiles_n, tm, tn, group_m);
      tm = __builtin_amdgcn_readfirstlane(tm);  // (the divisions run on the vector ALU: pin the uniform results in SGPRs)
      tn = __builtin_amdgcn_readfirstlane(tn);
      {  // bias of this wave's 64 columns (4 consecutive per fragment); no bias = a zero row (unconditional load)
        const int c0 = tn * BN + wc * 64;
        const char* bp = p.bias ? (const char*)(p.bias + c0) : (const char*)(nt8o_zero_row + (c0 & 1023));
        const unsigned lo_b = 16u * fg;
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = *(const f32x4*)(bp + opaque(lo_b) + 64 * j);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

      // One phase = 16 MFMAs with ONE LDS instruction pinned behind each of the first ones (the product kernel's
      // NT8_FINE order for the 128 x 64 wave tile): phases 0-2 read the next phase's A fragments from the current stage,
      // phase 3 reads the next K-tile's first A fragments and ALL its B fragments (each B register right after its last
      // use: the ks = 1 half of phase 3 runs fragment-major so that they die one by one).
#define NT8O_PHASE(ph)                                                                                              \
  _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                                  \
    const bool jm = (ph) == 3 && q >= 8;                                                                            \
    const int ks = q >> 3, i = jm ? (q & 1) : (q >> 2) & 1, j = jm ? (q - 8) >> 1 : q & 3;                          \
    acc[2 * (ph) + i][j] = mfma16(Br[j][ks], Ar[(ph) & 1][i][ks], acc[2 * (ph) + i][j]);                           \
    if ((ph) < 3) {                                                                                                 \
      if (q < 4) Ar[((ph) + 1) & 1][q & 1][q >> 1] = *(const bf16x8*)(smem + a_cur[q >> 1] + (2 * ((ph) + 1) + (q & 1)) * 2048); \
      if ((ph) == 2 && q == 5) seen = flag_ld_async(flags + 4u * (F_FULL + stn));                                                \
    } else {                                                                                                        \
      if (q < 4) Ar[0][q & 1][q >> 1] = *(const bf16x8*)(smem + a_nxt[q >> 1] + (q & 1) * 2048);                    \
      else if (q >= 5 && q <= 8) Br[q - 5][0] = *(const bf16x8*)(smem + b_nxt[0] + (q - 5) * 2048);                 \
      else if (q > 8 && (q & 1)) Br[(q - 8) >> 1][1] = *(const bf16x8*)(smem + b_nxt[1] + ((q - 8) >> 1) * 2048);   \
    }                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
      if (STATS && blockIdx.x == 0 && wave == 0 && lane == 0 && T < 32) nt8o_stamps[(0 * 32 + T) * 2] = __builtin_readcyclecounter();
      for (int kt = 0; kt < nk; ++kt) {
        const int stn = st == NS - 1 ? 0 : st + 1;
        const unsigned usen = use + (stn == 0 ? 1u : 0u);
        const bool has_next = (kt + 1 < nk) || (T + 1 < my_tiles);
        unsigned a_cur[2], a_nxt[2], b_nxt[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          a_cur[ks] = a_off[ks] + (unsigned)(st * STAGE);
          a_nxt[ks] = a_off[ks] + (unsigned)(stn * STAGE);
          b_nxt[ks] = b_off[ks] + (unsigned)(stn * STAGE);
        }
        if (E != E_PLAIN && T > 0 && kt == sig_kt) {  // tile T - 1 is in L2: hand it to the epilogue waves
          wait_vmcnt<0>();
          flag_add(flags + 4u * (F_YDONE + wave));
        }
        unsigned seen = 0;
        NT8O_PHASE(0)
        NT8O_PHASE(1)
        NT8O_PHASE(2)
        // the next K-tile must have landed before phase 3 reads its first fragments (the very last K-tile of the
        // workgroup has no successor: its phase-3 reads fetch stale bytes nobody uses)
        if (has_next) wait_ge<STATS>(flags, F_FULL + stn, NL * usen, flag_settle(seen), lane, st_a, gave_up);
        // phase 2 issued this wave's last reads of the current stage; LDS executes a wave's instructions in order, so
        // the counter moves after they have read
        flag_add(flags + 4u * (F_EMPTY + st));
        NT8O_PHASE(3)
        st = stn;
        use = usen;
      }
#undef NT8O_PHASE
      if (STATS && blockIdx.x == 0 && wave == 0 && lane == 0 && T < 32) nt8o_stamps[(0 * 32 + T) * 2 + 1] = __builtin_readcyclecounter();
      // ---- y = bf16(acc + bias) goes to `out` from HERE, as plain stores nobody waits for: 16 x 1 KiB per wave, then
      // straight on to the next tile's K loop (its first fragments are already in registers)
      if (!(DBG & DBG_NO_YSTORE)) {
        char* ub = (char*)p.out + ((long)(tm * BM + wr * 128) * p.ldo + tn * BN + wc * 64) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          unsigned lo[4], hi[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 v = acc[i][j] + bias[j];
            lo[j] = pack2(v[0], v[1]);
            hi[j] = pack2(v[2], v[3]);
          }
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            auto a = __builtin_amdgcn_permlane16_swap(lo[j], lo[j + 1], false, false);
            auto b = __builtin_amdgcn_permlane16_swap(hi[j], hi[j + 1], false, false);
            *(u32x4*)(ub + (long)i * 16 * p.ldo * 2 + opaque(y_lo) + 32 * j) = (u32x4){a[0], b[0], a[1], b[1]};
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3] + bias[j][0]));
      }
    }
    if (E != E_PLAIN) {  // the last tile
      wait_vmcnt<0>();
      flag_add(flags + 4u * (F_YDONE + wave));
    }
    if (STATS && lane == 0 && wave == 0) {
      atomicAdd(&nt8o_stats[S_MMA_FULL], st_a);
      atomicAdd(&nt8o_stats[S_MMA_TOTAL], __builtin_readcyclecounter() - t_begin);
      atomicAdd(&nt8o_stats[S_WGS], 1ull);
    }
  } else if (wave < 4 + NL) {
    // ================================================================ loader waves
    // One LDS-DMA instruction ("piece") = 8 tile rows x 128 B (lane -> row lane / 8, LDS chunk lane % 8; the global chunk
    // is XOR-swizzled with row & 7 = lane / 8, which makes the MMA waves' ds_read_b128 conflict-free).  A K-tile = 48 pieces
    // -- 32 of A, then 16 of B, which is also their order in the stage -- and loader L moves pieces NP L .. NP L + NP - 1.
    const int lr = lane >> 3, gch = (lane & 7) ^ lr;
    const unsigned a_lo = (unsigned)(lr * p.lda + gch * 8) * 2u, b_lo = (unsigned)(lr * p.ldb + gch * 8) * 2u;
    const long a_step = (long)p.lda * 16, b_step = (long)p.ldb * 16;  // bytes per 8 rows
    // (one straight-line copy of the walk per loader index: which of its pieces are A and which are B is then a compile-time fact)
    auto walk = [&](auto Lc) {
    constexpr int L = decltype(Lc)::value;
    constexpr int u0 = NP * L, ub0 = u0 > 32 ? u0 - 32 : 0;  // first piece; first B piece this loader ever touches
    int st = 0, st_prev = 0;
    unsigned use = 1;
    bool first = true;
    int vt = blockIdx.x;
    for (int T = 0; T < my_tiles; ++T, vt += gridDim.x) {
      int tm, tn;
      tile_coords(xcd_remap(vt, ntiles), tiles_m, tiles_n, tm, tn, group_m);
      tm = __builtin_amdgcn_readfirstlane(tm);
      tn = __builtin_amdgcn_readfirstlane(tn);
      const char* a_u = (const char*)(p.A + (long)(tm * BM) * p.lda) + u0 * a_step;
      const char* b_u = (const char*)(p.B + (long)(tn * BN) * p.ldb) + ub0 * b_step;
      for (int kt = 0; kt < nk; ++kt) {
        // the K-tile that lived in this stage (three K-tiles back in the walk) has been read by all four MMA waves
        if (use > 1) wait_ge<STATS>(flags, F_EMPTY + st, 4u * (use - 1), lane, st_a, gave_up);
        char* dst = smem + st * STAGE + u0 * 1024;
        const char* pa = sopaque(a_u + (long)kt * 128);
        const char* pb = sopaque(b_u + (long)kt * 128);
#define NT8O_PIECES(T0, T1)                                      \
  if (!(DBG & DBG_NO_DMA)) {                                     \
    _Pragma("unroll") for (int t = (T0); t < (T1); ++t) {        \
      if (u0 + t < 32) {                                         \
        glds16(pa + opaque(a_lo), dst + t * 1024);               \
        pa += a_step;                                            \
      } else {                                                   \
        glds16(pb + opaque(b_lo), dst + t * 1024);               \
        pb += b_step;                                            \
      }                                                          \
    }                                                            \
  }
        if (STATS && blockIdx.x == 0 && L == 0 && lane == 0 && T < 32 && kt == 0) nt8o_stamps[(1 * 32 + T) * 2] = __builtin_readcyclecounter();
        NT8O_PIECES(0, NP / 2)
        if (!first) {
          // NP / 2 pieces of THIS K-tile may be in flight; loads retire in order, so everything older -- the previous
          // K-tile -- has landed: publish it
          if (!(DBG & DBG_NO_DMA)) wait_vmcnt<NP / 2>();
          flag_add(flags + 4u * (F_FULL + st_prev));
        }
        NT8O_PIECES(NP / 2, NP)
#undef NT8O_PIECES
        if (STATS && blockIdx.x == 0 && L == 0 && lane == 0 && T < 32 && kt == nk - 1) nt8o_stamps[(1 * 32 + T) * 2 + 1] = __builtin_readcyclecounter();
        first = false;
        st_prev = st;
        st = st == NS - 1 ? 0 : st + 1;
        use += (st == 0 ? 1u : 0u);
      }
    }
    wait_vmcnt<0>();
    flag_add(flags + 4u * (F_FULL + st_prev));
    };
    switch (wave - 4) {
      case 0: walk(std::integral_constant<int, 0>{}); break;
      case 1: walk(std::integral_constant<int, 1>{}); break;
      case 2: walk(std::integral_constant<int, NL >= 3 ? 2 : 0>{}); break;
      default: walk(std::integral_constant<int, NL >= 4 ? 3 : 0>{}); break;
    }
    if (STATS && lane == 0 && wave == 4) {
      atomicAdd(&nt8o_stats[S_LD_EMPTY], st_a);
      atomicAdd(&nt8o_stats[S_LD_TOTAL], __builtin_readcyclecounter() - t_begin);
    }
  } else if (NE > 0 && !(DBG & DBG_NO_EPI_MEM)) {
    // ================================================================ epilogue waves
    // Wave e owns tile rows 128 e .. + 127 (the blocks of MMA waves (e, 0) and (e, 1)), all 128 columns.  A lane owns 8
    // CONSECUTIVE columns of a row (16 lanes per row, 4 rows per wave-instruction): fp32 arrays as two 16-byte accesses,
    // bf16 arrays as one -- the access shape that streams at 5.2-5.5 TB/s (tools/micro/stream_bench.hip); every row is
    // written in whole 128-byte lines (gemm_nt8's accumulator-shaped epilogue writes 64-byte segments).  The tile's y
    // comes back from `out` (L2: this CU's MMA waves wrote it a K-tile ago), everything streams with a DEPTH-group
    // register look-ahead; nothing but `ydone` ties these waves to the rest of the workgroup.
    constexpr int ROWS = 256 / (NE > 0 ? NE : 1), NG = ROWS / 4;  // rows / 4-row groups per epilogue wave
    const int e = wave - 4 - NL;
    const int r4 = lane >> 4, c = lane & 15;
    const int act = p.epi & 0xff;
    using TT = std::true_type;
    using FF = std::false_type;
    int vt = blockIdx.x;
    for (int T = 0; T < my_tiles; ++T, vt += gridDim.x) {
      int tm, tn;
      tile_coords(xcd_remap(vt, ntiles), tiles_m, tiles_n, tm, tn, group_m);
      tm = __builtin_amdgcn_readfirstlane(tm);
      tn = __builtin_amdgcn_readfirstlane(tn);
      const int m0 = tm * BM + ROWS * e, n0 = tn * BN;
      // group G (0..31): rows m0 + 4 G + r4, columns n0 + 8 c .. + 7.  Addresses = wave-uniform 64-bit base (+ 4 G rows)
      // + one 32-bit lane offset per array.
      auto rows = [&](const void* base, int ld, int es, int G) { return (char*)base + ((long)(m0 + 4 * G) * ld + n0) * es; };
      const unsigned ly = (unsigned)(r4 * p.ldo + 8 * c) * 2u;
      u32x4 Y[NG];
      auto ldy = [&](int G) { Y[G] = *(const u32x4*)(rows(p.out, p.ldo, 2, G) + opaque(ly)); };
      if constexpr (E == E_ACT) {
        constexpr int D = 24;
        wait_ge4<STATS>(flags, F_YDONE, (unsigned)(T + 1), lane, st_a, gave_up);
        if (STATS && blockIdx.x == 0 && e == 0 && lane == 0 && T < 32) nt8o_stamps[(2 * 32 + T) * 2] = __builtin_readcyclecounter();
        const unsigned la = (unsigned)(r4 * p.ldo2 + 8 * c) * 2u;
        auto body = [&](auto is_gelu) {
          constexpr bool GELU = decltype(is_gelu)::value;
#pragma unroll
          for (int G = 0; G < D; ++G) ldy(G);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int G = 0; G < NG; ++G) {
            if (G + D < NG) ldy(G + D);
            const f32x4 ya = unpack_lo4(Y[G]), yb = unpack_hi4(Y[G]);
            f32x4 aa, ab;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              aa[q] = GELU ? gelu_tanh(ya[q]) : silu(ya[q]);
              ab[q] = GELU ? gelu_tanh(yb[q]) : silu(yb[q]);
            }
            *(u32x4*)(rows(p.out2, p.ldo2, 2, G) + opaque(la)) =
                (u32x4){pack2(aa[0], aa[1]), pack2(aa[2], aa[3]), pack2(ab[0], ab[1]), pack2(ab[2], ab[3])};
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        if (act == MDT_EPI_GELU) body(TT{}); else body(FF{});
      } else {  // E_GATE: outf = res + gate[sample] * y.  rows_per_sample % 64 == 0: one sample per 64-row half
        constexpr int D = NE == 1 ? DEPTH - 2 : DEPTH;  // (one epilogue wave holds four gate rows: 16 registers more)
        const unsigned lr_ = (unsigned)(r4 * p.ldres + 8 * c) * 4u, lf = (unsigned)(r4 * p.ldof + 8 * c) * 4u;
        f32x4 gate[ROWS / 64][2];
#pragma unroll
        for (int h = 0; h < ROWS / 64; ++h) {
          const char* gp = (const char*)(p.gate + (long)((m0 + 64 * h) / p.rows_per_sample) * p.gate_ld + n0);
          gate[h][0] = *(const f32x4*)(gp + opaque(32u * c));
          gate[h][1] = *(const f32x4*)(gp + opaque(32u * c) + 16);
        }
        f32x4 r0[NG], r1[NG];
        auto ldr = [&](int G) {
          if (DBG & DBG_NO_RES_LOAD) {
            r0[G] = r1[G] = (f32x4){1.f, 1.f, 1.f, 1.f};
            return;
          }
          const char* rp = rows(p.res, p.ldres, 4, G) + opaque(lr_);
          r0[G] = *(const f32x4*)rp;
          r1[G] = *(const f32x4*)(rp + 16);
        };
        // the residual rows do not depend on this tile's GEMM: their first D groups are requested BEFORE the wait
#pragma unroll
        for (int G = 0; G < D; ++G) ldr(G);
        __builtin_amdgcn_sched_barrier(0);
        wait_ge4<STATS>(flags, F_YDONE, (unsigned)(T + 1), lane, st_a, gave_up);
        if (STATS && blockIdx.x == 0 && e == 0 && lane == 0 && T < 32) nt8o_stamps[(2 * 32 + T) * 2] = __builtin_readcyclecounter();
#pragma unroll
        for (int G = 0; G < D; ++G) ldy(G);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int G = 0; G < NG; ++G) {
          if (G + D < NG) {
            ldr(G + D);
            ldy(G + D);
          }
          const f32x4 o0 = r0[G] + gate[G >> 4][0] * unpack_lo4(Y[G]);
          const f32x4 o1 = r1[G] + gate[G >> 4][1] * unpack_hi4(Y[G]);
          if (DBG & DBG_NO_STORE) {
            asm volatile("" ::"v"(o0[0] + o0[1] + o0[2] + o0[3] + o1[0] + o1[1] + o1[2] + o1[3]));
          } else {
            char* fp = rows(p.outf, p.ldof, 4, G) + opaque(lf);
            *(f32x4*)fp = o0;
            *(f32x4*)(fp + 16) = o1;
          }
          __builtin_amdgcn_sched_barrier(0);  // (otherwise hipcc hoists every load of the tile to the top and spills)
        }
      }
      if (STATS && blockIdx.x == 0 && e == 0 && lane == 0 && T < 32) nt8o_stamps[(2 * 32 + T) * 2 + 1] = __builtin_readcyclecounter();
    }
    if (STATS && lane == 0 && wave == 4 + NL) {
      atomicAdd(&nt8o_stats[S_EP_DUMP], st_a);
      atomicAdd(&nt8o_stats[S_EP_TOTAL], __builtin_readcyclecounter() - t_begin);
    }
  }
  if (gave_up) {
    if (lane == 0) __hip_atomic_store(&nt8o_abort, gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef MDT_EXPERIMENTS
    // Product library: a wave that gave up a bounded spin has let every other wait of its workgroup fall through, so the
    // launch's outputs are garbage -- and the launch would still return MDT_OK (only mdt_nt8o_report, a test / tool entry,
    // reads nt8o_abort).  A training job that selected this A/B form through MDT_TUNE must not continue on corrupted
    // activations: the wave raises a hardware exception, the queue is torn down and the process ends with the
    // runtime's "HSA_STATUS_ERROR_EXCEPTION" instead.  (The experiments build keeps running: its tools read the code.)
    __builtin_trap();
#endif
  }
}

int nt8_num_cus();

// can mdt_gemm_nt's problem `p` (already validated for gemm_nt8's alignment rules) run on the overlap kernel?
bool nt8o_eligible(const NTParams& p) {
  const int epi = p.epi & 0xff;
  if (p.M % 256 || p.N % 128 || p.K % 64 || p.K < 256 || p.colsum || p.k_splits > 1) return false;
  if (!p.out) return false;  // y reaches the epilogue waves through `out` (training keeps it; inference runs gemm_nt8)
  if (epi == MDT_EPI_BF16 || epi == MDT_EPI_GELU || epi == MDT_EPI_SILU) return true;
  if (epi == MDT_EPI_GATE_RES) return p.rows_per_sample % 64 == 0;
  return false;
}

#define NT8O_GO(E, D) hipLaunchKernelGGL((gemm_nt8o_kernel<E, (E == nt8o::E_PLAIN ? 4 : NLV), D>), dim3(grid), dim3(512), 0, stream, p); break;
#ifdef MDT_EXPERIMENTS
// timing decompositions (tools/nt8o_bench.py): 1 = stall clocks; 3 = no epilogue waves; 9 / 17 / 25 = epilogue without its
// residual loads / its stores / both; 33 = no y stores; 35 = K loop + ring only; 39 = ... without LDS-DMA (MFMA + reads)
#define NT8O_DBG_CASES(E) case 1: NT8O_GO(E, 1) case 3: NT8O_GO(E, 3) case 9: NT8O_GO(E, 9) case 17: NT8O_GO(E, 17) \
  case 25: NT8O_GO(E, 25) case 33: NT8O_GO(E, 33) case 35: NT8O_GO(E, 35) case 39: NT8O_GO(E, 39)
#define NT8O_DBG_STATS(E) case 1: NT8O_GO(E, 1)
#else
#define NT8O_DBG_CASES(E)
#define NT8O_DBG_STATS(E)
#endif

// nl: loader waves of the GELU / GATE_RES classes (2 or 3); dbg: 0 = product, else nt8o::DBG_* bit sets (experiments build)
int launch_gemm_nt8o(const NTParams& p, int nl, int dbg, hipStream_t stream) {
  const int ntiles = (p.M / nt8o::BM) * (p.N / nt8o::BN);
  const int slots = nt8_num_cus();
  const int grid = ntiles < slots ? ntiles : slots;
  switch (p.epi & 0xff) {
    case MDT_EPI_BF16: {
      constexpr int NLV = 4;
      switch (dbg) { NT8O_DBG_STATS(nt8o::E_PLAIN) default: NT8O_GO(nt8o::E_PLAIN, 0) }
      break;
    }
    case MDT_EPI_GELU:
    case MDT_EPI_SILU:
      if (nl == 3) {
        constexpr int NLV = 3;
        switch (dbg) { NT8O_DBG_STATS(nt8o::E_ACT) default: NT8O_GO(nt8o::E_ACT, 0) }
      } else {
        constexpr int NLV = 2;
        switch (dbg) { NT8O_DBG_STATS(nt8o::E_ACT) default: NT8O_GO(nt8o::E_ACT, 0) }
      }
      break;
    case MDT_EPI_GATE_RES:
      if (nl == 3) {
        constexpr int NLV = 3;
        switch (dbg) { NT8O_DBG_CASES(nt8o::E_GATE) default: NT8O_GO(nt8o::E_GATE, 0) }
      } else {
        constexpr int NLV = 2;
        switch (dbg) { NT8O_DBG_CASES(nt8o::E_GATE) default: NT8O_GO(nt8o::E_GATE, 0) }
      }
      break;
    default: mdt_set_error("gemm_nt8o: no overlap form for this epilogue"); return MDT_ERR_ARG;
  }
  return mdt_check_launch("gemm_nt8o");
}

// tools/nt8o_bench.py --stamps: the per-tile time line of workgroup 0 written by the last STATS launch (3 roles x 32 tiles x 2)
extern "C" int mdt_nt8o_stamps(unsigned long long* host192) {
  if (hipDeviceSynchronize() != hipSuccess) return MDT_ERR_LAUNCH;
  return hipMemcpyFromSymbol(host192, HIP_SYMBOL(nt8o_stamps), sizeof(unsigned long long) * 192) == hipSuccess ? MDT_OK : MDT_ERR_LAUNCH;
}

// Host-side report of the overlap kernel's bounded-spin guard and (STATS launches) stall clocks.  Synchronises the
// device: test / tool support, not part of the training path.  abort_code: 0 = no wave ever gave up.
extern "C" int mdt_nt8o_report(unsigned* abort_code, unsigned long long* stats16, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return MDT_ERR_LAUNCH;
  unsigned a = 0;
  unsigned long long s[nt8o::S_COUNT] = {0};
  if (hipMemcpyFromSymbol(&a, HIP_SYMBOL(nt8o_abort), sizeof(a)) != hipSuccess) return MDT_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(s, HIP_SYMBOL(nt8o_stats), sizeof(s)) != hipSuccess) return MDT_ERR_LAUNCH;
  if (abort_code) *abort_code = a;
  if (stats16)
    for (int i = 0; i < nt8o::S_COUNT; ++i) stats16[i] = s[i];
  if (reset) {
    a = 0;
    for (int i = 0; i < nt8o::S_COUNT; ++i) s[i] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(nt8o_abort), &a, sizeof(a)) != hipSuccess) return MDT_ERR_LAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(nt8o_stats), s, sizeof(s)) != hipSuccess) return MDT_ERR_LAUNCH;
  }
  return MDT_OK;
}
