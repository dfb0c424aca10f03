// gemm_tn8: the large-problem weight-gradient GEMM  out[x, y] += sum_m X[m, x] * Y[m, y]
// (contraction over ROWS; X, Y row-major bf16; fp32 atomic accumulation, split over m).
//
// 256 (x) x 128 or 192 (y) output tile per 512-thread workgroup: 8 waves as 4(x) x 2(y), wave tile
// 64 x 64 or 64 x 96 = 4 x 4 / 4 x 6 MFMA 16x16x32 accumulators.  The contraction index is the slow (row) index of
// both operands, so tiles are kept in LDS exactly as they lie in memory ([m][n], 256-byte rows per
// 128-column half) and MFMA operands are fetched with the gfx950 LDS transpose read
// (ds_read_b64_tr_b16, two per fragment).  Pipeline: a ring of NSLOT slots of 32 contraction rows
// (one MFMA k-step); phase g = [tr-read the fragments of slot g+1 into the alternate register set]
// [LDS-DMA refill of the slot that was read during phase g-1 with rows of phase g+NSLOT]
// [16 / 24 MFMAs] [s_waitcnt vmcnt((NSLOT-2) * loads per slot) lgkmcnt(0)] [s_barrier]: every load has NSLOT-2 phases
// to land and vmcnt(0) only appears in the last ring pass.  Same swizzle as the 128x128 kernel
// (gemm.hip: tn_swz) through the DMA source address.
//
// Used by mdt_gemm_tn when both widths are multiples of 128, nothing is masked, and the row count
// per split is a multiple of 32 (>= NSLOT slots); ragged 256-wide x tiles clamp their source
// columns (never stored).  `swap` stores the tile transposed (C[y, x]) so that the caller can give
// the 256-wide role to whichever operand divides by 256.
#include "gemm_common.h"

// Two tile shapes (template YF = 16-column Y fragments per wave):
//   YF = 4: 256 x 128, ring of 6 slots (24 KiB each).  16 MFMAs per 16 transpose reads per wave and phase: the LDS
//           port is busy 100 % of the MFMA time and a CU pulls 47 B/clk from L2 -- the round-1 shape, kept for widths
//           that only divide by 128 (the 512-wide decoder).
//   YF = 6: 256 x 192, ring of 5 slots (28 KiB each), wave tile 64 x 96: 24 MFMAs per 20 transpose reads (LDS 83 %),
//           36 B/clk/CU from L2.  Every XL/2 encoder weight gradient has a 1152-wide side = 6 x 192.
//           The extra 64 columns of Y live in a second LDS piece with 128-byte rows (own swizzle); waves 0-3 move it,
//           so their vmcnt counts carry 4 loads per slot and those of waves 4-7 carry 3.
template <int YF> struct TN8Cfg {
  static constexpr int NSLOT = YF == 6 ? 5 : 6;
  static constexpr int SLOT_BYTES = YF == 6 ? 28672 : 24576;  // X half0 8K | X half1 8K | Y[0,128) 8K | Y[128,192) 4K
  static constexpr int TY = 32 * YF;
  static constexpr int UNROLL = YF == 6 ? 10 : 6;  // lcm(ring slots, 2 register sets)
};

__device__ __forceinline__ int tn8_swz(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }
// 128-byte-row piece: 4 units of 32 B per row, two rows per 256-byte bank line.  The 8 rows one 32-lane half of a
// transpose read touches (8g + j, g in a pair, j = 0..3) must land on 8 distinct (row & 1, unit) positions.
__device__ __forceinline__ int tn8_swz2(int r) { return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1; }

struct TN8Params {
  const bf16* X; int ldx;
  const bf16* Y; int ldy;
  int NX, NY;          // widths (NY % tile width == 0; NX % 128 == 0)
  float* C; int ldc;
  int swap;            // 0: C[x*ldc + y]   1: C[y*ldc + x]
  int slots_total;     // contraction rows / 32
  int slots_per_split;
  int tiles_x, tiles_y;
  float* colsum_x;     // optional [NX]: += column sums of X over this launch's rows (see the kernel)
  int group_x;         // X tiles per group of the tile order (the 32 concurrent workgroups of an XCD cover group_x x 32/group_x tiles)
  int dbg;             // timing experiments only (knob tn8_dbg): 1 = no slot refills, 2 = no fragment reads, 4 = refills re-read the first slots
  // "long + tail" partition of the contraction (round 6; long_k = 0: uniform splits).  The first tiles * long_k blocks are
  // LONG: every tile gets long_k of them, each walking long_len slots -- in lock-step over the same rows, like the uniform
  // splits, so operand rows are shared in L2.  They occupy tiles * long_k of the CUs for the whole launch; the rows that are
  // left (slots >= long_k * long_len) are one TAIL block per tile, which the dispatcher hands to the remaining CUs as they
  // come free.  Blocks per CU drop from ~3 (756 uniform blocks on 256 CUs for the XL/2 fc1 / fc2 weight gradients) to 1 on
  // five CUs of six; each block costs ~32 us on top of its slots (tools/tn8_fixed_cost.py).
  int long_k, long_len;
};

// ds_read_b64_tr_b16 through inline asm: with the builtin, hipcc's waitcnt pass assumes the read may
// alias the in-flight LDS-DMA writes and puts `s_waitcnt vmcnt(0)` in front of the first transpose
// read of every phase -- draining the whole pipeline.  The asm form is invisible to that pass; its
// completion is covered by the explicit lgkmcnt(0) at the end of the phase, and a sched_barrier
// after the s_barrier keeps every consumer (MFMA) behind that wait.
template <int OFF> __device__ __forceinline__ bf16x4 tn8_tr_read(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  short4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return __builtin_bit_cast(bf16x4, v);
}

template <int N> __device__ __forceinline__ void tn8_wait() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  asm volatile("; MDT_CHK hand_wait" ::: "memory");  // (a comment in the ISA: tools/check_waits.py audits the immediate that follows)
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

// SWAP = false: out tile rows = x (C[x, y]);  SWAP = true: the MFMA operands trade places so that the
// accumulator tile is [y rows][x cols] and the atomics to C[y, x] stay lane-contiguous.
// FINE (round 3): no half-phase stagger; instead ONE memory operation (a fragment = two transpose reads, or one LDS-DMA
// piece) pinned behind each of the phase's first MFMAs -- the form that gave gemm_nt8's K loop +13 %.
// CS: the fused column sums of X (a template flag since round 3: their 13 registers and the branch in the middle of
// every phase are only paid by the one launch per block that uses them, the qkv weight gradient).
template <bool SWAP, int YF, bool FINE, bool CS>
__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(TN8Params p) {
  using Cfg = TN8Cfg<YF>;
  constexpr int NSLOT = Cfg::NSLOT, SLOT_BYTES = Cfg::SLOT_BYTES;
  constexpr int NSET = (NSLOT + 2) / 3;
  static_assert(2 * SLOT_BYTES + 1024 < 65536, "three slots per address set");
  __shared__ __attribute__((aligned(16))) char smem[NSLOT * SLOT_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = wave >> 1, wy = wave & 1;
  const bool mover2 = YF == 6 && wave < 4;  // this wave also moves the 64-column piece of Y

  const int tiles = p.tiles_x * p.tiles_y;
  int tile, s_begin, S;
  if (p.long_k == 0) {  // uniform splits: block = (split, tile), XCD-contiguous ids
    const int sid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = sid / tiles;
    tile = sid - split * tiles;
    s_begin = split * p.slots_per_split;
    S = min(p.slots_per_split, p.slots_total - s_begin);  // >= NSLOT (host guarantees)
  } else {
    const int nlong = tiles * p.long_k;
    if ((int)blockIdx.x < nlong) {  // long block (dispatched first: lowest ids): row group sid / tiles of tile sid % tiles
      const int sid = xcd_remap(blockIdx.x, nlong);
      tile = sid % tiles;
      s_begin = (sid / tiles) * p.long_len;
      S = p.long_len;
    } else {  // tail block: the left-over rows of one tile
      tile = blockIdx.x - nlong;
      s_begin = p.long_k * p.long_len;
      S = p.slots_total - s_begin;
    }
  }
  int tx, ty;
  tile_coords(tile, p.tiles_x, p.tiles_y, tx, ty, p.group_x);
  const int x0 = tx * 256, y0 = ty * Cfg::TY;
  if (S <= 0) return;

  // ---- LDS-DMA addressing: per slot a wave moves rows 4w..4w+3 of X half 0, X half 1 and Y[0,128);
  // waves 0-3 also rows 8w..8w+7 of Y[128,192).
  const int lr = lane >> 4, cpos = lane & 15;
  const int rloc = 4 * wave + lr;  // slot-local row
  const int gch = cpos ^ tn8_swz(rloc);
  const int xc0 = min(x0 + gch * 8, p.NX - 8), xc1 = min(x0 + 128 + gch * 8, p.NX - 8);  // clamp ragged tiles
  const int rloc2 = 8 * (wave & 3) + (lane >> 3);
  // wave-uniform 64-bit bases (scalar registers, advanced per slot with scalar arithmetic) + one 32-bit lane offset per
  // stream: the LDS-DMA instructions take the saddr + voffset form (same idiom as gemm_nt8_impl.h)
  const char* const xu = (const char*)(p.X + (long)s_begin * 32 * p.ldx);
  const char* const yu = (const char*)(p.Y + (long)s_begin * 32 * p.ldy + y0);
  const unsigned x_lo0 = (unsigned)(rloc * p.ldx + xc0) * 2u, x_lo1 = (unsigned)(rloc * p.ldx + xc1) * 2u;
  const unsigned y_lo = (unsigned)(rloc * p.ldy + gch * 8) * 2u;
  const unsigned y_lo2 = (unsigned)(rloc2 * p.ldy + 128 + (((lane & 7) ^ tn8_swz2(rloc2)) << 3)) * 2u;
  const long x_step = 64L * p.ldx, y_step = 64L * p.ldy;  // bytes per slot
  char* const lds_w = smem + wave * 1024;

  auto issue = [&](int slot, int s, int parts = 15) {  // fill ring slot `slot` with contraction rows of phase s
    if (MDT_EXP(p.dbg & 4)) s &= 7;  // timing experiment: re-read the first 8 slots (cache-resident source)
    char* base = lds_w + slot * SLOT_BYTES;
    if (parts & 1) glds16(xu + s * x_step + opaque(x_lo0), base);
    if (parts & 2) glds16(xu + s * x_step + opaque(x_lo1), base + 8192);
    if (parts & 4) glds16(yu + s * y_step + opaque(y_lo), base + 16384);
    if ((parts & 8) && YF == 6 && mover2) glds16(yu + s * y_step + opaque(y_lo2), base + 24576);
  };
  // wait until all but this wave's loads of the newest `SLOTS` ring slots have landed
#define TN8_WAIT_SLOTS(SLOTS)                                     \
  {                                                               \
    if (YF == 6 && mover2) tn8_wait<4 * (SLOTS)>();               \
    else tn8_wait<3 * (SLOTS)>();                                 \
  }
  // The two waves of a SIMD (w and w + 4) run half a phase apart: while waves 0-3 issue their transpose reads and
  // LDS-DMA (half A), waves 4-7 issue MFMAs (half B), and vice versa -- in lock-step both would queue ~24 memory
  // instructions in front of an idle matrix pipe every phase.  One workgroup barrier per half; waves 4-7 take one
  // extra barrier before the loop, waves 0-3 one after it.  Consequence for the ring protocol: a slot read by waves
  // 0-3 in their half A of phase ph+1 needs waves 4-7's portion published one half earlier, so the late group waits
  // for one slot more (NSLOT-3 outstanding instead of NSLOT-2).
  // Measured alternatives at 131072 x 1152 x 4608 (tools/tn8_bench.py; this form 1165 us): lock-step waves 1233 us;
  // no stagger but one memory instruction pinned after each MFMA 1280 us; an L2 prefetch stream (one divergent dword
  // load per wave and phase touching the lines of the slot NSLOT phases ahead) 1719 us -- the extra vector-memory
  // instruction delays the DMAs queued behind it far more than the earlier L2 fill saves.  Where the time goes (knob
  // tn8_dbg): MFMAs + barriers alone 828 us, + transpose reads 871, + refills from cache-resident rows 1020, + refills
  // of the real stream (HBM latency) 1165.
  const bool late = !FINE && wave >= 4;
  // ---- optional column sums of X (the bias gradient of the layer whose weight gradient this is): the 32 x 256 X slot
  // of every phase is in LDS anyway.  The tiles_y workgroups that share an X tile split its 32 16-byte column chunks
  // between them; thread (row = tid >> 4, c = tid & 15) reads chunk c_begin + c of slot row `row` -- one ds_read_b128 per
  // phase in the same half (and under the same lgkmcnt(0)) as the transpose reads of that slot, consumed in the MFMA half.
  constexpr bool do_cs = CS;
  const int cs_begin = ty * 32 / p.tiles_y, cs_n = (ty + 1) * 32 / p.tiles_y - cs_begin;
  const int cs_row = tid >> 4, cs_c = tid & 15;
  const bool cs_on = do_cs && cs_c < cs_n;
  const int cs_chunk = cs_begin + (cs_c < cs_n ? cs_c : 0);  // 0..31 inside the 256-column X tile
  const unsigned cs_addr = (unsigned)(size_t)LDS_PTR(smem) + (cs_chunk >> 4) * 8192 + cs_row * 256 +
                           (((cs_chunk & 15) ^ tn8_swz(cs_row)) << 4);
  float cs_acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs_acc[e] = 0.f;
  bf16x8 cs_v;
#pragma unroll
  for (int e = 0; e < 8; ++e) cs_v[e] = (bf16)0.f;
#define TN8_CS_READ(slot)                                                                                   \
  if (do_cs) {                                                                                              \
    uint4 raw;                                                                                              \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(raw) : "v"(cs_addr + ((slot) / 3) * 3 * SLOT_BYTES), \
                 "n"(((slot) % 3) * SLOT_BYTES));                                                           \
    cs_v = __builtin_bit_cast(bf16x8, raw);                                                                 \
  }
#define TN8_CS_ADD()                                                                                        \
  if (cs_on) {                                                                                              \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) cs_acc[e] += bf2f(cs_v[e]);                               \
  }
#define TN8_WAIT_PHASE()                                          \
  {                                                               \
    if (late) tn8_wait<3 * (NSLOT - 3)>();                        \
    else TN8_WAIT_SLOTS(NSLOT - 2)                                \
  }
#define TN8_BARRIER()                                             \
  {                                                               \
    __builtin_amdgcn_s_barrier();                                 \
    asm volatile("" ::: "memory");                                \
    __builtin_amdgcn_sched_barrier(0);                            \
  }

  // ---- transpose-read addressing (gemm.hip gemm_tn_kernel): the 16-lane group g reads slot rows
  // 8g + 4t + (0..3) x 16 columns; lane i16 supplies row (i16>>2), 8-byte piece (i16&3).
  // Wave (wx, wy): X columns 64 wx + 16 i; Y columns 64 wy + 16 j (j < 4) and, YF = 6, 128 + 32 wy + 16 (j - 4).
  const int i16 = lane & 15, g = lane >> 4;
  const int trow = 8 * g + (i16 >> 2);
  const int sw = tn8_swz(trow), sw2 = tn8_swz2(trow);
  const int row_off = trow * 256 + ((i16 & 1) << 3);
  const int row_off2 = trow * 128 + ((i16 & 1) << 3);
  const int q1 = (i16 & 3) >> 1;
  const int cx0 = (wx & 1) * 8 + q1, cy0 = wy * 8 + q1;  // + 2*frag
  // LDS byte addresses per fragment, one set per THREE ring slots so that the remaining
  // (slot % 3) * SLOT_BYTES + row offset of t fits the 16-bit instruction offset
  const unsigned lds_base = (unsigned)(size_t)LDS_PTR(smem);
  unsigned xa[NSET][4], ya[NSET][YF];
#pragma unroll
  for (int pr = 0; pr < NSET; ++pr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xa[pr][i] = lds_base + pr * 3 * SLOT_BYTES + (wx >> 1) * 8192 + row_off + (((cx0 + 2 * i) ^ sw) << 4);
      ya[pr][i] = lds_base + pr * 3 * SLOT_BYTES + 16384 + row_off + (((cy0 + 2 * i) ^ sw) << 4);
    }
#pragma unroll
    for (int j = 4; j < YF; ++j)
      ya[pr][j] = lds_base + pr * 3 * SLOT_BYTES + 24576 + row_off2 + (((wy * 4 + 2 * (j - 4) + q1) ^ sw2) << 4);
  }

  f32x4 acc[4][YF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < YF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 Xr[2][4], Yr[2][YF];

#define TN8_LOAD(set, slot)                                                                             \
  {                                                                                                     \
    constexpr int so = ((slot) % 3) * SLOT_BYTES;                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
      Xr[set][i] = cat4(tn8_tr_read<so>(xa[(slot) / 3][i]), tn8_tr_read<so + 1024>(xa[(slot) / 3][i])); \
      Yr[set][i] = cat4(tn8_tr_read<so>(ya[(slot) / 3][i]), tn8_tr_read<so + 1024>(ya[(slot) / 3][i])); \
    }                                                                                                   \
    _Pragma("unroll") for (int j = 4; j < YF; ++j)                                                      \
      Yr[set][j] = cat4(tn8_tr_read<so>(ya[(slot) / 3][j]), tn8_tr_read<so + 512>(ya[(slot) / 3][j])); \
  }
#define TN8_MFMA(set, i, j)                                                                             \
  acc[i][j] = SWAP ? mfma16(Yr[set][j], Xr[set][i], acc[i][j]) : mfma16(Xr[set][i], Yr[set][j], acc[i][j]);

  // ---- prologue: fill the ring in steady-state order, wait for slots 0..2, load the fragments of slot 0
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) issue(s, s);
  TN8_WAIT_SLOTS(NSLOT - 3)
  TN8_BARRIER()
  TN8_LOAD(0, 0)
  TN8_CS_READ(0)
  tn8_wait<63>();  // lgkmcnt(0): every wave's reads of slot 0 retired before it is refilled
  TN8_BARRIER()
  TN8_CS_ADD()
  if (!FINE && late) TN8_BARRIER()

  // one phase; K is a literal so that ring slot, register set and instruction offsets are static
#define TN8_PHASE(K)                                                                                    \
  if constexpr ((K) < Cfg::UNROLL) {                                                                    \
    const int ph = g0 + (K);                                                                            \
    if (ph < S) {                                                                                       \
      const bool do_reads = ph + 1 < S && !MDT_EXP(p.dbg & 2), do_dma = ph + NSLOT < S && !MDT_EXP(p.dbg & 1);                                      \
      {                                                                                                 \
        /* half A (1) fragments of the next phase */                                                    \
        if (do_reads) {                                                                                 \
          TN8_LOAD(((K) + 1) & 1, ((K) + 1) % NSLOT)                                                    \
          TN8_CS_READ(((K) + 1) % NSLOT)                                                                \
        }                                                                                               \
        /* (2) slot K was read during the previous phase: refill it with the rows of phase ph+NSLOT */  \
        if (do_dma) issue((K) % NSLOT, ph + NSLOT);                                                     \
        tn8_wait<63>(); /* lgkmcnt(0): the other group refills the slot just read in ITS next half A */ \
        TN8_BARRIER()                                                                                   \
        /* half B (3) this phase's MFMAs */                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
        _Pragma("unroll") for (int j = 0; j < YF; ++j) TN8_MFMA((K) & 1, i, j)                          \
        __builtin_amdgcn_s_setprio(0);                                                                  \
        if (do_reads) TN8_CS_ADD() /* the X chunk of slot ph + 1 read in half A (landed: lgkmcnt(0) + barrier) */ \
        /* (4) publish this wave's share of the slot read two (early group) / three (late group) halves on */ \
        if (ph + NSLOT + 1 <= S) TN8_WAIT_PHASE()                                                       \
        else tn8_wait<0>();                                                                             \
        TN8_BARRIER()                                                                                   \
      }                                                                                                 \
    }                                                                                                   \
  }
  // FINE phase: MFMA q is followed by memory operation q: fragments X0..X3, Y0..Y(YF-1) of slot K+1 (two transpose
  // reads each) behind MFMAs 0 .. 3+YF, the 4 (YF = 6) / 3 LDS-DMA pieces of the refill of slot K behind MFMAs
  // YF+5, +DSTR, ...
  constexpr int DSTR = YF == 6 ? 3 : 2, NPART = YF == 6 ? 4 : 3;
#define TN8_FRAG(set, slot, f)                                                                          \
  {                                                                                                     \
    constexpr int so = ((slot) % 3) * SLOT_BYTES;                                                       \
    if ((f) < 4) Xr[set][(f)] = cat4(tn8_tr_read<so>(xa[(slot) / 3][(f)]), tn8_tr_read<so + 1024>(xa[(slot) / 3][(f)])); \
    else if ((f) < 8) Yr[set][(f) - 4] = cat4(tn8_tr_read<so>(ya[(slot) / 3][(f) - 4]), tn8_tr_read<so + 1024>(ya[(slot) / 3][(f) - 4])); \
    else Yr[set][(f) - 4] = cat4(tn8_tr_read<so>(ya[(slot) / 3][(f) - 4]), tn8_tr_read<so + 512>(ya[(slot) / 3][(f) - 4])); \
  }
  // STEADY: every condition is known (the main loop runs while whole unrolled groups have reads AND refills): no
  // branch inside the phase -- a run-time `if` around a read puts an s_cbranch behind every MFMA
#define TN8_PHASE_FINE(K, STEADY)                                                                       \
  if constexpr ((K) < Cfg::UNROLL) {                                                                    \
    const int ph = g0 + (K);                                                                            \
    if (STEADY || ph < S) {                                                                             \
      const bool do_reads = STEADY || (ph + 1 < S && !MDT_EXP(p.dbg & 2)), do_dma = STEADY || (ph + NSLOT < S && !MDT_EXP(p.dbg & 1)); \
      _Pragma("unroll") for (int q = 0; q < 4 * YF; ++q) {                                              \
        TN8_MFMA((K) & 1, q / YF, q % YF)                                                               \
        if (q < 4 + YF) { if (do_reads) TN8_FRAG(((K) + 1) & 1, ((K) + 1) % NSLOT, q) }                 \
        else if (q == 4 + YF) { if (do_reads) TN8_CS_READ(((K) + 1) % NSLOT) }                          \
        else if (q >= YF + 5 && (q - YF - 5) % DSTR == 0 && (q - YF - 5) / DSTR < NPART) {              \
          if (do_dma) issue((K) % NSLOT, ph + NSLOT, 1 << ((q - YF - 5) / DSTR));                       \
        }                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                              \
      }                                                                                                 \
      if (STEADY || ph + NSLOT + 1 <= S) TN8_WAIT_PHASE()                                               \
      else tn8_wait<0>();                                                                               \
      TN8_BARRIER()                                                                                     \
      if (do_reads) TN8_CS_ADD()                                                                        \
    }                                                                                                   \
  }
  if constexpr (FINE) {
    int g0 = 0;
    if (!MDT_EXP(p.dbg & 3)) {
      for (; g0 + Cfg::UNROLL + NSLOT <= S; g0 += Cfg::UNROLL) {
        TN8_PHASE_FINE(0, true) TN8_PHASE_FINE(1, true) TN8_PHASE_FINE(2, true) TN8_PHASE_FINE(3, true) TN8_PHASE_FINE(4, true)
        TN8_PHASE_FINE(5, true) TN8_PHASE_FINE(6, true) TN8_PHASE_FINE(7, true) TN8_PHASE_FINE(8, true) TN8_PHASE_FINE(9, true)
      }
    }
    for (; g0 < S; g0 += Cfg::UNROLL) {
      TN8_PHASE_FINE(0, false) TN8_PHASE_FINE(1, false) TN8_PHASE_FINE(2, false) TN8_PHASE_FINE(3, false) TN8_PHASE_FINE(4, false)
      TN8_PHASE_FINE(5, false) TN8_PHASE_FINE(6, false) TN8_PHASE_FINE(7, false) TN8_PHASE_FINE(8, false) TN8_PHASE_FINE(9, false)
    }
  } else {
    for (int g0 = 0; g0 < S; g0 += Cfg::UNROLL) {
      TN8_PHASE(0) TN8_PHASE(1) TN8_PHASE(2) TN8_PHASE(3) TN8_PHASE(4) TN8_PHASE(5)
      TN8_PHASE(6) TN8_PHASE(7) TN8_PHASE(8) TN8_PHASE(9)
    }
    if (!late) TN8_BARRIER()
  }
#undef TN8_PHASE
#undef TN8_PHASE_FINE
#undef TN8_FRAG
#undef TN8_LOAD
#undef TN8_MFMA
#undef TN8_WAIT_SLOTS
#undef TN8_WAIT_PHASE
#undef TN8_BARRIER

  if (do_cs) {  // the ring is dead: 32 row partials per chunk column -> LDS -> one atomic per column
    float* red = (float*)smem;  // [32 rows][16 chunk slots][8]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(cs_row * 16 + cs_c) * 8 + e] = cs_acc[e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tid < cs_n * 8) {
      float sum = 0.f;
      for (int r = 0; r < 32; ++r) sum += red[(r * 16 + (tid >> 3)) * 8 + (tid & 7)];
      const int col = x0 + (cs_begin + (tid >> 3)) * 8 + (tid & 7);
      if (col < p.NX) atomic_add_f32(p.colsum_x + col, sum);
    }
  }
#undef TN8_CS_READ
#undef TN8_CS_ADD

  // ---- epilogue: fp32 atomics.  MFMA C layout: col = lane&15 (second operand's index), row =
  // 4*(lane>>4) + r (first operand's index); either way the 16 lanes of a group hit 64 contiguous bytes.
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < YF; ++j) {
      const int yb = y0 + (j < 4 ? wy * 64 + j * 16 : 128 + wy * 32 + (j - 4) * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!SWAP) {
          const int x = x0 + wx * 64 + i * 16 + g * 4 + r;
          const int y = yb + i16;
          if (x < p.NX) atomic_add_f32(p.C + (long)x * p.ldc + y, acc[i][j][r]);
        } else {
          const int x = x0 + wx * 64 + i * 16 + i16;
          const int y = yb + g * 4 + r;
          if (x < p.NX) atomic_add_f32(p.C + (long)y * p.ldc + x, acc[i][j][r]);
        }
      }
    }
}

// colsum_a (optional): column sums of A.  Folded into the kernel when A takes the X role; returns 1 in *colsum_done then.
int launch_gemm_tn8(const bf16* A, int lda, const bf16* B, int ldb, int M, int N1, int N2, float* C, int ldc,
                    hipStream_t stream, float* colsum_a, int* colsum_done, int forced_splits) {
  TN8Params p;
  // Role assignment.  The 192-wide Y tile (YF = 6) wins whenever one width divides by 192: among the legal
  // assignments take the one that wastes the fewest MFMAs on the ragged 256-wide X tile; otherwise the 256 x 128 shape
  // with the 256-wide role on a width that divides by 256 if there is one.
  auto x_waste = [](int nx) { return (double)(((nx + 255) / 256) * 256) / nx; };
  const int knob = mdt_get_tuning_int(MDT_TUNE_TN8_WIDE);  // 1: never use the 256 x 192 shape (A/B runs)
  int yf = 4;
  bool a_is_x;
  const bool b192 = knob != 1 && N2 % 192 == 0, a192 = knob != 1 && N1 % 192 == 0;
  if (b192 || a192) {
    yf = 6;
    a_is_x = b192 && (!a192 || x_waste(N1) <= x_waste(N2));
  } else {
    a_is_x = (N1 % 256 == 0) ? true : (N2 % 256 == 0) ? false : (N1 >= N2);
  }
  if (a_is_x) { p.X = A; p.ldx = lda; p.NX = N1; p.Y = B; p.ldy = ldb; p.NY = N2; p.swap = 0; }
  else { p.X = B; p.ldx = ldb; p.NX = N2; p.Y = A; p.ldy = lda; p.NY = N1; p.swap = 1; }
  p.C = C; p.ldc = ldc;
  p.tiles_x = (p.NX + 255) / 256;
  p.tiles_y = p.NY / (32 * yf);
  // the fused column sums share an X tile's 256 columns out between the tiles_y workgroups of that tile, 16 chunk
  // lanes each: with a single Y tile half of the columns would have no owner -> separate pass then
  p.colsum_x = (colsum_a && a_is_x && p.tiles_y >= 2) ? colsum_a : nullptr;
  if (colsum_done) *colsum_done = p.colsum_x != nullptr;
  p.dbg = mdt_get_tuning_int(MDT_TUNE_TN8_DBG);
#ifndef MDT_EXPERIMENTS
  p.dbg &= ~7;  // bits 0-2 (skip refills / reads, re-read the first slots) produce garbage: experiments build only
#endif
  // tile order: groups of 2 X tiles, Y fastest -- the 32 concurrent workgroups of an XCD then cover ~5 X tiles x all 6 Y
  // tiles = 2517 distinct operand columns per streamed row instead of 2816 with groups of 8 (fc1 / fc2 weight gradients
  // -3..-7 %, gpurun_out/r3/tn8_group.log); tn8_dbg bits 4-7 override it for A/B runs
  p.group_x = ((p.dbg >> 4) & 15) ? ((p.dbg >> 4) & 15) : 2;
  const int nslot = yf == 6 ? TN8Cfg<6>::NSLOT : TN8Cfg<4>::NSLOT;
  p.slots_total = M / 32;
  const int tiles = p.tiles_x * p.tiles_y;
  const int cus = nt8_num_cus();  // the CU count the data-parallel wrapper leaves to compute ("nt8_max_cus"): the grid is not
                                  // persistent, so a concurrent RCCL kernel gets CUs as workgroups retire; the split only
                                  // has to be sized for the CUs that are really available
  // split the contraction: minimise (waves of `cus` workgroups) x (slots per split + fixed per-block cost
  // ~ prologue latency + 32K-48K epilogue atomics, worth about 48 slots of MFMA work)
  int best = 1;
  double best_cost = 1e30;
  // per-block fixed cost in slots of MFMA work: ring fill, dispatch, 32K-48K fp32 atomics per wave pair.  Fitted on
  // MI355X (tools/tn8_fixed_cost.py, profiles/r6_tn8_fixed_cost.txt): 32-38 us = 63-80 slots (rounds 2-5 assumed 48)
  const double FIXED = 64.0;
  // mdt_gemm_tn_args.splits > 0: the caller's split count (tools/tn8_fixed_cost.py fits the per-block fixed cost with it)
  if (forced_splits > 0) best = forced_splits < p.slots_total / nslot ? forced_splits : (p.slots_total / nslot > 0 ? p.slots_total / nslot : 1);
  for (int sp = 1; sp <= 64 && forced_splits <= 0; ++sp) {
    if (sp > 1 && p.slots_total / sp < 32) break;
    const long blocks = (long)tiles * sp;
    const double cost = (double)((blocks + cus - 1) / cus) * ((double)((p.slots_total + sp - 1) / sp) + FIXED);
    if (cost < best_cost * 0.98) { best_cost = cost; best = sp; }
  }
  p.slots_per_split = (p.slots_total + best - 1) / best;
  if (p.slots_per_split < nslot) p.slots_per_split = nslot;
  int splits = (p.slots_total + p.slots_per_split - 1) / p.slots_per_split;
  // a trailing split shorter than the ring would under-fill the prologue: merge it into its neighbour
  if (splits > 1 && p.slots_total - (splits - 1) * p.slots_per_split < nslot) {
    p.slots_per_split = (p.slots_total + splits - 2) / (splits - 1);
    splits = (p.slots_total + p.slots_per_split - 1) / p.slots_per_split;
  }
  int grid_n = tiles * splits;
  // ---- long + tail partition (TN8Params): k = cus / tiles long blocks per tile of L slots; the R = cus - k tiles other CUs
  // work through the `tiles` tail blocks in n = ceil(tiles / R) rounds.  L balances the two kinds of CU:
  // L + FIXED = n (TAIL (S - k L) + FIXED).  TAIL = 1.35: a tail block runs 1.25-1.35x slower per slot than a long one --
  // only R / 8 workgroups of its XCD read its rows at the same time (27-32 for the long blocks), so its refills are L2
  // misses (measured, profiles/r6_tn8_long_tail_ab.txt: with TAIL = 1.1 the model took this form for the XL/2 fc1 / fc2
  // weight gradients at 131072 rows and they ran 6 % SLOWER than 7 uniform splits; proj, where the tails are a 4 % share,
  // ran 4.6 % faster, and fc1 / fc2 at 16384 rows 2-4 % faster).  Taken when the model predicts >= 3 % less than the best
  // uniform split: proj at batch 1024, fc1 / fc2 at per-GPU batch 128.  "tn8_dbg" bit 8 = uniform splits only (A/B runs).
  p.long_k = p.long_len = 0;
  if (forced_splits <= 0 && !((p.dbg >> 8) & 1) && tiles < cus) {
    const int k = cus / tiles, R = cus - k * tiles;
    if (R > 0) {
      const int n = (tiles + R - 1) / R;
      const double TAIL = 1.35, S = (double)p.slots_total;
      int L = (int)((n * TAIL * S + (n - 1) * FIXED) / (1.0 + n * TAIL * k));
      if ((long)L * k > p.slots_total - 16) L = (p.slots_total - 16) / k;
      const int tail = p.slots_total - k * L;
      const double cost = fmax((double)L + FIXED, n * (TAIL * tail + FIXED));
      if (L >= 32 && tail >= 16 && tail >= nslot && cost < 0.97 * best_cost) {
        p.long_k = k;
        p.long_len = L;
        grid_n = tiles * k + tiles;
      }
    }
  }
  const dim3 grid(grid_n), block(512);
  const bool fine = (p.dbg & 8) == 0;  // tn8_dbg bit 3: the round-2 staggered form (A/B runs)
  const bool cs = p.colsum_x != nullptr;  // (only with swap == 0: the column sums are those of the X operand = A)
#define TN8_LAUNCH2(SW, YFV, CSV) { if (fine) hipLaunchKernelGGL((gemm_tn8_kernel<SW, YFV, true, CSV>), grid, block, 0, stream, p); \
                                    else hipLaunchKernelGGL((gemm_tn8_kernel<SW, YFV, false, CSV>), grid, block, 0, stream, p); }
#define TN8_LAUNCH(SW, YFV) { if (cs && !SW) TN8_LAUNCH2(SW, YFV, (!SW)) else TN8_LAUNCH2(SW, YFV, false) }
  if (yf == 6) { if (p.swap) TN8_LAUNCH(true, 6) else TN8_LAUNCH(false, 6) }
  else { if (p.swap) TN8_LAUNCH(true, 4) else TN8_LAUNCH(false, 4) }
#undef TN8_LAUNCH2
#undef TN8_LAUNCH
  return mdt_check_launch("gemm_tn8");
}
