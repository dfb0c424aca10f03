// gemm_tn8: the large-problem weight-gradient GEMM  out[x, y] += sum_m X[m, x] * Y[m, y]
// (contraction over ROWS; X, Y row-major bf16; fp32 atomic accumulation, split over m).
//
// 256 (x) x 128 (y) output tile per 512-thread workgroup: 8 waves as 4(x) x 2(y), wave tile
// 64 x 64 = 4 x 4 MFMA 16x16x32 accumulators.  The contraction index is the slow (row) index of
// both operands, so tiles are kept in LDS exactly as they lie in memory ([m][n], 256-byte rows per
// 128-column half) and MFMA operands are fetched with the gfx950 LDS transpose read
// (ds_read_b64_tr_b16, two per fragment).  Pipeline: a ring of NSLOT slots of 32 contraction rows
// (one MFMA k-step); phase g = [tr-read the fragments of slot g+1 into the alternate register set]
// [LDS-DMA refill of the slot that was read during phase g-1 with rows of phase g+NSLOT]
// [16 MFMAs] [s_waitcnt vmcnt((NSLOT-2)*3) lgkmcnt(0)] [s_barrier]: every load has NSLOT-2 phases
// to land and vmcnt(0) only appears in the last ring pass.  Same swizzle as the 128x128 kernel
// (gemm.hip: tn_swz) through the DMA source address.
//
// Used by mdt_gemm_tn when both widths are multiples of 128, nothing is masked, and the row count
// per split is a multiple of 32 (>= NSLOT slots); ragged 256-wide x tiles clamp their source
// columns (never stored).  `swap` stores the tile transposed (C[y, x]) so that the caller can give
// the 256-wide role to whichever operand divides by 256.
#include "gemm_common.h"

#define TN8_NSLOT 6
#define TN8_SLOT_BYTES 24576  // X half0 8K | X half1 8K | Y 8K   (32 rows x 256 B each)

__device__ __forceinline__ int tn8_swz(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }

struct TN8Params {
  const bf16* X; int ldx;
  const bf16* Y; int ldy;
  int NX, NY;          // widths (NY % 128 == 0; NX % 128 == 0)
  float* C; int ldc;
  int swap;            // 0: C[x*ldc + y]   1: C[y*ldc + x]
  int slots_total;     // contraction rows / 32
  int slots_per_split;
  int tiles_x, tiles_y;
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
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

// SWAP = false: out tile rows = x (C[x, y]);  SWAP = true: the MFMA operands trade places so that the
// accumulator tile is [y rows][x cols] and the atomics to C[y, x] stay lane-contiguous.
template <bool SWAP>
__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(TN8Params p) {
  __shared__ __attribute__((aligned(16))) char smem[TN8_NSLOT * TN8_SLOT_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = wave >> 1, wy = wave & 1;

  const int tiles = p.tiles_x * p.tiles_y;
  const int sid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = sid / tiles;
  int tx, ty;
  tile_coords(sid - split * tiles, p.tiles_x, p.tiles_y, tx, ty);
  const int x0 = tx * 256, y0 = ty * 128;
  const int s_begin = split * p.slots_per_split;
  const int S = min(p.slots_per_split, p.slots_total - s_begin);  // >= TN8_NSLOT (host guarantees)
  if (S <= 0) return;

  // ---- LDS-DMA addressing: per slot a wave moves rows 4w..4w+3 of X half 0, X half 1 and Y.
  const int lr = lane >> 4, cpos = lane & 15;
  const int rloc = 4 * wave + lr;  // slot-local row
  const int gch = cpos ^ tn8_swz(rloc);
  const long row0 = (long)s_begin * 32 + rloc;
  const int xc0 = min(x0 + gch * 8, p.NX - 8), xc1 = min(x0 + 128 + gch * 8, p.NX - 8);  // clamp ragged tiles
  const bf16* x_src0 = p.X + row0 * p.ldx + xc0;
  const bf16* x_src1 = p.X + row0 * p.ldx + xc1;
  const bf16* y_src = p.Y + row0 * p.ldy + y0 + gch * 8;
  const long x_step = 32L * p.ldx, y_step = 32L * p.ldy;
  char* const lds_w = smem + wave * 1024;

  auto issue = [&](int slot, int s) {  // fill ring slot `slot` with contraction rows of phase s
    char* base = lds_w + slot * TN8_SLOT_BYTES;
    glds16(x_src0 + s * x_step, base);
    glds16(x_src1 + s * x_step, base + 8192);
    glds16(y_src + s * y_step, base + 16384);
  };

  // ---- transpose-read addressing (gemm.hip gemm_tn_kernel): the 16-lane group g reads slot rows
  // 8g + 4t + (0..3) x 16 columns; lane i16 supplies row (i16>>2), 8-byte piece (i16&3).
  const int i16 = lane & 15, g = lane >> 4;
  const int sw = tn8_swz(8 * g + (i16 >> 2));
  const int row_off = (8 * g + (i16 >> 2)) * 256 + ((i16 & 1) << 3);
  const int cx0 = (wx & 1) * 8 + ((i16 & 3) >> 1), cy0 = wy * 8 + ((i16 & 3) >> 1);  // + 2*frag
  // LDS byte addresses per fragment, one set per PAIR of ring slots so that the remaining
  // (slot & 1) * SLOT_BYTES + 1024 * t fits the 16-bit instruction offset
  const unsigned lds_base = (unsigned)(size_t)LDS_PTR(smem);
  unsigned xa[TN8_NSLOT / 2][4], ya[TN8_NSLOT / 2][4];
#pragma unroll
  for (int pr = 0; pr < TN8_NSLOT / 2; ++pr)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xa[pr][i] = lds_base + pr * 2 * TN8_SLOT_BYTES + (wx >> 1) * 8192 + row_off + (((cx0 + 2 * i) ^ sw) << 4);
      ya[pr][i] = lds_base + pr * 2 * TN8_SLOT_BYTES + 16384 + row_off + (((cy0 + 2 * i) ^ sw) << 4);
    }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 Xr[2][4], Yr[2][4];

#define TN8_LOAD(set, slot)                                                                             \
  {                                                                                                     \
    constexpr int so = ((slot) & 1) * TN8_SLOT_BYTES;                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
      Xr[set][i] = cat4(tn8_tr_read<so>(xa[(slot) >> 1][i]), tn8_tr_read<so + 1024>(xa[(slot) >> 1][i])); \
      Yr[set][i] = cat4(tn8_tr_read<so>(ya[(slot) >> 1][i]), tn8_tr_read<so + 1024>(ya[(slot) >> 1][i])); \
    }                                                                                                   \
  }

  // ---- prologue: fill the ring in steady-state order, wait for slot 0, load its fragments
#pragma unroll
  for (int s = 0; s < TN8_NSLOT; ++s) issue(s, s);
  tn8_wait<3 * (TN8_NSLOT - 1)>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  TN8_LOAD(0, 0)
  tn8_wait<3 * (TN8_NSLOT - 2)>();  // slot 1 landed as well (read during phase 0); LDS reads of slot 0 retired
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // one phase; K is a literal so that ring slot, register set and instruction offsets are static
#define TN8_PHASE(K)                                                                                    \
  {                                                                                                     \
    const int ph = g0 + (K);                                                                            \
    if (ph < S) {                                                                                       \
      /* (1) fragments of the next phase */                                                             \
      if (ph + 1 < S) TN8_LOAD(((K) + 1) & 1, ((K) + 1) % TN8_NSLOT)                                    \
      /* (2) slot K was read during the previous phase: refill it with the rows of phase ph+NSLOT */    \
      if (ph + TN8_NSLOT < S) issue((K), ph + TN8_NSLOT);                                               \
      /* (3) this phase's MFMAs */                                                                      \
      __builtin_amdgcn_s_setprio(1);                                                                    \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
        acc[i][j] = SWAP ? mfma16(Yr[(K) & 1][j], Xr[(K) & 1][i], acc[i][j])                            \
                         : mfma16(Xr[(K) & 1][i], Yr[(K) & 1][j], acc[i][j]);                           \
      __builtin_amdgcn_s_setprio(0);                                                                    \
      /* (4) publish the slot that the next phase reads (data of phase ph+2) */                         \
      if (ph + TN8_NSLOT + 1 <= S) tn8_wait<3 * (TN8_NSLOT - 2)>();                                     \
      else tn8_wait<0>();                                                                               \
      __builtin_amdgcn_s_barrier();                                                                     \
      asm volatile("" ::: "memory");                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                \
    }                                                                                                   \
  }
  static_assert(TN8_NSLOT == 6, "the ring pass below is written out for 6 slots");
  for (int g0 = 0; g0 < S; g0 += TN8_NSLOT) {
    TN8_PHASE(0) TN8_PHASE(1) TN8_PHASE(2) TN8_PHASE(3) TN8_PHASE(4) TN8_PHASE(5)
  }
#undef TN8_PHASE
#undef TN8_LOAD

  // ---- epilogue: fp32 atomics.  MFMA C layout: col = lane&15 (second operand's index), row =
  // 4*(lane>>4) + r (first operand's index); either way the 16 lanes of a group hit 64 contiguous bytes.
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!SWAP) {
          const int x = x0 + wx * 64 + i * 16 + g * 4 + r;
          const int y = y0 + wy * 64 + j * 16 + i16;
          if (x < p.NX) atomic_add_f32(p.C + (long)x * p.ldc + y, acc[i][j][r]);
        } else {
          const int x = x0 + wx * 64 + i * 16 + i16;
          const int y = y0 + wy * 64 + j * 16 + g * 4 + r;
          if (x < p.NX) atomic_add_f32(p.C + (long)y * p.ldc + x, acc[i][j][r]);
        }
      }
    }
}

int launch_gemm_tn8(const bf16* A, int lda, const bf16* B, int ldb, int M, int N1, int N2, float* C, int ldc,
                    hipStream_t stream) {
  TN8Params p;
  // give the 256-wide role to an operand whose width divides by 256 if there is one, else to the wider
  const bool a_is_x = (N1 % 256 == 0) ? true : (N2 % 256 == 0) ? false : (N1 >= N2);
  if (a_is_x) { p.X = A; p.ldx = lda; p.NX = N1; p.Y = B; p.ldy = ldb; p.NY = N2; p.swap = 0; }
  else { p.X = B; p.ldx = ldb; p.NX = N2; p.Y = A; p.ldy = lda; p.NY = N1; p.swap = 1; }
  p.C = C; p.ldc = ldc;
  p.tiles_x = (p.NX + 255) / 256;
  p.tiles_y = p.NY / 128;
  p.slots_total = M / 32;
  const int tiles = p.tiles_x * p.tiles_y;
  // split the contraction: minimise (waves of 256 workgroups) x (slots per split + fixed per-block cost
  // ~ prologue latency + 32K epilogue atomics, worth about 48 slots of MFMA work)
  int best = 1;
  double best_cost = 1e30;
  for (int sp = 1; sp <= 64; ++sp) {
    if (sp > 1 && p.slots_total / sp < 32) break;
    const long blocks = (long)tiles * sp;
    const double cost = (double)((blocks + 255) / 256) * ((double)((p.slots_total + sp - 1) / sp) + 48.0);
    if (cost < best_cost * 0.98) { best_cost = cost; best = sp; }
  }
  p.slots_per_split = (p.slots_total + best - 1) / best;
  if (p.slots_per_split < TN8_NSLOT) p.slots_per_split = TN8_NSLOT;
  int splits = (p.slots_total + p.slots_per_split - 1) / p.slots_per_split;
  // a trailing split shorter than the ring would under-fill the prologue: merge it into its neighbour
  if (splits > 1 && p.slots_total - (splits - 1) * p.slots_per_split < TN8_NSLOT) {
    p.slots_per_split = (p.slots_total + splits - 2) / (splits - 1);
    splits = (p.slots_total + p.slots_per_split - 1) / p.slots_per_split;
  }
  if (p.swap) hipLaunchKernelGGL(gemm_tn8_kernel<true>, dim3(tiles * splits), dim3(512), 0, stream, p);
  else hipLaunchKernelGGL(gemm_tn8_kernel<false>, dim3(tiles * splits), dim3(512), 0, stream, p);
  return mdt_check_launch("gemm_tn8");
}
