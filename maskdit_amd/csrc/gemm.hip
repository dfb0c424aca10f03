// bf16 MFMA GEMMs for gfx950.
//
//   gemm_nt : C[M,N]  = A[M,K] * B[N,K]^T  (+bias, fused epilogues)       forward + dgrad
//   gemm_tn : C[N1,N2] += A[M,N1]^T * B[M,N2]  (f32 atomics, split over M)  wgrad
//
// Both: 128x128 output tile per 256-thread workgroup (4 waves, each 64x64 = 4x4 MFMA
// 16x16x32 fragments), K-step 64, operands staged HBM -> LDS with 16-byte
// global_load_lds (LDS-DMA, no VGPR round trip) into a 2-deep ring so the next K-tile is in
// flight while the current one feeds the matrix cores, XOR-swizzled through the *source*
// address (the LDS-DMA destination is lane-linear) so fragment reads are bank-conflict free,
// and an XCD-aware block-id remap + grouped tile order so the 32 CUs of one XCD share A/B
// panels in their private L2.
//
// Reference: every nn.Linear of models/maskdit.py (see include/maskdit_hip.h).
#include "common.h"
#include "../../include/maskdit_hip.h"

#include "gemm_common.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES 32768  // A 16 KiB + B 16 KiB

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NTParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- LDS-DMA source addressing: wave-instruction j covers tile rows 8j..8j+7 (128 B each);
  // lane -> (row 8j + lane/8, LDS chunk lane%8), global chunk = lds chunk ^ (row & 7).
  const int lr = lane >> 3, cpos = lane & 7;
  const int gch = cpos ^ lr;
  const bf16* a_src[4];
  const bf16* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = 8 * (wave * 4 + i) + lr;
    int am = min(m0 + r, p.M - 1);
    a_src[i] = p.A + (long)am * p.lda + gch * 8;
    b_src[i] = p.B + (long)(n0 + r) * p.ldb + gch * 8;
  }
  char* const lds_wave = smem + wave * 4096;  // + stage*STAGE_BYTES (+16384 for B) + i*1024

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // split-K (EPI_F32 accumulate): blockIdx.y selects a contiguous range of K-tiles
  const int nk_total = p.K / BK;
  const int nk_per = (nk_total + gridDim.y - 1) / gridDim.y;
  const int kt0 = blockIdx.y * nk_per;
  const int nk = min(nk_per, nk_total - kt0);
  if (nk <= 0) return;
  if (blockIdx.y > 0) p.bias = nullptr;
  // fragment read offsets (bytes) inside a stage: row r -> r*128 + ((kc ^ (r&7)) * 16)
  const int fr = lane & 15, fg = lane >> 4;
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    int kc = ks * 4 + fg;
    a_off[ks] = (wr * 64 + fr) * 128 + ((kc ^ (fr & 7)) << 4);
    b_off[ks] = 16384 + (wc * 64 + fr) * 128 + ((kc ^ (fr & 7)) << 4);
  }

#define ISSUE_STAGE(st, kt)                                                      \
  {                                                                              \
    char* base = lds_wave + (st) * STAGE_BYTES;                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                              \
      glds16(a_src[i] + (long)(kt0 + (kt)) * BK, base + i * 1024);               \
      glds16(b_src[i] + (long)(kt0 + (kt)) * BK, base + 16384 + i * 1024);       \
    }                                                                            \
  }

  ISSUE_STAGE(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) ISSUE_STAGE((kt + 1) & 1, kt + 1);
    const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(st + a_off[ks] + i * 16 * 128);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(st + b_off[ks] + j * 16 * 128);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
  }
#undef ISSUE_STAGE

  // ---- epilogue: restage each 16-row fragment band through LDS so every lane owns 16
  // consecutive columns of one row (coalesced, vectorised fused epilogue).
  float* stg = (float*)(smem + wave * (16 * 68 * 4));
  const int er = lane >> 2, ec = (lane & 3) * 16;
  const int n = n0 + wc * 64 + ec;
  float csum[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) csum[q] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) stg[(fg * 4 + r) * 68 + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    const int m = m0 + wr * 64 + i * 16 + er;
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 t = *(const f32x4*)(stg + er * 68 + ec + q * 4);
      v[q * 4 + 0] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
    }
    nt_epilogue_row<16>(p, m, n, v, csum);
  }
  if (p.colsum) nt_colsum_flush<16>(p, n, csum, lane);
}

// ------------------------------------------------------------------------------------------
// TN: contraction over rows.  Tiles are stored as loaded ([m][n], n contiguous, 256 B rows);
// MFMA operands (contraction index contiguous per lane) come from the gfx950 LDS
// transpose-read ds_read_b64_tr_b16.

struct TNParams {
  const bf16* A; int lda;
  const bf16* B; int ldb;
  int M, N1, N2;
  float* C; int ldc;
  int n1_valid, n2_valid;
  int splits, ksteps_per_split;
};

__device__ __forceinline__ int tn_swz(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TNParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int tiles_1 = (p.N1 + BM - 1) / BM, tiles_2 = (p.N2 + BN - 1) / BN;
  const int tiles = tiles_1 * tiles_2;
  const int s = xcd_remap(blockIdx.x, gridDim.x);
  const int split = s / tiles;
  int t1, t2;
  tile_coords(s - split * tiles, tiles_1, tiles_2, t1, t2);
  const int n1_0 = t1 * BM, n2_0 = t2 * BN;
  const int total_ksteps = p.M / BK;
  const int k_begin = split * p.ksteps_per_split;
  const int nk = min(p.ksteps_per_split, total_ksteps - k_begin);
  if (nk <= 0) return;

  // LDS-DMA addressing: wave-instruction j covers tile rows 4j..4j+3 (256 B each).
  const int lr = lane >> 4, cpos = lane & 15;
  const bf16* a_src[4];
  const bf16* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = 4 * (wave * 4 + i) + lr;
    int gch = cpos ^ tn_swz(r);
    long row = (long)k_begin * BK + r;
    a_src[i] = p.A + row * p.lda + n1_0 + gch * 8;
    b_src[i] = p.B + row * p.ldb + n2_0 + gch * 8;
  }
  char* const lds_wave = smem + wave * 4096;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transpose-read addressing.  For sub-step ks and half t the 16-lane group g reads block rows
  // 32ks + 8g + 4t + (0..3) x 16 columns; lane i16 supplies row (i16>>2), 8-byte piece (i16&3).
  // tn_swz of those rows depends only on (i16>>2) and (g&1), i.e. it is a per-lane constant.
  const int i16 = lane & 15, g = lane >> 4;
  const int sw = tn_swz(8 * g + (i16 >> 2));
  const int row_off = (8 * g + (i16 >> 2)) * 256 + ((i16 & 1) << 3);  // + 32ks*256 + 4t*256
  const int ca0 = wr * 8 + ((i16 & 3) >> 1), cb0 = wc * 8 + ((i16 & 3) >> 1);  // + 2*frag

#define ISSUE_STAGE(st, kt)                                                      \
  {                                                                              \
    char* base = lds_wave + (st) * STAGE_BYTES;                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                              \
      glds16(a_src[i] + (long)(kt) * BK * p.lda, base + i * 1024);               \
      glds16(b_src[i] + (long)(kt) * BK * p.ldb, base + 16384 + i * 1024);       \
    }                                                                            \
  }

  ISSUE_STAGE(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) ISSUE_STAGE((kt + 1) & 1, kt + 1);
    const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* q = st + ks * 32 * 256 + row_off + (((ca0 + 2 * i) ^ sw) << 4);
        af[i] = cat4(lds_tr_read(q), lds_tr_read(q + 4 * 256));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const char* q = st + 16384 + ks * 32 * 256 + row_off + (((cb0 + 2 * j) ^ sw) << 4);
        bfr[j] = cat4(lds_tr_read(q), lds_tr_read(q + 4 * 256));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
  }
#undef ISSUE_STAGE

  // C layout: col = lane&15 (n2), row = 4*(lane>>4) + r (n1)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n2 = n2_0 + wc * 64 + j * 16 + i16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int n1 = n1_0 + wr * 64 + i * 16 + g * 4 + r;
        if (n1 < p.n1_valid && n2 < p.n2_valid) atomic_add_f32(p.C + (long)n1 * p.ldc + n2, acc[i][j][r]);
      }
    }
}

// ------------------------------------------------------------------------------------------

extern "C" int mdt_gemm_nt(const mdt_gemm_nt_args* a, mdt_stream_t stream) {
  MDT_REQUIRE(a && a->A && a->B, "gemm_nt: null operand");
  MDT_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "gemm_nt: empty problem");
  MDT_REQUIRE(a->N % BN == 0, "gemm_nt: N must be a multiple of 128");
  MDT_REQUIRE(a->K % BK == 0, "gemm_nt: K must be a multiple of 64");
  MDT_REQUIRE(a->lda % 8 == 0 && a->ldb % 8 == 0, "gemm_nt: lda/ldb must be multiples of 8");
  MDT_REQUIRE(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "gemm_nt: operands must be 16-byte aligned");
  MDT_REQUIRE(!(a->colsum && a->k_splits > 1), "gemm_nt: colsum cannot be combined with k_splits");
  MDT_REQUIRE(!(a->colsum && !a->out), "gemm_nt: colsum sums the bf16 output, which needs `out`");
  switch (a->epi) {
    case MDT_EPI_BF16: MDT_REQUIRE(a->out, "gemm_nt: EPI_BF16 needs out"); break;
    case MDT_EPI_F32: MDT_REQUIRE(a->outf, "gemm_nt: EPI_F32 needs outf"); break;
    case MDT_EPI_GELU:
    case MDT_EPI_SILU: MDT_REQUIRE(a->out2, "gemm_nt: activation epilogue needs out2 (out = pre-activation is optional)"); break;
    case MDT_EPI_GATE_RES:
      MDT_REQUIRE(a->outf && a->res && a->gate && a->rows_per_sample > 0, "gemm_nt: EPI_GATE_RES needs outf/res/gate (out is optional)");
      break;
    case MDT_EPI_DGELU:
    case MDT_EPI_DSILU: MDT_REQUIRE(a->out && a->aux, "gemm_nt: d-activation epilogue needs out and aux"); break;
    default: MDT_REQUIRE(false, "gemm_nt: unknown epilogue");
  }
  NTParams p;
  p.A = (const bf16*)a->A; p.lda = a->lda; p.B = (const bf16*)a->B; p.ldb = a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K; p.bias = a->bias; p.epi = a->epi;
  p.out = (bf16*)a->out; p.ldo = a->ldo; p.out2 = (bf16*)a->out2; p.ldo2 = a->ldo2;
  p.outf = a->outf; p.ldof = a->ldof; p.res = a->res; p.ldres = a->ldres;
  p.gate = a->gate; p.gate_ld = a->gate_ld; p.rows_per_sample = a->rows_per_sample;
  p.aux = (const bf16*)a->aux; p.ldaux = a->ldaux;
  p.k_splits = 1;
  p.group_m = mdt_get_tuning_int(MDT_TUNE_NT8_GROUP_M);
  p.colsum = a->colsum;
  // large aligned problems: phase-pipelined persistent kernels (gemm_nt8_impl.h).  variant 0 = auto:
  // 256-row tiles, one 8-wave workgroup per CU (measured best inside the training step); M % 256 != 0
  // falls to 128-row tiles with two 4-wave workgroups per CU; 2 / 3 force either form.
  const int variant = mdt_get_tuning_int(MDT_TUNE_GEMM_NT_VARIANT);
  // the nt8 epilogues work on whole 16-byte pieces straight from the accumulators
  bool nt8_ok = variant != 1 && a->k_splits <= 1 && a->M % 128 == 0 && a->K % 128 == 0;
  if (a->out) nt8_ok = nt8_ok && a->ldo % 8 == 0 && ((uintptr_t)a->out & 15) == 0;
  if (a->out2) nt8_ok = nt8_ok && a->ldo2 % 8 == 0 && ((uintptr_t)a->out2 & 15) == 0;
  if (a->outf) nt8_ok = nt8_ok && a->ldof % 4 == 0 && ((uintptr_t)a->outf & 15) == 0;
  if (a->bias) nt8_ok = nt8_ok && ((uintptr_t)a->bias & 15) == 0;
  if (a->colsum) nt8_ok = nt8_ok && (a->epi == MDT_EPI_BF16 || a->epi == MDT_EPI_DGELU || a->epi == MDT_EPI_DSILU);
  if (a->epi == MDT_EPI_F32) nt8_ok = nt8_ok && a->out == nullptr;
  if (a->epi == MDT_EPI_GATE_RES)
    nt8_ok = nt8_ok && a->rows_per_sample % 64 == 0 && a->ldres % 4 == 0 && a->gate_ld % 4 == 0 &&
             (((uintptr_t)a->res | (uintptr_t)a->gate) & 15) == 0;
  if (a->epi == MDT_EPI_DGELU || a->epi == MDT_EPI_DSILU) nt8_ok = nt8_ok && a->ldaux % 8 == 0 && ((uintptr_t)a->aux & 15) == 0;  // 16-byte pair loads
  if (nt8_ok) {
    if (MDT_EXP(mdt_get_tuning_int(MDT_TUNE_NT8_SKIP_EPILOGUE))) p.epi |= 0x100;  // garbage results: experiments build only
    if (const int stg = mdt_get_tuning_int(MDT_TUNE_NT8_STAGGER)) p.epi |= 0x200 | ((stg < 255 ? stg : 255) << 16);
    // "nt8_overlap": 1 = the wave-specialised form (gemm_nt8o.hip: fused epilogue under the next tile's K loop) for the
    // epilogue classes that carry HBM traffic beyond one bf16 output (GELU / SiLU pairs, gate * y + residual); 2 = also
    // for the plain bf16 epilogue; 4 = three loader waves + one epilogue wave instead of 2 + 2; bits 4.. = its
    // timing-decomposition switches (experiments build)
    if (const int ov = mdt_get_tuning_int(MDT_TUNE_NT8_OVERLAP)) {
      if (nt8o_eligible(p) && ((ov & 2) || a->epi != MDT_EPI_BF16))
        return launch_gemm_nt8o(p, (ov & 4) ? 3 : 2, MDT_EXP(ov >> 4), (hipStream_t)stream);
    }
    const bool can8 = (a->M % 256 == 0);
    const int cus = nt8_num_cus();
    // column tile: 256 (NF = 4) where the epilogue class has it, else 192, else 128; "nt8_nf3" prefers 192-column
    // tiles whenever N allows (no register spill, deeper epilogue look-ahead; A/B knob)
    const int max_nf = nt8_max_nf(a->epi);
    const bool prefer3 = mdt_get_tuning_int(MDT_TUNE_NT8_NF3) != 0;
    int nf8 = (a->N % 256 == 0 && max_nf >= 4) ? 4 : (a->N % 192 == 0) ? 3 : 2;
    if (prefer3 && a->N % 192 == 0) nf8 = 3;
    const int nf4 = (a->N % 192 == 0) ? 3 : 2;
    const long tiles8 = can8 ? (long)(a->M / 256) * (a->N / (64 * nf8)) : 0;
    const long tiles4 = (long)(a->M / 128) * (a->N / (64 * nf4));
    if (variant == 2 && can8) return launch_gemm_nt8(p, nf8, 2, (hipStream_t)stream);
    if (variant == 3) return launch_gemm_nt8(p, nf4, 1, (hipStream_t)stream);
    if (variant == 0) {
      // Cost model (tools/gemm_bench.py --small-m / default): time ~ output area walked by the busiest CU.
      // 8-wave: whole rounds of 256 x 64*nf8 tiles.  4-wave: half-height tiles, two per CU at a time; a
      // workgroup that is alone on its CU (odd tail) runs ~1.6x faster.  With equal areas the 8-wave form
      // wins inside the training step (next-tile prefetch under the fused epilogues, more operand reuse), so
      // the 4-wave form is taken only when whole-tile rounds quantise badly -- per-GPU batch 128 of the 8-GPU
      // configuration.
      double w8 = 1e30, w4 = 1e30;
      if (can8 && tiles8 >= 192) w8 = (double)((tiles8 + cus - 1) / cus) * (64.0 * nf8);
      if (tiles4 >= 384) {
        const long q = (tiles4 + cus - 1) / cus;
        w4 = ((double)(q / 2) + 0.62 * (double)(q & 1)) * (64.0 * nf4) * (a->K < 2048 ? 1.0 : 1.10);
      }
      if (w4 < 0.97 * w8 && w4 < 1e29) return launch_gemm_nt8(p, nf4, 1, (hipStream_t)stream);
      if (w8 < 1e29) return launch_gemm_nt8(p, nf8, 2, (hipStream_t)stream);
    }
    p.epi &= 0xff;
  }
  p.epi &= 0xff;
  int tiles = cdiv(a->M, BM) * (a->N / BN);
  int ks = 1;
  if (a->k_splits > 1) {
    MDT_REQUIRE(a->epi == MDT_EPI_F32 && a->out == nullptr, "gemm_nt: k_splits needs MDT_EPI_F32 without a bf16 output");
    MDT_REQUIRE(a->ldof == a->N, "gemm_nt: k_splits needs a dense outf (ldof == N)");
    ks = a->k_splits < a->K / BK ? a->k_splits : a->K / BK;
    if (hipMemsetAsync(a->outf, 0, sizeof(float) * (size_t)a->M * a->N, (hipStream_t)stream) != hipSuccess) {
      mdt_set_error("gemm_nt: clearing the split-K accumulator failed");
      return MDT_ERR_LAUNCH;
    }
  }
  p.k_splits = ks;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(tiles, ks), dim3(256), 0, (hipStream_t)stream, p);
  return mdt_check_launch("gemm_nt");
}

extern "C" int mdt_gemm_tn(const mdt_gemm_tn_args* a, mdt_stream_t stream) {
  MDT_REQUIRE(a && a->A && a->B && a->C, "gemm_tn: null operand");
  MDT_REQUIRE(a->M > 0 && a->M % BK == 0, "gemm_tn: M must be a positive multiple of 64");
  MDT_REQUIRE(a->N1 > 0 && a->N2 > 0, "gemm_tn: empty output");
  MDT_REQUIRE(a->lda % 8 == 0 && a->ldb % 8 == 0, "gemm_tn: lda/ldb must be multiples of 8");
  MDT_REQUIRE(a->lda >= cdiv(a->N1, BM) * BM && a->ldb >= cdiv(a->N2, BN) * BN,
              "gemm_tn: operand rows must be padded to a multiple of 128 columns");
  MDT_REQUIRE(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "gemm_tn: operands must be 16-byte aligned");
  TNParams p;
  p.A = (const bf16*)a->A; p.lda = a->lda; p.B = (const bf16*)a->B; p.ldb = a->ldb;
  p.M = a->M; p.N1 = a->N1; p.N2 = a->N2; p.C = a->C; p.ldc = a->ldc;
  p.n1_valid = a->n1_valid > 0 ? a->n1_valid : a->N1;
  p.n2_valid = a->n2_valid > 0 ? a->n2_valid : a->N2;
  const int tn_variant = mdt_get_tuning_int(MDT_TUNE_GEMM_TN_VARIANT);
  if (tn_variant != 1 && p.n1_valid == a->N1 && p.n2_valid == a->N2 && a->N1 % 128 == 0 && a->N2 % 128 == 0 &&
      a->M % 32 == 0 && a->M >= 8192 && (long)a->N1 * a->N2 >= 256L * 1024) {
    int cs_done = 0;
    int rc = launch_gemm_tn8(p.A, p.lda, p.B, p.ldb, p.M, p.N1, p.N2, p.C, p.ldc, (hipStream_t)stream, a->colsum_a, &cs_done, a->splits);
    if (rc == MDT_OK && a->colsum_a && !cs_done) rc = mdt_colsum_bf16(a->A, a->lda, a->colsum_a, a->M, a->N1, stream);
    return rc;
  }
  if (a->colsum_a) {  // kernels without the fused column sums: separate pass
    int rc = mdt_colsum_bf16(a->A, a->lda, a->colsum_a, a->M, a->N1, stream);
    if (rc != MDT_OK) return rc;
  }
  int tiles = cdiv(a->N1, BM) * cdiv(a->N2, BN);
  int ksteps = a->M / BK;
  int splits = a->splits;
  if (splits <= 0) {
    splits = 2048 / tiles;
    if (splits > ksteps / 8) splits = ksteps / 8;
    if (splits < 1) splits = 1;
  }
  if (splits > ksteps) splits = ksteps;
  p.ksteps_per_split = cdiv(ksteps, splits);
  splits = cdiv(ksteps, p.ksteps_per_split);
  p.splits = splits;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * splits), dim3(256), 0, (hipStream_t)stream, p);
  return mdt_check_launch("gemm_tn");
}
