// The fp32-FAITHFUL inference path (VERDICT r5 missing #3 / next #4): the reference's sampler runs its network in fp32
// (sample.py:56 `net(x_hat.float(), ...)`, no autocast in generate.py) and `train.py --no_amp` does the same for training;
// rounds 1-5 only had the bf16-operand kernels, so BASELINE configs[4] was measured at narrower arithmetic than the
// reference's own.  Everything here computes in EXACT fp32: weights are read straight from the fp32 master arena (no
// shadow), activations stay fp32 in HBM, and the GEMMs use the fp32-input matrix instruction
// v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD = the fp32 vector rate, 157 TFLOP/s per chip: 1/16 of the bf16 rate; there is
// no TF32 / xf32 on gfx950).  What torch's fp32 nn.Linear / SDPA / LayerNorm do on this chip is the same arithmetic in
// another summation order, so the two agree to fp32 rounding (tests: 1e-5 of the output range after 99 evaluations).
//
//   mdt_gemm_f32        C = A * B^T (+ bias) with fp32 epilogues (none / GELU-tanh / SiLU / res + gate * y), batched over
//                       (sample, head) with independent strides -- also serves attention (scores = q k^T, out = p v)
//   mdt_softmax_rows_f32  in-place row softmax of the scores (timm Attention: softmax(q k^T * hd^-0.5))
//   mdt_ln_modulate_f32 LayerNorm(eps 1e-6, no affine) * (1 + scale) + shift -> fp32      (models/maskdit.py:19-20,177)
//   mdt_timestep_embed_f32, mdt_silu_f32, mdt_add_rows_f32   conditioning path / position embedding glue
//
// Reference: DiTBlock.forward (models/maskdit.py:188-192), timm Attention / Mlp (call sites :178,182),
// TimestepEmbedder (:41-60), LabelEmbedder (:75), DecoderLayer (:195-213), decoder_pos_embed add (:545).
//
// GEMM design.  f32 MFMA is 16x slower than the bf16 one while the operand bytes only double, so the kernel is
// matrix-pipe-bound by a wide margin with a plain structure: 128 x (32 NB WN) tile, 4 waves, K-tile of 32 floats staged
// through registers into a double-buffered LDS image (one workgroup barrier per K-tile; hipcc counts the waits -- nothing
// hand-counted here).  Per K-tile a wave issues 16 MB NB MFMAs (64 clocks each) against 4 (MB + NB) ds_read_b128.
// The contraction index is PERMUTED inside a K-tile: lane (r, kh) of a fragment reads four consecutive k of its row with
// one 16-byte LDS read and feeds them to four successive MFMAs -- A and B use the same permutation, so the sum is over
// all 32 k; a 16-byte-chunk XOR swizzle (chunk ^ row & 7) spreads the rows of a fragment over the banks.
#include "common.h"
#include "../../include/maskdit_hip.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace f32p {

constexpr int BK = 32;  // floats per K-tile = 128-byte LDS rows

struct Params {
  const float* A; long lda;
  const float* B; long ldb;
  int M, N, K;
  const float* bias;
  int epi;
  float* out; long ldo;
  const float* res; long ldres;
  const float* gate; long gate_ld; int rps;
  int heads;
  long a_sb, a_sh, b_sb, b_sh, o_sb, o_sh;
};

// exact-form activations (torch: F.gelu(approximate='tanh'), F.silu) -- the bf16 path's exp2 / rcp forms are 1-ulp
// approximations rounded to bf16 afterwards; here the result IS the fp32 output
__device__ __forceinline__ float gelu_tanh_f32(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.f + tanhf(u));
}
__device__ __forceinline__ float silu_f32(float x) { return x / (1.f + expf(-x)); }

template <int WM, int WN, int MB, int NB, bool BKM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const Params p, const int tiles_m, const int tiles_n) {
  constexpr int BM = WM * MB * 32, BN = WN * NB * 32;
  static_assert(WM * WN == 4, "four waves");
  constexpr int A_CH = BM * 8 / 256;                 // 16-byte chunks per thread and K-tile
  constexpr int B_CH = BN * 8 / 256;
  static_assert(A_CH >= 1 && B_CH >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-contiguous tile ids (block b runs on XCD b % 8): consecutive tiles -- which share an A row panel -- land in one L2
  const int nt = tiles_m * tiles_n;
  int t = blockIdx.x;
  {
    const int q = nt >> 3, r = nt & 7, x = t & 7, i = t >> 3;
    t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int tm = t / tiles_n, tn = t - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y, zb = z / p.heads, zh = z - zb * p.heads;
  const float* __restrict__ A = p.A + zb * p.a_sb + zh * p.a_sh;
  const float* __restrict__ Bm = p.B + zb * p.b_sb + zh * p.b_sh;
  float* __restrict__ out = p.out + zb * p.o_sb + zh * p.o_sh;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 ga[A_CH], gb[B_CH];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int q = tid + 256 * i, row = q >> 3, c = q & 7;
      const bool ok = (m0 + row) < p.M && (k0 + 4 * c) < p.K;
      const float* src = A + (long)min(m0 + row, p.M - 1) * p.lda + min(k0 + 4 * c, p.K - 4);
      const f32x4 v = *(const f32x4*)src;
      ga[i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < B_CH; ++i) {
        const int q = tid + 256 * i, row = q >> 3, c = q & 7;
        const bool ok = (n0 + row) < p.N && (k0 + 4 * c) < p.K;
        const float* src = Bm + (long)min(n0 + row, p.N - 1) * p.ldb + min(k0 + 4 * c, p.K - 4);
        const f32x4 v = *(const f32x4*)src;
        gb[i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    } else {  // B is [K][N]: a K-tile is 32 rows of BN floats
#pragma unroll
      for (int i = 0; i < B_CH; ++i) {
        const int q = tid + 256 * i, kr = q / (BN / 4), c = q % (BN / 4);
        const bool ok = (k0 + kr) < p.K && (n0 + 4 * c) < p.N;
        const float* src = Bm + (long)min(k0 + kr, p.K - 1) * p.ldb + min(n0 + 4 * c, p.N - 4);
        const f32x4 v = *(const f32x4*)src;
        gb[i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto lstore = [&](int buf) {
    float* la = lds[buf];
    float* lb = lds[buf] + BM * BK;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int q = tid + 256 * i, row = q >> 3, c = q & 7;
      *(f32x4*)(la + row * BK + 4 * (c ^ (row & 7))) = ga[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int q = tid + 256 * i;
      if (!BKM) {
        const int row = q >> 3, c = q & 7;
        *(f32x4*)(lb + row * BK + 4 * (c ^ (row & 7))) = gb[i];
      } else {
        *(f32x4*)(lb + 4 * q) = gb[i];  // [kr][BN], kr = q / (BN / 4)
      }
    }
  };

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int r = lane & 31, kh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const float* la = lds[buf] + (wm * MB * 32) * BK;
    const float* lb = lds[buf] + BM * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 af[MB], bf[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int row = 32 * i + r;  // (tile row & 7) == (row & 7): wm * MB * 32 is a multiple of 8
        af[i] = *(const f32x4*)(la + row * BK + 4 * ((2 * j + kh) ^ (row & 7)));
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int col = (wn * NB + i) * 32 + r;
        if (!BKM) {
          bf[i] = *(const f32x4*)(lb + col * BK + 4 * ((2 * j + kh) ^ (col & 7)));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bf[i][e] = lb[(8 * j + 4 * kh + e) * BN + col];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int jj = 0; jj < NB; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[jj][e], acc[i][jj], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds column n = lane & 31 of rows 8 (e / 4) + 4 (lane / 32) + e % 4 of each 32 x 32 block
  const int epi = p.epi;
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) {
    const int n = n0 + (wn * NB + jj) * 32 + r;
    if (n >= p.N) continue;
    const float bs = p.bias ? p.bias[n] : 0.f;
    // the gate of a 4-row group (rows 8 q + 4 kh .. + 3 of a 32-row block): one sample index per group instead of an
    // integer division per element (rows_per_sample is a multiple of 4 wherever a gate exists; checked by the host entry)
    float gq[MB][4];
    if (epi == MDT_F32EPI_GATE_RES) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int mq = min(m0 + (wm * MB + i) * 32 + 8 * q + 4 * kh, p.M - 1);
          gq[i][q] = p.gate ? p.gate[(long)(mq / p.rps) * p.gate_ld + n] : 1.f;
        }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (wm * MB + i) * 32 + 8 * (e >> 2) + 4 * kh + (e & 3);
        if (m >= p.M) continue;
        float y = acc[i][jj][e] + bs;
        if (epi == MDT_F32EPI_GELU) y = gelu_tanh_f32(y);
        else if (epi == MDT_F32EPI_SILU) y = silu_f32(y);
        else if (epi == MDT_F32EPI_GATE_RES) {
          y = p.res[(long)m * p.ldres + n] + gq[i][e >> 2] * y;
        }
        out[(long)m * p.ldo + n] = y;
      }
    }
  }
}

template <int WM, int WN, int MB, int NB>
int launch(const Params& p, bool bkm, int batch, hipStream_t stream) {
  constexpr int BM = WM * MB * 32, BN = WN * NB * 32;
  const int tm = cdiv(p.M, BM), tn = cdiv(p.N, BN);
  const dim3 grid(tm * tn, batch);
  if (bkm) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, MB, NB, true>), grid, dim3(256), 0, stream, p, tm, tn);
  else hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, MB, NB, false>), grid, dim3(256), 0, stream, p, tm, tn);
  return mdt_check_launch("gemm_f32");
}

// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(float* __restrict__ s, long R, int n, int n_valid, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float* r = s + row * n;
  // the row (1-4 KiB) stays in L1 / L2 over the three passes
  float mx = -3.0e38f;
  for (int i = lane; i < n_valid; i += 64) mx = fmaxf(mx, r[i] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < n_valid; i += 64) sum += expf(r[i] * scale - mx);
  const float inv = 1.f / wave_sum(sum);
  for (int i = lane; i < n; i += 64) r[i] = i < n_valid ? expf(r[i] * scale - mx) * inv : 0.f;
}

// the same with the row in registers (n = 256 NV4 floats per row, one 16-byte access per lane and 256 columns): one read,
// one exp per element, one write -- the three-pass form above runs at 2.6 TB/s on 256-wide rows, this one at the
// streaming rate
template <int NV4>
__global__ __launch_bounds__(256) void softmax_rows_f32_reg_kernel(float* __restrict__ s, long R, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float* r = s + row * (256L * NV4);
  f32x4 v[NV4];
#pragma unroll
  for (int i = 0; i < NV4; ++i) v[i] = *(const f32x4*)(r + 4 * (lane + 64 * i));
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) mx = fmaxf(fmaxf(mx, fmaxf(v[i][0], v[i][1])), fmaxf(v[i][2], v[i][3]));
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][e] = expf((v[i][e] - mx) * scale);
      sum += v[i][e];
    }
  const float inv = 1.f / wave_sum(sum);
#pragma unroll
  for (int i = 0; i < NV4; ++i) *(f32x4*)(r + 4 * (lane + 64 * i)) = (f32x4){v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv};
}

// ------------------------------------------------------------------------------------------------------------------
// Fused fp32 attention for L <= 256 tokens: one 8-wave workgroup per (sample, head), K and V of the item staged ONCE in
// LDS as fp32 rows (pitch HD + 4 floats: the 16 rows a ds_read_b128 lane group touches fall on 16 distinct 16-byte bank
// slots), a wave owns 32 queries.  Scores are computed TRANSPOSED -- S^T = K Q^T with v_mfma_f32_32x32x2_f32, K rows as the
// A operand from LDS, the wave's Q rows as the B operand from registers -- so a lane holds, for ITS query (column
// lane & 31), the scores of keys 32 kb + 8 (e / 4) + 4 (lane / 32) + e % 4: the softmax is a per-lane reduction plus one
// exchange with lane ^ 32, and the probabilities are ALREADY the B operand of O^T = V^T P^T (the two lane halves of
// accumulator register e are the two contraction slots of one MFMA; the matching V rows come from LDS as 4-byte reads,
// conflict-free).  Nothing but q, k, v is read and nothing but the output written: the three-launch form (q k^T -> HBM ->
// softmax -> HBM -> p v) moved 2.1 GB of scores per XL/2 layer at batch 128.
template <int HD, int L>
__global__ __launch_bounds__(512) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int H, float scale) {
  constexpr int HDP = HD + 4, KB = L / 32, NJ = HD / 8, NDB = (HD + 31) / 32, CPR = HD / 4;
  extern __shared__ __attribute__((aligned(16))) float attn_smem[];
  float* Ks = attn_smem;
  float* Vs = attn_smem + L * HDP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int W = H * HD;
  const long ld = 3L * W;
  const float* base = qkv + (long)b * L * ld + h * HD;
  constexpr int NCH = L * CPR, IT = (NCH + 511) / 512;
  {
    f32x4 kreg[IT], vreg[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int q = min(tid + 512 * i, NCH - 1), row = q / CPR, c = q - row * CPR;
      kreg[i] = *(const f32x4*)(base + row * ld + W + 4 * c);
      vreg[i] = *(const f32x4*)(base + row * ld + 2 * W + 4 * c);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int q = tid + 512 * i, row = q / CPR, c = q - row * CPR;
      if (q < NCH) {
        *(f32x4*)(Ks + row * HDP + 4 * c) = kreg[i];
        *(f32x4*)(Vs + row * HDP + 4 * c) = vreg[i];
      }
    }
  }
  const int r = lane & 31, kh = lane >> 5, q0 = wave * 32;
  f32x4 qf[NJ];
  {
    const float* qp = base + (long)min(q0 + r, L - 1) * ld + 4 * kh;
#pragma unroll
    for (int j = 0; j < NJ; ++j) qf[j] = *(const f32x4*)(qp + 8 * j);
  }
  __syncthreads();
  if (q0 >= L) return;
  f32x16 S[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
    const float* kp = Ks + (32 * kb + r) * HDP + 4 * kh;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const f32x4 kf = *(const f32x4*)(kp + 8 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) S[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[i], qf[j][i], S[kb], 0, 0, 0);
    }
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, S[kb][e]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      S[kb][e] = expf((S[kb][e] - mx) * scale);
      sum += S[kb][e];
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  f32x16 O[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[db][e] = 0.f;
  int dcol[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db) dcol[db] = min(32 * db + r, HD - 1);  // (columns >= HD of the last block are never stored)
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float* vp = Vs + (32 * kb + 8 * (e >> 2) + 4 * kh + (e & 3)) * HDP;
#pragma unroll
      for (int db = 0; db < NDB; ++db) O[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[dcol[db]], S[kb][e], O[db], 0, 0, 0);
    }
  float* op = out + ((long)b * L + q0 + r) * W + h * HD;
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int d0 = 32 * db + 8 * q4 + 4 * kh;
      if (d0 < HD)
        *(f32x4*)(op + d0) = (f32x4){O[db][4 * q4] * inv, O[db][4 * q4 + 1] * inv, O[db][4 * q4 + 2] * inv, O[db][4 * q4 + 3] * inv};
    }
}

template <int HD, int L>
int launch_attn(const float* qkv, float* out, int B, int H, hipStream_t stream) {
  constexpr int bytes = 2 * L * (HD + 4) * 4;
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute((const void*)attn_f32_kernel<HD, L>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
      mdt_set_error("attn_f32: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      return MDT_ERR_LAUNCH;
    }
    once = true;
  }
  const float scale = 1.f / sqrtf((float)HD);
  hipLaunchKernelGGL((attn_f32_kernel<HD, L>), dim3(B * H), dim3(512), bytes, stream, qkv, out, H, scale);
  return mdt_check_launch("attn_f32");
}

// one wave per row; NVT float4 per lane (branch-free, as norm.hip's bf16-output kernel)
template <int NVT>
__global__ __launch_bounds__(256) void ln_modulate_f32_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                              const float* __restrict__ scale, int mod_ld, int rows_per_sample,
                                                              float* __restrict__ xn, int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = D >> 2;
  const float* xr = x + (long)row * D;
  const long b = row / rows_per_sample;
  const float* sh = shift + b * mod_ld;
  const float* sc = scale + b * mod_ld;
  f32x4 v[NVT], a[NVT], m[NVT];
  int col[NVT];
  float own[NVT];
#pragma unroll
  for (int i = 0; i < NVT; ++i) {
    const int c = lane + 64 * i;
    own[i] = c < nv ? 1.f : 0.f;
    col[i] = 4 * min(c, nv - 1);
    v[i] = *(const f32x4*)(xr + col[i]);
    a[i] = *(const f32x4*)(sh + col[i]);
    m[i] = *(const f32x4*)(sc + col[i]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NVT; ++i) s += own[i] * (v[i][0] + v[i][1] + v[i][2] + v[i][3]);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NVT; ++i) {
    float qi = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      qi += d * d;
    }
    q += own[i] * qi;
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + 1e-6f);
  float* o = xn + (long)row * D;
#pragma unroll
  for (int i = 0; i < NVT; ++i) {
    f32x4 rr;
#pragma unroll
    for (int e = 0; e < 4; ++e) rr[e] = (v[i][e] - mean) * rstd * (1.f + m[i][e]) + a[i][e];
    *(f32x4*)(o + col[i]) = rr;
  }
}

__global__ void timestep_embed_f32_kernel(const float* __restrict__ t, float* __restrict__ out, int ld, int B, int dim) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (idx >= B * half) return;
  const int b = idx / half, i = idx - b * half;
  const float freq = expf(-logf(10000.f) * (float)i / (float)half);
  const float a = t[b] * freq;
  out[(long)b * ld + i] = cosf(a);
  out[(long)b * ld + half + i] = sinf(a);
}

__global__ void silu_f32_kernel(const float* __restrict__ in, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = silu_f32(in[i]);
}

// out[(b, j), :] = in[(b, j), :] + rows[j, :]   (4 floats per thread)
__global__ void add_rows_f32_kernel(const float* __restrict__ in, const float* __restrict__ rows, float* __restrict__ out, long nq,
                                    int T, int Dq) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const long row = i / Dq;
  const int cq = (int)(i - row * Dq), j = (int)(row % T);
  const f32x4 a = *(const f32x4*)(in + 4 * i);
  const f32x4 b = *(const f32x4*)(rows + ((long)j * Dq + cq) * 4);
  *(f32x4*)(out + 4 * i) = (f32x4){a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
}

}  // namespace f32p

extern "C" int mdt_gemm_f32(const mdt_gemm_f32_args* a, mdt_stream_t stream) {
  MDT_REQUIRE(a && a->A && a->B && a->out, "gemm_f32: null pointer");
  MDT_REQUIRE(a->M > 0 && a->N > 0 && a->K >= 4 && a->K % 4 == 0, "gemm_f32: M, N > 0 and K a positive multiple of 4");
  MDT_REQUIRE(a->lda % 4 == 0 && a->ldb % 4 == 0 && (((uintptr_t)a->A | (uintptr_t)a->B) & 15) == 0,
              "gemm_f32: operand rows must be 16-byte aligned");
  MDT_REQUIRE(a->epi >= MDT_F32EPI_NONE && a->epi <= MDT_F32EPI_GATE_RES, "gemm_f32: unknown epilogue");
  MDT_REQUIRE(a->epi != MDT_F32EPI_GATE_RES || (a->res && a->rows_per_sample > 0), "gemm_f32: GATE_RES needs res and rows_per_sample");
  MDT_REQUIRE(a->epi != MDT_F32EPI_GATE_RES || !a->gate || a->rows_per_sample % 4 == 0, "gemm_f32: a gate needs rows_per_sample % 4 == 0");
  const int batch = a->batch > 0 ? a->batch : 1;
  const int heads = a->heads > 0 ? a->heads : 1;
  MDT_REQUIRE(batch % heads == 0 && batch <= 65535, "gemm_f32: batch must be a multiple of heads and <= 65535");
  if (a->b_kmajor) MDT_REQUIRE(a->N % 4 == 0, "gemm_f32: k-major B needs N % 4 == 0");
  MDT_REQUIRE(((a->a_stride_b | a->a_stride_h | a->b_stride_b | a->b_stride_h) & 3) == 0, "gemm_f32: batch strides must be multiples of 4 elements");
  f32p::Params p;
  p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.bias = a->bias; p.epi = a->epi;
  p.out = a->out; p.ldo = a->ldo;
  p.res = a->res; p.ldres = a->ldres;
  p.gate = a->gate; p.gate_ld = a->gate_ld; p.rps = a->rows_per_sample > 0 ? a->rows_per_sample : 1;
  p.heads = heads;
  p.a_sb = a->a_stride_b; p.a_sh = a->a_stride_h; p.b_sb = a->b_stride_b; p.b_sh = a->b_stride_h;
  p.o_sb = a->o_stride_b; p.o_sh = a->o_stride_h;
  const hipStream_t st = (hipStream_t)stream;
  // column tile: 128 unless the problem is narrower (attention's p v with N = head_dim)
  if (a->N > 64) return f32p::launch<2, 2, 2, 2>(p, a->b_kmajor != 0, batch, st);
  if (a->N > 32) return f32p::launch<4, 1, 1, 2>(p, a->b_kmajor != 0, batch, st);
  return f32p::launch<4, 1, 1, 1>(p, a->b_kmajor != 0, batch, st);
}

extern "C" int mdt_softmax_rows_f32(float* s, long R, int n, int n_valid, float scale, mdt_stream_t stream) {
  MDT_REQUIRE(s && R > 0 && n > 0 && n_valid > 0 && n_valid <= n, "softmax_rows_f32: bad arguments");
  MDT_REQUIRE(cdiv(R, 4) > 0 && R / 4 < 2147483647L, "softmax_rows_f32: too many rows");
  const dim3 grid(cdiv(R, 4)), block(256);
  const hipStream_t st = (hipStream_t)stream;
  if (n_valid == n && n == 256) hipLaunchKernelGGL(f32p::softmax_rows_f32_reg_kernel<1>, grid, block, 0, st, s, R, scale);
  else if (n_valid == n && n == 512) hipLaunchKernelGGL(f32p::softmax_rows_f32_reg_kernel<2>, grid, block, 0, st, s, R, scale);
  else if (n_valid == n && n == 1024) hipLaunchKernelGGL(f32p::softmax_rows_f32_reg_kernel<4>, grid, block, 0, st, s, R, scale);
  else hipLaunchKernelGGL(f32p::softmax_rows_f32_kernel, grid, block, 0, st, s, R, n, n_valid, scale);
  return mdt_check_launch("softmax_rows_f32");
}

extern "C" int mdt_ln_modulate_f32(const float* x, const float* shift, const float* scale, int mod_ld, int rows_per_sample,
                                   float* xn, int M, int D, mdt_stream_t stream) {
  MDT_REQUIRE(x && shift && scale && xn, "ln_modulate_f32: null pointer");
  MDT_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1280 && M > 0 && rows_per_sample > 0 && mod_ld % 4 == 0, "ln_modulate_f32: bad shape");
  const dim3 grid(cdiv(M, 4)), block(256);
  const hipStream_t st = (hipStream_t)stream;
  switch (cdiv(D / 4, 64)) {
    case 1: hipLaunchKernelGGL(f32p::ln_modulate_f32_kernel<1>, grid, block, 0, st, x, shift, scale, mod_ld, rows_per_sample, xn, M, D); break;
    case 2: hipLaunchKernelGGL(f32p::ln_modulate_f32_kernel<2>, grid, block, 0, st, x, shift, scale, mod_ld, rows_per_sample, xn, M, D); break;
    case 3: hipLaunchKernelGGL(f32p::ln_modulate_f32_kernel<3>, grid, block, 0, st, x, shift, scale, mod_ld, rows_per_sample, xn, M, D); break;
    case 4: hipLaunchKernelGGL(f32p::ln_modulate_f32_kernel<4>, grid, block, 0, st, x, shift, scale, mod_ld, rows_per_sample, xn, M, D); break;
    default: hipLaunchKernelGGL(f32p::ln_modulate_f32_kernel<5>, grid, block, 0, st, x, shift, scale, mod_ld, rows_per_sample, xn, M, D); break;
  }
  return mdt_check_launch("ln_modulate_f32");
}

extern "C" int mdt_timestep_embed_f32(const float* t, float* out, int ld, int B, int dim, mdt_stream_t stream) {
  MDT_REQUIRE(t && out && dim % 2 == 0 && B > 0, "timestep_embed_f32: bad arguments");
  const int n = B * (dim / 2);
  hipLaunchKernelGGL(f32p::timestep_embed_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, t, out, ld, B, dim);
  return mdt_check_launch("timestep_embed_f32");
}

extern "C" int mdt_silu_f32(const float* in, float* out, long n, mdt_stream_t stream) {
  MDT_REQUIRE(in && out && n > 0, "silu_f32: bad arguments");
  hipLaunchKernelGGL(f32p::silu_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n);
  return mdt_check_launch("silu_f32");
}

extern "C" int mdt_add_rows_f32(const float* in, const float* rows, float* out, long n_rows, int T, int D, mdt_stream_t stream) {
  MDT_REQUIRE(in && rows && out && n_rows > 0 && T > 0 && D > 0 && D % 4 == 0, "add_rows_f32: bad arguments");
  const long nq = n_rows * (D / 4);
  hipLaunchKernelGGL(f32p::add_rows_f32_kernel, dim3(cdiv(nq, 256)), dim3(256), 0, (hipStream_t)stream, in, rows, out, nq, T, D / 4);
  return mdt_check_launch("add_rows_f32");
}

// timm Attention (call site models/maskdit.py:178) in exact fp32 on the packed qkv buffer [B * L, 3 * H * hd]:
// out [B * L, H * hd] = softmax(q k^T hd^-0.5) v.  Fused single-launch kernel for L in {64, 256} and hd in {32, 64, 72}
// (every shipped 256^2-latent configuration); otherwise three launches through `scores_ws` (B * H * L * L floats):
// batched q k^T, in-place row softmax, batched p v.
static bool attn_f32_fused(int L, int hd) { return (L == 64 || L == 256) && (hd == 32 || hd == 64 || hd == 72); }

extern "C" long mdt_attn_f32_ws_floats(int B, int L, int H, int hd) {
  return attn_f32_fused(L, hd) ? 0L : (long)B * H * L * L;
}

extern "C" int mdt_attn_f32(const float* qkv, float* out, float* scores_ws, int B, int L, int H, int hd, mdt_stream_t stream) {
  MDT_REQUIRE(qkv && out && B > 0 && L > 0 && H > 0 && hd > 0 && hd % 4 == 0, "attn_f32: bad arguments");
  MDT_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "attn_f32: 16-byte aligned buffers");
  const hipStream_t st = (hipStream_t)stream;
  if (attn_f32_fused(L, hd) && (long)B * H < 2147483647L) {
#define ATTN_F32_GO(HDV, LV) return f32p::launch_attn<HDV, LV>(qkv, out, B, H, st);
    if (L == 256) { if (hd == 72) ATTN_F32_GO(72, 256) else if (hd == 64) ATTN_F32_GO(64, 256) else ATTN_F32_GO(32, 256) }
    else { if (hd == 72) ATTN_F32_GO(72, 64) else if (hd == 64) ATTN_F32_GO(64, 64) else ATTN_F32_GO(32, 64) }
#undef ATTN_F32_GO
  }
  MDT_REQUIRE(scores_ws, "attn_f32: this shape takes the three-launch form and needs scores_ws (mdt_attn_f32_ws_floats)");
  MDT_REQUIRE((long)B * H <= 65535, "attn_f32: batch * heads must not exceed 65535 in the three-launch form");
  const int W = H * hd;
  mdt_gemm_f32_args g = {};
  g.A = qkv; g.lda = 3L * W; g.B = qkv + W; g.ldb = 3L * W; g.b_kmajor = 0;
  g.M = L; g.N = L; g.K = hd; g.epi = MDT_F32EPI_NONE;
  g.out = scores_ws; g.ldo = L; g.batch = B * H; g.heads = H;
  g.a_stride_b = (long)L * 3 * W; g.a_stride_h = hd; g.b_stride_b = (long)L * 3 * W; g.b_stride_h = hd;
  g.o_stride_b = (long)H * L * L; g.o_stride_h = (long)L * L;
  int rc = mdt_gemm_f32(&g, stream);
  if (rc != MDT_OK) return rc;
  rc = mdt_softmax_rows_f32(scores_ws, (long)B * H * L, L, L, 1.f / sqrtf((float)hd), stream);
  if (rc != MDT_OK) return rc;
  mdt_gemm_f32_args v = {};
  v.A = scores_ws; v.lda = L; v.B = qkv + 2 * W; v.ldb = 3L * W; v.b_kmajor = 1;
  v.M = L; v.N = hd; v.K = L; v.epi = MDT_F32EPI_NONE;
  v.out = out; v.ldo = W; v.batch = B * H; v.heads = H;
  v.a_stride_b = (long)H * L * L; v.a_stride_h = (long)L * L; v.b_stride_b = (long)L * 3 * W; v.b_stride_h = hd;
  v.o_stride_b = (long)L * W; v.o_stride_h = hd;
  return mdt_gemm_f32(&v, stream);
}
