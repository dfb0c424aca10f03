// Shared pieces of the bf16 NT GEMM kernels (gemm.hip: 128x128 tile; gemm_nt8.hip: 256 x 64*NF
// tile): parameter block, XCD-aware tile order, LDS-DMA helper and the fused row epilogue.
#pragma once
#include "common.h"
#include "../../include/maskdit_hip.h"

#define GROUP_M 8

struct NTParams {
  const bf16* A; int lda;
  const bf16* B; int ldb;
  int M, N, K;
  const float* bias;
  int epi;
  bf16* out; int ldo;
  bf16* out2; int ldo2;
  float* outf; int ldof;
  const float* res; int ldres;
  const float* gate; int gate_ld; int rows_per_sample;
  const bf16* aux; int ldaux;
  int k_splits;
  int group_m;  // tile-order group height (0 = GROUP_M)
  float* colsum;  // optional: colsum[n] += sum over rows of the bf16-rounded `out` (bias gradient of the producer)
  // implicit 3x3 convolution (gemm_nt8 CONV instantiations only; mdt_conv3x3_nhwc): A is not a matrix but an NHWC bf16
  // activation [B, Hi, Hi, C] whose 256 bytes in front are zero; row m = output pixel (b, y, x) of an Ho x Ho image
  // (Ho = Hi << conv_up, nearest-neighbour up-sampling folded into the gather), K index = (tap, channel)
  int conv_ho_log2, conv_up, conv_c;
  // ... and its fused epilogue options (round 4): outf = acc + bias (+ res, the ResnetBlock skip connection, fp32 [M, ldres]);
  // gn_sums[b, g, 0..1] += (sum, sum of squares) of the stored values of GroupNorm group g = column >> gn_cpg_log2 of
  // sample b = row >> (2 * conv_ho_log2) -- the statistics the NEXT layer's GroupNorm needs, instead of a separate pass
  float* gn_sums; int gn_cpg_log2;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // blocks are dispatched round-robin over the 8 XCDs; give each XCD a contiguous id range
  // (bijective for any nwg).
  int q = nwg >> 3, r = nwg & 7;
  int xcd = bid & 7, idx = bid >> 3;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ void tile_coords(int s, int tiles_m, int tiles_n, int& tm, int& tn, int group_m = GROUP_M) {
  int per_group = group_m * tiles_n;
  int group = s / per_group;
  int first_m = group * group_m;
  int gm = min(tiles_m - first_m, group_m);
  int in = s - group * per_group;
  tm = first_m + in % gm;
  tn = in / gm;
}


template <int CW> __device__ __forceinline__ void store_bf16_row(bf16* o, const float* v) {
  static_assert(CW % 4 == 0, "column group must be a multiple of 4");
#pragma unroll
  for (int q = 0; q + 8 <= CW; q += 8) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = f2bf(v[q + e]);
    *(bf16x8*)(o + q) = t;
  }
  if (CW % 8) {
    bf16x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = f2bf(v[CW - 4 + e]);
    *(bf16x4*)(o + CW - 4) = t;
  }
}

template <int CW> __device__ __forceinline__ void load_bf16_row(const bf16* a, float* h) {
#pragma unroll
  for (int q = 0; q + 8 <= CW; q += 8) {
    bf16x8 t = *(const bf16x8*)(a + q);
#pragma unroll
    for (int e = 0; e < 8; ++e) h[q + e] = bf2f(t[e]);
  }
  if (CW % 8) {
    bf16x4 t = *(const bf16x4*)(a + CW - 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) h[CW - 4 + e] = bf2f(t[e]);
  }
}

// Fused epilogue on CW consecutive columns [n, n+CW) of output row m (accumulators in v[]).
// Epilogue semantics: include/maskdit_hip.h (enum mdt_epilogue).  Split in two so that a kernel
// can issue the global loads of row i+1 (residual, gate, saved pre-activation) before it does the
// arithmetic and stores of row i:
//   nt_epilogue_prefetch : loads only       nt_epilogue_finish : arithmetic + stores
template <int CW> struct NtPre {
  float ra[CW];    // residual row (GATE_RES) or saved pre-activation (DGELU / DSILU): never both
  float gate[CW];
};

template <int CW> __device__ __forceinline__ void nt_load_bias(const NTParams& p, int n, float* bias) {
#pragma unroll
  for (int q = 0; q < CW; q += 4) {
    f32x4 b = p.bias ? *(const f32x4*)(p.bias + n + q) : (f32x4){0.f, 0.f, 0.f, 0.f};
    bias[q] = b[0]; bias[q + 1] = b[1]; bias[q + 2] = b[2]; bias[q + 3] = b[3];
  }
}

template <int CW> __device__ __forceinline__ void nt_epilogue_prefetch(const NTParams& p, int m, int n, NtPre<CW>& d) {
  // every field is written on every path so that the struct stays in registers (no stack object)
#pragma unroll
  for (int q = 0; q < CW; ++q) { d.ra[q] = 0.f; d.gate[q] = 0.f; }
  const int epi = p.epi & 0xff;
  if (m >= p.M) {
  } else if (epi == MDT_EPI_GATE_RES) {
    const float* g = p.gate + (long)(m / p.rows_per_sample) * p.gate_ld + n;
    const float* rs = p.res + (long)m * p.ldres + n;
#pragma unroll
    for (int q = 0; q < CW; q += 4) {
      f32x4 gv = *(const f32x4*)(g + q);
      f32x4 rv = *(const f32x4*)(rs + q);
      d.gate[q] = gv[0]; d.gate[q + 1] = gv[1]; d.gate[q + 2] = gv[2]; d.gate[q + 3] = gv[3];
      d.ra[q] = rv[0]; d.ra[q + 1] = rv[1]; d.ra[q + 2] = rv[2]; d.ra[q + 3] = rv[3];
    }
  } else if (epi == MDT_EPI_DGELU || epi == MDT_EPI_DSILU) {
    load_bf16_row<CW>(p.aux + (long)m * p.ldaux + n, d.ra);
  }
}

template <int CW>
__device__ __forceinline__ void nt_epilogue_finish(const NTParams& p, int m, int n, float* v, const float* bias,
                                                   const NtPre<CW>& d, float* csum) {
  if (m >= p.M) return;
#pragma unroll
  for (int q = 0; q < CW; ++q) v[q] += bias[q];
  const int epi = p.epi & 0xff;
  if (epi == MDT_EPI_DGELU || epi == MDT_EPI_DSILU) {
#pragma unroll
    for (int q = 0; q < CW; ++q) v[q] *= (epi == MDT_EPI_DGELU) ? gelu_tanh_grad(d.ra[q]) : silu_grad(d.ra[q]);
  }
  if (epi == MDT_EPI_F32) {
    float* o = p.outf + (long)m * p.ldof + n;
    if (p.k_splits > 1) {  // split-K partial sum: accumulate (bias was added by split 0 only)
#pragma unroll
      for (int q = 0; q < CW; ++q) atomic_add_f32(o + q, v[q]);
      return;
    }
#pragma unroll
    for (int q = 0; q < CW; q += 4) *(f32x4*)(o + q) = (f32x4){v[q], v[q + 1], v[q + 2], v[q + 3]};
  }
  // everything downstream sees the bf16-rounded value (it is what gets stored and re-read)
  float y[CW];
#pragma unroll
  for (int q = 0; q < CW; ++q) y[q] = bf2f(f2bf(v[q]));
#pragma unroll
  for (int q = 0; q < CW; ++q) csum[q] += y[q];  // column sums of the stored values (flushed only if p.colsum)
  if (p.out) store_bf16_row<CW>(p.out + (long)m * p.ldo + n, y);
  if (epi == MDT_EPI_GELU || epi == MDT_EPI_SILU) {
    float a[CW];
#pragma unroll
    for (int q = 0; q < CW; ++q) a[q] = (epi == MDT_EPI_GELU) ? gelu_tanh(y[q]) : silu(y[q]);
    store_bf16_row<CW>(p.out2 + (long)m * p.ldo2 + n, a);
  } else if (epi == MDT_EPI_GATE_RES) {
    float* o = p.outf + (long)m * p.ldof + n;
#pragma unroll
    for (int q = 0; q < CW; q += 4) {
      f32x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = d.ra[q + e] + d.gate[q + e] * y[q + e];
      *(f32x4*)(o + q) = ov;
    }
  }
}

template <int CW> __device__ __forceinline__ void nt_epilogue_row(const NTParams& p, int m, int n, float* v, float* csum) {
  float bias[CW];
  NtPre<CW> d;
  nt_load_bias<CW>(p, n, bias);
  nt_epilogue_prefetch<CW>(p, m, n, d);
  nt_epilogue_finish<CW>(p, m, n, v, bias, d, csum);
}

// A lane's csum[CW] holds partial column sums for columns [n, n+CW) over the rows it processed (row =
// lane >> 2 within each 16-row band): combine the 16 row-lanes that share lane & 3, one atomic per column.
template <int CW> __device__ __forceinline__ void nt_colsum_flush(const NTParams& p, int n, float* csum, int lane) {
#pragma unroll
  for (int q = 0; q < CW; ++q) {
    float s = csum[q];
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 4) atomic_add_f32(p.colsum + n + q, s);
  }
}

int launch_gemm_nt8(const NTParams& p, int nf, int wr, hipStream_t stream);
int nt8_num_cus();
int nt8_max_nf(int epi);
bool nt8o_eligible(const NTParams& p);                               // gemm_nt8o.hip: the wave-specialised overlap form
int launch_gemm_nt8o(const NTParams& p, int nl, int dbg, hipStream_t stream);
int launch_gemm_tn8(const bf16* A, int lda, const bf16* B, int ldb, int M, int N1, int N2, float* C, int ldc,
                    hipStream_t stream, float* colsum_a = nullptr, int* colsum_done = nullptr, int forced_splits = 0);
