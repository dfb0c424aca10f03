// Glue kernels of the KL-autoencoder DECODER that follows the sampler (reference:
// autoencoder.py:306-410 Decoder, :449-453 FrozenAutoencoderKL.decode, called at sample.py:248,273-284).
//
// The decoder is 3x3 convolutions (implicit GEMMs), GroupNorm(32, eps 1e-6, affine) + swish in front of each, nearest
// 2x up-sampling, one single-head attention block at 32x32, residual adds.  Here: activations are NHWC fp32
// [B*H*W, C] matrices; every convolution is ONE mdt_gemm_nt launch (bf16 MFMA, fp32 accumulate, fp32 output + bias)
// on an im2col matrix [B*Ho*Wo, 9*C] that the kernel below writes with GroupNorm + swish (+ the 2x nearest
// up-sampling and the zero padding) already applied -- the normalised activation is never materialised on its own.
//
//   mdt_gn_stats        : per (sample, group) sum / sum of squares   (HBM: one read of x)
//   mdt_gn_im2col       : [norm + affine] [swish] [2x nearest] k x k taps -> bf16 rows, zero padded
//   mdt_softmax_rows    : the attention block's row softmax (fp32 scores -> bf16 probabilities)
//   mdt_vae_prologue    : z / scale_factor -> post_quant_conv (1x1, 4 -> 4), NCHW -> NHWC
//   mdt_vae_epilogue    : NHWC [.., ld] fp32 -> NCHW image [B, C_out, H, W]
#include "common.h"
#include "../../include/maskdit_hip.h"

// ------------------------------------------------------------------------------------------
// GroupNorm statistics.  grid (pixel chunks, B); 256 threads = (C/4) channel quads x (1024/C) pixel lanes; every thread
// accumulates its 4 channels over its pixels, partials of the lanes / of the quads of one group are combined in LDS and
// added to sums[b, g, 0..1] (fp32 atomics; the host entry clears `sums` on the stream first).
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ sums, int HW, int C,
                                                       int groups, int px_per_block) {
  __shared__ float red[2][256];
  const int b = blockIdx.y;
  const int quads = C >> 2;                 // threads per pixel
  const int lanes = 256 / quads;            // pixels per iteration
  const int q = threadIdx.x % quads, pl = threadIdx.x / quads;
  const int p0 = blockIdx.x * px_per_block, p1 = min(p0 + px_per_block, HW);
  float s = 0.f, ss = 0.f;
  if (pl < lanes) {
    const float* base = x + ((long)b * HW) * C + 4 * q;
    for (int p = p0 + pl; p < p1; p += lanes) {
      const f32x4 v = *(const f32x4*)(base + (long)p * C);
      s += v[0] + v[1] + v[2] + v[3];
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  const int qpg = (C / groups) >> 2;        // channel quads per group (>= 1: C / groups is a multiple of 4)
  if (threadIdx.x < groups) {
    const int g = threadIdx.x;
    float a = 0.f, c = 0.f;
    for (int l = 0; l < lanes; ++l)
      for (int k = 0; k < qpg; ++k) {
        a += red[0][l * quads + g * qpg + k];
        c += red[1][l * quads + g * qpg + k];
      }
    atomic_add_f32(sums + ((long)b * groups + g) * 2, a);
    atomic_add_f32(sums + ((long)b * groups + g) * 2 + 1, c);
  }
}

// ------------------------------------------------------------------------------------------
// im2col with the pre-convolution pointwise work fused.  One thread = 8 consecutive bf16 of one output row
// (8 channels of one tap).  x: fp32 NHWC [B, H, W, C]; output rows = pixels of the (optionally 2x up-sampled) image,
// columns = (tap ky, kx | channel), zero beyond ks*ks*C up to Kp.
__global__ __launch_bounds__(256) void gn_im2col_kernel(const float* __restrict__ x, const float* __restrict__ sums,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        bf16* __restrict__ col, int B, int H, int W, int C, int groups,
                                                        int ks, int up, int swish, int Kp, float inv_n) {
  const int Ho = H << up, Wo = W << up;
  const int chunks = Kp >> 3;
  const long total = (long)B * Ho * Wo * chunks;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % chunks);
  const long row = idx / chunks;
  const int xo = (int)(row % Wo);
  const int yo = (int)((row / Wo) % Ho);
  const int b = (int)(row / ((long)Wo * Ho));
  const int k0 = ch * 8;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)0.f;
  if (k0 < ks * ks * C) {
    const int tap = k0 / C, c0 = k0 - tap * C;  // C % 8 == 0 (or C == 4 with ks*ks*C padded: handled below)
    const int pad = ks >> 1;
    if (C >= 8) {
      const int yy = yo + tap / ks - pad, xx = xo + tap % ks - pad;  // coordinates in the (up-sampled) input image
      if (yy >= 0 && yy < Ho && xx >= 0 && xx < Wo) {
        const float* src = x + (((long)b * H + (yy >> up)) * W + (xx >> up)) * C + c0;
        const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (sums) {
          // the eight channels of a chunk lie in one GroupNorm group (C / groups >= 8) or in two halves of four
          // (C = 128: four channels per group): mean / rstd once per half, the affine parameters as two 16-byte loads
          // (round 4: the per-element form -- sixteen scalar loads, eight rsqrt per thread -- ran the 128-channel 256 x 256
          // layers at 1.8 TB/s)
          const int cpg = C / groups;
          float mean[2], rstd[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int g = (c0 + 4 * hh) / cpg;
            const float2 sq = *(const float2*)(sums + ((long)b * groups + g) * 2);
            mean[hh] = sq.x * inv_n;
            rstd[hh] = rsqrtf(fmaxf(sq.y * inv_n - mean[hh] * mean[hh], 0.f) + 1e-6f);
          }
          const f32x4 ga0 = *(const f32x4*)(gamma + c0), ga1 = *(const f32x4*)(gamma + c0 + 4);
          const f32x4 be0 = *(const f32x4*)(beta + c0), be1 = *(const f32x4*)(beta + c0 + 4);
          if (cpg >= 4 && cpg % 4 == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = (v[e] - mean[0]) * rstd[0] * ga0[e] + be0[e];
              v[4 + e] = (v[4 + e] - mean[1]) * rstd[1] * ga1[e] + be1[e];
            }
          } else {  // fewer than four channels per group: the general form
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int g = (c0 + e) / cpg;
              const float m = sums[((long)b * groups + g) * 2] * inv_n;
              const float var = fmaxf(sums[((long)b * groups + g) * 2 + 1] * inv_n - m * m, 0.f);
              v[e] = (v[e] - m) * rsqrtf(var + 1e-6f) * gamma[c0 + e] + beta[c0 + e];
            }
          }
        }
        if (swish) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = silu(v[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
      }
    } else {  // C == 4 (conv_in): a chunk holds two taps of 4 channels; no norm on this path
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = (k0 >> 2) + half;
        if (t < ks * ks) {
          const int yy = yo + t / ks - pad, xx = xo + t % ks - pad;
          if (yy >= 0 && yy < Ho && xx >= 0 && xx < Wo) {
            const f32x4 v = *(const f32x4*)(x + (((long)b * H + (yy >> up)) * W + (xx >> up)) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * half + e] = f2bf(swish ? silu(v[e]) : v[e]);
          }
        }
      }
    }
  }
  *(bf16x8*)(col + row * Kp + k0) = o;
}

// ------------------------------------------------------------------------------------------
// row softmax: one wave per row of n (<= 4096, multiple of 64) fp32 scores, scaled; bf16 probabilities out
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ in, bf16* __restrict__ out, int R, int n,
                                                           float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* r = in + row * n;
  // three passes over the row (it sits in L2 / L1: 4 KiB at n = 1024): max, sum, write
  float mx = -1e30f;
  for (int i = lane; i < n; i += 64) mx = fmaxf(mx, r[i] * scale);
  mx = wave_max(mx);
  float s = 0.f;
  for (int i = lane; i < n; i += 64) s += __expf(r[i] * scale - mx);
  const float inv = 1.f / wave_sum(s);
  bf16* o = out + row * n;
  for (int i = lane; i < n; i += 64) o[i] = f2bf(__expf(r[i] * scale - mx) * inv);
}

__global__ __launch_bounds__(256) void vae_prologue_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int B, int HW,
                                                           float inv_scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * HW) return;
  const int b = (int)(i / HW), p = (int)(i % HW);
  float zi[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) zi[c] = z[((long)b * 4 + c) * HW + p] * inv_scale;
  f32x4 o;
#pragma unroll
  for (int m = 0; m < 4; ++m) o[m] = bias[m] + w[4 * m] * zi[0] + w[4 * m + 1] * zi[1] + w[4 * m + 2] * zi[2] + w[4 * m + 3] * zi[3];
  *(f32x4*)(y + i * 4) = o;
}

__global__ __launch_bounds__(256) void vae_epilogue_kernel(const float* __restrict__ in, int ld, float* __restrict__ img, int B,
                                                           int HW, int Cout) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * HW) return;
  const int b = (int)(i / HW), p = (int)(i % HW);
  for (int c = 0; c < Cout; ++c) img[((long)b * Cout + c) * HW + p] = in[i * ld + c];
}

// ------------------------------------------------------------------------------------------

extern "C" int mdt_gn_stats(const float* x, float* sums, int B, int HW, int C, int groups, mdt_stream_t stream) {
  MDT_REQUIRE(x && sums, "gn_stats: null pointer");
  MDT_REQUIRE(B > 0 && HW > 0 && C % 4 == 0 && C <= 1024 && 1024 % C == 0 && groups > 0 && groups <= 256 && C % groups == 0 &&
                  (C / groups) % 4 == 0,
              "gn_stats: C must divide 1024 with C / groups a multiple of 4");
  if (hipMemsetAsync(sums, 0, sizeof(float) * 2 * (size_t)B * groups, (hipStream_t)stream) != hipSuccess) {
    mdt_set_error("gn_stats: clearing the accumulator failed");
    return MDT_ERR_LAUNCH;
  }
  int chunks = 1;
  while ((long)B * chunks < 2048 && HW / (chunks * 2) >= 64) chunks *= 2;
  const int ppb = cdiv(HW, chunks);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(cdiv(HW, ppb), B), dim3(256), 0, (hipStream_t)stream, x, sums, HW, C, groups, ppb);
  return mdt_check_launch("gn_stats");
}

extern "C" int mdt_gn_im2col(const float* x, const float* sums, const float* gamma, const float* beta, mdt_bf16* col, int B,
                             int H, int W, int C, int groups, int ksize, int upsample, int swish, int Kp, mdt_stream_t stream) {
  MDT_REQUIRE(x && col, "gn_im2col: null pointer");
  MDT_REQUIRE(!sums || (gamma && beta && groups > 0 && C % groups == 0), "gn_im2col: normalisation needs sums, gamma, beta, groups");
  MDT_REQUIRE((ksize == 1 || ksize == 3) && (upsample == 0 || upsample == 1), "gn_im2col: 1x1 / 3x3 taps, optional 2x up-sampling");
  MDT_REQUIRE((C % 8 == 0 || (C == 4 && !sums)) && Kp % 8 == 0 && Kp >= ksize * ksize * C, "gn_im2col: C % 8 (or C == 4 without norm), Kp >= k*k*C");
  const long total = (long)B * (H << upsample) * (W << upsample) * (Kp / 8);
  MDT_REQUIRE(total > 0 && total / 256 < 2147483647L, "gn_im2col: problem size");
  const float inv_n = sums ? 1.f / ((float)H * (float)W * (float)(C / groups)) : 0.f;
  hipLaunchKernelGGL(gn_im2col_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, sums, gamma, beta,
                     (bf16*)col, B, H, W, C, groups > 0 ? groups : 1, ksize, upsample, swish, Kp, inv_n);
  return mdt_check_launch("gn_im2col");
}

extern "C" int mdt_softmax_rows(const float* in, mdt_bf16* out, int R, int n, float scale, mdt_stream_t stream) {
  MDT_REQUIRE(in && out && R > 0, "softmax_rows: null pointer / empty");
  MDT_REQUIRE(n % 64 == 0 && n >= 64 && n <= 4096, "softmax_rows: n must be a multiple of 64 in [64, 4096]");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, in, (bf16*)out, R, n, scale);
  return mdt_check_launch("softmax_rows");
}

extern "C" int mdt_vae_prologue(const float* z, const float* w, const float* bias, float* y, int B, int HW, float scale_factor,
                                mdt_stream_t stream) {
  MDT_REQUIRE(z && w && bias && y && B > 0 && HW > 0 && scale_factor != 0.f, "vae_prologue: bad arguments");
  hipLaunchKernelGGL(vae_prologue_kernel, dim3(cdiv((long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, z, w, bias, y, B, HW,
                     1.f / scale_factor);
  return mdt_check_launch("vae_prologue");
}

extern "C" int mdt_vae_epilogue(const float* in, int ld, float* img, int B, int HW, int Cout, mdt_stream_t stream) {
  MDT_REQUIRE(in && img && B > 0 && HW > 0 && Cout > 0 && ld >= Cout, "vae_epilogue: bad arguments");
  hipLaunchKernelGGL(vae_epilogue_kernel, dim3(cdiv((long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, in, ld, img, B, HW, Cout);
  return mdt_check_launch("vae_epilogue");
}
