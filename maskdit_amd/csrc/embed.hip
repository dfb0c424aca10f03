// Token-side entry/exit of the model: patch embedding + positional embedding + kept-token
// gather, timestep features, unmask (mask-token fill) + decoder positional embedding, and
// the final adaLN-LayerNorm-Linear(->p*p*C) + unpatchify, each with its backward.
//
// Reference: timm PatchEmbed as Conv2d(k=s=p) (models/maskdit.py:278,475), mask_out_token
// (:116-127), TimestepEmbedder.timestep_embedding (:41-60), unmask_tokens (:157-163,543-545),
// FinalLayer (:216-234), unpatchify (:411-424).
#include "common.h"
#include "../../include/maskdit_hip.h"

#define PE_TOK 8      // tokens per workgroup in patch_embed_fwd
#define MAX_PV 64     // max C*p*p supported (4*2*2 = 16 on the shipped configs)

// ------------------------------------------------------------------------------------------
// out[b, j, d] = bias[d] + pos[t, d] + sum_k W[d, k] * patch(b, t)[k],  t = ids[b, j]
// patch vector order (c, ph, pw) -- the Conv2d weight layout [D, C, p, p].
__global__ __launch_bounds__(256) void patch_embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                              const float* __restrict__ W, const float* __restrict__ bias,
                                                              const float* __restrict__ pos, const int32_t* __restrict__ ids,
                                                              int ids_ld, float* __restrict__ out, int C, int R, int p,
                                                              int L, int D) {
  __shared__ float pv[PE_TOK][MAX_PV];
  __shared__ int tok[PE_TOK];
  const int b = blockIdx.y;
  const int j0 = blockIdx.x * PE_TOK;
  const int kk = C * p * p;
  const int w = R / p;
  const float sc = in_scale ? in_scale[b] : 1.f;
  for (int idx = threadIdx.x; idx < PE_TOK * kk; idx += 256) {
    int tj = idx / kk, k = idx - tj * kk;
    int j = j0 + tj;
    float val = 0.f;
    if (j < L) {
      int t = ids ? ids[(long)b * ids_ld + j] : j;
      int c = k / (p * p), rem = k - c * p * p, py = rem / p, px = rem - py * p;
      int th = t / w, tw = t - th * w;
      val = sc * x[(((long)b * C + c) * R + th * p + py) * R + tw * p + px];
      if (k == 0) tok[tj] = t;
    }
    pv[tj][k] = val;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    const float* wr = W + (long)d * kk;
    float acc[PE_TOK];
#pragma unroll
    for (int t = 0; t < PE_TOK; ++t) acc[t] = 0.f;
    for (int k = 0; k < kk; ++k) {
      float wv = wr[k];
#pragma unroll
      for (int t = 0; t < PE_TOK; ++t) acc[t] += wv * pv[t][k];
    }
    const float bv = bias[d];
#pragma unroll
    for (int t = 0; t < PE_TOK; ++t) {
      int j = j0 + t;
      if (j < L) out[((long)b * L + j) * D + d] = acc[t] + bv + pos[(long)tok[t] * D + d];
    }
  }
}

// dW[d, k] += sum_{b,j} dout[b,j,d] * patch[b,j,k]; dbias[d] += sum dout[b,j,d]
// grid (token chunks of 64 over B*L, D/256)
#define PE_CHUNKS 8
__global__ __launch_bounds__(256) void patch_embed_bwd_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                              const float* __restrict__ dout, const int32_t* __restrict__ ids,
                                                              int ids_ld, float* __restrict__ dW, float* __restrict__ dbias,
                                                              int B, int C, int R, int p, int L, int D) {
  __shared__ float pv[64][17];
  const int kk = C * p * p;  // host guarantees kk <= 16
  const int w = R / p;
  const long nrows = (long)B * L;
  const int d = blockIdx.y * 256 + threadIdx.x;
  float acc[16], ab = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  // a workgroup walks PE_CHUNKS chunks of 64 rows before touching the (heavily shared) dW / dbias
  // accumulators: 8x fewer atomics than one chunk per workgroup
  for (int ch = 0; ch < PE_CHUNKS; ++ch) {
    const long row0 = ((long)blockIdx.x * PE_CHUNKS + ch) * 64;
    if (row0 >= nrows) break;
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * kk; idx += 256) {
      int tj = idx / kk, k = idx - tj * kk;
      long row = row0 + tj;
      float val = 0.f;
      if (row < nrows) {
        int b = (int)(row / L), j = (int)(row - (long)b * L);
        int t = ids ? ids[(long)b * ids_ld + j] : j;
        int c = k / (p * p), rem = k - c * p * p, py = rem / p, px = rem - py * p;
        int th = t / w, tw = t - th * w;
        val = (in_scale ? in_scale[b] : 1.f) * x[(((long)b * C + c) * R + th * p + py) * R + tw * p + px];
      }
      pv[tj][k] = val;
    }
    __syncthreads();
    if (d < D) {
      const int nt = (int)min((long)64, nrows - row0);
      // 8 independent 4-byte loads in flight per lane (the one-load-per-iteration form ran at 0.2 TB/s: every
      // iteration waited for its own HBM round trip)
      int t = 0;
      for (; t + 8 <= nt; t += 8) {
        float g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) g[u] = dout[(row0 + t + u) * D + d];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          ab += g[u];
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[k] += g[u] * pv[t + u][k];
        }
      }
      for (; t < nt; ++t) {
        float g = dout[(row0 + t) * D + d];
        ab += g;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] += g * pv[t][k];
      }
    }
  }
  if (d >= D) return;
  for (int k = 0; k < kk; ++k) atomic_add_f32(dW + (long)d * kk + k, acc[k]);
  atomic_add_f32(dbias + d, ab);
}

// ------------------------------------------------------------------------------------------
__global__ void timestep_embed_kernel(const float* __restrict__ t, bf16* __restrict__ out, int ld, int B, int dim) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int half = dim / 2;
  if (idx >= B * half) return;
  int b = idx / half, i = idx - b * half;
  float freq = expf(-logf(10000.f) * (float)i / (float)half);
  float a = t[b] * freq;
  out[(long)b * ld + i] = f2bf(cosf(a));
  out[(long)b * ld + half + i] = f2bf(sinf(a));
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, int ldi, bf16* __restrict__ out, int ldo, int rows,
                                     int cols, int act) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cols) return;
  int r = (int)(idx / cols), c = (int)(idx - (long)r * cols);
  float v = in[(long)r * ldi + c];
  if (act == 1) v = silu(v);
  out[(long)r * ldo + c] = f2bf(v);
}

__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

// dx = bf16(dy * silu'(x))
__global__ void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, bf16* __restrict__ dx, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = f2bf(dy[i] * silu_grad(x[i]));
}

// ------------------------------------------------------------------------------------------
// out[b, j, :] = (r < L ? xdec[b, r, :] : mask_token) + pos[j, :],  r = restore[b, j]
// one wave per output row; Dd % 4 == 0
__global__ __launch_bounds__(256) void unmask_fwd_kernel(const bf16* __restrict__ xdec, const int32_t* __restrict__ restore,
                                                         int ids_ld, const float* __restrict__ mask_token,
                                                         const float* __restrict__ pos, float* __restrict__ out, int B,
                                                         int T, int L, int Lp, int Dd) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * T) return;
  const int b = (int)(row / T), j = (int)(row - (long)b * T);
  const int r = restore ? restore[(long)b * ids_ld + j] : j;
  const float* pr = pos + (long)j * Dd;
  float* o = out + row * Dd;
  for (int c = lane * 4; c < Dd; c += 256) {
    f32x4 pvv = *(const f32x4*)(pr + c);
    f32x4 v;
    if (r < L) {
      bf16x4 xv = *(const bf16x4*)(xdec + ((long)b * Lp + r) * Dd + c);
      v = (f32x4){bf2f(xv[0]), bf2f(xv[1]), bf2f(xv[2]), bf2f(xv[3])};
    } else if (mask_token) {
      v = *(const f32x4*)(mask_token + c);
    } else {
      v = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    *(f32x4*)(o + c) = (f32x4){v[0] + pvv[0], v[1] + pvv[1], v[2] + pvv[2], v[3] + pvv[3]};
  }
}

// dxdec[b, r, :] = bf16(dout[b, shuffle[b, r], :]) for r < L; dmask_token += sum over r >= L.
// grid (B, row chunks of 32 over T); thread owns a float4 column quad.
__global__ __launch_bounds__(256) void unmask_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ shuffle,
                                                         int ids_ld, bf16* __restrict__ dxdec, float* __restrict__ dmask_token,
                                                         int T, int L, int Lp, int Dd) {
  const int b = blockIdx.x;
  const int r0 = blockIdx.y * 32, r1 = min(r0 + 32, T);
  for (int cq = threadIdx.x; cq * 4 < Dd; cq += 256) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int rb = r0; rb < r1; rb += 8) {
      // eight gathered rows in flight per lane (one dependent load per iteration ran at 1.4 TB/s); rows past the
      // chunk re-read its last row and are skipped below
      f32x4 gq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = min(rb + u, r1 - 1);
        const int j = shuffle ? shuffle[(long)b * ids_ld + rr] : rr;
        gq[u] = *(const f32x4*)(dout + ((long)b * T + j) * Dd + 4 * cq);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
      const int r = rb + u;
      if (r >= r1) break;
      const f32x4 g = gq[u];
      if (r < L) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(g[e]);
        *(bf16x4*)(dxdec + ((long)b * Lp + r) * Dd + 4 * cq) = o;
      } else {
        acc[0] += g[0]; acc[1] += g[1]; acc[2] += g[2]; acc[3] += g[3];
        if (r < Lp) {  // padding row of the encoder (kept count rounded up to 64): no gradient flows into it
          bf16x4 z;
#pragma unroll
          for (int e = 0; e < 4; ++e) z[e] = (bf16)0.f;
          *(bf16x4*)(dxdec + ((long)b * Lp + r) * Dd + 4 * cq) = z;
        }
      }
      }
    }
    if (dmask_token && r1 > L) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomic_add_f32(dmask_token + 4 * cq + e, acc[e]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Final layer.  One wave per token; Dd <= 512 => <= 2 float4 per lane.  O = p*p*C <= 16.
#define FV 2
#define FO 16

__device__ __forceinline__ void unpatch_index(int t, int k, int w, int p, int C, int R, int& c, int& yy, int& xx) {
  // 'nhwpqc->nchpwq' (models/maskdit.py:421-423): k = (py*p + px)*C + c
  int th = t / w, tw = t - th * w;
  c = k % C;
  int pq = k / C;
  int py = pq / p, px = pq - py * p;
  yy = th * p + py;
  xx = tw * p + px;
}

__global__ __launch_bounds__(256) void final_fwd_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                        const float* __restrict__ scale, int mod_ld,
                                                        const float* __restrict__ W, const float* __restrict__ bias,
                                                        float* __restrict__ F, float* __restrict__ stats, int B, int T,
                                                        int Dd, int C, int p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * T) return;
  const int b = (int)(row / T), t = (int)(row - (long)b * T);
  const int nv = Dd >> 2, O = p * p * C;
  const float* xr = x + row * Dd;
  f32x4 v[FV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < FV; ++i) {
    int c = lane + 64 * i;
    v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c < nv) {
      v[i] = *(const f32x4*)(xr + 4 * c);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(s) / (float)Dd;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < FV; ++i) {
    int c = lane + 64 * i;
    if (c < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d = v[i][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)Dd + 1e-6f);
  const float* sh = shift + (long)b * mod_ld;
  const float* sc = scale + (long)b * mod_ld;
#pragma unroll
  for (int i = 0; i < FV; ++i) {
    int c = lane + 64 * i;
    if (c < nv) {
      f32x4 a = *(const f32x4*)(sh + 4 * c), m = *(const f32x4*)(sc + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * (1.f + m[e]) + a[e];
    }
  }
  float outv = 0.f;  // lane k (< O) keeps output k
  for (int k = 0; k < O; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < FV; ++i) {
      int c = lane + 64 * i;
      if (c < nv) {
        f32x4 wv = *(const f32x4*)(W + (long)k * Dd + 4 * c);
        acc += v[i][0] * wv[0] + v[i][1] * wv[1] + v[i][2] * wv[2] + v[i][3] * wv[3];
      }
    }
    acc = wave_sum(acc);
    if (lane == k) outv = acc + bias[k];
  }
  if (lane < O) {
    int c, yy, xx;
    const int R = (int)(sqrtf((float)T) + 0.5f) * p;
    unpatch_index(t, lane, R / p, p, C, R, c, yy, xx);
    F[(((long)b * C + c) * R + yy) * R + xx] = outv;
  }
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

// grid (B, chunks); 4 waves, wave per token.  Accumulates dW [O, Dd], dbias [O], dshift/dscale.
__global__ __launch_bounds__(256, 2) void final_bwd_kernel(const float* __restrict__ dF, const float* __restrict__ x,
                                                        const float* __restrict__ stats, const float* __restrict__ shift,
                                                        const float* __restrict__ scale, int mod_ld,
                                                        const float* __restrict__ W, float* __restrict__ dx,
                                                        float* __restrict__ dW, float* __restrict__ dbias,
                                                        float* __restrict__ dshift, float* __restrict__ dscale,
                                                        int dmod_ld, int T, int chunk, int Dd, int C, int p) {
  extern __shared__ float red[];  // [4 waves][6 * Dd]: see the reduction at the end
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const int t_begin = blockIdx.y * chunk, t_end = min(t_begin + chunk, T);
  const int nv = Dd >> 2, O = p * p * C;
  const int w = (int)(sqrtf((float)T) + 0.5f), R = w * p;
  const float* sh = shift + (long)b * mod_ld;
  const float* sc = scale + (long)b * mod_ld;
  f32x4 shv[FV], scv[FV], a_sh[FV], a_sc[FV], a_w[FO][FV];
  float a_b = 0.f;  // lane k accumulates dbias[k]
#pragma unroll
  for (int i = 0; i < FV; ++i) {
    int c = lane + 64 * i;
    shv[i] = scv[i] = a_sh[i] = a_sc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c < nv) {
      shv[i] = *(const f32x4*)(sh + 4 * c);
      f32x4 m = *(const f32x4*)(sc + 4 * c);
      scv[i] = (f32x4){1.f + m[0], 1.f + m[1], 1.f + m[2], 1.f + m[3]};
    }
#pragma unroll
    for (int k = 0; k < FO; ++k) a_w[k][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float invD = 1.f / (float)Dd;
  for (int t = t_begin + wave; t < t_end; t += 4) {
    const long row = (long)b * T + t;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    // this token's 16 output gradients: lane k loads do[k], then broadcast
    float dok = 0.f;
    if (lane < O) {
      int c, yy, xx;
      unpatch_index(t, lane, w, p, C, R, c, yy, xx);
      dok = dF[(((long)b * C + c) * R + yy) * R + xx];
      a_b += dok;
    }
    f32x4 xh[FV], dxn[FV];
#pragma unroll
    for (int i = 0; i < FV; ++i) {
      int c = lane + 64 * i;
      xh[i] = dxn[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (c < nv) {
        f32x4 xv = *(const f32x4*)(x + row * Dd + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) xh[i][e] = (xv[e] - mean) * rstd;
      }
    }
#pragma unroll
    for (int k = 0; k < FO; ++k) {
      if (k < O) {
        float g = __shfl(dok, k, 64);
#pragma unroll
        for (int i = 0; i < FV; ++i) {
          int c = lane + 64 * i;
          if (c < nv) {
            f32x4 wv = *(const f32x4*)(W + (long)k * Dd + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float xn = xh[i][e] * scv[i][e] + shv[i][e];
              a_w[k][i][e] += g * xn;
              dxn[i][e] += g * wv[e];
            }
          }
        }
      }
    }
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < FV; ++i) {
      int c = lane + 64 * i;
      if (c < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a_sh[i][e] += dxn[i][e];
          a_sc[i][e] += dxn[i][e] * xh[i][e];
          float gm = dxn[i][e] * scv[i][e];
          dxn[i][e] = gm;
          c1 += gm;
          c2 += gm * xh[i][e];
        }
      }
    }
    c1 = wave_sum(c1) * invD;
    c2 = wave_sum(c2) * invD;
#pragma unroll
    for (int i = 0; i < FV; ++i) {
      int c = lane + 64 * i;
      if (c < nv) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (dxn[i][e] - c1 - xh[i][e] * c2);
        *(f32x4*)(dx + row * Dd + 4 * c) = o;
      }
    }
  }
  // cross-wave reduction through LDS, then atomics -- in three passes of six of the FO + 2 per-column sums, so that
  // the workgroup needs 48 KB of LDS instead of 144 KB (one workgroup = four waves per CU was all that fitted: 1.8 ms
  // for a pass whose HBM traffic is worth 0.25 ms)
  constexpr int RPP = 6;  // sums per pass
  static_assert((FO + 2) % RPP == 0, "passes must tile the FO + 2 sums");
  float* mine = red + (long)wave * RPP * Dd;
#pragma unroll
  for (int pass = 0; pass < (FO + 2) / RPP; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int i = 0; i < FV; ++i) {
      int c = lane + 64 * i;
      if (c < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int kk = 0; kk < RPP; ++kk) {
            const int k = pass * RPP + kk;
            mine[kk * Dd + 4 * c + e] = k < FO ? a_w[k < FO ? k : 0][i][e] : (k == FO ? a_sh[i][e] : a_sc[i][e]);
          }
        }
      }
    }
    __syncthreads();
    const int tot = RPP * Dd;
    for (int idx = threadIdx.x; idx < tot; idx += 256) {
      float s4 = red[idx] + red[tot + idx] + red[2 * tot + idx] + red[3 * tot + idx];
      int kk = idx / Dd, c = idx - kk * Dd;
      const int k = pass * RPP + kk;
      if (k < O) atomic_add_f32(dW + (long)k * Dd + c, s4);
      else if (k == FO) atomic_add_f32(dshift + (long)b * dmod_ld + c, s4);
      else if (k == FO + 1) atomic_add_f32(dscale + (long)b * dmod_ld + c, s4);
    }
  }
  if (lane < O) atomic_add_f32(dbias + lane, a_b);
}

// ------------------------------------------------------------------------------------------

extern "C" int mdt_patch_embed_fwd(const float* x, const float* in_scale, const float* W, const float* bias,
                                   const float* pos, const int32_t* ids, int ids_ld, float* out, int B, int C, int R,
                                   int p, int L, int D, mdt_stream_t stream) {
  MDT_REQUIRE(x && W && bias && pos && out, "patch_embed_fwd: null pointer");
  MDT_REQUIRE(C * p * p <= MAX_PV && R % p == 0, "patch_embed_fwd: unsupported patch geometry");
  dim3 grid(cdiv(L, PE_TOK), B);
  hipLaunchKernelGGL(patch_embed_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, in_scale, W, bias, pos, ids,
                     ids_ld, out, C, R, p, L, D);
  return mdt_check_launch("patch_embed_fwd");
}

extern "C" int mdt_patch_embed_bwd(const float* x, const float* in_scale, const float* dout, const int32_t* ids,
                                   int ids_ld, float* dW, float* dbias, int B, int C, int R, int p, int L, int D,
                                   mdt_stream_t stream) {
  MDT_REQUIRE(x && dout && dW && dbias, "patch_embed_bwd: null pointer");
  MDT_REQUIRE(C * p * p <= 16 && R % p == 0, "patch_embed_bwd: C*p*p must be <= 16");
  dim3 grid(cdiv((long)B * L, 64 * PE_CHUNKS), cdiv(D, 256));
  hipLaunchKernelGGL(patch_embed_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, in_scale, dout, ids, ids_ld,
                     dW, dbias, B, C, R, p, L, D);
  return mdt_check_launch("patch_embed_bwd");
}

extern "C" int mdt_timestep_embed(const float* t, mdt_bf16* out, int ld, int B, int dim, mdt_stream_t stream) {
  MDT_REQUIRE(t && out && dim % 2 == 0, "timestep_embed: bad arguments");
  int n = B * dim / 2;
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, t, (bf16*)out, ld, B, dim);
  return mdt_check_launch("timestep_embed");
}

extern "C" int mdt_cast_f32_bf16(const float* in, int ldi, mdt_bf16* out, int ldo, int rows, int cols, int act,
                                 mdt_stream_t stream) {
  MDT_REQUIRE(in && out, "cast: null pointer");
  long n = (long)rows * cols;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, ldi, (bf16*)out,
                     ldo, rows, cols, act);
  return mdt_check_launch("cast");
}

extern "C" int mdt_add_f32(const float* a, const float* b, float* out, long n, mdt_stream_t stream) {
  MDT_REQUIRE(a && b && out, "add: null pointer");
  hipLaunchKernelGGL(add_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  return mdt_check_launch("add");
}

extern "C" int mdt_silu_bwd(const float* dy, const float* x, mdt_bf16* dx, long n, mdt_stream_t stream) {
  MDT_REQUIRE(dy && x && dx, "silu_bwd: null pointer");
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, (bf16*)dx, n);
  return mdt_check_launch("silu_bwd");
}

extern "C" int mdt_unmask_fwd(const mdt_bf16* xdec, const int32_t* restore, int ids_ld, const float* mask_token,
                              const float* pos, float* out, int B, int T, int L, int Dd, int L_pitch,
                              mdt_stream_t stream) {
  MDT_REQUIRE(xdec && pos && out, "unmask_fwd: null pointer");
  if (L_pitch <= 0) L_pitch = L;
  MDT_REQUIRE(Dd % 4 == 0 && L <= T && L_pitch >= L, "unmask_fwd: bad shape");
  hipLaunchKernelGGL(unmask_fwd_kernel, dim3(cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)xdec, restore, ids_ld, mask_token, pos, out, B, T, L, L_pitch, Dd);
  return mdt_check_launch("unmask_fwd");
}

extern "C" int mdt_unmask_bwd(const float* dout, const int32_t* shuffle, int ids_ld, mdt_bf16* dxdec,
                              float* dmask_token, int B, int T, int L, int Dd, int L_pitch, mdt_stream_t stream) {
  MDT_REQUIRE(dout && dxdec, "unmask_bwd: null pointer");
  if (L_pitch <= 0) L_pitch = L;
  MDT_REQUIRE(Dd % 4 == 0 && L <= T && L_pitch >= L && L_pitch <= T, "unmask_bwd: bad shape");
  dim3 grid(B, cdiv(T, 32));
  hipLaunchKernelGGL(unmask_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dout, shuffle, ids_ld, (bf16*)dxdec,
                     dmask_token, T, L, L_pitch, Dd);
  return mdt_check_launch("unmask_bwd");
}

extern "C" int mdt_final_fwd(const float* x, const float* shift, const float* scale, int mod_ld, const float* W,
                             const float* bias, float* F, float* stats, int B, int T, int Dd, int C, int p,
                             mdt_stream_t stream) {
  MDT_REQUIRE(x && shift && scale && W && bias && F && stats, "final_fwd: null pointer");
  MDT_REQUIRE(Dd % 4 == 0 && Dd <= FV * 256 && p * p * C <= FO, "final_fwd: needs Dd <= 512 and p*p*C <= 16");
  hipLaunchKernelGGL(final_fwd_kernel, dim3(cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, x, shift, scale,
                     mod_ld, W, bias, F, stats, B, T, Dd, C, p);
  return mdt_check_launch("final_fwd");
}

extern "C" int mdt_final_bwd(const float* dF, const float* x, const float* stats, const float* shift,
                             const float* scale, int mod_ld, const float* W, float* dx, float* dW, float* dbias,
                             float* dshift, float* dscale, int dmod_ld, int B, int T, int Dd, int C, int p,
                             mdt_stream_t stream) {
  MDT_REQUIRE(dF && x && stats && shift && scale && W && dx && dW && dbias && dshift && dscale, "final_bwd: null pointer");
  MDT_REQUIRE(Dd % 4 == 0 && Dd <= FV * 256 && p * p * C <= FO, "final_bwd: needs Dd <= 512 and p*p*C <= 16");
  int splits = 1;
  while (B * splits < 1024 && T / (splits * 2) >= 16) splits *= 2;
  int chunk = cdiv(T, splits);
  dim3 grid(B, cdiv(T, chunk));
  size_t lds = (size_t)4 * 6 * Dd * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)final_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(final_bwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, dF, x, stats, shift, scale, mod_ld, W,
                     dx, dW, dbias, dshift, dscale, dmod_ld, T, chunk, Dd, C, p);
  return mdt_check_launch("final_bwd");
}
