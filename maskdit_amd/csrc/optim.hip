// Fused AdamW + EMA + bf16 weight-shadow refresh over the flat parameter arena, and the
// batched bf16 transposes that keep the K-major (dgrad) weight shadows in sync.
//
// Reference: apex.optimizers.FusedAdam(lr, adam_w_mode=True, weight_decay=0) (train.py:141,226)
// -- Adam with decoupled weight decay and bias correction (torch.optim.AdamW equivalent shown at
// train_wds.py:202) -- and update_ema (train_utils/helper.py:47-58: ema = d*ema + (1-d)*p).
// One pass: 20 B/param read (p, g, m, v, ema) + 18 B/param written (p, m, v, ema, bf16 shadow).
#include "common.h"
#include "../../include/maskdit_hip.h"

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ ema, bf16* __restrict__ w16, long n4,
                                                        long n, float lr, float b1, float b2, float eps, float wd,
                                                        float inv_bc1, float inv_sqrt_bc2, float ema_decay,
                                                        float gscale) {
  const long stride = (long)gridDim.x * blockDim.x;
  // two float4 per array and iteration: ten 16-byte loads in flight per lane before the first use
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 2 * stride) {
    const long i1 = i0 + stride;
    const bool two = i1 < n4;
    const long ix[2] = {i0, two ? i1 : i0};
    f32x4 pv[2], gv[2], mv[2], vv[2], ev[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pv[u] = *(const f32x4*)(p + 4 * ix[u]);
      gv[u] = *(const f32x4*)(g + 4 * ix[u]);
      mv[u] = *(const f32x4*)(m + 4 * ix[u]);
      vv[u] = *(const f32x4*)(v + 4 * ix[u]);
      ev[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ema) ev[u] = *(const f32x4*)(ema + 4 * ix[u]);  // (wave-uniform)
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      bf16x4 sh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gg = gv[u][e] * gscale;
        float pp = pv[u][e] * (1.f - lr * wd);
        float mm = b1 * mv[u][e] + (1.f - b1) * gg;
        float v2 = b2 * vv[u][e] + (1.f - b2) * gg * gg;
        float denom = sqrtf(v2) * inv_sqrt_bc2 + eps;
        pp -= (lr * inv_bc1) * (mm / denom);
        pv[u][e] = pp; mv[u][e] = mm; vv[u][e] = v2;
        sh[e] = f2bf(pp);
      }
      const long i = ix[u];
      *(f32x4*)(p + 4 * i) = pv[u];
      *(f32x4*)(m + 4 * i) = mv[u];
      *(f32x4*)(v + 4 * i) = vv[u];
      if (ema) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ev[u][e] = ema_decay * ev[u][e] + (1.f - ema_decay) * pv[u][e];
        *(f32x4*)(ema + 4 * i) = ev[u];
      }
      if (w16) *(bf16x4*)(w16 + 4 * i) = sh;
    }
  }
  // tail (n % 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    long i = (n4 << 2) + threadIdx.x;
    float gg = g[i] * gscale;
    float pp = p[i] * (1.f - lr * wd);
    float mm = b1 * m[i] + (1.f - b1) * gg;
    float v2 = b2 * v[i] + (1.f - b2) * gg * gg;
    pp -= (lr * inv_bc1) * (mm / (sqrtf(v2) * inv_sqrt_bc2 + eps));
    p[i] = pp; m[i] = mm; v[i] = v2;
    if (ema) ema[i] = ema_decay * ema[i] + (1.f - ema_decay) * pp;
    if (w16) w16[i] = f2bf(pp);
  }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, long n, float decay) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) ema[i] = decay * ema[i] + (1.f - decay) * p[i];
}

// table entry: src_off, dst_off, rows, cols, tile_start   (tiles are 64x64, row-major over ceil(rows/64) x ceil(cols/64))
__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                                const int64_t* __restrict__ table, int n_entries) {
  __shared__ bf16 tile[64][66];
  const int tid = blockIdx.x;
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {  // last entry with tile_start <= tid
    int mid = (lo + hi + 1) >> 1;
    if (table[mid * 5 + 4] <= tid) lo = mid; else hi = mid - 1;
  }
  const int64_t* e = table + lo * 5;
  const long so = e[0], dof = e[1];
  const int rows = (int)e[2], cols = (int)e[3];
  const int local = tid - (int)e[4];
  const int tc = (cols + 63) / 64;
  const int tr_ = local / tc, tc_ = local - tr_ * tc;
  const int r0 = tr_ * 64, c0 = tc_ * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  if (r0 + 64 <= rows && c0 + 64 <= cols && ((rows | cols) & 7) == 0 && ((so | dof) & 7) == 0) {
    // full tile, 16-byte accesses on both sides (round 6: the 2-byte form below moved the 2.8 GB of a step's K-major
    // shadows at 2.4 TB/s; every matrix of every shipped model takes this path).  Thread (rr, cq): row rr + 32 pass,
    // columns 8 cq .. 8 cq + 7 -> LDS; then output row (= source column) oc + 32 pass, source rows 8 rq .. 8 rq + 7.
    const int rr = threadIdx.x >> 3, cq = threadIdx.x & 7;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int r = rr + 32 * ps;
      const bf16x8 v = *(const bf16x8*)(src + so + (long)(r0 + r) * cols + c0 + 8 * cq);
#pragma unroll
      for (int e = 0; e < 8; ++e) tile[r][8 * cq + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int oc = rr + 32 * ps;  // output row = source column
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[8 * cq + e][oc];
      *(bf16x8*)(dst + dof + (long)(c0 + oc) * rows + r0 + 8 * cq) = v;
    }
    return;
  }
  for (int r = ty; r < 64; r += 4)
    if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = src[so + (long)(r0 + r) * cols + c0 + tx];
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < cols && r0 + tx < rows) dst[dof + (long)(c0 + c) * rows + r0 + tx] = tile[tx][c];
}

extern "C" int mdt_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, mdt_bf16* w16, long n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, float bc1,
                                  float bc2, float ema_decay, float grad_scale, mdt_stream_t stream) {
  MDT_REQUIRE(p && g && m && v, "adamw: null pointer");
  MDT_REQUIRE(n > 0 && bc1 > 0.f && bc2 > 0.f, "adamw: bad arguments");
  MDT_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) == 0 && ((uintptr_t)w16 & 7) == 0,
              "adamw: arenas must be 16-byte aligned");
  long n4 = n >> 2;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, (bf16*)w16, n4,
                     n, lr, beta1, beta2, eps, weight_decay, 1.f / bc1, 1.f / sqrtf(bc2), ema_decay, grad_scale);
  return mdt_check_launch("adamw_ema_step");
}

extern "C" int mdt_ema_update(float* ema, const float* p, long n, float decay, mdt_stream_t stream) {
  MDT_REQUIRE(ema && p && n > 0, "ema_update: bad arguments");
  int blocks = (int)((n + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(ema_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ema, p, n, decay);
  return mdt_check_launch("ema_update");
}

extern "C" int mdt_transpose_bf16_batched(const mdt_bf16* src, mdt_bf16* dst, const int64_t* table, int n_entries,
                                          int total_tiles, mdt_stream_t stream) {
  MDT_REQUIRE(src && dst && table && n_entries > 0 && total_tiles > 0, "transpose: bad arguments");
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const bf16*)src,
                     (bf16*)dst, table, n_entries);
  return mdt_check_launch("transpose_batched");
}
