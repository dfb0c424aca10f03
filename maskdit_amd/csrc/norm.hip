// HBM-bound glue around the GEMMs: LayerNorm + adaLN modulate (fwd / bwd), gated-residual
// backward, bias-gradient column sums.  All row-wise kernels are one wave per row with
// 16-byte vector accesses; per-sample reductions (d shift / d scale / d gate, shape [B, D])
// are accumulated per lane in registers across the rows a workgroup owns, then combined
// through LDS and one f32 atomic per column per workgroup.
//
// Reference: modulate (models/maskdit.py:19-20), LayerNorm(eps 1e-6, no affine) (:177,179),
// `x + gate.unsqueeze(1) * f(...)` (:190-191); backward = what autograd derives for them.
#include "common.h"
#include "../../include/maskdit_hip.h"

#define MAXV 5  // up to 5 float4 per lane => D <= 1280

// ------------------------------------------------------------------------------------------
// Branch-free: NVT = ceil(D / 4 / 64) float4 per lane; a lane whose quad lies beyond the row shadows the row's last
// quad (same loads, same result, duplicate store) and is masked out of the two sums -- all loads of the row, including
// the per-sample shift / scale vectors, are issued before the first use (with `if (c < nv)` around them hipcc waited
// for every quad separately: 4.1 TB/s).
template <int NVT>
__global__ __launch_bounds__(256) void ln_modulate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                              const float* __restrict__ scale, int mod_ld,
                                                              int rows_per_sample, bf16* __restrict__ xn,
                                                              float* __restrict__ stats, int M, int D) {
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
  const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-6f);
  bf16* o = xn + (long)row * D;
#pragma unroll
  for (int i = 0; i < NVT; ++i) {
    bf16x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = f2bf((v[i][e] - mean) * rstd * (1.f + m[i][e]) + a[i][e]);
    *(bf16x4*)(o + col[i]) = r;
  }
  if (lane == 0) {
    stats[2 * (long)row] = mean;
    stats[2 * (long)row + 1] = rstd;
  }
}

// ------------------------------------------------------------------------------------------
// The residual add that FEEDS a LayerNorm, folded into it (round 6): x = xres + gate[b] * y is what `x + gate * f(x)`
// (models/maskdit.py:190-191) hands to the next `modulate(norm(x))` (:188-189).  Rounds 1-5 formed x in the epilogue of
// the GEMM that produced y (MDT_EPI_GATE_RES: fp32 residual in, fp32 x out, bf16 y out = 10 B / element) and then read x
// again here (6 B / element).  Now that GEMM stores only y (2 B; plain class: its stores hide far better than the
// residual round trip did) and this pass reads xres (4) + y (2) and writes x (4) + xn (2): 14 instead of 16 B per element
// and step, and the slowest GEMM class of the training forward (0.17-0.27 of the MFMA peak) becomes the plain one.
// y is the bf16 value the GEMM stored -- the same rounded value MDT_EPI_GATE_RES added -- so x is bit-identical.
template <int NVT>
__global__ __launch_bounds__(256) void ln_modulate_fwd_res_kernel(const float* __restrict__ xres, const bf16* __restrict__ y,
                                                                  const float* __restrict__ gate, int gate_ld,
                                                                  const float* __restrict__ shift, const float* __restrict__ scale,
                                                                  int mod_ld, int rows_per_sample, float* __restrict__ x,
                                                                  bf16* __restrict__ xn, float* __restrict__ stats, int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = D >> 2;
  const float* xr = xres + (long)row * D;
  const bf16* yr = y + (long)row * D;
  const long b = row / rows_per_sample;
  const float* sh = shift + b * mod_ld;
  const float* sc = scale + b * mod_ld;
  const float* gt = gate + b * gate_ld;
  f32x4 v[NVT], a[NVT], m[NVT], g[NVT];
  bf16x4 yv[NVT];
  int col[NVT];
  float own[NVT];
#pragma unroll
  for (int i = 0; i < NVT; ++i) {
    const int c = lane + 64 * i;
    own[i] = c < nv ? 1.f : 0.f;
    col[i] = 4 * min(c, nv - 1);
    v[i] = *(const f32x4*)(xr + col[i]);
    yv[i] = *(const bf16x4*)(yr + col[i]);
    g[i] = *(const f32x4*)(gt + col[i]);
    a[i] = *(const f32x4*)(sh + col[i]);
    m[i] = *(const f32x4*)(sc + col[i]);
  }
  float s = 0.f;
  float* xo = x + (long)row * D;
#pragma unroll
  for (int i = 0; i < NVT; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = v[i][e] + g[i][e] * bf2f(yv[i][e]);  // the GATE_RES epilogue's expression, term for term
    *(f32x4*)(xo + col[i]) = v[i];
    s += own[i] * (v[i][0] + v[i][1] + v[i][2] + v[i][3]);
  }
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
  const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-6f);
  bf16* o = xn + (long)row * D;
#pragma unroll
  for (int i = 0; i < NVT; ++i) {
    bf16x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = f2bf((v[i][e] - mean) * rstd * (1.f + m[i][e]) + a[i][e]);
    *(bf16x4*)(o + col[i]) = r;
  }
  if (lane == 0) {
    stats[2 * (long)row] = mean;
    stats[2 * (long)row + 1] = rstd;
  }
}

// ------------------------------------------------------------------------------------------
// grid (B, splits): workgroup handles rows [s*chunk, (s+1)*chunk) of sample b, 4 waves.
template <bool FUSE_GATE>
__global__ __launch_bounds__(256) void ln_modulate_bwd_kernel(const bf16* __restrict__ dxn, const float* __restrict__ x,
                                                              const float* __restrict__ stats, const float* __restrict__ scale,
                                                              int mod_ld, int rows_per_sample, int chunk,
                                                              float* __restrict__ dx, int accumulate,
                                                              float* __restrict__ dshift, float* __restrict__ dscale,
                                                              int dmod_ld, int D,
                                                              // optional fused backward of the residual gate that FED this
                                                              // LayerNorm's input (x = x_prev + gate * y): the finished dx row is
                                                              // still in registers, so dys = bf16(gate*dx), dgate += dx*y and
                                                              // dbias += dys cost one extra read of y and one write of dys
                                                              const bf16* __restrict__ gy, const float* __restrict__ ggate,
                                                              int ggate_ld, bf16* __restrict__ gdys, float* __restrict__ gdgate,
                                                              int gdgate_ld, float* __restrict__ gdbias) {
  __shared__ float red[2][4][MAXV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const int r_begin = blockIdx.y * chunk, r_end = min(r_begin + chunk, rows_per_sample);
  const int nv = D >> 2;
  const float* sc = scale + (long)b * mod_ld;
  f32x4 scl[MAXV], a_sh[MAXV], a_sc[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int c = lane + 64 * i;
    a_sh[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    a_sc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    scl[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c < nv) {
      f32x4 t = *(const f32x4*)(sc + 4 * c);
      scl[i] = (f32x4){1.f + t[0], 1.f + t[1], 1.f + t[2], 1.f + t[3]};
    }
  }
  constexpr bool fuse_gate = FUSE_GATE;
  f32x4 gt[MAXV], a_g[MAXV], a_b[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int c = lane + 64 * i;
    a_g[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    a_b[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gt[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (fuse_gate && c < nv) gt[i] = *(const f32x4*)(ggate + (long)b * ggate_ld + 4 * c);
  }
  const float invD = 1.f / (float)D;
  for (int r = r_begin + wave; r < r_end; r += 4) {
    const long row = (long)b * rows_per_sample + r;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float* xr = x + row * D;
    const bf16* gr = dxn + row * D;
    f32x4 xh[MAXV], gm[MAXV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      int c = lane + 64 * i;
      if (c < nv) {
        f32x4 xv = *(const f32x4*)(xr + 4 * c);
        bf16x4 gv = *(const bf16x4*)(gr + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float gg = bf2f(gv[e]);
          float h = (xv[e] - mean) * rstd;
          a_sh[i][e] += gg;
          a_sc[i][e] += gg * h;
          float gmod = gg * scl[i][e];
          xh[i][e] = h;
          gm[i][e] = gmod;
          c1 += gmod;
          c2 += gmod * h;
        }
      }
    }
    c1 = wave_sum(c1) * invD;
    c2 = wave_sum(c2) * invD;
    float* dr = dx + row * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      int c = lane + 64 * i;
      if (c < nv) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (gm[i][e] - c1 - xh[i][e] * c2);
        if (accumulate) {
          f32x4 p = *(const f32x4*)(dr + 4 * c);
          o[0] += p[0]; o[1] += p[1]; o[2] += p[2]; o[3] += p[3];
        }
        *(f32x4*)(dr + 4 * c) = o;
        if (fuse_gate) {
          bf16x4 yv = *(const bf16x4*)(gy + row * D + 4 * c);
          bf16x4 dy;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a_g[i][e] += o[e] * bf2f(yv[e]);
            dy[e] = f2bf(o[e] * gt[i][e]);
            a_b[i][e] += bf2f(dy[e]);
          }
          *(bf16x4*)(gdys + row * D + 4 * c) = dy;
        }
      }
    }
  }
  // combine the 4 waves' per-column partials, one atomic per column per workgroup
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int c = lane + 64 * i;
    if (c < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[0][wave][4 * c + e] = a_sh[i][e];
        red[1][wave][4 * c + e] = a_sc[i][e];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float s0 = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    float s1 = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    atomic_add_f32(dshift + (long)b * dmod_ld + c, s0);
    atomic_add_f32(dscale + (long)b * dmod_ld + c, s1);
  }
  if (fuse_gate) {  // same reduction for the gate / bias partials (the LDS buffer is reused)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      int c = lane + 64 * i;
      if (c < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[0][wave][4 * c + e] = a_g[i][e];
          red[1][wave][4 * c + e] = a_b[i][e];
        }
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
      float s0 = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
      float s1 = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
      atomic_add_f32(gdgate + (long)b * gdgate_ld + c, s0);
      if (gdbias) atomic_add_f32(gdbias + c, s1);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Column-split form of the fused LayerNorm-modulate + residual-gate backward (round 2).  The row-per-wave kernel above
// needs 4 x 20 per-column running sums per lane (214 VGPRs, 2 waves/SIMD) and loses to the two separate kernels.  Here
// a PAIR of waves shares a row, each wave one half of the columns: 4 x 12 running sums per lane, the two row
// statistics (sum g', sum g' xhat) are exchanged through LDS with one workgroup barrier per row.  Workgroup = 2 pairs;
// pair p takes rows r_begin + p, + 2, ...  18 B/element in one pass instead of 14 + 8 in two.
// HV = float4 per lane and half row = ceil(D / 8 / 64): 1 (D <= 512), 2 (<= 1024), 3 (<= 1536)
// Branch-free body: a lane whose column quad lies beyond the half row (i = 2, lanes >= 16 at D = 1152) works on the
// LAST quad of the half row instead -- same loads, same arithmetic and therefore the same (duplicate) stores as the
// lane that owns it -- and is masked out of the sums.  With `if (c < nvh)` around every load hipcc puts each quad into
// its own divergent block and waits for its loads one by one; straight-line code issues them all up front.  The row
// barrier is the bare s_barrier + lgkmcnt(0): __syncthreads() carries a release fence for which hipcc drains vmcnt(0),
// i.e. waits for the previous row's dx / dys stores.
template <int HV>
__global__ __launch_bounds__(256, HV == 3 ? 2 : HV == 2 ? 3 : 4) void ln_bwd_gate_split_kernel(
    const bf16* __restrict__ dxn, const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ scale,
    int mod_ld, int rows_per_sample, int chunk, float* __restrict__ dx, int accumulate, float* __restrict__ dshift,
    float* __restrict__ dscale, int dmod_ld, int D, const bf16* __restrict__ gy, const float* __restrict__ ggate, int ggate_ld,
    bf16* __restrict__ gdys, float* __restrict__ gdgate, int gdgate_ld, float* __restrict__ gdbias) {
  __shared__ float part[2][2][2][2];        // [iteration parity][pair][half][sum g', sum g' xhat]
  __shared__ float red[2][2][MAXV * 256];   // [sum kind][pair][column]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = wave >> 1, hw = wave & 1;
  const int b = blockIdx.x;
  const int r_begin = blockIdx.y * chunk, r_end = min(r_begin + chunk, rows_per_sample);
  const int nvh = D >> 3;        // float4 per half row
  const int q0 = hw * nvh;       // first float4 of this wave's half
  const float* sc = scale + (long)b * mod_ld;
  const float* gg_ = ggate + (long)b * ggate_ld;
  int col[HV];      // element offset of this lane's quad i inside a row (clamped)
  float own[HV];    // 1 where the lane owns the quad, 0 where it shadows the last one
#pragma unroll
  for (int i = 0; i < HV; ++i) {
    const int c = lane + 64 * i;
    own[i] = c < nvh ? 1.f : 0.f;
    col[i] = 4 * (q0 + min(c, nvh - 1));
  }
  f32x4 a_sh[HV], a_sc[HV], a_g[HV], a_b[HV];
#pragma unroll
  for (int i = 0; i < HV; ++i) {
    a_sh[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    a_sc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    a_g[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    a_b[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float invD = 1.f / (float)D;
  const int iters = (r_end - r_begin + 1) >> 1;
  for (int it = 0; it < iters; ++it) {
    const int r = r_begin + 2 * it + pair;
    const bool valid = r < r_end;  // wave-uniform; only the last iteration of an odd chunk has an idle pair
    const long row = (long)b * rows_per_sample + (valid ? r : r_begin);
    const float live = valid ? 1.f : 0.f;  // an idle pair re-reads row r_begin and contributes / stores nothing
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float* xr = x + row * D;
    const bf16* gr = dxn + row * D;
    float* dr = dx + row * D;
    f32x4 xv[HV], sv[HV], pv[HV];
    bf16x4 gv[HV], yv[HV];
#pragma unroll
    for (int i = 0; i < HV; ++i) {  // every load of the row up front (the second half's operands do not depend on the statistics)
      xv[i] = *(const f32x4*)(xr + col[i]);
      gv[i] = *(const bf16x4*)(gr + col[i]);
      sv[i] = *(const f32x4*)(sc + col[i]);  // per-sample vector: L1 / L2 resident
      pv[i] = *(const f32x4*)(dr + col[i]);
      yv[i] = *(const bf16x4*)(gy + row * D + col[i]);
    }
    f32x4 xh[HV], gm[HV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < HV; ++i) {
      const float m = own[i] * live;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = bf2f(gv[i][e]);
        const float h = (xv[i][e] - mean) * rstd;
        a_sh[i][e] += m * g;
        a_sc[i][e] += m * (g * h);
        const float gmod = g * (1.f + sv[i][e]);
        xh[i][e] = h;
        gm[i][e] = gmod;
        c1 += m * gmod;
        c2 += m * (gmod * h);
      }
    }
    c1 = wave_sum(c1);
    c2 = wave_sum(c2);
    if (lane == 0) {
      part[it & 1][pair][hw][0] = c1;
      part[it & 1][pair][hw][1] = c2;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    c1 = (part[it & 1][pair][0][0] + part[it & 1][pair][1][0]) * invD;
    c2 = (part[it & 1][pair][0][1] + part[it & 1][pair][1][1]) * invD;
    if (valid) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < HV; ++i) {
        const f32x4 gt = *(const f32x4*)(gg_ + col[i]);
        f32x4 o;
        bf16x4 dy;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (accumulate ? pv[i][e] : 0.f) + rstd * (gm[i][e] - c1 - xh[i][e] * c2);
          dy[e] = f2bf(o[e] * gt[e]);
          a_g[i][e] += own[i] * (o[e] * bf2f(yv[i][e]));
          a_b[i][e] += own[i] * bf2f(dy[e]);
        }
        *(f32x4*)(dr + col[i]) = o;                       // shadow lanes write the owner's values again
        *(bf16x4*)(gdys + row * D + col[i]) = dy;
      }
    }
  }
  // combine the two pairs' per-column partials, one atomic per column per workgroup (two sum kinds at a time)
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (k) __syncthreads();
#pragma unroll
    for (int i = 0; i < HV; ++i) {
      if (lane + 64 * i < nvh) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[0][pair][col[i] + e] = k ? a_g[i][e] : a_sh[i][e];
          red[1][pair][col[i] + e] = k ? a_b[i][e] : a_sc[i][e];
        }
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
      const float s0 = red[0][0][c] + red[0][1][c], s1 = red[1][0][c] + red[1][1][c];
      if (!k) {
        atomic_add_f32(dshift + (long)b * dmod_ld + c, s0);
        atomic_add_f32(dscale + (long)b * dmod_ld + c, s1);
      } else {
        atomic_add_f32(gdgate + (long)b * gdgate_ld + c, s0);
        if (gdbias) atomic_add_f32(gdbias + c, s1);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// One workgroup per (sample b, 128-column strip): 256 threads = 32 column quads x 8 row lanes; a
// thread streams rows rl, rl+8, ... (4-way unrolled so several 16-byte loads are in flight),
// keeps its dgate / dbias partial sums in registers, and the 8 row lanes are combined through
// LDS.  One atomic per (b, column) for dgate (uncontended) and B-way contention for dbias.
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dx, const bf16* __restrict__ y,
                                                       const float* __restrict__ gate, int mod_ld, int rows_per_sample,
                                                       bf16* __restrict__ dys, float* __restrict__ dgate,
                                                       int dmod_ld, float* __restrict__ dbias, int D) {
  __shared__ float red[2][8][128];
  const int b = blockIdx.x;
  const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.y * 128 + 4 * cq;
  const f32x4 g = *(const f32x4*)(gate + (long)b * mod_ld + col);
  f32x4 ag = (f32x4){0.f, 0.f, 0.f, 0.f}, ab = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long base = (long)b * rows_per_sample * D + col;
  int r = rl;
  for (; r + 24 < rows_per_sample; r += 32) {
    f32x4 d[4];
    bf16x4 yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long off = base + (long)(r + 8 * u) * D;
      d[u] = *(const f32x4*)(dx + off);
      yv[u] = *(const bf16x4*)(y + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ag[e] += d[u][e] * bf2f(yv[u][e]);
        o[e] = f2bf(d[u][e] * g[e]);
        ab[e] += bf2f(o[e]);
      }
      *(bf16x4*)(dys + base + (long)(r + 8 * u) * D) = o;
    }
  }
  for (; r < rows_per_sample; r += 8) {
    const long off = base + (long)r * D;
    f32x4 d = *(const f32x4*)(dx + off);
    bf16x4 yv = *(const bf16x4*)(y + off);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ag[e] += d[e] * bf2f(yv[e]);
      o[e] = f2bf(d[e] * g[e]);
      ab[e] += bf2f(o[e]);
    }
    *(bf16x4*)(dys + off) = o;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][rl][4 * cq + e] = ag[e];
    red[1][rl][4 * cq + e] = ab[e];
  }
  __syncthreads();
  const int c = threadIdx.x & 127, which = threadIdx.x >> 7;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += red[which][k][c];
  const int oc = blockIdx.y * 128 + c;
  if (which == 0) atomic_add_f32(dgate + (long)b * dmod_ld + oc, s);
  else if (dbias) atomic_add_f32(dbias + oc, s);
}

// ------------------------------------------------------------------------------------------
// out[n] += sum_m in[m, n]; grid (N/256 column blocks of 256, row chunks); 256 threads =
// 32 column octets x 8 row lanes.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16* __restrict__ in, int ld, float* __restrict__ out,
                                                          int M, int N, int rows_per_block) {
  __shared__ float red[8][256];
  const int co = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int n = blockIdx.x * 256 + co * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, M);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (n < N) {
    for (int r = r0 + rl; r < r1; r += 8) {
      bf16x8 v = *(const bf16x8*)(in + (long)r * ld + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][co * 8 + e] = acc[e];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < N) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][c];
    atomic_add_f32(out + blockIdx.x * 256 + c, s);
  }
}

// ------------------------------------------------------------------------------------------

static int pick_chunk(int B, int rows_per_sample) {
  // enough workgroups to fill 256 CUs a few times over without shredding the per-sample sums
  int splits = 1;
  while (B * splits < 2048 && rows_per_sample / (splits * 2) >= 8) splits *= 2;
  return cdiv(rows_per_sample, splits);
}

extern "C" int mdt_ln_modulate_fwd(const float* x, const float* shift, const float* scale, int mod_ld,
                                   int rows_per_sample, mdt_bf16* xn, float* stats, int M, int D,
                                   mdt_stream_t stream) {
  MDT_REQUIRE(x && shift && scale && xn && stats, "ln_modulate_fwd: null pointer");
  MDT_REQUIRE(D % 4 == 0 && D <= MAXV * 256, "ln_modulate_fwd: D must be a multiple of 4 and <= 1280");
  MDT_REQUIRE(M > 0 && rows_per_sample > 0 && M % rows_per_sample == 0, "ln_modulate_fwd: M must be B*rows_per_sample");
#define LN_FWD_LAUNCH(N)                                                                                          \
  hipLaunchKernelGGL(ln_modulate_fwd_kernel<N>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, shift, scale, \
                     mod_ld, rows_per_sample, (bf16*)xn, stats, M, D)
  switch (cdiv(D >> 2, 64)) {
    case 1: LN_FWD_LAUNCH(1); break;
    case 2: LN_FWD_LAUNCH(2); break;
    case 3: LN_FWD_LAUNCH(3); break;
    case 4: LN_FWD_LAUNCH(4); break;
    default: LN_FWD_LAUNCH(5); break;
  }
#undef LN_FWD_LAUNCH
  return mdt_check_launch("ln_modulate_fwd");
}

extern "C" int mdt_ln_modulate_fwd_res(const float* xres, const mdt_bf16* y, const float* gate, int gate_ld, const float* shift,
                                       const float* scale, int mod_ld, int rows_per_sample, float* x, mdt_bf16* xn, float* stats,
                                       int M, int D, mdt_stream_t stream) {
  MDT_REQUIRE(xres && y && gate && shift && scale && x && xn && stats, "ln_modulate_fwd_res: null pointer");
  MDT_REQUIRE(D % 4 == 0 && D <= MAXV * 256, "ln_modulate_fwd_res: D must be a multiple of 4 and <= 1280");
  MDT_REQUIRE(M > 0 && rows_per_sample > 0 && M % rows_per_sample == 0, "ln_modulate_fwd_res: M must be B*rows_per_sample");
  MDT_REQUIRE(gate_ld % 4 == 0 && mod_ld % 4 == 0, "ln_modulate_fwd_res: gate / modulation pitches must be multiples of 4");
#define LN_FWDR_LAUNCH(N)                                                                                                  \
  hipLaunchKernelGGL(ln_modulate_fwd_res_kernel<N>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, xres, (const bf16*)y, \
                     gate, gate_ld, shift, scale, mod_ld, rows_per_sample, x, (bf16*)xn, stats, M, D)
  switch (cdiv(D >> 2, 64)) {
    case 1: LN_FWDR_LAUNCH(1); break;
    case 2: LN_FWDR_LAUNCH(2); break;
    case 3: LN_FWDR_LAUNCH(3); break;
    case 4: LN_FWDR_LAUNCH(4); break;
    default: LN_FWDR_LAUNCH(5); break;
  }
#undef LN_FWDR_LAUNCH
  return mdt_check_launch("ln_modulate_fwd_res");
}

extern "C" int mdt_ln_modulate_bwd(const mdt_bf16* dxn, const float* x, const float* stats, const float* scale,
                                   int mod_ld, int rows_per_sample, float* dx, int accumulate, float* dshift,
                                   float* dscale, int dmod_ld, int M, int D, mdt_stream_t stream) {
  MDT_REQUIRE(dxn && x && stats && scale && dx && dshift && dscale, "ln_modulate_bwd: null pointer");
  MDT_REQUIRE(D % 4 == 0 && D <= MAXV * 256, "ln_modulate_bwd: D must be a multiple of 4 and <= 1280");
  MDT_REQUIRE(M > 0 && rows_per_sample > 0 && M % rows_per_sample == 0, "ln_modulate_bwd: M must be B*rows_per_sample");
  int B = M / rows_per_sample;
  int chunk = pick_chunk(B, rows_per_sample);
  dim3 grid(B, cdiv(rows_per_sample, chunk));
  hipLaunchKernelGGL(ln_modulate_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)dxn, x, stats,
                     scale, mod_ld, rows_per_sample, chunk, dx, accumulate, dshift, dscale, dmod_ld, D,
                     (const bf16*)nullptr, (const float*)nullptr, 0, (bf16*)nullptr, (float*)nullptr, 0, (float*)nullptr);
  return mdt_check_launch("ln_modulate_bwd");
}

extern "C" int mdt_ln_modulate_bwd_gate(const mdt_bf16* dxn, const float* x, const float* stats, const float* scale,
                                        int mod_ld, int rows_per_sample, float* dx, int accumulate, float* dshift,
                                        float* dscale, int dmod_ld, int M, int D, const mdt_bf16* y, const float* gate,
                                        int gate_ld, mdt_bf16* dys, float* dgate, int dgate_ld, float* dbias,
                                        mdt_stream_t stream) {
  MDT_REQUIRE(dxn && x && stats && scale && dx && dshift && dscale, "ln_modulate_bwd_gate: null pointer");
  MDT_REQUIRE(y && gate && dys && dgate, "ln_modulate_bwd_gate: null gate operand");
  MDT_REQUIRE(D % 4 == 0 && D <= MAXV * 256, "ln_modulate_bwd_gate: D must be a multiple of 4 and <= 1280");
  MDT_REQUIRE(M > 0 && rows_per_sample > 0 && M % rows_per_sample == 0, "ln_modulate_bwd_gate: M must be B*rows_per_sample");
  int B = M / rows_per_sample;
  int chunk = pick_chunk(B, rows_per_sample);
  // the column-split kernel pays a two-pass LDS reduction + 4 D atomics per workgroup: at least 32 rows (16 row
  // iterations) per workgroup, or small batches lose to the separate kernels (per-GPU batch 128: 90 vs 88 us)
  if (D % 8 == 0 && chunk < 32) chunk = rows_per_sample < 32 ? rows_per_sample : 32;
  dim3 grid(B, cdiv(rows_per_sample, chunk));
#define LN_SPLIT_LAUNCH(HVV)                                                                                             \
  hipLaunchKernelGGL(ln_bwd_gate_split_kernel<HVV>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)dxn, x, stats,   \
                     scale, mod_ld, rows_per_sample, chunk, dx, accumulate, dshift, dscale, dmod_ld, D, (const bf16*)y,    \
                     gate, gate_ld, (bf16*)dys, dgate, dgate_ld, dbias)
  if (D % 8 == 0 && !mdt_get_tuning_int(MDT_TUNE_LN_GATE_ROWWISE)) {  // knob "ln_gate_rowwise": the row-per-wave build (A/B)
    if (D <= 512) LN_SPLIT_LAUNCH(1);
    else if (D <= 1024) LN_SPLIT_LAUNCH(2);
    else LN_SPLIT_LAUNCH(3);
  } else
    hipLaunchKernelGGL(ln_modulate_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)dxn, x, stats,
                       scale, mod_ld, rows_per_sample, chunk, dx, accumulate, dshift, dscale, dmod_ld, D, (const bf16*)y,
                       gate, gate_ld, (bf16*)dys, dgate, dgate_ld, dbias);
#undef LN_SPLIT_LAUNCH
  return mdt_check_launch("ln_modulate_bwd_gate");
}

extern "C" int mdt_gate_bwd(const float* dx, const mdt_bf16* y, const float* gate, int mod_ld, int rows_per_sample,
                            mdt_bf16* dys, float* dgate, int dmod_ld, float* dbias, int M, int D,
                            mdt_stream_t stream) {
  MDT_REQUIRE(dx && y && gate && dys && dgate, "gate_bwd: null pointer");
  MDT_REQUIRE(D % 128 == 0, "gate_bwd: D must be a multiple of 128");
  MDT_REQUIRE(M > 0 && rows_per_sample > 0 && M % rows_per_sample == 0, "gate_bwd: M must be B*rows_per_sample");
  int B = M / rows_per_sample;
  dim3 grid(B, D / 128);
  hipLaunchKernelGGL(gate_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dx, (const bf16*)y, gate, mod_ld,
                     rows_per_sample, (bf16*)dys, dgate, dmod_ld, dbias, D);
  return mdt_check_launch("gate_bwd");
}

extern "C" int mdt_colsum_bf16(const mdt_bf16* in, int ld, float* out, int M, int N, mdt_stream_t stream) {
  MDT_REQUIRE(in && out, "colsum: null pointer");
  MDT_REQUIRE(M > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0, "colsum: N and ld must be multiples of 8");
  int col_blocks = cdiv(N, 256);
  int row_blocks = 1;
  while (col_blocks * row_blocks < 1024 && M / (row_blocks * 2) >= 64) row_blocks *= 2;
  int rpb = cdiv(M, row_blocks);
  dim3 grid(col_blocks, cdiv(M, rpb));
  hipLaunchKernelGGL(colsum_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)in, ld, out, M, N, rpb);
  return mdt_check_launch("colsum");
}
