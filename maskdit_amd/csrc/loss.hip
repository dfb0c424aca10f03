// EDM preconditioning + EDM / MAE loss, forward and backward, fused elementwise kernels on the
// [B, C, R, R] latents.
//
// Reference: EDMLoss.__call__ (train_utils/loss.py:28-60), mae_loss / patchify (:73-101),
// EDMPrecond.forward coefficients (models/maskdit.py:756-773).
#include "common.h"
#include "../../include/maskdit_hip.h"

// coef is stored component-major: coef[i * B + b], i in (c_skip, c_out, c_in, c_noise, weight, sigma, -, -)
#define NCOEF 8

__device__ __forceinline__ void edm_coefs(float sigma, float sd, float* c) {
  float s2 = sigma * sigma, d2 = sd * sd;
  c[0] = d2 / (s2 + d2);
  c[1] = sigma * sd / sqrtf(s2 + d2);
  c[2] = 1.f / sqrtf(d2 + s2);
  c[3] = logf(sigma) * 0.25f;
  c[4] = (s2 + d2) / ((sigma * sd) * (sigma * sd));
  c[5] = sigma;
  c[6] = 0.f;
  c[7] = 0.f;
}

__global__ void edm_prep_kernel(const float* __restrict__ y, const float* __restrict__ rnd, const float* __restrict__ noise,
                                float* __restrict__ coef, float* __restrict__ yn, float* __restrict__ xin, int B, int chw,
                                float P_mean, float P_std, float sd) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * chw) return;
  int b = (int)(idx / chw);
  float sigma = expf(rnd[b] * P_std + P_mean);
  float c[8];
  edm_coefs(sigma, sd, c);
  if (idx - (long)b * chw == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) coef[(long)i * B + b] = c[i];
  }
  float v = y[idx] + noise[idx] * sigma;
  yn[idx] = v;
  xin[idx] = c[2] * v;
}

__global__ void precond_coef_kernel(const float* __restrict__ sigma, float* __restrict__ coef, int B, float sd) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float c[8];
  edm_coefs(sigma[b], sd, c);
#pragma unroll
  for (int i = 0; i < 8; ++i) coef[(long)i * B + b] = c[i];
}

__global__ void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ coef, int ci, float* __restrict__ out,
                                  int B, int chw) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * chw) return;
  int b = (int)(idx / chw);
  out[idx] = coef[(long)ci * B + b] * x[idx];
}

__global__ void precond_out_kernel(const float* __restrict__ x, const float* __restrict__ F, const float* __restrict__ coef,
                                   float* __restrict__ D, int B, int chw) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * chw) return;
  int b = (int)(idx / chw);
  D[idx] = coef[b] * x[idx] + coef[(long)B + b] * F[idx];
}

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

#define MAXP 64  // C*p*p upper bound

// one workgroup per sample; thread per patch (token)
__global__ __launch_bounds__(256) void edm_loss_fwd_kernel(const float* __restrict__ F, const float* __restrict__ yn,
                                                           const float* __restrict__ y, const float* __restrict__ coef,
                                                           const float* __restrict__ mask, float mae_coef,
                                                           float* __restrict__ D, float* __restrict__ loss, int C, int R,
                                                           int p) {
  __shared__ float sm[4];
  const int b = blockIdx.x, B = gridDim.x;
  const int w = R / p, T = w * w, n = C * p * p;
  const float c_skip = coef[b], c_out = coef[(long)B + b], wgt = coef[4L * B + b];
  float s_edm = 0.f, s_mae = 0.f, n_un = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    const int th = t / w, tw = t - th * w;
    const float mk = mask ? mask[(long)b * T + t] : 0.f;
    float se = 0.f, sum = 0.f, sumsq = 0.f;
    float tv[MAXP], dv[MAXP];
    int k = 0;
    for (int c = 0; c < C; ++c)
      for (int py = 0; py < p; ++py)
        for (int px = 0; px < p; ++px, ++k) {
          long idx = (((long)b * C + c) * R + th * p + py) * R + tw * p + px;
          float ynv = yn[idx];
          float d = c_skip * ynv + c_out * F[idx];
          D[idx] = d;
          float e = d - y[idx];
          se += e * e;
          sum += ynv;
          if (k < MAXP) { tv[k] = ynv; dv[k] = d; }
        }
    if (mk == 0.f) {
      s_edm += wgt * se / (float)n;
      n_un += 1.f;
    } else if (mae_coef > 0.f) {
      float mean = sum / (float)n;
      for (int i = 0; i < n; ++i) {
        float dd = tv[i] - mean;
        sumsq += dd * dd;
      }
      float inv = rsqrtf(sumsq / (float)(n - 1) + 1e-6f);
      float sl = 0.f;
      for (int i = 0; i < n; ++i) {
        float e = dv[i] - (tv[i] - mean) * inv;
        sl += e * e;
      }
      s_mae += sl / (float)n;
    }
  }
  s_edm = block_sum_256(s_edm, sm);
  s_mae = block_sum_256(s_mae, sm);
  n_un = block_sum_256(n_un, sm);
  if (threadIdx.x == 0) {
    float l = s_edm / n_un;
    if (mask && mae_coef > 0.f) l += mae_coef * s_mae / ((float)T - n_un);
    loss[b] = l;
  }
}

__global__ __launch_bounds__(256) void edm_loss_bwd_kernel(const float* __restrict__ dloss, const float* __restrict__ D,
                                                           const float* __restrict__ yn, const float* __restrict__ y,
                                                           const float* __restrict__ coef, const float* __restrict__ mask,
                                                           float mae_coef, float* __restrict__ dF, int C, int R, int p) {
  __shared__ float sm[4];
  const int b = blockIdx.x, B = gridDim.x;
  const int w = R / p, T = w * w, n = C * p * p;
  const float c_out = coef[(long)B + b], wgt = coef[4L * B + b];
  float cnt = 0.f;
  if (mask) {
    for (int t = threadIdx.x; t < T; t += 256) cnt += (mask[(long)b * T + t] == 0.f) ? 1.f : 0.f;
    cnt = block_sum_256(cnt, sm);
  } else {
    cnt = (float)T;
  }
  const float g = dloss[b];
  const float k_edm = g * wgt * 2.f / ((float)n * cnt) * c_out;
  const float k_mae = (mask && mae_coef > 0.f) ? g * mae_coef * 2.f / ((float)n * ((float)T - cnt)) * c_out : 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    const int th = t / w, tw = t - th * w;
    const float mk = mask ? mask[(long)b * T + t] : 0.f;
    if (mk == 0.f) {
      for (int c = 0; c < C; ++c)
        for (int py = 0; py < p; ++py)
          for (int px = 0; px < p; ++px) {
            long idx = (((long)b * C + c) * R + th * p + py) * R + tw * p + px;
            dF[idx] = k_edm * (D[idx] - y[idx]);
          }
    } else {
      float sum = 0.f, sumsq = 0.f;
      for (int c = 0; c < C; ++c)
        for (int py = 0; py < p; ++py)
          for (int px = 0; px < p; ++px) sum += yn[(((long)b * C + c) * R + th * p + py) * R + tw * p + px];
      float mean = sum / (float)n;
      for (int c = 0; c < C; ++c)
        for (int py = 0; py < p; ++py)
          for (int px = 0; px < p; ++px) {
            float dd = yn[(((long)b * C + c) * R + th * p + py) * R + tw * p + px] - mean;
            sumsq += dd * dd;
          }
      float inv = rsqrtf(sumsq / (float)(n - 1) + 1e-6f);
      for (int c = 0; c < C; ++c)
        for (int py = 0; py < p; ++py)
          for (int px = 0; px < p; ++px) {
            long idx = (((long)b * C + c) * R + th * p + py) * R + tw * p + px;
            dF[idx] = k_mae * (D[idx] - (yn[idx] - mean) * inv);
          }
    }
  }
}

// ------------------------------------------------------------------------------------------

extern "C" int mdt_edm_prep(const float* y, const float* rnd_normal, const float* noise, float* coef, float* yn,
                            float* xin, int B, int chw, float P_mean, float P_std, float sigma_data,
                            mdt_stream_t stream) {
  MDT_REQUIRE(y && rnd_normal && noise && coef && yn && xin, "edm_prep: null pointer");
  long n = (long)B * chw;
  hipLaunchKernelGGL(edm_prep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, y, rnd_normal, noise, coef,
                     yn, xin, B, chw, P_mean, P_std, sigma_data);
  return mdt_check_launch("edm_prep");
}

extern "C" int mdt_precond_coef(const float* sigma, float* coef, int B, float sigma_data, mdt_stream_t stream) {
  MDT_REQUIRE(sigma && coef, "precond_coef: null pointer");
  hipLaunchKernelGGL(precond_coef_kernel, dim3(cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, sigma, coef, B, sigma_data);
  return mdt_check_launch("precond_coef");
}

extern "C" int mdt_scale_rows(const float* x, const float* coef, int coef_idx, float* out, int B, int chw,
                              mdt_stream_t stream) {
  MDT_REQUIRE(x && coef && out && coef_idx >= 0 && coef_idx < NCOEF, "scale_rows: bad arguments");
  long n = (long)B * chw;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, coef, coef_idx, out, B, chw);
  return mdt_check_launch("scale_rows");
}

extern "C" int mdt_precond_out(const float* x, const float* F, const float* coef, float* D, int B, int chw,
                               mdt_stream_t stream) {
  MDT_REQUIRE(x && F && coef && D, "precond_out: null pointer");
  long n = (long)B * chw;
  hipLaunchKernelGGL(precond_out_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, F, coef, D, B, chw);
  return mdt_check_launch("precond_out");
}

extern "C" int mdt_edm_loss_fwd(const float* F, const float* yn, const float* y, const float* coef, const float* mask,
                                float mae_coef, float* D, float* loss, int B, int C, int R, int p,
                                mdt_stream_t stream) {
  MDT_REQUIRE(F && yn && y && coef && D && loss, "edm_loss_fwd: null pointer");
  MDT_REQUIRE(C * p * p <= MAXP && R % p == 0, "edm_loss_fwd: unsupported patch geometry");
  hipLaunchKernelGGL(edm_loss_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, F, yn, y, coef, mask, mae_coef, D,
                     loss, C, R, p);
  return mdt_check_launch("edm_loss_fwd");
}

extern "C" int mdt_edm_loss_bwd(const float* dloss, const float* D, const float* yn, const float* y,
                                const float* coef, const float* mask, float mae_coef, float* dF, int B, int C, int R,
                                int p, mdt_stream_t stream) {
  MDT_REQUIRE(dloss && D && yn && y && coef && dF, "edm_loss_bwd: null pointer");
  MDT_REQUIRE(C * p * p <= MAXP && R % p == 0, "edm_loss_bwd: unsupported patch geometry");
  hipLaunchKernelGGL(edm_loss_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dloss, D, yn, y, coef, mask,
                     mae_coef, dF, C, R, p);
  return mdt_check_launch("edm_loss_bwd");
}

// ------------------------------------------------------------------------------------------
// utils.sample (utils.py:59-65): moments [B, 2C, R, R] = (mean | logvar);
// z = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * randn)      (randn drawn by the caller)
__global__ void sample_moments_kernel(const float* __restrict__ moments, const float* __restrict__ rn,
                                      float* __restrict__ z, int B, int chw, float scale) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * chw;
  if (idx >= n) return;
  const int b = (int)(idx / chw);
  const int r = (int)(idx - (long)b * chw);
  const float mean = moments[(long)b * 2 * chw + r];
  float logvar = moments[(long)b * 2 * chw + chw + r];
  logvar = fminf(fmaxf(logvar, -30.f), 20.f);
  z[idx] = scale * (mean + expf(0.5f * logvar) * rn[idx]);
}

extern "C" int mdt_sample_moments(const float* moments, const float* randn, float* z, int B, int chw, float scale,
                                  mdt_stream_t stream) {
  MDT_REQUIRE(moments && randn && z && B > 0 && chw > 0, "sample_moments: bad arguments");
  long n = (long)B * chw;
  hipLaunchKernelGGL(sample_moments_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, moments, randn, z, B,
                     chw, scale);
  return mdt_check_launch("sample_moments");
}

// class dropout (train.py:208-209): y[b, :] *= (u[b] >= p)
__global__ void class_dropout_kernel(float* __restrict__ y, const float* __restrict__ u, float p, int B, int nc) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * nc) return;
  const int b = (int)(idx / nc);
  if (!(u[b] >= p)) y[idx] = 0.f;
}

extern "C" int mdt_class_dropout(float* y, const float* u, float p, int B, int num_classes, mdt_stream_t stream) {
  MDT_REQUIRE(y && u && B > 0 && num_classes > 0, "class_dropout: bad arguments");
  long n = (long)B * num_classes;
  hipLaunchKernelGGL(class_dropout_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, y, u, p, B, num_classes);
  return mdt_check_launch("class_dropout");
}
