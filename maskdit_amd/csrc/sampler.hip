// EDM Heun sampler state updates (fp64 state, fp32 network I/O) as fused elementwise kernels
// that read their per-step scalars from a device-side schedule, so that ONE captured hipGraph
// (2 network evaluations + these kernels) replays for every step.
//
// Reference: edm_sampler (sample.py:30-66) with S_churn = 0 (gamma = 0 => x_hat = x_cur,
// t_hat = t_cur); classifier-free guidance combine of DiT.forward_with_cfg
// (models/maskdit.py:580-583) and the EDMPrecond output (models/maskdit.py:764-772).
#include "common.h"
#include "../../include/maskdit_hip.h"

__device__ __forceinline__ void precond_f(float sigma, float sd, float& c_skip, float& c_out, float& c_in) {
  float s2 = sigma * sigma, d2 = sd * sd;
  c_skip = d2 / (s2 + d2);
  c_out = sigma * sd / sqrtf(s2 + d2);
  c_in = 1.f / sqrtf(d2 + s2);
}

// xin[dup copies] = c_in(t) * float(x);  sigma_out[b] = t   (which: 0 -> t_cur, 1 -> t_next)
__global__ void sampler_prep_kernel(const double* __restrict__ x, const double* __restrict__ t_steps,
                                    const int32_t* __restrict__ step_idx, int which, float* __restrict__ xin,
                                    float* __restrict__ sigma_out, int B, int chw, int dup, float sd) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * chw;
  const float sigma = (float)t_steps[*step_idx + which];
  if (idx < (long)B * dup) sigma_out[idx] = sigma;
  if (idx >= n) return;
  float cs, co, ci;
  precond_f(sigma, sd, cs, co, ci);
  float v = ci * (float)x[idx];
  xin[idx] = v;
  if (dup > 1) xin[n + idx] = v;
}

__device__ __forceinline__ double denoise(const double xh, const float* F, long idx, long n, float sigma, float sd,
                                          float cfg_scale, int use_cfg) {
  float f = F[idx];
  if (use_cfg) {
    float fu = F[n + idx];
    f = fu + cfg_scale * (f - fu);
  }
  float cs, co, ci;
  precond_f(sigma, sd, cs, co, ci);
  // the network sees x_hat.float(); D = c_skip * x + c_out * F in fp32, then .to(float64)
  return (double)(cs * (float)xh + co * f);
}

__global__ void sampler_euler_kernel(const double* __restrict__ x_hat, const float* __restrict__ F,
                                     const double* __restrict__ t_steps, const int32_t* __restrict__ step_idx,
                                     float cfg_scale, int use_cfg, double* __restrict__ x_next,
                                     double* __restrict__ d_cur, long n, float sd) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int i = *step_idx;
  const double t_hat = t_steps[i], t_next = t_steps[i + 1];
  const double xh = x_hat[idx];
  const double den = denoise(xh, F, idx, n, (float)t_hat, sd, cfg_scale, use_cfg);
  const double d = (xh - den) / t_hat;
  d_cur[idx] = d;
  x_next[idx] = xh + (t_next - t_hat) * d;
}

__global__ void sampler_heun_kernel(const double* __restrict__ x_hat, double* __restrict__ x_next,
                                    const float* __restrict__ F, const double* __restrict__ d_cur,
                                    const double* __restrict__ t_steps, const int32_t* __restrict__ step_idx,
                                    float cfg_scale, int use_cfg, long n, float sd) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int i = *step_idx;
  const double t_hat = t_steps[i], t_next = t_steps[i + 1];
  const double xn = x_next[idx];
  const double den = denoise(xn, F, idx, n, (float)t_next, sd, cfg_scale, use_cfg);
  const double d_prime = (xn - den) / t_next;
  x_next[idx] = x_hat[idx] + (t_next - t_hat) * (0.5 * d_cur[idx] + 0.5 * d_prime);
}

__global__ void sampler_advance_kernel(int32_t* step_idx) { *step_idx += 1; }

extern "C" int mdt_sampler_prep(const double* x, const double* t_steps, const int32_t* step_idx, int which, float* xin,
                                float* sigma_out, int B, int chw, int dup, float sigma_data, mdt_stream_t stream) {
  MDT_REQUIRE(x && t_steps && step_idx && xin && sigma_out, "sampler_prep: null pointer");
  MDT_REQUIRE(dup == 1 || dup == 2, "sampler_prep: dup must be 1 or 2");
  long n = (long)B * chw;
  hipLaunchKernelGGL(sampler_prep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, t_steps, step_idx,
                     which, xin, sigma_out, B, chw, dup, sigma_data);
  return mdt_check_launch("sampler_prep");
}

extern "C" int mdt_sampler_euler(const double* x_hat, const float* F, const double* t_steps, const int32_t* step_idx,
                                 float cfg_scale, int use_cfg, double* x_next, double* d_cur, int B, int chw,
                                 float sigma_data, mdt_stream_t stream) {
  MDT_REQUIRE(x_hat && F && t_steps && step_idx && x_next && d_cur, "sampler_euler: null pointer");
  long n = (long)B * chw;
  hipLaunchKernelGGL(sampler_euler_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x_hat, F, t_steps,
                     step_idx, cfg_scale, use_cfg, x_next, d_cur, n, sigma_data);
  return mdt_check_launch("sampler_euler");
}

extern "C" int mdt_sampler_heun(const double* x_hat, double* x_next, const float* F, const double* d_cur,
                                const double* t_steps, const int32_t* step_idx, float cfg_scale, int use_cfg, int B,
                                int chw, float sigma_data, mdt_stream_t stream) {
  MDT_REQUIRE(x_hat && x_next && F && d_cur && t_steps && step_idx, "sampler_heun: null pointer");
  long n = (long)B * chw;
  hipLaunchKernelGGL(sampler_heun_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x_hat, x_next, F, d_cur,
                     t_steps, step_idx, cfg_scale, use_cfg, n, sigma_data);
  return mdt_check_launch("sampler_heun");
}

extern "C" int mdt_sampler_advance(int32_t* step_idx, mdt_stream_t stream) {
  MDT_REQUIRE(step_idx, "sampler_advance: null pointer");
  hipLaunchKernelGGL(sampler_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_idx);
  return mdt_check_launch("sampler_advance");
}

// out[B] = F_uncond + s * (F_cond - F_uncond) with F = [cond; uncond] (models/maskdit.py:580-583)
__global__ void cfg_combine_kernel(const float* __restrict__ F, float s, float* __restrict__ out, long n) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float fu = F[n + idx];
  out[idx] = fu + s * (F[idx] - fu);
}

extern "C" int mdt_cfg_combine(const float* F, float cfg_scale, float* out, long n, mdt_stream_t stream) {
  MDT_REQUIRE(F && out && n > 0, "cfg_combine: bad arguments");
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, F, cfg_scale, out, n);
  return mdt_check_launch("cfg_combine");
}
