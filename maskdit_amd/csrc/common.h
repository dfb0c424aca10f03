// Shared device helpers for the MaskDiT gfx950 kernels (wave64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MDT_OK 0
#define MDT_ERR_ARG (-1)
#define MDT_ERR_LAUNCH (-2)

// error plumbing (capi.hip owns the storage)
void mdt_set_error(const char* msg);
int mdt_check_launch(const char* what);

// tuning knobs (capi.hip; set through mdt_set_tuning)
enum { MDT_TUNE_GEMM_NT_VARIANT = 0, MDT_TUNE_GEMM_TN_VARIANT = 1, MDT_TUNE_NT8_SKIP_EPILOGUE = 2, MDT_TUNE_NT8_STAGGER = 3, MDT_TUNE_NT8_GROUP_M = 4, MDT_TUNE_ATTN_QF = 5, MDT_TUNE_COUNT = 8 };
int mdt_get_tuning_int(int key);

#define MDT_REQUIRE(cond, msg)            \
  do {                                    \
    if (!(cond)) {                        \
      mdt_set_error(msg);                 \
      return MDT_ERR_ARG;                 \
    }                                     \
  } while (0)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ds_read_b64_tr_b16: within each 16-lane group the lanes supply 16 8-byte pieces forming a
// [4 rows][16 cols] bf16 block (row r from lanes 4r..4r+3); lane i receives column i
// (block[0..3][i]).  `p` must be 8-byte aligned.
__device__ __forceinline__ bf16x4 lds_tr_read(const void* p) {
  short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(p));
  return __builtin_bit_cast(bf16x4, v);
}

__device__ __forceinline__ bf16x8 cat4(bf16x4 lo, bf16x4 hi) {
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_tanh(float x) {
  // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))   (models/maskdit.py:181)
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  float t = 1.f - 2.f / (1.f + __expf(2.f * u));  // tanh(u)
  return 0.5f * x * (1.f + t);
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float x2 = x * x;
  float u = k0 * (x + k1 * x * x2);
  float t = 1.f - 2.f / (1.f + __expf(2.f * u));
  float du = k0 * (1.f + 3.f * k1 * x2);
  return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
}
__device__ __forceinline__ float silu(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float silu_grad(float x) {
  float s = 1.f / (1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
