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
enum { MDT_TUNE_GEMM_NT_VARIANT = 0, MDT_TUNE_GEMM_TN_VARIANT = 1, MDT_TUNE_NT8_SKIP_EPILOGUE = 2, MDT_TUNE_NT8_STAGGER = 3, MDT_TUNE_NT8_GROUP_M = 4, MDT_TUNE_ATTN_QF = 5, MDT_TUNE_NT8_NF3 = 6, MDT_TUNE_NT8_MAX_CUS = 7, MDT_TUNE_ATTN_SP = 8, MDT_TUNE_NT8_TRICKLE = 9, MDT_TUNE_TN8_WIDE = 10, MDT_TUNE_TN8_DBG = 11, MDT_TUNE_LN_GATE_ROWWISE = 12, MDT_TUNE_ATTN_DBG = 13, MDT_TUNE_NT8_SCHED = 14, MDT_TUNE_NT8_OVERLAP = 15, MDT_TUNE_COUNT = 16 };
int mdt_get_tuning_int(int key);
// Timing-decomposition switches that make a kernel skip part of its work (RESULTS ARE GARBAGE) exist only in the
// experiments build (`make experiments` -> libmaskdit_hip_exp.so, -DMDT_EXPERIMENTS; tools/* load it through
// MASKDIT_HIP_LIB).  In the product library the conditions below are the constant 0 and mdt_set_tuning refuses the keys.
#ifdef MDT_EXPERIMENTS
#define MDT_EXP(cond) (cond)
#else
#define MDT_EXP(cond) 0
#endif

#define MDT_REQUIRE(cond, msg)            \
  do {                                    \
    if (!(cond)) {                        \
      mdt_set_error(msg);                 \
      return MDT_ERR_ARG;                 \
    }                                     \
  } while (0)

// identity that hipcc cannot see through: keeps (uniform base) + (32-bit lane offset) address expressions in the
// saddr + voffset form instead of one 64-bit vector address per access
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

// the same for a wave-uniform pointer: pins it in a scalar register pair
__device__ __forceinline__ const char* sopaque(const char* p) {
  asm volatile("" : "+s"(p));
  return p;
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
// 16 bytes per lane HBM / L2 -> LDS without a register round trip (LDS address = wave-uniform base + 16 * lane)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gsrc), LDS_PTR(lds_dst), 16, 0, 0);
}

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

// Activations in "sigmoid form" with the hardware exp2 / reciprocal (1 ulp each): ~7 VALU per GELU instead of
// ~35 with an IEEE division.  Results are rounded to bf16 by every caller.
//   gelu_tanh(x) = 0.5 x (1 + tanh(u)) = x * sigmoid(2u),  u = sqrt(2/pi) (x + 0.044715 x^3)   (models/maskdit.py:181)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float gelu_sig(float x) {
  // sigmoid(2u) = 1 / (1 + 2^(-2 u log2 e))
  const float c0 = -2.f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
  const float x2 = x * x;
  return fast_rcp(1.f + fast_exp2(x * (c0 + c1 * x2)));
}
__device__ __forceinline__ float gelu_tanh(float x) { return x * gelu_sig(x); }
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  // d/dx [x s(x)] = s (1 + x (1 - s) 2u'),  2u' = 2 sqrt(2/pi) (1 + 3 * 0.044715 x^2)
  const float d0 = 2.f * 0.7978845608028654f, d1 = d0 * 3.f * 0.044715f;
  const float s = gelu_sig(x);
  return s * (1.f + x * (1.f - s) * (d0 + d1 * x * x));
}
__device__ __forceinline__ float sigmoid_fast(float x) { return fast_rcp(1.f + fast_exp2(-1.4426950408889634f * x)); }
__device__ __forceinline__ float silu(float x) { return x * sigmoid_fast(x); }
__device__ __forceinline__ float silu_grad(float x) {
  const float s = sigmoid_fast(x);
  return s * (1.f + x * (1.f - s));
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
