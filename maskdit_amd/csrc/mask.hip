// get_mask (models/maskdit.py:88-113) from a supplied noise tensor: per-row stable ascending
// argsort, its inverse permutation, and the binary mask.  One workgroup per row; the whole
// row (T <= 1024, power of two) is sorted in LDS with a bitonic network on 64-bit keys
// (float bits << 32 | index): noise is in [0,1) so the IEEE bit pattern orders like the value,
// and the index in the low word makes the order total => identical to a *stable* argsort
// (the tie rule the oracle uses; torch.argsort itself leaves ties unspecified).
#include "common.h"
#include "../../include/maskdit_hip.h"

__global__ __launch_bounds__(256) void mask_sort_kernel(const float* __restrict__ noise, int T, int len_keep,
                                                        int64_t* __restrict__ ids_shuffle, int64_t* __restrict__ ids_restore,
                                                        float* __restrict__ mask, int32_t* __restrict__ ids32) {
  __shared__ unsigned long long keys[1024];
  const int b = blockIdx.x;
  const float* nr = noise + (long)b * T;
  for (int i = threadIdx.x; i < T; i += 256) {
    unsigned int bits = __float_as_uint(nr[i]);
    // total order for any finite float (not only [0,1)): flip sign bit / all bits
    bits = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
    keys[i] = ((unsigned long long)bits << 32) | (unsigned int)i;
  }
  __syncthreads();
  for (int k = 2; k <= T; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < T; i += 256) {
        int ixj = i ^ j;
        if (ixj > i) {
          unsigned long long a = keys[i], c = keys[ixj];
          bool up = ((i & k) == 0);
          if ((a > c) == up) {
            keys[i] = c;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int r = threadIdx.x; r < T; r += 256) {
    int idx = (int)(keys[r] & 0xffffffffu);
    if (ids_shuffle) ids_shuffle[(long)b * T + r] = idx;
    if (ids_restore) ids_restore[(long)b * T + idx] = r;
    if (mask) mask[(long)b * T + idx] = (r >= len_keep) ? 1.f : 0.f;
    if (ids32) {
      ids32[(long)b * 2 * T + r] = idx;
      ids32[(long)b * 2 * T + T + idx] = r;
    }
  }
}

extern "C" int mdt_mask_sort(const float* noise, int B, int T, int len_keep, int64_t* ids_shuffle,
                             int64_t* ids_restore, float* mask, int32_t* ids32, mdt_stream_t stream) {
  MDT_REQUIRE(noise, "mask_sort: null noise");
  MDT_REQUIRE(B > 0 && T >= 2 && T <= 1024 && (T & (T - 1)) == 0, "mask_sort: T must be a power of two in [2, 1024]");
  MDT_REQUIRE(len_keep >= 0 && len_keep <= T, "mask_sort: bad len_keep");
  hipLaunchKernelGGL(mask_sort_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, noise, T, len_keep, ids_shuffle,
                     ids_restore, mask, ids32);
  return mdt_check_launch("mask_sort");
}
