// get_mask (models/maskdit.py:88-113) from a supplied noise tensor: per-row stable ascending argsort, its inverse
// permutation, and the binary mask.  Keys are 64-bit (order-preserving float bits << 32 | index): the index in the low
// word makes the order total => identical to a *stable* argsort (the tie rule the oracle uses; torch.argsort itself
// leaves ties unspecified), and ANY sorting network produces the same result bit for bit.
//
// Round 4: ONE WAVEFRONT PER ROW, the row in REGISTERS (T / 64 keys per lane, key e = r * 64 + lane), a bitonic network
// whose cross-lane exchanges (partner distance < 64) are wavefront shuffles (__shfl_xor -> DPP / ds_bpermute: no LDS
// storage, no barrier) and whose longer-distance exchanges are compare-swaps between a lane's own registers -- what
// BASELINE.json's north_star sketches ("wavefront shuffles for the mask-gather").  Rounds 1-3 sorted each row in LDS with
// one 256-thread workgroup and 36-55 __syncthreads(); that kernel remains for rows shorter than a wavefront (T < 64).
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

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int j) {
  const unsigned lo = __shfl_xor((unsigned)v, j, 64), hi = __shfl_xor((unsigned)(v >> 32), j, 64);
  return ((unsigned long long)hi << 32) | lo;
}

// KPL keys per lane; key e = r * 64 + lane (so that every global access of a row is a coalesced 64-lane stripe)
template <int KPL>
__global__ __launch_bounds__(256) void mask_sort_wave_kernel(const float* __restrict__ noise, int B, int len_keep,
                                                             int64_t* __restrict__ ids_shuffle, int64_t* __restrict__ ids_restore,
                                                             float* __restrict__ mask, int32_t* __restrict__ ids32) {
  constexpr int T = 64 * KPL;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);  // four rows (wavefronts) per workgroup
  if (b >= B) return;
  const float* nr = noise + (long)b * T;
  unsigned long long key[KPL];
#pragma unroll
  for (int r = 0; r < KPL; ++r) {
    unsigned bits = __float_as_uint(nr[r * 64 + lane]);
    bits = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);  // total order for any finite float
    key[r] = ((unsigned long long)bits << 32) | (unsigned)(r * 64 + lane);
  }
#pragma unroll
  for (int k = 2; k <= T; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {  // partner = another register of this lane
        const int jr = j >> 6;
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
          if ((r & jr) == 0) {
            const bool up = (((r * 64) & k) == 0);  // k >= 128 here: the direction bit is a register-index bit
            const unsigned long long a = key[r], c = key[r | jr];
            const bool sw = (a > c) == up;
            key[r] = sw ? c : a;
            key[r | jr] = sw ? a : c;
          }
        }
      } else {        // partner = the same register of lane ^ j: a wavefront shuffle
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
          const unsigned long long a = key[r], c = shfl_xor_u64(a, j);
          const bool up = (((r * 64 + lane) & k) == 0), lower = (lane & j) == 0;
          const unsigned long long mn = a < c ? a : c, mx = a < c ? c : a;
          key[r] = (lower == up) ? mn : mx;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < KPL; ++r) {
    const int rank = r * 64 + lane;
    const int idx = (int)(key[r] & 0xffffffffu);
    if (ids_shuffle) ids_shuffle[(long)b * T + rank] = idx;
    if (ids_restore) ids_restore[(long)b * T + idx] = rank;
    if (mask) mask[(long)b * T + idx] = (rank >= len_keep) ? 1.f : 0.f;
    if (ids32) {
      ids32[(long)b * 2 * T + rank] = idx;
      ids32[(long)b * 2 * T + T + idx] = rank;
    }
  }
}

extern "C" int mdt_mask_sort(const float* noise, int B, int T, int len_keep, int64_t* ids_shuffle,
                             int64_t* ids_restore, float* mask, int32_t* ids32, mdt_stream_t stream) {
  MDT_REQUIRE(noise, "mask_sort: null noise");
  MDT_REQUIRE(B > 0 && T >= 2 && T <= 1024 && (T & (T - 1)) == 0, "mask_sort: T must be a power of two in [2, 1024]");
  MDT_REQUIRE(len_keep >= 0 && len_keep <= T, "mask_sort: bad len_keep");
  const dim3 grid((B + 3) / 4), blk(256);
#define MASK_SORT_WAVE(KPL) hipLaunchKernelGGL(mask_sort_wave_kernel<KPL>, grid, blk, 0, (hipStream_t)stream, noise, B, len_keep, \
                                               ids_shuffle, ids_restore, mask, ids32)
  switch (T) {
    case 64: MASK_SORT_WAVE(1); break;
    case 128: MASK_SORT_WAVE(2); break;
    case 256: MASK_SORT_WAVE(4); break;
    case 512: MASK_SORT_WAVE(8); break;
    case 1024: MASK_SORT_WAVE(16); break;
    default:  // rows shorter than a wavefront: the LDS network
      hipLaunchKernelGGL(mask_sort_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, noise, T, len_keep, ids_shuffle,
                         ids_restore, mask, ids32);
  }
#undef MASK_SORT_WAVE
  return mdt_check_launch("mask_sort");
}
