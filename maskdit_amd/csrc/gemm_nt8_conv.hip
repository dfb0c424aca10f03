// Implicit-GEMM 3x3 convolution on NHWC bf16 activations: the fp32-output class of gemm_nt8 (gemm_nt8_impl.h) with the
// A operand GATHERED by the LDS-DMA refills -- row m = output pixel (b, y, x), K index = (tap, channel); a K-tile of 64
// channels of one tap is 128 contiguous bytes of the activation at the tap-shifted pixel, or of the zero line in front of
// the activation where the tap falls outside the image; nearest-neighbour 2x up-sampling is a shift of the source
// coordinates.  Replaces the materialised im2col matrix (9x the activation bytes per convolution) of rounds 1-2 in the VAE
// decoder (reference: autoencoder.py:35-52 Upsample, :78-140 ResnetBlock, :306-410 Decoder -- nn.Conv2d(k = 3, padding = 1)).
#define NT8_CLASS 1
#include "gemm_nt8_impl.h"

template __global__ void gemm_nt8_kernel<2, 2, 1, NT8_DEFAULT_SCHED | 4096>(NTParams);
template __global__ void gemm_nt8_kernel<4, 2, 1, NT8_DEFAULT_SCHED | 4096>(NTParams);
// the fused epilogue (skip connection + GroupNorm sums, round 4) exists for 128-column tiles only: in the 256-wide kernel
// (256 VGPRs) it spilled 38-48 registers -- tools/check_waits.py flagged the scratch traffic inside the counted
// cross-tile pair -- so requests with `res` / `gn_sums` run 128-wide tiles
template __global__ void gemm_nt8_kernel<2, 2, 1, NT8_DEFAULT_SCHED | 4096 | 8192>(NTParams);

extern "C" int mdt_conv3x3_nhwc(const mdt_bf16* act, int B, int Hi, int C, int up, const mdt_bf16* W, const float* bias,
                                const float* res, float* out, int ldo, int Np, float* gn_sums, int gn_groups, mdt_stream_t stream) {
  MDT_REQUIRE(act && W && out, "conv3x3_nhwc: null operand");
  MDT_REQUIRE(B > 0 && Hi >= 8 && (Hi & (Hi - 1)) == 0, "conv3x3_nhwc: the input size must be a power of two >= 8");
  MDT_REQUIRE(up == 0 || up == 1, "conv3x3_nhwc: up must be 0 or 1 (nearest-neighbour 2x)");
  MDT_REQUIRE(C >= 128 && C % 128 == 0, "conv3x3_nhwc: channels must be a multiple of 128 (K = 9 C in whole K-tile pairs)");
  MDT_REQUIRE(Np % 128 == 0 && ldo >= Np && ldo % 4 == 0, "conv3x3_nhwc: output columns must be padded to a multiple of 128");
  MDT_REQUIRE((((uintptr_t)act | (uintptr_t)W | (uintptr_t)out) & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0),
              "conv3x3_nhwc: operands must be 16-byte aligned");
  const int Ho = Hi << up;
  MDT_REQUIRE(Ho <= 2048 && B < 256, "conv3x3_nhwc: the packed pixel coordinates hold 12 + 12 + 8 bits");
  MDT_REQUIRE((long)B * Hi * Hi * C * 2 + 256 < (1L << 32), "conv3x3_nhwc: the activation must stay below 4 GB (32-bit source offsets)");
  const long M = (long)B * Ho * Ho;
  MDT_REQUIRE(M % 256 == 0, "conv3x3_nhwc: B * Ho * Ho must be a multiple of 256");
  int cpg_log2 = 0;
  if (gn_sums) {
    MDT_REQUIRE(gn_groups > 0 && Np % gn_groups == 0, "conv3x3_nhwc: gn_sums needs every output column to be a real channel (Np % groups == 0)");
    const int cpg = Np / gn_groups;
    MDT_REQUIRE(cpg >= 4 && (cpg & (cpg - 1)) == 0, "conv3x3_nhwc: channels per GroupNorm group must be a power of two >= 4");
    MDT_REQUIRE(((long)Ho * Ho) % 128 == 0, "conv3x3_nhwc: gn_sums needs Ho * Ho % 128 == 0 (a wave's 128 rows in one sample)");
    MDT_REQUIRE(gn_groups == 32, "conv3x3_nhwc: gn_sums is laid out [B, 32, 2]");
    while ((1 << cpg_log2) < cpg) ++cpg_log2;
  }
  MDT_REQUIRE(!res || (((uintptr_t)res & 15) == 0), "conv3x3_nhwc: res must be 16-byte aligned");
  NTParams p = {};
  p.res = res; p.ldres = ldo;
  p.gn_sums = gn_sums; p.gn_cpg_log2 = cpg_log2;
  p.A = (const bf16*)((const char*)act - 256);  // offset 0 .. 255 = the zero line the caller keeps in front of the activation
  p.lda = 0;
  p.B = (const bf16*)W; p.ldb = 9 * C;
  p.M = (int)M; p.N = Np; p.K = 9 * C;
  p.bias = bias; p.epi = MDT_EPI_F32;
  p.outf = out; p.ldof = ldo;
  p.k_splits = 1;
  p.group_m = mdt_get_tuning_int(MDT_TUNE_NT8_GROUP_M);
  int hl = 0;
  while ((1 << hl) < Ho) ++hl;
  p.conv_ho_log2 = hl; p.conv_up = up; p.conv_c = C;
  const bool fuse = res != nullptr || gn_sums != nullptr;
  const int nf = (Np % 256 == 0 && !fuse) ? 4 : 2;
  const int ntiles = (p.M / 256) * (Np / (64 * nf));
  const int slots = nt8_num_cus();
  const int grid = ntiles < slots ? ntiles : slots;
  if (fuse) hipLaunchKernelGGL((gemm_nt8_kernel<2, 2, 1, NT8_DEFAULT_SCHED | 4096 | 8192>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
  else if (nf == 4) hipLaunchKernelGGL((gemm_nt8_kernel<4, 2, 1, NT8_DEFAULT_SCHED | 4096>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((gemm_nt8_kernel<2, 2, 1, NT8_DEFAULT_SCHED | 4096>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
  return mdt_check_launch("conv3x3_nhwc");
}
