/*
 * maskdit_hip.h -- C ABI of libmaskdit_hip.so, the gfx950 (MI355X) kernel library behind
 * the MaskDiT training / sampling hot path.
 *
 * The reference (Anima-Lab/MaskDiT) has no FFI: its boundary is the Python object surface
 * (SURVEY.md section 8b).  Each entry point below therefore cites the reference Python code
 * whose arithmetic it replaces (paths relative to the reference repo).  All pointers are
 * DEVICE pointers unless stated; every call is asynchronous on `stream`, performs no
 * allocation and no host synchronisation (hipGraph-capturable), and returns 0 on success or
 * a negative code (mdt_last_error() gives the message).  bf16 = IEEE bfloat16 stored as
 * uint16_t; "f32" = float.
 */
#ifndef MASKDIT_HIP_H
#define MASKDIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mdt_stream_t; /* hipStream_t */
typedef uint16_t mdt_bf16;

const char* mdt_last_error(void);
/* ABI revision of this header: a binding must refuse a library whose mdt_version() differs (a changed signature would
 * otherwise be called with the wrong argument list). */
#define MDT_ABI_VERSION 4
int mdt_version(void);
/* Process-wide tuning knobs (benchmarking / A-B tests only; defaults are the product path).
 * "gemm_nt_variant": 0 = auto, 1 = force the 128x128-tile kernel, 2 = force the 256-row
 * phase-pipelined kernel wherever its shape constraints hold.
 * "gemm_tn_variant": 0 = auto, 1 = force the 128x128-tile weight-gradient kernel. */
int mdt_set_tuning(const char* key, int value);
/* Test support: fills the LDS of every CU with NaN bit patterns (0x7fc07fc0) so that the NEXT kernel on the stream
 * finds NaNs wherever it reads LDS it has not (yet) written -- how tests/test_00_kernels_gpu.py makes a missing
 * s_waitcnt / s_barrier in the LDS-DMA pipelines (the round-3 attention-backward race) deterministic.  `sink4` =
 * 4 writable device bytes (never written in practice). */
int mdt_lds_poison(void* sink4, mdt_stream_t stream);
/* Test / tool support for the wave-specialised GEMM form (csrc/gemm_nt8o.hip; mdt_set_tuning "nt8_overlap"): its waves
 * synchronise through LDS counters with BOUNDED spins; *abort_code != 0 means some wave gave up (the results of that
 * launch are garbage).  stats16 (may be NULL): stall-clock sums of its STATS launches (experiments build).  HOST
 * pointers; synchronises the device; reset != 0 clears both afterwards. */
int mdt_nt8o_report(unsigned* abort_code, unsigned long long* stats16, int reset);
/* ... and the per-tile time line of workgroup 0 of its last STATS launch: [role 0 MMA | 1 loader | 2 epilogue][tile < 32][start, done]
 * shader-clock stamps (192 values, HOST pointer; synchronises the device). */
int mdt_nt8o_stamps(unsigned long long* host192);

/* ---------------------------------------------------------------- GEMMs (MFMA bf16) ---- */

enum mdt_epilogue {
  MDT_EPI_BF16 = 0,     /* out = bf16(acc + bias)                                              */
  MDT_EPI_F32 = 1,      /* outf = acc + bias   (and out = bf16 of it when out != NULL)         */
  MDT_EPI_GELU = 2,     /* out = h = bf16(acc+bias); out2 = bf16(gelu_tanh(h))   (out may be NULL: */
  MDT_EPI_SILU = 3,     /* out = h; out2 = bf16(silu(h))                          inference)       */
  MDT_EPI_GATE_RES = 4, /* out = y = bf16(acc+bias); outf = res + gate[row/rows_per_sample]*y  (out may be NULL) */
  MDT_EPI_DGELU = 5,    /* out = bf16(acc * gelu_tanh'(aux))                                   */
  MDT_EPI_DSILU = 6     /* out = bf16(acc * silu'(aux))                                        */
};

/* C[M,N] = A[M,K] * B[N,K]^T (+bias) with a fused epilogue.  Replaces every nn.Linear on the
 * path: timm Attention.qkv/.proj, timm Mlp.fc1/.fc2 (models/maskdit.py:178,182), adaLN
 * Linear (:183-186), t_embedder (:34-38), y_embedder (:75), decoder_layer.linear (:203); the
 * GELU(tanh) (:181) and `x + gate * f(x)` (:190-191) are the fused epilogues.  Backward
 * data-gradients use the same kernel with the transposed weight shadow as B.
 * Requires N % 128 == 0, K % 64 == 0, 16-byte aligned rows; M arbitrary. */
typedef struct {
  const mdt_bf16* A; int lda;
  const mdt_bf16* B; int ldb;
  int M, N, K;
  const float* bias;
  int epi;
  mdt_bf16* out; int ldo;
  mdt_bf16* out2; int ldo2;
  float* outf; int ldof;
  const float* res; int ldres;
  const float* gate; int gate_ld; int rows_per_sample;
  const mdt_bf16* aux; int ldaux;
  int k_splits; /* MDT_EPI_F32 only (out must be NULL): > 1 splits the contraction over that many
                   workgroup groups; outf is cleared on the stream and accumulated with fp32 atomics.
                   For skinny problems with a huge K (the stacked adaLN data-gradient:
                   M = batch, N = D, K = all modulation outputs).  0/1 = off. */
  float* colsum; /* optional [N] fp32: colsum[n] += sum_m out[m, n] (the bf16-rounded values that are stored):
                    the bias gradient of the layer whose output-gradient this GEMM produces
                    (autograd of nn.Linear.bias), folded into the epilogue instead of a separate pass */
} mdt_gemm_nt_args;
int mdt_gemm_nt(const mdt_gemm_nt_args* a, mdt_stream_t stream);

/* C[N1,N2] += A[M,N1]^T * B[M,N2]  (f32 atomic accumulate, split over M).  Replaces the
 * autograd weight-gradient of every nn.Linear above (reference: torch autograd under
 * accelerator.backward, train.py:220).  Requires M % 64 == 0, row pitches padded to
 * multiples of 128 columns; only [n1_valid, n2_valid] is stored. */
typedef struct {
  const mdt_bf16* A; int lda;
  const mdt_bf16* B; int ldb;
  int M, N1, N2;
  float* C; int ldc;
  int n1_valid, n2_valid;
  int splits; /* 0 = auto */
  float* colsum_a; /* optional [N1] fp32: colsum_a[n] += sum_m A[m, n] -- the bias gradient of the layer whose weight
                      gradient this is (A = its output gradient), taken from the tiles the GEMM streams through LDS
                      anyway instead of a separate pass over A (mdt_colsum_bf16) */
} mdt_gemm_tn_args;
int mdt_gemm_tn(const mdt_gemm_tn_args* a, mdt_stream_t stream);

/* ---------------------------------------------------------------- attention ------------ */

/* softmax(q k^T / sqrt(hd)) v for packed qkv [B*L, 3*H*hd] (timm Attention, call site
 * models/maskdit.py:178).  out [B*L, H*hd] bf16, lse [B*H*L] f32 (log2 domain).
 * hd in {32, 64, 72, 80}; L % 64 == 0.  L_valid (0 = L): rows >= L_valid of every sample are padding
 * (a kept-token count rounded up to the 64-row tile): as KEYS they get zero probability, as queries their
 * outputs are don't-care. */
int mdt_attn_fwd(const mdt_bf16* qkv, mdt_bf16* out, float* lse, int B, int L, int H, int hd, int L_valid,
                 mdt_stream_t stream);
/* backward of the above: dqkv [B*L, 3*H*hd] from dout; delta is [B*H*L] f32 scratch.  With dout = 0 on the
 * padding rows, dqkv is exactly 0 there. */
int mdt_attn_bwd(const mdt_bf16* qkv, const mdt_bf16* out, const mdt_bf16* dout, const float* lse,
                 float* delta, mdt_bf16* dqkv, int B, int L, int H, int hd, int L_valid, mdt_stream_t stream);

/* ---------------------------------------------------------------- norm / modulate ------ */

/* xn = LayerNorm(x) * (1 + scale[b]) + shift[b]  (models/maskdit.py:19-20,177,190-191).
 * x f32 [M,D]; shift/scale f32 with row pitch mod_ld, sample = row / rows_per_sample;
 * xn bf16 [M,D]; stats[2*M] = (mean, rstd). */
int mdt_ln_modulate_fwd(const float* x, const float* shift, const float* scale, int mod_ld,
                        int rows_per_sample, mdt_bf16* xn, float* stats, int M, int D,
                        mdt_stream_t stream);
/* The residual add that feeds the LayerNorm, folded in (round 6): x = xres + gate[b] * y  (`x + gate * f(x)`,
 * models/maskdit.py:190-191, y = the bf16 branch output a plain mdt_gemm_nt stored), x is written (fp32: the next residual
 * / the backward's saved input), then xn, stats as mdt_ln_modulate_fwd of x.  Bit-identical to MDT_EPI_GATE_RES followed by
 * mdt_ln_modulate_fwd, 14 instead of 16 bytes per element. */
int mdt_ln_modulate_fwd_res(const float* xres, const mdt_bf16* y, const float* gate, int gate_ld, const float* shift,
                            const float* scale, int mod_ld, int rows_per_sample, float* x, mdt_bf16* xn, float* stats,
                            int M, int D, mdt_stream_t stream);
/* backward: dx (+)= dLN(dxn * (1+scale)); dshift[b] += sum_l dxn; dscale[b] += sum_l dxn*xhat. */
int mdt_ln_modulate_bwd(const mdt_bf16* dxn, const float* x, const float* stats, const float* scale,
                        int mod_ld, int rows_per_sample, float* dx, int accumulate, float* dshift,
                        float* dscale, int dmod_ld, int M, int D, mdt_stream_t stream);
/* ln_modulate_bwd fused with the backward of the residual gate that produced this LayerNorm's input
 * (x = x_prev + gate[b] * y, models/maskdit.py:190-191): after dx is final for a row, also
 * dys = bf16(gate[b] * dx), dgate[b] += sum_l dx * y, dbias += sum_rows dys -- i.e. mdt_gate_bwd without
 * re-reading dx. */
int mdt_ln_modulate_bwd_gate(const mdt_bf16* dxn, const float* x, const float* stats, const float* scale,
                             int mod_ld, int rows_per_sample, float* dx, int accumulate, float* dshift,
                             float* dscale, int dmod_ld, int M, int D, const mdt_bf16* y, const float* gate,
                             int gate_ld, mdt_bf16* dys, float* dgate, int dgate_ld, float* dbias,
                             mdt_stream_t stream);
/* backward of `x + gate * y` (models/maskdit.py:190-191): dys = bf16(gate[b] * dx),
 * dgate[b] += sum_l dx * y, dbias += sum_rows dys. */
int mdt_gate_bwd(const float* dx, const mdt_bf16* y, const float* gate, int mod_ld,
                 int rows_per_sample, mdt_bf16* dys, float* dgate, int dmod_ld, float* dbias, int M,
                 int D, mdt_stream_t stream);
/* out[n] += sum_m in[m,n]  (bias gradients). */
int mdt_colsum_bf16(const mdt_bf16* in, int ld, float* out, int M, int N, mdt_stream_t stream);

/* ---------------------------------------------------------------- masking -------------- */

/* get_mask (models/maskdit.py:88-113) from a noise tensor [B,T]: stable ascending argsort.
 * ids_* are int64 (reference dtype); ids32 = [B, 2T] int32 (shuffle | restore) for kernels. */
int mdt_mask_sort(const float* noise, int B, int T, int len_keep, int64_t* ids_shuffle,
                  int64_t* ids_restore, float* mask, int32_t* ids32, mdt_stream_t stream);

/* ---------------------------------------------------------------- embed / unmask / final  */

/* PatchEmbed (Conv2d k=s=p as a per-patch linear) + pos_embed + mask_out_token gather
 * (models/maskdit.py:116-127,278,475-483).  x [B,C,R,R] f32, scale[b] (c_in) optional,
 * W [D, C*p*p] f32, ids32 shuffle (NULL = all T tokens) -> out f32 [B, L, D]. */
int mdt_patch_embed_fwd(const float* x, const float* in_scale, const float* W, const float* bias,
                        const float* pos, const int32_t* ids, int ids_ld, float* out, int B, int C,
                        int R, int p, int L, int D, mdt_stream_t stream);
int mdt_patch_embed_bwd(const float* x, const float* in_scale, const float* dout, const int32_t* ids,
                        int ids_ld, float* dW, float* dbias, int B, int C, int R, int p, int L, int D,
                        mdt_stream_t stream);

/* timestep_embedding (models/maskdit.py:41-60): out bf16 [B,256] = [cos(t f), sin(t f)]. */
int mdt_timestep_embed(const float* t, mdt_bf16* out, int ld, int B, int dim, mdt_stream_t stream);
/* elementwise helpers on [rows, cols] */
int mdt_cast_f32_bf16(const float* in, int ldi, mdt_bf16* out, int ldo, int rows, int cols, int act,
                      mdt_stream_t stream); /* act: 0 none, 1 silu */
int mdt_add_f32(const float* a, const float* b, float* out, long n, mdt_stream_t stream);
int mdt_silu_bwd(const float* dy, const float* x, mdt_bf16* dx, long n, mdt_stream_t stream);

/* unmask_tokens + decoder_pos_embed (models/maskdit.py:157-163,543-545).
 * xdec bf16 [B,L_pitch,Dd] (L_pitch = 0 means L; L_pitch > L = encoder rows padded to the 64-row tile, the
 * padding rows are ignored forward and receive zero gradient); restore int32 [B,T] (NULL = identity, L == T);
 * out f32 [B,T,Dd]. */
int mdt_unmask_fwd(const mdt_bf16* xdec, const int32_t* restore, int ids_ld, const float* mask_token,
                   const float* pos, float* out, int B, int T, int L, int Dd, int L_pitch, mdt_stream_t stream);
int mdt_unmask_bwd(const float* dout, const int32_t* shuffle, int ids_ld, mdt_bf16* dxdec,
                   float* dmask_token, int B, int T, int L, int Dd, int L_pitch, mdt_stream_t stream);

/* FinalLayer + unpatchify (models/maskdit.py:216-234,411-424): x f32 [B*T, Dd] ->
 * LN-modulate -> Linear(Dd -> p*p*C) -> F [B,C,R,R] f32. */
int mdt_final_fwd(const float* x, const float* shift, const float* scale, int mod_ld, const float* W,
                  const float* bias, float* F, float* stats, int B, int T, int Dd, int C, int p,
                  mdt_stream_t stream);
int mdt_final_bwd(const float* dF, const float* x, const float* stats, const float* shift,
                  const float* scale, int mod_ld, const float* W, float* dx, float* dW, float* dbias,
                  float* dshift, float* dscale, int dmod_ld, int B, int T, int Dd, int C, int p,
                  mdt_stream_t stream);

/* ---------------------------------------------------------------- EDM precond / loss ---- */

/* EDMLoss noise draw + EDMPrecond coefficients (train_utils/loss.py:35-39,
 * models/maskdit.py:764-767): sigma = exp(rnd*P_std+P_mean); coef[i*B + b], i over (c_skip,
 * c_out, c_in, c_noise, weight, sigma, -, -) (8 x B floats); yn = y + noise*sigma; xin = c_in*yn. */
int mdt_edm_prep(const float* y, const float* rnd_normal, const float* noise, float* coef, float* yn,
                 float* xin, int B, int chw, float P_mean, float P_std, float sigma_data,
                 mdt_stream_t stream);
/* D = c_skip*yn + c_out*F; per-sample loss (train_utils/loss.py:44-52,88-101).  mask NULL =>
 * plain mean.  */
int mdt_edm_loss_fwd(const float* F, const float* yn, const float* y, const float* coef,
                     const float* mask, float mae_coef, float* D, float* loss, int B, int C, int R,
                     int p, mdt_stream_t stream);
int mdt_edm_loss_bwd(const float* dloss, const float* D, const float* yn, const float* y,
                     const float* coef, const float* mask, float mae_coef, float* dF, int B, int C,
                     int R, int p, mdt_stream_t stream);
/* Precond only (sampling / generic path): coef from sigma[b]; D = c_skip*x + c_out*F. */
int mdt_precond_coef(const float* sigma, float* coef, int B, float sigma_data, mdt_stream_t stream);
int mdt_scale_rows(const float* x, const float* coef, int coef_idx, float* out, int B, int chw,
                   mdt_stream_t stream);
int mdt_precond_out(const float* x, const float* F, const float* coef, float* D, int B, int chw,
                    mdt_stream_t stream);

/* utils.sample (utils.py:59-65): moments [B,2C,R,R] -> z = scale*(mean + exp(.5*clamp(logvar,-30,20))*randn),
 * chw = C*R*R, randn drawn by the caller (torch.randn_like(mean), same stream position as the reference). */
int mdt_sample_moments(const float* moments, const float* randn, float* z, int B, int chw, float scale,
                       mdt_stream_t stream);
/* class dropout (train.py:208-209): y[b,:] *= (u[b] >= p), u = torch.rand(B,1) drawn by the caller. */
int mdt_class_dropout(float* y, const float* u, float p, int B, int num_classes, mdt_stream_t stream);

/* ---------------------------------------------------------------- optimizer ------------- */

/* apex FusedAdam(adam_w_mode=True) step (train.py:141,226) fused with update_ema
 * (train_utils/helper.py:47-58) and the bf16 weight-shadow refresh, one pass over the flat
 * parameter arena.  grad is multiplied by grad_scale first.  ema / w16 may be NULL. */
int mdt_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, mdt_bf16* w16, long n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, float bc1,
                       float bc2, float ema_decay, float grad_scale, mdt_stream_t stream);
int mdt_ema_update(float* ema, const float* p, long n, float decay, mdt_stream_t stream);
/* batched [rows, cols] -> [cols, rows] bf16 transposes inside one arena; table = int64
 * (src_off, dst_off, rows, cols, tile_start) x n_entries (device), total_tiles over 64x64 tiles. */
int mdt_transpose_bf16_batched(const mdt_bf16* src, mdt_bf16* dst, const int64_t* table, int n_entries,
                               int total_tiles, mdt_stream_t stream);

/* ---------------------------------------------------------------- sampler --------------- */

/* edm_sampler (sample.py:30-66), S_churn = 0.  State fp64.  t_steps: device fp64 [N+1];
 * step_idx: device int32 advanced by mdt_sampler_advance so one captured hipGraph replays
 * every step.  cfg != 0 => F holds [cond; uncond] halves (models/maskdit.py:559-587). */
int mdt_sampler_prep(const double* x, const double* t_steps, const int32_t* step_idx, int which,
                     float* xin, float* sigma_out, int B, int chw, int dup, float sigma_data,
                     mdt_stream_t stream);
int mdt_sampler_euler(const double* x_hat, const float* F, const double* t_steps,
                      const int32_t* step_idx, float cfg_scale, int use_cfg, double* x_next,
                      double* d_cur, int B, int chw, float sigma_data, mdt_stream_t stream);
int mdt_sampler_heun(const double* x_hat, double* x_next, const float* F, const double* d_cur,
                     const double* t_steps, const int32_t* step_idx, float cfg_scale, int use_cfg,
                     int B, int chw, float sigma_data, mdt_stream_t stream);
int mdt_sampler_advance(int32_t* step_idx, mdt_stream_t stream);
/* classifier-free guidance combine on F = [cond; uncond] (models/maskdit.py:580-583), n = B*chw */
int mdt_cfg_combine(const float* F, float cfg_scale, float* out, long n, mdt_stream_t stream);

/* ---------------------------------------------------------------- fp32-faithful inference path ---- */

/* The reference's sampler evaluates its network in fp32 (sample.py:56 `net(x_hat.float(), ...)`; generate.py has no
 * autocast) and `train.py --no_amp` (train.py:39-46) runs fp32 too.  The entries below are that arithmetic on gfx950:
 * exact fp32 operands from the fp32 master weights, fp32 activations, the fp32-input matrix instruction
 * (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak).  csrc/f32path.hip. */
enum mdt_f32_epilogue {
  MDT_F32EPI_NONE = 0,     /* out = acc + bias                                                        */
  MDT_F32EPI_GELU = 1,     /* out = gelu_tanh(acc + bias)          (timm Mlp act, models/maskdit.py:181) */
  MDT_F32EPI_SILU = 2,     /* out = silu(acc + bias)               (TimestepEmbedder.mlp, :34-38)      */
  MDT_F32EPI_GATE_RES = 3  /* out = res + gate[row / rows_per_sample] * (acc + bias)   (:190-191); gate NULL = 1 */
};
/* out[z][M,N] = A[z][M,K] * B[z]^T (+ bias[N]) with an fp32 epilogue; z = b * heads + h runs over `batch` problems whose
 * operand bases are base + b * stride_b + h * stride_h (elements): nn.Linear (batch 1) and, with the packed qkv buffer,
 * timm Attention's q k^T (B = k rows, N = K-contiguous) and p v (b_kmajor = 1: B is [K, N] row-major, the v rows).
 * K % 4 == 0, 16-byte aligned operand rows / strides; M, N arbitrary. */
typedef struct {
  const float* A; long lda;
  const float* B; long ldb; int b_kmajor;
  int M, N, K;
  const float* bias;
  int epi;
  float* out; long ldo;
  const float* res; long ldres;
  const float* gate; long gate_ld; int rows_per_sample;
  int batch, heads; /* 0 = 1 */
  long a_stride_b, a_stride_h, b_stride_b, b_stride_h, o_stride_b, o_stride_h;
} mdt_gemm_f32_args;
int mdt_gemm_f32(const mdt_gemm_f32_args* a, mdt_stream_t stream);
/* s[r, :] = softmax(s[r, :n_valid] * scale), in place, columns >= n_valid set to 0 (timm Attention: softmax(q k^T * hd^-0.5)) */
int mdt_softmax_rows_f32(float* s, long R, int n, int n_valid, float scale, mdt_stream_t stream);
/* timm Attention (call site models/maskdit.py:178) in exact fp32 on the packed qkv buffer [B*L, 3*H*hd] -> out [B*L, H*hd].
 * One fused launch (K, V of a (sample, head) resident in LDS, scores in registers) for L in {64, 256}, hd in {32, 64, 72};
 * other shapes run q k^T -> softmax -> p v through `scores_ws`, which must then hold mdt_attn_f32_ws_floats() floats
 * (0 = not needed; scores_ws may be NULL). */
long mdt_attn_f32_ws_floats(int B, int L, int H, int hd);
int mdt_attn_f32(const float* qkv, float* out, float* scores_ws, int B, int L, int H, int hd, mdt_stream_t stream);
/* mdt_ln_modulate_fwd with an fp32 result (models/maskdit.py:19-20,177; eps 1e-6) */
int mdt_ln_modulate_f32(const float* x, const float* shift, const float* scale, int mod_ld, int rows_per_sample,
                        float* xn, int M, int D, mdt_stream_t stream);
/* mdt_timestep_embed with an fp32 result (models/maskdit.py:41-60) */
int mdt_timestep_embed_f32(const float* t, float* out, int ld, int B, int dim, mdt_stream_t stream);
/* out = silu(in)  (the SiLU in front of every adaLN Linear, models/maskdit.py:183-186) */
int mdt_silu_f32(const float* in, float* out, long n, mdt_stream_t stream);
/* out[(b, j), :] = in[(b, j), :] + rows[j, :], j = row % T: `x + decoder_pos_embed` (models/maskdit.py:545); D % 4 == 0 */
int mdt_add_rows_f32(const float* in, const float* rows, float* out, long n_rows, int T, int D, mdt_stream_t stream);

/* ---------------------------------------------------------------- VAE decoder glue ------ */

/* The KL-autoencoder decoder that follows the sampler (autoencoder.py:306-410 Decoder, :449-453 decode; call sites
 * sample.py:248,273-284).  Activations are NHWC fp32 [B*H*W, C]; mdt_gn_im2col applies the GroupNorm / swish in front of a
 * convolution and writes bf16: the normalised activation itself (ksize 1) for mdt_conv3x3_nhwc / the 1x1 convolutions, or an
 * im2col matrix for mdt_gemm_nt where the channel count is not a multiple of 64 (conv_in). */
/* sums[b, g, 0..1] = (sum, sum of squares) over the H*W*(C/groups) elements of group g (autoencoder.py:35-36
 * Normalize = GroupNorm(32, eps 1e-6, affine)); cleared on the stream by the call. */
int mdt_gn_stats(const float* x, float* sums, int B, int HW, int C, int groups, mdt_stream_t stream);
/* col[(b, yo, xo), (ky, kx, c)] = swish?(gamma * (x - mean) * rstd + beta) at input pixel ((yo+ky-1) >> up, (xo+kx-1) >> up),
 * zero outside the (up-sampled) image and in the padding columns [k*k*C, Kp): ResnetBlock.norm1/2 + nonlinearity +
 * conv padding (autoencoder.py:118-129), Upsample.interpolate(nearest, 2x) (:49), AttnBlock.norm (:176).  sums = NULL:
 * no normalisation (conv_in, the up-sampling convolutions, nin_shortcut). */
int mdt_gn_im2col(const float* x, const float* sums, const float* gamma, const float* beta, mdt_bf16* col, int B, int H,
                  int W, int C, int groups, int ksize, int upsample, int swish, int Kp, mdt_stream_t stream);
/* out[(b, y, x), n] = bias[n] + sum_{ky, kx, c} act[b, (y + ky - 1) >> up, (x + kx - 1) >> up, c] * W[n, (ky * 3 + kx) * C + c]
 * (zero outside the Ho x Ho image, Ho = Hi << up): nn.Conv2d(C, Cout, 3, padding = 1) of ResnetBlock.conv1/2, conv_out and
 * Upsample.conv behind interpolate(nearest, 2x) (autoencoder.py:35-52, 78-140, 306-410) as an IMPLICIT GEMM -- the bf16
 * MFMA kernel of mdt_gemm_nt gathers its A operand straight from the NHWC activation; no im2col matrix.  `act`: bf16
 * [B, Hi, Hi, C] (what mdt_gn_im2col writes with ksize 1: GroupNorm + swish applied), C % 128 == 0, Hi a power of two,
 * and THE 256 BYTES IN FRONT OF `act` MUST BE ZERO (the padding taps read them).  W: bf16 [Np, 9 C], Np % 128 == 0;
 * out: fp32 [B * Ho * Ho, ldo].  Fused epilogue options (round 4; both may be NULL): `res` fp32 [B * Ho * Ho, ldo] is added
 * to the result (ResnetBlock's `x + h`, autoencoder.py:129); `gn_sums` fp32 [B, 32, 2] (cleared by the CALLER) receives
 * += (sum, sum of squares) per sample and GroupNorm group of the values stored -- the statistics of the next layer's
 * Normalize (autoencoder.py:35-36) without a pass of mdt_gn_stats; needs Np == the real channel count and Ho * Ho % 128 == 0. */
int mdt_conv3x3_nhwc(const mdt_bf16* act, int B, int Hi, int C, int up, const mdt_bf16* W, const float* bias,
                     const float* res, float* out, int ldo, int Np, float* gn_sums, int gn_groups, mdt_stream_t stream);
/* out[r, :] = softmax(in[r, :] * scale) as bf16 (AttnBlock, autoencoder.py:188-190) */
int mdt_softmax_rows(const float* in, mdt_bf16* out, int R, int n, float scale, mdt_stream_t stream);
/* y[b, p, :] = W (z[b, :, p] / scale_factor) + bias: FrozenAutoencoderKL.decode's rescale + post_quant_conv
 * (autoencoder.py:449-451), NCHW fp32 in, NHWC fp32 out (4 channels) */
int mdt_vae_prologue(const float* z, const float* w, const float* bias, float* y, int B, int HW, float scale_factor,
                     mdt_stream_t stream);
/* img[b, c, p] = in[(b, p), c], c < Cout: the decoder output back in the reference's NCHW layout */
int mdt_vae_epilogue(const float* in, int ld, float* img, int B, int HW, int Cout, mdt_stream_t stream);

/* hipGraph helpers (stream capture of a sequence of the calls above). */
int mdt_graph_begin(mdt_stream_t stream);
int mdt_graph_end(mdt_stream_t stream, void** graph_exec_out);
int mdt_graph_launch(void* graph_exec, mdt_stream_t stream);
int mdt_graph_destroy(void* graph_exec);

/* timing helper: HIP events on an arbitrary stream (bench.py roofline leg). */
int mdt_event_create(void** ev);
int mdt_event_record(void* ev, mdt_stream_t stream);
int mdt_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */
int mdt_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif
