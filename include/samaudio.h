/* libsamaudio_hip.so - C ABI of the MI355X-native SAMAudio.separate() hot path.
 *
 * The reference (facebookresearch/sam-audio) is pure Python and has no FFI / operator interface; its
 * boundary is the Python class API of `sam_audio` (sam_audio/__init__.py:3-4).  This header is the
 * boundary a maintainer would bind with ctypes: plain pointers, sizes and a
 * hipStream_t - no torch types.  Each entry point cites the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the library never allocates or frees device memory: weights are borrowed (they must outlive the
 *     context), scratch comes from one caller-provided workspace;
 *   - every call is asynchronous on `stream` (pass the caller's current HIP stream);
 *   - return value: 0 = ok, negative = error, message via samaudio_last_error();
 *   - one context per (process, GPU, stream); not thread-safe.
 */
#ifndef SAMAUDIO_H_
#define SAMAUDIO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct samaudio_ctx samaudio_ctx;
typedef void* samaudio_stream; /* hipStream_t */

enum { SAMAUDIO_F32 = 0, SAMAUDIO_BF16 = 1 };           /* compute precision == dtype of GEMM operands */
enum { SAMAUDIO_DT_F32 = 0, SAMAUDIO_DT_BF16 = 1, SAMAUDIO_DT_I64 = 2, SAMAUDIO_DT_U8 = 3 };
enum { SAMAUDIO_ODE_EULER = 0, SAMAUDIO_ODE_MIDPOINT = 1 };

enum {
  SAMAUDIO_OK = 0,
  SAMAUDIO_ERR_ARG = -1,      /* bad argument / shape mismatch      (reference: AssertionError) */
  SAMAUDIO_ERR_WEIGHT = -2,   /* missing / mis-shaped weight tensor (reference: RuntimeError, model.py:356-359) */
  SAMAUDIO_ERR_WORKSPACE = -3,/* workspace missing or too small */
  SAMAUDIO_ERR_HIP = -4,      /* HIP runtime error */
  SAMAUDIO_ERR_STATE = -5     /* call order violated (e.g. solve before prepare) */
};

/* Hyper-parameters: reference sam_audio/model/config.py:86-135 (TransformerConfig), :204-231 (SAMAudioConfig),
 * :10-41 (DACVAEConfig). */
typedef struct {
  int32_t precision;        /* SAMAUDIO_F32 | SAMAUDIO_BF16 */
  int32_t dim, n_heads, n_layers, ffn_hidden;
  int32_t latent_channels;  /* 2 * codebook_dim = 256: ODE state width and DiT out_channels */
  int32_t text_dim;         /* 768  */
  int32_t video_dim;        /* 1024 */
  int32_t freq_dim;         /* 256  */
  int32_t anchor_dim;       /* 128  */
  int32_t anchor_vocab;     /* num_anchors + 1 = 4 */
  int32_t max_positions;    /* RoPE table rows */
  float norm_eps;
  /* DAC-VAE */
  int32_t codec_dim;        /* 128  */
  int32_t codec_latent;     /* 1024 */
  int32_t enc_dim;          /* 64   */
  int32_t dec_dim;          /* 1536 */
  int32_t enc_rates[4];     /* 2,8,10,12 */
  int32_t dec_rates[4];     /* 12,10,8,2 */
} samaudio_config;

const char* samaudio_last_error(void);
const char* samaudio_version(void);

int samaudio_create(const samaudio_config* cfg, samaudio_ctx** out);
void samaudio_destroy(samaudio_ctx* ctx);

/* Register one (re-laid-out) weight tensor by engine name; the pointer is borrowed.  The Python loader
 * (sam_audio_amd/weights.py) documents the mapping from the reference state_dict keys
 * (transformer.layers.{i}.attention.wq.weight, ...) to engine names. */
int samaudio_set_tensor(samaudio_ctx* ctx, const char* name, const void* data, int dtype, int ndim,
                        const int64_t* shape);
/* Check that the DiT (what=0) or codec (what=1) weight set is complete and shaped for the config. */
int samaudio_finalize(samaudio_ctx* ctx, int what);

/* Scratch.  `codec_items`: waveforms processed per codec pass (0 = no codec use), `samples`: padded
 * samples per waveform. */
size_t samaudio_workspace_bytes(samaudio_ctx* ctx, int rows, int frames, int text_len, int codec_items,
                                int64_t samples);
int samaudio_set_workspace(samaudio_ctx* ctx, void* workspace, size_t bytes);

/* ---- hot path ------------------------------------------------------------------------------------ */

/* Everything of SAMAudio.forward that does not depend on (noisy_audio, time): the audio_features / video /
 * anchor terms of align_inputs (model.py:108-128, align.py:30-50, model.py:54-65) and memory_proj(text)
 * (model.py:171).  rows = B * candidates (conditioning already repeated, model.py:193-229).
 *   audio_features [rows, frames, 256] f32      text [rows, text_len, text_dim] f32 or NULL
 *   text_mask      [rows, text_len] u8 or NULL  video [rows, frames, video_dim] f32 (channels-last) or NULL
 *   anchor_ids     [rows, n_ids] i64 or NULL    anchor_alignment [rows, frames] i64
 *   audio_pad_mask [rows, frames] u8 or NULL (1 = valid frame)                                          */
int samaudio_prepare(samaudio_ctx* ctx, int rows, int frames, int text_len, const float* audio_features,
                     const float* text, const uint8_t* text_mask, const float* video, const int64_t* anchor_ids,
                     int n_ids, const int64_t* anchor_alignment, const uint8_t* audio_pad_mask,
                     samaudio_stream stream);

/* One ODE function evaluation = SAMAudio.forward (model.py:130-180) -> DiT.forward (transformer.py:473-524).
 *   noisy [rows, frames, 256] f32, time [n_time] f32 with n_time in {1, rows}, out [rows, frames, 256] f32 */
int samaudio_forward(samaudio_ctx* ctx, const float* noisy, const float* time, int n_time, float* out,
                     samaudio_stream stream);

/* Fixed-grid ODE solve, replaces torchdiffeq.odeint at model.py:285-290 (method "midpoint" | "euler").
 * grid_host: n_grid increasing time points on the HOST (t0 .. t1); state [rows, frames, 256] f32 is updated
 * in place from the noise to states[-1]. */
int samaudio_ode_solve(samaudio_ctx* ctx, float* state, int method, const float* grid_host, int n_grid,
                       samaudio_stream stream);

/* DAC-VAE.  encode (codec.py:65-70): wav [items, samples] f32, samples % hop == 0 -> mean latent
 * [items, samples/hop, codec_dim] f32 (channels-last).  decode (codec.py:86-89):
 * latent [items, frames, codec_dim] f32 (channels-last) -> wav [items, frames*hop] f32. */
int samaudio_codec_encode(samaudio_ctx* ctx, const float* wav, int items, int64_t samples, float* latent,
                          samaudio_stream stream);
int samaudio_codec_decode(samaudio_ctx* ctx, const float* latent, int items, int frames, float* wav,
                          samaudio_stream stream);

/* ---- measurement ----------------------------------------------------------------------------------- */

/* Live per-kernel timing for bench.py's roofline leg (the reference has no counterpart: it publishes no
 * throughput numbers).  Between begin and end every GEMM launch on the hot path is
 * bracketed by a hipEvent pair recorded on the launch stream; end synchronises and returns, per GEMM tile
 * variant, the number of launches, the summed ALGORITHMIC flops (2*M*N*K of the contraction, zero padding
 * of K excluded) and the summed event time in ms. */
typedef struct {
  char name[64];
  int64_t launches;
  double flops;
  double ms;
} samaudio_kernel_stat;
int samaudio_profile_begin(samaudio_ctx* ctx);
/* Tuning / test hook: force the GEMM kernel variant (-1 automatic; 0..2 the 128-row tiles of gemm.hip;
 * 3 = 256x128 3-stage ring, 4 = 256x128 2-stage ring, 5 = 256x256 2-stage ring, 9 = 256x256 role-split of
 * gemm2.hip where eligible). */
void samaudio_debug_force_gemm_variant(int variant);
/* Tuning hook: A/B switches between kernel generations (flag 1 = first-generation bf16 qkv_prep); 0 = shipped. */
void samaudio_debug_set_flag(int flag, int value);
/* Test aid: leave the LDS of every CU filled with NaN bit patterns (LDS is not cleared between kernels), so that a
 * kernel consuming LDS it never wrote fails deterministically. */
int samaudio_debug_poison_lds(samaudio_stream stream);
int samaudio_profile_end(samaudio_ctx* ctx, samaudio_kernel_stat* out, int capacity, int* count);

/* ---- per-kernel hooks (parity tests; each is one kernel of the path above) -------------------------- */

/* C[b] = epilogue(A(b) @ W^T): generalised GEMM / implicit conv, see sam_audio_amd/csrc/common.h GemmParams.
 * `params` points to a HOST copy of sa::GemmParams (size checked against params_bytes). */
int samaudio_op_gemm(const void* params_host, size_t params_bytes, int precision, samaudio_stream stream);
int samaudio_op_rmsnorm_mod(const float* x, const float* w, const float* shift_tab, const float* scale_tab,
                            const float* tvec, int64_t tvec_ld, int shift_off, int scale_off, void* out,
                            int precision, int rows, int dim, int rows_per_batch, float eps, samaudio_stream stream);
int samaudio_op_groupnorm_silu(const float* x, const float* w, const float* b, void* partials_f64, void* out,
                               int precision, int batch, int frames, int channels, int halo, float eps,
                               samaudio_stream stream);
int samaudio_op_qkv_prep(const void* qkv, const float* q_w, const float* k_w, const float* rope_cos,
                         const float* rope_sin, void* q, void* k, void* vt, int precision, int batch, int frames,
                         int frames_padded, int heads, float eps, samaudio_stream stream);
int samaudio_op_self_attention(const void* q, const void* k, const void* vt, const uint8_t* key_mask, void* out,
                               int precision, int batch, int frames, int frames_padded, int heads,
                               samaudio_stream stream);
int samaudio_op_cross_attention(const void* q, const float* q_w, void* kv, const float* k_w, const uint8_t* mask,
                                void* out, int precision, int batch, int frames, int text_len, int heads, float eps,
                                samaudio_stream stream);
/* U^T[b][n][h*ltp + j] = sum_d wo[n][h*128+d] * V[b][j][h*128+d] (bf16; V = columns [D, 2D) of kv rows b*text_len+j):
 * the per-batch operand of the folded cross-attention output projection. */
int samaudio_op_cross_attn_fold(const void* wo, const void* kv, int64_t kv_ld, void* ut, int kp, int batch, int text_len,
                                int ltp, int heads, samaudio_stream stream);
int samaudio_op_layernorm_accum(const float* x, const float* w, const float* b, const float* gate, float* acc,
                                int rows, int dim, float eps, samaudio_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* SAMAUDIO_H_ */
