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
/* Check that the DiT (what=0), codec (what=1) or codec-encoder-only (what=2: the Judge's DACVAEEncoder,
 * reference codec.py:42-78) weight set is complete and shaped for the config. */
int samaudio_finalize(samaudio_ctx* ctx, int what);

/* Scratch.  `codec_items`: waveforms processed per codec pass (0 = no codec use), `samples`: padded
 * samples per waveform.  Besides the activations of one evaluation the DiT part holds, in 16-bit contexts with text_len <= 16 and
 * 128-wide heads, the folded cross-attention operands of ALL layers (rows * dim * pad64(heads * (8 | 16)) * n_layers 16-bit
 * elements: 0.76 GB for 32 rows at the large stand-in dims, zeroed once per samaudio_prepare), and in fp32 contexts with
 * SAMAUDIO_OPT_X3_CLASSES the split activation operands (rows * (frames + 2) * 3 * dim and rows * frames * 3 * ffn_hidden 16-bit
 * elements).  Query it AFTER setting the options that change the plan (SAMAUDIO_OPT_X3_CLASSES). */
size_t samaudio_workspace_bytes(samaudio_ctx* ctx, int rows, int frames, int text_len, int codec_items,
                                int64_t samples);
int samaudio_set_workspace(samaudio_ctx* ctx, void* workspace, size_t bytes);

/* Execution options of a context (no reference counterpart: scheduling only, results are bitwise unaffected).
 *   SAMAUDIO_OPT_TAIL_SPLIT (default 1): a GEMM whose last round of 256x256 tiles would leave most CUs idle is issued as
 *   two kernels (whole rounds + a 128x128-tile tail).  Set to 0 for contexts that share the GPU with another context's
 *   stream (SAMAudio(streams=2)): there the other stream's workgroups fill the idle CUs and the extra launches cost 2 %. */
#define SAMAUDIO_OPT_TAIL_SPLIT 1
/* Precision of single GEMM classes (the reference computes everything in fp32: README.md:48, no autocast anywhere).
 *   SAMAUDIO_OPT_F32_CLASSES (16-bit contexts; value = mask of SAMAUDIO_CLS_* bits, default 0): the named classes run on
 *   exact-fp32 operands (v_mfma_f32_16x16x4f32) inside an otherwise 16-bit context.  Needs the class's weights registered
 *   a second time in fp32 under "<name>.f32" (samaudio_set_tensor) - checked by samaudio_finalize(ctx, 0) and, on a finalized
 *   context, by samaudio_set_option itself (SAMAUDIO_ERR_WEIGHT names the missing copy); only the classes in SAMAUDIO_CLS_F32_CAPABLE - the
 *   ones whose fp32 cost is < 1 % of a step - are accepted.
 *   SAMAUDIO_OPT_QUANT_CLASSES / SAMAUDIO_OPT_QUANT_FORMAT (fp32 contexts; measurement aid for the error budget of
 *   DESIGN.md section 4): the GEMMs of the named classes round BOTH operands to the 16-bit format (1 = bfloat16,
 *   2 = IEEE fp16) before multiplying, everything else stays exact - the numerical effect of running only that class on
 *   16-bit operands. */
#define SAMAUDIO_OPT_F32_CLASSES 2
#define SAMAUDIO_OPT_QUANT_CLASSES 3
#define SAMAUDIO_OPT_QUANT_FORMAT 4
/*   SAMAUDIO_OPT_ALT16_CLASSES (16-bit contexts; value = mask of the five big GEMM classes of the DiT layers - QKV, WO, CWQ, W13,
 *   W2 -, default 0): precision "mixed".  The named classes read their operands in the library's OTHER 16-bit format - bfloat16
 *   in libsamaudio_hip_f16.so - i.e. BASELINE's dtype for 96 % of the flops, IEEE fp16 for the classes that carry the error
 *   (DESIGN.md section 4); their weight tensors are registered in that format (same dtype code: a 16-bit tensor is bits), and
 *   the kernels that produce their activation operands (RMSNorm + modulate, self-attention, the wo / w13 epilogues) write that
 *   format.  In libsamaudio_hip.so both formats are bfloat16 and the option changes nothing. */
#define SAMAUDIO_OPT_ALT16_CLASSES 5
/*   SAMAUDIO_OPT_PREFETCH_ROWS (16-bit contexts; default 0 = off): in evaluations of at most this many rows (rows = batch x
 *   frames of one samaudio_forward), every big GEMM of a DiT layer that leaves CUs idle (< 256 workgroups: the few-row launches of
 *   a batch shard, 4 clips per GPU) uses them to read the NEXT GEMM's weights once, linearly, so that the next launch finds them
 *   in the memory-side cache instead of fetching them cold.  Scheduling only: results are bitwise unaffected.
 * Weight layout of the five big GEMM classes of the DiT layers (L<i>.wqkv, wo, c_wq, w13, w2), chosen per tensor by the shape it is
 * registered with: [N, K] row-major, or [K/64, N, 64] K-TILE-MAJOR - the 64-element K slab of all N rows contiguous (16-bit contexts
 * only; sam_audio_amd/weights.py ktm_layout).  Same values, same results; a launch of few rows then streams its weights front to back. */
#define SAMAUDIO_OPT_PREFETCH_ROWS 6
/*   SAMAUDIO_OPT_SENTINEL (default 0; validation aid, never on in timed runs): value 1 makes the context scan every 16-bit tensor a
 *   GEMM of the hot path writes (and the RMSNorm / attention outputs that feed GEMMs) for its largest magnitude and for
 *   non-finite values, per GEMM class, on the launch stream - an IEEE-fp16 operand that overflowed (|x| > 65504 -> inf) is then
 *   REPORTED by samaudio_sentinel_read with the class it came from instead of propagating silently.  Allocates 4 KiB of device
 *   memory on first use (the only allocation of the library besides the checksum trace). */
#define SAMAUDIO_OPT_SENTINEL 7
/* (option 8 was SAMAUDIO_OPT_ODE_GRAPH, rounds 5: a solve's launches replayed as one HIP graph.  Bitwise equal and measured at
 * + 0.0 ... 0.3 % - the host already runs ahead of the GPU, there is no launch gap to close - so it was removed in round 6;
 * profiles/r5_call13/ holds the measurement, the git history the code.) */
/*   SAMAUDIO_OPT_X3_CLASSES (fp32 contexts; value = mask of SAMAUDIO_CLS_X3_CAPABLE bits, default 0): COMPENSATED 16-bit operands -
 *   precision "fp16x3" of the host classes.  The reference computes in fp32 (README.md:48); no plain 16-bit operand format holds
 *   the 1e-3 parity bound on trained-like weight statistics.  A GEMM of a named class keeps its fp32
 *   activation operand and fp32 outputs, but multiplies on the 16-bit MFMA: each operand is split into hi = rn16(x) and
 *   lo = rn16(x - hi), the activation row becomes [lo | hi | hi] (3K elements, written by a streaming kernel on the launch stream)
 *   and the weight row [W_hi | W_lo | W_hi], so that ONE launch of the library's 16-bit GEMM over K' = 3K accumulates
 *   x_lo W_hi + x_hi W_lo + x_hi W_hi in fp32 - the fp32 product to ~2^-21 relative for three times the MFMA work (the 8-phase kernels
 *   know the layout and stage / read what the three products share once: csrc/gemm8.hip gemm8x_kernel).  Needs the
 *   class's weights registered in that split form under "<name>.x3" - 16-bit, [N, 3K] row-major or [3K/64, N, 64] K-tile-major
 *   (sam_audio_amd/weights.py x3_weight) - for L<i>.wqkv, wo, c_wq, c_wo, w13, w2; checked like the ".f32" copies above.  Classes
 *   PATCH (the patcher's k3 convolutions: "patch1.w.x3" / "patch2.w.x3", [D, 9D] with EACH tap's D columns split into 3D) and CKV
 *   ("c_wkv_all.x3") can be switched on as well.  Class CODEC works without a second copy of anything: the fp32 convolution kernel splits
 *   the fp32 fragments of both operands in registers and multiplies them as lo*hi + hi*lo + hi*hi on the 16-bit MFMA.  Optional twins make
 *   it faster: "<name>.x3" ([N, K / Cin, 3 Cin], weights.py convert_codec_x3) sends a convolution with >= 256 output channels through the
 *   16-bit 8-phase kernels over K' = 3K; "<name>.fly" (16-bit, [N, 2K]: the weight split ONCE in the fragment layout of the fp32 kernel,
 *   csrc/common.h GEMM_FLAG_W_FLY16, weights.py convert_codec_fly16) takes the split of the weight out of that kernel's K loop.  Same
 *   bits with or without them.
 *   With class CWO on, text memories of <= 16 tokens (128-wide heads) take the folded form of the cross-attention output projection on
 *   compensated operands (h += P . U, U = Wo V per layer and batch item: K' = 3 * pad64(heads * (8 | 16)) instead of 3 * dim).
 *   Bit SAMAUDIO_X3_ATTENTION does the same for the two contractions of the self-attention.  Everything else of the context
 *   (norms, softmax, the small GEMM classes, the codec) stays exact fp32.  In
 *   libsamaudio_hip_f16.so the halves are IEEE fp16 (22 mantissa bits per operand); in libsamaudio_hip.so bfloat16 (16 bits). */
#define SAMAUDIO_OPT_X3_CLASSES 9
#define SAMAUDIO_SENTINEL_SLOTS 16   /* SAMAUDIO_CLS_COUNT GEMM classes (bit order) + slot 14: RMSNorm outputs, 15: attention outputs */
#define SAMAUDIO_CLS_ALT16_CAPABLE (SAMAUDIO_CLS_QKV | SAMAUDIO_CLS_WO | SAMAUDIO_CLS_CWQ | SAMAUDIO_CLS_W13 | SAMAUDIO_CLS_W2)
#define SAMAUDIO_CLS_TIME (1 << 0)   /* t_embedder MLP + t_block (transformer.py:236-257,462-467): 1 row per time value */
#define SAMAUDIO_CLS_OUT (1 << 1)    /* DiT output projection D -> 256 (transformer.py:519): feeds the ODE state */
#define SAMAUDIO_CLS_IN (1 << 2)     /* proj, noisy-audio columns (model.py:116-125): reads the ODE state */
#define SAMAUDIO_CLS_PREP (1 << 3)   /* hoisted conditioning: proj feature columns, video conv1x1, anchors, memory_proj */
#define SAMAUDIO_CLS_YEMB (1 << 4)   /* y_embedder (transformer.py:260-288) */
#define SAMAUDIO_CLS_CKV (1 << 5)    /* cross-attention K | V projections of all layers */
#define SAMAUDIO_CLS_PATCH (1 << 6)  /* patcher k3 convolutions (patcher.py:48-67) */
#define SAMAUDIO_CLS_QKV (1 << 7)
#define SAMAUDIO_CLS_WO (1 << 8)
#define SAMAUDIO_CLS_CWQ (1 << 9)
#define SAMAUDIO_CLS_CWO (1 << 10)
#define SAMAUDIO_CLS_W13 (1 << 11)
#define SAMAUDIO_CLS_W2 (1 << 12)
#define SAMAUDIO_CLS_CODEC (1 << 13) /* every DAC-VAE convolution */
#define SAMAUDIO_CLS_COUNT 14
/* SAMAUDIO_OPT_X3_CLASSES only (not a GEMM class): the self-attention of an fp32 context on hi/lo-split operands on the 16-bit MFMA
 * (S = Ql Kh + Qh Kl + Qh Kh, O likewise) instead of the fp32 vector-ALU kernel; fp32 tensors in and out */
#define SAMAUDIO_X3_ATTENTION (1 << 14)
#define SAMAUDIO_CLS_X3_CAPABLE \
  (SAMAUDIO_CLS_QKV | SAMAUDIO_CLS_WO | SAMAUDIO_CLS_CWQ | SAMAUDIO_CLS_CWO | SAMAUDIO_CLS_W13 | SAMAUDIO_CLS_W2 | SAMAUDIO_CLS_PATCH | \
   SAMAUDIO_CLS_CKV | SAMAUDIO_CLS_CODEC | SAMAUDIO_X3_ATTENTION)
#define SAMAUDIO_CLS_F32_CAPABLE (SAMAUDIO_CLS_TIME | SAMAUDIO_CLS_OUT | SAMAUDIO_CLS_IN | SAMAUDIO_CLS_PREP | SAMAUDIO_CLS_YEMB)
int samaudio_set_option(samaudio_ctx* ctx, int option, int value);

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
/* The same with the conditioning as separate() holds it (no concatenated / repeated copies):
 *   latent [rows / candidates, frames, latent_channels / 2] f32: the DAC-VAE mean latent z; audio_features = (z | z)
 *   (model.py:182-184) is never materialised - the feature GEMM reads each z row twice (two K taps, stride 0);
 *   candidates >= 1: text / text_mask / video / anchor_ids / anchor_alignment / audio_pad_mask hold rows / candidates clips; the
 *   per-clip conditioning is computed once and repeated sample-major for the clip's candidates (model.py:193-203) inside the engine. */
int samaudio_prepare_latent(samaudio_ctx* ctx, int rows, int frames, int text_len, int candidates, const float* latent,
                            const float* text, const uint8_t* text_mask, const float* video, const int64_t* anchor_ids,
                            int n_ids, const int64_t* anchor_alignment, const uint8_t* audio_pad_mask, samaudio_stream stream);

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
/* Decode straight from the ODE state (reference model.py:291-295 without the transposed copy): state [rows, frames, 2 * codec_dim]
 * f32 -> wav [2 * rows, frames * hop]: waveform 2b = channels [0, codec_dim) of row block b (target), 2b + 1 the rest (residual). */
int samaudio_codec_decode_pairs(samaudio_ctx* ctx, const float* state, int rows, int frames, float* wav, samaudio_stream stream);

/* ---- reranking and span prediction ------------------------------------------ */

/* One PE-AV transformer (perception_models core.audio_visual_encoder.transformer.Transformer, un-vendored; the
 * reference instantiates it at sam_audio/model/judge.py:46-47; restated from its Hugging Face port
 * transformers/models/pe_audio/modeling_pe_audio.py:241-287,344-490,616-680). */
typedef struct {
  int32_t dim, n_heads, n_layers, ffn_hidden; /* hidden_size, num_attention_heads, num_hidden_layers, intermediate_size */
  int32_t in_dim;                             /* width of the features entering the input projection */
  int32_t max_positions;                      /* rows of the RoPE table */
  int32_t attn_bias;                          /* 1: q/k/v/o projections carry biases */
  float norm_eps;
} samaudio_peav_dims;

/* SAMAudioJudgeModel (reference sam_audio/model/judge.py:35-132, config.py:234-251).  The ModernBERT text tower and
 * the tokenizer stay with the caller (PyTorch-ROCm); the DAC encoder is a codec-only samaudio_ctx. */
typedef struct {
  int32_t precision;
  samaudio_peav_dims transformer, finetune_transformer;
  int32_t codec_dim;      /* 128: transformer.in_dim */
  int32_t text_hidden;    /* text_model.hidden_size */
  int32_t bottleneck_dim; /* 256: finetune_transformer.in_dim */
} samaudio_judge_config;

typedef struct samaudio_judge samaudio_judge;
int samaudio_judge_create(const samaudio_judge_config* cfg, samaudio_judge** out);
void samaudio_judge_destroy(samaudio_judge* j);
/* engine names: sam_audio_amd/judge.py documents the mapping from the reference state_dict keys */
int samaudio_judge_set_tensor(samaudio_judge* j, const char* name, const void* data, int dtype, int ndim,
                              const int64_t* shape);
int samaudio_judge_finalize(samaudio_judge* j);
size_t samaudio_judge_workspace_bytes(samaudio_judge* j, int inputs, int candidates, int frames);
int samaudio_judge_set_workspace(samaudio_judge* j, void* workspace, size_t bytes);
/* SAMAudioJudgeModel.forward after the codec and the text tower (judge.py:98-132):
 *   input_latent     [inputs, frames, codec_dim] f32              DAC mean latents of the mixtures (codec.py:65-70)
 *   separated_latent [inputs*candidates, frames, codec_dim] f32   ... of the candidate separations, sample-major
 *   text_pooled      [inputs*candidates, text_hidden] f32         _get_text_output(...).pooler_output (judge.py:76-88)
 *   pad_mask         [inputs, frames] u8 or NULL                  padding_mask[:, ::hop] (judge.py:104-107), 1 = valid
 *   scores           [inputs*candidates, 4] f32                   overall, recall, precision, faithfulness (de-normalised)
 * The reference repeats every mixture once per candidate (ranking/judge.py:31-33); all ops are per-row, so the
 * mixture branch runs once per clip here.  candidates = 1 is the plain forward(). */
int samaudio_judge_score(samaudio_judge* j, const float* input_latent, const float* separated_latent, int inputs,
                         int candidates, int frames, const float* text_pooled, const uint8_t* pad_mask, float* scores,
                         samaudio_stream stream);
/* parity hook: transformer `which` (0 = transformer, 1 = finetune_transformer) alone:
 * x [rows, frames, in_dim] f32 -> hidden [rows, frames + 1, dim] f32 (row 0 of each item = pooler_output). */
int samaudio_judge_encode(samaudio_judge* j, int which, const float* x, const uint8_t* pad_mask, int rows, int frames,
                          float* hidden, samaudio_stream stream);

/* PE-A-Frame span predictor (reference model.py:96-102,231-245; un-vendored PEAudioFrame; Hugging Face port
 * modeling_pe_audio.py:810-868): per-frame audio-text logits for batch-paired (audio, description) rows. */
typedef struct {
  int32_t precision;
  samaudio_peav_dims audio;
  int32_t codec_dim;  /* 128 */
  int32_t embed_dim;  /* text hidden size = width of the joint space */
} samaudio_frame_config;

typedef struct samaudio_frame samaudio_frame;
int samaudio_frame_create(const samaudio_frame_config* cfg, samaudio_frame** out);
void samaudio_frame_destroy(samaudio_frame* f);
int samaudio_frame_set_tensor(samaudio_frame* f, const char* name, const void* data, int dtype, int ndim,
                              const int64_t* shape);
int samaudio_frame_finalize(samaudio_frame* f);
size_t samaudio_frame_workspace_bytes(samaudio_frame* f, int rows, int frames);
int samaudio_frame_set_workspace(samaudio_frame* f, void* workspace, size_t bytes);
/* codec_features [rows, frames, codec_dim] f32 (= audio_features[:, :, :128], model.py:239), text_pooled
 * [rows, embed_dim] f32 (token 0 of the text tower's last hidden state), pad_mask [rows, frames] u8 or NULL
 * -> logits [rows, frames] f32 */
int samaudio_frame_logits(samaudio_frame* f, const float* codec_features, const float* text_pooled,
                          const uint8_t* pad_mask, int rows, int frames, float* logits, samaudio_stream stream);

/* ---- visual-prompt tower -------------------------------------------------------
 * PE-Core vision tower behind `PerceptionEncoder.encode` (reference sam_audio/model/vision_encoder.py:80-89:
 * `pe.CLIP.from_config("PE-Core-L14-336")`, `encode_image(x, normalize=...)`).  Resize / scaling / normalisation of the
 * frames (vision_encoder.py:91-113) stay with the caller; everything from the patch embedding to the L2-normalised
 * feature runs here.  Engine tensor names: sam_audio_amd/vision_tower.py documents the mapping from the `visual.*`
 * state_dict keys. */
typedef struct {
  int32_t precision;                          /* SAMAUDIO_F32 | SAMAUDIO_BF16 */
  int32_t image_size, patch_size;             /* 336, 14 */
  int32_t width, layers, heads, mlp_width;    /* 1024, 24, 16 (head dim 64 or 128), 4096 */
  int32_t output_dim;                         /* 1024 */
  int32_t use_cls_token, use_rope2d, use_ln_pre, use_ln_post;
  int32_t pool_type;                          /* 0 = class token, 1 = token mean, 2 = attention pooling */
  int32_t pool_heads;                         /* heads of the pooling attention (8; head dim 64 or 128) */
  int32_t act;                                /* 4 = GELU (erf), 5 = quick GELU */
  float ln_eps;
} samaudio_vit_config;

typedef struct samaudio_vit samaudio_vit;
int samaudio_vit_create(const samaudio_vit_config* cfg, samaudio_vit** out);
void samaudio_vit_destroy(samaudio_vit* v);
int samaudio_vit_set_tensor(samaudio_vit* v, const char* name, const void* data, int dtype, int ndim,
                            const int64_t* shape);
int samaudio_vit_finalize(samaudio_vit* v);
size_t samaudio_vit_workspace_bytes(samaudio_vit* v, int frames);
int samaudio_vit_set_workspace(samaudio_vit* v, void* workspace, size_t bytes);
/* frames [n, 3, image_size, image_size] f32 (resized, scaled, normalised) -> features [n, output_dim] f32, L2-normalised
 * when `normalize` != 0.  tokens_out (nullable, parity hook): the residual stream after the last block,
 * [n, tokens, width] f32 (before ln_post). */
int samaudio_vit_encode(samaudio_vit* v, const float* frames, int n, int normalize, float* features, float* tokens_out,
                        samaudio_stream stream);

/* ---- text-prompt encoder ---------------------------------------------------------
 * T5 encoder stack behind `T5TextEncoder.forward` (reference sam_audio/model/text_encoder.py:19-37:
 * `transformers.T5EncoderModel("t5-base")(input_ids, attention_mask)["last_hidden_state"]`).  Tokenisation stays with the
 * caller (the Hugging Face tokenizer, text_encoder.py:21-27); everything from the embedding lookup to the final
 * T5LayerNorm runs here.  Engine tensor names: sam_audio_amd/t5_encoder.py documents the mapping from the
 * `shared.* / encoder.*` state_dict keys. */
typedef struct {
  int32_t precision;                 /* SAMAUDIO_F32 | SAMAUDIO_BF16 (GEMM operands; the residual stream is f32) */
  int32_t vocab, d_model, d_kv;      /* 32128, 768, 64 (d_kv <= 128) */
  int32_t heads, d_ff, layers;       /* 12, 3072, 12; heads * d_kv and d_ff must be multiples of 64 */
  int32_t max_len;                   /* longest sequence the relative-position table covers (<= 512) */
  int32_t act;                       /* 6 = ReLU (t5-base), 7 = tanh-form GELU ("gelu_new"); gated variants unsupported */
  float ln_eps;                      /* 1e-6 */
} samaudio_t5_config;

typedef struct samaudio_t5 samaudio_t5;
int samaudio_t5_create(const samaudio_t5_config* cfg, samaudio_t5** out);
void samaudio_t5_destroy(samaudio_t5* t);
int samaudio_t5_set_tensor(samaudio_t5* t, const char* name, const void* data, int dtype, int ndim,
                           const int64_t* shape);
int samaudio_t5_finalize(samaudio_t5* t);
size_t samaudio_t5_workspace_bytes(samaudio_t5* t, int rows, int tokens);
int samaudio_t5_set_workspace(samaudio_t5* t, void* workspace, size_t bytes);
/* input_ids [rows, tokens] i64 (device), attention_mask [rows, tokens] u8 (1 = token, device) ->
 * last_hidden_state [rows, tokens, d_model] f32, every position (padding rows included, as transformers returns them).
 * An id outside [0, vocab) is the caller's error: the lookup clamps it. */
int samaudio_t5_encode(samaudio_t5* t, const int64_t* input_ids, const unsigned char* attention_mask, int rows, int tokens,
                       float* last_hidden_state, samaudio_stream stream);

/* ---- Judge / span-predictor text tower -------------------------------------------
 * ModernBERT encoder behind `SAMAudioJudgeModel._get_text_output` and `PEAudioFrame` (reference sam_audio/model/judge.py:48,
 * 74-88: `transformers.AutoModel.from_config(ModernBertConfig(...))`, `hidden_states[nth_text_layer]`).  Tokenisation stays
 * with the caller; everything from the embedding lookup to the requested hidden state runs here.  Engine tensor names:
 * sam_audio_amd/mbert_encoder.py documents the mapping from the `embeddings.* / layers.* / final_norm.*` state_dict keys. */
typedef struct {
  int32_t precision;                  /* SAMAUDIO_F32 | SAMAUDIO_BF16 (GEMM operands; the residual stream is f32) */
  int32_t vocab, hidden, heads;       /* 50368, 768, 12 (head dim even, <= 128) */
  int32_t intermediate, layers;       /* 1152 (Wi projects to 2x that: input | gate), 22 */
  int32_t global_every;               /* 3: layers l % 3 == 0 attend globally, the others within +-window tokens */
  int32_t window;                     /* 64 = local_attention / 2 */
  int32_t max_len;                    /* longest sequence the rotary tables cover (<= 512) */
  float ln_eps;                       /* 1e-5 */
} samaudio_mbert_config;

typedef struct samaudio_mbert samaudio_mbert;
int samaudio_mbert_create(const samaudio_mbert_config* cfg, samaudio_mbert** out);
void samaudio_mbert_destroy(samaudio_mbert* t);
int samaudio_mbert_set_tensor(samaudio_mbert* t, const char* name, const void* data, int dtype, int ndim,
                              const int64_t* shape);
int samaudio_mbert_finalize(samaudio_mbert* t);
size_t samaudio_mbert_workspace_bytes(samaudio_mbert* t, int rows, int tokens);
int samaudio_mbert_set_workspace(samaudio_mbert* t, void* workspace, size_t bytes);
/* input_ids [rows, tokens] i64, attention_mask [rows, tokens] u8 (1 = token) -> hidden [rows, tokens, hidden] f32.
 * nth_hidden_state: 0 <= n <= layers = the residual stream after n layers (0 = the normalised embeddings), NEVER passed
 * through final_norm; n < 0 = last_hidden_state (after the final LayerNorm).  The two transformers generations disagree
 * about hidden_states[layers] - 4.48 .. 4.5x (what the reference pins, pyproject.toml: transformers>=4.54, and what the
 * released Judge was trained with) append the last layer's output BEFORE final_norm, 5.x records the normalised tensor -
 * and the reference's default nth_text_layer = 22 = layers selects exactly that entry (judge.py:74-88).  The ABI is
 * explicit: n == layers is the pre-norm tensor; a caller that wants 5.x semantics passes -1 (the host classes do this
 * under SAMAudioJudgeConfig.last_text_layer_prenorm = False). */
int samaudio_mbert_encode(samaudio_mbert* t, const int64_t* input_ids, const unsigned char* attention_mask, int rows, int tokens,
                          int nth_hidden_state, float* hidden, samaudio_stream stream);

/* ---- measurement ----------------------------------------------------------------------------------- */

/* Live per-kernel timing for bench.py's roofline leg (the reference has no counterpart: it publishes no
 * throughput numbers).  Between begin and end every GEMM launch on the hot path is
 * bracketed by a hipEvent pair recorded on the launch stream; end synchronises and returns, per GEMM tile
 * variant, the number of launches, the summed ALGORITHMIC flops (2*M*N*K of the contraction, zero padding
 * of K excluded) and the summed event time in ms. */
typedef struct {
  char name[64];
  int64_t launches;
  double flops; /* algorithmic flops of the launches (2*M*N*K for a contraction; 0 for pure streaming kernels) */
  double bytes; /* algorithmic bytes of the launches: every operand, output and residual element counted once */
  double ms;
} samaudio_kernel_stat;
int samaudio_profile_begin(samaudio_ctx* ctx);
/* SAMAUDIO_OPT_SENTINEL: synchronises `stream`, copies out absmax[SAMAUDIO_SENTINEL_SLOTS] / nonfinite[SAMAUDIO_SENTINEL_SLOTS]
 * (counts as doubles) accumulated since the last read, and resets them. */
int samaudio_sentinel_read(samaudio_ctx* ctx, float* absmax, double* nonfinite, samaudio_stream stream);
/* Test hook: force the GEMM kernel variant (-1 automatic; 0..2 the 128-row tiles of gemm.hip; 22 = gemm8 256x256 8-phase,
 * 27 = gemm8s 128x128, 25 / 26 / 28 / 29 / 32 / 33 / 34 = the 32x32x16-family tiles, 35 = conv7h; csrc/gemm.hip
 * gemm_variant_name).  A launch the forced kernel does not cover falls back to gemm.hip's tiles - except launches on
 * K-tile-major weights or with the split-form output, which exist in the 8-phase family only: forcing anything but 22 / 27 on
 * them is refused with SAMAUDIO_ERR_ARG and that reason (a model's own launches never leave the family). */
void samaudio_debug_force_gemm_variant(int variant);
/* Test hooks (csrc/kernels.h lists them; 0 = shipped behaviour): 11 = k7 convolutions as implicit GEMMs, 16 = DAC residual
 * units as two launches, 18 = fuse residual units whatever the launch size, 19 = residual-unit kernel form (1 / 3 = weight-
 * stationary, 2 = ring), 21 = gemm8s always in its plain double-buffered form, 24 = epilogue form of the 8-phase family (1 = the
 * general one for every launch, 2 / 3 = one lean form for every eligible launch), 26 = 1: one workgroup per tile (shipped:
 * persistent above 256 tiles). */
void samaudio_debug_set_flag(int flag, int value);
/* Test aid: leave the LDS of every CU filled with NaN bit patterns (LDS is not cleared between kernels), so that a
 * kernel consuming LDS it never wrote fails deterministically. */
int samaudio_debug_poison_lds(samaudio_stream stream);
int samaudio_profile_end(samaudio_ctx* ctx, samaudio_kernel_stat* out, int capacity, int* count);

/* ---- per-kernel hooks (parity tests; each is one kernel of the path above) -------------------------- */

/* C[b] = epilogue(A(b) @ W^T): generalised GEMM / implicit conv, see sam_audio_amd/csrc/common.h GemmParams.
 * `params` points to a HOST copy of sa::GemmParams (size checked against params_bytes). */
int samaudio_op_gemm(const void* params_host, size_t params_bytes, int precision, samaudio_stream stream);
/* One DAC residual unit as ONE kernel: the k7 convolution `conv7_params` and the k1 convolution `conv1_params` that reads
 * its output (two host GemmParams as for samaudio_op_gemm, 16-bit operands only).  The intermediate activation
 * (conv7_params.out_act == conv1_params.A) is never written; conv1_params.out_act must not be conv7_params.A.  Bitwise
 * equal to the two samaudio_op_gemm launches; ERR_ARG when the pair is not one the fused kernel covers. */
int samaudio_op_resunit(const void* conv7_params_host, const void* conv1_params_host, size_t params_bytes,
                        samaudio_stream stream);
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
/* masked GroupNorm(1) + SiLU of the PE-AV patch embedder: x [batch, frames, channels] f32, mask [batch, frames] u8,
 * partials_f64: batch*64*3 doubles of scratch, out [batch][halo + frames + halo][channels] (halo rows untouched) */
int samaudio_op_masked_groupnorm_silu(const float* x, const float* w, const float* b, const uint8_t* mask,
                                      void* partials_f64, void* out, int precision, int batch, int frames,
                                      int channels, int halo, float eps, samaudio_stream stream);
int samaudio_op_layernorm_rows(const float* x, int64_t x_ld, const float* w, const float* b, float* out_f32,
                               void* out_act, int precision, int64_t rows, int dim, float eps, samaudio_stream stream);
/* SAMAUDIO_OPT_X3_CLASSES: the activation operand of a compensated GEMM - x [rows, k] f32 (row stride x_ld) -> out [rows, 3k]
 * in the library's 16-bit format = [lo | hi | hi] per row, hi = rn16(x) (clamped to the largest finite value), lo = rn16(x - hi) */
int samaudio_op_split3(const float* x, int64_t x_ld, void* out, int64_t rows, int k, samaudio_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* SAMAUDIO_H_ */
