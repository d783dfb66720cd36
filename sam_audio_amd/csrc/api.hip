// extern "C" surface of libsamaudio_hip.so (declared in include/samaudio.h).
#include <cstring>
#include <string>

#include "engine.h"
#include "peav.h"

struct samaudio_ctx {
  sa::Engine* engine;
};
struct samaudio_judge {
  sa::Judge* judge;
};
struct samaudio_frame {
  sa::FramePredictor* frame;
};

namespace {
thread_local std::string g_err;
}  // namespace
namespace sa {
void set_last_error(const std::string& msg) { g_err = msg; }  // for the C entry points that live in other files (vit.hip)
}  // namespace sa
namespace {
int ret(const sa::Status& s) {
  if (!s.ok()) g_err = s.msg;
  return s.code;
}
int hip_ret(hipError_t e, const char* what) {
  if (e == hipSuccess) return SAMAUDIO_OK;
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return SAMAUDIO_ERR_HIP;
}
int bad(const char* msg) {
  g_err = msg;
  return SAMAUDIO_ERR_ARG;
}
}  // namespace

extern "C" {

const char* samaudio_last_error(void) { return g_err.c_str(); }
#ifdef SA_OPERAND_FP16
const char* samaudio_version(void) { return "samaudio-hip 0.1 (gfx950, fp16 operands)"; }
#else
const char* samaudio_version(void) { return "samaudio-hip 0.1 (gfx950)"; }
#endif

int samaudio_create(const samaudio_config* cfg, samaudio_ctx** out) {
  if (!cfg || !out) return bad("samaudio_create: null argument");
  if (cfg->precision != SAMAUDIO_F32 && cfg->precision != SAMAUDIO_BF16) return bad("samaudio_create: precision");
  if (cfg->dim <= 0 || cfg->n_heads <= 0 || cfg->n_layers < 0 || cfg->ffn_hidden <= 0) return bad("samaudio_create: dims");
  samaudio_ctx* c = new samaudio_ctx;
  c->engine = new sa::Engine(*cfg);
  *out = c;
  return SAMAUDIO_OK;
}

void samaudio_destroy(samaudio_ctx* ctx) {
  if (!ctx) return;
  delete ctx->engine;
  delete ctx;
}

int samaudio_set_tensor(samaudio_ctx* ctx, const char* name, const void* data, int dtype, int ndim,
                        const int64_t* shape) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->set_tensor(name, data, dtype, ndim, shape));
}

int samaudio_set_option(samaudio_ctx* ctx, int option, int value) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->set_option(option, value));
}

int samaudio_finalize(samaudio_ctx* ctx, int what) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->finalize(what));
}

size_t samaudio_workspace_bytes(samaudio_ctx* ctx, int rows, int frames, int text_len, int codec_items,
                                int64_t samples) {
  if (!ctx) return 0;
  return ctx->engine->workspace_bytes(rows, frames, text_len, codec_items, samples);
}

int samaudio_set_workspace(samaudio_ctx* ctx, void* workspace, size_t bytes) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->set_workspace(workspace, bytes));
}

int samaudio_prepare(samaudio_ctx* ctx, int rows, int frames, int text_len, const float* audio_features,
                     const float* text, const uint8_t* text_mask, const float* video, const int64_t* anchor_ids,
                     int n_ids, const int64_t* anchor_alignment, const uint8_t* audio_pad_mask,
                     samaudio_stream stream) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->prepare(rows, frames, text_len, audio_features, text, text_mask, video, anchor_ids, n_ids,
                                  anchor_alignment, audio_pad_mask, (hipStream_t)stream));
}

int samaudio_prepare_latent(samaudio_ctx* ctx, int rows, int frames, int text_len, int candidates, const float* latent,
                            const float* text, const uint8_t* text_mask, const float* video, const int64_t* anchor_ids,
                            int n_ids, const int64_t* anchor_alignment, const uint8_t* audio_pad_mask, samaudio_stream stream) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->prepare(rows, frames, text_len, latent, text, text_mask, video, anchor_ids, n_ids, anchor_alignment,
                                  audio_pad_mask, (hipStream_t)stream, candidates, true));
}

int samaudio_forward(samaudio_ctx* ctx, const float* noisy, const float* time, int n_time, float* out,
                     samaudio_stream stream) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->forward(noisy, time, n_time, out, (hipStream_t)stream));
}

int samaudio_ode_solve(samaudio_ctx* ctx, float* state, int method, const float* grid_host, int n_grid,
                       samaudio_stream stream) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->ode_solve(state, method, grid_host, n_grid, (hipStream_t)stream));
}

int samaudio_codec_encode(samaudio_ctx* ctx, const float* wav, int items, int64_t samples, float* latent,
                          samaudio_stream stream) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->codec_encode(wav, items, samples, latent, (hipStream_t)stream));
}

int samaudio_codec_decode(samaudio_ctx* ctx, const float* latent, int items, int frames, float* wav,
                          samaudio_stream stream) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->codec_decode(latent, items, frames, wav, (hipStream_t)stream));
}

int samaudio_codec_decode_pairs(samaudio_ctx* ctx, const float* state, int rows, int frames, float* wav, samaudio_stream stream) {
  if (!ctx) return bad("null context");
  if (rows <= 0) return bad("codec_decode_pairs: rows");
  return ret(ctx->engine->codec_decode(state, 2 * rows, frames, wav, (hipStream_t)stream, true));
}

void samaudio_debug_force_gemm_variant(int variant) { sa::gemm_force_variant(variant); }
void samaudio_debug_set_flag(int flag, int value) { sa::set_debug_flag(flag, value); }
int samaudio_debug_poison_lds(samaudio_stream stream) {
  return hip_ret(sa::launch_poison_lds((hipStream_t)stream), "poison_lds");
}

int samaudio_profile_begin(samaudio_ctx* ctx) {
  if (!ctx) return bad("null context");
  return ret(ctx->engine->profile_begin());
}

int samaudio_sentinel_read(samaudio_ctx* ctx, float* absmax, double* nonfinite, samaudio_stream stream) {
  if (!ctx || !absmax || !nonfinite) return bad("samaudio_sentinel_read: null argument");
  return ret(ctx->engine->sentinel_read(absmax, nonfinite, (hipStream_t)stream));
}


int samaudio_profile_end(samaudio_ctx* ctx, samaudio_kernel_stat* out, int capacity, int* count) {
  if (!ctx || !count || (capacity > 0 && !out)) return bad("samaudio_profile_end: null argument");
  std::vector<sa::Engine::KernelStat> st;
  const int rc = ret(ctx->engine->profile_end(st));
  if (rc) return rc;
  int n = 0;
  for (const auto& k : st) {
    if (k.launches == 0) continue;
    if (n >= capacity) break;
    std::memset(&out[n], 0, sizeof(out[n]));
    std::strncpy(out[n].name, k.name.c_str(), sizeof(out[n].name) - 1);
    out[n].launches = k.launches;
    out[n].flops = k.flops;
    out[n].bytes = k.bytes;
    out[n].ms = k.ms;
    ++n;
  }
  *count = n;
  return SAMAUDIO_OK;
}

// ---- per-kernel hooks ------------------------------------------------------------------------------
int samaudio_op_gemm(const void* params_host, size_t params_bytes, int precision, samaudio_stream stream) {
  if (!params_host || params_bytes != sizeof(sa::GemmParams)) return bad("samaudio_op_gemm: GemmParams size mismatch");
  sa::GemmParams p;
  std::memcpy(&p, params_host, sizeof(p));
  const bool bf16 = precision == SAMAUDIO_BF16;
  if (const char* why = sa::gemm_check(p, bf16)) return bad(why);
  return hip_ret(sa::launch_gemm(p, bf16, (hipStream_t)stream), "gemm");
}

int samaudio_op_resunit(const void* conv7_params_host, const void* conv1_params_host, size_t params_bytes,
                        samaudio_stream stream) {
  if (!conv7_params_host || !conv1_params_host || params_bytes != sizeof(sa::GemmParams))
    return bad("samaudio_op_resunit: GemmParams size mismatch");
  sa::GemmParams p, q;
  std::memcpy(&p, conv7_params_host, sizeof(p));
  std::memcpy(&q, conv1_params_host, sizeof(q));
  if (const char* why = sa::gemm_check(p, true)) return bad(why);
  if (const char* why = sa::gemm_check(q, true)) return bad(why);
  if (!sa::resunit_ok(p, q)) return bad("samaudio_op_resunit: not a (k7, k1) residual-unit pair the fused kernel covers");
  return hip_ret(sa::launch_resunit(p, q, (hipStream_t)stream), "resunit");
}

int samaudio_op_rmsnorm_mod(const float* x, const float* w, const float* shift_tab, const float* scale_tab,
                            const float* tvec, int64_t tvec_ld, int shift_off, int scale_off, void* out,
                            int precision, int rows, int dim, int rows_per_batch, float eps, samaudio_stream stream) {
  if (dim % 4) return bad("rmsnorm_mod: dim % 4");
  return hip_ret(sa::launch_rmsnorm_mod(x, w, shift_tab, scale_tab, tvec, tvec_ld, shift_off, scale_off, out,
                                        precision == SAMAUDIO_BF16, rows, dim, rows_per_batch, eps,
                                        (hipStream_t)stream), "rmsnorm_mod");
}

int samaudio_op_groupnorm_silu(const float* x, const float* w, const float* b, void* partials_f64, void* out,
                               int precision, int batch, int frames, int channels, int halo, float eps,
                               samaudio_stream stream) {
  if (channels % 4) return bad("groupnorm: channels % 4");
  return hip_ret(sa::launch_groupnorm_silu(x, w, b, (double*)partials_f64, out, precision == SAMAUDIO_BF16, batch,
                                           frames, channels, halo, eps, (hipStream_t)stream), "groupnorm_silu");
}

int samaudio_op_qkv_prep(const void* qkv, const float* q_w, const float* k_w, const float* rope_cos,
                         const float* rope_sin, void* q, void* k, void* vt, int precision, int batch, int frames,
                         int frames_padded, int heads, float eps, samaudio_stream stream) {
  if (frames_padded % 64 || frames_padded < frames) return bad("qkv_prep: frames_padded");
  return hip_ret(sa::launch_qkv_prep(qkv, q_w, k_w, rope_cos, rope_sin, q, k, vt, precision == SAMAUDIO_BF16, batch,
                                     frames, frames_padded, heads, eps, (hipStream_t)stream), "qkv_prep");
}

int samaudio_op_self_attention(const void* q, const void* k, const void* vt, const uint8_t* key_mask, void* out,
                               int precision, int batch, int frames, int frames_padded, int heads,
                               samaudio_stream stream) {
  if (frames_padded % 64 || frames_padded < frames) return bad("self_attention: frames_padded");
  if (precision == 2)   // fp32 tensors, both contractions on hi/lo-split operands (SAMAUDIO_X3_ATTENTION)
    return hip_ret(sa::launch_self_attention_x3((const float*)q, (const float*)k, (const float*)vt, key_mask, (float*)out, batch, frames,
                                                frames_padded, heads, 128, (hipStream_t)stream), "self_attention_x3");
  return hip_ret(sa::launch_self_attention(q, k, vt, key_mask, out, precision == SAMAUDIO_BF16, batch, frames,
                                           frames_padded, heads, (hipStream_t)stream), "self_attention");
}

int samaudio_op_cross_attention(const void* q, const float* q_w, void* kv, const float* k_w, const uint8_t* mask,
                                void* out, int precision, int batch, int frames, int text_len, int heads, float eps,
                                samaudio_stream stream) {
  const bool bf16 = precision == SAMAUDIO_BF16;
  hipError_t e = sa::launch_headnorm(kv, k_w, bf16, batch * text_len, 2L * heads * 128, 0, heads, eps,
                                     (hipStream_t)stream);
  if (e != hipSuccess) return hip_ret(e, "headnorm");
  return hip_ret(sa::launch_cross_attention(q, q_w, kv, 2L * heads * 128, mask, out, bf16, batch, frames, text_len, heads, eps,
                                            (hipStream_t)stream), "cross_attention");
}

int samaudio_op_cross_attn_fold(const void* wo, const void* kv, int64_t kv_ld, void* ut, int kp, int batch, int text_len,
                                int ltp, int heads, samaudio_stream stream) {
  if (text_len > 16 || (ltp != 8 && ltp != 16) || kp % 64 || kp < heads * ltp) return bad("cross_attn_fold: shape");
  return hip_ret(sa::launch_cross_attn_fold(wo, kv, kv_ld, ut, kp, batch, text_len, ltp, heads, (hipStream_t)stream),
                 "cross_attn_fold");
}

int samaudio_op_layernorm_accum(const float* x, const float* w, const float* b, const float* gate, float* acc,
                                int rows, int dim, float eps, samaudio_stream stream) {
  return hip_ret(sa::launch_layernorm_accum(x, w, b, gate, acc, rows, dim, eps, (hipStream_t)stream),
                 "layernorm_accum");
}

int samaudio_op_masked_groupnorm_silu(const float* x, const float* w, const float* b, const uint8_t* mask,
                                      void* partials_f64, void* out, int precision, int batch, int frames,
                                      int channels, int halo, float eps, samaudio_stream stream) {
  if (channels % 4 || !mask) return bad("masked_groupnorm: channels % 4 / null mask");
  return hip_ret(sa::launch_masked_groupnorm_silu(x, w, b, mask, (double*)partials_f64, out, precision == SAMAUDIO_BF16,
                                                  batch, frames, channels, halo, eps, (hipStream_t)stream),
                 "masked_groupnorm_silu");
}

int samaudio_op_layernorm_rows(const float* x, int64_t x_ld, const float* w, const float* b, float* out_f32,
                               void* out_act, int precision, int64_t rows, int dim, float eps, samaudio_stream stream) {
  if (dim % 4 || x_ld % 4) return bad("layernorm_rows: dim % 4");
  return hip_ret(sa::launch_layernorm_rows(x, x_ld, w, b, out_f32, out_act, precision == SAMAUDIO_BF16, rows, dim, eps,
                                           (hipStream_t)stream), "layernorm_rows");
}

int samaudio_op_split3(const float* x, int64_t x_ld, void* out, int64_t rows, int k, samaudio_stream stream) {
  if (!x || !out || rows <= 0 || k <= 0 || k % 8 || x_ld % 4) return bad("split3: k % 8, x_ld % 4");
  return hip_ret(sa::launch_split3(x, x_ld, out, rows, k, (hipStream_t)stream), "split3");
}

// ---- Judge reranker ----------------------------------------------------------------------------------
int samaudio_judge_create(const samaudio_judge_config* cfg, samaudio_judge** out) {
  if (!cfg || !out) return bad("samaudio_judge_create: null argument");
  if (cfg->precision != SAMAUDIO_F32 && cfg->precision != SAMAUDIO_BF16) return bad("samaudio_judge_create: precision");
  samaudio_judge* j = new samaudio_judge;
  j->judge = new sa::Judge(*cfg);
  *out = j;
  return SAMAUDIO_OK;
}

void samaudio_judge_destroy(samaudio_judge* j) {
  if (!j) return;
  delete j->judge;
  delete j;
}

int samaudio_judge_set_tensor(samaudio_judge* j, const char* name, const void* data, int dtype, int ndim,
                              const int64_t* shape) {
  if (!j) return bad("null judge");
  return ret(j->judge->set_tensor(name, data, dtype, ndim, shape));
}

int samaudio_judge_finalize(samaudio_judge* j) {
  if (!j) return bad("null judge");
  return ret(j->judge->finalize());
}

size_t samaudio_judge_workspace_bytes(samaudio_judge* j, int inputs, int candidates, int frames) {
  if (!j) return 0;
  return j->judge->workspace_bytes(inputs, candidates, frames);
}

int samaudio_judge_set_workspace(samaudio_judge* j, void* workspace, size_t bytes) {
  if (!j) return bad("null judge");
  return ret(j->judge->set_workspace(workspace, bytes));
}

int samaudio_judge_score(samaudio_judge* j, const float* input_latent, const float* separated_latent, int inputs,
                         int candidates, int frames, const float* text_pooled, const uint8_t* pad_mask, float* scores,
                         samaudio_stream stream) {
  if (!j) return bad("null judge");
  return ret(j->judge->score(input_latent, separated_latent, inputs, candidates, frames, text_pooled, pad_mask, scores,
                             (hipStream_t)stream));
}

int samaudio_judge_encode(samaudio_judge* j, int which, const float* x, const uint8_t* pad_mask, int rows, int frames,
                          float* hidden, samaudio_stream stream) {
  if (!j) return bad("null judge");
  return ret(j->judge->encode(which, x, pad_mask, rows, frames, hidden, (hipStream_t)stream));
}

// ---- PE-A-Frame span predictor -------------------------------------------------------------------------
int samaudio_frame_create(const samaudio_frame_config* cfg, samaudio_frame** out) {
  if (!cfg || !out) return bad("samaudio_frame_create: null argument");
  if (cfg->precision != SAMAUDIO_F32 && cfg->precision != SAMAUDIO_BF16) return bad("samaudio_frame_create: precision");
  samaudio_frame* f = new samaudio_frame;
  f->frame = new sa::FramePredictor(*cfg);
  *out = f;
  return SAMAUDIO_OK;
}

void samaudio_frame_destroy(samaudio_frame* f) {
  if (!f) return;
  delete f->frame;
  delete f;
}

int samaudio_frame_set_tensor(samaudio_frame* f, const char* name, const void* data, int dtype, int ndim,
                              const int64_t* shape) {
  if (!f) return bad("null frame predictor");
  return ret(f->frame->set_tensor(name, data, dtype, ndim, shape));
}

int samaudio_frame_finalize(samaudio_frame* f) {
  if (!f) return bad("null frame predictor");
  return ret(f->frame->finalize());
}

size_t samaudio_frame_workspace_bytes(samaudio_frame* f, int rows, int frames) {
  if (!f) return 0;
  return f->frame->workspace_bytes(rows, frames);
}

int samaudio_frame_set_workspace(samaudio_frame* f, void* workspace, size_t bytes) {
  if (!f) return bad("null frame predictor");
  return ret(f->frame->set_workspace(workspace, bytes));
}

int samaudio_frame_logits(samaudio_frame* f, const float* codec_features, const float* text_pooled,
                          const uint8_t* pad_mask, int rows, int frames, float* logits, samaudio_stream stream) {
  if (!f) return bad("null frame predictor");
  return ret(f->frame->logits(codec_features, text_pooled, pad_mask, rows, frames, logits, (hipStream_t)stream));
}

}  // extern "C"
