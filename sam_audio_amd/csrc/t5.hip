// T5 prompt encoder: host code that sequences kernels of gemm*.hip / kernels.hip / t5_kernels.hip, and its C entry
// points (include/samaudio.h "text-prompt encoder").  "oracle:" = oracle/t5_oracle.py, the CPU restatement of
// transformers' T5Stack every step below is checked against (itself pinned to T5EncoderModel in tests/).
#include "t5.h"

#include <cstring>

struct samaudio_t5 {
  sa::T5Encoder* enc;
};

namespace sa {

#define SA_TRY(expr)                     \
  do {                                   \
    Status _s = (expr);                  \
    if (!_s.ok()) return _s;             \
  } while (0)
#define SA_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return Status{SAMAUDIO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)}; \
  } while (0)

namespace {
Status tfail(int code, const std::string& m) { return Status{code, m}; }

GemmParams tlin(const void* A, long lda, const void* W, long M, int N, int K) {
  GemmParams p;
  std::memset(&p, 0, sizeof(p));
  p.A = A; p.W = W; p.lda = lda; p.kc = K; p.tap_stride = 0;
  p.M = (int)M; p.N = N; p.K = K; p.nbatch = 1; p.alpha = 1.f; p.rows_per_gate = 1;
  return p;
}
Status tgemm(const GemmParams& p, bool bf16, hipStream_t st) {
  if (const char* why = gemm_check(p, bf16)) return tfail(SAMAUDIO_ERR_ARG, std::string("t5 encoder: ") + why);
  SA_HIP(launch_gemm(p, bf16, st));
  return Status{};
}
}  // namespace

T5Encoder::T5Encoder(const samaudio_t5_config& c)
    : cfg_(c), bf16_(c.precision == SAMAUDIO_BF16), esz_(bf16_ ? 2 : 4),
      at_dtype_(bf16_ ? SAMAUDIO_DT_BF16 : SAMAUDIO_DT_F32), inner_(c.heads * c.d_kv) {}

Status T5Encoder::set_tensor(const char* name, const void* p, int dtype, int ndim, const int64_t* shape) {
  ready_ = false;
  return reg_.set(name, p, dtype, ndim, shape);
}

Status T5Encoder::finalize() {
  const samaudio_t5_config& c = cfg_;
  if (c.vocab <= 0 || c.d_model <= 0 || c.d_kv <= 0 || c.heads <= 0 || c.d_ff <= 0 || c.layers < 0 || c.max_len <= 0)
    return tfail(SAMAUDIO_ERR_ARG, "t5 encoder: non-positive dimension");
  if (c.d_model % 64 || inner_ % 64 || c.d_ff % 64)
    return tfail(SAMAUDIO_ERR_ARG, "t5 encoder: d_model, heads * d_kv and d_ff must be multiples of 64");
  if (c.d_kv > 128) return tfail(SAMAUDIO_ERR_ARG, "t5 encoder: d_kv must be <= 128");
  if (c.max_len > 512) return tfail(SAMAUDIO_ERR_ARG, "t5 encoder: max_len must be <= 512");
  if (c.act != ACT_RELU && c.act != ACT_GELU_TANH)
    return tfail(SAMAUDIO_ERR_ARG, "t5 encoder: act must be 6 (relu) or 7 (gelu_new); gated feed-forward variants are not built");
  const int D = c.d_model, F = c.d_ff, I = inner_;
  const int F32 = SAMAUDIO_DT_F32, AT = at_dtype_;
#define NEEDF(field, name, ...) SA_TRY(reg_.need(name, F32, {__VA_ARGS__}, (const void**)&(field)))
#define NEEDW(field, name, ...) SA_TRY(reg_.need(name, AT, {__VA_ARGS__}, (const void**)&(field)))
  NEEDF(g_.emb, "emb", c.vocab, D);                          // shared.weight
  NEEDF(g_.rel_bias, "rel_bias", c.heads, 2 * c.max_len - 1);  // relative_attention_bias[bucket(k - q)][h], per distance
  NEEDF(g_.final_ln, "final_ln", D);
  layers_.assign(c.layers, LayerW{});
  for (int i = 0; i < c.layers; ++i) {
    const std::string L = "L" + std::to_string(i) + ".";
    LayerW& w = layers_[i];
    NEEDF(w.ln1, L + "ln1", D);
    NEEDW(w.wqkv, L + "wqkv", 3 * I, D);
    NEEDW(w.wo, L + "wo", D, I);
    NEEDF(w.ln2, L + "ln2", D);
    NEEDW(w.wi, L + "wi", F, D);
    NEEDW(w.wo2, L + "wo2", D, F);
  }
#undef NEEDF
#undef NEEDW
  ready_ = true;
  return Status{};
}

void T5Encoder::plan(Bump& b, long M, bool assign) {
  const long D = cfg_.d_model, F = cfg_.d_ff, I = inner_;
  float* h = (float*)b.take((size_t)M * D * 4);
  void* xn = b.take((size_t)M * D * esz_);
  void* qkv = b.take((size_t)M * 3 * I * esz_);
  void* attn = b.take((size_t)M * I * esz_);
  void* u = b.take((size_t)M * F * esz_);
  if (assign) { w_.h = h; w_.xn = xn; w_.qkv = qkv; w_.attn = attn; w_.u = u; }
}

size_t T5Encoder::workspace_bytes(int rows, int tokens) {
  if (rows <= 0 || tokens <= 0) return 0;
  Bump b;
  plan(b, (long)rows * tokens, false);
  return b.used();
}

Status T5Encoder::set_workspace(void* p, size_t bytes) {
  if (!p || (reinterpret_cast<uintptr_t>(p) & 255)) return tfail(SAMAUDIO_ERR_WORKSPACE, "t5 encoder: workspace must be 256-byte aligned");
  ws_ = (char*)p;
  ws_bytes_ = bytes;
  planned_m_ = 0;
  return Status{};
}

Status T5Encoder::encode(const long long* ids, const unsigned char* mask, int rows, int tokens, float* out, hipStream_t st) {
  if (!ready_) return tfail(SAMAUDIO_ERR_STATE, "t5 encoder: weights not finalized");
  if (!ids || !mask || !out || rows <= 0 || tokens <= 0) return tfail(SAMAUDIO_ERR_ARG, "t5 encoder: bad argument");
  if (tokens > cfg_.max_len)
    return tfail(SAMAUDIO_ERR_ARG, "t5 encoder: " + std::to_string(tokens) + " tokens exceed max_len " + std::to_string(cfg_.max_len));
  if (!ws_) return tfail(SAMAUDIO_ERR_WORKSPACE, "t5 encoder: no workspace");
  const long M = (long)rows * tokens;
  if (planned_m_ != M) {
    Bump b(ws_, ws_bytes_);
    plan(b, M, true);
    if (!b.fits()) return tfail(SAMAUDIO_ERR_WORKSPACE, "t5 encoder: workspace too small for " + std::to_string(M) + " token rows");
    planned_m_ = M;
  }
  const samaudio_t5_config& c = cfg_;
  const int D = c.d_model, F = c.d_ff, I = inner_;
  const float eps = c.ln_eps;
  auto rms = [&](const float* w, void* dst, bool as_act) {  // oracle: t5_layer_norm
    return launch_rmsnorm_mod(w_.h, w, nullptr, nullptr, nullptr, 0, 0, 0, dst, as_act && bf16_, (int)M, D, 1, eps, st);
  };
  SA_HIP(launch_t5_embed(ids, g_.emb, w_.h, M, D, c.vocab, st));  // oracle: shared(input_ids); T5 does not scale it
  for (int l = 0; l < c.layers; ++l) {  // oracle: T5Block = T5LayerSelfAttention + T5LayerFF
    const LayerW& w = layers_[l];
    SA_HIP(rms(w.ln1, w_.xn, true));
    {
      GemmParams p = tlin(w_.xn, D, w.wqkv, M, 3 * I, D);
      p.out_act = w_.qkv; p.act_ld = 3L * I;
      SA_TRY(tgemm(p, bf16_, st));
    }
    SA_HIP(launch_t5_attention(w_.qkv, mask, g_.rel_bias, w_.attn, bf16_, rows, tokens, c.heads, c.d_kv, c.max_len, 1.f, 0, st));
    {
      GemmParams p = tlin(w_.attn, I, w.wo, M, D, I);  // h = h + o(attn)
      p.res = w_.h; p.res_ld = D;
      p.out_f32 = w_.h; p.f32_ld = D;
      SA_TRY(tgemm(p, bf16_, st));
    }
    SA_HIP(rms(w.ln2, w_.xn, true));
    {
      GemmParams p = tlin(w_.xn, D, w.wi, M, F, D);  // act(wi(x))
      p.act = c.act;
      p.out_act = w_.u; p.act_ld = F;
      SA_TRY(tgemm(p, bf16_, st));
      p = tlin(w_.u, F, w.wo2, M, D, F);  // h = h + wo(...)
      p.res = w_.h; p.res_ld = D;
      p.out_f32 = w_.h; p.f32_ld = D;
      SA_TRY(tgemm(p, bf16_, st));
    }
  }
  SA_HIP(rms(g_.final_ln, out, false));  // oracle: final_layer_norm -> last_hidden_state (f32)
  return Status{};
}

}  // namespace sa

namespace {
int tret(const sa::Status& s) {
  if (!s.ok()) sa::set_last_error(s.msg);
  return s.code;
}
int tbad(const char* msg) {
  sa::set_last_error(msg);
  return SAMAUDIO_ERR_ARG;
}
}  // namespace

extern "C" {

int samaudio_t5_create(const samaudio_t5_config* cfg, samaudio_t5** out) {
  if (!cfg || !out) return tbad("samaudio_t5_create: null argument");
  if (cfg->precision != SAMAUDIO_F32 && cfg->precision != SAMAUDIO_BF16) return tbad("samaudio_t5_create: precision");
  samaudio_t5* t = new samaudio_t5;
  t->enc = new sa::T5Encoder(*cfg);
  *out = t;
  return SAMAUDIO_OK;
}

void samaudio_t5_destroy(samaudio_t5* t) {
  if (!t) return;
  delete t->enc;
  delete t;
}

int samaudio_t5_set_tensor(samaudio_t5* t, const char* name, const void* data, int dtype, int ndim, const int64_t* shape) {
  if (!t) return tbad("null t5 encoder");
  return tret(t->enc->set_tensor(name, data, dtype, ndim, shape));
}

int samaudio_t5_finalize(samaudio_t5* t) {
  if (!t) return tbad("null t5 encoder");
  return tret(t->enc->finalize());
}

size_t samaudio_t5_workspace_bytes(samaudio_t5* t, int rows, int tokens) {
  if (!t) return 0;
  return t->enc->workspace_bytes(rows, tokens);
}

int samaudio_t5_set_workspace(samaudio_t5* t, void* workspace, size_t bytes) {
  if (!t) return tbad("null t5 encoder");
  return tret(t->enc->set_workspace(workspace, bytes));
}

int samaudio_t5_encode(samaudio_t5* t, const int64_t* input_ids, const unsigned char* attention_mask, int rows, int tokens,
                       float* last_hidden_state, samaudio_stream stream) {
  if (!t) return tbad("null t5 encoder");
  return tret(t->enc->encode((const long long*)input_ids, attention_mask, rows, tokens, last_hidden_state, (hipStream_t)stream));
}

}  // extern "C"
