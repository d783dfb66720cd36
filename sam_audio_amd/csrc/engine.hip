// Engine: sequences the HIP kernels of SAMAudio.separate() (reference sam_audio/model/model.py:247-338).
// Host code only - every arithmetic step is a kernel from gemm.hip / kernels.hip / attention.hip.
#include "engine.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>

namespace sa {

#define SA_TRY(expr)                     \
  do {                                   \
    Status _s = (expr);                  \
    if (!_s.ok()) return _s;             \
  } while (0)
#define SA_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess)                                                             \
      return Status{SAMAUDIO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)}; \
  } while (0)

static Status fail(int code, const std::string& m) { return Status{code, m}; }

// SAMAUDIO_TRACE=1: after each stage of an evaluation, synchronise, copy the stage's buffer to the host and print how
// many values are non-finite plus the largest magnitude (debugging aid; never on in timed runs).
static bool trace_on() {
  static const bool on = std::getenv("SAMAUDIO_TRACE") != nullptr;
  return on;
}
// SAMAUDIO_TRACE_HASH=1: the same stages, WITHOUT synchronising: a checksum kernel per stage writes one 64-bit word per batch
// item into a debug buffer on the launch stream; ode_solve / forward print them afterwards ("[samaudio hash] <context>
// <sequence> <stage> item <b> <hex>").  Two runs whose rows must agree (one stream vs two streams, whole batch vs shards)
// are compared stage by stage offline (tools/diag_hash.py): the first stage whose checksums differ names the kernel.
// Debugging aid; the only place the library allocates device memory (hipMalloc of 8 MiB, on first use).
struct HashTrace {
  struct Rec { std::string name; int items; size_t slot; };
  unsigned long long* dev = nullptr;
  size_t used = 0, seq = 0;
  std::vector<Rec> recs;
  static constexpr size_t CAP = 1 << 20;
  bool full_warned = false;
};
// eval_field sets the running context's recorder for the duration of ONE evaluation: cleared on every way out, so that a later
// trace() on this thread (the codec, another context) cannot append to this context's recorder
struct HashScope {
  explicit HashScope(HashTrace* h, int items);
  ~HashScope();
};
static bool hash_on() {
  static const bool on = std::getenv("SAMAUDIO_TRACE_HASH") != nullptr;
  return on;
}
static thread_local HashTrace* g_hash = nullptr;   // the running context's recorder (set by eval_field)
static thread_local int g_hash_items = 1;
HashScope::HashScope(HashTrace* h, int items) { g_hash = h; g_hash_items = items; }
HashScope::~HashScope() { g_hash = nullptr; g_hash_items = 1; }
static void hash_stage(const char* name, const void* dev, size_t bytes, hipStream_t st) {
  HashTrace* h = g_hash;
  if (!h || !dev || bytes < 4) return;
  if (!h->dev && !(h->dev = (unsigned long long*)debug_device_alloc(HashTrace::CAP * 8))) return;
  int items = g_hash_items;
  if (items <= 0 || (bytes / 4) % (size_t)items) items = 1;
  if (h->used + items > HashTrace::CAP) {
    if (!h->full_warned) std::fprintf(stderr, "[samaudio hash] recorder full (%zu words): later stages are NOT recorded\n", HashTrace::CAP);
    h->full_warned = true;
    return;
  }
  (void)launch_hash_items((const unsigned*)dev, bytes / 4 / items, items, h->dev + h->used, st);
  h->recs.push_back({name, items, h->used});
  h->used += items;
}
static void hash_flush(HashTrace* h, const void* ctx, hipStream_t st) {
  if (!h || !h->dev || h->recs.empty()) return;
  (void)hipStreamSynchronize(st);
  std::vector<unsigned long long> host(h->used);
  if (hipMemcpy(host.data(), h->dev, h->used * 8, hipMemcpyDeviceToHost) == hipSuccess)
    for (const auto& r : h->recs) {
      for (int b = 0; b < r.items; ++b)
        std::fprintf(stderr, "[samaudio hash] %p %zu %s item %d of %d %016llx\n", ctx, h->seq, r.name.c_str(), b, r.items,
                     host[r.slot + b]);
      ++h->seq;
    }
  h->recs.clear();
  h->used = 0;
}

static void trace(const char* name, const void* dev, size_t count, bool is_bf16, hipStream_t st) {
  if (hash_on()) hash_stage(name, dev, count * (is_bf16 ? 2 : 4), st);
  if (!trace_on() || !dev || !count) return;
  (void)hipStreamSynchronize(st);
  std::vector<unsigned char> host(count * (is_bf16 ? 2 : 4));
  if (hipMemcpy(host.data(), dev, host.size(), hipMemcpyDeviceToHost) != hipSuccess) return;
  size_t bad = 0;
  double mx = 0.0;
  for (size_t i = 0; i < count; ++i) {
    float v;
    if (is_bf16) {
#ifdef SA_OPERAND_FP16
      const unsigned short hw = ((const unsigned short*)host.data())[i];
      const int e = (hw >> 10) & 31, m = hw & 1023;
      const float mag = e == 31 ? (m ? NAN : INFINITY) : (e ? std::ldexp(1.f + m / 1024.f, e - 15) : std::ldexp(m / 1024.f, -14));
      v = (hw & 0x8000) ? -mag : mag;
#else
      const unsigned u = (unsigned)((const unsigned short*)host.data())[i] << 16;
      std::memcpy(&v, &u, 4);
#endif
    } else {
      v = ((const float*)host.data())[i];
    }
    if (!std::isfinite(v)) ++bad;
    else if (std::fabs(v) > mx) mx = std::fabs(v);
  }
  std::fprintf(stderr, "[samaudio trace] %-22s n=%zu non-finite=%zu max|x|=%.4g\n", name, count, bad, mx);
}
static long round_up(long v, long m) { return (v + m - 1) / m * m; }

Engine::Engine(const samaudio_config& c) : cfg_(c) {
  bf16_ = c.precision == SAMAUDIO_BF16;
  esz_ = bf16_ ? 2 : 4;
  at_dtype_ = bf16_ ? SAMAUDIO_DT_BF16 : SAMAUDIO_DT_F32;
  std::memset(&g_, 0, sizeof(g_));
  std::memset(&g32_, 0, sizeof(g32_));
  std::memset(&g3_, 0, sizeof(g3_));
  std::memset(&enc_, 0, sizeof(enc_));
  std::memset(&dec_, 0, sizeof(dec_));
  std::memset(&d_, 0, sizeof(d_));
}

Status Engine::set_tensor(const char* name, const void* p, int dtype, int ndim, const int64_t* shape) {
  if (!name || !p || ndim < 0 || ndim > 4) return fail(SAMAUDIO_ERR_ARG, "set_tensor: bad argument");
  if ((reinterpret_cast<uintptr_t>(p) & 15) != 0)
    return fail(SAMAUDIO_ERR_ARG, std::string("set_tensor: ") + name + " is not 16-byte aligned");
  TensorRef t;
  t.p = p;
  t.dtype = dtype;
  t.shape.assign(shape, shape + ndim);
  // Re-registering a name invalidates the resolved pointers: finalize() must run again.  Adding new names
  // (e.g. the codec set after the DiT set) leaves an already finalized set valid.
  if (tensors_.count(name)) dit_ready_ = codec_ready_ = enc_ready_ = false;
  tensors_[name] = t;
  return Status{};
}

const TensorRef* Engine::find(const std::string& name) const {
  auto it = tensors_.find(name);
  return it == tensors_.end() ? nullptr : &it->second;
}

Status Engine::need(const std::string& name, int dtype, std::vector<int64_t> shape, const void** out) {
  const TensorRef* t = find(name);
  if (!t) return fail(SAMAUDIO_ERR_WEIGHT, "missing weight tensor '" + name + "'");
  if (t->dtype != dtype) return fail(SAMAUDIO_ERR_WEIGHT, "weight '" + name + "' has the wrong dtype");
  if (t->shape != shape) {
    std::string s = "weight '" + name + "' has shape [";
    for (auto v : t->shape) s += std::to_string(v) + ",";
    s += "] expected [";
    for (auto v : shape) s += std::to_string(v) + ",";
    return fail(SAMAUDIO_ERR_WEIGHT, s + "]");
  }
  *out = t->p;
  return Status{};
}

const void* Engine::opt(const std::string& name, std::vector<int64_t> shape) const {
  const TensorRef* t = find(name);
  return t && t->dtype == SAMAUDIO_DT_F32 && t->shape == shape ? t->p : nullptr;
}

Status Engine::need_w5(const std::string& name, int N, int K, const void** out, int* ktm_bits, int bit) {
  const TensorRef* t = find(name);
  if (t && bf16_ && t->dtype == at_dtype_ && t->shape == std::vector<int64_t>{K / 64, N, 64} && K % 64 == 0) {
    *out = t->p;
    *ktm_bits |= 1 << bit;
    return Status{};
  }
  return need(name, at_dtype_, {N, K}, out);
}

static int kpad(int k, bool bf16) { return (int)round_up(k, bf16 ? 64 : 32); }

Status Engine::finalize(int what) {
  const int D = cfg_.dim, F = cfg_.ffn_hidden, L = cfg_.n_layers, C2 = cfg_.latent_channels;
  const int F32 = SAMAUDIO_DT_F32, AT = at_dtype_;
#define NEEDF(field, name, ...) SA_TRY(need(name, F32, {__VA_ARGS__}, (const void**)&(field)))
#define NEEDW(field, name, ...) SA_TRY(need(name, AT, {__VA_ARGS__}, (const void**)&(field)))
  if (what == 0) {
    // head_dim = dim / n_heads: 128 (every kernel tuned for it) or 64 (general forms of qkv_prep / the norms / cross-attention,
    // the self-attention kernel's 64-wide instantiation, no folded cross-attention projection)
    if (D % 256 || cfg_.n_heads <= 0 || D % cfg_.n_heads || (D / cfg_.n_heads != 128 && D / cfg_.n_heads != 64))
      return fail(SAMAUDIO_ERR_ARG, "dim must be a multiple of 256 and dim / n_heads (the head width) 128 or 64");
    const int hd = D / cfg_.n_heads;
    if (F % 64 || C2 % 64 || cfg_.text_dim % 64 || cfg_.video_dim % 64 || cfg_.freq_dim % 64 || cfg_.anchor_dim % 64)
      return fail(SAMAUDIO_ERR_ARG, "channel widths must be multiples of 64");
    layers_.assign(L, LayerW{});
    for (int i = 0; i < L; ++i) {
      const std::string P = "L" + std::to_string(i) + ".";
      LayerW& w = layers_[i];
      NEEDF(w.attn_norm, P + "attn_norm", D);
      NEEDF(w.ffn_norm, P + "ffn_norm", D);
      NEEDF(w.mod_table, P + "mod_table", 6, D);
      NEEDF(w.q_norm, P + "q_norm", hd);
      NEEDF(w.k_norm, P + "k_norm", hd);
      NEEDF(w.c_q_norm, P + "c_q_norm", hd);
      SA_TRY(need_w5(P + "wqkv", 3 * D, D, &w.wqkv, &w.ktm, 0));
      SA_TRY(need_w5(P + "wo", D, D, &w.wo, &w.ktm, 1));
      SA_TRY(need_w5(P + "c_wq", D, D, &w.c_wq, &w.ktm, 2));
      NEEDW(w.c_wo, P + "c_wo", D, D);
      SA_TRY(need_w5(P + "w13", 2 * F, D, &w.w13, &w.ktm, 3));
      SA_TRY(need_w5(P + "w2", D, F, &w.w2, &w.ktm, 4));
      if (!bf16_) {   // SAMAUDIO_OPT_X3_CLASSES: optional split copies, checked when a class is switched on / at the end of finalize
        const struct { const char* leaf; int N, K; const void** out; } x3w[6] = {
            {"wqkv", 3 * D, D, &w.wqkv3}, {"wo", D, D, &w.wo3}, {"c_wq", D, D, &w.c_wq3},
            {"c_wo", D, D, &w.c_wo3},     {"w13", 2 * F, D, &w.w13_3}, {"w2", D, F, &w.w2_3}};
        for (int j = 0; j < 6; ++j) {
          const TensorRef* t = find(P + x3w[j].leaf + ".x3");
          *x3w[j].out = nullptr;
          if (!t || t->dtype != SAMAUDIO_DT_BF16) continue;
          const int64_t N = x3w[j].N, K3 = 3L * x3w[j].K;
          if (t->shape == std::vector<int64_t>{K3 / 64, N, 64}) { *x3w[j].out = t->p; w.ktm3 |= 1 << j; }
          else if (t->shape == std::vector<int64_t>{N, K3}) *x3w[j].out = t->p;
        }
      }
    }
    NEEDF(g_.final_table, "final_table", 2, D);
    NEEDF(g_.final_norm, "final_norm", D);
    NEEDW(g_.w_out, "w_out", C2, D);
    NEEDF(g_.gn1_w, "patch1.gn_w", D);
    NEEDF(g_.gn1_b, "patch1.gn_b", D);
    NEEDW(g_.pw1, "patch1.w", D, 3 * D);
    NEEDF(g_.pb1, "patch1.b", D);
    NEEDF(g_.gn2_w, "patch2.gn_w", D);
    NEEDF(g_.gn2_b, "patch2.gn_b", D);
    NEEDW(g_.pw2, "patch2.w", D, 3 * D);
    NEEDF(g_.pb2, "patch2.b", D);
    NEEDW(g_.y_w13, "y_w13", 2 * D, D);
    NEEDW(g_.y_w2, "y_w2", D, D);
    NEEDW(g_.t_w13, "t_w13", 2 * D, cfg_.freq_dim);
    NEEDW(g_.t_w2, "t_w2", D, D);
    NEEDW(g_.tb_w, "tb_w", 6 * D, D);
    NEEDF(g_.tb_b, "tb_b", 6 * D);
    NEEDF(g_.t_freqs, "t_freqs", cfg_.freq_dim / 2);
    NEEDF(g_.mem_inv_freq, "mem_inv_freq", D / 2);
    NEEDF(g_.rope_cos, "rope_cos", cfg_.max_positions, hd / 2);
    NEEDF(g_.rope_sin, "rope_sin", cfg_.max_positions, hd / 2);
    NEEDW(g_.proj_wy, "proj_wy", D, C2);
    NEEDW(g_.proj_wf, "proj_wf", D, C2);
    NEEDF(g_.proj_b, "proj_b", D);
    NEEDW(g_.mem_w, "mem_w", D, cfg_.text_dim);
    NEEDF(g_.mem_b, "mem_b", D);
    NEEDW(g_.vid_w, "vid_w", D, cfg_.video_dim);
    NEEDF(g_.vid_b, "vid_b", D);
    NEEDF(g_.vid_ln_w, "vid_ln_w", D);
    NEEDF(g_.vid_ln_b, "vid_ln_b", D);
    NEEDF(g_.vid_gate, "vid_gate", 1);
    NEEDF(g_.anc_emb, "anc_emb", cfg_.anchor_vocab, cfg_.anchor_dim);
    NEEDW(g_.anc_w, "anc_w", D, cfg_.anchor_dim);
    // cross-attention K|V projections of ALL layers as one operand: the text memory changes with t only through
    // the y-embedder, so one GEMM per evaluation serves the 22 layers (reference transformer.py:382-388, :102-114)
    NEEDW(g_.c_wkv_all, "c_wkv_all", (int64_t)L * 2 * D, D);
    NEEDF(g_.c_k_norm_all, "c_k_norm_all", L, hd);
    if (bf16_) {  // fp32 copies for SAMAUDIO_OPT_F32_CLASSES: optional, checked when a class is switched on / used
#define OPTF(field, name, ...) g32_.field = (const float*)opt(name ".f32", {__VA_ARGS__})
      OPTF(w_out, "w_out", C2, D);
      OPTF(t_w13, "t_w13", 2 * D, cfg_.freq_dim);
      OPTF(t_w2, "t_w2", D, D);
      OPTF(tb_w, "tb_w", 6 * D, D);
      OPTF(proj_wy, "proj_wy", D, C2);
      OPTF(proj_wf, "proj_wf", D, C2);
      OPTF(mem_w, "mem_w", D, cfg_.text_dim);
      OPTF(vid_w, "vid_w", D, cfg_.video_dim);
      OPTF(anc_w, "anc_w", D, cfg_.anchor_dim);
      OPTF(y_w13, "y_w13", 2 * D, D);
      OPTF(y_w2, "y_w2", D, D);
#undef OPTF
      SA_TRY(check_f32_weights(f32_classes_));
    }
    if (!bf16_) {   // SAMAUDIO_OPT_X3_CLASSES, classes PATCH / CKV: optional split copies
      std::memset(&g3_, 0, sizeof(g3_));
      const struct { const char* name; int64_t N, K3; const void** out; } x3g[3] = {
          {"patch1.w.x3", D, 9L * D, &g3_.pw1}, {"patch2.w.x3", D, 9L * D, &g3_.pw2}, {"c_wkv_all.x3", (int64_t)L * 2 * D, 3L * D, &g3_.c_wkv_all}};
      for (int j = 0; j < 3; ++j) {
        const TensorRef* t = find(x3g[j].name);
        if (!t || t->dtype != SAMAUDIO_DT_BF16) continue;
        if (t->shape == std::vector<int64_t>{x3g[j].K3 / 64, x3g[j].N, 64}) { *x3g[j].out = t->p; g3_.ktm |= 1 << j; }
        else if (t->shape == std::vector<int64_t>{x3g[j].N, x3g[j].K3}) *x3g[j].out = t->p;
      }
    }
    dit_ready_ = true;   // (check_x3_weights looks at the resolved layers)
    if (const Status s3 = check_x3_weights(x3_classes_); !s3.ok()) { dit_ready_ = false; return s3; }
  } else {
    const int CD = cfg_.codec_dim, CL = cfg_.codec_latent;
    auto res_units = [&](const std::string& P, StageW& s, int C) -> Status {
      for (int j = 0; j < 3; ++j) {
        const std::string R = P + "r" + std::to_string(j) + ".";
        ResUnitW& r = s.r[j];
        r.k1pad = kpad(7 * C, bf16_);
        r.k2pad = kpad(C, bf16_);
        NEEDF(r.a1, R + "a1", C);
        NEEDW(r.w1, R + "w1", C, r.k1pad);
        NEEDF(r.b1, R + "b1", C);
        NEEDF(r.a2, R + "a2", C);
        NEEDW(r.w2, R + "w2", C, r.k2pad);
        NEEDF(r.b2, R + "b2", C);
      }
      return Status{};
    };
    // encoder
    NEEDW(enc_.in_w, "enc.in.w", cfg_.enc_dim, 64);
    NEEDF(enc_.in_b, "enc.in.b", cfg_.enc_dim);
    int C = cfg_.enc_dim;
    for (int i = 0; i < 4; ++i) {
      const std::string P = "enc.s" + std::to_string(i) + ".";
      const int s = cfg_.enc_rates[i];
      if (s % 2) return fail(SAMAUDIO_ERR_ARG, "codec strides must be even");
      SA_TRY(res_units(P, enc_.s[i], C));
      NEEDF(enc_.s[i].a, P + "a", C);
      NEEDW(enc_.s[i].w, P + "down.w", 2 * C, 2 * s * C);
      NEEDF(enc_.s[i].b, P + "down.b", 2 * C);
      C *= 2;
    }
    NEEDF(enc_.out_a, "enc.out.a", C);
    NEEDW(enc_.out_w, "enc.out.w", CL, 3 * C);
    NEEDF(enc_.out_b, "enc.out.b", CL);
    NEEDW(enc_.proj_w, "enc.proj.w", CD, CL);
    NEEDF(enc_.proj_b, "enc.proj.b", CD);
    enc_ready_ = true;
    // SAMAUDIO_OPT_X3_CLASSES bit CODEC: "<name>.x3" twins of registered codec weights, keyed by the weight's own pointer
    x3_codec_.clear();
    if (!bf16_)
      for (const auto& kv : tensors_) {
        const std::string& name = kv.first;
        if (name.size() < 4 || name.compare(name.size() - 3, 3, ".x3") != 0 || (name.rfind("enc.", 0) != 0 && name.rfind("dec.", 0) != 0)) continue;
        const TensorRef* base = find(name.substr(0, name.size() - 3));
        const TensorRef& t = kv.second;
        if (!base || base->shape.size() != 2 || t.dtype != SAMAUDIO_DT_BF16 || t.shape.size() != 3 || t.shape[2] % 3) continue;
        const int64_t cin = t.shape[2] / 3;
        if (t.shape[0] != base->shape[0] || t.shape[1] * cin != base->shape[1] || cin % 8) continue;
        x3_codec_[base->p] = X3CodecW{t.p, (int)cin};
      }
    // ... and "<name>.fly" twins (weights.py convert_codec_fly16): the narrow convolutions' weights already split, in the layout the
    // fp32 kernel's on-the-fly multiply reads (common.h GEMM_FLAG_W_FLY16)
    fly_codec_.clear();
    if (!bf16_)
      for (const auto& kv : tensors_) {
        const std::string& name = kv.first;
        if (name.size() < 5 || name.compare(name.size() - 4, 4, ".fly") != 0 || (name.rfind("enc.", 0) != 0 && name.rfind("dec.", 0) != 0)) continue;
        const TensorRef* base = find(name.substr(0, name.size() - 4));
        const TensorRef& t = kv.second;
        if (!base || base->shape.size() != 2 || t.dtype != SAMAUDIO_DT_BF16 || t.shape.size() != 2) continue;
        if (t.shape[0] != base->shape[0] || t.shape[1] != 2 * base->shape[1] || base->shape[1] % 32) continue;
        fly_codec_[base->p] = t.p;
      }
    if (what == 2) return Status{};  // encoder only: the Judge's DACVAEEncoder (reference codec.py:42-78)
    // decoder
    NEEDW(dec_.proj_w, "dec.proj.w", CL, CD);
    NEEDF(dec_.proj_b, "dec.proj.b", CL);
    NEEDW(dec_.in_w, "dec.in.w", cfg_.dec_dim, 7 * CL);
    NEEDF(dec_.in_b, "dec.in.b", cfg_.dec_dim);
    C = cfg_.dec_dim;
    for (int i = 0; i < 4; ++i) {
      const std::string P = "dec.s" + std::to_string(i) + ".";
      const int s = cfg_.dec_rates[i];
      if (s % 2) return fail(SAMAUDIO_ERR_ARG, "codec strides must be even");
      NEEDF(dec_.s[i].a, P + "a", C);
      NEEDW(dec_.s[i].w, P + "up.w", s * (C / 2), 2 * C);
      NEEDF(dec_.s[i].b, P + "up.b", C / 2);
      C /= 2;
      SA_TRY(res_units(P, dec_.s[i], C));
    }
    dec_.out_kpad = kpad(7 * C, bf16_);
    NEEDF(dec_.out_a, "dec.out.a", C);
    NEEDW(dec_.out_w, "dec.out.w", 1, dec_.out_kpad);
    NEEDF(dec_.out_b, "dec.out.b", 1);
    codec_ready_ = true;
  }
#undef NEEDF
#undef NEEDW
  return Status{};
}

// ---------------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------------
Status Engine::plan_dit(Bump& b, int rows, int T, int Lt, bool assign) {
  const long D = cfg_.dim, F = cfg_.ffn_hidden, C2 = cfg_.latent_channels, H = cfg_.n_heads;
  const long M = (long)rows * T, Mt = (long)rows * Lt, Tp = round_up(T, 64);
  const long nt = rows;  // worst case: one time value per row
  auto f32 = [&](long n) { return (float*)b.take((size_t)n * 4); };
  auto act = [&](long n) { return b.take((size_t)n * esz_); };
  auto& d = d_;
  float* ymid = f32(M * C2); float* aligned = f32(M * D); float* cond = f32(M * D); float* h = f32(M * D);
  float* hp1 = f32(M * D); float* text_proj = f32(Mt * D); float* t_emb = f32(nt * D); float* t0 = f32(nt * 6 * D);
  float* tsin = f32(nt * D); float* vtmp = f32(M * D); float* times = f32(4096);
  float* modgs = f32(2L * cfg_.n_layers * nt * 2 * D);   // pre-combined RMSNorm + modulate operands of an evaluation
  void* ybf = act(M * C2); void* xn = act(M * D); void* qkv = act(M * 3 * D);
  void* Q = act((long)rows * Tp * D); void* K = act((long)rows * Tp * D);   // [rows, H, Tp, head_dim]
  void* Vt = act((long)rows * D * Tp);
  void* attn = act(M * D); void* hbf = act(M * D); void* qc = act(M * D); void* ca = act(M * D); void* u = act(M * F);
  void* gnbuf = act((long)rows * (T + 2) * D); void* mem = act(Mt * D); void* yu = act(Mt * D); void* yemb = act(Mt * D);
  void* kvc = act(Mt * 2 * D * cfg_.n_layers); void* temb = act(nt * cfg_.freq_dim); void* tu = act(nt * D); void* tsilu = act(nt * D);
  void* feats = act(M * C2); void* text = act(Mt * cfg_.text_dim); void* video = act(M * cfg_.video_dim);
  void* anch = act(M * cfg_.anchor_dim);
  // fp32 operands of the classes SAMAUDIO_OPT_F32_CLASSES may switch to exact fp32 (16-bit contexts only)
  float *temb32 = nullptr, *tu32 = nullptr, *tsilu32 = nullptr, *xn32 = nullptr, *prep32 = nullptr, *mem32 = nullptr,
        *yu32 = nullptr, *yemb32 = nullptr;
  if (bf16_) {
    temb32 = f32(nt * cfg_.freq_dim); tu32 = f32(nt * D); tsilu32 = f32(nt * D); xn32 = f32(M * D);
    prep32 = f32(M * (cfg_.video_dim > cfg_.anchor_dim ? cfg_.video_dim : cfg_.anchor_dim));
    mem32 = f32(Mt * D); yu32 = f32(Mt * D); yemb32 = f32(Mt * D);
  }
  // folded cross-attention (bf16, Lt <= 16): probabilities [M, KP] and the per-batch operand U^T [rows][D][KP]
  // (the condition prepare() folds under: 16-bit context, short memory, 128-wide heads, not switched off)
  const long ltp = Lt <= 8 ? 8 : 16, kp = round_up(H * ltp, 64);
  const bool fold = bf16_ && Lt <= 16 && D / cfg_.n_heads == 128 && !std::getenv("SAMAUDIO_NO_FOLD");
  void* probs = fold ? act(M * kp) : nullptr;
  // (one slice per layer: the folds of an evaluation run as one launch in front of the layer loop; debug flag 31 - one launch per
  // layer - only ever uses the first slice)
  const bool fold_all = cfg_.n_layers <= kMaxFoldLayers;
  void* ut = fold ? act((long)rows * D * kp * (fold_all ? cfg_.n_layers : 1)) : nullptr;
  // SAMAUDIO_OPT_X3_CLASSES: the split activation operand [lo | hi | hi] of the widest GEMM input (16-bit, 3 K elements per row)
  // (x3a: D-wide operands and the patcher's halo-padded rows; x3u: the SwiGLU hidden, written by the w13 launch while it reads x3a)
  const bool x3g = !bf16_ && (x3_classes_ & ~(SAMAUDIO_X3_ATTENTION | SAMAUDIO_CLS_CODEC));
  void* x3a = x3g ? b.take((size_t)rows * (T + 2) * 3 * (size_t)D * 2) : nullptr;
  void* x3u = x3g ? b.take((size_t)M * 3 * (size_t)F * 2) : nullptr;
  // folded cross-attention on compensated operands (class CWO of an x3 context, short memory, 128-wide heads): probabilities
  // [M][3 kp] and the per-batch operands of all layers [L][rows][D][3 kp]
  const bool fold3 = x3g && (x3_classes_ & SAMAUDIO_CLS_CWO) && Lt <= 16 && D / cfg_.n_heads == 128 && cfg_.n_layers <= kMaxFoldLayers &&
                     !std::getenv("SAMAUDIO_NO_FOLD");
  void* x3p = fold3 ? b.take((size_t)M * 3 * kp * 2) : nullptr;
  void* ut3 = fold3 ? b.take((size_t)rows * D * 3 * kp * 2 * cfg_.n_layers) : nullptr;
  unsigned char* pad_mask = (unsigned char*)b.take((size_t)M);
  unsigned char* text_mask = (unsigned char*)b.take((size_t)Mt);
  double* gn_part = (double*)b.take((size_t)rows * 64 * 2 * 8);
  if (assign) {
    d.ymid = ymid; d.aligned = aligned; d.cond = cond; d.h = h; d.hp1 = hp1; d.text_proj = text_proj; d.t_emb = t_emb;
    d.t0 = t0; d.modgs = modgs; d.tsin = tsin; d.vtmp = vtmp; d.times = times; d.ybf = ybf; d.xn = xn; d.qkv = qkv; d.Q = Q; d.K = K;
    d.Vt = Vt; d.attn = attn; d.hbf = hbf; d.qc = qc; d.ca = ca; d.u = u; d.gnbuf = gnbuf; d.mem = mem; d.yu = yu;
    d.yemb = yemb; d.kvc = kvc; d.temb = temb; d.tu = tu; d.tsilu = tsilu; d.feats = feats; d.text = text;
    d.video = video; d.anch = anch; d.temb32 = temb32; d.tu32 = tu32; d.tsilu32 = tsilu32; d.xn32 = xn32; d.prep32 = prep32;
    d.mem32 = mem32; d.yu32 = yu32; d.yemb32 = yemb32; d.probs = probs; d.ut = ut; d.x3a = x3a; d.x3u = x3u; d.x3p = x3p; d.ut3 = ut3; d.pad_mask = pad_mask; d.text_mask = text_mask; d.gn_part = gn_part;
  }
  return Status{};
}

// per-item element counts of the codec stage buffers (see codec_encode / codec_decode)
static void codec_stage_dims(const samaudio_config& c, int64_t samples, long encT[5], int encC[5], long decT[5],
                             int decC[5]) {
  long T = samples;
  int C = c.enc_dim;
  for (int i = 0; i < 5; ++i) {
    encT[i] = T; encC[i] = C;
    if (i < 4) { T /= c.enc_rates[i]; C *= 2; }
  }
  long hop = 1;
  for (int i = 0; i < 4; ++i) hop *= c.enc_rates[i];
  T = samples / hop;
  C = c.dec_dim;
  for (int i = 0; i < 5; ++i) {
    decT[i] = T; decC[i] = C;
    if (i < 4) { T *= c.dec_rates[i]; C /= 2; }
  }
}

size_t Engine::codec_bytes(int items, int64_t samples) const {
  if (items <= 0) return 0;
  long encT[5], decT[5];
  int encC[5], decC[5];
  codec_stage_dims(cfg_, samples, encT, encC, decT, decC);
  const size_t per_elem = 4 + 2 * esz_;  // raw f32 + act + tmp
  size_t enc = (size_t)(samples + 2 * HALO) * 8 * esz_, dec = 0;
  for (int i = 0; i < 5; ++i) enc += (size_t)(encT[i] + 2 * HALO) * encC[i] * per_elem + 1024;
  enc += (size_t)(encT[4] + 2 * HALO) * cfg_.codec_latent * esz_;
  dec += (size_t)(decT[0] + 2 * HALO) * (cfg_.codec_dim + cfg_.codec_latent) * esz_;
  for (int i = 0; i < 5; ++i) dec += (size_t)(decT[i] + 2 * HALO) * decC[i] * per_elem + 1024;
  return (size_t)items * ((enc > dec ? enc : dec) + codec_x3_per_item(samples)) + (1 << 16);
}

// bytes per waveform of the split activation operand of the widest codec launch that can run as a compensated 16-bit launch
// (gemm_codec_x3): the whole halo buffer a launch reads, 3 x 16 bits per element; 0 unless the option and the twins are there
size_t Engine::codec_x3_per_item(int64_t samples) const {
  if (!x3(SAMAUDIO_CLS_CODEC) || x3_codec_.empty()) return 0;
  long encT[5], decT[5];
  int encC[5], decC[5];
  codec_stage_dims(cfg_, samples, encT, encC, decT, decC);
  size_t m = 0;
  auto see = [&](long T, int C) { const size_t v = (size_t)(T + 2 * HALO) * C * 6 + 256; if (v > m) m = v; };
  for (int i = 0; i < 5; ++i) {
    if (2 * encC[i] >= 256) see(encT[i], encC[i]);   // (the strided convolution out of stage i has 2 C outputs)
    if (decC[i] >= 256) see(decT[i], decC[i]);
  }
  see(decT[0], cfg_.codec_dim);
  see(decT[0], cfg_.codec_latent);
  return m;
}

size_t Engine::workspace_bytes(int rows, int frames, int text_len, int codec_items, int64_t samples) {
  size_t dit = 0;
  if (rows > 0) {
    Bump b;
    plan_dit(b, rows, frames, text_len < 1 ? 1 : text_len, false);
    dit = b.used() + 4096;
  }
  size_t codec = codec_bytes(codec_items, samples);
  return dit > codec ? dit : codec;
}

Status Engine::set_workspace(void* p, size_t bytes) {
  if (!p || (reinterpret_cast<uintptr_t>(p) & 255)) return fail(SAMAUDIO_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  ws_ = (char*)p;
  ws_bytes_ = bytes;
  prepared_ = false;
  return Status{};
}

// algorithmic bytes of one GEMM / implicit-convolution launch: every operand, output and residual element once
static double gemm_alg_bytes(const GemmParams& p, size_t esz) {
  const double rows = (double)p.M * p.nbatch;
  const double n_out = p.swiglu ? p.N / 2 : p.N;
  const long a_row = p.lda > 0 && p.lda < p.K ? p.lda : p.K;  // implicit convolutions re-read a row once per tap
  double b = rows * (double)a_row * esz + (double)p.N * p.K * esz * (p.w_bstride ? p.nbatch : 1);
  if (p.out_f32) b += rows * n_out * 4;
  if (p.out_act) b += rows * n_out * esz;
  if (p.res) b += (p.res_ld ? rows : 1.0) * n_out * 4;
  return b;
}

Status Engine::check_f32_weights(int classes) const {
  const struct { int cls; const float* w; const char* name; } need[] = {
      {SAMAUDIO_CLS_OUT, g32_.w_out, "w_out"},       {SAMAUDIO_CLS_TIME, g32_.t_w13, "t_w13"},   {SAMAUDIO_CLS_TIME, g32_.t_w2, "t_w2"},
      {SAMAUDIO_CLS_TIME, g32_.tb_w, "tb_w"},        {SAMAUDIO_CLS_IN, g32_.proj_wy, "proj_wy"}, {SAMAUDIO_CLS_PREP, g32_.proj_wf, "proj_wf"},
      {SAMAUDIO_CLS_PREP, g32_.mem_w, "mem_w"},      {SAMAUDIO_CLS_PREP, g32_.vid_w, "vid_w"},   {SAMAUDIO_CLS_PREP, g32_.anc_w, "anc_w"},
      {SAMAUDIO_CLS_YEMB, g32_.y_w13, "y_w13"},      {SAMAUDIO_CLS_YEMB, g32_.y_w2, "y_w2"}};
  for (const auto& n : need)
    if ((classes & n.cls) && !n.w)
      return fail(SAMAUDIO_ERR_WEIGHT, std::string("SAMAUDIO_OPT_F32_CLASSES: the fp32 operand copy '") + n.name +
                                           ".f32' of a class that is switched to fp32 is not registered");
  return Status{};
}

Status Engine::check_x3_weights(int classes) const {
  if (!classes) return Status{};
  if ((classes & SAMAUDIO_CLS_PATCH) && !(g3_.pw1 && g3_.pw2))
    return fail(SAMAUDIO_ERR_WEIGHT, "SAMAUDIO_OPT_X3_CLASSES: the split weights 'patch1.w.x3' / 'patch2.w.x3' (16-bit, [D, 9D] or [9D/64, D, 64]) are not registered");
  if ((classes & SAMAUDIO_CLS_CKV) && !g3_.c_wkv_all)
    return fail(SAMAUDIO_ERR_WEIGHT, "SAMAUDIO_OPT_X3_CLASSES: the split weight 'c_wkv_all.x3' (16-bit, [L*2D, 3D] or [3D/64, L*2D, 64]) is not registered");
  for (size_t i = 0; i < layers_.size(); ++i) {
    const LayerW& w = layers_[i];
    const struct { int cls; const void* p; const char* leaf; } need[6] = {
        {SAMAUDIO_CLS_QKV, w.wqkv3, "wqkv"}, {SAMAUDIO_CLS_WO, w.wo3, "wo"},     {SAMAUDIO_CLS_CWQ, w.c_wq3, "c_wq"},
        {SAMAUDIO_CLS_CWO, w.c_wo3, "c_wo"}, {SAMAUDIO_CLS_W13, w.w13_3, "w13"}, {SAMAUDIO_CLS_W2, w.w2_3, "w2"}};
    for (const auto& n : need)
      if ((classes & n.cls) && !n.p)
        return fail(SAMAUDIO_ERR_WEIGHT, "SAMAUDIO_OPT_X3_CLASSES: the split weight 'L" + std::to_string(i) + "." + n.leaf +
                                             ".x3' (16-bit, [N, 3K] or [3K/64, N, 64]) of a class that is switched on is not registered");
  }
  return Status{};
}

Status Engine::set_option(int option, int value) {
  if (option == SAMAUDIO_OPT_TAIL_SPLIT) {
    tail_split_ = value != 0;
    return Status{};
  }
  if (option == SAMAUDIO_OPT_F32_CLASSES) {
    if (value && !bf16_) return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_F32_CLASSES applies to 16-bit contexts (an fp32 context is exact already)");
    if (value & ~SAMAUDIO_CLS_F32_CAPABLE)
      return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_F32_CLASSES: only the classes of SAMAUDIO_CLS_F32_CAPABLE can run in fp32");
    if (dit_ready_) SA_TRY(check_f32_weights(value));   // (before finalize(0): checked there)
    f32_classes_ = value;
    return Status{};
  }
  if (option == SAMAUDIO_OPT_ALT16_CLASSES) {
    if (value && !bf16_) return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_ALT16_CLASSES applies to 16-bit contexts");
    if (value & ~SAMAUDIO_CLS_ALT16_CAPABLE)
      return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_ALT16_CLASSES: only the five big GEMM classes of the DiT layers (qkv, wo, cwq, w13, w2)");
    // (what eval_field needs for these classes: the pre-combined RMSNorm operands - checked here, not at the first evaluation)
    if (value && !(2 * cfg_.n_layers <= kMaxModNorms && cfg_.n_layers > 0 && cfg_.dim <= 256 * 12))
      return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_ALT16_CLASSES (precision 'mixed') supports 1 .. " + std::to_string(kMaxModNorms / 2) +
                                        " layers and dim <= 3072: this config has " + std::to_string(cfg_.n_layers) + " layers, dim " +
                                        std::to_string(cfg_.dim) + " - use precision 'fp16' or 'bf16'");
    alt_classes_ = value;
    return Status{};
  }
  if (option == SAMAUDIO_OPT_PREFETCH_ROWS) {
    if (value && !bf16_) return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_PREFETCH_ROWS applies to 16-bit contexts");
    if (value < 0) return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_PREFETCH_ROWS: a row count (0 = off)");
    prefetch_rows_ = value;
    return Status{};
  }
  if (option == SAMAUDIO_OPT_X3_CLASSES) {
    if (value && bf16_) return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_X3_CLASSES applies to fp32 contexts (compensated 16-bit operands under fp32 storage)");
    if (value & ~SAMAUDIO_CLS_X3_CAPABLE)
      return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_X3_CLASSES: only the six big GEMM classes of the DiT layers (qkv, wo, cwq, cwo, w13, w2), patch, ckv, codec and SAMAUDIO_X3_ATTENTION");
    if (dit_ready_) SA_TRY(check_x3_weights(value));   // (before finalize(0): checked there)
    if (((value & ~(SAMAUDIO_X3_ATTENTION | SAMAUDIO_CLS_CODEC)) != 0) != ((x3_classes_ & ~(SAMAUDIO_X3_ATTENTION | SAMAUDIO_CLS_CODEC)) != 0)) prepared_ = false;   // the scratch operand is part of the workspace plan
    x3_classes_ = value;
    return Status{};
  }
  if (option == SAMAUDIO_OPT_SENTINEL) {
    sentinel_on_ = value != 0;
    return Status{};
  }
  if (option == SAMAUDIO_OPT_QUANT_CLASSES || option == SAMAUDIO_OPT_QUANT_FORMAT) {
    if (value && bf16_) return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_QUANT_*: operand-rounding emulation needs an fp32 context");
    if (option == SAMAUDIO_OPT_QUANT_FORMAT && (value < 0 || value > 2))
      return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_QUANT_FORMAT: 0 (off), 1 (bfloat16) or 2 (fp16)");
    (option == SAMAUDIO_OPT_QUANT_CLASSES ? quant_classes_ : quant_fmt_) = value;
    return Status{};
  }
  return fail(SAMAUDIO_ERR_ARG, "samaudio_set_option: unknown option " + std::to_string(option));
}

Status Engine::sentinel(int slot, const void* x, int fmt, long rows, int cols, long ld, hipStream_t st) {
  if (!sentinel_on_ || !x) return Status{};
  constexpr size_t kFloats = 2 * (SAMAUDIO_SENTINEL_SLOTS + kSentinelPartials);
  if (!sentinel_dev_) {
    sentinel_dev_ = (float*)debug_device_alloc(kFloats * 4);
    if (!sentinel_dev_) return fail(SAMAUDIO_ERR_HIP, "sentinel: device allocation failed");
    SA_HIP(hipMemsetAsync(sentinel_dev_, 0, kFloats * 4, st));
  }
  SA_HIP(launch_sentinel(x, fmt, rows, cols, ld, sentinel_dev_ + 2 * SAMAUDIO_SENTINEL_SLOTS, sentinel_dev_ + 2 * slot, st));
  return Status{};
}

Status Engine::sentinel_read(float* absmax, double* nonfinite, hipStream_t st) {
  for (int i = 0; i < SAMAUDIO_SENTINEL_SLOTS; ++i) { absmax[i] = 0.f; nonfinite[i] = 0.0; }
  if (!sentinel_dev_) return Status{};
  float host[2 * SAMAUDIO_SENTINEL_SLOTS];
  SA_HIP(hipStreamSynchronize(st));
  SA_HIP(hipMemcpy(host, sentinel_dev_, sizeof(host), hipMemcpyDeviceToHost));
  SA_HIP(hipMemsetAsync(sentinel_dev_, 0, sizeof(host), st));
  for (int i = 0; i < SAMAUDIO_SENTINEL_SLOTS; ++i) { absmax[i] = host[2 * i]; nonfinite[i] = host[2 * i + 1]; }
  return Status{};
}

// the 16-bit output of a GEMM launch, scanned into its class's slot (outputs with a window mask - transposed convolutions - have
// rows the launch does not write: skipped)
static int cls_slot(int cls) {
  int bit = 0;
  while (bit < SAMAUDIO_CLS_COUNT - 1 && !(cls & (1 << bit))) ++bit;
  return bit;
}

Status Engine::gemm(const GemmParams& p_in, hipStream_t st, double alg_flops, int cls, int mode) {
  const bool f32 = mode == 1, x3m = mode == 2 || mode == 3;   // mode 3: an x3 launch whose K' is split per input block, not as a whole
  const bool is16 = bf16_ || x3m;   // the launch's operand format (an X3 launch: 16-bit operands inside an fp32 context)
  if (sentinel_on_ && p_in.out_act && !p_in.c_ld_rel && !x3m) {
    sentinel_on_ = false;   // (the launch itself, without recursion)
    const Status s = gemm(p_in, st, alg_flops, cls, mode);
    sentinel_on_ = true;
    if (!s.ok()) return s;
    const int c = prof_cls_[0] == 'c' ? SAMAUDIO_CLS_CODEC : cls;
    const int fmt = (f32 || !bf16_) ? 0 : ((p_in.flags & 512) ? 2 : 1);
    const int n_out = p_in.swiglu ? p_in.N / 2 : p_in.N;
    for (int b = 0; b < p_in.nbatch; ++b) {
      const size_t esz = fmt == 0 ? 4 : 2;
      const char* base = (const char*)p_in.out_act + ((size_t)p_in.act_off + (size_t)b * p_in.act_bstride) * esz;
      SA_TRY(sentinel(cls_slot(c), base, fmt, p_in.M, n_out, p_in.act_ld, st));
    }
    return Status{};
  }
  if (mode == 0 && !bf16_ && prof_cls_[0] == 'c' && p_in.N >= 256 && x3_codec_scratch_ && x3(SAMAUDIO_CLS_CODEC)) {
    const auto it = x3_codec_.find(p_in.W);
    if (it != x3_codec_.end()) {
      const int c = it->second.cin;
      const bool both = p_in.out_f32 && p_in.out_act;
      const bool flat = !p_in.c_ld_rel ? (p_in.out_f32 ? p_in.f32_ld == p_in.N : p_in.act_ld == p_in.N) : true;
      if (p_in.a_bstride > 0 && !p_in.w_bstride && !p_in.swiglu && !p_in.gate && !(p_in.kc % c) && !(p_in.lda % c) && !(p_in.a_off % c) &&
          !(p_in.a_bstride % c) && !(p_in.tap_stride % c) && !(p_in.K % c) && !((3L * p_in.K) % 64) && flat && (!both || !p_in.f32_act) &&
          (size_t)p_in.nbatch * (p_in.a_bstride / c) * 3 * c * 2 <= x3_codec_scratch_bytes_)
        return gemm_codec_x3(p_in, it->second, st, alg_flops);
    }
  }
  GemmParams p = p_in;
  p.tag = prof_cls_[0] == 'c' ? 1 : 0;  // codec launches run under their own kernel symbols
  // bit 1: no tail split (gemm.hip gemm_tail_split); bit 9 (from the caller): 16-bit output in the alt format; bit 10: operands
  // in the alt format (SAMAUDIO_OPT_ALT16_CLASSES, mixed mode)
  // bit 11 (from the caller): W is K-tile-major
  p.flags = (p_in.flags & (512 | GEMM_FLAG_W_KTM | GEMM_FLAG_OUT_SPLIT3)) | (tail_split_ ? 0 : 2) | (alt16(cls) && !f32 ? 1024 : 0);
  if (p.tag) cls = SAMAUDIO_CLS_CODEC;
  // an x3 launch on K-concatenated split operands: let the 8-phase kernels share the operand tiles the three products have in common
  // (common.h GEMM_FLAG_X3_SHARE) wherever the launch qualifies; debug flag 38 = 1: the plain walk over K' (A/B, tests)
  // mode 3 (the convolutions: every Cin-block of K' is its own [hi | lo | hi]) never qualifies
  if (mode == 2 && debug_flag(38) != 1 && (debug_flag(38) < 2 || (cls & (debug_flag(38) >> 1)))) {   // (flag 38 >= 2: class mask << 1, diagnosis)
    GemmParams q = p;
    q.flags |= GEMM_FLAG_X3_SHARE;
    if (q.kc == q.K && q.K % 192 == 0 && !gemm_check(q, true)) p = q;
  }
  if (f32) {  // a class of SAMAUDIO_OPT_F32_CLASSES: exact-fp32 kernel inside a 16-bit context
    if (!p.W) return fail(SAMAUDIO_ERR_WEIGHT, "SAMAUDIO_OPT_F32_CLASSES: the class's \"<name>.f32\" weight copy is not registered");
    if (const char* why = gemm_check(p, false)) return fail(SAMAUDIO_ERR_ARG, why);
    const double flops = alg_flops >= 0 ? alg_flops : 2.0 * p.M * (double)p.N * p.K * p.nbatch;
    return op(gemm_variant_name(gemm_variant(p, false), false), gemm_alg_bytes(p, 4), flops, st,
              [&] { return launch_gemm(p, false, st); });
  }
  if (!is16 && quant_fmt_ && (quant_classes_ & cls)) p.flags |= (quant_fmt_ << 2) | (quant_fmt_ << 4);
  // SAMAUDIO_OPT_X3_CLASSES bit CODEC (fp32 contexts): the convolution multiplies on split operands, split in registers (gemm.hip)
  if (!is16 && p.tag && x3(SAMAUDIO_CLS_CODEC)) {
    p.flags |= GEMM_FLAG_X3_FLY;
    if (!p.w_bstride && !fly_codec_.empty()) {
      const auto it = fly_codec_.find(p.W);
      if (it != fly_codec_.end()) { p.W = it->second; p.flags |= GEMM_FLAG_W_FLY16; }
    }
  }
  if (const char* why = gemm_check(p, is16)) return fail(SAMAUDIO_ERR_ARG, why);
  if (!prof_on_) {
    SA_HIP(launch_gemm(p, is16, st));
    return Status{};
  }
  const double flops = alg_flops >= 0 ? alg_flops : 2.0 * p.M * (double)p.N * p.K * p.nbatch;
  const double bytes = gemm_alg_bytes(p, is16 ? 2 : 4);
  const int full = gemm_tail_split(p, is16);
  const long tiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * p.nbatch;
  for (int part = 0; part < (full ? 2 : 1); ++part) {
    // a split launch (gemm.hip gemm_tail_split) is two kernels: each gets its own record, flops / bytes by tile share
    const double share = !full ? 1.0 : (part == 0 ? (double)full / tiles : 1.0 - (double)full / tiles);
    ProfRec r;
    // implicit convolutions (kc < K) run the 8-phase kernels under their own instantiation (gemm8_kernel<true>): own record
    const int variant = gemm_variant(p, is16);
    const char* conv = (variant == 22 || variant == 27) && p.kc < p.K ? "_conv" : "";
    r.key = std::string(prof_cls_) + "/" + (part ? "gemm8s_bf16_128x128_tail" : gemm_variant_name(variant, is16)) + conv + (x3m ? "_x3" : "");
    if (static const bool by_class = std::getenv("SAMAUDIO_PROF_BY_CLASS") != nullptr; by_class) {   // diagnosis: one record per GEMM class
      int bit = 0;
      while (bit < SAMAUDIO_CLS_COUNT && !(cls & (1 << bit))) ++bit;
      r.key += "#" + std::to_string(bit);
    }
    r.flops = flops * share;
    r.bytes = bytes * share;
    SA_TRY(prof_event(&r.e0));
    SA_TRY(prof_event(&r.e1));
    SA_HIP(hipEventRecord(r.e0, st));
    if (full) SA_HIP(launch_gemm_part(p, is16, part, st));
    else SA_HIP(launch_gemm(p, is16, st));
    SA_HIP(hipEventRecord(r.e1, st));
    prof_.push_back(r);
  }
  return Status{};
}

Status Engine::gemm_x3(GemmParams p, const void* w3, bool ktm, hipStream_t st, int cls, const void* presplit) {
  if (!w3 || !d_.x3a) return fail(SAMAUDIO_ERR_STATE, "SAMAUDIO_OPT_X3_CLASSES: split weight or scratch operand missing (set the option before samaudio_prepare)");
  if (p.nbatch != 1 || p.kc != p.K || p.a_off || p.tap_stride || (p.out_act && p.out_f32))
    return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_X3_CLASSES: plain single-batch launches with one output only");
  const int K = p.K;
  // algorithmic bytes of the split: the fp32 row in, three 16-bit copies out
  if (!presplit)
    SA_TRY(op("split3", (double)p.M * K * (4 + 6), 0, st, [&] { return launch_split3((const float*)p.A, p.lda, d_.x3a, p.M, K, st); }));
  p.A = presplit ? presplit : d_.x3a; p.lda = 3L * K; p.K = 3 * K; p.kc = 3 * K; p.W = w3;
  if (p.out_act) {   // an fp32 context's "activation" outputs are fp32 tensors: the 16-bit kernel writes them as its fp32 output
    p.out_f32 = (float*)p.out_act; p.f32_ld = p.act_ld; p.f32_bstride = p.act_bstride; p.f32_off = p.act_off;
    p.f32_act = p.act != ACT_NONE;
    p.out_act = nullptr; p.act_ld = p.act_bstride = p.act_off = 0;
  }
  if (ktm) p.flags |= GEMM_FLAG_W_KTM;
  if ((p.flags & GEMM_FLAG_OUT_SPLIT3) && p.out_f32) {   // the result leaves as the next GEMM's split operand (16-bit, 3 x n_out per row)
    p.out_act = p.out_f32; p.act_ld = 3L * (p.swiglu ? p.N / 2 : p.N); p.act_bstride = p.act_off = 0;
    p.out_f32 = nullptr; p.f32_ld = p.f32_bstride = p.f32_off = 0; p.f32_act = 0;
  }
  return gemm(p, st, 2.0 * p.M * (double)p.N * K, cls, 2);   // flops as the reference counts them: one product over K
}

Status Engine::gemm_codec_x3(const GemmParams& p_in, const X3CodecW& w, hipStream_t st, double alg_flops) {
  const int c = w.cin;
  const long rows = (long)p_in.nbatch * (p_in.a_bstride / c);   // every row of the halo buffers the launch reads (halo rows are zeros)
  SA_TRY(op("split3", (double)rows * c * (4 + 6), 0, st, [&] { return launch_split3((const float*)p_in.A, c, x3_codec_scratch_, rows, c, st); }));
  GemmParams p = p_in;
  p.A = x3_codec_scratch_; p.W = w.w;
  p.a_off *= 3; p.a_bstride *= 3; p.lda *= 3; p.tap_stride *= 3; p.kc *= 3; p.K *= 3;
  // the 16-bit launch writes the RAW fp32 result (into the raw stream, or - a launch with an activated output only - into that
  // buffer); the activation follows as an elementwise pass with the fp32 kernel's own expressions
  const int act = p_in.act;
  float* const raw = p_in.out_f32 ? p_in.out_f32 : (float*)p_in.out_act;
  const long raw_off = p_in.out_f32 ? p_in.f32_off : p_in.act_off, raw_bs = p_in.out_f32 ? p_in.f32_bstride : p_in.act_bstride;
  if (!p_in.out_f32) { p.out_f32 = raw; p.f32_ld = p_in.act_ld; p.f32_bstride = p_in.act_bstride; p.f32_off = p_in.act_off; }
  p.out_act = nullptr; p.act_ld = p.act_bstride = p.act_off = 0; p.act = ACT_NONE; p.f32_act = 0;
  const double flops = alg_flops >= 0 ? alg_flops : 2.0 * p_in.M * (double)p_in.N * p_in.K * p_in.nbatch;
  SA_TRY(gemm(p, st, flops, SAMAUDIO_CLS_CODEC, p_in.K == c && p_in.kc == c ? 2 : 3));   // one block: [lo | hi | hi] x [W_hi | W_lo | W_hi] over the whole K'
  if (!p_in.out_act || (act == ACT_NONE && !p_in.out_f32)) return Status{};
  // region the launch wrote, per item: [c_lo, c_hi) of the windowed (transposed) convolutions, else M rows of N
  const long start = p_in.c_ld_rel ? p_in.c_lo : 0, count = p_in.c_ld_rel ? p_in.c_hi - p_in.c_lo : (long)p_in.M * p_in.N;
  const int chan = p_in.chan_mod ? p_in.chan_mod : p_in.N;
  if (start % chan) return fail(SAMAUDIO_ERR_ARG, "gemm_codec_x3: window start is not a whole channel row");
  return op("codec_act", (double)count * p_in.nbatch * 8, 0, st, [&] {
    return launch_act_flat(raw + raw_off + start, raw_bs, (float*)p_in.out_act + p_in.act_off + start, p_in.act_bstride, p_in.nbatch, count,
                           chan, act, p_in.act_alpha, st);
  });
}

// One DAC residual unit: k7 convolution `p` (Snake'd bf16 intermediate) followed by the k1 convolution `q` on it
// (+ fp32 residual, Snake'd bf16 copy out).  `cur` = the unit's input activation, `alt` = a second halo-zeroed buffer of
// the same shape.  Large bf16 launches run as ONE kernel (gemm2.hip resunit_kernel: the intermediate stays in LDS) that
// writes its activation to `alt` - it must not overwrite rows neighbouring tiles still read - and the buffers swap roles;
// everything else runs as the two launches with `alt` as the intermediate.  Both forms are bitwise identical.
Status Engine::res_unit(GemmParams p, GemmParams q, void*& cur, void*& alt, double flops7, double flops1, hipStream_t st) {
  p.out_act = alt;
  q.A = alt;
  q.out_act = cur;
  GemmParams fp = p, fq = q;
  fq.out_act = alt;
  fp.tag = fq.tag = prof_cls_[0] == 'c' ? 1 : 0;
  const bool covered = p.N == 64 || p.N == 96 || p.N == 128 || p.N == 192;
  // fused for all four channel counts: since the residual-unit kernels issue their direct-to-LDS loads as inline assembly
  // (gemm2.hip dma16a) the fused form is the faster one everywhere (profiles/r3_call12/op_bench.log, 8 waveforms, fused vs
  // two launches: C = 64 923 vs 1193 us, C = 96 1697 vs 2269, C = 128 1353 vs 1391, C = 192 2515 vs 2784)
  const bool fuse = bf16_ && covered && !debug_flag(16) && resunit_ok(fp, fq) &&
                    ((long)((p.M + 255) / 256) * p.nbatch >= 256 || debug_flag(18));
  if (!fuse) {
    SA_TRY(gemm(p, st, flops7));
    return gemm(q, st, flops1);
  }
  if (const char* why = gemm_check(fp, bf16_)) return fail(SAMAUDIO_ERR_ARG, why);
  if (const char* why = gemm_check(fq, bf16_)) return fail(SAMAUDIO_ERR_ARG, why);
  const double inter = (double)p.M * p.N * p.nbatch * esz_;   // the intermediate: neither written nor read
  SA_TRY(op("resunit_bf16", gemm_alg_bytes(fp, esz_) + gemm_alg_bytes(fq, esz_) - 2 * inter, flops7 + flops1, st,
            [&] { return launch_resunit(fp, fq, st); }));
  if (sentinel_on_)   // the fused unit's 16-bit output (halo layout: the rows the launch writes)
    for (int b = 0; b < fq.nbatch; ++b)
      SA_TRY(sentinel(cls_slot(SAMAUDIO_CLS_CODEC), (const char*)fq.out_act + ((size_t)fq.act_off + (size_t)b * fq.act_bstride) * esz_,
                      bf16_ ? 1 : 0, fq.M, fq.N, fq.act_ld, st));
  std::swap(cur, alt);
  return Status{};
}

template <class F>
Status Engine::op(const char* name, double alg_bytes, double alg_flops, hipStream_t st, F&& launch) {
  if (!prof_on_) {
    SA_HIP(launch());
    return Status{};
  }
  ProfRec r;
  r.key = std::string(prof_cls_) + "/" + name;
  r.flops = alg_flops;
  r.bytes = alg_bytes;
  SA_TRY(prof_event(&r.e0));
  SA_TRY(prof_event(&r.e1));
  SA_HIP(hipEventRecord(r.e0, st));
  SA_HIP(launch());
  SA_HIP(hipEventRecord(r.e1, st));
  prof_.push_back(r);
  return Status{};
}

Status Engine::prof_event(hipEvent_t* e) {
  if (ev_used_ == ev_pool_.size()) {
    hipEvent_t ev;
    SA_HIP(hipEventCreate(&ev));
    ev_pool_.push_back(ev);
  }
  *e = ev_pool_[ev_used_++];
  return Status{};
}

Status Engine::profile_begin() {
  prof_.clear();
  ev_used_ = 0;
  prof_on_ = true;
  return Status{};
}

Status Engine::profile_end(std::vector<KernelStat>& out) {
  prof_on_ = false;
  out.clear();
  std::map<std::string, size_t> index;
  for (const ProfRec& r : prof_) {
    SA_HIP(hipEventSynchronize(r.e1));
    float ms = 0.f;
    SA_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    auto it = index.find(r.key);
    if (it == index.end()) {
      it = index.emplace(r.key, out.size()).first;
      out.push_back(KernelStat{});
      out.back().name = r.key;
    }
    KernelStat& k = out[it->second];
    k.launches += 1;
    k.flops += r.flops;
    k.bytes += r.bytes;
    k.ms += ms;
  }
  prof_.clear();
  ev_used_ = 0;
  return Status{};
}

Engine::~Engine() {
  for (hipEvent_t e : ev_pool_) (void)hipEventDestroy(e);
  debug_device_free(sentinel_dev_);
  if (hash_) {   // SAMAUDIO_TRACE_HASH recorder
    HashTrace* h = (HashTrace*)hash_;
    debug_device_free(h->dev);
    delete h;
  }
}

static GemmParams lin(const void* A, long lda, const void* W, long M, int N, int K) {
  GemmParams p;
  std::memset(&p, 0, sizeof(p));
  p.A = A; p.W = W; p.lda = lda; p.kc = K; p.tap_stride = 0;
  p.M = (int)M; p.N = N; p.K = K; p.nbatch = 1; p.alpha = 1.f; p.rows_per_gate = 1;
  return p;
}
static void out_f32(GemmParams& p, float* o, long ld) { p.out_f32 = o; p.f32_ld = ld; }
static void out_act(GemmParams& p, void* o, long ld, int act = ACT_NONE) { p.out_act = o; p.act_ld = ld; p.act = act; }
static void with_res(GemmParams& p, const float* r, long ld) { p.res = r; p.res_ld = ld; }

// ---------------------------------------------------------------------------------------------------
// conditioning that is constant over the ODE (hoisted out of the 32 evaluations)
// ---------------------------------------------------------------------------------------------------
Status Engine::prepare(int rows, int T, int Lt, const float* feats, const float* text, const uint8_t* text_mask,
                       const float* video, const int64_t* anchor_ids, int n_ids, const int64_t* anchor_alignment,
                       const uint8_t* pad_mask, hipStream_t st, int cand, bool latent_feats) {
  if (!dit_ready_) return fail(SAMAUDIO_ERR_STATE, "prepare: DiT weights not finalized");
  if (rows <= 0 || T <= 0 || !feats) return fail(SAMAUDIO_ERR_ARG, "prepare: bad shape");
  if (cand < 1 || rows % cand) return fail(SAMAUDIO_ERR_ARG, "prepare: rows must be a multiple of candidates");
  if (T > cfg_.max_positions) return fail(SAMAUDIO_ERR_ARG, "prepare: more frames than RoPE positions");
  if (text && Lt <= 0) return fail(SAMAUDIO_ERR_ARG, "prepare: text_len must be positive");
  if (!text) Lt = 1;
  if (anchor_ids && (!anchor_alignment || n_ids <= 0)) return fail(SAMAUDIO_ERR_ARG, "prepare: anchors incomplete");
  prof_cls_ = "prep";
  Bump b(ws_, ws_bytes_);
  plan_dit(b, rows, T, Lt, true);
  if (!ws_ || !b.fits())
    return fail(SAMAUDIO_ERR_WORKSPACE, "prepare: workspace too small (" + std::to_string(b.used()) + " bytes needed)");
  rows_ = rows; frames_ = T; text_len_ = Lt; frames_pad_ = (int)round_up(T, 64);
  const int D = cfg_.dim, C2 = cfg_.latent_channels;
  const long M = (long)rows * T, Mt = (long)rows * Lt;
  // the conditioning is computed once per CLIP (B = rows / candidates of them) and then repeated for the clip's candidates: the
  // per-clip results live in buffers the evaluations overwrite anyway (aligned, hp1) until the repeat kernels have read them
  const int B = rows / cand;
  const long Mb = (long)B * T, Mtb = (long)B * Lt;
  float* const cond_b = cand > 1 ? d_.aligned : d_.cond;
  float* const textp_b = cand > 1 ? d_.hp1 : d_.text_proj;

  if (pad_mask && cand > 1) SA_HIP(launch_repeat_rows_u8(pad_mask, d_.pad_mask, B, cand, T, st));
  else if (pad_mask) SA_HIP(hipMemcpyAsync(d_.pad_mask, pad_mask, M, hipMemcpyDeviceToDevice, st));
  else SA_HIP(hipMemsetAsync(d_.pad_mask, 1, M, st));
  if (text && text_mask && cand > 1) SA_HIP(launch_repeat_rows_u8(text_mask, d_.text_mask, B, cand, Lt, st));
  else if (text && text_mask) SA_HIP(hipMemcpyAsync(d_.text_mask, text_mask, Mt, hipMemcpyDeviceToDevice, st));
  else SA_HIP(hipMemsetAsync(d_.text_mask, 1, Mt, st));
  // patcher conv input: halo rows stay zero for the whole solve
  SA_HIP(hipMemsetAsync(d_.gnbuf, 0, (size_t)rows * (T + 2) * D * esz_, st));

  // SAMAUDIO_CLS_PREP in exact fp32 (16-bit contexts): the caller's fp32 tensors are the operands themselves
  const bool pf = f32c(SAMAUDIO_CLS_PREP);
  const int PREP = SAMAUDIO_CLS_PREP;
  // cond = proj_b + audio_features @ Wf^T                           (model.py:116-125, columns 512..767)
  // latent_feats: audio_features = (z | z) of the codec latent z [Mb, C2 / 2] (model.py:182-184): the K axis of this GEMM is two
  // taps of C2 / 2 channels that read the SAME row (tap stride 0) - the same products in the same order as on the concatenation
  const int fw = latent_feats ? C2 / 2 : C2;   // width of the rows `feats` holds
  if (!pf) SA_HIP(launch_to_act(feats, 0, fw, 0, d_.feats, 0, bf16_, 1, Mb, fw, fw, 0, st));
  {
    GemmParams p = pf ? lin(feats, fw, g32_.proj_wf, Mb, D, C2) : lin(d_.feats, fw, g_.proj_wf, Mb, D, C2);
    if (latent_feats) { p.kc = fw; p.tap_stride = 0; }
    p.bias = g_.proj_b;
    out_f32(p, cond_b, D);
    SA_TRY(gemm(p, st, -1.0, PREP, pf));
  }
  // cond += tanh(g_v) * LayerNorm(conv1x1(video))                   (align.py:41-50; zeros if no video: Q8)
  const void* vid_op = d_.video;
  if (pf && video) vid_op = video;
  else if (pf) { vid_op = d_.prep32; SA_HIP(hipMemsetAsync(d_.prep32, 0, (size_t)Mb * cfg_.video_dim * 4, st)); }
  else if (video) SA_HIP(launch_to_act(video, 0, cfg_.video_dim, 0, d_.video, 0, bf16_, 1, Mb, cfg_.video_dim, cfg_.video_dim, 0, st));
  else SA_HIP(hipMemsetAsync(d_.video, 0, (size_t)Mb * cfg_.video_dim * esz_, st));
  {
    GemmParams p = lin(vid_op, cfg_.video_dim, pf ? (const void*)g32_.vid_w : g_.vid_w, Mb, D, cfg_.video_dim);
    p.bias = g_.vid_b;
    out_f32(p, d_.vtmp, D);
    SA_TRY(gemm(p, st, -1.0, PREP, pf));
    SA_HIP(launch_layernorm_accum(d_.vtmp, g_.vid_ln_w, g_.vid_ln_b, g_.vid_gate, cond_b, (int)Mb, D, 1e-5f, st));
  }
  // cond += tanh(g_a) * proj(Emb[ids.gather(alignment)])            (model.py:54-65; tanh folded into anc_w)
  // folded cross-attention output projection: zero the probability buffer once (its K padding columns stay zero)
  fold_ltp_ = fold_kp_ = 0;
  if (bf16_ && Lt <= 16 && D / cfg_.n_heads == 128 && !std::getenv("SAMAUDIO_NO_FOLD")) {
    fold_ltp_ = Lt <= 8 ? 8 : 16;
    fold_kp_ = (int)round_up((long)cfg_.n_heads * fold_ltp_, 64);
    SA_HIP(hipMemsetAsync(d_.probs, 0, (size_t)M * fold_kp_ * esz_, st));
    // 0 * (K padding of U) must stay 0
    SA_HIP(hipMemsetAsync(d_.ut, 0, (size_t)rows * D * fold_kp_ * esz_ * (cfg_.n_layers <= kMaxFoldLayers ? cfg_.n_layers : 1), st));
  }
  fold3_ = false;
  if (d_.ut3 && d_.x3p) {   // x3 context: the fold on compensated operands (zero K padding of P and U, once per prepare)
    fold3_ = true;
    fold_ltp_ = Lt <= 8 ? 8 : 16;
    fold_kp_ = (int)round_up((long)cfg_.n_heads * fold_ltp_, 64);
    SA_HIP(hipMemsetAsync(d_.x3p, 0, (size_t)M * 3 * fold_kp_ * 2, st));
    SA_HIP(hipMemsetAsync(d_.ut3, 0, (size_t)rows * D * 3 * fold_kp_ * 2 * cfg_.n_layers, st));
  }
  has_anchor_ = anchor_ids != nullptr;
  if (anchor_ids) {
    SA_HIP(launch_anchor_gather(g_.anc_emb, (const long*)anchor_ids, n_ids, (const long*)anchor_alignment,
                                pf ? (void*)d_.prep32 : d_.anch, pf ? false : bf16_, B, T, cfg_.anchor_dim,
                                cfg_.anchor_vocab, st));
    GemmParams p = pf ? lin(d_.prep32, cfg_.anchor_dim, g32_.anc_w, Mb, D, cfg_.anchor_dim)
                      : lin(d_.anch, cfg_.anchor_dim, g_.anc_w, Mb, D, cfg_.anchor_dim);
    with_res(p, cond_b, D);
    out_f32(p, cond_b, D);
    SA_TRY(gemm(p, st, -1.0, PREP, pf));
  }
  // text_proj = memory_proj(text)                                   (model.py:171)
  if (text) {
    if (!pf) SA_HIP(launch_to_act(text, 0, cfg_.text_dim, 0, d_.text, 0, bf16_, 1, Mtb, cfg_.text_dim, cfg_.text_dim, 0, st));
    GemmParams p = pf ? lin(text, cfg_.text_dim, g32_.mem_w, Mtb, D, cfg_.text_dim)
                      : lin(d_.text, cfg_.text_dim, g_.mem_w, Mtb, D, cfg_.text_dim);
    p.bias = g_.mem_b;
    out_f32(p, textp_b, D);
    SA_TRY(gemm(p, st, -1.0, PREP, pf));
  } else {
    SA_HIP(hipMemsetAsync(textp_b, 0, (size_t)Mtb * D * 4, st));
  }
  if (cand > 1) {   // sample-major repeat of the per-clip conditioning (model.py:193-203)
    SA_HIP(launch_repeat_items_f32(cond_b, d_.cond, B, cand, (long)T * D, st));
    SA_HIP(launch_repeat_items_f32(textp_b, d_.text_proj, B, cand, (long)Lt * D, st));
  }
  prepared_ = true;
  return Status{};
}

// ---------------------------------------------------------------------------------------------------
// one evaluation of the vector field: out = res + alpha * DiT(align(noisy), t)
// ---------------------------------------------------------------------------------------------------
Status Engine::eval_field(const float* noisy, const float* time, int nt, float* out, const float* res, float alpha,
                          hipStream_t st) {
  if (!prepared_) return fail(SAMAUDIO_ERR_STATE, "forward: call samaudio_prepare first");
  if (nt != 1 && nt != rows_) return fail(SAMAUDIO_ERR_ARG, "forward: n_time must be 1 or rows");
  const int D = cfg_.dim, F = cfg_.ffn_hidden, C2 = cfg_.latent_channels, H = cfg_.n_heads, T = frames_,
            Lt = text_len_, Tp = frames_pad_, rows = rows_;
  const long M = (long)rows * T, Mt = (long)rows * Lt;
  const float eps = cfg_.norm_eps;
  const int hd = D / H;   // 128 | 64 (finalize)
  const long t6 = nt == 1 ? 0 : 6L * D, t1 = nt == 1 ? 0 : (long)D;
  prof_cls_ = "dit";
  const double MD = (double)M * D;
  if (hash_on() && !hash_) hash_ = new HashTrace();
  const HashScope hash_scope(hash_on() ? (HashTrace*)hash_ : nullptr, rows);
  hash_stage("noisy", noisy, (size_t)M * C2 * 4, st);

  // aligned = noisy @ Wy^T + cond                                   (model.py:116-125, columns 0..255)
  {
    const bool f = f32c(SAMAUDIO_CLS_IN);   // exact fp32: the ODE state itself is the operand
    if (!f) SA_HIP(launch_to_act(noisy, 0, C2, 0, d_.ybf, 0, bf16_, 1, M, C2, C2, 0, st));
    GemmParams p = f ? lin(noisy, C2, g32_.proj_wy, M, D, C2) : lin(d_.ybf, C2, g_.proj_wy, M, D, C2);
    with_res(p, d_.cond, D);
    out_f32(p, d_.aligned, D);
    SA_TRY(gemm(p, st, -1.0, SAMAUDIO_CLS_IN, f));
  }
  // patcher: (GroupNorm(1) -> SiLU -> conv k3) x 2 + skip           (patcher.py:138-141)
  auto patch_conv = [&](const void* W, const void* W3, bool ktm3, const float* bias, const float* skip, float* dst) -> Status {
    GemmParams p = lin(d_.gnbuf, D, W, T, D, 3 * D);
    p.kc = D; p.tap_stride = D; p.a_off = 0; p.a_bstride = (long)(T + 2) * D; p.nbatch = rows;
    p.bias = bias;
    if (skip) { with_res(p, skip, D); p.res_bstride = (long)T * D; }
    out_f32(p, dst, D);
    p.f32_bstride = (long)T * D;
    if (!x3(SAMAUDIO_CLS_PATCH)) return gemm(p, st, -1.0, SAMAUDIO_CLS_PATCH);
    // compensated operands: every row of the halo-padded GroupNorm output (halo rows are zeros: they split into zeros) becomes
    // [lo | hi | hi], a tap of the convolution then is 3 D contiguous elements against that tap's [W_hi | W_lo | W_hi]
    if (!W3 || !d_.x3a) return fail(SAMAUDIO_ERR_STATE, "SAMAUDIO_OPT_X3_CLASSES: patcher split weight or scratch operand missing");
    const long prow = (long)rows * (T + 2);
    SA_TRY(op("split3", (double)prow * D * (4 + 6), 0, st, [&] { return launch_split3((const float*)d_.gnbuf, D, d_.x3a, prow, D, st); }));
    p.A = d_.x3a; p.W = W3; p.lda = 3L * D; p.kc = 3 * D; p.tap_stride = 3L * D; p.a_bstride = (long)(T + 2) * 3 * D; p.K = 9 * D;
    if (ktm3) p.flags |= GEMM_FLAG_W_KTM;
    return gemm(p, st, 2.0 * T * (double)D * 3 * D * rows, SAMAUDIO_CLS_PATCH, 3);
  };
  trace("cond", d_.cond, (size_t)M * D, false, st);
  trace("aligned", d_.aligned, (size_t)M * D, false, st);
  SA_TRY(op("groupnorm_silu", MD * (4 + esz_), 0, st, [&] {
    return launch_groupnorm_silu(d_.aligned, g_.gn1_w, g_.gn1_b, d_.gn_part, d_.gnbuf, bf16_, rows, T, D, 1, 1e-5f, st);
  }));
  SA_TRY(patch_conv(g_.pw1, g3_.pw1, g3_.ktm & 1, g_.pb1, nullptr, d_.hp1));
  SA_TRY(op("groupnorm_silu", MD * (4 + esz_), 0, st, [&] {
    return launch_groupnorm_silu(d_.hp1, g_.gn2_w, g_.gn2_b, d_.gn_part, d_.gnbuf, bf16_, rows, T, D, 1, 1e-5f, st);
  }));
  SA_TRY(patch_conv(g_.pw2, g3_.pw2, g3_.ktm & 2, g_.pb2, d_.aligned, d_.h));

  // timestep embeddings                                             (transformer.py:490-493, model.py:170)
  {
    // one row per time value: in exact fp32 these three GEMMs cost nothing, and their rounding would reach the shift /
    // scale / gate of every row of every layer coherently
    const bool f = f32c(SAMAUDIO_CLS_TIME);
    const int TIME = SAMAUDIO_CLS_TIME;
    void *temb = f ? (void*)d_.temb32 : d_.temb, *tu = f ? (void*)d_.tu32 : d_.tu, *tsilu = f ? (void*)d_.tsilu32 : d_.tsilu;
    SA_HIP(launch_time_features(time, nt, g_.t_freqs, cfg_.freq_dim, g_.mem_inv_freq, D, temb, d_.tsin, f ? false : bf16_, st));
    GemmParams p = lin(temb, cfg_.freq_dim, f ? (const void*)g32_.t_w13 : g_.t_w13, nt, 2 * D, cfg_.freq_dim);
    p.swiglu = 1;
    out_act(p, tu, D);
    SA_TRY(gemm(p, st, -1.0, TIME, f));
    p = lin(tu, D, f ? (const void*)g32_.t_w2 : g_.t_w2, nt, D, D);
    out_f32(p, d_.t_emb, D);
    out_act(p, tsilu, D, ACT_SILU);
    SA_TRY(gemm(p, st, -1.0, TIME, f));
    p = lin(tsilu, D, f ? (const void*)g32_.tb_w : g_.tb_w, nt, 6 * D, D);
    p.bias = g_.tb_b;
    out_f32(p, d_.t0, 6L * D);
    SA_TRY(gemm(p, st, -1.0, TIME, f));
  }
  // RMSNorm + modulate operands of this evaluation, pre-combined for every layer's two norms (kernels.hip mod_tables)
  const bool mod_gs = 2 * cfg_.n_layers <= kMaxModNorms && cfg_.n_layers > 0 && D <= 256 * 12;
  if (alt_classes_ && bf16_ && !mod_gs)
    return fail(SAMAUDIO_ERR_ARG, "SAMAUDIO_OPT_ALT16_CLASSES: the mixed mode needs the pre-combined RMSNorm operands (<= 48 layers, D <= 3072)");
  const long gs_ld = nt == 1 ? 0 : 2L * D;
  if (mod_gs) {
    ModTables mt;
    for (int l = 0; l < cfg_.n_layers; ++l)
      for (int k = 0; k < 2; ++k) {
        const int n = 2 * l + k;
        mt.w[n] = k ? layers_[l].ffn_norm : layers_[l].attn_norm;
        mt.shift_tab[n] = layers_[l].mod_table + (k ? 3 : 0) * D;
        mt.scale_tab[n] = layers_[l].mod_table + (k ? 4 : 1) * D;
        mt.shift_off[n] = (k ? 3 : 0) * D;
        mt.scale_off[n] = (k ? 4 : 1) * D;
      }
    for (int n = 2 * cfg_.n_layers; n < kMaxModNorms; ++n) {
      mt.w[n] = mt.shift_tab[n] = mt.scale_tab[n] = nullptr;
      mt.shift_off[n] = mt.scale_off[n] = 0;
    }
    SA_TRY(op("mod_tables", 2.0 * cfg_.n_layers * (5.0 + 2.0 * nt) * D * 4, 0, st, [&] {
      return launch_mod_tables(mt, 2 * cfg_.n_layers, d_.t0, t6, nt, d_.modgs, D, st);
    }));
  }
  // memory = memory_proj(text) + sincos(t); y = y_embedder(memory)  (model.py:170-172, transformer.py:495)
  {
    const bool f = f32c(SAMAUDIO_CLS_YEMB);
    const int YEMB = SAMAUDIO_CLS_YEMB;
    void *mem = f ? (void*)d_.mem32 : d_.mem, *yu = f ? (void*)d_.yu32 : d_.yu;
    SA_HIP(launch_add_rowvec(d_.text_proj, d_.tsin, t1, mem, f ? false : bf16_, (int)Mt, D, Lt, st));
    GemmParams p = lin(mem, D, f ? (const void*)g32_.y_w13 : g_.y_w13, Mt, 2 * D, D);
    p.swiglu = 1;
    out_act(p, yu, D);
    SA_TRY(gemm(p, st, -1.0, YEMB, f));
    p = lin(yu, D, f ? (const void*)g32_.y_w2 : g_.y_w2, Mt, D, D);
    if (f) {  // the K | V projections read the 16-bit copy
      out_f32(p, d_.yemb32, D);
      SA_TRY(gemm(p, st, -1.0, YEMB, true));
      SA_HIP(launch_to_act(d_.yemb32, 0, D, 0, d_.yemb, 0, true, 1, Mt, D, D, 0, st));
    } else {
      out_act(p, d_.yemb, D);
      SA_TRY(gemm(p, st, -1.0, YEMB));
    }
  }

  trace("patcher out h", d_.h, (size_t)M * D, false, st);
  trace("t0", d_.t0, (size_t)nt * 6 * D, false, st);
  trace("yemb", d_.yemb, (size_t)Mt * D, bf16_, st);
  const long kv_ld = 2L * D * cfg_.n_layers;
  if (cfg_.n_layers > 0) {  // cross-attention keys / values of every layer (k-normed), [Mt, L*2D]
    GemmParams p = lin(d_.yemb, D, g_.c_wkv_all, Mt, (int)kv_ld, D);
    out_act(p, d_.kvc, kv_ld);
    if (x3(SAMAUDIO_CLS_CKV)) SA_TRY(gemm_x3(p, g3_.c_wkv_all, g3_.ktm & 4, st, SAMAUDIO_CLS_CKV));
    else SA_TRY(gemm(p, st, -1.0, SAMAUDIO_CLS_CKV));
    SA_HIP(launch_headnorm_layers(d_.kvc, g_.c_k_norm_all, bf16_, (int)Mt, cfg_.n_layers, H, eps, st, hd));
  }
  trace("kvc", d_.kvc, (size_t)Mt * kv_ld, bf16_, st);
  // folded cross-attention: U_l = Wo_l V_l of EVERY layer in one launch (it depends on the text memory only, not on h)
  const bool fold_all = fold_ltp_ && !fold3_ && cfg_.n_layers <= kMaxFoldLayers && !debug_flag(31);   // flag 31: one launch per layer (A/B, tests)
  if (fold3_) {   // x3 context: U = Wo V of every layer on split operands, [L][rows][D][3 kp] = [U_hi | U_lo | U_hi]
    const float* wos[kMaxFoldLayers];
    for (int l = 0; l < cfg_.n_layers; ++l) wos[l] = (const float*)layers_[l].c_wo;
    SA_TRY(op("cross_attn_fold3", ((double)D * D * 4 + (double)rows * D * fold_kp_ * 6 + (double)Mt * D * 4) * cfg_.n_layers, 0, st, [&] {
      return launch_cross_attn_fold3_layers(wos, cfg_.n_layers, (const float*)d_.kvc, kv_ld, d_.ut3, fold_kp_, rows, Lt, fold_ltp_, H, st);
    }));
  }
  const size_t ut_layer = (size_t)rows * D * fold_kp_ * esz_;
  if (fold_all) {
    const void* wos[kMaxFoldLayers];
    for (int l = 0; l < cfg_.n_layers; ++l) wos[l] = layers_[l].c_wo;
    SA_TRY(op("cross_attn_fold", ((double)D * D + (double)rows * D * fold_kp_ + (double)Mt * D) * esz_ * cfg_.n_layers, 0, st, [&] {
      return launch_cross_attn_fold_layers(wos, cfg_.n_layers, d_.kvc, kv_ld, d_.ut, fold_kp_, rows, Lt, fold_ltp_, H, st);
    }));
  }
  // SAMAUDIO_OPT_PREFETCH_ROWS: a launch's idle workgroups read the next big GEMM's weights (gemm8.hip prefetch_lines).  Chain per
  // layer: qkv -> wo -> c_wq -> c_wo (read by the fold kernel); w13 -> w2 -> the next layer's qkv.  Nobody prefetches w13: the only
  // launch in front of it with idle CUs is the folded cross-attention GEMM (K = 192: 15 us), which the 85 MB read stretched to 27 us,
  // and w13 itself (236 tiles of 256 x 256) measured the same warm or cold (round 5, profiles/r5_call2/).
  const bool pf_on = bf16_ && prefetch_rows_ > 0 && M <= prefetch_rows_;
  auto prefetch = [&](GemmParams& p, const void* w_next, double elems) {
    if (pf_on && w_next) { p.pf_ptr = w_next; p.pf_bytes = (long)(elems * esz_); }
  };
  auto ktm = [](GemmParams& p, const LayerW& w, int bit) { if (w.ktm & (1 << bit)) p.flags |= GEMM_FLAG_W_KTM; };
  for (int l = 0; l < cfg_.n_layers; ++l) {  // DiTBlock.forward, transformer.py:354-391
    const LayerW& w = layers_[l];
    const float* tab = w.mod_table;
    // self-attention branch
    // compensated operands (fp32 contexts): the producers write the split form [lo | hi | hi] themselves where they can
    const bool qkv_pre = x3(SAMAUDIO_CLS_QKV) && mod_gs, w13_pre = x3(SAMAUDIO_CLS_W13) && mod_gs;
    const bool wo_pre = x3(SAMAUDIO_CLS_WO) && x3(SAMAUDIO_X3_ATTENTION);
    const bool w2_pre = x3(SAMAUDIO_CLS_W2) && x3(SAMAUDIO_CLS_W13) && F % 16 == 0;
    SA_TRY(op("rmsnorm_mod", MD * (4 + (qkv_pre ? 6 : esz_)), 0, st, [&] {
      if (qkv_pre)
        return launch_rmsnorm_gs_split3(d_.h, d_.modgs + (2L * l) * nt * 2 * D, gs_ld, d_.x3a, (int)M, D, T, eps, st);
      if (mod_gs)
        return launch_rmsnorm_gs(d_.h, d_.modgs + (2L * l) * nt * 2 * D, gs_ld, d_.xn, bf16_, (int)M, D, T, eps, st,
                                 alt16(SAMAUDIO_CLS_QKV));
      return launch_rmsnorm_mod(d_.h, w.attn_norm, tab + 0 * D, tab + 1 * D, d_.t0, t6, 0 * D, 1 * D, d_.xn, bf16_, (int)M, D,
                                T, eps, st);
    }));
    SA_TRY(sentinel(14, d_.xn, !bf16_ ? 0 : (alt16(SAMAUDIO_CLS_QKV) ? 2 : 1), M, D, D, st));
    {
      GemmParams p = lin(d_.xn, D, w.wqkv, M, 3 * D, D);
      ktm(p, w, 0);
      prefetch(p, w.wo, (double)D * D);
      out_act(p, d_.qkv, 3L * D);
      if (x3(SAMAUDIO_CLS_QKV)) SA_TRY(gemm_x3(p, w.wqkv3, w.ktm3 & 1, st, SAMAUDIO_CLS_QKV, qkv_pre ? d_.x3a : nullptr));
      else SA_TRY(gemm(p, st, -1.0, SAMAUDIO_CLS_QKV));
    }
    SA_TRY(op("qkv_prep", 2 * 3 * MD * esz_, 0, st, [&] {
      if (x3(SAMAUDIO_X3_ATTENTION) && hd == 128)   // fp32 tensors, the fast access pattern (its consumer is the compensated attention)
        return launch_qkv_prep_f32x((const float*)d_.qkv, w.q_norm, w.k_norm, g_.rope_cos, g_.rope_sin, (float*)d_.Q, (float*)d_.K,
                                    (float*)d_.Vt, rows, T, Tp, H, eps, st);
      return launch_qkv_prep(d_.qkv, w.q_norm, w.k_norm, g_.rope_cos, g_.rope_sin, d_.Q, d_.K, d_.Vt, bf16_, rows, T, Tp, H,
                             eps, st, hd);
    }));
    trace("  xn", d_.xn, (size_t)M * D, bf16_, st);
    trace("  qkv", d_.qkv, (size_t)M * 3 * D, bf16_, st);
    trace("  Q", d_.Q, (size_t)rows * Tp * D, bf16_, st);
    trace("  K", d_.K, (size_t)rows * Tp * D, bf16_, st);
    trace("  Vt", d_.Vt, (size_t)rows * Tp * D, bf16_, st);
    if (x3(SAMAUDIO_X3_ATTENTION))
      SA_TRY(op("self_attention_x3", 4 * MD * 4, 4.0 * T * T * D * rows, st, [&] {
        return launch_self_attention_x3((const float*)d_.Q, (const float*)d_.K, (const float*)d_.Vt, d_.pad_mask, (float*)d_.attn, rows,
                                        T, Tp, H, hd, st, wo_pre ? d_.x3a : nullptr);
      }));
    else
    SA_TRY(op("self_attention", 4 * MD * esz_, 4.0 * T * T * D * rows, st, [&] {
      return launch_self_attention_hd(d_.Q, d_.K, d_.Vt, d_.pad_mask, d_.attn, bf16_, rows, T, Tp, H, hd, st, alt16(SAMAUDIO_CLS_WO));
    }));
    trace("  attn", d_.attn, (size_t)M * D, bf16_, st);
    SA_TRY(sentinel(15, d_.attn, !bf16_ ? 0 : (alt16(SAMAUDIO_CLS_WO) ? 2 : 1), M, D, D, st));
    {
      GemmParams p = lin(d_.attn, D, w.wo, M, D, D);  // h = x + gate_msa * attn
      p.gate_tab = tab + 2 * D; p.gate = d_.t0 + 2 * D; p.gate_ld = t6; p.rows_per_gate = T;
      with_res(p, d_.h, D);
      out_f32(p, d_.h, D);
      out_act(p, d_.hbf, D);
      if (alt16(SAMAUDIO_CLS_CWQ)) p.flags |= 512;   // hbf is c_wq's operand
      ktm(p, w, 1);
      prefetch(p, w.c_wq, (double)D * D);
      if (x3(SAMAUDIO_CLS_WO)) {   // (fp32 outputs only: c_wq then reads h itself)
        p.out_act = nullptr; p.act_ld = 0;
        SA_TRY(gemm_x3(p, w.wo3, w.ktm3 & 2, st, SAMAUDIO_CLS_WO, wo_pre ? d_.x3a : nullptr));
      } else SA_TRY(gemm(p, st, -1.0, SAMAUDIO_CLS_WO));
    }
    trace("  h after wo", d_.h, (size_t)M * D, false, st);
    trace("  hbf", d_.hbf, (size_t)M * D, bf16_, st);
    // cross-attention branch: h = h + CA(h, y)   (no norm, no gate: quirk Q4)
    {
      // (an fp32 context whose wo ran on compensated operands has no second copy of h)
      GemmParams p = lin(x3(SAMAUDIO_CLS_WO) ? (const void*)d_.h : d_.hbf, D, w.c_wq, M, D, D);
      ktm(p, w, 2);
      if (!fold_all) prefetch(p, w.c_wo, (double)D * D);   // (read by the per-layer fold kernel)
      out_act(p, d_.qc, D);
      if (x3(SAMAUDIO_CLS_CWQ)) SA_TRY(gemm_x3(p, w.c_wq3, w.ktm3 & 4, st, SAMAUDIO_CLS_CWQ));
      else SA_TRY(gemm(p, st, -1.0, SAMAUDIO_CLS_CWQ));
    }
    const void* kv_l = (const char*)d_.kvc + (size_t)l * 2 * D * esz_;
    if (fold3_) {
      // h += P . U on compensated operands: K' = 3 kp = 576 instead of 3 D
      SA_TRY(op("cross_attn_probs3", (MD * 4 + (double)M * fold_kp_ * 6 + (double)Mt * 2 * D * 4), 0, st, [&] {
        return launch_cross_attn_probs3((const float*)d_.qc, w.c_q_norm, (const float*)kv_l, kv_ld, d_.text_mask, d_.x3p, fold_kp_, rows, T, Lt,
                                        fold_ltp_, H, eps, st);
      }));
      const int K3 = 3 * fold_kp_;
      GemmParams p = lin(d_.x3p, K3, (const char*)d_.ut3 + (size_t)l * rows * D * K3 * 2, T, D, K3);
      p.nbatch = rows;
      p.a_bstride = (long)T * K3;
      p.w_bstride = (long)D * K3;
      with_res(p, d_.h, D);
      p.res_bstride = (long)T * D;
      out_f32(p, d_.h, D);
      p.f32_bstride = (long)T * D;
      SA_TRY(gemm(p, st, 2.0 * M * (double)D * H * Lt, SAMAUDIO_CLS_CWO, 2));
    } else if (fold_ltp_) {
      // h += P . U with U = Wo V folded per (batch, head, token): K = H*Lt instead of D (see attention.hip)
      SA_TRY(op("cross_attn_probs", (MD + (double)M * fold_kp_ + (double)Mt * 2 * D) * esz_, 0, st, [&] {
        return launch_cross_attn_probs(d_.qc, w.c_q_norm, kv_l, kv_ld, d_.text_mask, d_.probs, fold_kp_, rows, T, Lt,
                                       fold_ltp_, H, eps, st);
      }));
      const void* ut_l = fold_all ? (const void*)((const char*)d_.ut + (size_t)l * ut_layer) : d_.ut;
      if (!fold_all)
        SA_TRY(op("cross_attn_fold", ((double)D * D + (double)rows * D * fold_kp_ + (double)Mt * D) * esz_, 0, st, [&] {
          return launch_cross_attn_fold(w.c_wo, kv_l, kv_ld, d_.ut, fold_kp_, rows, Lt, fold_ltp_, H, st);
        }));
      GemmParams p = lin(d_.probs, fold_kp_, ut_l, T, D, fold_kp_);
      p.nbatch = rows;
      p.a_bstride = (long)T * fold_kp_;
      p.w_bstride = (long)D * fold_kp_;
      with_res(p, d_.h, D);
      p.res_bstride = (long)T * D;
      out_f32(p, d_.h, D);
      p.f32_bstride = (long)T * D;
      trace("  probs", d_.probs, (size_t)M * fold_kp_, bf16_, st);
      trace("  ut", ut_l, (size_t)rows * D * fold_kp_, bf16_, st);
      SA_TRY(gemm(p, st, -1.0, SAMAUDIO_CLS_CWO));
    } else {
      SA_TRY(op("cross_attention", (2 * MD + (double)Mt * 2 * D) * esz_, 4.0 * M * Lt * D, st, [&] {
        return launch_cross_attention(d_.qc, w.c_q_norm, kv_l, kv_ld, d_.text_mask, d_.ca, bf16_, rows, T, Lt, H, eps, st, hd);
      }));
      GemmParams p = lin(d_.ca, D, w.c_wo, M, D, D);
      with_res(p, d_.h, D);
      out_f32(p, d_.h, D);
      if (x3(SAMAUDIO_CLS_CWO)) SA_TRY(gemm_x3(p, w.c_wo3, w.ktm3 & 8, st, SAMAUDIO_CLS_CWO));
      else SA_TRY(gemm(p, st, -1.0, SAMAUDIO_CLS_CWO));
    }
    trace("  qc", d_.qc, (size_t)M * D, bf16_, st);
    trace("  h after cross", d_.h, (size_t)M * D, false, st);
    // feed-forward branch
    SA_TRY(op("rmsnorm_mod", MD * (4 + (w13_pre ? 6 : esz_)), 0, st, [&] {
      if (w13_pre)
        return launch_rmsnorm_gs_split3(d_.h, d_.modgs + (2L * l + 1) * nt * 2 * D, gs_ld, d_.x3a, (int)M, D, T, eps, st);
      if (mod_gs)
        return launch_rmsnorm_gs(d_.h, d_.modgs + (2L * l + 1) * nt * 2 * D, gs_ld, d_.xn, bf16_, (int)M, D, T, eps, st,
                                 alt16(SAMAUDIO_CLS_W13));
      return launch_rmsnorm_mod(d_.h, w.ffn_norm, tab + 3 * D, tab + 4 * D, d_.t0, t6, 3 * D, 4 * D, d_.xn, bf16_, (int)M, D,
                                T, eps, st);
    }));
    SA_TRY(sentinel(14, d_.xn, !bf16_ ? 0 : (alt16(SAMAUDIO_CLS_W13) ? 2 : 1), M, D, D, st));
    {
      GemmParams p = lin(d_.xn, D, w.w13, M, 2 * F, D);
      p.swiglu = 1;
      out_act(p, d_.u, F);
      if (alt16(SAMAUDIO_CLS_W2)) p.flags |= 512;    // u is w2's operand
      ktm(p, w, 3);
      prefetch(p, w.w2, (double)D * F);
      if (x3(SAMAUDIO_CLS_W13)) {
        if (w2_pre) { p.out_act = d_.x3u; p.flags |= GEMM_FLAG_OUT_SPLIT3; }   // the SwiGLU epilogue writes w2's split operand itself
        SA_TRY(gemm_x3(p, w.w13_3, w.ktm3 & 16, st, SAMAUDIO_CLS_W13, w13_pre ? d_.x3a : nullptr));
      }
      else SA_TRY(gemm(p, st, -1.0, SAMAUDIO_CLS_W13));
      trace("  xn (ffn)", d_.xn, (size_t)M * D, bf16_, st);
      trace("  u", d_.u, (size_t)M * F, bf16_, st);
      p = lin(d_.u, F, w.w2, M, D, F);  // out = h + gate_mlp * ff
      p.gate_tab = tab + 5 * D; p.gate = d_.t0 + 5 * D; p.gate_ld = t6; p.rows_per_gate = T;
      with_res(p, d_.h, D);
      out_f32(p, d_.h, D);
      ktm(p, w, 4);
      if (l + 1 < cfg_.n_layers) prefetch(p, layers_[l + 1].wqkv, 3.0 * D * D);
      if (x3(SAMAUDIO_CLS_W2)) SA_TRY(gemm_x3(p, w.w2_3, w.ktm3 & 32, st, SAMAUDIO_CLS_W2, w2_pre ? d_.x3u : nullptr));
      else SA_TRY(gemm(p, st, -1.0, SAMAUDIO_CLS_W2));
      trace("  h after ffn", d_.h, (size_t)M * D, false, st);
    }
  }
  trace("h after layers", d_.h, (size_t)M * D, false, st);
  // final modulated norm + output projection                         (transformer.py:507-519)
  {
    const bool f = f32c(SAMAUDIO_CLS_OUT);   // exact fp32: the result is the ODE's vector field itself
    void* xn = f ? (void*)d_.xn32 : d_.xn;
    SA_HIP(launch_rmsnorm_mod(d_.h, g_.final_norm, g_.final_table, g_.final_table + D, d_.t_emb, t1, 0, 0, xn,
                              f ? false : bf16_, (int)M, D, T, eps, st));
    GemmParams p = lin(xn, D, f ? (const void*)g32_.w_out : g_.w_out, M, C2, D);
    p.alpha = alpha;
    if (res) with_res(p, res, C2);
    out_f32(p, out, C2);
    SA_TRY(gemm(p, st, -1.0, SAMAUDIO_CLS_OUT, f));
  }
  hash_stage("field out", out, (size_t)M * C2 * 4, st);
  return Status{};
}

Status Engine::forward(const float* noisy, const float* time, int n_time, float* out, hipStream_t st) {
  if (!noisy || !time || !out) return fail(SAMAUDIO_ERR_ARG, "forward: null pointer");
  const Status s = eval_field(noisy, time, n_time, out, nullptr, 1.f, st);
  if (hash_on()) hash_flush((HashTrace*)hash_, this, st);
  return s;
}

Status Engine::ode_solve(float* y, int method, const float* grid, int n_grid, hipStream_t st) {
  if (!prepared_) return fail(SAMAUDIO_ERR_STATE, "ode_solve: call samaudio_prepare first");
  if (method != SAMAUDIO_ODE_EULER && method != SAMAUDIO_ODE_MIDPOINT)
    return fail(SAMAUDIO_ERR_ARG, "ode_solve: unsupported method");
  if (!y || !grid || n_grid < 2 || 2 * n_grid > 4096) return fail(SAMAUDIO_ERR_ARG, "ode_solve: bad grid");
  for (int k = 0; k + 1 < n_grid; ++k)
    if (!(grid[k + 1] > grid[k])) return fail(SAMAUDIO_ERR_ARG, "ode_solve: grid must be increasing");
  return solve_launches(y, method, grid, n_grid, st);
}

Status Engine::solve_launches(float* y, int method, const float* grid, int n_grid, hipStream_t st) {
  std::vector<float> ev(2 * (size_t)n_grid);
  for (int k = 0; k + 1 < n_grid; ++k) {
    ev[2 * k] = grid[k];
    ev[2 * k + 1] = (float)((double)grid[k] + 0.5 * ((double)grid[k + 1] - (double)grid[k]));
  }
  SA_HIP(launch_set_floats(d_.times, ev.data(), (int)ev.size(), st));   // as kernel arguments: no host synchronisation
  for (int k = 0; k + 1 < n_grid; ++k) {
    const float dt = (float)((double)grid[k + 1] - (double)grid[k]);
    if (method == SAMAUDIO_ODE_EULER) {
      SA_TRY(eval_field(y, d_.times + 2 * k, 1, y, y, dt, st));
    } else {
      SA_TRY(eval_field(y, d_.times + 2 * k, 1, d_.ymid, y, 0.5f * dt, st));
      SA_TRY(eval_field(d_.ymid, d_.times + 2 * k + 1, 1, y, y, dt, st));
    }
  }
  if (hash_on()) hash_flush((HashTrace*)hash_, this, st);
  return Status{};
}

// ---------------------------------------------------------------------------------------------------
// DAC-VAE: every Conv1d / ConvTranspose1d is one launch of the generalised GEMM over channels-last,
// halo-padded activations; Snake is fused into the producer's epilogue (raw f32 stream for residuals,
// activated copy as the next convolution's operand).
// ---------------------------------------------------------------------------------------------------
namespace {
struct SBuf {
  float* raw; void* act; void* tmp;
  long T; int C;
};
}  // namespace

static GemmParams conv_same(const void* x, long T, int Cin, int taps, int dil, const void* W, int Kp, int Cout,
                            int items) {
  GemmParams p = lin(x, Cin, W, T, Cout, Kp);
  p.kc = Cin;
  p.tap_stride = taps == 1 ? (long)Cin : (long)dil * Cin;
  p.a_off = (long)(HALO - (taps / 2) * dil) * Cin;
  p.a_bstride = (T + 2L * HALO) * Cin;
  p.nbatch = items;
  return p;
}
static void halo_out(GemmParams& p, float* raw, void* act, long T, int C, int actfn, const float* alpha) {
  if (raw) { p.out_f32 = raw; p.f32_bstride = (T + 2L * HALO) * C; p.f32_ld = C; p.f32_off = (long)HALO * C; }
  if (act) { p.out_act = act; p.act_bstride = (T + 2L * HALO) * C; p.act_ld = C; p.act_off = (long)HALO * C; }
  p.act = actfn;
  p.act_alpha = alpha;
}

Status Engine::codec_encode(const float* wav, int items, int64_t S, float* latent, hipStream_t st) {
  if (!enc_ready_) return fail(SAMAUDIO_ERR_STATE, "codec_encode: codec weights not finalized");
  long hop = 1;
  for (int i = 0; i < 4; ++i) hop *= cfg_.enc_rates[i];
  if (!wav || !latent || items <= 0 || S <= 0 || S % hop) return fail(SAMAUDIO_ERR_ARG, "codec_encode: samples % hop != 0");
  // codec_bytes(n) = n * per_item + a fixed 64 KiB: dividing the workspace by codec_bytes(1) would turn a workspace sized
  // for exactly n waveforms into passes of n - 1 and 1 (and the stray single-waveform pass runs at a fraction of the rate)
  const size_t fixed = (size_t)1 << 16;
  const size_t per_item = codec_bytes(1, S) - fixed;
  int chunk = ws_bytes_ > fixed ? (int)((ws_bytes_ - fixed) / (per_item ? per_item : 1)) : 0;
  if (!ws_ || chunk < 1) return fail(SAMAUDIO_ERR_WORKSPACE, "codec_encode: workspace too small");
  if (chunk > items) chunk = items;
  prepared_ = false;  // the codec scratch aliases the DiT scratch
  prof_cls_ = "codec";
  long encT[5], decT[5];
  int encC[5], decC[5];
  codec_stage_dims(cfg_, S, encT, encC, decT, decC);
  const int CL = cfg_.codec_latent, CD = cfg_.codec_dim;
  for (int i0 = 0; i0 < items; i0 += chunk) {
    const int n = items - i0 < chunk ? items - i0 : chunk;
    Bump b(ws_, ws_bytes_);
    void* in8 = b.take((size_t)n * (S + 2 * HALO) * 8 * esz_);
    SBuf sb[5];
    for (int i = 0; i < 5; ++i) {
      const size_t e = (size_t)n * (encT[i] + 2 * HALO) * encC[i];
      sb[i] = SBuf{(float*)b.take(e * 4), b.take(e * esz_), b.take(e * esz_), encT[i], encC[i]};
    }
    void* eout = b.take((size_t)n * (encT[4] + 2 * HALO) * CL * esz_);
    x3_codec_scratch_bytes_ = (size_t)n * codec_x3_per_item(S);
    x3_codec_scratch_ = x3_codec_scratch_bytes_ ? b.take(x3_codec_scratch_bytes_) : nullptr;
    if (!b.fits()) return fail(SAMAUDIO_ERR_WORKSPACE, "codec_encode: workspace too small");
    // waveform -> [n][HALO + S + HALO][8] (channel 0), zero halos
    SA_HIP(hipMemsetAsync(in8, 0, (size_t)n * (S + 2 * HALO) * 8 * esz_, st));
    SA_HIP(launch_to_act(wav + (long)i0 * S, S, 1, 0, in8, 0, bf16_, n, S, 1, 8, HALO, st));
    for (int i = 0; i < 5; ++i) {
      SA_HIP(launch_zero_halo(sb[i].act, bf16_, n, sb[i].T, sb[i].C, HALO, st));
      SA_HIP(launch_zero_halo(sb[i].tmp, bf16_, n, sb[i].T, sb[i].C, HALO, st));
    }
    SA_HIP(launch_zero_halo(eout, bf16_, n, encT[4], CL, HALO, st));
    {  // conv k7 (1 -> 64): window of 8 samples x 8 padded channels = one 64-wide row
      GemmParams p = lin(in8, 8, enc_.in_w, S, encC[0], 64);
      p.a_off = (long)(HALO - 3) * 8; p.a_bstride = (S + 2L * HALO) * 8; p.nbatch = n; p.bias = enc_.in_b;
      halo_out(p, sb[0].raw, sb[0].act, S, encC[0], ACT_SNAKE, enc_.s[0].r[0].a1);
      SA_TRY(gemm(p, st, 2.0 * S * encC[0] * 7 * n));
    }
    for (int i = 0; i < 4; ++i) {
      const StageW& sw = enc_.s[i];
      const long T = sb[i].T;
      const int C = sb[i].C, s = cfg_.enc_rates[i];
      const int dil[3] = {1, 3, 9};
      for (int j = 0; j < 3; ++j) {
        const ResUnitW& r = sw.r[j];
        GemmParams p = conv_same(sb[i].act, T, C, 7, dil[j], r.w1, r.k1pad, C, n);
        p.bias = r.b1;
        halo_out(p, nullptr, sb[i].tmp, T, C, ACT_SNAKE, r.a2);
        GemmParams q = conv_same(sb[i].tmp, T, C, 1, 1, r.w2, r.k2pad, C, n);
        q.bias = r.b2;
        q.res = sb[i].raw; q.res_bstride = (T + 2L * HALO) * C; q.res_ld = C; q.res_off = (long)HALO * C;
        halo_out(q, sb[i].raw, sb[i].act, T, C, ACT_SNAKE, j < 2 ? sw.r[j + 1].a1 : sw.a);
        SA_TRY(res_unit(p, q, sb[i].act, sb[i].tmp, 2.0 * T * C * 7 * C * n, 2.0 * T * C * C * n, st));
      }
      // strided conv k = 2s, stride s, pad s/2: the 2s input rows of one output are contiguous
      const int pad = (s + 1) / 2;
      GemmParams p = lin(sb[i].act, (long)s * C, sw.w, T / s, 2 * C, 2 * s * C);
      p.a_off = (long)(HALO - pad) * C; p.a_bstride = (T + 2L * HALO) * C; p.nbatch = n; p.bias = sw.b;
      halo_out(p, sb[i + 1].raw, sb[i + 1].act, T / s, 2 * C, ACT_SNAKE, i < 3 ? enc_.s[i + 1].r[0].a1 : enc_.out_a);
      SA_TRY(gemm(p, st));
    }
    {
      const long T = sb[4].T;
      const int C = sb[4].C;
      GemmParams p = conv_same(sb[4].act, T, C, 3, 1, enc_.out_w, 3 * C, CL, n);
      p.bias = enc_.out_b;
      halo_out(p, nullptr, eout, T, CL, ACT_NONE, nullptr);
      SA_TRY(gemm(p, st));
      p = conv_same(eout, T, CL, 1, 1, enc_.proj_w, CL, CD, n);  // quantizer.in_proj, mean half only
      p.bias = enc_.proj_b;
      p.out_f32 = latent + (long)i0 * T * CD; p.f32_bstride = T * CD; p.f32_ld = CD; p.f32_off = 0;
      SA_TRY(gemm(p, st));
    }
  }
  return Status{};
}

Status Engine::codec_decode(const float* latent, int items, int T0, float* wav, hipStream_t st, bool pairs) {
  if (!codec_ready_) return fail(SAMAUDIO_ERR_STATE, "codec_decode: codec weights not finalized");
  if (!latent || !wav || items <= 0 || T0 <= 0) return fail(SAMAUDIO_ERR_ARG, "codec_decode: bad argument");
  if (pairs && items % 2) return fail(SAMAUDIO_ERR_ARG, "codec_decode: the state layout holds (target, residual) pairs");
  long hop = 1;
  for (int i = 0; i < 4; ++i) hop *= cfg_.enc_rates[i];
  const int64_t S = (int64_t)T0 * hop;
  // codec_bytes(n) = n * per_item + a fixed 64 KiB: dividing the workspace by codec_bytes(1) would turn a workspace sized
  // for exactly n waveforms into passes of n - 1 and 1 (and the stray single-waveform pass runs at a fraction of the rate)
  const size_t fixed = (size_t)1 << 16;
  const size_t per_item = codec_bytes(1, S) - fixed;
  int chunk = ws_bytes_ > fixed ? (int)((ws_bytes_ - fixed) / (per_item ? per_item : 1)) : 0;
  if (!ws_ || chunk < 1) return fail(SAMAUDIO_ERR_WORKSPACE, "codec_decode: workspace too small");
  if (chunk > items) chunk = items;
  if (pairs && chunk > 1) chunk &= ~1;   // a pass holds whole pairs
  if (pairs && chunk < 2) return fail(SAMAUDIO_ERR_WORKSPACE, "codec_decode: workspace too small for one (target, residual) pair");
  prepared_ = false;
  prof_cls_ = "codec";
  long encT[5], decT[5];
  int encC[5], decC[5];
  codec_stage_dims(cfg_, S, encT, encC, decT, decC);
  const int CL = cfg_.codec_latent, CD = cfg_.codec_dim;
  for (int i0 = 0; i0 < items; i0 += chunk) {
    const int n = items - i0 < chunk ? items - i0 : chunk;
    Bump b(ws_, ws_bytes_);
    void* lat = b.take((size_t)n * (T0 + 2 * HALO) * CD * esz_);
    void* p0 = b.take((size_t)n * (T0 + 2 * HALO) * CL * esz_);
    SBuf sb[5];
    for (int i = 0; i < 5; ++i) {
      const size_t e = (size_t)n * (decT[i] + 2 * HALO) * decC[i];
      sb[i] = SBuf{(float*)b.take(e * 4), b.take(e * esz_), b.take(e * esz_), decT[i], decC[i]};
    }
    x3_codec_scratch_bytes_ = (size_t)n * codec_x3_per_item(S);
    x3_codec_scratch_ = x3_codec_scratch_bytes_ ? b.take(x3_codec_scratch_bytes_) : nullptr;
    if (!b.fits()) return fail(SAMAUDIO_ERR_WORKSPACE, "codec_decode: workspace too small");
    SA_HIP(launch_zero_halo(lat, bf16_, n, T0, CD, HALO, st));
    SA_HIP(launch_zero_halo(p0, bf16_, n, T0, CL, HALO, st));
    for (int i = 0; i < 5; ++i) {
      SA_HIP(launch_zero_halo(sb[i].act, bf16_, n, sb[i].T, sb[i].C, HALO, st));
      SA_HIP(launch_zero_halo(sb[i].tmp, bf16_, n, sb[i].T, sb[i].C, HALO, st));
    }
    if (pairs) {   // items (2b, 2b + 1) = channels [0, CD) / [CD, 2 CD) of state row block b: two strided gathers, no transposed copy
      const long item = (long)(T0 + 2 * HALO) * CD;
      for (int sgn = 0; sgn < 2; ++sgn)
        SA_HIP(launch_to_act(latent + (long)(i0 / 2) * T0 * 2 * CD, (long)T0 * 2 * CD, 2L * CD, sgn * CD, (char*)lat + (size_t)sgn * item * esz_,
                             2 * item, bf16_, n / 2, T0, CD, CD, HALO, st));
    } else
    SA_HIP(launch_to_act(latent + (long)i0 * T0 * CD, (long)T0 * CD, CD, 0, lat, 0, bf16_, n, T0, CD, CD, HALO, st));
    {
      GemmParams p = conv_same(lat, T0, CD, 1, 1, dec_.proj_w, CD, CL, n);  // quantizer.out_proj
      p.bias = dec_.proj_b;
      halo_out(p, nullptr, p0, T0, CL, ACT_NONE, nullptr);
      SA_TRY(gemm(p, st));
      p = conv_same(p0, T0, CL, 7, 1, dec_.in_w, 7 * CL, decC[0], n);
      p.bias = dec_.in_b;
      halo_out(p, nullptr, sb[0].act, T0, decC[0], ACT_SNAKE, dec_.s[0].a);
      SA_TRY(gemm(p, st));
    }
    for (int i = 0; i < 4; ++i) {
      const StageW& sw = dec_.s[i];
      const long Tin = sb[i].T, Tout = sb[i + 1].T;
      const int Cin = sb[i].C, C = sb[i + 1].C, s = cfg_.dec_rates[i];
      const int pad = (s + 1) / 2;
      {  // ConvTranspose1d(k=2s, stride s, pad s/2): out rows q*s + r - pad = x[q-1] W[r+s] + x[q] W[r]
        GemmParams p = lin(sb[i].act, Cin, sw.w, Tin + 1, s * C, 2 * Cin);
        p.a_off = (long)(HALO - 1) * Cin; p.a_bstride = (Tin + 2L * HALO) * Cin; p.nbatch = n;
        p.bias = sw.b; p.chan_mod = C;
        halo_out(p, sb[i + 1].raw, sb[i + 1].act, Tout, C, ACT_SNAKE, sw.r[0].a1);
        p.f32_ld = p.act_ld = (long)s * C;
        p.f32_off = p.act_off = (long)(HALO - pad) * C;
        p.c_ld_rel = (long)s * C; p.c_lo = (long)pad * C; p.c_hi = (Tout + pad) * (long)C;
        SA_TRY(gemm(p, st, 2.0 * Tout * C * 2 * Cin * n));
      }
      const int dil[3] = {1, 3, 9};
      for (int j = 0; j < 3; ++j) {
        const ResUnitW& r = sw.r[j];
        GemmParams p = conv_same(sb[i + 1].act, Tout, C, 7, dil[j], r.w1, r.k1pad, C, n);
        p.bias = r.b1;
        halo_out(p, nullptr, sb[i + 1].tmp, Tout, C, ACT_SNAKE, r.a2);
        GemmParams q = conv_same(sb[i + 1].tmp, Tout, C, 1, 1, r.w2, r.k2pad, C, n);
        q.bias = r.b2;
        q.res = sb[i + 1].raw; q.res_bstride = (Tout + 2L * HALO) * C; q.res_ld = C; q.res_off = (long)HALO * C;
        const float* next_alpha = j < 2 ? sw.r[j + 1].a1 : (i < 3 ? dec_.s[i + 1].a : dec_.out_a);
        halo_out(q, sb[i + 1].raw, sb[i + 1].act, Tout, C, ACT_SNAKE, next_alpha);
        SA_TRY(res_unit(p, q, sb[i + 1].act, sb[i + 1].tmp, 2.0 * Tout * C * 7 * C * n, 2.0 * Tout * C * C * n, st));
      }
    }
    {  // conv k7 (C -> 1) + tanh
      const long T = sb[4].T;
      const int C = sb[4].C;
      GemmParams p = conv_same(sb[4].act, T, C, 7, 1, dec_.out_w, dec_.out_kpad, 1, n);
      p.bias = dec_.out_b;
      p.act = ACT_TANH; p.f32_act = 1;
      p.out_f32 = wav + (long)i0 * T; p.f32_bstride = T; p.f32_ld = 1; p.f32_off = 0;
      SA_TRY(gemm(p, st, 2.0 * T * 7 * C * n));
    }
  }
  return Status{};
}

}  // namespace sa
