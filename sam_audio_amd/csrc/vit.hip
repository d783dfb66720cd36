// PE-Core vision tower: host code that sequences kernels of gemm*.hip / attention.hip / peav_kernels.hip /
// vit_kernels.hip, and its C entry points (include/samaudio.h "visual-prompt tower").  "oracle:" = oracle/vit_oracle.py,
// the CPU restatement of the published architecture every step below is checked against.
#include "vit.h"

#include <cstring>

struct samaudio_vit {
  sa::VisionTower* tower;
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
Status vfail(int code, const std::string& m) { return Status{code, m}; }
long vround_up(long v, long m) { return (v + m - 1) / m * m; }

GemmParams vlin(const void* A, long lda, const void* W, long M, int N, int K) {
  GemmParams p;
  std::memset(&p, 0, sizeof(p));
  p.A = A; p.W = W; p.lda = lda; p.kc = K; p.tap_stride = 0;
  p.M = (int)M; p.N = N; p.K = K; p.nbatch = 1; p.alpha = 1.f; p.rows_per_gate = 1;
  return p;
}
Status vgemm(const GemmParams& p, bool bf16, hipStream_t st) {
  if (const char* why = gemm_check(p, bf16)) return vfail(SAMAUDIO_ERR_ARG, std::string("vision tower: ") + why);
  SA_HIP(launch_gemm(p, bf16, st));
  return Status{};
}
}  // namespace

VisionTower::VisionTower(const samaudio_vit_config& c)
    : cfg_(c), bf16_(c.precision == SAMAUDIO_BF16), esz_(bf16_ ? 2 : 4),
      at_dtype_(bf16_ ? SAMAUDIO_DT_BF16 : SAMAUDIO_DT_F32) {
  grid_ = c.patch_size > 0 ? c.image_size / c.patch_size : 0;
  kp_ = (int)vround_up(3L * c.patch_size * c.patch_size, 64);
  hd_ = c.heads > 0 ? c.width / c.heads : 0;
  pool_hd_ = c.pool_heads > 0 ? c.width / c.pool_heads : 0;
}

Status VisionTower::set_tensor(const char* name, const void* p, int dtype, int ndim, const int64_t* shape) {
  ready_ = false;
  return reg_.set(name, p, dtype, ndim, shape);
}

Status VisionTower::finalize() {
  const samaudio_vit_config& c = cfg_;
  if (c.image_size <= 0 || c.patch_size <= 0 || c.image_size % c.patch_size || c.width <= 0 || c.layers < 0 ||
      c.heads <= 0 || c.mlp_width <= 0 || c.output_dim <= 0)
    return vfail(SAMAUDIO_ERR_ARG, "vision tower: non-positive dimension / image_size % patch_size");
  if (c.width % 64 || c.mlp_width % 64 || c.output_dim % 4)
    return vfail(SAMAUDIO_ERR_ARG, "vision tower: width / mlp_width must be multiples of 64, output_dim of 4");
  if (c.width % c.heads || (hd_ != 64 && hd_ != 128)) return vfail(SAMAUDIO_ERR_ARG, "vision tower: head dim must be 64 or 128");
  if (c.pool_type < 0 || c.pool_type > 2) return vfail(SAMAUDIO_ERR_ARG, "vision tower: pool_type");
  if (c.pool_type == 0 && !c.use_cls_token) return vfail(SAMAUDIO_ERR_ARG, "vision tower: class-token pooling without a class token");
  if (c.pool_type == 2 && (c.pool_heads <= 0 || c.width % c.pool_heads || (pool_hd_ != 64 && pool_hd_ != 128)))
    return vfail(SAMAUDIO_ERR_ARG, "vision tower: pooling head dim must be 64 or 128");
  if (c.act != ACT_GELU && c.act != ACT_QUICK_GELU) return vfail(SAMAUDIO_ERR_ARG, "vision tower: act must be 4 (gelu) or 5 (quick gelu)");
  const int W = c.width, F = c.mlp_width, S = tokens();
  const int F32 = SAMAUDIO_DT_F32, AT = at_dtype_;
#define NEEDF(field, name, ...) SA_TRY(reg_.need(name, F32, {__VA_ARGS__}, (const void**)&(field)))
#define NEEDW(field, name, ...) SA_TRY(reg_.need(name, AT, {__VA_ARGS__}, (const void**)&(field)))
  NEEDW(g_.patch_w, "patch.w", W, kp_);        // conv1.weight [W,3,P,P] flattened, K zero-padded to a multiple of 64
  NEEDF(g_.pos, "pos", S, W);                  // positional_embedding (zeros without it); row 0 += class_embedding
  g_.ln_pre_w = g_.ln_pre_b = g_.ln_post_w = g_.ln_post_b = g_.rope_cos = g_.rope_sin = nullptr;
  if (c.use_ln_pre) { NEEDF(g_.ln_pre_w, "ln_pre.w", W); NEEDF(g_.ln_pre_b, "ln_pre.b", W); }
  if (c.use_ln_post) { NEEDF(g_.ln_post_w, "ln_post.w", W); NEEDF(g_.ln_post_b, "ln_post.b", W); }
  if (c.use_rope2d) { NEEDF(g_.rope_cos, "rope_cos", S, hd_ / 2); NEEDF(g_.rope_sin, "rope_sin", S, hd_ / 2); }
  layers_.assign(c.layers, LayerW{});
  for (int i = 0; i < c.layers; ++i) {
    const std::string L = "L" + std::to_string(i) + ".";
    LayerW& w = layers_[i];
    NEEDF(w.ln1_w, L + "ln1.w", W); NEEDF(w.ln1_b, L + "ln1.b", W);
    NEEDW(w.wqkv, L + "wqkv", 3 * W, W); NEEDF(w.bqkv, L + "bqkv", 3 * W);
    NEEDW(w.wo, L + "wo", W, W); NEEDF(w.bo, L + "bo", W);
    NEEDF(w.ln2_w, L + "ln2.w", W); NEEDF(w.ln2_b, L + "ln2.b", W);
    NEEDW(w.w1, L + "w1", F, W); NEEDF(w.b1, L + "b1", F);
    NEEDW(w.w2, L + "w2", W, F); NEEDF(w.b2, L + "b2", W);
  }
  if (c.pool_type == 2) {
    NEEDF(g_.pool_q, "pool.q", W);             // in_proj_q(probe): the same query for every frame
    NEEDW(g_.pool_wkv, "pool.wkv", 2 * W, W); NEEDF(g_.pool_bkv, "pool.bkv", 2 * W);
    NEEDW(g_.pool_wo, "pool.wo", W, W); NEEDF(g_.pool_bo, "pool.bo", W);
    NEEDF(g_.pool_ln_w, "pool.ln.w", W); NEEDF(g_.pool_ln_b, "pool.ln.b", W);
    NEEDW(g_.pool_w1, "pool.w1", F, W); NEEDF(g_.pool_b1, "pool.b1", F);
    NEEDW(g_.pool_w2, "pool.w2", W, F); NEEDF(g_.pool_b2, "pool.b2", W);
  }
  NEEDW(g_.proj, "proj", c.output_dim, W);     // proj^T (features = pooled @ proj)
#undef NEEDF
#undef NEEDW
  ready_ = true;
  return Status{};
}

void VisionTower::plan(Bump& b, int n, bool assign) {
  const long W = cfg_.width, F = cfg_.mlp_width, H = cfg_.heads, S = tokens(), Sp = vround_up(S, 128);
  const long M = (long)n * S;
  auto f32 = [&](long k) { return (float*)b.take((size_t)k * 4); };
  auto act = [&](long k) { return b.take((size_t)k * esz_); };
  float* h = f32(M * W);
  void* xn = act(M * W); void* qkv = act(M * 3 * W);
  void* Q = act((long)n * H * Sp * hd_); void* K = act((long)n * H * Sp * hd_); void* Vt = act((long)n * H * hd_ * Sp);
  void* attn = act(M * W);
  void* u = b.take((size_t)M * F * esz_ > (size_t)M * W * 4 ? (size_t)M * F * esz_ : (size_t)M * W * 4);  // also the pre-LN embedding (f32)
  void* patches = act((long)n * grid_ * grid_ * kp_);
  unsigned char* mask = (unsigned char*)b.take((size_t)M);
  void* kv = act(M * 2 * W); void* pooled = act((long)n * W); float* y = f32((long)n * W); void* yn = act((long)n * W);
  void* u2 = act((long)n * F); float* z = f32((long)n * W); void* z_act = act((long)n * W);
  if (assign) {
    w_.h = h; w_.xn = xn; w_.qkv = qkv; w_.Q = Q; w_.K = K; w_.Vt = Vt; w_.attn = attn; w_.u = u; w_.patches = patches;
    w_.mask = mask; w_.kv = kv; w_.pooled = pooled; w_.y = y; w_.yn = yn; w_.u2 = u2; w_.z = z; w_.z_act = z_act;
  }
}

size_t VisionTower::workspace_bytes(int n) {
  if (n <= 0) return 0;
  Bump b;
  plan(b, n, false);
  return b.used();
}

Status VisionTower::set_workspace(void* p, size_t bytes) {
  if (!p || (reinterpret_cast<uintptr_t>(p) & 255)) return vfail(SAMAUDIO_ERR_WORKSPACE, "vision tower: workspace must be 256-byte aligned");
  ws_ = (char*)p;
  ws_bytes_ = bytes;
  planned_n_ = 0;
  return Status{};
}

Status VisionTower::encode(const float* frames, int n, bool normalize, float* features, float* tokens_out, hipStream_t st) {
  if (!ready_) return vfail(SAMAUDIO_ERR_STATE, "vision tower: weights not finalized");
  if (!frames || !features || n <= 0) return vfail(SAMAUDIO_ERR_ARG, "vision tower: bad argument");
  if (!ws_) return vfail(SAMAUDIO_ERR_WORKSPACE, "vision tower: no workspace");
  if (planned_n_ != n) {
    Bump b(ws_, ws_bytes_);
    plan(b, n, true);
    if (!b.fits()) return vfail(SAMAUDIO_ERR_WORKSPACE, "vision tower: workspace too small for " + std::to_string(n) + " frames");
    planned_n_ = n;
  }
  const samaudio_vit_config& c = cfg_;
  const int W = c.width, F = c.mlp_width, H = c.heads, S = tokens(), Sp = (int)vround_up(S, 128), G2 = grid_ * grid_;
  const int cls = c.use_cls_token ? 1 : 0;
  const long M = (long)n * S;
  const float eps = c.ln_eps;
  float* emb = c.use_ln_pre ? (float*)w_.u : w_.h;  // the pre-LN embedding lives in the (still unused) MLP scratch

  // patch embedding: conv1 (k = stride = P, no bias) as one GEMM per frame over im2col rows, + position rows 1..   (oracle: conv2d, + positional_embedding)
  SA_HIP(launch_patchify(frames, w_.patches, bf16_, n, c.image_size, c.patch_size, kp_, st));
  {
    GemmParams p = vlin(w_.patches, kp_, g_.patch_w, G2, W, kp_);
    p.nbatch = n; p.a_bstride = (long)G2 * kp_;
    p.res = g_.pos; p.res_ld = W; p.res_off = (long)cls * W; p.res_bstride = 0;
    p.out_f32 = emb; p.f32_ld = W; p.f32_bstride = (long)S * W; p.f32_off = (long)cls * W;
    SA_TRY(vgemm(p, bf16_, st));
  }
  // class-token row (= class_embedding + positional_embedding[0], folded at load) and the all-valid key mask
  if (cls) {
    SA_HIP(launch_peav_cls_mask(emb, g_.pos, nullptr, w_.mask, n, S - 1, W, st));
  } else {
    SA_HIP(hipMemsetAsync(w_.mask, 1, (size_t)M, st));
  }
  if (c.use_ln_pre) SA_HIP(launch_layernorm_rows(emb, W, g_.ln_pre_w, g_.ln_pre_b, w_.h, nullptr, bf16_, M, W, eps, st));

  for (int l = 0; l < c.layers; ++l) {  // oracle: resblocks
    const LayerW& w = layers_[l];
    SA_HIP(launch_layernorm_rows(w_.h, W, w.ln1_w, w.ln1_b, nullptr, w_.xn, bf16_, M, W, eps, st));
    {
      GemmParams p = vlin(w_.xn, W, w.wqkv, M, 3 * W, W);
      p.bias = w.bqkv;
      p.out_act = w_.qkv; p.act_ld = 3L * W;
      SA_TRY(vgemm(p, bf16_, st));
    }
    SA_HIP(launch_rope2d_split(w_.qkv, g_.rope_cos, g_.rope_sin, w_.Q, w_.K, w_.Vt, bf16_, n, S, Sp, H, hd_, st));
    SA_HIP(launch_self_attention_hd(w_.Q, w_.K, w_.Vt, w_.mask, w_.attn, bf16_, n, S, Sp, H, hd_, st));
    {
      GemmParams p = vlin(w_.attn, W, w.wo, M, W, W);  // x = x + out_proj(attn)
      p.bias = w.bo;
      p.res = w_.h; p.res_ld = W;
      p.out_f32 = w_.h; p.f32_ld = W;
      SA_TRY(vgemm(p, bf16_, st));
    }
    SA_HIP(launch_layernorm_rows(w_.h, W, w.ln2_w, w.ln2_b, nullptr, w_.xn, bf16_, M, W, eps, st));
    {
      GemmParams p = vlin(w_.xn, W, w.w1, M, F, W);  // act(c_fc(x))
      p.bias = w.b1; p.act = c.act;
      p.out_act = w_.u; p.act_ld = F;
      SA_TRY(vgemm(p, bf16_, st));
      p = vlin(w_.u, F, w.w2, M, W, F);  // x = x + c_proj(...)
      p.bias = w.b2;
      p.res = w_.h; p.res_ld = W;
      p.out_f32 = w_.h; p.f32_ld = W;
      SA_TRY(vgemm(p, bf16_, st));
    }
  }
  if (tokens_out) SA_HIP(hipMemcpyAsync(tokens_out, w_.h, (size_t)M * W * 4, hipMemcpyDeviceToDevice, st));

  // ln_post on every token, then pooling                                                   (oracle: ln_post, _pool)
  const void* pooled_act = nullptr;  // [n, W] GEMM operand of the projection
  if (c.pool_type == 2) {
    if (c.use_ln_post) SA_HIP(launch_layernorm_rows(w_.h, W, g_.ln_post_w, g_.ln_post_b, nullptr, w_.xn, bf16_, M, W, eps, st));
    else SA_HIP(launch_to_act(w_.h, 0, W, 0, w_.xn, 0, bf16_, 1, M, W, W, 0, st));
    {
      GemmParams p = vlin(w_.xn, W, g_.pool_wkv, M, 2 * W, W);  // k | v of every token
      p.bias = g_.pool_bkv;
      p.out_act = w_.kv; p.act_ld = 2L * W;
      SA_TRY(vgemm(p, bf16_, st));
    }
    SA_HIP(launch_pool_attention(g_.pool_q, w_.kv, w_.pooled, bf16_, n, S, c.pool_heads, pool_hd_, st));
    {
      GemmParams p = vlin(w_.pooled, W, g_.pool_wo, n, W, W);  // y = out_proj(attention)
      p.bias = g_.pool_bo;
      p.out_f32 = w_.y; p.f32_ld = W;
      SA_TRY(vgemm(p, bf16_, st));
    }
    SA_HIP(launch_layernorm_rows(w_.y, W, g_.pool_ln_w, g_.pool_ln_b, nullptr, w_.yn, bf16_, n, W, eps, st));
    {
      GemmParams p = vlin(w_.yn, W, g_.pool_w1, n, F, W);
      p.bias = g_.pool_b1; p.act = c.act;
      p.out_act = w_.u2; p.act_ld = F;
      SA_TRY(vgemm(p, bf16_, st));
      p = vlin(w_.u2, F, g_.pool_w2, n, W, F);  // z = y + mlp(layernorm(y))
      p.bias = g_.pool_b2;
      p.res = w_.y; p.res_ld = W;
      p.out_f32 = w_.z; p.f32_ld = W;
      p.out_act = w_.z_act; p.act_ld = W;
      SA_TRY(vgemm(p, bf16_, st));
    }
    pooled_act = w_.z_act;
  } else if (c.pool_type == 0) {
    // class token: LayerNorm of row 0 of every frame (rows are S*W apart)
    if (c.use_ln_post) SA_HIP(launch_layernorm_rows(w_.h, (long)S * W, g_.ln_post_w, g_.ln_post_b, nullptr, w_.z_act, bf16_, n, W, eps, st));
    else SA_HIP(launch_to_act(w_.h, 0, (long)S * W, 0, w_.z_act, 0, bf16_, 1, n, W, W, 0, st));
    pooled_act = w_.z_act;
  } else {
    float* src = w_.h;
    if (c.use_ln_post) {
      SA_HIP(launch_layernorm_rows(w_.h, W, g_.ln_post_w, g_.ln_post_b, (float*)w_.u, nullptr, bf16_, M, W, eps, st));
      src = (float*)w_.u;
    }
    SA_HIP(launch_token_mean(src, W, nullptr, w_.z_act, bf16_, n, S, W, st));
    pooled_act = w_.z_act;
  }
  {
    GemmParams p = vlin(pooled_act, W, g_.proj, n, c.output_dim, W);  // features = pooled @ proj
    p.out_f32 = features; p.f32_ld = c.output_dim;
    SA_TRY(vgemm(p, bf16_, st));
  }
  if (normalize) SA_HIP(launch_l2_normalize(features, n, c.output_dim, st));
  return Status{};
}

}  // namespace sa

namespace {
int vret(const sa::Status& s) {
  if (!s.ok()) sa::set_last_error(s.msg);
  return s.code;
}
int vbad(const char* msg) {
  sa::set_last_error(msg);
  return SAMAUDIO_ERR_ARG;
}
}  // namespace

extern "C" {

int samaudio_vit_create(const samaudio_vit_config* cfg, samaudio_vit** out) {
  if (!cfg || !out) return vbad("samaudio_vit_create: null argument");
  if (cfg->precision != SAMAUDIO_F32 && cfg->precision != SAMAUDIO_BF16) return vbad("samaudio_vit_create: precision");
  samaudio_vit* v = new samaudio_vit;
  v->tower = new sa::VisionTower(*cfg);
  *out = v;
  return SAMAUDIO_OK;
}

void samaudio_vit_destroy(samaudio_vit* v) {
  if (!v) return;
  delete v->tower;
  delete v;
}

int samaudio_vit_set_tensor(samaudio_vit* v, const char* name, const void* data, int dtype, int ndim, const int64_t* shape) {
  if (!v) return vbad("null vision tower");
  return vret(v->tower->set_tensor(name, data, dtype, ndim, shape));
}

int samaudio_vit_finalize(samaudio_vit* v) {
  if (!v) return vbad("null vision tower");
  return vret(v->tower->finalize());
}

size_t samaudio_vit_workspace_bytes(samaudio_vit* v, int frames) {
  if (!v) return 0;
  return v->tower->workspace_bytes(frames);
}

int samaudio_vit_set_workspace(samaudio_vit* v, void* workspace, size_t bytes) {
  if (!v) return vbad("null vision tower");
  return vret(v->tower->set_workspace(workspace, bytes));
}

int samaudio_vit_encode(samaudio_vit* v, const float* frames, int n, int normalize, float* features, float* tokens_out,
                        samaudio_stream stream) {
  if (!v) return vbad("null vision tower");
  return vret(v->tower->encode(frames, n, normalize != 0, features, tokens_out, (hipStream_t)stream));
}

}  // extern "C"
