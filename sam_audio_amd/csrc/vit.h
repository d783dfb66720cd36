// Host-side orchestration of the PE-Core vision tower: patch embedding ->
// [class token ;] + positions -> ln_pre -> layers x { LayerNorm, fused q|k|v (+bias), 2-D RoPE, flash attention,
// out_proj (+bias, residual) ; LayerNorm, c_fc (+bias, GELU), c_proj (+bias, residual) } -> ln_post -> pooling
// (class token | mean | attention pooling head) -> projection -> optional L2 normalisation.
// Reference: sam_audio/model/vision_encoder.py:80-89 (`pe.CLIP.encode_image`); architecture restated in
// oracle/vit_oracle.py.  Like Engine it owns no device memory: borrowed weights, one caller-provided workspace.
#pragma once
#include "peav.h"

namespace sa {

class VisionTower {
 public:
  explicit VisionTower(const samaudio_vit_config& c);
  Status set_tensor(const char* name, const void* p, int dtype, int ndim, const int64_t* shape);
  Status finalize();
  size_t workspace_bytes(int frames);
  Status set_workspace(void* p, size_t bytes);
  Status encode(const float* frames, int n, bool normalize, float* features, float* tokens_out, hipStream_t st);

 private:
  void plan(Bump& b, int n, bool assign);
  int tokens() const { return grid_ * grid_ + (cfg_.use_cls_token ? 1 : 0); }
  samaudio_vit_config cfg_;
  bool bf16_;
  size_t esz_;
  int at_dtype_;
  int grid_, kp_, hd_, pool_hd_;
  Registry reg_;
  bool ready_ = false;
  char* ws_ = nullptr;
  size_t ws_bytes_ = 0;
  int planned_n_ = 0;
  struct LayerW {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *bqkv, *bo, *b1, *b2;
    const void *wqkv, *wo, *w1, *w2;
  };
  std::vector<LayerW> layers_;
  struct {
    const void *patch_w, *proj, *pool_wkv, *pool_wo, *pool_w1, *pool_w2;
    const float *pos, *ln_pre_w, *ln_pre_b, *ln_post_w, *ln_post_b, *rope_cos, *rope_sin;
    const float *pool_q, *pool_bkv, *pool_bo, *pool_ln_w, *pool_ln_b, *pool_b1, *pool_b2;
  } g_{};
  struct {
    void *patches, *xn, *qkv, *Q, *K, *Vt, *attn, *u, *kv, *pooled, *yn, *u2, *z_act;
    float *h, *y, *z;
    unsigned char* mask;
  } w_{};
};

}  // namespace sa
