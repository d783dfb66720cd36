// Host-side orchestration of the reranking / span-prediction rows: the PE-AV
// transformer (used twice by the Judge and once by PE-A-Frame), the Judge's glue (reference
// sam_audio/model/judge.py:90-132) and the PE-A-Frame frame logits.  Like Engine, these classes own no device
// memory: weights are borrowed, scratch is one caller-provided workspace.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "engine.h"

namespace sa {

class Registry {  // name -> borrowed weight tensor
 public:
  Status set(const char* name, const void* p, int dtype, int ndim, const int64_t* shape);
  Status need(const std::string& name, int dtype, std::vector<int64_t> shape, const void** out) const;
  bool has(const std::string& name) const { return tensors_.count(name) != 0; }

 private:
  std::map<std::string, TensorRef> tensors_;
};

// One PE-AV transformer: input projection -> [class token ; frames] -> ResNet block with masked GroupNorm ->
// n_layers x (RMSNorm, qk-norm RoPE attention, RMSNorm, SwiGLU) -> RMSNorm -> output projection.
class PeavEncoder {
 public:
  PeavEncoder(const samaudio_peav_dims& d, bool bf16, std::string prefix);
  Status finalize(const Registry& reg);
  void plan(Bump& b, int rows, int frames, bool assign);
  // x_act [rows, frames, in_dim] GEMM-operand dtype; pad_mask [rows, frames] u8 (1 = valid) or null.
  Status forward(const void* x_act, const unsigned char* pad_mask, int rows, int frames, hipStream_t st);
  // results of the last forward: [rows][frames + 1][dim], row 0 of each item = class token (pooler_output)
  const float* out_f32() const { return w_.out; }
  const void* out_act() const { return w_.out_act; }
  const unsigned char* seq_mask() const { return w_.mask_s; }  // [rows][frames + 1]
  int dim() const { return d_.dim; }
  int in_dim() const { return d_.in_dim; }

 private:
  samaudio_peav_dims d_;
  bool bf16_;
  size_t esz_;
  std::string prefix_;
  bool ready_ = false;
  struct LayerW {
    const float *attn_norm, *ffn_norm, *q_norm, *k_norm, *bqkv, *bo;
    const void *wqkv, *wo, *w13, *w2;
  };
  std::vector<LayerW> layers_;
  struct {
    const void *in_w, *conv1_w, *conv2_w, *out_w;
    const float *in_b, *cls, *gn1_w, *gn1_b, *gn2_w, *gn2_b, *conv1_b, *conv2_b, *norm, *rope_cos, *rope_sin;
  } g_{};
  struct {
    float *h0, *r1, *h, *out;
    void *out_act, *xn, *qkv, *Q, *K, *Vt, *attn, *u, *gnbuf;
    unsigned char* mask_s;
    double* gn_part;
  } w_{};
};

class Judge {
 public:
  explicit Judge(const samaudio_judge_config& c);
  Status set_tensor(const char* name, const void* p, int dtype, int ndim, const int64_t* shape);
  Status finalize();
  size_t workspace_bytes(int inputs, int candidates, int frames);
  Status set_workspace(void* p, size_t bytes);
  // reference judge.py:90-132 with the mixture branch evaluated once per clip instead of once per candidate
  // (ranking/judge.py:31-33 repeats it; every op is per-row, so this is exact):
  //   input_latent [inputs, frames, codec_dim], separated_latent [inputs*candidates, frames, codec_dim] f32,
  //   text_pooled [inputs*candidates, text_hidden] f32, pad_mask [inputs, frames] u8 or null,
  //   scores [inputs*candidates, 4] f32 = (overall, recall, precision, faithfulness)
  Status score(const float* input_latent, const float* separated_latent, int inputs, int candidates, int frames,
               const float* text_pooled, const unsigned char* pad_mask, float* scores, hipStream_t st);
  // test hook: one transformer alone; x [rows, frames, in_dim] f32 -> hidden [rows, frames + 1, dim] f32
  Status encode(int which, const float* x, const unsigned char* pad_mask, int rows, int frames, float* hidden,
                hipStream_t st);

 private:
  void plan(Bump& b, int inputs, int candidates, int frames, bool assign);
  samaudio_judge_config cfg_;
  bool bf16_;
  size_t esz_;
  int at_dtype_;
  Registry reg_;
  PeavEncoder enc_, fin_;
  bool ready_ = false;
  char* ws_ = nullptr;
  size_t ws_bytes_ = 0;
  struct {
    const void *cat_wh, *cat_wi, *tp1_w, *tp2_w, *pat_wa, *pat_wt;
    const float *cat_b, *tp2_b, *ln_w, *ln_b, *pat_b, *head_w, *mean, *std_;
  } g_{};
  struct {
    void *xa, *audio, *tp_act, *t1, *tl, *at;
    float *inp_part, *t2, *tpart;
    unsigned char* mask;
  } w_{};
};

class FramePredictor {  // PE-A-Frame span predictor: per-frame audio-text logits for batch-paired rows
 public:
  explicit FramePredictor(const samaudio_frame_config& c);
  Status set_tensor(const char* name, const void* p, int dtype, int ndim, const int64_t* shape);
  Status finalize();
  size_t workspace_bytes(int rows, int frames);
  Status set_workspace(void* p, size_t bytes);
  // codec_features [rows, frames, codec_dim] f32, text_pooled [rows, embed_dim] f32, pad_mask [rows, frames] u8 or
  // null -> logits [rows, frames] f32
  Status logits(const float* codec_features, const float* text_pooled, const unsigned char* pad_mask, int rows,
                int frames, float* out, hipStream_t st);

 private:
  void plan(Bump& b, int rows, int frames, bool assign);
  samaudio_frame_config cfg_;
  bool bf16_;
  size_t esz_;
  int at_dtype_;
  Registry reg_;
  PeavEncoder enc_;
  bool ready_ = false;
  char* ws_ = nullptr;
  size_t ws_bytes_ = 0;
  struct {
    const void *ah_w, *th_w;
    const float *ah_ln_w, *ah_ln_b, *th_ln_w, *th_ln_b, *scale, *bias;
  } g_{};
  struct {
    void *xa, *a_ln, *t_ln;
    float *a_emb, *t_emb;
  } w_{};
};

}  // namespace sa
