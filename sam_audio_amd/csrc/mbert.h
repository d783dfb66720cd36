// Host-side orchestration of the ModernBERT text tower of the Judge reranker and the PE-A-Frame span predictor (SURVEY.md
// section 8 rows a17 / a18; reference sam_audio/model/judge.py:48,74-88 -> transformers' ModernBertModel, restated in
// oracle/mbert_oracle.py): embedding lookup -> LayerNorm -> layers x { LayerNorm (identity in layer 0), fused q|k|v,
// rotate-half RoPE (theta per layer type), attention (every n-th layer global, the others limited to +-window tokens), o
// (+residual) ; LayerNorm, Wi, gelu(input) * gate, Wo (+residual) } -> final LayerNorm.  No biases anywhere.
// Like Engine / VisionTower / T5Encoder it owns no device memory: borrowed weights, one caller-provided workspace.
#pragma once
#include "peav.h"

namespace sa {

class MBertEncoder {
 public:
  explicit MBertEncoder(const samaudio_mbert_config& c);
  Status set_tensor(const char* name, const void* p, int dtype, int ndim, const int64_t* shape);
  Status finalize();
  size_t workspace_bytes(int rows, int tokens);
  Status set_workspace(void* p, size_t bytes);
  // transformers' hidden_states[nth]: 0 <= nth < layers = the residual stream after `nth` layers (0 = the normalised
  // embeddings); nth == layers or nth < 0 = last_hidden_state (after the final LayerNorm: transformers 5 records the
  // normalised tensor as the last hidden state)
  Status encode(const long long* ids, const unsigned char* mask, int rows, int tokens, int nth, float* out, hipStream_t st);

 private:
  void plan(Bump& b, long M, bool assign);
  samaudio_mbert_config cfg_;
  bool bf16_;
  size_t esz_;
  int at_dtype_;
  int hd_;
  Registry reg_;
  bool ready_ = false;
  char* ws_ = nullptr;
  size_t ws_bytes_ = 0;
  long planned_m_ = 0;
  struct LayerW {
    const float *ln1, *ln2;   // ln1 is null in layer 0 (attn_norm = Identity)
    const void *wqkv, *wo, *wi, *wo2;
  };
  std::vector<LayerW> layers_;
  struct {
    const float *emb, *emb_ln, *final_ln, *zeros, *rope_cos, *rope_sin;   // rope tables [2][max_len][hd]: 0 global, 1 local
  } g_{};
  struct {
    float *h, *e;
    void *xn, *qkv, *attn, *u, *u2;
  } w_{};
};

}  // namespace sa
