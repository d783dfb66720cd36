// Host-side orchestration of the T5 prompt encoder: embedding lookup ->
// layers x { T5LayerNorm (RMS, no bias), fused q|k|v (no bias), attention with the shared relative-position bias and
// no 1/sqrt(d) scaling, o (+residual) ; T5LayerNorm, wi (+ReLU / gelu_new), wo (+residual) } -> final T5LayerNorm.
// Reference: sam_audio/model/text_encoder.py:19-37 (`transformers.T5EncoderModel`); restated in oracle/t5_oracle.py.
// Like Engine / VisionTower it owns no device memory: borrowed weights, one caller-provided workspace.
#pragma once
#include "peav.h"

namespace sa {

class T5Encoder {
 public:
  explicit T5Encoder(const samaudio_t5_config& c);
  Status set_tensor(const char* name, const void* p, int dtype, int ndim, const int64_t* shape);
  Status finalize();
  size_t workspace_bytes(int rows, int tokens);
  Status set_workspace(void* p, size_t bytes);
  Status encode(const long long* ids, const unsigned char* mask, int rows, int tokens, float* out, hipStream_t st);

 private:
  void plan(Bump& b, long M, bool assign);
  samaudio_t5_config cfg_;
  bool bf16_;
  size_t esz_;
  int at_dtype_;
  int inner_;
  Registry reg_;
  bool ready_ = false;
  char* ws_ = nullptr;
  size_t ws_bytes_ = 0;
  long planned_m_ = 0;
  struct LayerW {
    const float *ln1, *ln2;
    const void *wqkv, *wo, *wi, *wo2;
  };
  std::vector<LayerW> layers_;
  struct {
    const float *emb, *rel_bias, *final_ln;
  } g_{};
  struct {
    float* h;
    void *xn, *qkv, *attn, *u;
  } w_{};
};

}  // namespace sa
