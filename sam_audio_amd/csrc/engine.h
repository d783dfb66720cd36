// Host-side orchestration of the separate() hot path: owns no device memory, sequences the kernels.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/samaudio.h"
#include "kernels.h"

namespace sa {

struct TensorRef {
  const void* p = nullptr;
  int dtype = 0;
  std::vector<int64_t> shape;
};

struct Status {
  int code = 0;
  std::string msg;
  bool ok() const { return code == 0; }
};

void set_last_error(const std::string& msg);  // api.hip: the thread-local string behind samaudio_last_error()

constexpr int HALO = 40;  // zero rows either side of codec activations (>= 4 * max dilation 9)

class Bump {  // workspace carving (also used dry to size the workspace)
 public:
  explicit Bump(char* base = nullptr, size_t cap = 0) : base_(base), cap_(cap) {}
  void* take(size_t bytes) {
    size_t off = (used_ + 255) & ~size_t(255);
    used_ = off + bytes;
    return base_ ? base_ + off : nullptr;
  }
  size_t used() const { return (used_ + 255) & ~size_t(255); }
  bool fits() const { return used() <= cap_; }
  void reset_to(size_t mark) { used_ = mark; }
  size_t mark() const { return used_; }

 private:
  char* base_;
  size_t cap_;
  size_t used_ = 0;
};

class Engine {
 public:
  explicit Engine(const samaudio_config& c);
  Status set_tensor(const char* name, const void* p, int dtype, int ndim, const int64_t* shape);
  Status finalize(int what);
  Status set_option(int option, int value);
  Status check_f32_weights(int classes) const;   // every "<name>.f32" operand copy the classes read is registered
  size_t workspace_bytes(int rows, int frames, int text_len, int codec_items, int64_t samples);
  Status set_workspace(void* p, size_t bytes);

  // `candidates` > 1: every conditioning tensor holds rows / candidates clips and serves `candidates` consecutive rows each
  // (reference model.py:193-203, sample-major); `latent_feats`: feats is the codec latent z [.., frames, latent_channels / 2] and the
  // audio features are (z | z) (model.py:182-184) - read twice through a zero tap stride, never materialised
  Status prepare(int rows, int frames, int text_len, const float* feats, const float* text, const uint8_t* text_mask,
                 const float* video, const int64_t* anchor_ids, int n_ids, const int64_t* anchor_alignment,
                 const uint8_t* pad_mask, hipStream_t st, int candidates = 1, bool latent_feats = false);
  Status forward(const float* noisy, const float* time, int n_time, float* out, hipStream_t st);
  Status ode_solve(float* state, int method, const float* grid_host, int n_grid, hipStream_t st);
  Status codec_encode(const float* wav, int items, int64_t samples, float* latent, hipStream_t st);
  // `pairs`: latent is the ODE state [items / 2, frames, 2 * codec_dim] - item 2b = the first codec_dim channels of row b (target),
  // 2b + 1 the second (residual): reference model.py:291-295 without the transposed copy
  Status codec_decode(const float* latent, int items, int frames, float* wav, hipStream_t st, bool pairs = false);

  // Per-kernel timing with HIP events on the launch stream (bench.py's roofline leg): between begin and end
  // every GEMM launch is bracketed by an event pair; end synchronises and folds them per tile variant.
  // name = "<class>/<kernel>": class dit | codec | prep (conditioning hoisted out of the ODE); GEMM launches carry their
  // algorithmic flops AND bytes (operands + outputs + residual, each counted once), the streaming kernels their bytes.
  struct KernelStat {
    std::string name;
    long launches = 0;
    double flops = 0, bytes = 0, ms = 0;
  };
  Status sentinel_read(float* absmax, double* nonfinite, hipStream_t st);   // SAMAUDIO_OPT_SENTINEL
  Status profile_begin();
  Status profile_end(std::vector<KernelStat>& out);
  ~Engine();

 private:
  struct DitBuffers;
  struct DitW;
  struct CodecW;
  // one field evaluation; out = res + alpha * v(noisy, t) (res may be null)
  Status eval_field(const float* noisy, const float* time, int n_time, float* out, const float* res, float alpha,
                    hipStream_t st);
  Status plan_dit(Bump& b, int rows, int frames, int text_len, bool assign);
  size_t codec_bytes(int items, int64_t samples) const;
  // alg_flops < 0: 2*M*N*K*nbatch (exact unless K carries zero padding, then the caller passes the true count).
  // cls: SAMAUDIO_CLS_* bit of the launch (0: codec launches are classed by prof_cls_); f32: exact-fp32 operands inside a
  // 16-bit context (SAMAUDIO_OPT_F32_CLASSES - the caller hands fp32 A / W / out_act pointers)
  // mode 2: a SAMAUDIO_OPT_X3_CLASSES launch inside an fp32 context (16-bit A / W over K' = 3K, fp32 outputs; gemm_x3 builds it), both operands split over the
  // whole K (A rows [lo | hi | hi], W rows [W_hi | W_lo | W_hi]: the launch may share operand tiles, common.h GEMM_FLAG_X3_SHARE); mode 3: the same
  // with K' split per input block (the convolutions of gemm_codec_x3 / the patcher: [block][3 Cin]) - a plain walk over K' only
  Status gemm(const GemmParams& p, hipStream_t st, double alg_flops = -1.0, int cls = 0, int mode = 0);
  // SAMAUDIO_OPT_X3_CLASSES: `p` = the fp32 context's plain launch (fp32 A rows, fp32-typed outputs) of a class that is switched
  // on; `w3` = its "<name>.x3" weight.  Splits A into the scratch operand [lo | hi | hi] and runs ONE 16-bit GEMM over K' = 3K.
  // `presplit`: the activation operand is already in its split form at that address (written by the kernel that produced it)
  Status gemm_x3(GemmParams p, const void* w3, bool ktm, hipStream_t st, int cls, const void* presplit = nullptr);
  bool x3(int cls) const { return !bf16_ && (x3_classes_ & cls) != 0; }
  // SAMAUDIO_OPT_X3_CLASSES bit CODEC, convolutions with >= 256 output channels whose weight has a registered "<name>.x3" twin
  // ([N, K / Cin, 3 Cin]: every Cin-block of a weight row as [W_hi | W_lo | W_hi]): the fp32 activation buffer is split row by row
  // into a scratch operand and the launch runs on the 8-phase 16-bit kernels over K' = 3K; the activation is applied to the raw fp32
  // result by an elementwise kernel.  Everything else of the codec multiplies on operands split in registers (GEMM_FLAG_X3_FLY).
  struct X3CodecW { const void* w; int cin; };
  std::map<const void*, X3CodecW> x3_codec_;
  std::map<const void*, const void*> fly_codec_;   // fp32 codec weight -> its "<name>.fly" twin (GEMM_FLAG_W_FLY16)
  void* x3_codec_scratch_ = nullptr;
  size_t x3_codec_scratch_bytes_ = 0;
  size_t codec_x3_per_item(int64_t samples) const;
  Status gemm_codec_x3(const GemmParams& p, const X3CodecW& w, hipStream_t st, double alg_flops);
  Status check_x3_weights(int classes) const;
  bool f32c(int cls) const { return bf16_ && (f32_classes_ & cls) != 0; }
  bool alt16(int cls) const { return bf16_ && (alt_classes_ & cls) != 0; }   // SAMAUDIO_OPT_ALT16_CLASSES (mixed mode)
  const void* opt(const std::string& name, std::vector<int64_t> shape) const;  // optional fp32 tensor, null if absent / mis-shaped
  struct ProfRec {
    std::string key;
    double flops, bytes;
    hipEvent_t e0, e1;
  };
  const char* prof_cls_ = "dit";
  // non-GEMM launch, event-bracketed while profiling
  template <class F>
  Status op(const char* name, double alg_bytes, double alg_flops, hipStream_t st, F&& launch);
  Status res_unit(GemmParams p, GemmParams q, void*& cur, void*& alt, double flops7, double flops1, hipStream_t st);
  void* hash_ = nullptr;   // SAMAUDIO_TRACE_HASH recorder (engine.hip HashTrace; debugging aid)
  bool prof_on_ = false;
  std::vector<ProfRec> prof_;
  std::vector<hipEvent_t> ev_pool_;
  size_t ev_used_ = 0;
  Status prof_event(hipEvent_t* e);
  const TensorRef* find(const std::string& name) const;
  Status need(const std::string& name, int dtype, std::vector<int64_t> shape, const void** out);

  samaudio_config cfg_;
  bool bf16_;
  bool tail_split_ = true;  // SAMAUDIO_OPT_TAIL_SPLIT
  int f32_classes_ = 0;     // SAMAUDIO_OPT_F32_CLASSES (16-bit contexts)
  int alt_classes_ = 0;     // SAMAUDIO_OPT_ALT16_CLASSES (16-bit contexts)
  int prefetch_rows_ = 0;   // SAMAUDIO_OPT_PREFETCH_ROWS (16-bit contexts)
  int x3_classes_ = 0;      // SAMAUDIO_OPT_X3_CLASSES (fp32 contexts)
  Status solve_launches(float* y, int method, const float* grid, int n_grid, hipStream_t st);   // the launches of one solve
  bool sentinel_on_ = false;   // SAMAUDIO_OPT_SENTINEL
  float* sentinel_dev_ = nullptr;   // [SAMAUDIO_SENTINEL_SLOTS][2] slots + [kSentinelPartials][2] partials (debug_device_alloc)
  // fold the scan of a tensor into `slot` (no-op unless the sentinel is on); fmt as launch_sentinel
  Status sentinel(int slot, const void* x, int fmt, long rows, int cols, long ld, hipStream_t st);
  int quant_classes_ = 0, quant_fmt_ = 0;  // SAMAUDIO_OPT_QUANT_CLASSES / _FORMAT (fp32 contexts)
  size_t esz_;  // bytes per activation / GEMM-operand element
  int at_dtype_;
  std::map<std::string, TensorRef> tensors_;
  bool dit_ready_ = false, codec_ready_ = false, enc_ready_ = false, prepared_ = false;
  char* ws_ = nullptr;
  size_t ws_bytes_ = 0;
  int rows_ = 0, frames_ = 0, text_len_ = 0, frames_pad_ = 0;
  bool fold3_ = false;   // the fold runs on compensated operands (x3 context, class CWO)
  int fold_ltp_ = 0, fold_kp_ = 0;  // folded cross-attention: tokens per head slot (8 | 16; 0 = not folded), padded K
  bool has_anchor_ = false;

  // resolved weights (pointers into caller memory)
  struct LayerW {
    const float *attn_norm, *ffn_norm, *mod_table, *q_norm, *k_norm, *c_q_norm;
    const void *wqkv, *wo, *c_wq, *c_wo, *w13, *w2;
    int ktm;   // which of (wqkv, wo, c_wq, w13, w2) - bits 0..4 - are registered K-tile-major [K/64][N][64] (samaudio.h)
    // SAMAUDIO_OPT_X3_CLASSES (fp32 contexts): the "<name>.x3" split weights [W_hi | W_lo | W_hi], 16-bit; null = not registered
    const void *wqkv3, *wo3, *c_wq3, *c_wo3, *w13_3, *w2_3;
    int ktm3;  // which of them - bits 0..5 in that order - are K-tile-major [3K/64][N][64]
  };
  // a big-five weight: [N, K] row-major or (16-bit contexts) [K/64, N, 64] K-tile-major
  Status need_w5(const std::string& name, int N, int K, const void** out, int* ktm_bits, int bit);
  std::vector<LayerW> layers_;
  struct {
    const float *final_table, *final_norm, *gn1_w, *gn1_b, *gn2_w, *gn2_b, *pb1, *pb2, *tb_b, *t_freqs, *mem_inv_freq,
        *rope_cos, *rope_sin, *proj_b, *mem_b, *vid_b, *vid_ln_w, *vid_ln_b, *vid_gate, *anc_emb, *c_k_norm_all;
    const void *w_out, *pw1, *pw2, *y_w13, *y_w2, *t_w13, *t_w2, *tb_w, *proj_wy, *proj_wf, *mem_w, *vid_w, *anc_w,
        *c_wkv_all;
  } g_;
  struct {  // SAMAUDIO_OPT_X3_CLASSES, classes PATCH and CKV: "patch1.w.x3", "patch2.w.x3" (per tap [W_hi | W_lo | W_hi]), "c_wkv_all.x3"
    const void *pw1, *pw2, *c_wkv_all;
    int ktm;   // bits 0..2 in that order: K-tile-major
  } g3_;
  struct {  // optional fp32 copies ("<name>.f32") of the weights of the SAMAUDIO_CLS_F32_CAPABLE classes
    const float *w_out, *t_w13, *t_w2, *tb_w, *proj_wy, *proj_wf, *mem_w, *vid_w, *anc_w, *y_w13, *y_w2;
  } g32_;
  struct ResUnitW {
    const float *a1, *b1, *a2, *b2;
    const void *w1, *w2;
    int k1pad, k2pad;
  };
  struct StageW {
    ResUnitW r[3];
    const float *a, *b;   // snake before the resampling conv, its bias
    const void* w;        // down conv [2C, 2s*C] (encoder)  /  up conv [s*Cout, 2*Cin] (decoder)
  };
  struct {
    const void *in_w, *out_w, *proj_w;
    const float *in_b, *out_a, *out_b, *proj_b;
    int out_kpad;
    StageW s[4];
  } enc_, dec_;

  // DiT workspace (assigned by plan_dit)
  struct {
    float *ymid, *aligned, *cond, *h, *hp1, *text_proj, *t_emb, *t0, *modgs, *tsin, *vtmp, *times;
    void *ybf, *xn, *qkv, *Q, *K, *Vt, *attn, *hbf, *qc, *ca, *u, *gnbuf, *mem, *yu, *yemb, *kvc, *temb, *tu, *tsilu,
        *feats, *text, *video, *anch, *probs, *ut, *x3a, *x3u, *x3p, *ut3;
    float *temb32, *tu32, *tsilu32, *xn32, *prep32, *mem32, *yu32, *yemb32;  // fp32 operands of the f32 classes (16-bit contexts)
    unsigned char *pad_mask, *text_mask;
    double* gn_part;
  } d_;
};

}  // namespace sa
