// Launchers of the non-GEMM kernels (kernels.hip, attention.hip).  `bf16` selects the activation
// element type AT: bf16_t (bf16 compute mode) or float (fp32 parity mode).
#pragma once
#include "common.h"

namespace sa {

// test aid: fill the LDS of every CU with NaN bit patterns (LDS persists between kernels)
hipError_t launch_poison_lds(hipStream_t st);

// test / tuning switches (samaudio_debug_set_flag); 0 = shipped path.  Round 3 removed the A/B generations that had lost
// their measurements (the logs are under profiles/, the code in the git history); what is left are hooks the tests use:
//   flag 11: k7 convolutions as implicit GEMMs (no conv7h kernel) - the bitwise-equality tests of conv7h
//   flag 16: DAC residual units as two launches (k7 + k1) instead of the fused resunit kernel - its bitwise-equality tests
//   flag 18: fuse residual units whatever the launch size (tests: small launches otherwise stay two launches)
//   flag 19: 1 = residual units on the weight-stationary kernel whatever the launch size (its tests; otherwise >= 1024
//            tiles), 3 = the same on 3 workgroups (small cases then walk several tiles each), 2 = never (ring kernel: A/B)
//   flag 21: gemm8s always in its plain double-buffered form (launches of <= 256 workgroups use the pipelined form)
//   flag 24: epilogue of the 8-phase family: 0 = shipped choice (register form for 16-bit-only outputs, LDS-staged lean form
//            for fp32 output / residual, general contract for everything else), 1 = the general epilogue for every launch
//            (bitwise-equality tests of the lean forms), 2 / 3 = the register / LDS form for every eligible launch (A/B)
//   flag 26: 1 = the 8-phase kernel launches one workgroup per tile (shipped: persistent above 256 tiles) - its bitwise test
//   flag 29: 1 = qkv_prep with its 16-bit rounding written out (the form before round 4: reproducer of the run-to-run difference its
//            SDWA instruction sequence showed beside another kernel's waves - kernels.hip, tools/stress_qkv_prep.py)
//   flag 27: wave roles of gemm8s' pipelined form (gemm8.hip): 0 = shipped choice, 1 = none (4 waves request and multiply, round 3),
//            2 = 4 requesting waves beside 4 multiplying ones, 3 = the same with the multiplying waves issuing 2 of their 8 loads
//   flag 33: (A/B) smallest last round, in 256x256 tiles, that is split off as a 128x128-tile tail launch (0 = shipped: 8)
//   flag 31: 1 = the folded cross-attention operand U = Wo V of every layer in its own launch (shipped: all layers of an evaluation in
//            one launch in front of the layer loop) - its bitwise test
//   flag 35: (A/B) M-tiles per raster group of the 8-phase family (0 = shipped: 8; GemmParams.raster_gm)
//   flag 36: 1 = (A/B, tests) split-weight launches of the fp32 kernel (GEMM_FLAG_W_FLY16) on the tiles of gemm1_variant instead of fly_variant's
//   flag 38: 1 = x3 launches walk K' = 3K as a plain GEMM (shipped: the operand-sharing order of GEMM_FLAG_X3_SHARE) - A/B, bitwise tests
//   flag 30: (A/B) number of 256x256 tiles from which the policy uses gemm8 instead of gemm8s (0 = shipped: 128)
//   flag 25: only in the ablation build (tools/build_abl.sh): selects an ablation of the round-3 8-phase loop
void set_debug_flag(int flag, int value);
// SAMAUDIO_TRACE_HASH debugging aid (engine.hip): per-item checksums of a buffer; the only device allocation of the library
hipError_t launch_hash_items(const unsigned* x, size_t words_per_item, int items, unsigned long long* out, hipStream_t st);
// SAMAUDIO_OPT_SENTINEL: absmax / non-finite scan of a tensor folded into slot[0..1] (kernels.hip)
constexpr int kSentinelPartials = 256;
hipError_t launch_sentinel(const void* x, int fmt, long rows, int cols, long ld, float* partial, float* slot, hipStream_t st);
void* debug_device_alloc(size_t bytes);
void debug_device_free(void* p);
int debug_flag(int flag);
void debug_touch();   // a process-wide debugging switch changed
unsigned long long debug_epoch();

hipError_t launch_gemm(const GemmParams& p, bool is_bf16, hipStream_t st);
const char* gemm_check(const GemmParams& p, bool is_bf16);
int gemm_variant(const GemmParams& p, bool is_bf16);       // which kernel / tile shape launch_gemm picks
const char* gemm_variant_name(int variant, bool is_bf16);
constexpr int kGemmVariants = 40;  // 36 .. 39 = the fp32 kernel's 4 x 1-wave tiles of split-weight launches (gemm.hip fly_variant);  // 35 = conv7h (k7 convolution, halo tile resident in LDS; conv7h_ok launches only)  // 0..2 gemm.hip tiles, 3.. = 3 + gemm2.hip variant; 25 / 26 = 128x128 / 64x128 tiles for small M
                                   // (32x32x16 family); 27 = gemm8s, the 128x128 tile of the 16x16x32 (8-phase) family;
                                   // 28 = 256x64 tile of the 32x32x16 family for 64-channel convolutions
// gemm2.hip: 256-row-tile bf16 kernels (variants 3.. in gemm_variant's numbering are gemm2 variants 0..)
bool gemm2_ok(const GemmParams& p);
hipError_t launch_gemm2(const GemmParams& p, int variant, hipStream_t st);
// gemm8.hip: 256x256 8-phase kernel (variant 22); needs gemm2_ok(p)
hipError_t launch_gemm8(const GemmParams& p, hipStream_t st);
bool gemm8_share_ok(const GemmParams& p);   // GemmParams.flags bit 15 is well-formed for this launch (gemm8.hip)
// gemm8.hip: 128x128 tile with gemm8's arithmetic (bitwise identical results), two workgroups per CU; needs gemm2_ok(p)
hipError_t launch_gemm8s(const GemmParams& p, hipStream_t st);
// gemm8.hip: GemmParams.flags bits 9 / 10 (mixed mode: out_act written / operands read in the alt 16-bit format) are well-formed;
// only the 8-phase family (variants 22 / 27) implements them
bool gemm8_alt_ok(const GemmParams& p);
// gemm8.hip: GemmParams.flags bit 12 (out_act in the compensated-operand form [lo | hi | hi]) is well-formed: SwiGLU launches of the
// 8-phase family with a 16-bit output only
bool gemm8_split3_ok(const GemmParams& p);
// gemm2.hip: dilated k = 7 'same' convolution C -> C (C = 64 / 96 / 128 / 192) with the activation halo tile resident in
// LDS; bitwise equal to the implicit GEMM of the 32x32x16 family
bool conv7h_ok(const GemmParams& p);
hipError_t launch_conv7h(const GemmParams& p, hipStream_t st);
// one DAC residual unit (k7 launch p + k1 launch q on its output) as ONE kernel, bitwise equal to the two launches
bool resunit_ok(const GemmParams& p, const GemmParams& q);
hipError_t launch_resunit(const GemmParams& p, const GemmParams& q, hipStream_t st);
// gemm8.hip: one GEMM as two launches - part 0: gemm8 on the first `full` 256x256 tiles (whole rounds of the chip),
// part 1: the rest as 128x128 quadrants on gemm8s.  gemm_tail_split() = `full` for a launch (0: no split).
hipError_t launch_gemm8_split(const GemmParams& p, int full, int part, hipStream_t st);
int gemm_tail_split(const GemmParams& p, bool is_bf16);
hipError_t launch_gemm_part(const GemmParams& p, bool is_bf16, int part, hipStream_t st);
// test / tuning hook: force a variant for every eligible bf16 GEMM (-1 = automatic, 0..2 = gemm.hip tiles only,
// 3.. = gemm2 variant when gemm2_ok)
void gemm_force_variant(int v);

// out[m,:] = AT( rmsnorm(x[m,:]) * w * (1 + scale) + shift ),  shift = shift_tab + tvec[b, shift_off:],
// scale likewise; b = m / rows_per_b; tvec_ld = 0 shares one conditioning row.  tvec == nullptr: no modulation.
hipError_t launch_rmsnorm_mod(const float* x, const float* w, const float* shift_tab, const float* scale_tab,
                              const float* tvec, long tvec_ld, int shift_off, int scale_off, void* out, bool bf16,
                              int M, int D, int rows_per_b, float eps, hipStream_t st);

// acc[m,:] += tanh(gate[0]) * (layernorm(x[m,:]) * w + b)
hipError_t launch_layernorm_accum(const float* x, const float* w, const float* b, const float* gate, float* acc,
                                  int M, int D, float eps, hipStream_t st);

// GroupNorm(1 group) over (T x C) per sample: partial sums, then normalise+affine+SiLU into a halo-padded
// channels-last buffer out[b][halo + t][c].
hipError_t launch_groupnorm_silu(const float* x, const float* w, const float* b, double* partials, void* out,
                                 bool bf16, int B, int T, int C, int halo, float eps, hipStream_t st);

// q/k: per-head RMSNorm (shared weight) + RoPE (adjacent pairs) -> Q,K [B,H,Tp,128]; v -> Vt [B,H,128,Tp]
// RMSNorm + modulate with pre-combined operands (kernels.hip): one launch per evaluation folds (w, shift / scale tables, the
// evaluation's shift / scale vectors) of up to kMaxModNorms norms into [norm][time value][g | s][D]
constexpr int kMaxModNorms = 96;
struct ModTables {
  const float* w[kMaxModNorms];
  const float* shift_tab[kMaxModNorms];
  const float* scale_tab[kMaxModNorms];
  int shift_off[kMaxModNorms], scale_off[kMaxModNorms];   // offsets of the norm's shift / scale inside a time vector
};
hipError_t launch_mod_tables(const ModTables& t, int n_norms, const float* tvec, long tvec_ld, int nt, float* gs, int D,
                             hipStream_t st);
hipError_t launch_rmsnorm_gs(const float* x, const float* gs, long gs_ld, void* out, bool bf16, int M, int D, int rows_per_b,
                             float eps, hipStream_t st, bool out_alt = false);   // out_alt: 16-bit output in the alt format (mixed mode)
// the same with the result in the compensated-operand form: out [M, 3D] 16-bit = [lo | hi | hi] (SAMAUDIO_OPT_X3_CLASSES)
hipError_t launch_rmsnorm_gs_split3(const float* x, const float* gs, long gs_ld, void* out, int M, int D, int rows_per_b, float eps,
                                    hipStream_t st);
hipError_t launch_qkv_prep(const void* qkv, const float* qw, const float* kw, const float* rope_cos,
                           const float* rope_sin, void* Q, void* K, void* Vt, bool bf16, int B, int T, int Tp, int H,
                           float eps, hipStream_t st, int head_dim = 128);   // head_dim 64 | 128 (rope tables [T, head_dim / 2])

// x3 contexts, head_dim 128: the same on fp32 tensors with the 16-lane-per-row access pattern of the 16-bit fast path (row statistics
// summed in another order than the general fp32 kernel: not the exact-fp32 parity mode's kernel)
hipError_t launch_qkv_prep_f32x(const float* qkv, const float* qw, const float* kw, const float* rope_cos, const float* rope_sin, float* Q,
                                float* K, float* Vt, int B, int T, int Tp, int H, float eps, hipStream_t st);

// self-attention over the padded layout above; key_mask [B,T] bytes (1 = attend); out [B*T, H*128]
hipError_t launch_self_attention(const void* Q, const void* K, const void* Vt, const unsigned char* key_mask,
                                 void* out, bool bf16, int B, int T, int Tp, int H, hipStream_t st, bool out_alt = false);

// the same with head_dim 64 or 128 (Q, K [B,H,Tp,hd], Vt [B,H,hd,Tp], out [B*T, H*hd]; scale hd^-0.5)
hipError_t launch_self_attention_hd(const void* Q, const void* K, const void* Vt, const unsigned char* key_mask,
                                    void* out, bool bf16, int B, int T, int Tp, int H, int head_dim, hipStream_t st,
                                    bool out_alt = false);

// fp32 contexts, SAMAUDIO_OPT_X3_CLASSES bit SAMAUDIO_X3_ATTENTION: the same contract on fp32 tensors with both contractions on the
// 16-bit MFMA over hi/lo-split operands (attention.hip self_attn_x3_kernel); Tp % 64 == 0
// out3 != nullptr: the context rows leave in the compensated-operand form instead, out3 [B*T, 3*H*hd] 16-bit = [lo | hi | hi]
hipError_t launch_self_attention_x3(const float* Q, const float* K, const float* Vt, const unsigned char* key_mask, float* out,
                                    int B, int T, int Tp, int H, int head_dim, hipStream_t st, void* out3 = nullptr);

// in-place per-(row, head) RMSNorm of x[rows, ld] columns [col0, col0 + H*128)
hipError_t launch_headnorm(void* x, const float* w, bool bf16, int rows, long ld, int col0, int H, float eps,
                           hipStream_t st, int head_dim = 128);

// cross attention: q [M, D] (raw, q-norm applied here), kv rows b*Lt + j with row stride kv_ld elements holding
// (k already normalised | v) in columns [0, 2D); mask [B, Lt] bytes; out [M, D]
hipError_t launch_cross_attention(const void* q, const float* qw, const void* kv, long kv_ld,
                                  const unsigned char* mask, void* out, bool bf16, int B, int T, int Lt, int H,
                                  float eps, hipStream_t st, int head_dim = 128);
// folded cross-attention output projection (bf16, Lt <= 16; see attention.hip): P [M, ldp] = softmax probabilities
// at column h*LtP + token; UT [B][D][KP] = per-batch weight operand of the GEMM h += P . U
hipError_t launch_cross_attn_probs(const void* q, const float* qw, const void* kv, long kv_ld, const unsigned char* mask,
                                   void* P, int ldp, int B, int T, int Lt, int LtP, int H, float eps, hipStream_t st);
hipError_t launch_cross_attn_fold(const void* wo, const void* kv, long kv_ld, void* UT, int KP, int B, int Lt, int LtP,
                                  int H, hipStream_t st);
// every layer of an evaluation in one launch: wo[l] = layer l's output projection, kv_all [B * Lt, kv_ld] with layer l's (k | v) in
// columns [l * 2 D, (l + 1) * 2 D), UT_all [n_layers][B][D][KP]; bitwise the per-layer launches
constexpr int kMaxFoldLayers = 48;
hipError_t launch_cross_attn_fold_layers(const void* const* wo, int n_layers, const void* kv_all, long kv_ld, void* UT_all, int KP,
                                         int B, int Lt, int LtP, int H, hipStream_t st);
// the same fold for x3 contexts (fp32 tensors in, compensated operands out; attention.hip): P3 [M, 3 KP] = [P_lo | P_hi | P_hi],
// UT3_all [n_layers][B][D][3 KP] = [U_hi | U_lo | U_hi]; q raw fp32 [M, D] (q-norm applied here), kv fp32 with k already normalised
hipError_t launch_cross_attn_probs3(const float* q, const float* qw, const float* kv, long kv_ld, const unsigned char* mask, void* P3,
                                    int KP, int B, int T, int Lt, int LtP, int H, float eps, hipStream_t st);
hipError_t launch_cross_attn_fold3_layers(const float* const* wo, int n_layers, const float* kv_all, long kv_ld, void* UT3_all, int KP,
                                          int B, int Lt, int LtP, int H, hipStream_t st);
// per-(row, layer, head) RMSNorm of the K halves of kv_all [rows, L*2D] (all layers' cross-attention keys at
// once); w_all [L, 128]
hipError_t launch_headnorm_layers(void* kv_all, const float* w_all, bool bf16, int rows, int L, int H, float eps,
                                  hipStream_t st, int head_dim = 128);

// timestep features: temb [nt, fdim] (AT) = cat(cos, sin)(t*freqs);  tsin [nt, D] fp32 likewise with inv_freq
hipError_t launch_time_features(const float* t, int nt, const float* freqs, int fdim, const float* inv_freq, int D,
                                void* temb, float* tsin, bool bf16, hipStream_t st);

// dst[0 .. n) = host_values, passed as kernel arguments (no host buffer has to outlive the call, no synchronisation)
struct FloatPack {
  static constexpr int N = 64;
  float v[N];
};
hipError_t launch_set_floats(float* dst, const float* host_values, int n, hipStream_t st);

// out[r,:] = AT(x[r,:] + vec[(r / rows_per_b) * vec_ld + :])
hipError_t launch_add_rowvec(const float* x, const float* vec, long vec_ld, void* out, bool bf16, int rows, int D,
                             int rows_per_b, hipStream_t st);

// out[r,:] = AT(emb[ids[b, align[b,t]], :])   (model.py:61: embed(anchor_ids.gather(1, anchor_alignment)))
hipError_t launch_anchor_gather(const float* emb, const long* ids, int n_ids, const long* align, void* out, bool bf16,
                                int B, int T, int E, int vocab, hipStream_t st);

// generic fp32 -> AT copy with optional halo layout: out[b][halo + t][c_out_pad] <- in[b][t][c_in] (extra channels 0)
// out_bstride in elements (0 = dense (T + 2*halo) * C_out)
hipError_t launch_to_act(const float* in, long in_bstride, long in_ld, int in_col0, void* out, long out_bstride,
                         bool bf16, int B, long T, int C_in, int C_out, int halo, hipStream_t st);

// compensated 16-bit GEMM operand (SAMAUDIO_OPT_X3_CLASSES): x fp32 [M, K] (row stride ldx) -> out [M, 3K] in the library's 16-bit
// format = [lo | hi | hi] per row, hi = rn16(x), lo = rn16(x - hi); K % 8 == 0
hipError_t launch_split3(const float* x, long ldx, void* out, long M, int K, hipStream_t st);

// out[b][e] = act(in[b][e]; alpha[e % chan]) for e < count (fp32, contiguous per item, in-place allowed; count, chan % 4 == 0)
hipError_t launch_act_flat(const float* in, long in_bstride, float* out, long out_bstride, int items, long count, int chan, int act,
                           const float* alpha, hipStream_t st);

// zero the halo rows of a [B][halo + T + halo][C] buffer
hipError_t launch_zero_halo(void* buf, bool bf16, int B, long T, int C, int halo, hipStream_t st);

// ---- PE-AV transformer / Judge / span predictor (peav_kernels.hip) ------------------------------------------
// h[b][0][:] = cls; mask_s[b][0] = pad[b][0], mask_s[b][1+t] = pad[b][t]  (pad == nullptr: all valid); h is [B][T+1][D]
hipError_t launch_peav_cls_mask(float* h, const float* cls, const unsigned char* pad, unsigned char* mask_s, int B,
                                int T, int D, hipStream_t st);
// dst[b*rep + c][:] = src[b][:]
hipError_t launch_repeat_rows_u8(const unsigned char* src, unsigned char* dst, int rows, int rep, int T,
                                 hipStream_t st);
// dst[b*rep + c][:] = src[b][:], items of `elems` fp32 values (elems % 4 == 0)
hipError_t launch_repeat_items_f32(const float* src, float* dst, int items, int rep, long elems, hipStream_t st);
// masked GroupNorm(1 group) + SiLU into a halo-padded channels-last buffer; partials: B * 64 * 3 doubles
hipError_t launch_masked_groupnorm_silu(const float* x, const float* w, const float* b, const unsigned char* mask,
                                        double* partials, void* out, bool bf16, int B, int S, int C, int halo,
                                        float eps, hipStream_t st);
// out[m,:] = LayerNorm(x[m,:]) * w + b; either output may be null
hipError_t launch_layernorm_rows(const float* x, long x_ld, const float* w, const float* b, float* out_f32,
                                 void* out_act, bool bf16, long M, int D, float eps, hipStream_t st);
// scores[b][j] = (masked mean over frames of hidden[b][1+t][:]) . head_w[j][:] * std[j] + mean[j]
hipError_t launch_judge_pool_head(const float* hidden, const unsigned char* mask_s, const float* head_w,
                                  const float* mean, const float* std_, float* out, int B, int T, int D,
                                  hipStream_t st);
// logits[b][t] = <audio[a_off + b*a_bstride + t*E + :], text[b][:]> * scale[0] + bias[0]
hipError_t launch_frame_logits(const float* audio, long a_bstride, long a_off, const float* text, const float* scale,
                               const float* bias, float* out, int B, int T, int E, hipStream_t st);

// ---- PE-Core vision tower (vit_kernels.hip) ---------------------------------------------------------------------
// im2col of the k = stride = P patch convolution: frames [n,3,S,S] f32 -> rows [n*(S/P)^2, Kp], column c*P*P + py*P + px
hipError_t launch_patchify(const float* frames, void* out, bool bf16, int n, int S, int P, int Kp, hipStream_t st);
// q|k|v rows [n*T, 3*H*hd] -> Q, K [n,H,Tp,hd] (adjacent-pair rotation by rc / rs [T, hd/2]; null = none), V^T [n,H,hd,Tp]
hipError_t launch_rope2d_split(const void* qkv, const float* rc, const float* rs, void* Q, void* K, void* Vt, bool bf16,
                               int n, int T, int Tp, int H, int head_dim, hipStream_t st);
// one query per head over all tokens: q [H*hd] f32, kv rows [n*T, 2*H*hd] = (k | v) -> out [n, H*hd]
hipError_t launch_pool_attention(const float* q, const void* kv, void* out, bool bf16, int n, int T, int H, int head_dim,
                                 hipStream_t st);
hipError_t launch_l2_normalize(float* x, int rows, int D, hipStream_t st);
// out[f, :] = mean over the T tokens of x[f, t, :]
hipError_t launch_token_mean(const float* x, long x_ld, float* out_f32, void* out_act, bool bf16, int n, int T, int D,
                             hipStream_t st);

// ---- T5 prompt encoder (t5_kernels.hip) ------------------------------------------------------------
hipError_t launch_t5_embed(const long long* ids, const float* table, float* out, long M, int D, int vocab, hipStream_t st);
// softmax(q k^T + bias[h][k - q] + key mask) v per (item, head, query row); qkv [B*Lt, 3*H*dkv]; Lt <= 512, dkv <= 128
// (bias nullable; scale multiplies q k^T; window > 0: only keys within +-window tokens)
hipError_t launch_t5_attention(const void* qkv, const unsigned char* mask, const float* bias, void* out, bool bf16, int B,
                               int Lt, int H, int dkv, int max_len, float scale, int window, hipStream_t st);
// ModernBERT text tower (mbert.hip): rotate-half RoPE of the q / k segments of q|k|v rows in place; gated GELU
hipError_t launch_mbert_rope(void* qkv, const float* cs, const float* sn, bool bf16, long M, int Lt, int H, int hd,
                             hipStream_t st);
hipError_t launch_geglu(const void* x, void* out, bool bf16, long M, int F, hipStream_t st);

}  // namespace sa
