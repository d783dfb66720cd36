// CPU emulation of every kernel launcher declared in sam_audio_amd/csrc/kernels.h  --  TEST INFRASTRUCTURE ONLY.
//
// Purpose: the HOST orchestration code of the product (sam_audio_amd/csrc/{engine,peav,api}.hip: which kernel runs
// when, with which pointers, strides, offsets, aliasing of the workspace) is compiled UNCHANGED into
// oracle/_emu/libsamaudio_emu.so and linked against the plain-C++ loops below instead of the HIP kernels, so that the
// orchestration can be checked against the oracle on a machine without a GPU (tests/test_emu_cpu.py).  Each function
// restates the documented contract of the launcher it stands in for (kernels.h / common.h), not its tiling.
// Nothing under sam_audio_amd/ loads this library; only tests do, by explicit path.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../sam_audio_amd/csrc/kernels.h"

namespace sa {

namespace {
inline float bf2f_h(unsigned short h) {
  unsigned u = ((unsigned)h) << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
inline unsigned short f2bf_h(float f) {  // round-to-nearest-even, same as common.h f2bf
  unsigned u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
template <typename T> struct HE;
template <> struct HE<float> {
  static float ld(const float* p) { return *p; }
  static void st(float* p, float v) { *p = v; }
};
template <> struct HE<bf16_t> {
  static float ld(const bf16_t* p) { return bf2f_h(p->v); }
  static void st(bf16_t* p, float v) { p->v = f2bf_h(v); }
};
inline float silu_h(float x) { return x / (1.0f + std::exp(-x)); }
inline float snake_h(float x, float a) {
  const float s = std::sin(a * x);
  return x + s * s / (a + 1e-9f);
}
int g_flags[64] = {0};
int g_force = -1;
unsigned long long g_epoch = 0;
}  // namespace

void debug_touch() { ++g_epoch; }
unsigned long long debug_epoch() { return g_epoch; }

hipError_t launch_poison_lds(hipStream_t) { return hipSuccess; }
void set_debug_flag(int flag, int value) {
  if (flag >= 0 && flag < 64) g_flags[flag] = value;
}
int debug_flag(int flag) { return flag >= 0 && flag < 64 ? g_flags[flag] : 0; }
void gemm_force_variant(int v) { g_force = v; }
int gemm_variant(const GemmParams&, bool) { return 0; }
const char* gemm_variant_name(int v, bool) { return v == 0 ? "emu_gemm" : ""; }
bool gemm2_ok(const GemmParams&) { return false; }
hipError_t launch_gemm2(const GemmParams&, int, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_gemm8(const GemmParams&, int, hipStream_t) { return hipErrorNotSupported; }
int gemm_tail_split(const GemmParams&, bool) { return 0; }
hipError_t launch_gemm_part(const GemmParams&, bool, int, hipStream_t) { return hipErrorNotSupported; }

// same argument checks as the product (gemm.hip gemm_check): the orchestration must satisfy them on the GPU too
const char* gemm_check(const GemmParams& p, bool is_bf16) {
  const int bk = is_bf16 ? 64 : 32, ch = is_bf16 ? 8 : 4;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.nbatch <= 0) return "gemm: empty problem";
  if (p.K % bk) return "gemm: K must be a multiple of 64 (bf16) / 32 (f32) - pad W with zeros";
  if (p.kc % ch || p.kc <= 0) return "gemm: kc must be a positive multiple of the 16-byte chunk";
  if ((p.lda % ch) || (p.tap_stride % ch) || (p.a_off % ch) || (p.a_bstride % ch))
    return "gemm: A strides/offsets must be 16-byte aligned";
  if (p.swiglu && (p.N % 32)) return "gemm: swiglu needs N % 32 == 0";
  if (p.gate && p.rows_per_gate <= 0) return "gemm: rows_per_gate";
  return nullptr;
}

// C[b][m][n] = epilogue(sum_k A(b,m,k) * W[b][n][k])  -  contract in common.h (GemmParams)
template <typename T>
static void gemm_t(const GemmParams& p) {
  const T* A0 = (const T*)p.A;
  const T* W0 = (const T*)p.W;
  for (int b = 0; b < p.nbatch; ++b) {
    const T* W = W0 + (long)b * p.w_bstride;
#pragma omp parallel for schedule(static)
    for (int m = 0; m < p.M; ++m) {
      std::vector<float> arow((size_t)p.K), acc((size_t)p.N);
      const T* Ar = A0 + p.a_off + (long)b * p.a_bstride + (long)m * p.lda;
      for (int k = 0; k < p.K; ++k) arow[k] = HE<T>::ld(Ar + (long)(k / p.kc) * p.tap_stride + (k % p.kc));
      const bool ktm = (p.flags & GEMM_FLAG_W_KTM) != 0;   // W stored [K/64][N][64] (common.h) instead of [N][K]
      for (int n = 0; n < p.N; ++n) {
        const T* w = W + (long)n * p.K;
        float s = 0.f;
        if constexpr (sizeof(T) == 4) {
          if (p.flags & GEMM_FLAG_W_FLY16) {   // split weight (common.h): slab of 32 k = [4 chunks hi | 4 chunks lo], chunk c = k in {4c.., 16+4c..}
            const bf16_t* h = (const bf16_t*)w;
            for (int k = 0; k < p.K; ++k) {
              const int kk = k % 32, pos = (kk % 16 / 4) * 8 + (kk / 16) * 4 + kk % 4;
              const bf16_t* slab = h + (long)(k / 32) * 64;
              s += arow[k] * (HE<bf16_t>::ld(slab + pos) + HE<bf16_t>::ld(slab + 32 + pos));
            }
            acc[n] = s;
            continue;
          }
        }
        if (ktm)
          for (int k = 0; k < p.K; ++k) s += arow[k] * HE<T>::ld(W + ((long)(k / 64) * p.N + n) * 64 + k % 64);
        else
          for (int k = 0; k < p.K; ++k) s += arow[k] * HE<T>::ld(w + k);
        acc[n] = s;
      }
      const int n_out = p.swiglu ? p.N / 2 : p.N;
      const float* grow = p.gate ? p.gate + (((long)b * p.M + m) / p.rows_per_gate) * p.gate_ld : nullptr;
      for (int n = 0; n < n_out; ++n) {
        float v;
        if (p.swiglu) {
          const int blk = n / 16, i = n % 16;
          v = silu_h(acc[blk * 32 + i]) * acc[blk * 32 + 16 + i];
        } else {
          v = acc[n];
        }
        const int ch = p.chan_mod ? n % p.chan_mod : n;
        if (p.bias) v += p.bias[ch];
        if (grow) v *= (p.gate_tab ? p.gate_tab[n] : 0.f) + grow[n];
        v *= p.alpha;
        const long erel = (long)m * p.c_ld_rel + n;
        if (p.c_ld_rel && (erel < p.c_lo || erel >= p.c_hi)) continue;
        if (p.res) v += p.res[p.res_off + (long)b * p.res_bstride + (long)m * p.res_ld + n];
        float a = v;
        if (p.act == ACT_SNAKE) a = snake_h(v, p.act_alpha[ch]);
        else if (p.act == ACT_TANH) a = std::tanh(v);
        else if (p.act == ACT_SILU) a = silu_h(v);
        if (p.out_f32) p.out_f32[p.f32_off + (long)b * p.f32_bstride + (long)m * p.f32_ld + n] = p.f32_act ? a : v;
        if (p.out_act && (p.flags & GEMM_FLAG_OUT_SPLIT3)) {   // [lo | hi | hi] rows of 3 n_out elements (common.h)
          T* row = (T*)p.out_act + p.act_off + (long)b * p.act_bstride + (long)m * p.act_ld;
          T hi;
          HE<T>::st(&hi, a);
          HE<T>::st(row + n, a - HE<T>::ld(&hi));
          row[n_out + n] = hi;
          row[2 * n_out + n] = hi;
        } else
        if (p.out_act) HE<T>::st((T*)p.out_act + p.act_off + (long)b * p.act_bstride + (long)m * p.act_ld + n, a);
      }
    }
  }
}

hipError_t launch_gemm(const GemmParams& p, bool is_bf16, hipStream_t) {
  // in-place residual updates (res == out_f32) are element-wise safe: each output element is read once, then written
  if (is_bf16) gemm_t<bf16_t>(p);
  else gemm_t<float>(p);
  return hipSuccess;
}

template <typename TO>
static void rmsnorm_mod_t(const float* x, const float* w, const float* shift_tab, const float* scale_tab,
                          const float* tvec, long tvec_ld, int shift_off, int scale_off, TO* out, int M, int D,
                          int rows_per_b, float eps) {
  for (int row = 0; row < M; ++row) {
    const float* xr = x + (long)row * D;
    float ss = 0.f;
    for (int i = 0; i < D; ++i) ss += xr[i] * xr[i];
    const float inv = 1.0f / std::sqrt(ss / (float)D + eps);
    const float* trow = tvec ? tvec + (long)(row / rows_per_b) * tvec_ld : nullptr;
    for (int i = 0; i < D; ++i) {
      float o = xr[i] * inv * w[i];
      if (trow) o = o * (1.f + (scale_tab[i] + trow[scale_off + i])) + (shift_tab[i] + trow[shift_off + i]);
      HE<TO>::st(out + (long)row * D + i, o);
    }
  }
}
hipError_t launch_mod_tables(const ModTables& t, int n_norms, const float* tvec, long tvec_ld, int nt, float* gs, int D,
                             hipStream_t) {
  for (int n = 0; n < n_norms; ++n)
    for (int tt = 0; tt < nt; ++tt) {
      const float* trow = tvec + (long)tt * tvec_ld;
      float* g = gs + (((long)n * nt + tt) * 2) * D;
      float* s = g + D;
      for (int i = 0; i < D; ++i) {
        const float sc = t.scale_tab[n][i] + trow[t.scale_off[n] + i];
        g[i] = t.w[n][i] * (1.f + sc);
        s[i] = t.shift_tab[n][i] + trow[t.shift_off[n] + i];
      }
    }
  return hipSuccess;
}
hipError_t launch_rmsnorm_gs(const float* x, const float* gs, long gs_ld, void* out, bool bf16, int M, int D, int rows_per_b,
                             float eps, hipStream_t, bool);
hipError_t launch_rmsnorm_gs_split3(const float* x, const float* gs, long gs_ld, void* out, int M, int D, int rows_per_b, float eps,
                                    hipStream_t st) {
  std::vector<float> tmp((size_t)M * D);
  launch_rmsnorm_gs(x, gs, gs_ld, tmp.data(), false, M, D, rows_per_b, eps, st, false);
  return launch_split3(tmp.data(), D, out, M, D, st);
}
hipError_t launch_rmsnorm_gs(const float* x, const float* gs, long gs_ld, void* out, bool bf16, int M, int D, int rows_per_b,
                             float eps, hipStream_t, bool) {
  for (int row = 0; row < M; ++row) {
    const float* xr = x + (long)row * D;
    double ss = 0;
    for (int i = 0; i < D; ++i) ss += (double)xr[i] * xr[i];
    const float inv = 1.f / std::sqrt((float)(ss / D) + eps);
    const float* g = gs + (long)(row / rows_per_b) * gs_ld;
    const float* s = g + D;
    for (int i = 0; i < D; ++i) {
      const float o = xr[i] * inv * g[i] + s[i];
      if (bf16) HE<bf16_t>::st((bf16_t*)out + (long)row * D + i, o);
      else ((float*)out)[(long)row * D + i] = o;
    }
  }
  return hipSuccess;
}
hipError_t launch_rmsnorm_mod(const float* x, const float* w, const float* shift_tab, const float* scale_tab,
                              const float* tvec, long tvec_ld, int shift_off, int scale_off, void* out, bool bf16,
                              int M, int D, int rows_per_b, float eps, hipStream_t) {
  if (bf16) rmsnorm_mod_t<bf16_t>(x, w, shift_tab, scale_tab, tvec, tvec_ld, shift_off, scale_off, (bf16_t*)out, M, D, rows_per_b, eps);
  else rmsnorm_mod_t<float>(x, w, shift_tab, scale_tab, tvec, tvec_ld, shift_off, scale_off, (float*)out, M, D, rows_per_b, eps);
  return hipSuccess;
}

hipError_t launch_layernorm_accum(const float* x, const float* w, const float* b, const float* gate, float* acc, int M,
                                  int D, float eps, hipStream_t) {
  const float g = std::tanh(gate[0]);
  for (int r = 0; r < M; ++r) {
    const float* xr = x + (long)r * D;
    double s = 0, q = 0;
    for (int i = 0; i < D; ++i) s += xr[i];
    const float mean = (float)(s / D);
    for (int i = 0; i < D; ++i) q += (double)(xr[i] - mean) * (xr[i] - mean);
    const float rstd = 1.0f / std::sqrt((float)(q / D) + eps);
    for (int i = 0; i < D; ++i) acc[(long)r * D + i] += g * ((xr[i] - mean) * rstd * w[i] + b[i]);
  }
  return hipSuccess;
}

template <typename TO>
static void gn_t(const float* x, const float* w, const float* b, const unsigned char* mask, TO* out, int B, int S, int C,
                 int halo, float eps) {
  for (int bi = 0; bi < B; ++bi) {
    const float* xs = x + (long)bi * S * C;
    double s = 0, q = 0, rows = 0;
    for (int t = 0; t < S; ++t) {
      if (mask && !mask[(long)bi * S + t]) continue;
      rows += 1;
      for (int c = 0; c < C; ++c) { s += xs[(long)t * C + c]; q += (double)xs[(long)t * C + c] * xs[(long)t * C + c]; }
    }
    double n = rows * C;
    if (n < 1) n = 1;
    const double mean_d = s / n;
    double var_d = q / n - mean_d * mean_d;
    if (var_d < 0) var_d = 0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / std::sqrt(var_d + (double)eps));
    TO* ob = out + ((long)bi * (S + 2 * halo) + halo) * C;
    for (int t = 0; t < S; ++t)
      for (int c = 0; c < C; ++c) {
        const bool on = !mask || mask[(long)bi * S + t];
        HE<TO>::st(ob + (long)t * C + c, on ? silu_h((xs[(long)t * C + c] - mean) * rstd * w[c] + b[c]) : 0.f);
      }
  }
}
hipError_t launch_groupnorm_silu(const float* x, const float* w, const float* b, double*, void* out, bool bf16, int B,
                                 int T, int C, int halo, float eps, hipStream_t) {
  if (bf16) gn_t<bf16_t>(x, w, b, nullptr, (bf16_t*)out, B, T, C, halo, eps);
  else gn_t<float>(x, w, b, nullptr, (float*)out, B, T, C, halo, eps);
  return hipSuccess;
}
hipError_t launch_masked_groupnorm_silu(const float* x, const float* w, const float* b, const unsigned char* mask,
                                        double*, void* out, bool bf16, int B, int S, int C, int halo, float eps,
                                        hipStream_t) {
  if (bf16) gn_t<bf16_t>(x, w, b, mask, (bf16_t*)out, B, S, C, halo, eps);
  else gn_t<float>(x, w, b, mask, (float*)out, B, S, C, halo, eps);
  return hipSuccess;
}

template <typename TA>
static void qkv_prep_t(const TA* qkv, const float* qw, const float* kw, const float* rc, const float* rs, TA* Q, TA* K,
                       TA* Vt, int B, int T, int Tp, int H, float eps) {
  const int D = H * 128;
  const long ld = 3L * D;
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h) {
      const long bh = (long)b * H + h;
      for (int t = 0; t < Tp; ++t) {
        float q[128], k[128], v[128];
        for (int d = 0; d < 128; ++d) q[d] = k[d] = v[d] = 0.f;
        if (t < T) {
          const TA* row = qkv + ((long)b * T + t) * ld + h * 128;
          float sq = 0, sk = 0;
          for (int d = 0; d < 128; ++d) {
            q[d] = HE<TA>::ld(row + d); k[d] = HE<TA>::ld(row + D + d); v[d] = HE<TA>::ld(row + 2 * D + d);
            sq += q[d] * q[d]; sk += k[d] * k[d];
          }
          const float iq = 1.0f / std::sqrt(sq / 128.f + eps), ik = 1.0f / std::sqrt(sk / 128.f + eps);
          for (int i = 0; i < 64; ++i) {
            const float c = rc[(long)t * 64 + i], s = rs[(long)t * 64 + i];
            const float a0 = q[2 * i] * iq * qw[2 * i], a1 = q[2 * i + 1] * iq * qw[2 * i + 1];
            q[2 * i] = a0 * c - a1 * s; q[2 * i + 1] = a0 * s + a1 * c;
            const float b0 = k[2 * i] * ik * kw[2 * i], b1 = k[2 * i + 1] * ik * kw[2 * i + 1];
            k[2 * i] = b0 * c - b1 * s; k[2 * i + 1] = b0 * s + b1 * c;
          }
        }
        for (int d = 0; d < 128; ++d) {
          HE<TA>::st(Q + (bh * Tp + t) * 128 + d, q[d]);
          HE<TA>::st(K + (bh * Tp + t) * 128 + d, k[d]);
          HE<TA>::st(Vt + (bh * 128 + d) * Tp + t, v[d]);
        }
      }
    }
}
hipError_t launch_qkv_prep(const void* qkv, const float* qw, const float* kw, const float* rope_cos,
                           const float* rope_sin, void* Q, void* K, void* Vt, bool bf16, int B, int T, int Tp, int H,
                           float eps, hipStream_t, int head_dim) {
  if (Tp % 64 || Tp < T || head_dim != 128) return hipErrorInvalidValue;   // the launcher emulation covers head_dim 128 only
  if (bf16) qkv_prep_t<bf16_t>((const bf16_t*)qkv, qw, kw, rope_cos, rope_sin, (bf16_t*)Q, (bf16_t*)K, (bf16_t*)Vt, B, T, Tp, H, eps);
  else qkv_prep_t<float>((const float*)qkv, qw, kw, rope_cos, rope_sin, (float*)Q, (float*)K, (float*)Vt, B, T, Tp, H, eps);
  return hipSuccess;
}

template <typename TA>
static void self_attn_t(const TA* Q, const TA* K, const TA* Vt, const unsigned char* key_mask, TA* out, int B, int T,
                        int Tp, int H) {
  const float scale = 0.08838834764831845f;
  std::vector<float> sc((size_t)T);
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h) {
      const long bh = (long)b * H + h;
      for (int t = 0; t < T; ++t) {
        float mx = -INFINITY;
        for (int j = 0; j < T; ++j) {
          if (!key_mask[(long)b * T + j]) { sc[j] = -INFINITY; continue; }
          float s = 0.f;
          for (int d = 0; d < 128; ++d) s += HE<TA>::ld(Q + (bh * Tp + t) * 128 + d) * HE<TA>::ld(K + (bh * Tp + j) * 128 + d);
          sc[j] = s * scale;
          mx = std::fmax(mx, sc[j]);
        }
        float l = 0.f;
        for (int j = 0; j < T; ++j) { sc[j] = sc[j] == -INFINITY ? 0.f : std::exp(sc[j] - mx); l += sc[j]; }
        for (int d = 0; d < 128; ++d) {
          float o = 0.f;
          for (int j = 0; j < T; ++j) o += sc[j] * HE<TA>::ld(Vt + (bh * 128 + d) * Tp + j);
          HE<TA>::st(out + ((long)b * T + t) * (H * 128) + h * 128 + d, o / l);
        }
      }
    }
}
hipError_t launch_split3(const float* x, long ldx, void* out, long M, int K, hipStream_t);
hipError_t launch_self_attention_x3(const float* Q, const float* K, const float* Vt, const unsigned char* key_mask, float* out, int B,
                                    int T, int Tp, int H, int head_dim, hipStream_t st, void* out3) {   // fp32 tensors: the emulation is the fp32 product
  if (head_dim != 128) return hipErrorInvalidValue;
  if (!out3) {
    self_attn_t<float>(Q, K, Vt, key_mask, out, B, T, Tp, H);
    return hipSuccess;
  }
  std::vector<float> tmp((size_t)B * T * H * 128);
  self_attn_t<float>(Q, K, Vt, key_mask, tmp.data(), B, T, Tp, H);
  return launch_split3(tmp.data(), (long)H * 128, out3, (long)B * T, H * 128, st);
}
hipError_t launch_self_attention_hd(const void* Q, const void* K, const void* Vt, const unsigned char* key_mask, void* out,
                                    bool bf16, int B, int T, int Tp, int H, int head_dim, hipStream_t st, bool alt) {
  if (head_dim != 128) return hipErrorInvalidValue;
  return launch_self_attention(Q, K, Vt, key_mask, out, bf16, B, T, Tp, H, st, alt);
}
hipError_t launch_self_attention(const void* Q, const void* K, const void* Vt, const unsigned char* key_mask, void* out,
                                 bool bf16, int B, int T, int Tp, int H, hipStream_t, bool) {
  if (bf16) self_attn_t<bf16_t>((const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt, key_mask, (bf16_t*)out, B, T, Tp, H);
  else self_attn_t<float>((const float*)Q, (const float*)K, (const float*)Vt, key_mask, (float*)out, B, T, Tp, H);
  return hipSuccess;
}

template <typename TA>
static void headnorm_t(TA* x, const float* w, long rows, long ld, int col0, int H, float eps) {
  for (long r = 0; r < rows; ++r)
    for (int h = 0; h < H; ++h) {
      TA* p = x + r * ld + col0 + h * 128;
      float ss = 0.f;
      for (int d = 0; d < 128; ++d) ss += HE<TA>::ld(p + d) * HE<TA>::ld(p + d);
      const float inv = 1.0f / std::sqrt(ss / 128.f + eps);
      for (int d = 0; d < 128; ++d) HE<TA>::st(p + d, HE<TA>::ld(p + d) * inv * w[d]);
    }
}
hipError_t launch_headnorm(void* x, const float* w, bool bf16, int rows, long ld, int col0, int H, float eps,
                           hipStream_t, int head_dim) {
  if (head_dim != 128) return hipErrorInvalidValue;
  if (bf16) headnorm_t<bf16_t>((bf16_t*)x, w, rows, ld, col0, H, eps);
  else headnorm_t<float>((float*)x, w, rows, ld, col0, H, eps);
  return hipSuccess;
}
hipError_t launch_headnorm_layers(void* kv_all, const float* w_all, bool bf16, int rows, int L, int H, float eps,
                                  hipStream_t st, int head_dim) {
  if (head_dim != 128) return hipErrorInvalidValue;
  const long D2 = 2L * H * 128;
  for (int l = 0; l < L; ++l) {
    hipError_t e = launch_headnorm(kv_all, w_all + l * 128, bf16, rows, D2 * L, (int)(l * D2), H, eps, st, 128);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <typename TA>
static void cross_attn_t(const TA* q, const float* qw, const TA* kv, long kv_ld, const unsigned char* mask, TA* out,
                         long M, int T, int Lt, int H, float eps) {
  const int D = H * 128;
  const float scale = 0.08838834764831845f;
  std::vector<float> sc((size_t)Lt);
  for (long m = 0; m < M; ++m)
    for (int h = 0; h < H; ++h) {
      const long b = m / T;
      float qn[128], ss = 0.f;
      for (int d = 0; d < 128; ++d) { qn[d] = HE<TA>::ld(q + m * D + h * 128 + d); ss += qn[d] * qn[d]; }
      const float inv = 1.0f / std::sqrt(ss / 128.f + eps);
      for (int d = 0; d < 128; ++d) qn[d] *= inv * qw[d];
      float mx = -INFINITY;
      for (int j = 0; j < Lt; ++j) {
        if (!mask[b * Lt + j]) { sc[j] = -INFINITY; continue; }
        float s = 0.f;
        for (int d = 0; d < 128; ++d) s += qn[d] * HE<TA>::ld(kv + (b * Lt + j) * kv_ld + h * 128 + d);
        sc[j] = s * scale;
        mx = std::fmax(mx, sc[j]);
      }
      float l = 0.f;
      for (int j = 0; j < Lt; ++j) { sc[j] = sc[j] == -INFINITY ? 0.f : std::exp(sc[j] - mx); l += sc[j]; }
      for (int d = 0; d < 128; ++d) {
        float o = 0.f;
        for (int j = 0; j < Lt; ++j) o += sc[j] * HE<TA>::ld(kv + (b * Lt + j) * kv_ld + D + h * 128 + d);
        HE<TA>::st(out + m * D + h * 128 + d, o / l);
      }
    }
}
hipError_t launch_cross_attention(const void* q, const float* qw, const void* kv, long kv_ld, const unsigned char* mask,
                                  void* out, bool bf16, int B, int T, int Lt, int H, float eps, hipStream_t, int head_dim) {
  if (head_dim != 128) return hipErrorInvalidValue;
  if (bf16) cross_attn_t<bf16_t>((const bf16_t*)q, qw, (const bf16_t*)kv, kv_ld, mask, (bf16_t*)out, (long)B * T, T, Lt, H, eps);
  else cross_attn_t<float>((const float*)q, qw, (const float*)kv, kv_ld, mask, (float*)out, (long)B * T, T, Lt, H, eps);
  return hipSuccess;
}
// the folded cross-attention projection is a bf16 fast path of the product; the emulation runs the
// unfolded order (tests set SAMAUDIO_NO_FOLD=1)
hipError_t launch_cross_attn_probs(const void*, const float*, const void*, long, const unsigned char*, void*, int, int,
                                   int, int, int, int, float, hipStream_t) { return hipErrorNotSupported; }
// the fused residual-unit kernel is not emulated: the engine then issues the two launches
bool resunit_ok(const GemmParams&, const GemmParams&) { return false; }
hipError_t launch_resunit(const GemmParams&, const GemmParams&, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_cross_attn_fold_layers(const void* const*, int, const void*, long, void*, int, int, int, int, int, hipStream_t) {
  return hipErrorNotSupported;   // (the emulation runs with SAMAUDIO_NO_FOLD=1)
}
hipError_t launch_cross_attn_fold(const void*, const void*, long, void*, int, int, int, int, int, hipStream_t) {
  return hipErrorNotSupported;
}

hipError_t launch_sentinel(const void*, int, long, int, long, float*, float*, hipStream_t) { return hipSuccess; }   // not emulated
hipError_t launch_set_floats(float* dst, const float* host_values, int n, hipStream_t) {
  for (int i = 0; i < n; ++i) dst[i] = host_values[i];
  return hipSuccess;
}
hipError_t launch_time_features(const float* t, int nt, const float* freqs, int fdim, const float* inv_freq, int D,
                                void* temb, float* tsin, bool bf16, hipStream_t) {
  const int hf = fdim / 2, hd = D / 2;
  for (int j = 0; j < nt; ++j) {
    for (int i = 0; i < hf; ++i) {
      const float a = t[j] * freqs[i];
      if (bf16) { HE<bf16_t>::st((bf16_t*)temb + (long)j * fdim + i, std::cos(a)); HE<bf16_t>::st((bf16_t*)temb + (long)j * fdim + hf + i, std::sin(a)); }
      else { ((float*)temb)[(long)j * fdim + i] = std::cos(a); ((float*)temb)[(long)j * fdim + hf + i] = std::sin(a); }
    }
    for (int i = 0; i < hd; ++i) {
      const float a = t[j] * inv_freq[i];
      tsin[(long)j * D + i] = std::cos(a);
      tsin[(long)j * D + hd + i] = std::sin(a);
    }
  }
  return hipSuccess;
}

hipError_t launch_add_rowvec(const float* x, const float* vec, long vec_ld, void* out, bool bf16, int rows, int D,
                             int rows_per_b, hipStream_t) {
  for (long r = 0; r < rows; ++r)
    for (int c = 0; c < D; ++c) {
      const float v = x[r * D + c] + vec[(r / rows_per_b) * vec_ld + c];
      if (bf16) HE<bf16_t>::st((bf16_t*)out + r * D + c, v);
      else ((float*)out)[r * D + c] = v;
    }
  return hipSuccess;
}

hipError_t launch_anchor_gather(const float* emb, const long* ids, int n_ids, const long* align, void* out, bool bf16,
                                int B, int T, int E, int vocab, hipStream_t) {
  for (long r = 0; r < (long)B * T; ++r) {
    long slot = align[r];
    slot = slot < 0 ? 0 : (slot >= n_ids ? n_ids - 1 : slot);
    long tok = ids[(r / T) * n_ids + slot];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    for (int i = 0; i < E; ++i) {
      if (bf16) HE<bf16_t>::st((bf16_t*)out + r * E + i, emb[tok * E + i]);
      else ((float*)out)[r * E + i] = emb[tok * E + i];
    }
  }
  return hipSuccess;
}

hipError_t launch_hash_items(const unsigned* x, size_t words, int items, unsigned long long* out, hipStream_t) {
  for (int b = 0; b < items; ++b) {
    unsigned long long h = 0;
    for (size_t i = 0; i < words; ++i) h += (unsigned long long)(x[(size_t)b * words + i] ^ (unsigned)(i * 0x9E3779B1u)) * (2 * i + 1);
    out[b] = h;
  }
  return hipSuccess;
}
void* debug_device_alloc(size_t bytes) { return std::malloc(bytes); }
void debug_device_free(void* p) { std::free(p); }

hipError_t launch_to_act(const float* in, long in_bstride, long in_ld, int in_col0, void* out, long out_bstride,
                         bool bf16, int B, long T, int C_in, int C_out, int halo, hipStream_t) {
  if (out_bstride == 0) out_bstride = (T + 2L * halo) * C_out;
  for (int b = 0; b < B; ++b)
    for (long t = 0; t < T; ++t)
      for (int c = 0; c < C_out; ++c) {
        const float v = c < C_in ? in[(long)b * in_bstride + in_col0 + t * in_ld + c] : 0.f;
        const long o = (long)b * out_bstride + (long)halo * C_out + t * C_out + c;
        if (bf16) HE<bf16_t>::st((bf16_t*)out + o, v);
        else ((float*)out)[o] = v;
      }
  return hipSuccess;
}

hipError_t launch_split3(const float* x, long ldx, void* out, long M, int K, hipStream_t) {   // [lo | hi | hi], kernels.hip split3_kernel
  bf16_t* o = (bf16_t*)out;
  for (long m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) {
      const float v = x[m * ldx + k];
      bf16_t hi;
      HE<bf16_t>::st(&hi, v);
      HE<bf16_t>::st(o + m * 3L * K + k, v - HE<bf16_t>::ld(&hi));
      o[m * 3L * K + K + k] = hi;
      o[m * 3L * K + 2L * K + k] = hi;
    }
  return hipSuccess;
}

hipError_t launch_act_flat(const float* in, long in_bstride, float* out, long out_bstride, int items, long count, int chan, int act,
                           const float* alpha, hipStream_t) {
  for (int b = 0; b < items; ++b)
    for (long i = 0; i < count; ++i) {
      const float v = in[(long)b * in_bstride + i];
      out[(long)b * out_bstride + i] = act == ACT_SNAKE ? snake_h(v, alpha[i % chan]) : act == ACT_TANH ? std::tanh(v) : act == ACT_SILU ? silu_h(v) : v;
    }
  return hipSuccess;
}

hipError_t launch_zero_halo(void* buf, bool bf16, int B, long T, int C, int halo, hipStream_t) {
  const size_t e = bf16 ? 2 : 4;
  for (int b = 0; b < B; ++b) {
    char* base = (char*)buf + (size_t)b * (T + 2L * halo) * C * e;
    std::memset(base, 0, (size_t)halo * C * e);
    std::memset(base + (size_t)(T + halo) * C * e, 0, (size_t)halo * C * e);
  }
  return hipSuccess;
}

// ---- PE-AV / Judge / PE-A-Frame (peav_kernels.hip) ----------------------------------------------------------
hipError_t launch_peav_cls_mask(float* h, const float* cls, const unsigned char* pad, unsigned char* mask_s, int B,
                                int T, int D, hipStream_t) {
  const int S = T + 1;
  for (int b = 0; b < B; ++b) {
    std::memcpy(h + (long)b * S * D, cls, (size_t)D * 4);
    for (int s = 0; s < S; ++s) mask_s[(long)b * S + s] = pad ? (pad[(long)b * T + (s == 0 ? 0 : s - 1)] ? 1 : 0) : 1;
  }
  return hipSuccess;
}

hipError_t launch_cross_attn_probs3(const float*, const float*, const float*, long, const unsigned char*, void*, int, int, int, int, int,
                                    int, float, hipStream_t) {
  return hipErrorNotSupported;   // (the launcher emulation runs with SAMAUDIO_NO_FOLD: the folded kernels are not emulated)
}
hipError_t launch_cross_attn_fold3_layers(const float* const*, int, const float*, long, void*, int, int, int, int, int, hipStream_t) {
  return hipErrorNotSupported;
}
hipError_t launch_qkv_prep(const void* qkv, const float* qw, const float* kw, const float* rope_cos, const float* rope_sin, void* Q, void* K,
                           void* Vt, bool bf16, int B, int T, int Tp, int H, float eps, hipStream_t st, int head_dim);
hipError_t launch_qkv_prep_f32x(const float* qkv, const float* qw, const float* kw, const float* rope_cos, const float* rope_sin, float* Q,
                                float* K, float* Vt, int B, int T, int Tp, int H, float eps, hipStream_t st) {
  return launch_qkv_prep(qkv, qw, kw, rope_cos, rope_sin, Q, K, Vt, false, B, T, Tp, H, eps, st, 128);
}
hipError_t launch_repeat_items_f32(const float* src, float* dst, int items, int rep, long elems, hipStream_t) {
  for (long r = (long)items * rep - 1; r >= 0; --r) std::memmove(dst + r * elems, src + (r / rep) * elems, (size_t)elems * 4);
  return hipSuccess;
}
hipError_t launch_repeat_rows_u8(const unsigned char* src, unsigned char* dst, int rows, int rep, int T, hipStream_t) {
  for (long r = 0; r < (long)rows * rep; ++r) std::memcpy(dst + r * T, src + (r / rep) * T, (size_t)T);
  return hipSuccess;
}

hipError_t launch_layernorm_rows(const float* x, long x_ld, const float* w, const float* b, float* out_f32,
                                 void* out_act, bool bf16, long M, int D, float eps, hipStream_t) {
  for (long r = 0; r < M; ++r) {
    const float* xr = x + r * x_ld;
    double s = 0, q = 0;
    for (int i = 0; i < D; ++i) s += xr[i];
    const float mean = (float)(s / D);
    for (int i = 0; i < D; ++i) q += (double)(xr[i] - mean) * (xr[i] - mean);
    const float rstd = 1.0f / std::sqrt((float)(q / D) + eps);
    for (int i = 0; i < D; ++i) {
      const float o = (xr[i] - mean) * rstd * w[i] + b[i];
      if (out_f32) out_f32[r * D + i] = o;
      if (out_act) {
        if (bf16) HE<bf16_t>::st((bf16_t*)out_act + r * D + i, o);
        else ((float*)out_act)[r * D + i] = o;
      }
    }
  }
  return hipSuccess;
}

hipError_t launch_judge_pool_head(const float* hidden, const unsigned char* mask_s, const float* head_w,
                                  const float* mean, const float* std_, float* out, int B, int T, int D, hipStream_t) {
  if (D > 4096) return hipErrorInvalidValue;
  const int S = T + 1;
  std::vector<float> pooled((size_t)D);
  for (int b = 0; b < B; ++b) {
    int valid = 0;
    for (int t = 1; t < S; ++t) valid += mask_s[(long)b * S + t] ? 1 : 0;
    const float inv = 1.f / (float)(valid > 0 ? valid : 1);
    for (int d = 0; d < D; ++d) {
      float a = 0.f;
      for (int t = 1; t < S; ++t)
        if (mask_s[(long)b * S + t]) a += hidden[((long)b * S + t) * D + d];
      pooled[d] = a * inv;
    }
    for (int j = 0; j < 4; ++j) {
      float v = 0.f;
      for (int d = 0; d < D; ++d) v += pooled[d] * head_w[(long)j * D + d];
      out[(long)b * 4 + j] = v * std_[j] + mean[j];
    }
  }
  return hipSuccess;
}

hipError_t launch_frame_logits(const float* audio, long a_bstride, long a_off, const float* text, const float* scale,
                               const float* bias, float* out, int B, int T, int E, hipStream_t) {
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t) {
      float a = 0.f;
      for (int e = 0; e < E; ++e) a += audio[a_off + (long)b * a_bstride + (long)t * E + e] * text[(long)b * E + e];
      out[(long)b * T + t] = a * scale[0] + bias[0];
    }
  return hipSuccess;
}

}  // namespace sa
