// Shared device/host helpers for libsamaudio_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sa {

// ---- activation element types -------------------------------------------------------------
// The 16-bit GEMM-operand type is stored as raw 16-bit words; all arithmetic is fp32.  Default build: bfloat16
// (bit-compatible with torch.bfloat16).  -DSA_OPERAND_FP16 builds libsamaudio_hip_f16.so, in which the SAME type and the
// same kernels carry IEEE fp16 (torch.float16): identical MFMA rate on gfx950 (v_mfma_*_f16 vs *_bf16), 10 instead of 7
// mantissa bits per operand rounding - the precision="fp16" mode of the host classes.  Only the two
// conversions, the packed-word unpackers and the MFMA builtins below depend on the format.
struct bf16_t {
  unsigned short v;
};

#ifdef SA_OPERAND_FP16
__device__ __forceinline__ float bf2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ unsigned short f2bf(float f) {  // v_cvt_f16_f32: round-to-nearest-even, overflow -> inf
  return __builtin_bit_cast(unsigned short, (_Float16)f);
}
// the two halves of a packed 32-bit word
__device__ __forceinline__ float h16_lo(unsigned w) { return bf2f((unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float h16_hi(unsigned w) { return bf2f((unsigned short)(w >> 16)); }
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8_t;
#define SA_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define SA_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
#ifdef SA_BF16_HW_ROUND
// The hardware conversion (v_cvt_pk_bf16_f32): the same bits as the written-out form below for every finite value and for infinities;
// NaN stays NaN (written out, a NaN with a large payload - the 0xFF poison bytes of the tests - wraps around to a zero).  Built with
// SAMAUDIO_BF16_HW_ROUND=1 (csrc/build.sh, oracle/simt/build.sh); NOT the shipped build yet: it changes the instruction sequence of
// every bf16 kernel that rounds, and round 4 ended before a GPU run of the whole suite on it (it removes the SDWA
// word-select instructions behind packed-fp32 results from the five kernels that still have one or two such pairs).
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
#else
__device__ __forceinline__ unsigned short f2bf(float f) {  // round-to-nearest-even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
#endif
__device__ __forceinline__ float h16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
typedef __attribute__((ext_vector_type(8))) __bf16 h16x8_t;
#define SA_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define SA_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

// two fp32 values -> one packed word of the 16-bit operand format, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 /
// v_cvt_f16_f32 x 2): the same bits as f2bf() for every finite value
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#ifdef SA_OPERAND_FP16
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2_t;
#else
typedef __attribute__((ext_vector_type(2))) __bf16 h16x2_t;
#endif
__device__ __forceinline__ unsigned pack_h16x2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2_t));
}

// the largest finite value of the library's 16-bit operand format
#ifdef SA_OPERAND_FP16
constexpr float kH16Max = 65504.f;
#else
constexpr float kH16Max = 3.3895313892515355e38f;
#endif

// ---- the OTHER 16-bit format (mixed mode) ------------------------------------------------------------------------------
// precision = "mixed" runs on the fp16 build of the library with the five big GEMM classes of the DiT (qkv, wo, c_wq, w13, w2:
// 96 % of the flops, 2e-4 of the error each in bf16 - DESIGN.md section 4) on bfloat16 operands, i.e. BASELINE's dtype where
// the time is and fp16 where the error is.  "alt" = the format that is not the library's own: bf16 in the fp16 build.  In the
// bf16 build alt is bf16 as well (the mode is the plain bf16 mode there).
struct alt16_t {
  unsigned short v;
};
typedef __attribute__((ext_vector_type(2))) __bf16 alt16x2_t;
typedef __attribute__((ext_vector_type(8))) __bf16 alt16x8_t;
__device__ __forceinline__ unsigned pack_alt16x2(float a, float b) {   // v_cvt_pk_bf16_f32
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, alt16x2_t));
}
__device__ __forceinline__ unsigned short f2alt(float f) { return (unsigned short)(pack_alt16x2(f, 0.f) & 0xffffu); }
#define SA_MFMA_16x16x32_ALT(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(alt16x8_t, a), __builtin_bit_cast(alt16x8_t, b), c, 0, 0, 0)

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf2f(p->v); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { p->v = f2bf(v); }
};

template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
  *(float4*)p = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  ushort4 v;
  v.x = f2bf(a); v.y = f2bf(b); v.z = f2bf(c); v.w = f2bf(d);
  *(ushort4*)p = v;
}
template <> __device__ __forceinline__ void store4<alt16_t>(alt16_t* p, float a, float b, float c, float d) {
  *(uint2*)p = make_uint2(pack_alt16x2(a, b), pack_alt16x2(c, d));
}
template <typename T> __device__ __forceinline__ void load2(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void load2<float>(const float* p, float& a, float& b) {
  float2 v = *(const float2*)p; a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void load2<bf16_t>(const bf16_t* p, float& a, float& b) {
  ushort2 v = *(const ushort2*)p; a = bf2f(v.x); b = bf2f(v.y);
}
template <typename T> __device__ __forceinline__ void store2(T* p, float a, float b);
template <> __device__ __forceinline__ void store2<float>(float* p, float a, float b) { *(float2*)p = make_float2(a, b); }
// (the pair goes through the hardware conversion, not through f2bf: the same bits for every finite value, and no SDWA word-select
// instructions behind the arithmetic that produced a / b - DESIGN.md section 8, round 4)
template <> __device__ __forceinline__ void store2<bf16_t>(bf16_t* p, float a, float b) { *(unsigned*)p = pack_h16x2(a, b); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// nn.GELU() (erf form) and CLIP's x * sigmoid(1.702 x): the vision tower's MLP activations
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// T5's feed-forward activations: ReLU (t5-base) and the tanh-form "gelu_new" (non-gated T5 variants)
__device__ __forceinline__ float relu_f(float x) { return fmaxf(x, 0.f); }
__device__ __forceinline__ float gelu_tanh_f(float x) {
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float snake_f(float x, float a) {
  float s = sinf(a * x);
  return x + s * s / (a + 1e-9f);
}
// Snake in the epilogues of the 16-bit GEMM / convolution kernels, whose result is rounded to a 16-bit operand (or is the
// fp32 copy of one): hardware sine and hardware reciprocal (v_sin_f32 / v_rcp_f32, 1 ulp each).  An IEEE division costs
// ~10 VALU instructions per element, and the DAC residual-unit kernels were VALU-bound on it (two Snakes per element).
// Every 16-bit kernel uses this one expression, so fused and unfused forms of a convolution stay bitwise identical.
__device__ __forceinline__ float snake16_f(float x, float a) {
  const float s = __sinf(a * x);
  return x + s * s * __builtin_amdgcn_rcpf(a + 1e-9f);
}

// Reductions over the 16 lanes of one DPP row (lanes 16k .. 16k+15) on the VALU: v_max / v_add with a row_ror
// DPP modifier, no LDS round trip (__shfl_xor lowers to ds_bpermute_b32: ~100 cycles of lgkmcnt wait per step, which
// PMC showed as half of the attention kernel's wave cycles).  Every lane ends up with the reduction of its row; for
// the sum the association order differs from lane to lane by design (each lane's result is still deterministic).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0x128>(v));  // row_ror:8
  v = fmaxf(v, dpp_mov<0x124>(v));  // row_ror:4
  v = fmaxf(v, dpp_mov<0x122>(v));  // row_ror:2
  v = fmaxf(v, dpp_mov<0x121>(v));  // row_ror:1
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0x128>(v);
  v += dpp_mov<0x124>(v);
  v += dpp_mov<0x122>(v);
  v += dpp_mov<0x121>(v);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- epilogue activation codes --------------------------------------------------------------
enum : int { ACT_NONE = 0, ACT_SNAKE = 1, ACT_TANH = 2, ACT_SILU = 3, ACT_GELU = 4, ACT_QUICK_GELU = 5, ACT_RELU = 6,
              ACT_GELU_TANH = 7 };

// ---- generalised (implicit-convolution) GEMM ---------------------------------------------------
// C[b][m][n] = sum_k A(b, m, k) * W[n][k]
//   A(b,m,k): k is split into `ntaps` segments of `kc` contiguous elements;
//             element address = A + a_off + b*a_bstride + m*lda + (k / kc)*tap_stride + (k % kc).
//   plain GEMM: ntaps=1, kc=K.  dilated conv: ntaps=taps, kc=C_in, tap_stride=dil*C_in.
//   strided conv / transposed conv: a single contiguous window per row with lda < kc (overlapping).
// Rows are clamped to M-1 on load (halo rows make shifted reads legal) and masked on store.
struct GemmParams {
  const void* A;
  const void* W;  // [N][K] row-major, K contiguous; same element type as A (flags bit 11: K-tile-major [K/64][N][64])
  long a_off, a_bstride, lda, tap_stride;
  int kc;
  int M, N, K, nbatch;
  // epilogue:  v = acc (+ bias[n % chan_mod]);  swiglu: v = silu(v_even_blk) * v_odd_blk
  //            v *= gate_tab[n] + gate[gate_row*gate_ld + n]   (if gate)   ; v *= alpha
  //            v += res[...]                                    (if res)
  //            out_f32[...] = v ; out_act[...] = act(v)
  const float* bias;
  int chan_mod;
  int swiglu;
  const float* gate_tab;
  const float* gate;
  long gate_ld;
  int rows_per_gate;
  float alpha;
  const float* res;
  long res_bstride, res_ld, res_off;
  float* out_f32;
  long f32_bstride, f32_ld, f32_off;
  void* out_act;
  long act_bstride, act_ld, act_off;
  int act;
  int f32_act;             // 1: out_f32 receives act(v) instead of v
  const float* act_alpha;  // snake alpha per channel (n % chan_mod)
  long c_lo, c_hi;         // valid range of (m*c_ld_rel + n); c_ld_rel = f32_ld or act_ld
  long c_ld_rel;
  long w_bstride;          // per-batch offset of W in elements (0: one weight matrix for every batch)
  int raster_gm;           // gemm2/gemm3 tile raster: M-tiles per group (0 = default 8)
  int flags;               // launch switches: bit 0 (set by launch_gemm2) accumulator-layout epilogue; bit 1 (set by the
                           // caller) never split the launch into whole rounds + tail (gemm.hip gemm_tail_split);
                           // bits 2-3 / 4-5 (fp32 kernel only): round the A / W operand to bf16 (1) or fp16 (2) first;
                           // bit 6 (set by the gemm8.hip launchers): linear epilogue (gemm8_linear_epilogue)
                           // bits 9 / 10: 16-bit output / operands in the alt format (mixed mode)
                           // bit 11: W is K-TILE-MAJOR, [K/64][N][64] - the 64-element K slab of ALL N rows contiguous, so a
                           // launch streams W front to back (8-phase family only, plain operands; weights.py ktm_layout)
  int tag;                 // 1: DAC-VAE launch - same code under its own kernel symbol (rocprofv3 / roofline attribution)
  // Optional (8-phase family, launches of fewer than 256 workgroups): the workgroups that would otherwise idle touch these bytes
  // once, line by line - the NEXT launch's weights, so that it finds them in the memory-side cache instead of fetching them
  // cold.  Results do not depend on it.
  const void* pf_ptr;
  long pf_bytes;
};
constexpr int GEMM_FLAG_W_KTM = 2048;
// flags bit 12 (8-phase family, SwiGLU launches with a 16-bit output only): out_act receives the COMPENSATED-operand form of the
// result - row stride act_ld = 3 * (N / 2), [lo | hi | hi] with hi = rn16(v), lo = rn16(v - hi) - i.e. the next GEMM's split
// activation operand straight from the fp32 accumulators (SAMAUDIO_OPT_X3_CLASSES; kernels.hip split3_kernel is the stand-alone form)
constexpr int GEMM_FLAG_OUT_SPLIT3 = 4096;
// flags bit 13 (fp32 kernel of gemm.hip only): COMPENSATED 16-bit multiply of fp32 operands, split ON THE FLY - the fp32 fragments of
// a K slab are split in registers into hi = rn16(x), lo = rn16(x - hi) and multiplied as lo*hi + hi*lo + hi*hi on the 16-bit MFMA
// (3 x 16x16x32 instead of 8 x 16x16x4f32 per 32 k: the fp32 product to ~2^-21, 2.7x less matrix-core time; no second copy of
// anything).  SAMAUDIO_OPT_X3_CLASSES bit SAMAUDIO_CLS_CODEC: the DAC-VAE convolutions of an fp32 context.
constexpr int GEMM_FLAG_X3_FLY = 8192;
// flags bit 14 (with bit 13 only): W is the launch's weight ALREADY SPLIT, in the layout the on-the-fly kernel's fragment reads want -
// same size and row stride as the fp32 matrix; the 128 bytes of a row's 32-k slab hold eight 16-byte chunks: chunk c < 4 = the hi
// halves of k = 4c .. 4c+3 and 16+4c .. 16+4c+3 (one lane's operand of a 16x16x32 MFMA), chunk 4 + c = their lo halves
// (weights.py codec_fly16 makes it; the same hi / lo bits the kernel would compute from the fp32 weight, so the results are the same
// bits).  Weights are constants: splitting them once per model instead of once per tile and slab takes the split of W out of the
// K loop, which for the narrow DAC-VAE stages (N = 64 .. 192) is most of its vector-ALU work.
constexpr int GEMM_FLAG_W_FLY16 = 16384;
// flags bit 15 (8-phase family, plain 16-bit launches): the operands are the K-CONCATENATED split operands of the compensated mode -
// A rows [x_lo | x_hi | x_hi], W rows [W_hi | W_lo | W_hi], K = 3 x the original K, a multiple of 192 - and the kernel may use that: the
// three products of one original K-tile run back to back in the order lo.hi, hi.hi, hi.lo, sharing the operand tiles they have in
// common (gemm8.hip gemm8x_kernel: a third less staging traffic; gemm8s_kernel: the same order through its K-tile index map).  The
// result is the same sum in another order.
constexpr int GEMM_FLAG_X3_SHARE = 32768;

}  // namespace sa
