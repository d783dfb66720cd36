// Generalised GEMM / implicit 1-D convolution on the CDNA4 matrix cores (gfx950).
//
// One kernel serves every dense contraction on the separate() path: the DiT linears
// (reference transformer.py:102-114,186-189,426-428,462-467), the patcher k3 convolutions
// (patcher.py:48-67), proj / memory_proj / align conv1x1 / anchor proj (model.py:90-95, align.py:17-19)
// and every DAC-VAE Conv1d / ConvTranspose1d (codec.py:65-70,86-89) - see GemmParams in common.h
// for how convolutions are expressed as overlapping-row GEMMs over channels-last, halo-padded
// activations.
//
// Structure (per workgroup, 256 threads = 4 waves of 64):
//   * BM x BN output tile, K consumed in 128-byte slabs per row (64 bf16 / 32 f32).
//   * both operands are K-contiguous ("NT"), staged HBM -> LDS with direct-to-LDS loads
//     (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR round trip), double-buffered:
//     slab k+1 streams in while slab k feeds the MFMAs; one barrier per slab.
//   * LDS image = rows of 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 so that the
//     16 lanes serviced together by ds_read_b128 hit 16 distinct 16-B slots (conflict-free).  As the
//     DMA writes lane-linear, the swizzle is applied to the per-lane SOURCE address and to the read.
//   * bf16: v_mfma_f32_16x16x32_bf16 (8 k per lane per chunk); f32: 4 x v_mfma_f32_16x16x4_f32 per
//     chunk (exact fp32, k-ordered fma chain) - identical staging, identical fragment addressing.
//   * accumulation order depends only on k => results are bitwise independent of M / batch sharding.
#include "common.h"
#include "kernels.h"

#include <type_traits>

namespace sa {

typedef h16x8_t bf16x8_t;  // 8 x 16-bit operand words (bf16, or fp16 with -DSA_OPERAND_FP16: common.h)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16x8_t frag_t;
  static __device__ __forceinline__ f32x4_t run(const frag_t& a, const frag_t& b, f32x4_t c) {
    return SA_MFMA_16x16x32(a, b, c);
  }
};
template <> struct Mma<float> {
  typedef f32x4_t frag_t;
  static __device__ __forceinline__ f32x4_t run(const frag_t& a, const frag_t& b, f32x4_t c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
  }
};

// fp32 kernel only (GemmParams.flags bits 2-3: A operand, 4-5: W operand): round an operand to a 16-bit format before the
// multiply - 1 = bfloat16, 2 = IEEE fp16, both round-to-nearest-even as the 16-bit kernels' stores do.  Products of two
// such values are exact in fp32, so the launch computes what the 16-bit MFMA would, operand class by operand class
// (SAMAUDIO_OPT_QUANT_CLASSES: the error budget of DESIGN.md section 4).
__device__ __forceinline__ float quant16(float x, int fmt) {
  if (fmt == 1) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xffff0000u);
  }
  if (fmt == 2) return (float)(_Float16)x;
  return x;
}
__device__ __forceinline__ f32x4_t quant16x4(f32x4_t v, int fmt) {
  return f32x4_t{quant16(v[0], fmt), quant16(v[1], fmt), quant16(v[2], fmt), quant16(v[3], fmt)};
}

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
  // lane i's 16 bytes land at lds_wave_base + 16*i  (wave-uniform base, lane-linear destination)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// FLY (fp32 only): the instantiation serves GEMM_FLAG_X3_FLY | GEMM_FLAG_W_FLY16 launches and nothing else (the wide tiles of the
// narrow DAC-VAE stages, fly_variant below) - no exact-fp32 path, no split of W in the K loop.
// (Measured and removed, round 6: a three-stage ring of the FLY tiles - two slabs in flight, the DMA as inline assembly with counted
// vmcnt - is 84 KB of LDS at 128 x 96, one workgroup per CU instead of two, and ran 1.35x SLOWER than this loop on the k7 shapes, on
// 128- and on 256-row tiles; profiles/r6_call15/fly_probe.log, columns ring3 / ring3 256.)
template <typename T, int BM, int BN, int WM_, int WN_, bool FLY = false>
__global__ __launch_bounds__(256, (FLY ? 2 : 1)) void gemm_kernel(const GemmParams p) {   // (two waves per SIMD: two workgroups per CU)
  constexpr int NW = WM_ * WN_;
  static_assert(NW == 4, "4 waves per workgroup");
  constexpr int CH = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = 128 / (int)sizeof(T);  // elements per 128-byte row slab
  constexpr int WTM = BM / WM_, WTN = BN / WN_;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int AI = BM / 32, BI = BN / 32;  // DMA instructions per wave per slab
  static_assert(BM % 32 == 0 && BN % 32 == 0 && WTM % 16 == 0 && WTN % 16 == 0, "tile shape");
  constexpr int TILE_A = BM * 128, TILE_B = BN * 128, STAGE = TILE_A + TILE_B;
  constexpr int NS = 2;   // stages: one slab in flight while one is multiplied
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_, wn = wave % WN_;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  int id = blockIdx.x;
  const int tn = id % tiles_n;
  id /= tiles_n;
  const int tm = id % tiles_m;
  const int b = id / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane DMA source bookkeeping ---------------------------------------------------------
  // wave-instruction j (j = wave + 4*i) fills tile rows 8j..8j+7; lane -> (row = 8j + lane/8,
  // 16-byte slot = lane%8); slot s of row r holds source chunk  s ^ ((r>>1)&7).
  const int r8 = lane >> 3;
  const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (r8 >> 1)) & 7);
  const T* a_rows[AI];
  const T* w_rows[BI];
  {
    const T* A = (const T*)p.A + p.a_off + (long)b * p.a_bstride;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int m = m0 + (wave + 4 * i) * 8 + r8;
      m = m < p.M ? m : p.M - 1;
      a_rows[i] = A + (long)m * p.lda;
    }
    const T* W = (const T*)p.W + (long)b * p.w_bstride;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int n = n0 + (wave + 4 * i) * 8 + r8;
      n = n < p.N ? n : p.N - 1;
      w_rows[i] = W + (long)n * p.K + chunk * CH;
    }
  }
  // (Measured without effect and removed, round 6: a first-touch pass - every thread requesting up to six of the 128-byte lines of the
  // tile's activation rows with ordinary loads in front of the K loop, so that the slabs of the first tap find them in L2: k7 at 96
  // channels 1.057 ms with, 0.980 without; end to end 89.17 s-audio/s both ways; profiles/r6_call17/.)
  // running position of this lane's chunk inside the (tap, offset) structure of A's k axis
  int a_in = chunk * CH;   // offset inside the current tap segment
  long a_tap = 0;          // element offset of the current tap
  while (a_in >= p.kc) { a_in -= p.kc; a_tap += p.tap_stride; }

  auto issue = [&](int stage) {
    char* sA = smem + stage * STAGE;
    char* sB = sA + TILE_A;
#pragma unroll
    for (int i = 0; i < AI; ++i) dma16(a_rows[i] + a_tap + a_in, sA + (wave + 4 * i) * 1024);
#pragma unroll
    for (int i = 0; i < BI; ++i) dma16(w_rows[i], sB + (wave + 4 * i) * 1024);
    // advance to the next slab
    a_in += BK;
    while (a_in >= p.kc) { a_in -= p.kc; a_tap += p.tap_stride; }
#pragma unroll
    for (int i = 0; i < BI; ++i) w_rows[i] += BK;
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nslab = p.K / BK;
  const int lr = lane & 15, lg = lane >> 4;
  const int qa = (p.flags >> 2) & 3, qw = (p.flags >> 4) & 3;  // operand rounding (fp32 kernel only, see quant16)
  const bool x3fly = FLY || (sizeof(T) == 4 && (p.flags & GEMM_FLAG_X3_FLY) != 0);   // (uniform)
  const bool wfly = FLY || (p.flags & GEMM_FLAG_W_FLY16) != 0;                         // (uniform) W arrives split: common.h
  // one 32-k slab of an X3_FLY launch (stage image sA | sB)
  auto fly_slab = [&](const char* sA, const char* sB) {
        // GEMM_FLAG_X3_FLY: both k-steps of the 32-element slab at once.  A lane's 8 values (k = 4 lg .. + 3 and 16 + 4 lg .. + 3) form
        // one 16x16x32 fragment; A and W use the same lane -> k map, so the MFMA pairs equal k.  Small terms first.
#pragma clang fp contract(off)
        bf16x8_t ah[FM], al[FM], bh[FN], bl[FN];
        auto cl = [](float x) { return __builtin_amdgcn_fmed3f(x, -kH16Max, kH16Max); };   // (one instruction; finite x: fmin(fmax()))
        auto split = [&](const char* base, int row, bf16x8_t& hi, bf16x8_t& lo) {
          const int sw = (row >> 1) & 7;
          const f32x4_t v0 = *(const f32x4_t*)(base + row * 128 + ((lg ^ sw) << 4));
          const f32x4_t v1 = *(const f32x4_t*)(base + row * 128 + (((4 + lg) ^ sw) << 4));
          unsigned h[4], l[4];
          h[0] = pack_h16x2(cl(v0[0]), cl(v0[1]));
          h[1] = pack_h16x2(cl(v0[2]), cl(v0[3]));
          h[2] = pack_h16x2(cl(v1[0]), cl(v1[1]));
          h[3] = pack_h16x2(cl(v1[2]), cl(v1[3]));
          l[0] = pack_h16x2(v0[0] - h16_lo(h[0]), v0[1] - h16_hi(h[0]));
          l[1] = pack_h16x2(v0[2] - h16_lo(h[1]), v0[3] - h16_hi(h[1]));
          l[2] = pack_h16x2(v1[0] - h16_lo(h[2]), v1[1] - h16_hi(h[2]));
          l[3] = pack_h16x2(v1[2] - h16_lo(h[3]), v1[3] - h16_hi(h[3]));
          typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
          hi = __builtin_bit_cast(bf16x8_t, (u32x4_t){h[0], h[1], h[2], h[3]});
          lo = __builtin_bit_cast(bf16x8_t, (u32x4_t){l[0], l[1], l[2], l[3]});
        };
        if (wfly) {   // GEMM_FLAG_W_FLY16: chunk lg of the row's slab IS the lane's hi operand, chunk 4 + lg its lo operand
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int row = wn * WTN + j * 16 + lr, sw = (row >> 1) & 7;
            bh[j] = *(const bf16x8_t*)(sB + row * 128 + ((lg ^ sw) << 4));
            bl[j] = *(const bf16x8_t*)(sB + row * 128 + (((4 + lg) ^ sw) << 4));
          }
        } else {
#pragma unroll
          for (int j = 0; j < FN; ++j) split(sB, wn * WTN + j * 16 + lr, bh[j], bl[j]);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) split(sA, wm * WTM + i * 16 + lr, ah[i], al[i]);
        // term-major: FM x FN independent accumulators between two MFMAs on the same one (per element the order stays lo.hi, hi.lo, hi.hi)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = SA_MFMA_16x16x32(bh[j], al[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = SA_MFMA_16x16x32(bl[j], ah[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = SA_MFMA_16x16x32(bh[j], ah[i], acc[i][j]);
  };
  issue(0);
  for (int s = 0; s < nslab; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // slab s landed for every wave; everyone is done reading the other stage
    if (s + 1 < nslab) issue((s + 1) & 1);
    const char* sA = smem + (s & 1) * STAGE;
    const char* sB = sA + TILE_A;
    if constexpr (sizeof(T) == 4) {
      if (x3fly) {
        fly_slab(sA, sB);
        continue;
      }
    }
    if constexpr (!FLY) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename Mma<T>::frag_t af[FM], bfr[FN];
      const int c = ks * 4 + lg;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * WTM + i * 16 + lr;
        af[i] = *(const typename Mma<T>::frag_t*)(sA + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * WTN + j * 16 + lr;
        bfr[j] = *(const typename Mma<T>::frag_t*)(sB + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
      if constexpr (sizeof(T) == 4) {
        if (qa) {
#pragma unroll
          for (int i = 0; i < FM; ++i) af[i] = quant16x4(af[i], qa);
        }
        if (qw) {
#pragma unroll
          for (int j = 0; j < FN; ++j) bfr[j] = quant16x4(bfr[j], qw);
        }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = Mma<T>::run(bfr[j], af[i], acc[i][j]);   // swapped: see the epilogue
    }
    }   // !FLY
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  // The operands go into the MFMA swapped (W fragment first): per fragment D = (A W^T)^T, i.e. a lane holds row m = lane & 15
  // of m-fragment i and the FOUR CONSECUTIVE columns n = 16 j + 4 (lane >> 4) .. + 3 of n-fragment j.  Bias / gate / residual
  // are 16-byte loads and the outputs 16- / 8-byte stores wherever the addresses allow it (the plain layout - one column,
  // four rows per lane - moved every element on its own: the 2816-wide fp32 input projection of the 16-bit modes ran at
  // 16 TF/s, 355 us for 68 MB of output; profiles/r3_call14/bench.log).  Element arithmetic and its order are unchanged.
  const long bM = (long)b * p.M;
  const int n_out = p.swiglu ? p.N >> 1 : p.N;
  auto finish = [&](float v, const float bias, const float gate, const float res, const float sa, float& a) -> float {
    v += bias;
    if (p.gate) v *= gate;
    v *= p.alpha;
    v += res;
    a = v;
    if (p.act == ACT_SNAKE) a = snake_f(v, sa);
    else if (p.act == ACT_TANH) a = tanhf(v);
    else if (p.act == ACT_SILU) a = silu_f(v);
    else if (p.act == ACT_GELU) a = gelu_f(v);
    else if (p.act == ACT_QUICK_GELU) a = quick_gelu_f(v);
    else if (p.act == ACT_RELU) a = relu_f(v);
    else if (p.act == ACT_GELU_TANH) a = gelu_tanh_f(v);
    return v;
  };
  if constexpr (sizeof(T) == 4) {
    // LEAN form: fp32 launches whose operands allow 16-byte accesses throughout and whose activation is none or Snake - every
    // DAC-VAE convolution of an fp32 context except the windowed transposed ones and the final tanh; the input projections of the DiT.
    // The general loop below is unrolled over the FM x FN fragments with the seven-way activation chain of finish() inlined per
    // element: ~50 000 instructions per instantiation, walked once per tile from a cold instruction cache, each fragment's loads
    // issued behind the previous fragment's stores (vmcnt retires in order: the load is waited for together with the store's round
    // trip).  Measured per 128 x 96 tile: 26 - 32 us of epilogue in front of 21 us of K loop; a k1 convolution - three slabs per tile -
    // ran at 1.4 TB/s of its 1.47 GB (profiles/r6_call13/, r6_call14/fly_probe.log).  With the form below the same launch runs at
    // 3.3 TB/s and the k7 convolutions 1.5 - 2x faster (r6_call15/).  Same expressions in the same order: the same bits.
    auto al16 = [](const void* q, long a, long b2, long c) { return (((size_t)q) & 15) == 0 && !((a | b2 | c) & 3); };
    const bool lean = !p.swiglu && !p.gate && !p.chan_mod && !p.c_ld_rel && !(p.N & 3) && (!p.bias || al16(p.bias, 0, 0, 0)) &&
                      (p.act != ACT_SNAKE || al16(p.act_alpha, 0, 0, 0)) && (!p.res || al16(p.res, p.res_off, p.res_ld, p.res_bstride)) &&
                      (!p.out_f32 || al16(p.out_f32, p.f32_off, p.f32_ld, p.f32_bstride)) &&
                      (!p.out_act || al16(p.out_act, p.act_off, p.act_ld, p.act_bstride));
    if (lean && (p.act == ACT_NONE || p.act == ACT_SNAKE)) {
      // accumulators -> LDS (the K loop's stages are free after one barrier): rows of BN floats, 16-byte chunks XOR-swizzled with the
      // row.  Then a ROLLED loop whose body exists once: a wave instruction covers 64 / (BN / 4) whole rows - full cache lines per
      // load and store -, the residual row of iteration it + 1 is requested before the stores of iteration it.
      constexpr int CPR = BN / 4, RPI = 64 / CPR, WR = BM / 4, NIT = (WR + RPI - 1) / RPI;   // chunks per row, rows per wave instruction, rows per wave
      static_assert(BN % 32 == 0 && BM * BN * 4 <= NS * STAGE, "epilogue staging area");
      __syncthreads();
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = wm * WTM + i * 16 + lr, c = wn * (WTN / 4) + j * 4 + lg;
          *(f32x4_t*)(smem + row * (BN * 4) + ((c ^ (row & 7)) << 4)) = acc[i][j];
        }
      __syncthreads();
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const int c = lane % CPR, rsub = lane / CPR;
      const bool lane_on = rsub < RPI;
      const int n = n0 + c * 4;
      const int ncl = n < p.N ? n : p.N - 4;   // clamped address for the loads of columns past N (stores masked)
      const float4 bb = p.bias ? *(const float4*)(p.bias + ncl) : z;
      const float4 sa = p.act == ACT_SNAKE ? *(const float4*)(p.act_alpha + ncl) : z;
      const float* const res0 = p.res ? p.res + p.res_off + (long)b * p.res_bstride + ncl : nullptr;
      float* const f320 = p.out_f32 ? p.out_f32 + p.f32_off + (long)b * p.f32_bstride + n : nullptr;
      float* const act0 = p.out_act ? (float*)p.out_act + p.act_off + (long)b * p.act_bstride + n : nullptr;
      const int rw = wave * WR;   // this wave's rows of the tile
      auto body = [&](auto SNAKE) {
        // finish() without the gate and with the activation known (same expressions, same order)
        auto fin = [&](float v, const float bias, const float res, const float sal, float& a) -> float {
          v += bias;
          v *= p.alpha;
          v += res;
          a = decltype(SNAKE)::value ? snake_f(v, sal) : v;
          return v;
        };
        auto res_of = [&](int it) {
          const int m = m0 + rw + it * RPI + rsub;
          return res0 ? *(const float4*)(res0 + (long)(m < p.M ? m : p.M - 1) * p.res_ld) : z;
        };
        float4 rr = res_of(0);
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
          const float4 rn = it + 1 < NIT ? res_of(it + 1) : z;   // requested ahead of this iteration's stores (vmcnt retires in order)
          const int rl = it * RPI + rsub, row = rw + (rl < WR ? rl : 0), m = m0 + row;
          const float4 sv = *(const float4*)(smem + row * (BN * 4) + ((c ^ (row & 7)) << 4));
          float a0, a1, a2, a3;
          const float v0 = fin(sv.x, bb.x, rr.x, sa.x, a0), v1 = fin(sv.y, bb.y, rr.y, sa.y, a1), v2 = fin(sv.z, bb.z, rr.z, sa.z, a2),
                      v3 = fin(sv.w, bb.w, rr.w, sa.w, a3);
          if (lane_on && rl < WR && m < p.M && n < p.N) {
            if (f320) *(float4*)(f320 + (long)m * p.f32_ld) = p.f32_act ? make_float4(a0, a1, a2, a3) : make_float4(v0, v1, v2, v3);
            if (act0) *(float4*)(act0 + (long)m * p.act_ld) = make_float4(a0, a1, a2, a3);
          }
          rr = rn;
        }
      };
      if (p.act == ACT_SNAKE) body(std::true_type{});
      else body(std::false_type{});
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + lr;
    if (m >= p.M) continue;
    const float* grow = nullptr;
    if (p.gate) grow = p.gate + ((bM + m) / p.rows_per_gate) * p.gate_ld;
    const float* rrow = p.res ? p.res + p.res_off + (long)b * p.res_bstride + (long)m * p.res_ld : nullptr;
    float* frow = p.out_f32 ? p.out_f32 + p.f32_off + (long)b * p.f32_bstride + (long)m * p.f32_ld : nullptr;
    T* arow = p.out_act ? (T*)p.out_act + p.act_off + (long)b * p.act_bstride + (long)m * p.act_ld : nullptr;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if (p.swiglu && (j & 1)) continue;
      f32x4_t v4 = acc[i][j];
      int n = n0 + wn * WTN + j * 16 + lg * 4;
      if (p.swiglu) {  // w1 / w3 rows interleaved in 16-row blocks: fragment j | 1 holds the matching w3 columns
        const f32x4_t g = acc[i][j | 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = silu_f(v4[e]) * g[e];
        n = ((n0 + wn * WTN) >> 1) + (j >> 1) * 16 + lg * 4;
      }
      if (n >= n_out) continue;
      // 16-byte path: all four columns exist, no per-element channel folding / output window, every operand aligned
      size_t al = 0;
      if (p.bias) al |= (size_t)(p.bias + n);
      if (grow) al |= (size_t)(grow + n);
      if (p.gate_tab) al |= (size_t)(p.gate_tab + n);
      if (rrow) al |= (size_t)(rrow + n);
      if (p.act == ACT_SNAKE) al |= (size_t)(p.act_alpha + n);
      if (frow) al |= (size_t)(frow + n);
      if (arow) al |= (size_t)(arow + n) * (16 / (4 * sizeof(T)));   // 4 elements of T: 16 bytes (fp32) or 8 (16-bit)
      if (n + 4 <= n_out && !p.chan_mod && !p.c_ld_rel && (al & 15) == 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 bb = p.bias ? *(const float4*)(p.bias + n) : z;
        float4 gg = grow ? *(const float4*)(grow + n) : z;
        if (grow && p.gate_tab) { const float4 tt = *(const float4*)(p.gate_tab + n); gg.x += tt.x; gg.y += tt.y; gg.z += tt.z; gg.w += tt.w; }
        const float4 rr = rrow ? *(const float4*)(rrow + n) : z;
        const float4 sa = p.act == ACT_SNAKE ? *(const float4*)(p.act_alpha + n) : z;
        float a0, a1, a2, a3;
        const float v0 = finish(v4[0], bb.x, gg.x, rr.x, sa.x, a0), v1 = finish(v4[1], bb.y, gg.y, rr.y, sa.y, a1),
                    v2 = finish(v4[2], bb.z, gg.z, rr.z, sa.z, a2), v3 = finish(v4[3], bb.w, gg.w, rr.w, sa.w, a3);
        if (frow) *(float4*)(frow + n) = p.f32_act ? make_float4(a0, a1, a2, a3) : make_float4(v0, v1, v2, v3);
        if (arow) store4<T>(arow + n, a0, a1, a2, a3);
        continue;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ne = n + e;
        if (ne >= n_out) continue;
        const int ch = p.chan_mod ? ne % p.chan_mod : ne;
        const long erel = (long)m * p.c_ld_rel + ne;
        if (p.c_ld_rel && (erel < p.c_lo || erel >= p.c_hi)) continue;
        float a;
        const float v = finish(v4[e], p.bias ? p.bias[ch] : 0.f, grow ? (p.gate_tab ? p.gate_tab[ne] : 0.f) + grow[ne] : 0.f,
                               rrow ? rrow[ne] : 0.f, p.act == ACT_SNAKE ? p.act_alpha[ch] : 0.f, a);
        if (frow) frow[ne] = p.f32_act ? a : v;
        if (arow) Elem<T>::store(arow + ne, a);
      }
    }
  }
}

template <typename T, int BM, int BN, int WM_, int WN_>
static hipError_t launch_cfg(const GemmParams& p, hipStream_t st) {
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.nbatch;
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM_, WN_>), dim3((unsigned)tiles), dim3(256), 0, st, p);
  return hipGetLastError();
}

static int g_force = -1;
void gemm_force_variant(int v) { g_force = v; debug_touch(); }

// tile variant of gemm.hip: 0 = 128x128, 1 = 128x64, 2 = 128x32.  Narrow outputs (codec stages with 1 / 64 /
// 96 / 192 channels) use narrower N tiles.
static int gemm1_variant(const GemmParams& p) {
  const int N = p.N;
  if (p.swiglu) return 0;
  if (N <= 32 || N == 96) return 2;
  if (N <= 64 || (N % 128 != 0 && N % 64 == 0 && N <= 448)) return 1;
  // Few tiles (the fp32 classes of a 16-bit context: D -> 256 output projection at 8000 rows = 126 tiles of 128x128; one
  // time row; 256 text rows): narrower tiles until every CU has one.  Every tile of this file accumulates an output
  // element over k in the same order, so the choice may depend on M without breaking batch-sharding invariance.
  const long t128 = (long)((p.M + 127) / 128) * ((N + 127) / 128) * p.nbatch;
  if (t128 < 192) return 2 * t128 < 192 ? 2 : 1;
  return 0;
}
// Which kernel launch_gemm runs: 0..2 = gemm.hip tiles, 3 + v = gemm2.hip / gemm8.hip variant v (16-bit operands only).  The
// choice depends on (N, K, epilogue) and - only INSIDE one accumulation-order family - on the number of rows, so that
// sharding the batch cannot change the accumulation order of any output element:
//   16x16x32 family: 22 = gemm8 (256x256, 8-phase K loop), 27 = gemm8s (128x128) - every N >= 256;
//   32x32x16 family: 25 / 26 (128x128, 64x128 BK 64), 28 (256x64), 29 / 32 / 33 / 34 (BK 32 multi-workgroup tiles),
//                    35 = conv7h - the DAC-VAE stages with 64 .. 192 channels.
// How each entry was chosen (one-box A/B per step, round 2): DESIGN.md sections 3.1 and 3.4, profiles/r2_call*/.
// GEMM_FLAG_X3_FLY | GEMM_FLAG_W_FLY16 launches (the DAC-VAE stages with < 256 channels in an fp32 context): 4 x 1 waves, each 32
// rows x the whole tile width.  In the K loop a wave then splits only ITS OWN two activation fragments (2 x 28 vector-ALU
// instructions per 32-k slab) and multiplies them with every column fragment of the tile: 36 MFMAs per slab at N = 96 where the
// 128 x 32 tile of gemm1_variant had 12 MFMAs behind the split of two activation AND two weight fragments, three times per row block
// (one workgroup per 32 columns).  128 x 96 for N = 96 / 192, 128 x 128 for N = 128, 128 x 64 / 128 x 32 below.  Same k order per
// output element as every tile of this file.
static int fly_variant(const GemmParams& p) {
  if (p.N <= 32) return 39;
  if (p.N <= 64) return 38;
  if (p.N % 96 == 0 || p.N <= 96) return 36;
  return 37;
}
int gemm_variant(const GemmParams& p, bool is_bf16) {
  const bool g2 = is_bf16 && gemm2_ok(p);
  if (!is_bf16 && g_force < 0 && (p.flags & GEMM_FLAG_W_FLY16) && debug_flag(36) != 1) return fly_variant(p);   // flag 36 = 1 (A/B): the old tiles
  // (K-tile-major weights / the split-form output exist in the 8-phase family only: a FORCED variant other than 22 / 27 on such a
  // launch is refused by gemm_check with that reason - tests/test_gemm2_gpu.py::test_ktm_weights_are_refused_outside_the_8phase_family;
  // the policy itself never leaves the family for them, and every side operand the engine registers is 16-byte aligned by
  // samaudio_set_tensor, so gemm2_ok cannot fail for a model's launches)
  if (g_force >= 3) {
    const bool known = g_force == 22 || (g_force >= 25 && g_force <= 29) || (g_force >= 32 && g_force < kGemmVariants);
    if (g_force == 35 && !(g2 && conv7h_ok(p))) return gemm1_variant(p);   // conv7h computes convolutions only
    return g2 && known ? g_force : gemm1_variant(p);
  }
  if (g_force >= 0 || !g2) return gemm1_variant(p);
  const long rows256 = (long)((p.M + 255) / 256) * p.nbatch, rows128 = (long)((p.M + 127) / 128) * p.nbatch;
  // k7 'same' convolutions of the DAC stages with <= 96 channels: halo tile resident in LDS (C = 64 485 vs 567 us, C = 96
  // 1149 vs 1307 us for the implicit GEMM; C = 128 / 192 measured slower and stay implicit GEMMs).  Flag 11 = off (tests).
  if (!debug_flag(11) && p.N <= 96 && conv7h_ok(p) && rows256 >= 256) return 35;
  // 64-channel outputs (first DAC encoder stage): one 64-wide tile of the DMA-fed family.  The small-launch fallback stays
  // in the SAME MFMA family: how many waveforms one codec pass holds depends on the caller's workspace, and a family
  // switch at a row-count threshold made two identical calls differ in the last bits (round 2, GPU call 14).
  if (p.N >= 64 && p.N < 96 && p.K >= 64) return rows256 >= 256 ? 28 : 32;
  // 96 - 192 channels with very many rows: a K-tile of such a tile is ~0.4 us of MFMA work behind ~2 us of L2 latency, so
  // what pays is MORE TILES IN FLIGHT per CU: BK 32, 36 - 60 KiB per workgroup, 2 - 4 workgroups per CU.
  if (p.N >= 96 && p.N <= 192 && p.K >= 64 && rows128 >= 1024) {
    if (p.N <= 128) return p.K <= 256 ? 33 : 29;   // 64x128 k32 s3 (k1) | 128x128 k32 s3 (k7)
    return p.K <= 256 ? 29 : 34;                   // 128x128 k32 s3 (k1) | 128x192 k32 s3 (k7)
  }
  if (p.N >= 96 && p.K >= 128) {
    if (p.N >= 256) {
      // The 8-phase family: gemm8 when the 256x256 tiling yields >= 128 tiles, else its 128x128 tile (few rows: 4 clips per
      // GPU, the Judge's small batches; N = 384 whatever M).  (Round 3, GPU call 2: the same kernel on a 256x192 tile - 480
      // instead of 352 tiles at N = D - was 8 - 13 % faster per launch on the N = D shapes and 5 % SLOWER end to end: a
      // 256x192 tile delivers 20 % fewer flops per CU-second, and the CUs a 352-tile launch leaves idle are not idle in the
      // timed configuration - the other row group's HBM-bound kernels run on them.  profiles/r3_call2/.)
      const long t256 = rows256 * ((p.N + 255) / 256);
      if (p.N < 1024 && p.N % 256) return 27;
      const long min256 = debug_flag(30) > 0 ? debug_flag(30) : 128;   // flag 30 (A/B): the tile count from which gemm8 is used
      return t256 >= min256 ? 22 : 27;
    }
    // 96 <= N < 256 with few rows: 128- / 64-row tiles of the 32x32x16 family
    return rows128 * ((p.N + 127) / 128) >= 192 ? 25 : 26;
  }
  return gemm1_variant(p);
}
const char* gemm_variant_name(int v, bool is_bf16) {
  if (v < 0 || v >= kGemmVariants) return "";
  if (v >= 36) {
    static const char* fly[4] = {"gemm_f32x3_128x96", "gemm_f32x3_128x128", "gemm_f32x3_128x64", "gemm_f32x3_128x32"};
    return is_bf16 ? "" : fly[v - 36];
  }
  if (v < 3) {
    static const char* base[2][3] = {{"gemm_f32_128x128", "gemm_f32_128x64", "gemm_f32_128x32"},
                                     {"gemm_bf16_128x128", "gemm_bf16_128x64", "gemm_bf16_128x32"}};
    return base[is_bf16 ? 1 : 0][v];
  }
  if (!is_bf16) return "";
  switch (v) {
    case 22: return "gemm8_bf16_256x256_8phase";
    case 25: return "gemm2_bf16_128x128_s2";
    case 26: return "gemm2_bf16_64x128_s3";
    case 27: return "gemm8s_bf16_128x128";
    case 28: return "gemm2_bf16_256x64_s2";
    case 29: return "gemm2_bf16_128x128_k32_s3";
    case 32: return "gemm2_bf16_128x64_k32_s2";
    case 33: return "gemm2_bf16_64x128_k32_s3";
    case 34: return "gemm2_bf16_128x192_k32_s3";
    case 35: return "conv7h_bf16";
    default: return "";
  }
}

template <typename T>
static hipError_t launch_t(const GemmParams& p, int variant, hipStream_t st) {
  switch (variant) {
    case 2: return launch_cfg<T, 128, 32, 4, 1>(p, st);
    case 1: return launch_cfg<T, 128, 64, 2, 2>(p, st);
    default: return launch_cfg<T, 128, 128, 2, 2>(p, st);
  }
}

// 8-phase launches whose last round of 256x256 tiles would be mostly empty are split (gemm8.hip launch_gemm8_split):
// returns the number of tiles the main launch keeps, 0 = one launch.  Only when the policy (not a forced variant) chose
// the kernel; the tail must be worth a launch (>= 16 tiles) and the last round must be at most half full.
int gemm_tail_split(const GemmParams& p, bool is_bf16) {
  if (g_force >= 0 || (p.flags & 2) || gemm_variant(p, is_bf16) != 22) return 0;
  const long tiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * p.nbatch;
  const long full = tiles / 256 * 256, rem = tiles - full;
  // ... and only for launches of a few rounds: with many rounds the idle part of the last one is a small share of the
  // launch and the second kernel costs more than it recovers (vision tower, M = 144 250: qkv 749 -> 646 TF/s, out_proj
  // 557 -> 500 with the split; profiles/r2_call15/)
  // (round 4: with the lean epilogues a last round that is half full is cheaper as it stands - w13 at 4000 rows, 944 tiles = 3 x 256
  // + 176: 249 us in one launch, 268 split; qkv, 528 = 2 x 256 + 16: 179 vs 160 split; profiles/r4_call5/gemm_bench_f3.log)
  // (round 5: the floor of the tail went from 16 to 8 tiles - qkv at 8 clips per GPU, the 4-GPU share of the 32-clip batch, is 264
  //  tiles = one round + 8, i.e. a second round of 8 tiles as long as the first: 164.8 / 165.3 -> 170.0 s-audio/s at 8 clips, nothing
  //  at 4 clips or small* (profiles/r5_call12/); flag 33 = another floor for the A/B)
  const long min_tail = debug_flag(33) > 0 ? debug_flag(33) : 8;
  return full >= 256 && full <= 1024 && rem >= min_tail && rem <= 128 ? (int)full : 0;
}
hipError_t launch_gemm_part(const GemmParams& p, bool is_bf16, int part, hipStream_t st) {
  return launch_gemm8_split(p, gemm_tail_split(p, is_bf16), part, st);
}

// host entry used by the engine and by the C-ABI test hook; is_bf16 selects the element type
hipError_t launch_gemm(const GemmParams& p, bool is_bf16, hipStream_t st) {
  if (const int full = gemm_tail_split(p, is_bf16)) {
    const hipError_t e = launch_gemm8_split(p, full, 0, st);
    return e != hipSuccess ? e : launch_gemm8_split(p, full, 1, st);
  }
  const int v = gemm_variant(p, is_bf16);
  if (v >= 36) {   // fly_variant (fp32 operands, weights already split)
    const auto go = [&](auto kern, int bm, int bn) {
      const long tiles = (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.nbatch;
      hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), 0, st, p);
      return hipGetLastError();
    };
    switch (v) {
      case 36: return go(gemm_kernel<float, 128, 96, 4, 1, true>, 128, 96);
      case 37: return go(gemm_kernel<float, 128, 128, 4, 1, true>, 128, 128);
      case 38: return go(gemm_kernel<float, 128, 64, 4, 1, true>, 128, 64);
      default: return go(gemm_kernel<float, 128, 32, 4, 1, true>, 128, 32);
    }
  }
  if (v >= 3) return launch_gemm2(p, v - 3, st);
  return is_bf16 ? launch_t<bf16_t>(p, v, st) : launch_t<float>(p, v, st);
}

const char* gemm_check(const GemmParams& p, bool is_bf16) {
  const int bk = is_bf16 ? 64 : 32, ch = is_bf16 ? 8 : 4;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.nbatch <= 0) return "gemm: empty problem";
  if (p.K % bk) return "gemm: K must be a multiple of 64 (bf16) / 32 (f32) - pad W with zeros";
  if (p.kc % ch || p.kc <= 0) return "gemm: kc must be a positive multiple of the 16-byte chunk";
  if ((p.lda % ch) || (p.tap_stride % ch) || (p.a_off % ch) || (p.a_bstride % ch))
    return "gemm: A strides/offsets must be 16-byte aligned";
  if (p.swiglu && (p.N % 32)) return "gemm: swiglu needs N % 32 == 0";
  if (p.gate && p.rows_per_gate <= 0) return "gemm: rows_per_gate";
  if ((p.flags & GEMM_FLAG_W_FLY16) && (is_bf16 || !(p.flags & GEMM_FLAG_X3_FLY)))
    return "gemm: split-weight layout (flags bit 14): fp32 launches with operands split on the fly (bit 13) only";
  if (p.flags & (512 | 1024)) {   // mixed mode: alt-format output / operands exist in the 8-phase family only
    if (!is_bf16 || !gemm2_ok(p) || !gemm8_alt_ok(p)) return "gemm: alt 16-bit format: plain 16-bit launches with a lean epilogue only";
    const int v = gemm_variant(p, is_bf16);
    if (v != 22 && v != 27) return "gemm: alt 16-bit format: the tile policy did not pick the 8-phase family for this launch";
  }
  if (p.flags & GEMM_FLAG_OUT_SPLIT3) {   // compensated-operand output: the register epilogue of the 8-phase family, SwiGLU launches
    if (!is_bf16 || !gemm2_ok(p) || !gemm8_split3_ok(p)) return "gemm: split3 output: 16-bit SwiGLU launches with a lean epilogue only";
    const int v = gemm_variant(p, is_bf16);
    if (v != 22 && v != 27) return "gemm: split3 output: the tile policy did not pick the 8-phase family for this launch";
  }
  if (p.flags & GEMM_FLAG_X3_SHARE) {   // operand-sharing walk of K-concatenated split operands: the 8-phase family, plain launches
    if (!is_bf16 || !gemm2_ok(p) || !gemm8_share_ok(p)) return "gemm: shared split operands (flags bit 15): plain 16-bit launches with K' = 3K, K % 64 == 0";
    const int v = gemm_variant(p, is_bf16);
    if (v != 22 && v != 27) return "gemm: shared split operands: the tile policy did not pick the 8-phase family for this launch";
  }
  if (p.flags & GEMM_FLAG_W_KTM) {   // K-tile-major weights: only the 8-phase family addresses them
    if (!is_bf16 || !gemm2_ok(p) || p.w_bstride) return "gemm: K-tile-major W: plain 16-bit launches with one weight matrix only";
    const int v = gemm_variant(p, is_bf16);
    if (v != 22 && v != 27) return "gemm: K-tile-major W: the tile policy did not pick the 8-phase family for this launch";
  }
  return nullptr;
}

}  // namespace sa
