// gemm8: 256x256 bf16 GEMM with the "8-phase" K loop of the CDNA4 guide (cdna_hip_programming.md section 5, "The 256^2
// 8-phase template": 16x16x32 MFMA, 8 waves as 2 (M) x 4 (N), 128 KiB of LDS as 2 K-tile buffers x 4 half-tiles, one
// half-tile staged per phase, counted vmcnt - never 0 in the steady state -, two wave groups one barrier apart so that
// one group's MFMA cluster overlaps the other's LDS reads on every SIMD).
//
// Shipped since round 2 for every GEMM with N >= 1024 (gemm.hip gemm_variant): measured on MI355X at 1 349 TF/s on 8192^3
// and 860 - 1 134 TF/s on the DiT's shapes (the round-1 ablation, profiles/r1_gemm_ablation.log,
// had shown the LDS side of the earlier kernels - not their wave schedules - as the limiter).  Epilogue = the full GemmParams contract,
// operands swapped (W fragment as the MFMA's A operand) so that a lane owns 4 consecutive output columns of one row.
// gemm8s_kernel below is the same arithmetic on a 128 x 128 tile (few rows; the tail of a split launch).
//
// Geometry.  Tile 256 x 256, BK = 64.  Wave w: wr = w >> 2 (M half), wc = w & 3 (N quarter) -> output 128 x 64 =
// acc[8 m-fragments][4 n-fragments] of 16 x 16.  A K-tile in LDS = 4 half-tiles of 128 rows x 128 B:
// HA0 / HA1 (activation rows 0-127 / 128-255: exactly what the waves with wr = 0 / 1 read), HB0 / HB1 (weight rows =
// output columns 0-127 / 128-255: waves with wc>>1 = 0 / 1).  16-byte chunks XOR-swizzled with (row>>1)&7 - applied to
// the DMA's per-lane SOURCE address (the LDS image of global_load_lds is lane-linear) and again on the ds_read_b128.
//
// Phases of K-tile t (buffer t&1; "R" = reads + stage, then barrier, lgkmcnt(0), 16 MFMAs under s_setprio 1, barrier):
//   P1  R: Bs0 (4 reads), As0 (8)      stage HA0(t+1)                       M: As0 x Bs0
//   P2  R: Bs1 (4), lgkmcnt(0)         stage HA1(t+1)                       M: As0 x Bs1
//   P3  R: As1 (8)                     stage HB0(t+2)   [HB(t) last read in P2, retired by its lgkmcnt(0)]
//                                                                           M: As1 x Bs1
//   P4  R: -                           stage HB1(t+2), vmcnt(4 | 0)         M: As1 x Bs0
// (As = 64-row half of the wave's 128 rows, Bs = 32-column half of its 64 columns.)  HA(t) is last read in P3 and
// restaged in P1 / P2 of tile t+1.  The wait in P4 leaves the 4 youngest loads (HB0, HB1 of t+2) in flight and retires
// everything tile t+1 needs; its first read is one barrier later (two for the lagging group's loads: that group waits
// in its own P4-R interval, one barrier before the leading group's P1-R of the next tile).
#include "common.h"
#include "kernels.h"

#include <type_traits>

namespace sa {

typedef h16x8_t bf16x8_t;  // 8 x 16-bit operand words (bf16, or fp16 with -DSA_OPERAND_FP16: common.h)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

__device__ __forceinline__ void dma16_8(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same DMA with the source as (wave-uniform base, per-lane 32-bit byte offset) and issued as inline assembly
// (gemm8_kernel, gemm8s_kernel).  Two reasons: the scalar-base form needs no 64-bit address arithmetic per issue and half the address
// registers; and hipcc treats a global_load_lds it can see as a "flat" access pending on BOTH counters - while one is in
// flight every LDS fragment read is waited for with lgkmcnt(0) (and every ordinary load with vmcnt(0)), whatever the order
// of issue, so a kernel cannot start its MFMAs on the fragments that have already arrived.  Invisible to the compiler,
// the DMA is ordered by the kernel's own counted s_waitcnt vmcnt + barriers alone (a __syncthreads() implies NO wait for
// it).  M0 = LDS base; one wait state between the M0 write and the DMA.
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, size_t lds_wave_addr) {
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)lds_wave_addr);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory", "m0");  // SIMT-DMA8
}
// the same with a per-lane 64-bit source pointer (implicit convolutions: the tap walk; operands beyond 4 GiB)
__device__ __forceinline__ void dma16v(const void* gsrc, size_t lds_wave_addr) {
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)lds_wave_addr);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(gsrc) : "memory", "m0");  // SIMT-DMA8V
}
__device__ __forceinline__ float act_apply8(float v, int act, float snake_alpha) {
  if (act == ACT_SILU) return silu_f(v);
  if (act == ACT_GELU) return gelu_f(v);
  if (act == ACT_QUICK_GELU) return quick_gelu_f(v);
  if (act == ACT_RELU) return relu_f(v);
  if (act == ACT_GELU_TANH) return gelu_tanh_f(v);
  if (act == ACT_TANH) return tanhf(v);
  if (act == ACT_SNAKE) return snake16_f(v, snake_alpha);
  return v;
}

// Epilogue of the 16x16x32 family (contract of GemmParams, common.h).  A wave owns NH x 64 rows x 64 columns as
// acc[NH * 4 m-fragments][4 n-fragments]; `stg` = 16 KiB of LDS private to the wave (free after the K loop).
// Swapped operands give D[n][m]: a lane holds row m = lr of fragment i and columns 4*lg .. 4*lg+3 of fragment j, i.e.
// one accumulator register group covers 16 rows x 64 B - a store instruction straight from that layout touches 16
// different cache lines for 32 B each.  The tile is therefore re-laid through LDS (rows of 64 fp32 columns, 16-byte
// chunks XOR-swizzled with the row): after it a lane owns 4 consecutive columns and the 16 lanes of a row group cover
// 256 contiguous bytes, so residual / gate loads and fp32 / bf16 stores move whole cache lines per row
// (MI355X_MICROARCH.md "store-ISSUE-bound" epilogues, cdna_hip_programming.md T21).
template <int NH>
__device__ __forceinline__ void epilogue8(const GemmParams& p, f32x4_t (&acc)[NH * 4][4], char* const stg, const int b,
                                          const int m_wave0, const int n_wave0, const int lane) {
  const int lr = lane & 15, lg = lane >> 4;
  const long bM = (long)b * p.M;
  const bool has_bias = p.bias != nullptr, has_gate = p.gate != nullptr, has_tab = p.gate_tab != nullptr,
             has_res = p.res != nullptr, has_snake = p.act == ACT_SNAKE;
  const int n_out = p.swiglu ? p.N >> 1 : p.N;
  const int cpr = p.swiglu ? 8 : 16;                                  // 4-column chunks per staged row
  const int col0 = p.swiglu ? n_wave0 >> 1 : n_wave0;                 // first output column of this wave
  const int rsub = p.swiglu ? lane >> 3 : lane >> 4, csub = p.swiglu ? lane & 7 : lane & 15;
  const int rows_per_it = 64 / cpr;
  const int nit = cpr;  // read-phase iterations per 64-row half (64 / rows_per_it)
  // column-only operands of this lane (its 4 columns are the same in every iteration)
  const int n = col0 + csub * 4;
  const int nc = n + 4 <= n_out ? n : n_out - 4;  // clamped address for the loads of masked columns
  const int ch = p.chan_mod ? nc % p.chan_mod : nc;
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), tt = bb, sa = bb;
  if (has_bias) bb = *(const float4*)(p.bias + ch);
  if (has_tab) tt = *(const float4*)(p.gate_tab + nc);
  if (has_snake) sa = *(const float4*)(p.act_alpha + ch);
#pragma unroll
  for (int half = 0; half < NH; ++half) {
    // write phase: rows half*64 + i*16 + lr of the wave's rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 16 + lr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4_t a = acc[half * 4 + i][j];
        if (p.swiglu) {  // odd fragments hold the w3 rows matching the previous fragment's w1 rows
          if (j & 1) continue;
          const f32x4_t g = acc[half * 4 + i][j | 1];
          const int chunk = (j >> 1) * 4 + lg;
          *(float4*)(stg + row * 256 + ((chunk ^ (row & 15)) << 4)) =
              make_float4(silu_f(a[0]) * g[0], silu_f(a[1]) * g[1], silu_f(a[2]) * g[2], silu_f(a[3]) * g[3]);
        } else {
          const int chunk = j * 4 + lg;
          *(float4*)(stg + row * 256 + ((chunk ^ (row & 15)) << 4)) = make_float4(a[0], a[1], a[2], a[3]);
        }
      }
    }
    __syncthreads();
    // read phase: rows_per_it rows x cpr chunks per wave instruction.  The residual / gate operands of a row are loaded
    // PD iterations ahead of their use: vmcnt retires in issue order, so a load issued AFTER the previous iteration's
    // stores cannot be waited for without also waiting for those stores to reach L2 - every iteration would pay a full
    // store round trip (measured: gated-residual tiles +28 us over plain ones).  Issued ahead, the loads overtake nothing.
    // Explicit modulo schedule of depth 4: four bodies per trip, each consuming ITS queue slot and refilling it for the
    // row four iterations later (no register rotation, which would need the data at copy time).  Not unrolled further: fully
    // unrolled the epilogue is 6x the kernel's code - five inlined activation functions x 4 elements x 16 iterations x 2
    // halves - and the instruction fetch then costs more than the pipelining wins (798 -> 622 TF/s on a plain GEMM,
    // profiles/r2_call8/).
    float4 r0, r1, r2, r3, g0, g1, g2, g3;
    auto issue = [&](const int it, float4& rr, float4& gg) {
      const int m = m_wave0 + half * 64 + it * rows_per_it + rsub;
      const int mc = m < p.M ? m : p.M - 1;
      if (has_res) rr = *(const float4*)(p.res + p.res_off + (long)b * p.res_bstride + (long)mc * p.res_ld + nc);
      if (has_gate) gg = *(const float4*)(p.gate + ((bM + mc) / p.rows_per_gate) * p.gate_ld + nc);
    };
    const bool ahead = has_res || has_gate;
    auto body = [&](const int it, float4& rslot, float4& gslot) {
      const float4 rr = rslot;
      float4 gg = gslot;
      if (ahead && it + 4 < nit) issue(it + 4, rslot, gslot);
      const int row = it * rows_per_it + rsub;
      const float4 sv = *(const float4*)(stg + row * 256 + ((csub ^ (row & 15)) << 4));
      const int m = m_wave0 + half * 64 + row;
      const bool m_ok = m < p.M;
      const int mc = m_ok ? m : p.M - 1;
      float v[4] = {sv.x, sv.y, sv.z, sv.w};
      if (has_bias) { v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
      if (has_gate) {
        if (has_tab) { gg.x += tt.x; gg.y += tt.y; gg.z += tt.z; gg.w += tt.w; }
        v[0] *= gg.x; v[1] *= gg.y; v[2] *= gg.z; v[3] *= gg.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
      if (has_res) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
      const float a0 = act_apply8(v[0], p.act, sa.x), a1 = act_apply8(v[1], p.act, sa.y),
                  a2 = act_apply8(v[2], p.act, sa.z), a3 = act_apply8(v[3], p.act, sa.w);
      bool ok = m_ok && n < n_out;
      if (p.c_ld_rel) {
        const long erel = (long)m * p.c_ld_rel + n;
        ok = ok && erel >= p.c_lo && erel < p.c_hi;
      }
      if (ok) {
        if (p.out_f32) {
          float* frow = p.out_f32 + p.f32_off + (long)b * p.f32_bstride + (long)mc * p.f32_ld;
          *(float4*)(frow + n) = p.f32_act ? make_float4(a0, a1, a2, a3) : make_float4(v[0], v[1], v[2], v[3]);
        }
        if (p.out_act) {
          bf16_t* arow = (bf16_t*)p.out_act + p.act_off + (long)b * p.act_bstride + (long)mc * p.act_ld;
          store4<bf16_t>(arow + n, a0, a1, a2, a3);
        }
      }
    };
    if (ahead) {  // nit is 8 or 16
      issue(0, r0, g0);
      issue(1, r1, g1);
      issue(2, r2, g2);
      issue(3, r3, g3);
    }
#pragma unroll 1
    for (int it = 0; it < nit; it += 4) {
      body(it, r0, g0);
      body(it + 1, r1, g1);
      body(it + 2, r2, g2);
      body(it + 3, r3, g3);
    }
    if (half + 1 < NH) __syncthreads();  // the reads of this half precede the next half's writes
  }
}

// The LINEAR epilogue (p.flags bit 6, set by gemm8_linear_epilogue() on the host): what every Linear of the DiT needs -
// bias, adaLN gate, alpha, residual, fp32 and / or 16-bit output, SwiGLU - and nothing else (no activation, no per-channel
// period, no window mask, N a multiple of 64), straight from the accumulator layout.  Round 4, GPU call 1
// (profiles/r4_call1/ksweep.log): one 256 x 256 tile of the general epilogue above costs 24 us (plain 16-bit output) to 32 us
// (gated residual) on top of its K loop - as much as 22 - 30 K-tiles, a third of a K = 2816 launch - because the general
// contract is evaluated per element: 64-bit divisions for the gate row, 64-bit multiplies per address, seven activation
// branches, two LDS passes and two barriers per 64-row half.  Here a lane keeps the 4 consecutive columns the swapped MFMA
// gives it (16 bytes of fp32: residual / gate loads and fp32 stores are 16-byte accesses as they stand), the 16-bit
// output pairs two column blocks with v_permlane16_swap so that a lane stores 8 consecutive columns (16 bytes) - no LDS,
// no barrier, ~25 VALU instructions per 16 x 16 fragment.  Row operands are requested two row blocks ahead of their use,
// before the stores of the current block (vmcnt retires in order: a load issued behind a store waits for the store).
// Same expression order as the general epilogue, contraction off: both give the same bits.
template <int NH>
__device__ __forceinline__ void epilogue8_linear(const GemmParams& p, f32x4_t (&acc)[NH * 4][4], const int b,
                                                 const int m_wave0, const int n_wave0, const int lane) {
#pragma clang fp contract(off)
  constexpr int NI = NH * 4;
  const int lr = lane & 15, lg = lane >> 4;
  const bool out_alt = (p.flags & 512) != 0;   // mixed mode: the 16-bit output feeds a GEMM on alt-format operands
  auto pack = [&](float x, float y) { return out_alt ? pack_alt16x2(x, y) : pack_h16x2(x, y); };
  if (n_wave0 >= p.N) return;   // N % 64 == 0: a wave's 64 columns are all inside or all outside
  const long bM = (long)b * p.M;
  const int m_last = p.M - 1;
  bf16_t* const act0 = p.out_act ? (bf16_t*)p.out_act + p.act_off + (long)b * p.act_bstride : nullptr;
  if (p.swiglu) {
    // fragments (j, j | 1) = the w1 / w3 rows of 16 outputs: block jp = j >> 1 -> output columns (n_wave0 >> 1) + 16 jp
    const int c_own = (n_wave0 >> 1) + lg * 4;                              // + 16 jp: the lane's own 4 outputs
    const int c_st = (n_wave0 >> 1) + (lg & 1) * 16 + (lg >> 1) * 8;        // after the swap: 8 consecutive outputs
    (void)c_own;
    const bool split3 = (p.flags & GEMM_FLAG_OUT_SPLIT3) != 0;   // [lo | hi | hi] rows of 3 * (N / 2) elements (common.h)
    const int n_out = p.N >> 1;
#pragma unroll
    for (int I = 0; I < NI; ++I) {
      const int m = m_wave0 + I * 16 + lr;
      unsigned lo[2], hi[2], rlo[2], rhi[2];
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const f32x4_t a = acc[I][2 * jp], g = acc[I][2 * jp + 1];
        const float v0 = silu_f(a[0]) * g[0], v1 = silu_f(a[1]) * g[1], v2 = silu_f(a[2]) * g[2], v3 = silu_f(a[3]) * g[3];
        lo[jp] = pack(v0, v1);
        hi[jp] = pack(v2, v3);
        if (split3) {   // the residuals of the 16-bit rounding, rounded themselves: v = h16 + r16 to ~2^-22
          rlo[jp] = pack_h16x2(v0 - h16_lo(lo[jp]), v1 - h16_hi(lo[jp]));
          rhi[jp] = pack_h16x2(v2 - h16_lo(hi[jp]), v3 - h16_hi(hi[jp]));
        }
      }
      const auto s0 = __builtin_amdgcn_permlane16_swap(lo[0], lo[1], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(hi[0], hi[1], false, false);
      if (!split3) {
        if (m <= m_last) *(uint4*)(act0 + (long)m * p.act_ld + c_st) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        continue;
      }
      const auto t0 = __builtin_amdgcn_permlane16_swap(rlo[0], rlo[1], false, false);
      const auto t1 = __builtin_amdgcn_permlane16_swap(rhi[0], rhi[1], false, false);
      if (m <= m_last) {
        bf16_t* row = act0 + (long)m * p.act_ld + c_st;
        const uint4 h4 = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        *(uint4*)row = make_uint4(t0[0], t1[0], t0[1], t1[1]);
        *(uint4*)(row + n_out) = h4;
        *(uint4*)(row + 2 * n_out) = h4;
      }
    }
    return;
  }
  const bool has_bias = p.bias != nullptr, has_gate = p.gate != nullptr, has_tab = p.gate_tab != nullptr,
             has_res = p.res != nullptr, has_f32 = p.out_f32 != nullptr, has_act = p.out_act != nullptr;
  const int ncol = n_wave0 + lg * 4;   // + 16 j
  // column-only operand: the bias, or the gate table (never both: gemm8_linear_epilogue)
  float4 cc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    cc[j] = has_bias ? *(const float4*)(p.bias + ncol + 16 * j)
                     : (has_tab ? *(const float4*)(p.gate_tab + ncol + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f));
  const float* const res0 = has_res ? p.res + p.res_off + (long)b * p.res_bstride + ncol : nullptr;
  float* const f320 = has_f32 ? p.out_f32 + p.f32_off + (long)b * p.f32_bstride + ncol : nullptr;
  const float* const gate0 = has_gate ? p.gate + ncol : nullptr;
  const unsigned rpg = (unsigned)p.rows_per_gate;
  const int c_st = n_wave0 + (lg & 1) * 16 + (lg >> 1) * 8;   // + 32 jp: 8 consecutive columns after the swap
  // one step = one row block I x one pair of column blocks jp; the row operands of step s + 2 are requested before the
  // stores of step s
  constexpr int NS = NI * 2;
  float4 rr[2][2], gg[2][2];
  auto request = [&](const int s_, float4 (&r)[2], float4 (&g)[2]) {
    const int I = s_ >> 1, jp = s_ & 1;
    int m = m_wave0 + I * 16 + lr;
    m = m <= m_last ? m : m_last;   // rows past M: clamped loads, masked stores
    if (has_res) {
      const float* rrow = res0 + (long)m * p.res_ld + 32 * jp;
      r[0] = *(const float4*)rrow;
      r[1] = *(const float4*)(rrow + 16);
    }
    if (has_gate) {
      const float* grow = gate0 + (long)((unsigned)(bM + m) / rpg) * p.gate_ld + 32 * jp;
      g[0] = *(const float4*)grow;
      g[1] = *(const float4*)(grow + 16);
    }
  };
  request(0, rr[0], gg[0]);
  request(1, rr[1], gg[1]);
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_) {
    const int I = s_ >> 1, jp = s_ & 1;
    const int m = m_wave0 + I * 16 + lr;
    const bool m_ok = m <= m_last;
    float v[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = 2 * jp + h;
      const float4 r = rr[s_ & 1][h];
      float4 g = gg[s_ & 1][h];
      v[h][0] = acc[I][j][0]; v[h][1] = acc[I][j][1]; v[h][2] = acc[I][j][2]; v[h][3] = acc[I][j][3];
      if (has_bias) { v[h][0] += cc[j].x; v[h][1] += cc[j].y; v[h][2] += cc[j].z; v[h][3] += cc[j].w; }
      if (has_gate) {
        if (has_tab) { g.x += cc[j].x; g.y += cc[j].y; g.z += cc[j].z; g.w += cc[j].w; }
        v[h][0] *= g.x; v[h][1] *= g.y; v[h][2] *= g.z; v[h][3] *= g.w;
      }
      if (has_res) { v[h][0] += r.x; v[h][1] += r.y; v[h][2] += r.z; v[h][3] += r.w; }
    }
    if (s_ + 2 < NS) request(s_ + 2, rr[s_ & 1], gg[s_ & 1]);   // ahead of this step's stores
    if (has_f32 && m_ok) {
      float* frow = f320 + (long)m * p.f32_ld + 32 * jp;
      *(float4*)frow = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
      *(float4*)(frow + 16) = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
    }
    if (has_act) {
      const unsigned lo0 = pack(v[0][0], v[0][1]), hi0 = pack(v[0][2], v[0][3]);
      const unsigned lo1 = pack(v[1][0], v[1][1]), hi1 = pack(v[1][2], v[1][3]);
      const auto s0 = __builtin_amdgcn_permlane16_swap(lo0, lo1, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(hi0, hi1, false, false);
      if (m_ok) *(uint4*)(act0 + (long)m * p.act_ld + c_st + 32 * jp) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    }
  }
}

// The same contract as epilogue8_linear for launches WITH an fp32 output / residual (the adaLN-gated residual updates of
// the DiT: wo, w2, the folded cross-attention projection, the patcher): those move 10 bytes per element and are bound by
// memory transactions, and the accumulator layout gives a wave instruction 16 rows x 64 bytes - sixteen half lines (round 4,
// GPU call 2: 5 us per tile slower than the general epilogue).  So the tile goes through the wave's private LDS area as in
// the general epilogue - a lane then owns 4 consecutive columns and 16 lanes cover a row's 256 contiguous bytes - but with
// the lean arithmetic of the linear contract: no activation chain, 32-bit gate-row division, row pointers advanced by
// addition, operands of row block it + 4 requested ahead of the stores of block it.
template <int NH>
__device__ __forceinline__ void epilogue8_rows(const GemmParams& p, f32x4_t (&acc)[NH * 4][4], char* const stg, const int b,
                                               const int m_wave0, const int n_wave0, const int lane) {
#pragma clang fp contract(off)
  const int lr = lane & 15, lg = lane >> 4;
  const bool out_alt = (p.flags & 512) != 0;
  const long bM = (long)b * p.M;
  const bool has_bias = p.bias != nullptr, has_gate = p.gate != nullptr, has_tab = p.gate_tab != nullptr,
             has_res = p.res != nullptr, has_f32 = p.out_f32 != nullptr, has_act = p.out_act != nullptr;
  const int rsub = lane >> 4, csub = lane & 15;   // read phase: 4 rows x 16 chunks of 4 columns per wave instruction
  const int n = n_wave0 + csub * 4;
  const bool n_ok = n_wave0 < p.N;                // N % 64 == 0
  float4 cc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n_ok && has_bias) cc = *(const float4*)(p.bias + n);
  if (n_ok && has_tab) cc = *(const float4*)(p.gate_tab + n);
  const int m_last = p.M - 1;
  const unsigned rpg = (unsigned)p.rows_per_gate;
  const float* const res0 = has_res ? p.res + p.res_off + (long)b * p.res_bstride + n : nullptr;
  float* const f320 = has_f32 ? p.out_f32 + p.f32_off + (long)b * p.f32_bstride + n : nullptr;
  bf16_t* const act0 = has_act ? (bf16_t*)p.out_act + p.act_off + (long)b * p.act_bstride + n : nullptr;
  const float* const gate0 = has_gate ? p.gate + n : nullptr;
#pragma unroll
  for (int half = 0; half < NH; ++half) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 16 + lr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4_t a = acc[half * 4 + i][j];
        *(float4*)(stg + row * 256 + (((j * 4 + lg) ^ (row & 15)) << 4)) = make_float4(a[0], a[1], a[2], a[3]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging area is private to the wave: no barrier
    float4 r0, r1, r2, r3, g0, g1, g2, g3;
    auto request = [&](const int it, float4& rr, float4& gg) {
      int m = m_wave0 + half * 64 + it * 4 + rsub;
      m = m <= m_last ? m : m_last;
      if (has_res) rr = *(const float4*)(res0 + (long)m * p.res_ld);
      if (has_gate) gg = *(const float4*)(gate0 + (long)((unsigned)(bM + m) / rpg) * p.gate_ld);
    };
    auto body = [&](const int it, float4& rslot, float4& gslot) {
      const float4 rr = rslot;
      float4 gg = gslot;
      if (it + 4 < 16) request(it + 4, rslot, gslot);
      const int row = it * 4 + rsub;
      const float4 sv = *(const float4*)(stg + row * 256 + ((csub ^ (row & 15)) << 4));
      const int m = m_wave0 + half * 64 + row;
      float v0 = sv.x, v1 = sv.y, v2 = sv.z, v3 = sv.w;
      if (has_bias) { v0 += cc.x; v1 += cc.y; v2 += cc.z; v3 += cc.w; }
      if (has_gate) {
        if (has_tab) { gg.x += cc.x; gg.y += cc.y; gg.z += cc.z; gg.w += cc.w; }
        v0 *= gg.x; v1 *= gg.y; v2 *= gg.z; v3 *= gg.w;
      }
      if (has_res) { v0 += rr.x; v1 += rr.y; v2 += rr.z; v3 += rr.w; }
      if (m <= m_last) {
        if (has_f32) *(float4*)(f320 + (long)m * p.f32_ld) = make_float4(v0, v1, v2, v3);
        if (has_act) *(uint2*)(act0 + (long)m * p.act_ld) = out_alt ? make_uint2(pack_alt16x2(v0, v1), pack_alt16x2(v2, v3)) : make_uint2(pack_h16x2(v0, v1), pack_h16x2(v2, v3));
      }
    };
    if (n_ok) {
      request(0, r0, g0);
      request(1, r1, g1);
      request(2, r2, g2);
      request(3, r3, g3);
#pragma unroll
      for (int it = 0; it < 16; it += 4) {
        body(it, r0, g0);
        body(it + 1, r1, g1);
        body(it + 2, r2, g2);
        body(it + 3, r3, g3);
      }
    }
    if (half + 1 < NH) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this half's reads precede the next half's writes
  }
}

// workgroup -> tile: contiguous run of tiles per XCD, GM M-tiles x all N-tiles per group (as gemm2.hip).
// `L` = position in the linear raster order; tile_of() maps it to (batch, M-tile, N-tile).
__device__ __forceinline__ void tile_of(const GemmParams& p, const int BM, const int BN, const int L, int& b, int& tm, int& tn) {
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int per_batch = tiles_m * tiles_n;
  b = L / per_batch;
  const int l2 = L - b * per_batch;
  const int GM = p.raster_gm > 0 ? p.raster_gm : 8;
  const int per_group = GM * tiles_n;
  const int gi = l2 / per_group;
  const int first_m = gi * GM;
  const int gsz = tiles_m - first_m < GM ? tiles_m - first_m : GM;
  const int in_grp = l2 - gi * per_group;
  tm = first_m + in_grp % gsz;
  tn = in_grp / gsz;
}
// blockIdx.x -> position in a run of `total` units dealt to the 8 XCDs as contiguous sub-runs
__device__ __forceinline__ int xcd_run_pos(const int total) {
  const int bid = blockIdx.x;
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ int xcd_run_pos_of(const int total, const int bid) {
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ void tile_raster8(const GemmParams& p, const int BM, const int BN, int& b, int& tm, int& tn) {
  const int total = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.nbatch;
  tile_of(p, BM, BN, xcd_run_pos(total), b, tm, tn);
}


// W addressing (GemmParams.flags bit 11): row-major [N][K] - rows K elements apart, a K-tile 128 bytes further along the row -
// or K-TILE-MAJOR [K/64][N][64] - rows 64 elements apart inside a slab, a K-tile N * 128 bytes further.  Row-major, a launch of
// few rows fetches each K-tile of its cold weights as one 128-byte piece out of every one of N DRAM rows 2 K bytes apart, all
// workgroups at the same K offset at the same time (round 4, profiles/r4_call13 .. 17: that order of requests, not issue time or
// depth, is what a few-row launch in the model waits for); K-tile-major the same K-tile is ONE contiguous N * 128-byte run and the
// launch streams W front to back.  Same fragments in LDS either way: the same bits.
struct WGeom {
  long ns;      // elements from one W row to the next
  long kstep;   // bytes from one K-tile to the next
};
__device__ __forceinline__ WGeom w_geom(const GemmParams& p) {
  const bool ktm = (p.flags & GEMM_FLAG_W_KTM) != 0;
  return WGeom{ktm ? 64L : (long)p.K, ktm ? (long)p.N * 128 : 128L};
}
// The workgroups of a launch beyond its tiles (launches of < 256 workgroups are padded to 256 when GemmParams.pf_ptr is set): touch
// every 128-byte line of [pf_ptr, pf_ptr + pf_bytes) once, eight requests in flight per lane - a linear read that leaves the NEXT
// launch's weights in the memory-side cache (round 4, the warm-weights measurement: 46.5 -> 36.8 us per few-row launch in the model).
__device__ __forceinline__ void prefetch_lines(const GemmParams& p, const int wg, const int nwg, const int nthreads) {
  const long lines = p.pf_bytes >> 7;
  const char* const base = (const char*)p.pf_ptr;
  const long stride = (long)nwg * nthreads;
  for (long i = (long)wg * nthreads + threadIdx.x; i < lines; i += stride * 8) {
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long j = i + u * stride;
      v[u] = j < lines ? *(const volatile unsigned*)(base + (j << 7)) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
  }
}

}  // namespace

#ifdef SAMAUDIO_GEMM8_ABL
// gemm8o: the round-2 / round-3 form of the 8-phase loop (DMA through the builtin, buffer-major LDS layout), kept for the
// ablation build only (tools/build_abl.sh, tools/gemm8_ablate.py): its ablations are what located the read side as the
// longer leg (profiles/r4_call4/ablate.log).  The shipped kernel is gemm8_kernel below.
// The two wave groups run one barrier apart and every MFMA cluster runs under s_setprio 1 (round-2 A/B builds of the
// template on one box: without the stagger -11 %, without the priority -9 %; profiles/r2_call3/).
// CONV: A's k axis is split into taps (implicit convolutions: kc < K); plain GEMMs compile the per-K-tile tap walk - a
// per-lane loop under an exec mask, twice per K-tile - out of the K loop.
// (Round 3, GPU call 3: issuing the second staging instruction of every phase from inside the wave's own MFMA cluster -
// to shorten the read sections, which carry 2 global_load_lds at 100 - 185 issue cycles each - measured 4 - 7 % SLOWER on
// every DiT shape than this loop (profiles/r3_call3/gemm_bench_r3.log); removed.  Compiling the tap walk out of plain
// GEMMs measured 3 - 5 % faster and is what CONV = false is.  GPU call 10: issuing a phase's staging instructions BEFORE its
// ds_reads, and merging the four phases into two super-phases (32-MFMA clusters, 4 barriers per K-tile instead of 8), both
// measured within +-1 % of this loop on every shape (profiles/r3_call10/): neither the barrier count nor the order inside
// a read section is what bounds it.)
// ABL (only instantiated with -DSAMAUDIO_GEMM8_ABL, tools/build_abl.sh; timing experiments, wrong results): 1 = no DMA inside the K
// loop, 2 = no LDS fragment reads inside the K loop, 3 = no MFMA, 4 = no barriers, 5 = no s_setprio, 9 = correct results +
// s_memtime stamps of (entry, prologue done, K loop done, epilogue done) written per tile to p.act_alpha
template <bool CONV, int ABL = 0>
__global__ __launch_bounds__(512) void gemm8o_kernel(const GemmParams p, const int tile_count) {
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
  if constexpr (ABL == 9) ts0 = __builtin_readcyclecounter();
  constexpr bool STAGGER = true, PRIO = ABL != 5;
  constexpr int BM = 256, BN = 256, BK = 64, HT = 128 * 128;  // HT: bytes of one half-tile
  __shared__ __attribute__((aligned(16))) char smem[2 * 4 * HT];  // [K-tile buffer][HA0, HA1, HB0, HB1]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;  // waves w and w+4 share a SIMD: one of each group per SIMD
  const int lr = lane & 15, lg = lane >> 4;

  // tile_count > 0: this launch covers only the first tile_count tiles of the raster order (the full rounds of the chip);
  // the rest runs as 128x128 tiles of gemm8s_kernel (launch_gemm8_split below)
  int b, tm, tn;
  if (tile_count > 0) tile_of(p, BM, BN, xcd_run_pos(tile_count), b, tm, tn);
  else tile_raster8(p, BM, BN, b, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging bookkeeping ---------------------------------------------------------------------------------
  // A half-tile = 128 rows; wave w stages rows 16w .. 16w+15 as two wave-instructions of 8 rows (1 KiB each):
  // lane -> row 16w + 8q + (lane>>3), 16-byte slot lane&7, which must hold source chunk slot ^ ((row>>1)&7).
  const int r8 = lane >> 3;
  const bf16_t* a_row[2][2];  // [half][q] activation row pointers (row clamped to M-1), without the k offset
  const bf16_t* w_row[2][2];  // [half][q] weight row pointers incl. the lane's chunk
  int chunk[2];
  {
    const bf16_t* A = (const bf16_t*)p.A + p.a_off + (long)b * p.a_bstride;
    const bf16_t* W = (const bf16_t*)p.W + (long)b * p.w_bstride;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = wave * 16 + q * 8 + r8;  // row inside a half-tile
      chunk[q] = (lane & 7) ^ ((row >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int m = m0 + h * 128 + row;
        m = m < p.M ? m : p.M - 1;
        a_row[h][q] = A + (long)m * p.lda;
        int n = n0 + h * 128 + row;
        n = n < p.N ? n : p.N - 1;
        w_row[h][q] = W + (long)n * p.K + chunk[q] * 8;
      }
    }
  }
  // position of the lane's chunk in the (tap, offset) structure of A's k axis, per q, for the K-tile being staged.
  // HA0 and HA1 of a K-tile are staged in consecutive phases and share it; it advances once both are out.
  int a_in[2];
  long a_tap[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    a_in[q] = chunk[q] * 8;
    a_tap[q] = 0;
    if constexpr (CONV)
      while (a_in[q] >= p.kc) { a_in[q] -= p.kc; a_tap[q] += p.tap_stride; }
  }
  const int nt = p.K / BK;
  auto stage_a = [&](int h, int buf) {  // HA_h of the K-tile the a_in / a_tap state points at
    char* dst = smem + buf * (4 * HT) + h * HT + wave * 2048;
#pragma unroll
    for (int q = 0; q < 2; ++q) dma16_8(a_row[h][q] + a_tap[q] + a_in[q], dst + q * 1024);
  };
  auto advance_a = [&]() {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      a_in[q] += BK;
      if constexpr (CONV)
        while (a_in[q] >= p.kc) { a_in[q] -= p.kc; a_tap[q] += p.tap_stride; }
    }
  };
  auto stage_w = [&](int h, int buf, int kt) {  // HB_h of K-tile kt
    char* dst = smem + buf * (4 * HT) + (2 + h) * HT + wave * 2048;
#pragma unroll
    for (int q = 0; q < 2; ++q) dma16_8(w_row[h][q] + (long)kt * BK, dst + q * 1024);
  };

  // ---- fragments --------------------------------------------------------------------------------------------
  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[4][2];      // activation fragments of the current 64-row half: [m-fragment][k-step]
  bf16x8_t wf[2][2][2];   // weight fragments: [32-column half][n-fragment][k-step]

  auto frag = [&](const char* half_base, int row, int ks) -> bf16x8_t {
    return *(const bf16x8_t*)(half_base + row * 128 + ((((ks << 2) + lg) ^ ((row >> 1) & 7)) << 4));
  };
  auto read_a = [&](int buf, int sub) {  // rows 64*sub .. +63 of the wave's half-tile HA_wr
    const char* base = smem + buf * (4 * HT) + wr * HT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = frag(base, sub * 64 + i * 16 + lr, ks);
  };
  // weight rows (output columns) 64*(wc&1) + 32*SUB .. +31 of HB_(wc>>1); SUB compile-time (static register index)
#define SA_GEMM8_READ_W(BUF, SUB)                                                                                 \
  do {                                                                                                            \
    const char* base_ = smem + (BUF) * (4 * HT) + (2 + (wc >> 1)) * HT;                                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
        wf[SUB][j][ks] = frag(base_, (wc & 1) * 64 + (SUB) * 32 + j * 16 + lr, ks);                               \
  } while (0)
  // one C quadrant x K = 64: 16 MFMAs.  ASUB / WSUB are compile-time so that acc[][] is indexed statically and stays in
  // registers (a run-time quadrant index sends the whole accumulator to scratch).
#define SA_GEMM8_MMA(ASUB, WSUB)                                                                                  \
  do {                                                                                                            \
    if constexpr (ABL == 3) {                                                                                     \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(wf[WSUB][j][ks]));                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(af[i][ks]));                          \
      }                                                                                                           \
      break;                                                                                                      \
    }                                                                                                             \
    if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
          acc[(ASUB) * 4 + i][(WSUB) * 2 + j] = SA_MFMA_16x16x32(                          \
              wf[WSUB][j][ks], af[i][ks], acc[(ASUB) * 4 + i][(WSUB) * 2 + j]);                          \
    if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                      \
  } while (0)

#define SA_BAR()                                   \
  do {                                             \
    if constexpr (ABL != 4) __builtin_amdgcn_s_barrier(); \
  } while (0)
  // ---- prologue: K-tile 0 complete, HB0 / HB1 of K-tile 1 in flight (what the steady state expects) ----------
  stage_w(0, 0, 0);
  stage_w(1, 0, 0);
  stage_a(0, 0);
  stage_a(1, 0);
  advance_a();
  if (nt > 1) {
    stage_w(0, 1, 1);
    stage_w(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

  if constexpr (ABL == 2) {
    SA_GEMM8_READ_W(0, 0);
    SA_GEMM8_READ_W(0, 1);
    read_a(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if constexpr (ABL == 9) ts1 = __builtin_readcyclecounter();
  for (int t = 0; t < nt; ++t) {
    const int cb = t & 1, nb = cb ^ 1;
    const bool s1 = t + 1 < nt, s2 = t + 2 < nt;
    // P1
    if constexpr (ABL != 2) SA_GEMM8_READ_W(cb, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABL != 2) read_a(cb, 0);
    if (ABL != 1 && s1) stage_a(0, nb);
    SA_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SA_GEMM8_MMA(0, 0);
    SA_BAR();
    // P2
    if constexpr (ABL != 2) SA_GEMM8_READ_W(cb, 1);
    if (ABL != 1 && s1) { stage_a(1, nb); advance_a(); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // HB(t) is restaged in the next phase: its reads end here
    SA_BAR();
    SA_GEMM8_MMA(0, 1);
    SA_BAR();
    // P3
    if constexpr (ABL != 2) read_a(cb, 1);
    if (ABL != 1 && s2) stage_w(0, cb, t + 2);
    SA_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SA_GEMM8_MMA(1, 1);
    SA_BAR();
    // P4
    if (ABL == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (s2) {
      stage_w(1, cb, t + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // HB0 / HB1 of t+2 stay in flight; K-tile t+1 has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SA_BAR();
    SA_GEMM8_MMA(1, 0);
    SA_BAR();
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();  // every wave passes the same number of barriers
#undef SA_GEMM8_MMA
#undef SA_GEMM8_READ_W
#undef SA_BAR
  if constexpr (ABL == 9) ts2 = __builtin_readcyclecounter();

  // ---- epilogue (contract of GemmParams, common.h): shared with gemm8s_kernel below ----------------------------
  if (p.flags & 64) {   // the linear epilogue needs no LDS: no barrier either
    epilogue8_linear<2>(p, acc, b, m0 + wr * 128, n0 + wc * 64, lane);
  } else {
    __syncthreads();
    if (p.flags & 128) epilogue8_rows<2>(p, acc, smem + wave * 16384, b, m0 + wr * 128, n0 + wc * 64, lane);
    else epilogue8<2>(p, acc, smem + wave * 16384, b, m0 + wr * 128, n0 + wc * 64, lane);
  }
  if constexpr (ABL == 9) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the wave
    const unsigned long long ts3 = __builtin_readcyclecounter();
    if (lane == 0) {
      unsigned long long* o = (unsigned long long*)p.act_alpha + ((size_t)blockIdx.x * 8 + wave) * 4;
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
    }
  }
}

#endif  // SAMAUDIO_GEMM8_ABL

// gemm8_kernel (round 4): the 8-phase loop above with the DMA issued as inline assembly.  GPU call 4
// (profiles/r4_call4/ablate.log): a K-tile of the round-3 loop took 2 533 cycles for 2 048 cycles of MFMA work; without its
// DMA instructions 2 105, without its LDS fragment reads 2 006, without barriers / priorities no less - the read side of the
// heavy phases was the longer leg, and the ISA showed why: hipcc treats a global_load_lds it can see as a "flat" access
// pending on both counters, so every MFMA cluster began with s_waitcnt lgkmcnt(0) - all 12 fragment reads of a phase
// landed before its first MFMA issued.  With the DMA invisible to it (dma16s / dma16v) the compiler counts the reads itself
// (lgkmcnt(9), (8), ... in front of the MFMAs that need them): the fragments of k-step 1 land underneath the MFMAs of
// k-step 0.  Reads are issued k-step-major for that; the steady-state loop is unrolled by two K-tiles and free of branches
// (buffer offsets become immediates; the LDS layout is half-tile-major so that both buffers are within the 64 KiB
// immediate range of one lane base address); the last one to three K-tiles run a copy with the end-of-K conditions.
// Plain operands are addressed as (uniform base advancing 128 bytes per K-tile) + (32-bit lane offset): no vector address
// arithmetic in the loop at all - its VALU content is the 128 MFMAs.  GPU call 5 (profiles/r4_call5/): 1.091 -> 0.999 us
// per K-tile at 176 tiles, 1.398 -> 1.311 at 256; 4096^3 1 101 -> 1 388 TF/s (the guide's template: 1 320 - 1 340);
// end to end 234.5 -> 240.7 s-audio/s.  Also measured there and dropped: reading the next K-tile's first W fragments in P4
// (8 / 4 / 8 / 4 reads per phase instead of 12 / 4 / 8 / 0; +-0.5 %), and in call 1 a deeper staging pipeline (five
// half-tiles in flight instead of two: 1 - 3 % slower - load latency was never the limiter).
// Same tile, MFMA order and epilogues as gemm8o / gemm8s: the same bits.
// ALT: the operands are in the alt 16-bit format (mixed mode: bf16 inside the fp16 build) - the MFMA opcode is the only difference
template <bool CONV, bool ALT = false>
__global__ __launch_bounds__(512) void gemm8_kernel(const GemmParams p, const int tile_count) {
  constexpr int BM = 256, BN = 256, BK = 64, HT = 128 * 128;
  // [HA0, HA1, HB0, HB1][K-tile buffer]: the two buffers of a half-tile are 16 KiB apart, so that every fragment read of a wave
  // is one of four lane base addresses + an immediate offset (a buffer-major layout puts buffer 1 beyond the 64 KiB
  // immediate range: eight more address registers in the loop unrolled by two)
  __shared__ __attribute__((aligned(16))) char smem[4 * 2 * HT];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;

  // PERSISTENT (grid < tiles: launch_gemm8_tiles): a workgroup walks the tiles vb = blockIdx.x, + gridDim.x, ...
  // of the XCD-contiguous raster (gridDim.x is a multiple of 8, so every tile of a workgroup lies in its XCD's run).  The
  // 16-bit epilogue's stores are not waited for: they drain underneath the next tile's prologue loads, which in turn are in
  // flight while the stores are issued.
  const int total = tile_count > 0 ? tile_count : ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.nbatch;
  const WGeom wg_ = w_geom(p);
  if (p.pf_ptr && (int)blockIdx.x >= total) {   // a workgroup beyond the tiles: warm the next launch's weights
    prefetch_lines(p, (int)blockIdx.x - total, (int)gridDim.x - total, 512);
    return;
  }
  for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
  int b, tm, tn;
  tile_of(p, BM, BN, xcd_run_pos_of(total, vb), b, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging: plain GEMMs address a row as (uniform base advancing 128 bytes per K-tile) + (per-lane 32-bit byte offset)
  // (launch_gemm8_tiles checks that every offset fits 32 bits); implicit convolutions keep per-lane 64-bit pointers and the
  // tap walk of gemm8_kernel
  const int r8 = lane >> 3;
  const bf16_t* a_row[2][2];
  const bf16_t* w_row[2][2];
  unsigned a_off[2][2], w_off[2][2];
  int chunk[2];
  const bf16_t* const A0 = (const bf16_t*)p.A + p.a_off + (long)b * p.a_bstride;
  const bf16_t* const W0 = (const bf16_t*)p.W + (long)b * p.w_bstride;
  {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = wave * 16 + q * 8 + r8;
      chunk[q] = (lane & 7) ^ ((row >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int m = m0 + h * 128 + row;
        m = m < p.M ? m : p.M - 1;
        a_row[h][q] = A0 + (long)m * p.lda;
        a_off[h][q] = (unsigned)(((long)m * p.lda + chunk[q] * 8) * 2);
        int n = n0 + h * 128 + row;
        n = n < p.N ? n : p.N - 1;
        w_row[h][q] = W0 + (long)n * wg_.ns + chunk[q] * 8;
        w_off[h][q] = (unsigned)(((long)n * wg_.ns + chunk[q] * 8) * 2);
      }
    }
  }
  int a_in[2];
  long a_tap[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    a_in[q] = chunk[q] * 8;
    a_tap[q] = 0;
    if constexpr (CONV)
      while (a_in[q] >= p.kc) { a_in[q] -= p.kc; a_tap[q] += p.tap_stride; }
  }
  const int nt = p.K / BK;
  const char* a_base = (const char*)A0;   // the K-tile the next stage_a() stages
  const size_t lds0 = (size_t)(__attribute__((address_space(3))) char*)smem;   // LDS address of the tile buffers
  auto stage_a = [&](int h, int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const size_t dst = lds0 + (size_t)((h * 2 + buf) * HT + wave * 2048 + q * 1024);
      if constexpr (CONV) dma16v(a_row[h][q] + a_tap[q] + a_in[q], dst);
      else dma16s(a_base, a_off[h][q], dst);
    }
  };
  auto advance_a = [&]() {
    a_base += BK * 2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      a_in[q] += BK;
      if constexpr (CONV)
        while (a_in[q] >= p.kc) { a_in[q] -= p.kc; a_tap[q] += p.tap_stride; }
    }
  };
  auto stage_w = [&](int h, int buf, int kt) {
    const char* w_base = (const char*)W0 + (long)kt * wg_.kstep;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const size_t dst = lds0 + (size_t)(((2 + h) * 2 + buf) * HT + wave * 2048 + q * 1024);
      if constexpr (CONV) dma16v((const char*)w_row[h][q] + (long)kt * wg_.kstep, dst);
      else dma16s(w_base, w_off[h][q], dst);
    }
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[4][2];      // [m-fragment][k-step]
  bf16x8_t wf[2][2][2];   // [register set][n-fragment][k-step]

  auto frag = [&](const char* half_base, int row, int ks) -> bf16x8_t {
    return *(const bf16x8_t*)(half_base + row * 128 + ((((ks << 2) + lg) ^ ((row >> 1) & 7)) << 4));
  };
  // fragment reads of ONE k-step (KS compile-time): A rows 64*sub .. +63 of HA_wr; W rows of the 32-column half SUB
#define SA_G8P_READ_A(BUF, ASUB, KS)                                                                              \
  do {                                                                                                            \
    const char* base_ = smem + (wr * 2 + (BUF)) * HT;                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i][KS] = frag(base_, (ASUB) * 64 + i * 16 + lr, KS);        \
  } while (0)
#define SA_G8P_READ_W(BUF, SUB, SET, KS)                                                                          \
  do {                                                                                                            \
    const char* base_ = smem + ((2 + (wc >> 1)) * 2 + (BUF)) * HT;                                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
      wf[SET][j][KS] = frag(base_, (wc & 1) * 64 + (SUB) * 32 + j * 16 + lr, KS);                                 \
  } while (0)
  // the 8 MFMAs of one k-step of a C quadrant (same order as gemm8_kernel: ks outer, j, i)
#define SA_G8P_MMA(ASUB, WSUB, SET, KS)                                                                           \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
        acc[(ASUB) * 4 + i][(WSUB) * 2 + j] =                                                                     \
            ALT ? SA_MFMA_16x16x32_ALT(wf[SET][j][KS], af[i][KS], acc[(ASUB) * 4 + i][(WSUB) * 2 + j])            \
                : SA_MFMA_16x16x32(wf[SET][j][KS], af[i][KS], acc[(ASUB) * 4 + i][(WSUB) * 2 + j]);              \
  } while (0)
  // one K-tile; CB = its buffer (compile-time), S0 / S1 = the register sets of Bs0 / Bs1 for this K-tile.  STEADY: K-tiles
  // t+1 and t+2 exist - no branch between the fragment reads and the MFMAs (at a control-flow merge hipcc waits for EVERY
  // outstanding LDS read before the first MFMA; without one it counts them itself)
#define SA_G8P_TILE(CB, S0, S1, STEADY)                                                                                  \
  do {                                                                                                            \
    constexpr int NB = (CB) ^ 1;                                                                                  \
    const bool s1 = (STEADY) || t + 1 < nt, s2 = (STEADY) || t + 2 < nt;                                                                \
    /* P1: (Bs0,) As0 */                                                                                          \
    SA_G8P_READ_W(CB, 0, S0, 0);                                                                               \
    SA_G8P_READ_A(CB, 0, 0);                                                                                      \
    SA_G8P_READ_W(CB, 0, S0, 1);                                                                               \
    SA_G8P_READ_A(CB, 0, 1);                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    if (s1) stage_a(0, NB);                                                                                       \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8P_MMA(0, 0, S0, 0);                                                                                      \
    SA_G8P_MMA(0, 0, S0, 1);                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
    /* P2: Bs1 (HB(t) is restaged in P3: its reads end before the barrier) */                                     \
    SA_G8P_READ_W(CB, 1, S1, 0);                                                                                  \
    SA_G8P_READ_W(CB, 1, S1, 1);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    if (s1) { stage_a(1, NB); advance_a(); }                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8P_MMA(0, 1, S1, 0);                                                                                      \
    SA_G8P_MMA(0, 1, S1, 1);                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
    /* P3: As1 */                                                                                                 \
    SA_G8P_READ_A(CB, 1, 0);                                                                                      \
    SA_G8P_READ_A(CB, 1, 1);                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    if (s2) stage_w(0, CB, t + 2);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8P_MMA(1, 1, S1, 0);                                                                                      \
    SA_G8P_MMA(1, 1, S1, 1);                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
    /* P4: no reads */                                                                                            \
    if (s2) {                                                                                                     \
      stage_w(1, CB, t + 2);                                                                                      \
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                            \
    } else {                                                                                                      \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
    }                                                                                                             \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8P_MMA(1, 0, S0, 0);                                                                                      \
    SA_G8P_MMA(1, 0, S0, 1);                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
  } while (0)

  // ---- prologue (as gemm8_kernel) ---------------------------------------------------------------------------------
  stage_w(0, 0, 0);
  stage_w(1, 0, 0);
  stage_a(0, 0);
  stage_a(1, 0);
  advance_a();
  if (nt > 1) {
    stage_w(0, 1, 1);
    stage_w(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0), as the builtin: hipcc then knows that nothing is pending at the loop entry

  int t = 0;
  for (; t + 3 < nt; t += 2) {   // both K-tiles of a trip have two successors
    SA_G8P_TILE(0, 0, 1, true);
    ++t;
    SA_G8P_TILE(1, 0, 1, true);
    --t;
  }
  // the last one to three K-tiles (t is even here): straight-line code, so that the roles of the W register sets stay static
  SA_G8P_TILE(0, 0, 1, false);
  ++t;
  if (t < nt) {
    SA_G8P_TILE(1, 0, 1, false);
    ++t;
    if (t < nt) SA_G8P_TILE(0, 0, 1, false);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef SA_G8P_TILE
#undef SA_G8P_MMA
#undef SA_G8P_READ_W
#undef SA_G8P_READ_A

  if (p.flags & 64) {
    epilogue8_linear<2>(p, acc, b, m0 + wr * 128, n0 + wc * 64, lane);
  } else {
    __syncthreads();
    if (p.flags & 128) epilogue8_rows<2>(p, acc, smem + wave * 16384, b, m0 + wr * 128, n0 + wc * 64, lane);
    else epilogue8<2>(p, acc, smem + wave * 16384, b, m0 + wr * 128, n0 + wc * 64, lane);
    if (vb + (int)gridDim.x < total) __syncthreads();   // the staging area is the next tile's K-tile buffers
  }
  }   // tiles of this workgroup
}

// gemm8x_kernel (round 6): gemm8_kernel for the K-CONCATENATED operands of the compensated mode (GemmParams.flags bit 15,
// GEMM_FLAG_X3_SHARE: A rows [x_lo | x_hi | x_hi], W rows [W_hi | W_lo | W_hi], K' = 3K - common.h), aware that a third of what the plain
// kernel stages is a copy.  Per 64 original k the plain kernel walks three K-tiles far apart in K' - (x_lo, W_hi), (x_hi, W_lo),
// (x_hi, W_hi) - and stages six operand tiles; here the three products of one original k-tile run back to back in the order
//     (x_lo, W_hi)   (x_hi, W_hi)   (x_hi, W_lo)
// so that consecutive products share an operand tile that is already in LDS: FOUR staged tiles for the same 192 MFMAs per wave (a
// third less DMA, LDS-write and L2 / HBM traffic; the third copies of both operands are never read).  The buffers have fixed roles -
// A: x_lo in buffer 0, x_hi in buffer 1; W: W_hi in buffer 0, W_lo in buffer 1 - and every restaging keeps the distance to the last
// read of its buffer that gemm8_kernel has (A: restaged in P1 / P2 of the tile after its last reader; W: in P3 / P4, behind the P2
// reads of its last reader or later):
//     product 0 of t:  reads A0 W0   stages x_hi(t) -> A1 (P1, P2), W_lo(t) -> W1 (P3, P4)        vmcnt(4): x_hi(t) landed
//     product 1 of t:  reads A1 W0   stages x_lo(t+1) -> A0,        W_hi(t+1) -> W0                vmcnt(8): W_lo(t) landed
//     product 2 of t:  reads A1 W1   stages nothing                                                vmcnt(0): x_lo / W_hi(t+1) landed
// Same phases, barriers, stagger, priorities and epilogues as gemm8_kernel.  The accumulation order of an output element is
// lo.hi, hi.hi, hi.lo per original k-tile; gemm8s_kernel walks the same order for such launches (its K-tile index map), so the tile
// policy may still pick by row count without changing a bit.  Plain operands within 32-bit offsets only (no implicit convolutions).
// Fragments two consecutive products have in common stay in registers (A/B against reading every fragment of every product from LDS:
// +2-3 % per launch, profiles/r6_call22).  A two-phase form of the products (32 MFMAs per wave and phase, half the barriers) was built
// and measured in round 6: no gain on the timed step (profiles/r6_call23); so was requesting x_hi a phase and a half earlier (4 - 5
// phases in front of its vmcnt(4) instead of 2 - 3; profiles/r6_call24: no gain) - the loop does not wait for its loads - and issuing the
// last 2 / 4 MFMAs of a phase behind its closing barrier (the wave arrives early, the other group's MFMAs queue behind its own:
// profiles/r6_call25, 5 % SLOWER - the strict alternation of the two groups is what the loop lives on).
__global__ __launch_bounds__(512) void gemm8x_kernel(const GemmParams p, const int tile_count) {
  constexpr int BM = 256, BN = 256, HT = 128 * 128;
  __shared__ __attribute__((aligned(16))) char smem[4 * 2 * HT];   // [HA0, HA1, HB0, HB1][buffer] as gemm8_kernel

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;
  const int total = tile_count > 0 ? tile_count : ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.nbatch;
  const WGeom wg_ = w_geom(p);
  if (p.pf_ptr && (int)blockIdx.x >= total) {
    prefetch_lines(p, (int)blockIdx.x - total, (int)gridDim.x - total, 512);
    return;
  }
  const int T3 = p.K / 192;   // original K-tiles: K' = 3K, 64 k each
  for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
  int b, tm, tn;
  tile_of(p, BM, BN, xcd_run_pos_of(total, vb), b, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int r8 = lane >> 3;
  unsigned a_off[2][2], w_off[2][2];
  const bf16_t* const A0 = (const bf16_t*)p.A + p.a_off + (long)b * p.a_bstride;
  const bf16_t* const W0 = (const bf16_t*)p.W + (long)b * p.w_bstride;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = wave * 16 + q * 8 + r8;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int m = m0 + h * 128 + row;
      m = m < p.M ? m : p.M - 1;
      a_off[h][q] = (unsigned)(((long)m * p.lda + chunk * 8) * 2);
      int n = n0 + h * 128 + row;
      n = n < p.N ? n : p.N - 1;
      w_off[h][q] = (unsigned)(((long)n * wg_.ns + chunk * 8) * 2);
    }
  }
  const size_t lds0 = (size_t)(__attribute__((address_space(3))) char*)smem;
  // K-tile `kt` of K' (64 elements = 128 bytes of an A row; wg_.kstep bytes of W) -> half-tile h of buffer `buf`
  auto stage_a = [&](int h, int buf, int kt) {
    const char* a_base = (const char*)A0 + (long)kt * 128;
#pragma unroll
    for (int q = 0; q < 2; ++q) dma16s(a_base, a_off[h][q], lds0 + (size_t)((h * 2 + buf) * HT + wave * 2048 + q * 1024));
  };
  auto stage_w = [&](int h, int buf, int kt) {
    const char* w_base = (const char*)W0 + (long)kt * wg_.kstep;
#pragma unroll
    for (int q = 0; q < 2; ++q) dma16s(w_base, w_off[h][q], lds0 + (size_t)(((2 + h) * 2 + buf) * HT + wave * 2048 + q * 1024));
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[4][2];
  bf16x8_t wf[2][2][2];
  auto frag = [&](const char* half_base, int row, int ks) -> bf16x8_t {
    return *(const bf16x8_t*)(half_base + row * 128 + ((((ks << 2) + lg) ^ ((row >> 1) & 7)) << 4));
  };
#define SA_G8X_READ_A(BUF, ASUB, KS)                                                                              \
  do {                                                                                                            \
    const char* base_ = smem + (wr * 2 + (BUF)) * HT;                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i][KS] = frag(base_, (ASUB) * 64 + i * 16 + lr, KS);        \
  } while (0)
#define SA_G8X_READ_W(BUF, SUB, SET, KS)                                                                          \
  do {                                                                                                            \
    const char* base_ = smem + ((2 + (wc >> 1)) * 2 + (BUF)) * HT;                                                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
      wf[SET][j][KS] = frag(base_, (wc & 1) * 64 + (SUB) * 32 + j * 16 + lr, KS);                                 \
  } while (0)
#define SA_G8X_MMA(ASUB, WSUB, SET, KS)                                                                           \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
        acc[(ASUB) * 4 + i][(WSUB) * 2 + j] =                                                                     \
            SA_MFMA_16x16x32(wf[SET][j][KS], af[i][KS], acc[(ASUB) * 4 + i][(WSUB) * 2 + j]);                     \
  } while (0)
  // one product (a K-tile of 64): reads A buffer AB / W buffer WB; SA / SW: stage K-tile KA of A into buffer AD (P1, P2) / K-tile KW
  // of W into buffer WD (P3, P4); WAITS = the s_waitcnt of P4 (text: the simulator reads the count).  RW false: the W fragments of
  // the previous product are still in wf[][][] (both column halves, both k-steps) and are not read again.
#define SA_G8X_TILE(AB, WB, RW, SA, AD, KA, SW, WD, KW, WAITS)                                                    \
  do {                                                                                                            \
    if (RW) SA_G8X_READ_W(WB, 0, 0, 0);                                                                           \
    SA_G8X_READ_A(AB, 0, 0);                                                                                      \
    if (RW) SA_G8X_READ_W(WB, 0, 0, 1);                                                                           \
    SA_G8X_READ_A(AB, 0, 1);                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    if (SA) stage_a(0, AD, KA);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8X_MMA(0, 0, 0, 0);                                                                                       \
    SA_G8X_MMA(0, 0, 0, 1);                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
    if (RW) {                                                                                                     \
      SA_G8X_READ_W(WB, 1, 1, 0);                                                                                 \
      SA_G8X_READ_W(WB, 1, 1, 1);                                                                                 \
    }                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    if (SA) stage_a(1, AD, KA);                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8X_MMA(0, 1, 1, 0);                                                                                       \
    SA_G8X_MMA(0, 1, 1, 1);                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
    SA_G8X_READ_A(AB, 1, 0);                                                                                      \
    SA_G8X_READ_A(AB, 1, 1);                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    if (SW) stage_w(0, WD, KW);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8X_MMA(1, 1, 1, 0);                                                                                       \
    SA_G8X_MMA(1, 1, 1, 1);                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
    if (SW) stage_w(1, WD, KW);                                                                                   \
    asm volatile(WAITS ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8X_MMA(1, 0, 0, 0);                                                                                       \
    SA_G8X_MMA(1, 0, 0, 1);                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
  } while (0)
  // the product behind one that read the same A buffer: af[][] still holds that buffer's LOWER 64 rows of this wave (read in P3), so
  // the quadrants run bottom first - (1,0) (1,1) (0,1) (0,0) - and only the upper rows are read again; all of W is new.  Stages
  // nothing.  An output element belongs to one quadrant: its accumulation order does not depend on the order of the quadrants.
#define SA_G8X_TILE_REV(AB, WB, WAITS)                                                                            \
  do {                                                                                                            \
    SA_G8X_READ_W(WB, 0, 0, 0);                                                                                   \
    SA_G8X_READ_W(WB, 0, 0, 1);                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8X_MMA(1, 0, 0, 0);                                                                                       \
    SA_G8X_MMA(1, 0, 0, 1);                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
    SA_G8X_READ_W(WB, 1, 1, 0);                                                                                   \
    SA_G8X_READ_W(WB, 1, 1, 1);                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8X_MMA(1, 1, 1, 0);                                                                                       \
    SA_G8X_MMA(1, 1, 1, 1);                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
    SA_G8X_READ_A(AB, 0, 0);                                                                                      \
    SA_G8X_READ_A(AB, 0, 1);                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8X_MMA(0, 1, 1, 0);                                                                                       \
    SA_G8X_MMA(0, 1, 1, 1);                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
    asm volatile(WAITS ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    SA_G8X_MMA(0, 0, 0, 0);                                                                                       \
    SA_G8X_MMA(0, 0, 0, 1);                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                 \
  } while (0)

  // prologue: x_lo(0) and W_hi(0) complete
  stage_w(0, 0, 0);
  stage_w(1, 0, 0);
  stage_a(0, 0, 0);
  stage_a(1, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0
  __builtin_amdgcn_s_waitcnt(0xC07F);
  int t = 0;
  for (; t + 1 < T3; ++t) {   // the next original K-tile exists: no branch between fragment reads and MFMAs
    SA_G8X_TILE(0, 0, true, true, 1, T3 + t, true, 1, T3 + t, "s_waitcnt vmcnt(4)");
    SA_G8X_TILE(1, 0, false, true, 0, t + 1, true, 0, t + 1, "s_waitcnt vmcnt(8)");
    SA_G8X_TILE_REV(1, 1, "s_waitcnt vmcnt(0)");
  }
  SA_G8X_TILE(0, 0, true, true, 1, T3 + t, true, 1, T3 + t, "s_waitcnt vmcnt(4)");
  SA_G8X_TILE(1, 0, false, false, 0, 0, false, 0, 0, "s_waitcnt vmcnt(0)");
  SA_G8X_TILE_REV(1, 1, "s_waitcnt vmcnt(0)");
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef SA_G8X_TILE_REV
#undef SA_G8X_TILE
#undef SA_G8X_MMA
#undef SA_G8X_READ_W
#undef SA_G8X_READ_A

  if (p.flags & 64) {
    epilogue8_linear<2>(p, acc, b, m0 + wr * 128, n0 + wc * 64, lane);
  } else {
    __syncthreads();
    if (p.flags & 128) epilogue8_rows<2>(p, acc, smem + wave * 16384, b, m0 + wr * 128, n0 + wc * 64, lane);
    else epilogue8<2>(p, acc, smem + wave * 16384, b, m0 + wr * 128, n0 + wc * 64, lane);
    if (vb + (int)gridDim.x < total) __syncthreads();
  }
  }   // tiles of this workgroup
}

// gemm8s: the SAME arithmetic as gemm8_kernel on a 128 x 128 tile - 16x16x32 MFMA with swapped operands, identical
// fragment <-> k mapping, K walked in slabs of 64 with the two k-steps of a slab in the same order - so an output element
// is accumulated bit for bit as the 256 x 256 kernel accumulates it.  The tile policy (gemm.hip gemm_variant) may
// therefore switch between the two with the number of rows of a launch (few rows: strong scaling at 4 clips per GPU,
// the Judge's and the vision tower's small batches) without breaking batch-sharding invariance.
// 4 waves as 2 (M) x 2 (N), 64 x 64 outputs each; two 32 KiB stages (A tile | W tile) = 64 KiB of LDS and 256 threads, so
// two workgroups share a CU and one's barriers / epilogue are covered by the other's K loop; a plain double buffer: the
// DMA of K-tile t+1 is issued before the reads of K-tile t, one counted vmcnt and two barriers per K-tile.
// PIPE: the form for launches that cannot give a CU a second workgroup (<= 256 workgroups: 176 at M = 1000, N = D).  Alone
// on its CU the double-buffered loop runs read fragments -> barrier -> MFMA strictly in sequence - one wave per SIMD, nothing
// to overlap with: ~1 200 cycles per K-tile for 512 cycles of MFMA work.  PIPE keeps a 4-stage ring (128 KiB) and two
// fragment sets: the LDS reads of K-tile t+1 are issued BEFORE the MFMAs of K-tile t and complete underneath them, one
// barrier per K-tile.  Same MFMA order per output element: bitwise identical to the plain form and to gemm8_kernel.
// CONV as in gemm8_kernel: plain GEMMs (kc == K) compile the tap walk out of the staging step.
// The pipelined form's ring has 4 stages (128 KiB: as deep as LDS allows) = 3 K-tiles of L2 latency in flight: a launch of
// <= 256 workgroups lasts nt x (a per-K-tile time set by that depth) whatever its workgroup count (round 3, GPU call 8).
// 3 -> 4 stages: c_wq at 1000 rows 36.9 -> 34.9 us, w2 83.4 -> 79.0; 4 clips per GPU 114.5 -> 119.8 s-audio/s, small* 8 clips
// 424.0 -> 435.3 (profiles/r3_call9/).  A 5-stage ring (160 KiB, all of the CU's LDS) measured slower again: c_wq 35.9 vs 34.9 us, w2
// 80.5 vs 76.2, 4 clips 120.7 vs 121.5 (profiles/r3_call28/) - three K-tiles in flight already cover the latency.
// PROD >= 0 (pipelined form only): WAVE ROLES.  Alone on its CU the pipelined form has one wave per SIMD, and a wave that issues a
// K-tile's 8 direct-to-LDS loads holds its instruction stream for ~100 cycles each (the issue cost the 4-wave 256 x 256 experiment
// of round 3 ran into): ~800 cycles of load issue in front of 512 cycles of MFMAs per K-tile, in series - the measured ~1 400 cycles
// per K-tile (0.58 - 0.61 us: c_wq at 1 000 rows 27 us for 44 K-tiles).  With roles the workgroup has 8 waves: waves 0 - 3 multiply
// exactly as before (64 x 64 outputs each, same fragments, same MFMA order: bitwise identical), waves 4 - 7 - one per SIMD, beside a
// multiplying wave - only request K-tiles: their issue time runs underneath the other wave's MFMAs.  PROD = how many of its row
// block's 4 A-tile loads per K-tile a multiplying wave still issues itself (0: none; 2 balances 6 + 2 when the requesting waves are
// the longer side).  One barrier per K-tile as before; the requesting waves leave at the last barrier.
template <int N> __device__ __forceinline__ void wait_vm_lit() {   // literal counts: the simulator reads the number from the text
  static_assert(N == 0 || N == 1 || N == 2 || N == 4 || N == 6 || N == 7 || N == 8 || N == 12 || N == 14 || N == 16, "add the literal");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}
template <bool PIPE, bool CONV, bool ALT = false, int PROD = -1>
__global__ __launch_bounds__(PROD >= 0 ? 512 : 256) void gemm8s_kernel(const GemmParams p, const int skip256) {
  constexpr int BM = 128, BN = 128, BK = 64, TB = 128 * 128;  // TB: bytes of one operand tile (128 rows x 128 B)
  constexpr int S = PIPE ? 4 : 2;
  constexpr bool ROLES = PROD >= 0;
  static_assert(!ROLES || PIPE, "wave roles belong to the pipelined form");
  __shared__ __attribute__((aligned(16))) char smem[S * 2 * TB];  // [stage][A tile, W tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave_id & 3;   // ROLES: waves 4 .. 7 request the row blocks waves 0 .. 3 own
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;

  // skip256 >= 0 ("tail" mode): the first skip256 tiles of the 256x256 raster order were computed by gemm8_kernel; this
  // launch covers the remaining ones as 4 quadrants each - consecutive workgroups of an XCD's run share a 256-tile's
  // operand panels.  Same arithmetic either way, so the split is invisible in the results.
  int b, tm, tn, m0, n0;
  if (skip256 >= 0) {
    const int total256 = ((p.M + 255) / 256) * ((p.N + 255) / 256) * p.nbatch;
    const int pos = xcd_run_pos((total256 - skip256) * 4);
    tile_of(p, 256, 256, skip256 + (pos >> 2), b, tm, tn);
    m0 = tm * 256 + ((pos >> 1) & 1) * 128;
    n0 = tn * 256 + (pos & 1) * 128;
    if (m0 >= p.M || n0 >= p.N) return;  // quadrant outside the problem (uniform for the workgroup)
  } else {
    const int total = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.nbatch;
    if ((int)blockIdx.x >= total) {   // a workgroup beyond the tiles (launch_gemm8s pads to 256): warm the next launch's weights
      if (p.pf_ptr) prefetch_lines(p, (int)blockIdx.x - total, (int)gridDim.x - total, (int)blockDim.x);
      return;
    }
    tile_of(p, BM, BN, xcd_run_pos(total), b, tm, tn);
    m0 = tm * BM;
    n0 = tn * BN;
  }
  const WGeom wg_ = w_geom(p);

  // staging: wave w moves rows 32w .. 32w+31 of both tiles as 4 + 4 wave instructions of 8 rows (1 KiB each):
  // lane -> row 32w + 8q + (lane>>3), 16-byte slot lane&7, which must hold source chunk slot ^ ((row>>1)&7).
  const int r8 = lane >> 3;
  const bf16_t* a_row[4];
  const bf16_t* w_row[4];
  unsigned a_off[4], w_off[4];   // plain GEMMs: 32-bit byte offsets from the batch item's base (as gemm8_kernel)
  int a_in[4];
  long a_tap[4];
  const bf16_t* const A0 = (const bf16_t*)p.A + p.a_off + (long)b * p.a_bstride;
  const bf16_t* const W0 = (const bf16_t*)p.W + (long)b * p.w_bstride;
  {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = wave * 32 + q * 8 + r8;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      a_row[q] = A0 + (long)m * p.lda;
      a_off[q] = (unsigned)(((long)m * p.lda + chunk * 8) * 2);
      int n = n0 + row;
      n = n < p.N ? n : p.N - 1;
      w_row[q] = W0 + (long)n * wg_.ns + chunk * 8;
      w_off[q] = (unsigned)(((long)n * wg_.ns + chunk * 8) * 2);
      a_in[q] = chunk * 8;
      a_tap[q] = 0;
      if constexpr (CONV)
        while (a_in[q] >= p.kc) { a_in[q] -= p.kc; a_tap[q] += p.tap_stride; }
    }
  }
  const int nt = p.K / BK;
  const size_t lds0 = (size_t)(__attribute__((address_space(3))) char*)smem;
  // DMA as inline assembly (dma16s / dma16v, see there): the compiler then counts the fragment reads itself, which is what
  // lets the pipelined form's reads of K-tile t+1 really complete underneath the MFMAs of K-tile t
  // K-tile kt (CONV: the a_in / a_tap state points at it) -> stage buf: the A-tile loads QA0 <= q < QA1 and the W-tile loads q < QW1
  // of this wave's row block
  // GEMM_FLAG_X3_SHARE (common.h): K-tile kt of the walk = product kt % 3 of original K-tile kt / 3, in gemm8x_kernel's order
  // (x_lo, W_hi) (x_hi, W_hi) (x_hi, W_lo) - the same accumulation order per output element, whichever of the two kernels runs
  const bool share = (p.flags & GEMM_FLAG_X3_SHARE) != 0;
  const int T3 = p.K / 192;
  auto stage_q = [&](int buf, int kt, auto QA0, auto QA1, auto QW1) {
    const size_t dst = lds0 + (size_t)(buf * (2 * TB) + wave * 4096);
    int ka = kt, kw = kt;
    if (share) {
      const int t3 = kt / 3, r3 = kt - 3 * t3;
      ka = r3 == 0 ? t3 : T3 + t3;
      kw = r3 == 2 ? T3 + t3 : t3;
    }
    const char* a_base = (const char*)A0 + (long)ka * (BK * 2);
    const char* w_base = (const char*)W0 + (long)kw * wg_.kstep;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < decltype(QA0)::value || q >= decltype(QA1)::value) continue;
      if constexpr (CONV) dma16v(a_row[q] + a_tap[q] + a_in[q], dst + q * 1024);
      else dma16s(a_base, a_off[q], dst + q * 1024);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q >= decltype(QW1)::value) continue;
      if constexpr (CONV) dma16v((const char*)w_row[q] + (long)kw * wg_.kstep, dst + TB + q * 1024);
      else dma16s(w_base, w_off[q], dst + TB + q * 1024);
    }
    if constexpr (CONV) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a_in[q] += BK;
        while (a_in[q] >= p.kc) { a_in[q] -= p.kc; a_tap[q] += p.tap_stride; }
      }
    }
  };
  using Q0 = std::integral_constant<int, 0>;
  using Q4 = std::integral_constant<int, 4>;
  auto stage = [&](int buf, int kt) { stage_q(buf, kt, Q0{}, Q4{}, Q4{}); };

  if constexpr (ROLES) {
    if (wave_id >= 4) {   // a requesting wave: the ring's producer side, no arithmetic
      using QC = std::integral_constant<int, PROD>;
      constexpr int MINE = 8 - PROD;   // loads per K-tile of this wave
      auto request = [&](int buf, int kt) { stage_q(buf, kt, QC{}, Q4{}, Q4{}); };
      request(0, 0);
      if (nt > 1) request(1, 1);
      if (nt > 2) request(2, 2);
      if (nt > 2) wait_vm_lit<2 * MINE>();
      else if (nt > 1) wait_vm_lit<MINE>();
      else wait_vm_lit<0>();
      __builtin_amdgcn_s_barrier();
      for (int t = 0; t + 1 < nt; ++t) {
        // K-tile t+3 -> the buffer K-tile t-1 was read from: those reads completed before the barrier of step t-1
        if (t + 3 < nt) request((t + 3) & 3, t + 3);
        if (t + 3 < nt) wait_vm_lit<2 * MINE>();        // K-tile t+1 has landed; t+2, t+3 may be in flight
        else if (t + 2 < nt) wait_vm_lit<MINE>();
        else wait_vm_lit<0>();
        __builtin_amdgcn_s_barrier();
      }
      if (!(p.flags & 64)) __syncthreads();   // the multiplying waves' barrier in front of the LDS-staged epilogues
      return;
    }
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto frag = [&](const char* tile_base, int row, int ks) -> bf16x8_t {
    return *(const bf16x8_t*)(tile_base + row * 128 + ((((ks << 2) + lg) ^ ((row >> 1) & 7)) << 4));
  };

  if constexpr (PIPE) {
    bf16x8_t af[2][4][2], wf[2][4][2];   // [fragment set][16-row block][k-step]
    // loads per K-tile of a multiplying wave: all 8 of its row block, or - with requesting waves - PROD of the A tile's
    constexpr int MINE = ROLES ? PROD : 8;
    auto stage_mine = [&](int buf, int kt) {
      if constexpr (!ROLES) stage(buf, kt);
      else if constexpr (PROD > 0) stage_q(buf, kt, Q0{}, std::integral_constant<int, PROD>{}, Q0{});
    };
    auto read_frags = [&](int buf, auto SET) {
      const char* At = smem + buf * (2 * TB);
      const char* Wt = At + TB;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          af[SET()][i][ks] = frag(At, wr * 64 + i * 16 + lr, ks);
          wf[SET()][i][ks] = frag(Wt, wc * 64 + i * 16 + lr, ks);
        }
    };
    // NEXT: K-tile t+1 exists.  A compile-time switch, not a branch: at a control-flow merge the compiler would wait for
    // EVERY outstanding LDS read before the MFMAs (it cannot keep a per-path count), which serialises read and multiply again.
    auto step = [&](int t, auto SET, auto NEXT) {   // fragments of K-tile t are set SET (reads issued one step earlier)
      constexpr int OTHER = 1 - decltype(SET)::value;
      // K-tile t+S-1 -> the buffer K-tile t-1 was read from: those reads COMPLETED before the barrier of step t-1
      if (t + S - 1 < nt) stage_mine((t + S - 1) % S, t + S - 1);
      if constexpr (decltype(NEXT)::value) {
        // K-tile t+1 has landed; the younger ones (t+2 .. t+S-1, as far as they exist) may be in flight
        if constexpr (MINE > 0) {   // (a multiplying wave that requests nothing has nothing to wait for)
          if (S == 4 && t + 3 < nt) wait_vm_lit<2 * MINE>();
          else if (t + 2 < nt) wait_vm_lit<MINE>();
          else wait_vm_lit<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // this wave's reads of K-tile t are complete
        __builtin_amdgcn_s_barrier();
        read_frags((t + 1) % S, std::integral_constant<int, OTHER>{});      // in flight underneath the MFMAs below
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i][j] = ALT ? SA_MFMA_16x16x32_ALT(wf[SET()][j][ks], af[SET()][i][ks], acc[i][j])
                            : SA_MFMA_16x16x32(wf[SET()][j][ks], af[SET()][i][ks], acc[i][j]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    stage_mine(0, 0);
    if (nt > 1) stage_mine(1, 1);
    if (S == 4 && nt > 2) stage_mine(2, 2);
    // K-tile 0 has landed; up to S - 2 younger ones stay in flight
    if constexpr (MINE > 0) {
      if (S == 4 && nt > 2) wait_vm_lit<2 * MINE>();
      else if (nt > 1) wait_vm_lit<MINE>();
      else wait_vm_lit<0>();
    }
    __builtin_amdgcn_s_barrier();
    read_frags(0, I0{});
    int t = 0;
    for (; t + 2 < nt; t += 2) {   // both steps have a successor
      step(t, I0{}, std::true_type{});
      step(t + 1, I1{}, std::true_type{});
    }
    if (nt - t == 2) {
      step(t, I0{}, std::true_type{});
      step(t + 1, I1{}, std::false_type{});
    } else {
      step(t, I0{}, std::false_type{});
    }
    if (p.flags & 64) {
      epilogue8_linear<1>(p, acc, b, m0 + wr * 64, n0 + wc * 64, lane);
      return;
    }
    __syncthreads();
    if (p.flags & 128) epilogue8_rows<1>(p, acc, smem + wave * 16384, b, m0 + wr * 64, n0 + wc * 64, lane);
    else epilogue8<1>(p, acc, smem + wave * 16384, b, m0 + wr * 64, n0 + wc * 64, lane);
    return;
  }
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int cb = t & 1;
    if (t + 1 < nt) {
      stage(cb ^ 1, t + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the 8 loads just issued stay in flight; K-tile t has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* At = smem + cb * (2 * TB);
    const char* Wt = At + TB;
    bf16x8_t af[4][2], wf[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        af[i][ks] = frag(At, wr * 64 + i * 16 + lr, ks);
        wf[i][ks] = frag(Wt, wc * 64 + i * 16 + lr, ks);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave holds its fragments: stage cb may be overwritten by K-tile t+2
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][j] = ALT ? SA_MFMA_16x16x32_ALT(wf[j][ks], af[i][ks], acc[i][j]) : SA_MFMA_16x16x32(wf[j][ks], af[i][ks], acc[i][j]);
  }
  if (p.flags & 64) {
    epilogue8_linear<1>(p, acc, b, m0 + wr * 64, n0 + wc * 64, lane);
    return;
  }
  __syncthreads();
  if (p.flags & 128) epilogue8_rows<1>(p, acc, smem + wave * 16384, b, m0 + wr * 64, n0 + wc * 64, lane);
  else epilogue8<1>(p, acc, smem + wave * 16384, b, m0 + wr * 64, n0 + wc * 64, lane);
}

// CONV instantiation = per-lane 64-bit source pointers: implicit convolutions, and plain operands whose byte offsets from the
// batch item's base do not fit 32 bits
static bool gemm8_wide(const GemmParams& p) {
  return p.kc < p.K || (long)p.M * p.lda * 2 >= (1L << 32) || (long)p.N * p.K * 2 >= (1L << 32);
}
// The launches whose epilogue is "linear" (every Linear of the DiT / the towers): bit 6 = epilogue8_linear (16-bit output
// only: straight from the accumulator layout), bit 7 = epilogue8_rows (fp32 output / residual: through the wave's LDS area).
// Debug flag 24: 1 = the general epilogue for everything (the bitwise-equality tests), 2 / 3 = the register / LDS form for
// every eligible launch (A/B).
static int gemm8_linear_epilogue(const GemmParams& p) {
  const int mode = debug_flag(24);
  if (mode == 1) return 0;
  if (p.act != ACT_NONE || p.chan_mod || p.c_ld_rel || (p.flags & 1) || p.N % 64 || p.alpha != 1.f) return 0;
  if (!p.out_act && !p.out_f32) return 0;
  if (p.swiglu && (!p.out_act || p.out_f32 || p.bias || p.gate || p.res)) return 0;
  if ((p.flags & GEMM_FLAG_OUT_SPLIT3) && (!p.swiglu || (p.flags & 512) || (p.N / 2) % 8)) return 0;   // (gemm8_split3_ok)
  if (p.bias && (p.gate || p.gate_tab)) return 0;
  if (p.gate_tab && !p.gate) return 0;
  auto al = [](long v, long a) { return v % a == 0; };
  if (p.out_act && !(al(p.act_ld, 8) && al(p.act_off, 8) && al(p.act_bstride, 8) && ((uintptr_t)p.out_act & 15) == 0))
    return 0;
  if (p.gate && (p.rows_per_gate <= 0 || (long)p.M * p.nbatch >= (1L << 31))) return 0;
  // 16-byte alignment of the fp32 operands: gemm2_ok(), checked by the policy for every launch of this file
  if (p.swiglu || mode == 2) return 64;
  if (mode == 3) return 128;
  return p.out_f32 || p.res ? 128 : 64;
}
static GemmParams with_epilogue_choice(const GemmParams& p) {
  GemmParams q = p;
  q.flags = (q.flags & ~192) | gemm8_linear_epilogue(p);
  if (!q.raster_gm && debug_flag(35) > 0) q.raster_gm = debug_flag(35);   // (A/B) M-tiles per raster group of the 8-phase family
  return q;
}

// eligibility: the vectorised-epilogue conditions of gemm2_ok() (checked by the caller) - any M, N, K % 64 == 0
// (Round 3, GPU call 8: the same kernel on 96 x 128 tiles - 242 instead of 176 workgroups at 1000 rows x N = 2816, bitwise
// identical - ran exactly as fast: c_wq 36.0 vs 35.3 us, 4 clips 114.8 vs 115.1 s-audio/s, small* 423 vs 424.  With few
// rows a launch lasts nt x ~0.8 us whatever its workgroup count: it is bound by the depth of the K-tile prefetch (two K-tiles
// of L2 latency in flight), not by how many CUs hold a tile.  Removed; profiles/r3_call8/.)
// alt-format operands (flags bit 10, mixed mode) exist for plain GEMMs only - the DiT's Linears - gemm8_alt_ok()
#define SA_GEMM8S_ROLES_DEFAULT 2   // the shipped form of the pipelined kernel: -1 no roles, 0 / 2 = PROD
static void launch_gemm8s_grid(const GemmParams& p, bool pipe, bool conv, dim3 grid, int skip256, hipStream_t st) {
  const dim3 block(256);
  const bool alt = (p.flags & 1024) != 0;
  // wave roles of the pipelined form (see the kernel): debug flag 27 = 1 the form without roles (round 3), 2 / 3 force PROD = 0 / 2
  const int roles = !pipe || debug_flag(27) == 1 ? -1 : debug_flag(27) == 3 ? 2 : debug_flag(27) == 2 ? 0 : SA_GEMM8S_ROLES_DEFAULT;
  if (roles >= 0) {
    const dim3 block8(512);
    if (conv && roles == 0) hipLaunchKernelGGL((gemm8s_kernel<true, true, false, 0>), grid, block8, 0, st, p, skip256);
    else if (conv) hipLaunchKernelGGL((gemm8s_kernel<true, true, false, 2>), grid, block8, 0, st, p, skip256);
    else if (alt && roles == 0) hipLaunchKernelGGL((gemm8s_kernel<true, false, true, 0>), grid, block8, 0, st, p, skip256);
    else if (alt) hipLaunchKernelGGL((gemm8s_kernel<true, false, true, 2>), grid, block8, 0, st, p, skip256);
    else if (roles == 0) hipLaunchKernelGGL((gemm8s_kernel<true, false, false, 0>), grid, block8, 0, st, p, skip256);
    else hipLaunchKernelGGL((gemm8s_kernel<true, false, false, 2>), grid, block8, 0, st, p, skip256);
    return;
  }
  if (pipe && conv) hipLaunchKernelGGL((gemm8s_kernel<true, true>), grid, block, 0, st, p, skip256);
  else if (pipe && alt) hipLaunchKernelGGL((gemm8s_kernel<true, false, true>), grid, block, 0, st, p, skip256);
  else if (pipe) hipLaunchKernelGGL((gemm8s_kernel<true, false>), grid, block, 0, st, p, skip256);
  else if (conv) hipLaunchKernelGGL((gemm8s_kernel<false, true>), grid, block, 0, st, p, skip256);
  else if (alt) hipLaunchKernelGGL((gemm8s_kernel<false, false, true>), grid, block, 0, st, p, skip256);
  else hipLaunchKernelGGL((gemm8s_kernel<false, false>), grid, block, 0, st, p, skip256);
}
// flags bit 12 is well-formed: only the register epilogue of a SwiGLU launch writes the split form
bool gemm8_split3_ok(const GemmParams& p) {
  return !(p.flags & GEMM_FLAG_OUT_SPLIT3) || (p.swiglu && p.out_act && gemm8_linear_epilogue(p) == 64);
}
// flags bit 15 is well-formed: plain operands within 32-bit offsets (no implicit convolution), K' = 3K with K a multiple of 64, the
// library's own operand format
bool gemm8_share_ok(const GemmParams& p) {
  return !(p.flags & GEMM_FLAG_X3_SHARE) || (!gemm8_wide(p) && !(p.flags & 1024) && p.K % 192 == 0 && p.kc == p.K);
}
// flags bits 9 / 10 are well-formed for this launch: plain operands within 32-bit offsets, a lean epilogue for an alt-format output
bool gemm8_alt_ok(const GemmParams& p) {
  if (!(p.flags & (512 | 1024))) return true;
  if ((p.flags & 1024) && gemm8_wide(p)) return false;
  if ((p.flags & 512) && (!p.out_act || gemm8_linear_epilogue(p) == 0)) return false;
  return true;
}

hipError_t launch_gemm8s(const GemmParams& p_in, hipStream_t st) {
  const GemmParams p = with_epilogue_choice(p_in);
  const long tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.nbatch;
  // flag 21 (A/B): the plain double-buffered form for every launch, as before GPU call 25 of round 2
  const bool pipe = tiles <= 256 && !debug_flag(21), conv = gemm8_wide(p);
  // pf_ptr: the pipelined form holds one workgroup per CU - a launch of fewer than 256 tiles is padded with workgroups that
  // touch the next launch's weights on the CUs it leaves idle (prefetch_lines)
  const dim3 grid((unsigned)(p.pf_ptr && p.pf_bytes > 0 && pipe && tiles < 256 ? 256 : tiles));
  launch_gemm8s_grid(p, pipe, conv, grid, -1, st);
  return hipGetLastError();
}

static void launch_gemm8_tiles(const GemmParams& p, dim3 grid, int tile_count, hipStream_t st) {
  const dim3 block(512);
#ifdef SAMAUDIO_GEMM8_ABL   // timing experiments (tools/build_abl.sh): debug flag 25 selects an ablation of the round-3 loop
  if (!(p.kc < p.K)) switch (debug_flag(25)) {
    case 1: hipLaunchKernelGGL((gemm8o_kernel<false, 1>), grid, block, 0, st, p, tile_count); return;
    case 2: hipLaunchKernelGGL((gemm8o_kernel<false, 2>), grid, block, 0, st, p, tile_count); return;
    case 3: hipLaunchKernelGGL((gemm8o_kernel<false, 3>), grid, block, 0, st, p, tile_count); return;
    case 4: hipLaunchKernelGGL((gemm8o_kernel<false, 4>), grid, block, 0, st, p, tile_count); return;
    case 5: hipLaunchKernelGGL((gemm8o_kernel<false, 5>), grid, block, 0, st, p, tile_count); return;
    case 8: hipLaunchKernelGGL((gemm8o_kernel<false, 0>), grid, block, 0, st, p, tile_count); return;
    case 9: hipLaunchKernelGGL((gemm8o_kernel<false, 9>), grid, block, 0, st, p, tile_count); return;
    default: break;
  }
#endif
  // Persistent above one round of the chip: at most one workgroup per CU, each walking its XCD's run of tiles (debug flag 26 =
  // 1: one workgroup per tile, as in round 3).  Per launch the walk is worth 1 - 4 % (w13 at 4 000 rows 259 -> 254 us: the
  // epilogue stores drain under the next prologue); end to end, with two row groups on two streams, 223.5 -> 234.2 s-audio/s
  // (+4.7 %, profiles/r4_call7/): a launch now keeps its CUs for its whole duration instead of re-competing for them with
  // the other group's launch after every tile.
  if (debug_flag(26) != 1 && grid.x > 256) grid.x = 256;
  if (p.pf_ptr && p.pf_bytes > 0 && tile_count == 0 && grid.x < 256) grid.x = 256;   // idle CUs warm the next launch's weights
  if ((p.flags & GEMM_FLAG_X3_SHARE) && !gemm8_wide(p) && !(p.flags & 1024)) hipLaunchKernelGGL(gemm8x_kernel, grid, block, 0, st, p, tile_count);
  else if (gemm8_wide(p)) hipLaunchKernelGGL((gemm8_kernel<true>), grid, block, 0, st, p, tile_count);
  else if (p.flags & 1024) hipLaunchKernelGGL((gemm8_kernel<false, true>), grid, block, 0, st, p, tile_count);   // alt-format operands
  else hipLaunchKernelGGL((gemm8_kernel<false>), grid, block, 0, st, p, tile_count);
}

hipError_t launch_gemm8(const GemmParams& p_in, hipStream_t st) {
  const GemmParams p = with_epilogue_choice(p_in);
  const long tiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * p.nbatch;
  const dim3 grid((unsigned)tiles), block(512);
  launch_gemm8_tiles(p, grid, 0, st);
  return hipGetLastError();
}

// Tile-quantisation split: 256x256 tiles fill the chip only in whole rounds of 256 workgroups (one per CU); the last,
// partial round of a launch leaves CUs idle for a full tile time (352 tiles at N = D: 2 rounds for 1.375 rounds of
// work).  part 0 = the 8-phase kernel on the first `full` tiles of the raster order, part 1 = the remaining tiles as
// 128x128 quadrants on gemm8s_kernel (two workgroups per CU, 4x finer granularity).  Bitwise the same results.
hipError_t launch_gemm8_split(const GemmParams& p_in, int full, int part, hipStream_t st) {
  const GemmParams p = with_epilogue_choice(p_in);
  const long tiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * p.nbatch;
  if (full <= 0 || full >= tiles) return hipErrorInvalidValue;
  if (part == 0) launch_gemm8_tiles(p, dim3((unsigned)full), full, st);
  else {
    const bool pipe = (tiles - full) * 4 <= 256 && !debug_flag(21);   // a tail that cannot give a CU two workgroups
    const bool conv = gemm8_wide(p);
    const dim3 grid((unsigned)((tiles - full) * 4));
    launch_gemm8s_grid(p, pipe, conv, grid, full, st);
  }
  return hipGetLastError();
}

}  // namespace sa
