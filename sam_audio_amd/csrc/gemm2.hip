// Second-generation bf16 GEMM / implicit convolution for the large DiT contractions (gfx950).
//
// Same operand model as gemm.hip (GemmParams: "NT" operands, K-contiguous rows, taps folded into the row
// address) and the same LDS image (128-byte row slabs, 16-byte chunks XOR-swizzled with (row>>1)&7, filled by
// direct-to-LDS DMA with the swizzle on the per-lane SOURCE address), re-tiled for the shapes that carry
// >95 % of the separate() FLOPs (M = B*250 rows, N,K in {D, 3D, 2F, F}; reference transformer.py:102-114,
// 186-189):
//   * 256-row tiles, 8 waves (512 threads), one workgroup per CU: the 128x128 tile of gemm.hip needs
//     ~39 TB/s of L2->LDS traffic at the MFMA peak, above what the 8 L2s deliver; 256x128 needs 29, 256x256 19.
//   * v_mfma_f32_32x32x16_bf16 with the operands SWAPPED (W fragment as the row operand): every lane then
//     owns 4 consecutive output COLUMNS of one output row, so bias / gate / residual are float4 loads and the
//     fp32 / bf16 outputs are 16-byte / 8-byte stores (4x fewer epilogue memory instructions than gemm.hip).
//   * STAGES-deep LDS ring.  The K loop keeps STAGES-2 slabs of DMA in flight ACROSS the per-slab barrier:
//     counted s_waitcnt vmcnt(G) + raw s_barrier (a __syncthreads() would drain the DMA queue), so HBM/L2
//     latency hides under the previous slab's MFMAs although only one workgroup lives on the CU.
//   * XCD-aware rasterisation: hardware deals consecutive workgroups round-robin to the 8 XCDs; the remap
//     gives every XCD one contiguous run of tiles, walked in groups of 8 M-tiles x all N-tiles, so the 32
//     tiles resident on one XCD share A-row and W-column panels in that XCD's private L2.
//   * accumulation order depends only on k, never on M / batch: results are bitwise invariant to sharding.
#include "common.h"
#include "kernels.h"

namespace sa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_SILU) return silu_f(v);
  if (act == ACT_TANH) return tanhf(v);
  return v;
}

}  // namespace

template <int BM, int BN, int WM_, int WN_, int STAGES>
__global__ __launch_bounds__(WM_* WN_ * 64) void gemm2_kernel(const GemmParams p) {
  constexpr int NW = WM_ * WN_;
  constexpr int NT = NW * 64;
  constexpr int WTM = BM / WM_, WTN = BN / WN_;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  constexpr int AI = BM / (8 * NW), BI = BN / (8 * NW);  // DMA instructions per wave per slab
  constexpr int G = AI + BI;
  constexpr int TILE_A = BM * 128, TILE_B = BN * 128, STAGE = TILE_A + TILE_B;
  constexpr int BK = 64, CH = 8;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && WTM % 32 == 0 && WTN % 32 == 0, "tile shape");
  static_assert(STAGES >= 2 && STAGES * STAGE <= 160 * 1024, "LDS budget");
  static_assert((STAGES - 2) * G <= 63, "vmcnt range");
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE];
  (void)NT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_, wn = wave % WN_;

  // ---- workgroup -> tile: XCD-contiguous, grouped raster ---------------------------------------------
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int per_batch = tiles_m * tiles_n;
  int b, tm, tn;
  {
    const int total = per_batch * p.nbatch;
    const int bid = blockIdx.x;
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective for any total
    b = L / per_batch;
    const int l2 = L - b * per_batch;
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int grp = l2 / per_group;
    const int first_m = grp * GM;
    const int gsz = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    const int in_grp = l2 - grp * per_group;
    tm = first_m + in_grp % gsz;
    tn = in_grp / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane DMA sources (see gemm.hip): wave-instruction j = wave + NW*i fills tile rows 8j..8j+7 ----
  const int r8 = lane >> 3;
  const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (r8 >> 1)) & 7);
  const bf16_t* a_rows[AI];
  const bf16_t* w_rows[BI];
  {
    const bf16_t* A = (const bf16_t*)p.A + p.a_off + (long)b * p.a_bstride;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int m = m0 + (wave + NW * i) * 8 + r8;
      m = m < p.M ? m : p.M - 1;
      a_rows[i] = A + (long)m * p.lda;
    }
    const bf16_t* W = (const bf16_t*)p.W;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int n = n0 + (wave + NW * i) * 8 + r8;
      n = n < p.N ? n : p.N - 1;
      w_rows[i] = W + (long)n * p.K + chunk * CH;
    }
  }
  int a_in = chunk * CH;
  long a_tap = 0;
  while (a_in >= p.kc) { a_in -= p.kc; a_tap += p.tap_stride; }

  auto issue = [&](int stage) {
    char* sA = smem + stage * STAGE;
    char* sB = sA + TILE_A;
#pragma unroll
    for (int i = 0; i < AI; ++i) dma16(a_rows[i] + a_tap + a_in, sA + (wave + NW * i) * 1024);
#pragma unroll
    for (int i = 0; i < BI; ++i) dma16(w_rows[i], sB + (wave + NW * i) * 1024);
    a_in += BK;
    while (a_in >= p.kc) { a_in -= p.kc; a_tap += p.tap_stride; }
#pragma unroll
    for (int i = 0; i < BI; ++i) w_rows[i] += BK;
  };

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int swz = (l31 >> 1) & 7;  // (row>>1)&7 for every fragment row of this lane (fragment bases are multiples of 32)
  const int a_base = (wm * WTM + l31) * 128, b_base = (wn * WTN + l31) * 128;

  const int nslab = p.K / BK;
  // prologue: STAGES-1 slabs in flight
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nslab) issue(s);
  int st_c = 0, st_i = (STAGES - 1) % STAGES;
  for (int s = 0; s < nslab; ++s) {
    // slab s has landed once at most the (STAGES-2) younger slabs' DMAs are still outstanding
    if (STAGES > 2 && s + STAGES - 2 < nslab) wait_vmcnt<(STAGES - 2) * G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // slab s visible to every wave; every wave is done reading slab s-1's stage
    __builtin_amdgcn_sched_barrier(0);
    if (s + STAGES - 1 < nslab) issue(st_i);
    const char* sA = smem + st_c * STAGE;
    const char* sB = sA + TILE_A;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((ks * 2 + lh) ^ swz) << 4;
      bf16x8_t af[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8_t*)(sA + a_base + i * 32 * 128 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8_t*)(sB + b_base + j * 32 * 128 + coff);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    st_c = st_c + 1 == STAGES ? 0 : st_c + 1;
    st_i = st_i + 1 == STAGES ? 0 : st_i + 1;
  }

  // ---- epilogue --------------------------------------------------------------------------------------
  // swapped operands: D[n][m]; lane holds m = l31, n = 8*g + 4*lh + (0..3) for register group g = reg>>2.
  // Per 32x32 fragment: all loads (bias / gate / residual, float4 each) are issued first, then the arithmetic,
  // then the 16-byte fp32 / 8-byte bf16 stores - one memory round trip per fragment instead of one per value.
  const long bM = (long)b * p.M;
  const bool has_bias = p.bias != nullptr, has_gate = p.gate != nullptr, has_tab = p.gate_tab != nullptr,
             has_res = p.res != nullptr;
  const int NG = p.swiglu ? 2 : 4;       // swiglu: groups 0,1 = w1 rows of the 32-row block, groups 2,3 = matching w3 rows
  const int n_out = p.swiglu ? p.N >> 1 : p.N;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * WTM + i * 32 + l31;
    const bool m_ok = m < p.M;
    const int mc = m_ok ? m : p.M - 1;
    const float* grow = has_gate ? p.gate + ((bM + mc) / p.rows_per_gate) * p.gate_ld : nullptr;
    const float* rrow = has_res ? p.res + p.res_off + (long)b * p.res_bstride + (long)mc * p.res_ld : nullptr;
    float* frow = p.out_f32 ? p.out_f32 + p.f32_off + (long)b * p.f32_bstride + (long)mc * p.f32_ld : nullptr;
    bf16_t* arow = p.out_act ? (bf16_t*)p.out_act + p.act_off + (long)b * p.act_bstride + (long)mc * p.act_ld : nullptr;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nf = n0 + wn * WTN + j * 32;  // first GEMM column of this 32-wide fragment
      const int nb = (p.swiglu ? nf >> 1 : nf) + 4 * lh;
      int ncol[4];
      float4 bb[4], gg[4], tt[4], rr[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nb + 8 * g;
        ncol[g] = n;
        const int nc = n + 4 <= n_out ? n : n_out - 4;  // clamped address for the loads of masked columns
        if (has_bias) bb[g] = *(const float4*)(p.bias + nc);
        if (has_gate) gg[g] = *(const float4*)(grow + nc);
        if (has_tab) tt[g] = *(const float4*)(p.gate_tab + nc);
        if (has_res) rr[g] = *(const float4*)(rrow + nc);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g >= NG) continue;
        float v[4];
        if (p.swiglu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f(acc[i][j][4 * g + e]) * acc[i][j][4 * ((g + 2) & 3) + e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
        }
        if (has_bias) { v[0] += bb[g].x; v[1] += bb[g].y; v[2] += bb[g].z; v[3] += bb[g].w; }
        if (has_gate) {
          float4 q = gg[g];
          if (has_tab) { q.x += tt[g].x; q.y += tt[g].y; q.z += tt[g].z; q.w += tt[g].w; }
          v[0] *= q.x; v[1] *= q.y; v[2] *= q.z; v[3] *= q.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
        if (has_res) { v[0] += rr[g].x; v[1] += rr[g].y; v[2] += rr[g].z; v[3] += rr[g].w; }
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = act_apply(v[e], p.act);
        if (m_ok && ncol[g] < n_out) {
          if (frow) {
            if (p.f32_act) *(float4*)(frow + ncol[g]) = make_float4(a[0], a[1], a[2], a[3]);
            else *(float4*)(frow + ncol[g]) = make_float4(v[0], v[1], v[2], v[3]);
          }
          if (arow) store4<bf16_t>(arow + ncol[g], a[0], a[1], a[2], a[3]);
        }
      }
    }
  }
}

template <int BM, int BN, int WM_, int WN_, int STAGES>
static hipError_t launch2(const GemmParams& p, hipStream_t st) {
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.nbatch;
  hipLaunchKernelGGL((gemm2_kernel<BM, BN, WM_, WN_, STAGES>), dim3((unsigned)tiles), dim3(WM_ * WN_ * 64), 0, st, p);
  return hipGetLastError();
}

// Can this problem take the vectorised-epilogue kernels?  (everything else stays on gemm.hip)
bool gemm2_ok(const GemmParams& p) {
  if (p.chan_mod || p.c_ld_rel || p.act == ACT_SNAKE) return false;
  if (p.N % 4 || p.K % 64) return false;
  if (p.swiglu && p.N % 32) return false;
  auto al4 = [](long v) { return (v & 3) == 0; };
  if (p.out_f32 && !(al4(p.f32_ld) && al4(p.f32_off) && al4(p.f32_bstride) && ((uintptr_t)p.out_f32 & 15) == 0))
    return false;
  if (p.out_act && !(al4(p.act_ld) && al4(p.act_off) && al4(p.act_bstride) && ((uintptr_t)p.out_act & 7) == 0))
    return false;
  if (p.res && !(al4(p.res_ld) && al4(p.res_off) && al4(p.res_bstride) && ((uintptr_t)p.res & 15) == 0)) return false;
  if (p.gate && !(al4(p.gate_ld) && ((uintptr_t)p.gate & 15) == 0)) return false;
  if (p.gate_tab && ((uintptr_t)p.gate_tab & 15)) return false;
  if (p.bias && ((uintptr_t)p.bias & 15)) return false;
  return true;
}

// variants: 0 = 256x128 3-stage, 1 = 256x128 2-stage, 2 = 256x256 2-stage
hipError_t launch_gemm2(const GemmParams& p, int variant, hipStream_t st) {
  switch (variant) {
    case 2: return launch2<256, 256, 2, 4, 2>(p, st);
    case 1: return launch2<256, 128, 4, 2, 2>(p, st);
    default: return launch2<256, 128, 4, 2, 3>(p, st);
  }
}

}  // namespace sa
