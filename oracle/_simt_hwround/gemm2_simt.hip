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

typedef h16x8_t bf16x8_t;  // 8 x 16-bit operand words (bf16, or fp16 with -DSA_OPERAND_FP16: common.h)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

// Direct-to-LDS load of 16 bytes per lane (lane i lands at lds_wave_base + 16 i).
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same as inline assembly, for the residual-unit kernels.  They also WRITE the DMA-filled LDS region with ordinary
// stores (the intermediate activation), and for such a kernel the compiler makes every LDS read that follows a
// direct-to-LDS load it knows about wait for that load: s_waitcnt vmcnt(0) in front of the ds_reads of the next k-step,
// i.e. issue -> wait -> compute instead of a ring (gemm2_kernel / conv7h_kernel, which only read their LDS, do not get
// these waits).  Completion is ordered by the explicit s_waitcnt vmcnt / barriers of the kernel - a __syncthreads() does
// NOT imply vmcnt(0) for loads the compiler cannot see.  M0 = LDS base; one wait state between the M0 write and the DMA.
__device__ __forceinline__ void dma16a(const void* gsrc, char* lds_wave_base) {
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  simt::dma_asm = true; __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0); simt::dma_asm = false; (void)lds;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  simt::wait_vmcnt(N);
}
__device__ __forceinline__ float act_apply(float v, int act, float snake_alpha) {
  if (act == ACT_SILU) return silu_f(v);
  if (act == ACT_GELU) return gelu_f(v);
  if (act == ACT_QUICK_GELU) return quick_gelu_f(v);
  if (act == ACT_RELU) return relu_f(v);
  if (act == ACT_GELU_TANH) return gelu_tanh_f(v);
  if (act == ACT_TANH) return tanhf(v);
  if (act == ACT_SNAKE) return snake16_f(v, snake_alpha);  // x + sin^2(a x) / a
  return v;
}

}  // namespace

// ---- LDS-staged variant of the same epilogue ----------------------------------------------------------------------
// In the accumulator layout above one store instruction covers 32 different rows for 32 contiguous bytes each: 32 cache
// lines per instruction, for the fp32 residual read, the fp32 write and the bf16 write alike - the three residual GEMMs
// of a DiT layer spend 30-60 us of their 150-420 us in it (profiles/r2_gemm_variants.log: wo 601 vs c_wq 845 TF/s on the
// same shape).  Here every 32-row slice of the wave's tile goes through a PRIVATE LDS region of the wave (FN x 4 KiB,
// free after the K loop; 16-byte chunks XOR-swizzled with row & 7) and comes back with 4 consecutive columns per lane
// and FN*8 lanes per row, so all global traffic of the epilogue moves whole 128-/256-/512-byte row segments.
// Wave-private: only wave-level ordering (s_waitcnt lgkmcnt) is needed between the two phases.
template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams& p, f32x16_t (&acc)[FM][FN], int b, int m_first,
                                                  int n_first, int lane, char* stg) {
  const int l31 = lane & 31, lh = lane >> 5;
  const long bM = (long)b * p.M;
  const bool has_bias = p.bias != nullptr, has_gate = p.gate != nullptr, has_tab = p.gate_tab != nullptr,
             has_res = p.res != nullptr, has_snake = p.act == ACT_SNAKE;
  const int n_out = p.swiglu ? p.N >> 1 : p.N;
  const int cpr = p.swiglu ? FN * 4 : FN * 8;                   // 4-column chunks per staged row
  const int rb = cpr * 16;                                      // bytes per staged row
  const int col0 = p.swiglu ? n_first >> 1 : n_first;
  const int units = 32 * cpr;                                   // (row, chunk) pairs of one 32-row slice
  const int smask = (cpr & 7) ? 3 : 7;                          // the XOR must stay inside the row: cpr = 4 / 12 -> groups of 4
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    // write phase: fragment row l31, columns j*32 + 8g + 4lh .. +3
#pragma unroll
    for (int j = 0; j < FN; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        int chunk;
        if (p.swiglu) {  // groups 0,1 = w1 rows of the 32-row block, groups 2,3 = the matching w3 rows
          if (g >= 2) continue;
          v = make_float4(silu_f(acc[i][j][4 * g + 0]) * acc[i][j][4 * (g + 2) + 0],
                          silu_f(acc[i][j][4 * g + 1]) * acc[i][j][4 * (g + 2) + 1],
                          silu_f(acc[i][j][4 * g + 2]) * acc[i][j][4 * (g + 2) + 2],
                          silu_f(acc[i][j][4 * g + 3]) * acc[i][j][4 * (g + 2) + 3]);
          chunk = j * 4 + 2 * g + lh;
        } else {
          v = make_float4(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          chunk = j * 8 + 2 * g + lh;
        }
        *(float4*)(stg + l31 * rb + ((chunk ^ (l31 & smask)) << 4)) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // read phase
    for (int u = lane; u < units; u += 64) {
      const int row = u / cpr, chunk = u - row * cpr;
      const float4 sv = *(const float4*)(stg + row * rb + ((chunk ^ (row & smask)) << 4));
      const int m = m_first + i * 32 + row;
      const int n = col0 + chunk * 4;
      const bool m_ok = m < p.M;
      const int mc = m_ok ? m : p.M - 1;
      const int nc = n + 4 <= n_out ? n : n_out - 4;  // clamped address for the loads of masked columns
      const int ch = p.chan_mod ? nc % p.chan_mod : nc;
      float v[4] = {sv.x, sv.y, sv.z, sv.w};
      float4 bb, gg, tt, rr, sa = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_bias) bb = *(const float4*)(p.bias + ch);
      if (has_gate) gg = *(const float4*)(p.gate + ((bM + mc) / p.rows_per_gate) * p.gate_ld + nc);
      if (has_tab) tt = *(const float4*)(p.gate_tab + nc);
      if (has_res) rr = *(const float4*)(p.res + p.res_off + (long)b * p.res_bstride + (long)mc * p.res_ld + nc);
      if (has_snake) sa = *(const float4*)(p.act_alpha + ch);
      if (has_bias) { v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
      if (has_gate) {
        if (has_tab) { gg.x += tt.x; gg.y += tt.y; gg.z += tt.z; gg.w += tt.w; }
        v[0] *= gg.x; v[1] *= gg.y; v[2] *= gg.z; v[3] *= gg.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
      if (has_res) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
      const float a0 = act_apply(v[0], p.act, sa.x), a1 = act_apply(v[1], p.act, sa.y),
                  a2 = act_apply(v[2], p.act, sa.z), a3 = act_apply(v[3], p.act, sa.w);
      bool ok = m_ok && n < n_out;
      if (p.c_ld_rel) {  // transposed conv: keep only the (row, phase) pairs that fall inside the output
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
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this slice's reads precede the next slice's writes
  }
}

// BK = k-elements per LDS slab (64: 128-byte rows, 8 chunks, swizzle (row>>1)&7;  32: 64-byte rows, 4 chunks,
// swizzle (row>>2)&3 - both make the 16 rows of a ds_read_b128 lane group hit 16 distinct 16-byte bank slots).
// 4-wave configurations (BK = 32, <= 80 KiB LDS) run TWO workgroups per CU: the two are not barrier-coupled, so
// one's MFMAs cover the other's barrier / LDS-latency / epilogue time.
// TAG: 0 = DiT / generic, 1 = DAC-VAE launches (identical code under a second symbol so that rocprofv3 and bench.py can
// attribute the MFMA-bound DiT contractions and the codec convolutions separately; GemmParams.tag selects it).
template <int BM, int BN, int WM_, int WN_, int STAGES, int BK, int TAG = 0>
__global__ __launch_bounds__(WM_* WN_ * 64, (WM_ * WN_ == 4 && BM * BN >= 256 * 256 ? 1 : (WM_ * WN_ == 8 && BK == 32 ? 4 : 2)))
    void gemm2_kernel(
    const GemmParams p) {
  constexpr int NW = WM_ * WN_;
  constexpr int WTM = BM / WM_, WTN = BN / WN_;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  constexpr int RB = BK * 2;        // bytes per LDS row
  constexpr int CPR = RB / 16;      // 16-byte chunks per row
  constexpr int RPI = 1024 / RB;    // rows filled by one wave-wide DMA instruction
  constexpr int KS = BK / 16;       // MFMA k-steps per slab
  constexpr int AI = BM / (RPI * NW), BI = BN / (RPI * NW);  // DMA instructions per wave per slab
  constexpr int G = AI + BI;
  constexpr int TILE_A = BM * RB, TILE_B = BN * RB, STAGE = TILE_A + TILE_B;
  constexpr int CH = 8;
  static_assert(BK == 64 || BK == 32, "BK");
  static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0 && WTM % 32 == 0 && WTN % 32 == 0, "tile shape");
  static_assert(STAGES >= 2 && STAGES * STAGE <= 160 * 1024, "LDS budget");
  static_assert((STAGES - 2) * G <= 63, "vmcnt range");
  static_assert(NW % 2 == 0, "swizzle bookkeeping assumes an even wave count");
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE];

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
    const int GM = p.raster_gm > 0 ? p.raster_gm : 8;
    const int per_group = GM * tiles_n;
    const int grp = l2 / per_group;
    const int first_m = grp * GM;
    const int gsz = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    const int in_grp = l2 - grp * per_group;
    tm = first_m + in_grp % gsz;
    tn = in_grp / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane DMA sources (see gemm.hip): wave-instruction j = wave + NW*i fills tile rows RPI*j .. RPI*j+RPI-1;
  // lane -> (row RPI*j + lane/CPR, 16-byte slot lane%CPR); slot s of row r holds source chunk s ^ swz(r) ----
  const int r8 = lane / CPR;
  const int chunk = BK == 64 ? (lane & 7) ^ ((4 * (wave & 1) + (r8 >> 1)) & 7) : (lane & 3) ^ ((r8 >> 2) & 3);
  const bf16_t* a_rows[AI];
  const bf16_t* w_rows[BI];
  {
    const bf16_t* A = (const bf16_t*)p.A + p.a_off + (long)b * p.a_bstride;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int m = m0 + (wave + NW * i) * RPI + r8;
      m = m < p.M ? m : p.M - 1;
      a_rows[i] = A + (long)m * p.lda;
    }
    const bf16_t* W = (const bf16_t*)p.W + (long)b * p.w_bstride;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int n = n0 + (wave + NW * i) * RPI + r8;
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
  {
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = zero16;
  }

  const int l31 = lane & 31, lh = lane >> 5;
  // swizzle of every fragment row of this lane (fragment bases are multiples of 32 rows)
  const int swz = BK == 64 ? (l31 >> 1) & 7 : (l31 >> 2) & 3;
  const int a_base = (wm * WTM + l31) * RB, b_base = (wn * WTN + l31) * RB;

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
    for (int ks = 0; ks < KS; ++ks) {
      const int coff = ((ks * 2 + lh) ^ swz) << 4;
      bf16x8_t af[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8_t*)(sA + a_base + i * 32 * RB + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8_t*)(sB + b_base + j * 32 * RB + coff);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = SA_MFMA_32x32x16(wf[j], af[i], acc[i][j]);
    }
    st_c = st_c + 1 == STAGES ? 0 : st_c + 1;
    st_i = st_i + 1 == STAGES ? 0 : st_i + 1;
  }

  static_assert(NW * FN * 4096 <= STAGES * STAGE, "epilogue staging fits the ring");
  __syncthreads();  // every wave is done with the last slab (no DMA is in flight any more)
  gemm_epilogue_lds<FM, FN>(p, acc, b, m0 + wm * WTM, n0 + wn * WTN, lane, smem + wave * (FN * 4096));
}

// ---- conv7h: dilated k = 7 convolution with the activation HALO TILE RESIDENT in LDS ---------------------------------
// The implicit GEMM above re-stages every activation row once per tap (7x the L2 -> LDS traffic) and, with C <= 192
// output channels, a K-tile is ~0.4 us of MFMA work behind ~2 us of L2 latency (GPU call 13).  Here
// a workgroup stages the (BM + 6 dil) x C halo tile of its BM output rows ONCE (global_load_lds, row stride padded by 16
// bytes against bank conflicts - the LDS image of the DMA is lane-linear, so the padding is produced by the per-lane source
// address), walks the 7 taps by shifting the fragment rows inside LDS, and streams only the weight K-tiles (C x 128 B each)
// through a ring.  Same MFMA (32x32x16, operands swapped), same fragment <-> k mapping and the same K order as the
// implicit GEMM: results are bit for bit those of gemm2_kernel (tests/test_gemm2_gpu.py), so the policy may fall back to
// it for small launches.  8 waves; the whole N = C in one tile.
template <int C, int BM, int WM_, int WN_, int STAGES, int TAG>
__global__ __launch_bounds__(512) void conv7h_kernel(const GemmParams p) {
  constexpr int NW = 8, MAXD = 9;
  static_assert(WM_ * WN_ == NW, "8 waves");
  constexpr int WTM = BM / WM_, WTN = C / WN_;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0 && C % 16 == 0, "tile shape");
  constexpr int HS = C * 2 + 16;                 // padded halo row stride in bytes
  constexpr int CPRH = HS / 16;                  // 16-byte chunks per halo row (the last one is padding)
  constexpr int HROWS = BM + 6 * MAXD;
  constexpr int HALO_B = (HROWS * HS + 1023) / 1024 * 1024;
  constexpr int WR = (C + 63) / 64 * 64;         // weight rows per K-tile as staged (8 waves x 8 rows per instruction)
  constexpr int RB = 128, TILE_W = WR * RB;      // BK = 64
  constexpr int BI = WR / 64;                    // weight DMA instructions per wave per K-tile
  static_assert(HALO_B + STAGES * TILE_W <= 160 * 1024, "LDS budget");
  static_assert(NW * FN * 4096 <= HALO_B + STAGES * TILE_W, "epilogue staging fits");
  static_assert((STAGES - 2) * BI <= 63, "vmcnt range");
  __shared__ __attribute__((aligned(16))) char smem[HALO_B + STAGES * TILE_W];
  char* const halo = smem;
  char* const ring = smem + HALO_B;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_, wn = wave % WN_;
  const int dil = (int)(p.tap_stride / C);

  // workgroup -> (batch item, M-tile): contiguous run of tiles per XCD (neighbouring tiles share halo rows in L2)
  const int tiles_m = (p.M + BM - 1) / BM;
  int b, tm;
  {
    const int total = tiles_m * p.nbatch;
    const int bid = blockIdx.x;
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    b = L / tiles_m;
    tm = L - b * tiles_m;
  }
  const int m0 = tm * BM;

  // ---- halo tile: rows m0 .. m0 + BM + 6 dil - 1 of the (already tap-0-shifted) activation window, contiguous in memory ----
  {
    const bf16_t* A = (const bf16_t*)p.A + p.a_off + (long)b * p.a_bstride;
    const int rows_used = BM + 6 * dil;
    const int last_row = p.M - 1 + 6 * dil;       // last row of the item's window that exists
    const int chunks = rows_used * CPRH;
    for (int c0 = wave * 64; c0 < chunks; c0 += NW * 64) {   // uniform per wave
      const int g = c0 + lane;
      int row = g / CPRH;
      int cc = g - row * CPRH;
      if (cc >= C / 8) cc = 0;                    // the padding chunk: any valid address
      int mr = m0 + row;
      mr = mr < last_row ? mr : last_row;
      dma16(A + (long)mr * C + cc * 8, halo + c0 * 16);
    }
  }
  // ---- weight ring ------------------------------------------------------------------------------------------
  const int r8 = lane >> 3;
  const int wchunk = (lane & 7) ^ ((4 * (wave & 1) + (r8 >> 1)) & 7);
  const bf16_t* w_rows[BI];
  {
    const bf16_t* W = (const bf16_t*)p.W + (long)b * p.w_bstride;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int n = (wave + NW * i) * 8 + r8;
      n = n < p.N ? n : p.N - 1;
      w_rows[i] = W + (long)n * p.K + wchunk * 8;
    }
  }
  auto issue = [&](int stage) {
    char* sB = ring + stage * TILE_W;
#pragma unroll
    for (int i = 0; i < BI; ++i) dma16(w_rows[i], sB + (wave + NW * i) * 1024);
#pragma unroll
    for (int i = 0; i < BI; ++i) w_rows[i] += 64;
  };

  f32x16_t acc[FM][FN];
  {
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = zero16;
  }
  const int l31 = lane & 31, lh = lane >> 5;
  const int swz = (l31 >> 1) & 7;
  const int b_base = (wn * WTN + l31) * RB;
  const char* const a_lane = halo + (wm * WTM + l31) * HS + lh * 16;   // + (i*32 + tap*dil) * HS + c * 2

  const int nslab = p.K / 64;
  const int k_real = 7 * C;                       // columns beyond are the zero padding of W: skipped
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nslab) issue(s);
  int st_c = 0, st_i = (STAGES - 1) % STAGES;
  for (int s = 0; s < nslab; ++s) {
    if (STAGES > 2 && s + STAGES - 2 < nslab) wait_vmcnt<(STAGES - 2) * BI>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // K-tile s (and, at s = 0, the halo tile) visible; everybody done with K-tile s-1's stage
    __builtin_amdgcn_sched_barrier(0);
    if (s + STAGES - 1 < nslab) issue(st_i);
    const char* sB = ring + st_c * TILE_W;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int k = s * 64 + ks * 16;             // the 16 k of this step lie inside one tap (C % 16 == 0)
      if (k < k_real) {                           // uniform
        const int tap = k / C, c = k - tap * C;
        const char* a_k = a_lane + tap * dil * HS + c * 2;
        const int coff = ((ks * 2 + lh) ^ swz) << 4;
        bf16x8_t af[FM], wf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8_t*)(a_k + i * 32 * HS);
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8_t*)(sB + b_base + j * 32 * RB + coff);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = SA_MFMA_32x32x16(wf[j], af[i], acc[i][j]);
      }
    }
    st_c = st_c + 1 == STAGES ? 0 : st_c + 1;
    st_i = st_i + 1 == STAGES ? 0 : st_i + 1;
  }
  __syncthreads();  // every wave is done with the halo tile and the last K-tile (no DMA in flight)
  gemm_epilogue_lds<FM, FN>(p, acc, b, m0 + wm * WTM, wn * WTN, lane, smem + wave * (FN * 4096));
}

// ------------------------------------------------------------------------------------------------
// resunit_kernel: one DAC residual unit  x -> x + conv1(snake(conv7_dil(snake(x))))  per launch (reference codec:
// dacvae ResidualUnit; engine.hip codec_encode / codec_decode issue it as a k7 launch `p` followed by a k1 launch `q`).
// Phase 1 is conv7h_kernel's K loop (halo tile resident, W7 K-tiles through the ring).  Its epilogue (+bias, Snake, bf16
// rounding - the arithmetic of gemm_epilogue_lds) leaves the BM x C tile of the intermediate activation in LDS in the halo
// tile's row layout instead of HBM; phase 2 multiplies it with W1, whose K-tiles follow W7's through the SAME ring (so
// their DMA latency hides behind phase 1's tail), and the ordinary epilogue applies q (+bias, +fp32 residual, fp32 stream
// and Snake'd bf16 copy out).  Same MFMA, same K order, same rounding points as the two launches: bitwise identical
// results (tests/test_gemm2_gpu.py), so a batch may mix fused and unfused launches.  The unit's output activation must
// NOT alias its input (neighbouring tiles still read input halo rows): the engine ping-pongs two buffers.
// Saves the intermediate's write + read (4 of every 16 algorithmic bytes per element) and one launch per unit.
// ------------------------------------------------------------------------------------------------
template <int C, int BM, int WM_, int WN_, int STAGES, int TAG>
__global__ __launch_bounds__(WM_ * WN_ * 64) void resunit_kernel(const GemmParams p, const GemmParams q) {
  constexpr int NW = WM_ * WN_, MAXD = 9;      // 8 waves, or 4 (half the rows per workgroup: two workgroups per CU)
  static_assert(NW == 8 || NW == 4, "4 or 8 waves");
  constexpr int WTM = BM / WM_, WTN = C / WN_;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0 && C % 16 == 0, "tile shape");
  constexpr int HS = C * 2 + 16;
  constexpr int CPRH = HS / 16;
  constexpr int HROWS = BM + 6 * MAXD;
  constexpr int HALO_B = (HROWS * HS + 1023) / 1024 * 1024;
  constexpr int WR = (C + 63) / 64 * 64;
  constexpr int RB = 128, TILE_W = WR * RB;
  constexpr int BI = WR / (NW * 8);              // weight DMA instructions per wave per K-tile (8 rows each)
  static_assert(HALO_B + STAGES * TILE_W <= 160 * 1024, "LDS budget");
  static_assert(NW * FN * 4096 <= HALO_B + STAGES * TILE_W, "epilogue staging fits");
  static_assert((STAGES - 2) * BI <= 63, "vmcnt range");
  __shared__ __attribute__((aligned(16))) char smem[HALO_B + STAGES * TILE_W];
  __shared__ __attribute__((aligned(16))) float colv[2 * C];   // phase 1's bias | Snake alpha (read between the phases)
  char* const halo = smem;
  char* const ring = smem + HALO_B;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_, wn = wave % WN_;
  const int dil = (int)(p.tap_stride / C);

  const int tiles_m = (p.M + BM - 1) / BM;
  int b, tm;
  {
    const int total = tiles_m * p.nbatch;
    const int bid = blockIdx.x;
    const int qd = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int L = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
    b = L / tiles_m;
    tm = L - b * tiles_m;
  }
  const int m0 = tm * BM;
  if (threadIdx.x < C / 4) {   // published by the K loop's first barrier
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), av = bv;
    if (p.bias) bv = ((const float4*)p.bias)[threadIdx.x];
    if (p.act == ACT_SNAKE) av = ((const float4*)p.act_alpha)[threadIdx.x];
    ((float4*)colv)[threadIdx.x] = bv;
    ((float4*)colv)[C / 4 + threadIdx.x] = av;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the table is IN LDS before the (raw) barrier that publishes it

  {  // halo tile of the input activation (as conv7h_kernel)
    const bf16_t* A = (const bf16_t*)p.A + p.a_off + (long)b * p.a_bstride;
    const int rows_used = BM + 6 * dil;
    const int last_row = p.M - 1 + 6 * dil;
    const int chunks = rows_used * CPRH;
    for (int c0 = wave * 64; c0 < chunks; c0 += NW * 64) {
      const int g = c0 + lane;
      int row = g / CPRH;
      int cc = g - row * CPRH;
      if (cc >= C / 8) cc = 0;
      int mr = m0 + row;
      mr = mr < last_row ? mr : last_row;
      dma16a(A + (long)mr * C + cc * 8, halo + c0 * 16);
    }
  }
  // ---- weight ring: the K-tiles of W7 (p.W, row stride p.K) followed by those of W1 (q.W, row stride q.K) ----------
  const int r8 = lane >> 3;
  const int wchunk = (lane & 7) ^ ((4 * (wave & 1) + (r8 >> 1)) & 7);
  const int nslab1 = p.K / 64, nslab2 = q.K / 64, nslab = nslab1 + nslab2;
  const bf16_t* w_rows[BI];
  const bf16_t* w2_rows[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    int n = (wave + NW * i) * 8 + r8;
    n = n < C ? n : C - 1;
    w_rows[i] = (const bf16_t*)p.W + (long)n * p.K + wchunk * 8;
    w2_rows[i] = (const bf16_t*)q.W + (long)n * q.K + wchunk * 8;
  }
  int issued = 0;
  auto issue = [&](int stage) {
    char* sB = ring + stage * TILE_W;
    const bool first = issued < nslab1;  // uniform
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      dma16a(first ? w_rows[i] : w2_rows[i], sB + (wave + NW * i) * 1024);
      w_rows[i] += first ? 64 : 0;
      w2_rows[i] += first ? 0 : 64;
    }
    ++issued;
  };

  f32x16_t acc[FM][FN];
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = zero16;
  const int l31 = lane & 31, lh = lane >> 5;
  const int swz = (l31 >> 1) & 7;
  const int b_base = (wn * WTN + l31) * RB;
  const char* const a_lane = halo + (wm * WTM + l31) * HS + lh * 16;

  const int k_real = 7 * C;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nslab) issue(s);
  int st_c = 0, st_i = (STAGES - 1) % STAGES;
  for (int s = 0; s < nslab; ++s) {
    if (s == nslab1) {
      // ---- between the phases: y = snake(acc + bias) rounded to bf16 -> rows 0 .. BM-1 of the (now free) halo region ----
      // every wave has finished reading the halo tile once it arrives here (its fragment reads completed before its last
      // MFMAs issued); a raw barrier, so that W1's first K-tiles stay in flight
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const bool has_bias = p.bias != nullptr, snake_on = p.act == ACT_SNAKE;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = wn * WTN + j * 32 + 8 * g + 4 * lh;   // 4 consecutive channels of row l31 (gemm_epilogue_lds)
            const float4 bb = *(const float4*)(colv + n), sa = *(const float4*)(colv + C + n);
            float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            if (has_bias) { v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
            // p.act is ACT_SNAKE or ACT_NONE (resunit_ok): act_apply's Snake expression, without its other branches
            const float sv[4] = {sa.x, sa.y, sa.z, sa.w};
            float a[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = snake_on ? snake16_f(v[e], sv[e]) : v[e];
            store4<bf16_t>((bf16_t*)(halo + (wm * WTM + i * 32 + l31) * HS) + n, a[0], a[1], a[2], a[3]);
            acc[i][j][4 * g + 0] = 0.f; acc[i][j][4 * g + 1] = 0.f; acc[i][j][4 * g + 2] = 0.f; acc[i][j][4 * g + 3] = 0.f;
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tile is in LDS before the barrier below publishes it
    }
    if (STAGES > 2 && s + STAGES - 2 < nslab) wait_vmcnt<(STAGES - 2) * BI>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (s + STAGES - 1 < nslab) issue(st_i);
    const char* sB = ring + st_c * TILE_W;
    const bool second = s >= nslab1;               // uniform
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int k = (second ? s - nslab1 : s) * 64 + ks * 16;
      if (k < (second ? C : k_real)) {             // columns beyond are the zero padding of W: skipped
        const int tap = second ? 0 : k / C, c = k - tap * C;
        const char* a_k = a_lane + tap * dil * HS + c * 2;
        const int coff = ((ks * 2 + lh) ^ swz) << 4;
        bf16x8_t af[FM], wf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8_t*)(a_k + i * 32 * HS);
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = *(const bf16x8_t*)(sB + b_base + j * 32 * RB + coff);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = SA_MFMA_32x32x16(wf[j], af[i], acc[i][j]);
      }
    }
    st_c = st_c + 1 == STAGES ? 0 : st_c + 1;
    st_i = st_i + 1 == STAGES ? 0 : st_i + 1;
  }
  __syncthreads();
  gemm_epilogue_lds<FM, FN>(q, acc, b, m0 + wm * WTM, wn * WTN, lane, smem + wave * (FN * 4096));
}

// ------------------------------------------------------------------------------------------------
// resws_kernel: the same residual unit, WEIGHT-STATIONARY and persistent, for launches with many tiles (C = 64 / 96 / 128).
// resunit_kernel at 8 x 480 000 samples spends ~34 us per 128-row tile for ~2 us of MFMA work: 14 K-tiles of 12 MFMAs each
// behind a barrier + DMA wait, then an epilogue whose residual reads the compiler cannot hoist over the stores of the
// previous iteration (they alias: the fp32 stream is updated in place), i.e. one memory round trip per 64 x 16 bytes - the
// kernel has too few bytes in flight to load HBM (2.2 TB/s, profiles/r3_call5/op_bench_prev.log).  Here
//   * one workgroup per CU, C/32 waves, wave w owns output channels [32 w, 32 w + 32) of ALL 128 rows of a tile and keeps
//     its slice of W7 (7 C / 16 fragments) and W1 (C / 16) in REGISTERS for the whole launch (<= 256 VGPRs): no weight
//     ring, no per-K-tile barrier - phase 1 is one straight run of 4 independent MFMAs per fragment;
//   * the workgroup walks tiles; the halo tile of tile t+1 streams into the second halo buffer while tile t computes, and
//     the fp32 residual rows of tile t are requested at the top of the tile, ~3 us before the epilogue consumes them, by
//     DMA into a wave-private LDS area in exactly the (lane, step) order the epilogue reads them back (in registers the
//     compiler shuffled them through AGPR copies that waited for the loads at once): ~90 KiB in flight per CU;
//   * 3 barriers per tile (halo landed | phase-1 reads done | intermediate published).
// C = 128 does not fit (2 halo buffers + 64 KiB of residual rows > 160 KiB) and stays on the ring kernel / two launches.
// Arithmetic, K order and rounding points are those of the two launches: bitwise identical (tests/test_gemm2_gpu.py).
// ------------------------------------------------------------------------------------------------
template <int C, int TAG>
__global__ __launch_bounds__(C / 32 * 64, 1) void resws_kernel(const GemmParams p, const GemmParams q) {
  constexpr int NW = C / 32, BM = 128, FM = 4, MAXD = 9;
  constexpr int HS = C * 2 + 16;
  constexpr int CPRH = HS / 16;
  constexpr int HROWS = BM + 6 * MAXD;
  constexpr int HALO_B = (HROWS * HS + 1023) / 1024 * 1024;
  constexpr int K7S = 7 * C / 16, K1S = C / 16;
  constexpr int RES_B = BM * 32 * 4;   // one wave's residual rows: 128 rows x 32 channels of fp32
  static_assert(C % 32 == 0 && 2 * HALO_B + NW * (4096 + RES_B) + 2 * C * 4 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem[2 * HALO_B + NW * (4096 + RES_B)];
  __shared__ __attribute__((aligned(16))) float colv[2 * C];   // phase 1's bias | Snake alpha
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int dil = (int)(p.tap_stride / C);
  const int tiles_m = (p.M + BM - 1) / BM, total = tiles_m * p.nbatch;
  const int bid = blockIdx.x, G = gridDim.x;
  // a full grid gives every XCD (workgroup id mod 8) 32 consecutive tiles per sweep: neighbours share halo rows in one L2
  auto tile_of = [&](int it) { return G == 256 ? it * 256 + (bid & 7) * 32 + (bid >> 3) : it * G + bid; };
  if (tid < C / 4) {   // published by the first tile's barrier
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), av = bv;
    if (p.bias) bv = ((const float4*)p.bias)[tid];
    if (p.act == ACT_SNAKE) av = ((const float4*)p.act_alpha)[tid];
    ((float4*)colv)[tid] = bv;
    ((float4*)colv)[C / 4 + tid] = av;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the table is IN LDS before the (raw) barrier that publishes it
  const int rows_used = BM + 6 * dil, last_row = p.M - 1 + 6 * dil, chunks = rows_used * CPRH;
  auto issue_halo = [&](int L, char* dst) {   // as conv7h_kernel; rows past the clip re-read its last halo row
    const int hb = L / tiles_m, hm0 = (L - hb * tiles_m) * BM;
    const bf16_t* A = (const bf16_t*)p.A + p.a_off + (long)hb * p.a_bstride;
    for (int c0 = wave * 64; c0 < chunks; c0 += NW * 64) {
      const int g = c0 + lane;
      const int row = g / CPRH;
      int cc = g - row * CPRH;
      if (cc >= C / 8) cc = 0;
      int mr = hm0 + row;
      mr = mr < last_row ? mr : last_row;
      dma16a(A + (long)mr * C + cc * 8, dst + c0 * 16);
    }
  };
  if (tile_of(0) < total) issue_halo(tile_of(0), smem);

  bf16x8_t w7[K7S], w1[K1S];
  {
    const bf16_t* W7 = (const bf16_t*)p.W + (long)(wave * 32 + l31) * p.K + lh * 8;
    const bf16_t* W1 = (const bf16_t*)q.W + (long)(wave * 32 + l31) * q.K + lh * 8;
#pragma unroll
    for (int kk = 0; kk < K7S; ++kk) w7[kk] = *(const bf16x8_t*)(W7 + kk * 16);
#pragma unroll
    for (int kk = 0; kk < K1S; ++kk) w1[kk] = *(const bf16x8_t*)(W1 + kk * 16);
  }
  // q's epilogue operands: a lane keeps columns ncol .. ncol+3 in every staged slice
  const int ncol = wave * 32 + (lane & 7) * 4, erow = lane >> 3;
  const bool q_bias = q.bias != nullptr, q_res = q.res != nullptr, p_bias = p.bias != nullptr, p_snake = p.act == ACT_SNAKE,
             q_snake = q.act == ACT_SNAKE;
  float4 qb = make_float4(0.f, 0.f, 0.f, 0.f), qsa = qb;
  if (q_bias) qb = *(const float4*)(q.bias + ncol);
  if (q.act == ACT_SNAKE) qsa = *(const float4*)(q.act_alpha + ncol);
  char* const stg = smem + 2 * HALO_B + wave * 4096;
  char* const resb = smem + 2 * HALO_B + NW * 4096 + wave * RES_B;
  const int dilHS = dil * HS;
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  int cur = 0;
  for (int it = 0;; ++it, cur ^= 1) {
    const int L = tile_of(it);
    if (L >= total) break;   // uniform; tile_of grows with it
    wait_vmcnt<0>();         // this tile's halo has landed (the compiler does not see dma16's loads: explicit) ...
    __syncthreads();         // ... in every wave
    const int Ln = tile_of(it + 1);
    if (Ln < total) issue_halo(Ln, smem + (cur ^ 1) * HALO_B);
    const int b = L / tiles_m, m0 = (L - b * tiles_m) * BM;
    if (q_res) {   // step (i, t) of the epilogue: lane's row i*32 + t*8 + erow, columns ncol .. ncol+3 -> resb + ((4i + t) * 64 + lane) * 16
      const float* R = q.res + q.res_off + (long)b * q.res_bstride + ncol;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int m = m0 + i * 32 + t * 8 + erow;
          dma16a(R + (long)(m < p.M ? m : p.M - 1) * q.res_ld, resb + (i * 4 + t) * 1024);
        }
    }
    char* const hb = smem + cur * HALO_B;
    const char* const a_lane = hb + l31 * HS + lh * 16;
    f32x16_t acc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = zero16;
    // ---- phase 1: k = 7 dilated convolution, K order tap-major as the ring kernels ------------------------------------
    // One wave per SIMD: nothing but this wave's own instruction stream hides the LDS latency, and left alone the compiler
    // issues read -> wait -> MFMA one at a time (12 us per tile).  Fragments are requested two k-steps ahead into a rotating
    // register set and the scheduler is told the interleave: 4 reads, 4 MFMAs.
    auto halo_frags = [&](int kk, bf16x8_t (&af)[FM]) {
      const int tap = kk * 16 / C, c = kk * 16 - tap * C;
      const char* a_k = a_lane + tap * dilHS + c * 2;
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8_t*)(a_k + i * 32 * HS);
    };
    bf16x8_t af[3][FM];
    halo_frags(0, af[0]);
    halo_frags(1, af[1]);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * FM, 0);   // the two steps requested ahead form a group of their own
#pragma unroll
    for (int kk = 0; kk < K7S; ++kk) {
      if (kk + 2 < K7S) halo_frags(kk + 2, af[(kk + 2) % 3]);
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i] = SA_MFMA_32x32x16(w7[kk], af[kk % 3][i], acc[i]);
      __builtin_amdgcn_sched_group_barrier(0x100, FM, 0);   // DS reads of step kk+2
      __builtin_amdgcn_sched_group_barrier(0x008, FM, 0);   // MFMAs of step kk
    }
    __builtin_amdgcn_s_barrier();   // every wave is through with the halo tile (raw: the prefetches stay in flight)
    __builtin_amdgcn_sched_barrier(0);
    // ---- between the phases: y = snake(acc + bias) rounded to bf16 -> rows 0 .. BM-1 of this tile's halo buffer --------
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = wave * 32 + 8 * g + 4 * lh;
        const float4 bb = *(const float4*)(colv + n), sa = *(const float4*)(colv + C + n);
        float v[4] = {acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
        if (p_bias) { v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
        const float sv[4] = {sa.x, sa.y, sa.z, sa.w};
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = p_snake ? snake16_f(v[e], sv[e]) : v[e];
        store4<bf16_t>((bf16_t*)(hb + (i * 32 + l31) * HS) + n, a[0], a[1], a[2], a[3]);
      }
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = zero16;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2: k = 1 convolution of the intermediate --------------------------------------------------------------
    auto mid_frags = [&](int kk, bf16x8_t (&mf)[FM]) {
#pragma unroll
      for (int i = 0; i < FM; ++i) mf[i] = *(const bf16x8_t*)(a_lane + kk * 32 + i * 32 * HS);
    };
    mid_frags(0, af[0]);
    mid_frags(1, af[1]);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * FM, 0);
#pragma unroll
    for (int kk = 0; kk < K1S; ++kk) {
      if (kk + 2 < K1S) mid_frags(kk + 2, af[(kk + 2) % 3]);
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i] = SA_MFMA_32x32x16(w1[kk], af[kk % 3][i], acc[i]);
      __builtin_amdgcn_sched_group_barrier(0x100, FM, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, FM, 0);
    }
    // ---- q's epilogue (gemm_epilogue_lds with FN = 1, its operands already on chip) -----------------------------------
    wait_vmcnt<0>();   // this wave's residual rows (and its share of the next halo tile) have landed
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(stg + l31 * 128 + (((2 * g + lh) ^ (l31 & 7)) << 4)) =
            make_float4(acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = t * 8 + erow;
        const float4 sv = *(const float4*)(stg + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
        const int m = m0 + i * 32 + row;
        float v[4] = {sv.x, sv.y, sv.z, sv.w};
        if (q_bias) { v[0] += qb.x; v[1] += qb.y; v[2] += qb.z; v[3] += qb.w; }
        if (q_res) {
          const float4 rr = *(const float4*)(resb + ((i * 4 + t) * 64 + lane) * 16);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        // q.act is ACT_SNAKE or ACT_NONE (resunit_ws): act_apply's Snake expression without its other branches
        const float qs[4] = {qsa.x, qsa.y, qsa.z, qsa.w};
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = q_snake ? snake16_f(v[e], qs[e]) : v[e];
        const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
        if (m < p.M) {
          if (q.out_f32) {
            float* frow = q.out_f32 + q.f32_off + (long)b * q.f32_bstride + (long)m * q.f32_ld;
            *(float4*)(frow + ncol) = q.f32_act ? make_float4(a0, a1, a2, a3) : make_float4(v[0], v[1], v[2], v[3]);
          }
          bf16_t* arow = (bf16_t*)q.out_act + q.act_off + (long)b * q.act_bstride + (long)m * q.act_ld;
          store4<bf16_t>(arow + ncol, a0, a1, a2, a3);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// the launches conv7h covers: k = 7 'same' convolution of C -> C channels as the engine issues it (conv_same(): kc = lda =
// C, tap_stride = dil * C, W tap-major with K = 7 C rounded up to 64), dilation <= 9, bf16, no per-batch weights
bool conv7h_ok(const GemmParams& p) {
  if (!(p.N == 64 || p.N == 96 || p.N == 128 || p.N == 192)) return false;
  if (p.kc != p.N || p.lda != p.kc || p.w_bstride != 0 || p.swiglu) return false;
  if (p.tap_stride <= 0 || p.tap_stride % p.kc || p.tap_stride / p.kc > 9) return false;
  if (p.K != (7 * p.kc + 63) / 64 * 64) return false;
  return gemm2_ok(p);
}

template <int C, int BM, int WM_, int WN_, int STAGES>
static hipError_t launch_c7(const GemmParams& p, hipStream_t st) {
  const long tiles = (long)((p.M + BM - 1) / BM) * p.nbatch;
  if (p.tag == 1)
    hipLaunchKernelGGL((conv7h_kernel<C, BM, WM_, WN_, STAGES, 1>), dim3((unsigned)tiles), dim3(512), 0, st, p);
  else
    hipLaunchKernelGGL((conv7h_kernel<C, BM, WM_, WN_, STAGES, 0>), dim3((unsigned)tiles), dim3(512), 0, st, p);
  return hipGetLastError();
}

hipError_t launch_conv7h(const GemmParams& p, hipStream_t st) {
  switch (p.N) {
    case 64: return launch_c7<64, 256, 8, 1, 3>(p, st);     // halo 44 KiB + 3 x 8 KiB
    case 96: return launch_c7<96, 256, 8, 1, 3>(p, st);     // halo 63 KiB + 3 x 16 KiB
    case 128: return launch_c7<128, 256, 4, 2, 3>(p, st);   // halo 83 KiB + 3 x 16 KiB
    case 192: return launch_c7<192, 128, 4, 2, 3>(p, st);   // halo 72 KiB + 3 x 24 KiB
    default: return hipErrorInvalidValue;
  }
}

// a (k7 launch p, k1 launch q) pair the fused residual-unit kernel covers: p as conv7h_ok, q the k1 convolution of the
// same rows and channels reading p's output, with a residual stream and an output activation that is not p's input
bool resunit_ok(const GemmParams& p, const GemmParams& q) {
  if (!conv7h_ok(p) || !gemm2_ok(q)) return false;
  if (q.N != p.N || q.M != p.M || q.nbatch != p.nbatch || q.kc != p.N || q.lda != p.N || q.w_bstride != 0) return false;
  if (q.K != (p.N + 63) / 64 * 64 || q.swiglu || q.gate || q.chan_mod || q.c_ld_rel || q.alpha != 1.f) return false;
  if (p.gate || p.chan_mod || p.c_ld_rel || p.res || p.out_f32 || p.f32_act || !p.out_act) return false;
  if (p.act != ACT_SNAKE && p.act != ACT_NONE) return false;
  // q reads exactly what p writes (the intermediate never reaches memory in the fused form)
  if (q.A != p.out_act || q.a_off != p.act_off || q.a_bstride != p.act_bstride || p.act_ld != p.N) return false;
  if (!q.out_act || q.out_act == p.A) return false;
  return true;
}

template <int C, int BM, int WM_, int WN_, int STAGES>
static hipError_t launch_ru(const GemmParams& p, const GemmParams& q, hipStream_t st) {
  const long tiles = (long)((p.M + BM - 1) / BM) * p.nbatch;
  if (p.tag == 1)
    hipLaunchKernelGGL((resunit_kernel<C, BM, WM_, WN_, STAGES, 1>), dim3((unsigned)tiles), dim3(WM_ * WN_ * 64), 0, st, p, q);
  else
    hipLaunchKernelGGL((resunit_kernel<C, BM, WM_, WN_, STAGES, 0>), dim3((unsigned)tiles), dim3(WM_ * WN_ * 64), 0, st, p, q);
  return hipGetLastError();
}

template <int C>
static hipError_t launch_ws(const GemmParams& p, const GemmParams& q, long tiles, hipStream_t st) {
  const long cap = debug_flag(19) == 3 ? 3 : 256;                // one workgroup per CU, walking tiles
  const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
  if (p.tag == 1)
    hipLaunchKernelGGL((resws_kernel<C, 1>), dim3(grid), dim3(C / 32 * 64), 0, st, p, q);
  else
    hipLaunchKernelGGL((resws_kernel<C, 0>), dim3(grid), dim3(C / 32 * 64), 0, st, p, q);
  return hipGetLastError();
}

// 96-channel launches of >= 1024 tiles of 128 rows (four sweeps of the chip) run the weight-stationary kernel; flag 19 = whatever
// the launch size (its tests; 3 = the same on a grid of 3 workgroups, so that small cases walk several tiles), 2 = never
static bool resunit_ws(const GemmParams& p, const GemmParams& q) {
  if (!(p.N == 64 || p.N == 96) || debug_flag(19) == 2) return false;
  if (q.act != ACT_SNAKE && q.act != ACT_NONE) return false;
  if (debug_flag(19) == 1 || debug_flag(19) == 3) return true;
  // 96 channels: 1 697 vs 1 939 us for 8 waveforms (ring kernel), profiles/r3_call12/op_bench.log.  64 channels would run
  // two waves per CU - too few to cover its LDS / VALU latencies: 1 564 vs 923 us - and stays on the ring kernel.
  return p.N == 96 && (long)((p.M + 127) / 128) * p.nbatch >= 1024;
}

hipError_t launch_resunit(const GemmParams& p, const GemmParams& q, hipStream_t st) {
  if (resunit_ws(p, q)) {
    const long tiles = (long)((p.M + 127) / 128) * p.nbatch;
    return p.N == 64 ? launch_ws<64>(p, q, tiles, st) : launch_ws<96>(p, q, tiles, st);
  }
  switch (p.N) {  // tile shapes of launch_conv7h
    case 64: return launch_ru<64, 256, 8, 1, 3>(p, q, st);
    // 96 channels: 128-row tiles on 4 waves and a 2-stage ring = 72 KiB, two workgroups per CU, so that one workgroup's
    // memory-bound phase-2 epilogue overlaps the other's MFMA-bound phase 1 (2023 vs 2102 us for 8 waveforms, two launches
    // 2204; profiles/r2_call21/).  Flag 20 = the 8-wave 256-row shape.
    case 96: return launch_ru<96, 128, 4, 1, 2>(p, q, st);   // 72 KiB: two workgroups per CU (2 023 vs 2 127 us on 256 rows / 8 waves)
    case 128: return launch_ru<128, 256, 4, 2, 3>(p, q, st);
    case 192: return launch_ru<192, 128, 4, 2, 3>(p, q, st);
    default: return hipErrorInvalidValue;
  }
}

template <int BM, int BN, int WM_, int WN_, int STAGES, int BK, bool TAGGED = false>
static hipError_t launch2(const GemmParams& p, hipStream_t st) {
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.nbatch;
  if (TAGGED && p.tag == 1)
    hipLaunchKernelGGL((gemm2_kernel<BM, BN, WM_, WN_, STAGES, BK, TAGGED ? 1 : 0>), dim3((unsigned)tiles),
                       dim3(WM_ * WN_ * 64), 0, st, p);
  else
    hipLaunchKernelGGL((gemm2_kernel<BM, BN, WM_, WN_, STAGES, BK, 0>), dim3((unsigned)tiles), dim3(WM_ * WN_ * 64), 0,
                       st, p);
  return hipGetLastError();
}

// Can this problem take the vectorised-epilogue kernels?  (everything else stays on gemm.hip)
bool gemm2_ok(const GemmParams& p) {
  if ((p.chan_mod & 3) || (p.c_ld_rel & 3) || (p.c_lo & 3) || (p.c_hi & 3)) return false;
  if (p.act == ACT_SNAKE && (!p.act_alpha || ((uintptr_t)p.act_alpha & 15))) return false;
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

// variants: 0 = 256x128 3-stage ring, 1 = 256x128 2-stage ring, 2 = 256x256 2-stage ring (8 waves, BK 64, one
// workgroup per CU); 3 = 256x192 2-stage ring; 6 = 256x256 role-split (half-slab phases, 2 stages).
// Measured and dropped (profiles/r1_gemm_variants_*.log): 4-wave BK-32 tiles with two workgroups per CU (3-5),
// 256x128 role-split with 3 stages (7, 8), 256x256 with one 512-register wave per SIMD (12), hand-pipelined asm
// fragment reads (13, 14) - all within +-3 % of the kept kernels or slower.  Ablation builds (9-11: no DMA / no MFMA /
// no LDS reads; wrong results, timing only) compile with -DSAMAUDIO_GEMM_ABLATIONS.
// variant = gemm_variant()'s number - 3 (gemm.hip): only the tiles the policy selects are built
hipError_t launch_gemm2(const GemmParams& p, int variant, hipStream_t st) {
  switch (variant) {
    case 19: return launch_gemm8(p, st);   // gemm8.hip: 256x256 tile, the guide's 8-phase K loop, 16x16x32 MFMA
    case 24: return launch_gemm8s(p, st);  // gemm8.hip: 128x128 tile of the same arithmetic (few rows; split tails)
    // 32x32x16 family (DAC-VAE stages with 64 - 192 channels): M-aware tiles, bitwise equal among themselves
    case 22: return launch2<128, 128, 2, 2, 2, 64>(p, st);  // few rows: 4 waves, 64 KiB => two workgroups per CU
    case 23: return launch2<64, 128, 1, 4, 3, 64>(p, st);   // fewer rows: 4 waves, 72 KiB => two workgroups per CU
    case 25: return launch2<256, 64, 8, 1, 2, 64, true>(p, st);  // N = 64 outputs (first DAC encoder stage) in one 64-wide tile
    case 26: return launch2<128, 128, 2, 2, 3, 32>(p, st);  // BK 32: 48 KiB => three workgroups per CU, 3 stages each
    case 29: return launch2<128, 64, 2, 2, 2, 32>(p, st);   // N <= 64: 24 KiB => six workgroups per CU
    case 30: return launch2<64, 128, 1, 4, 3, 32>(p, st);   // 36 KiB => four workgroups per CU
    case 31: return launch2<128, 192, 2, 2, 3, 32>(p, st);  // N = 192 in one tile, 60 KiB => two workgroups per CU
    case 32: return launch_conv7h(p, st);  // k7 convolution with the halo tile resident in LDS (conv7h_ok launches only)
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sa
