// Attention kernels of the DiT block (reference transformer.py:153-160: non-causal SDPA, bool key-padding
// mask, scale 1/sqrt(128)).  Sequence lengths on this path are short (T = 250 latent frames for a 10 s clip),
// so one workgroup owns 64 query rows of one (batch, head) and streams the keys through LDS in tiles of 64.
//
//   * bf16 mode: flash-style, both contractions on v_mfma_f32_16x16x32_bf16.  S = Q K^T with Q fragments
//     held in registers; the softmax runs in the accumulator layout (row statistics by 16-lane shuffles);
//     (16-lane DPP row reductions, no LDS round trip); P is re-laid as an A operand through a per-wave LDS tile; V arrives pre-transposed ([d][key]) so that
//     P@V is again a K-contiguous contraction.  LDS tiles are XOR-swizzled against ds_read_b128 conflicts.
//   * fp32 mode (parity path): same tiling on the vector ALU, exact fp32 with expf.
//   * cross-attention (Lt ~ a handful of T5 tokens): one wave per (row, head), q-norm fused, online softmax.
#include "kernels.h"

namespace sa {

typedef h16x8_t bf16x8_t;  // 8 x 16-bit operand words (bf16, or fp16 with -DSA_OPERAND_FP16: common.h)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---------------------------------------------------------------------------------------------------
// bf16 MFMA self-attention.  grid (Tp/64, H, B), 256 threads (4 waves x 16 query rows).
// ---------------------------------------------------------------------------------------------------
// NW waves x 16 query rows per workgroup: NW = 8 (128 rows) halves the K / V^T staging traffic and barrier count per
// query row; it needs Tp % 128 == 0 (the launcher falls back to NW = 4 otherwise).  HD = head dim: 128 (DiT, PE-AV
// transformers, the vision tower's pooling head) or 64 (PE-Core vision tower blocks); scale = HD^-0.5.
// (Round 2 tried three restructurings of this kernel on hardware, all slower or flat, profiles/r2_call9/ and r2_call11/:
// a register prefetch of the next K / V^T tile - the compiler sinks the loads back behind the barrier -, DMA double
// buffering with one barrier per tile and the key mask staged through LDS (79 vs 68 us), and both together.  At T = 250
// the kernel moves 275 MB for 23.6 GFLOP in 68 us = 4 TB/s: it is bandwidth-bound, not latency-bound.)
// OUT_ALT: the context rows leave in the alt 16-bit format (mixed mode: they feed the wo GEMM on bf16 operands)
template <int NW, int HD, bool OUT_ALT = false>
__global__ __launch_bounds__(NW * 64, 4) void self_attn_bf16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                 const bf16_t* __restrict__ Vt,
                                                                 const unsigned char* __restrict__ key_mask,
                                                                 bf16_t* __restrict__ out, int T, int Tp, int H) {
  constexpr int CH = HD / 8;       // 16-byte chunks per K row
  constexpr int KS = HD / 32;      // k-steps of the S = Q K^T contraction
  constexpr int NF = HD / 16;      // output fragments (16 head channels each)
  __shared__ __attribute__((aligned(16))) char Ks[64 * HD * 2];   // [key][HD d] bf16, chunk ^= key & (CH - 1)
  __shared__ __attribute__((aligned(16))) char Vs[HD * 128];      // [d][64 keys] bf16, chunk ^= (d >> 1) & 7
  __shared__ __attribute__((aligned(16))) char Ps[NW * 16 * 128];  // per wave [16 q][64 keys] bf16
  // (Round 2, call 28: a 1-D grid dealing the query blocks of a (batch, head) onto one XCD, so that the second block finds
  // K / V^T in that L2 - PMC: 225 MB fetched per launch for 135 MB of operands - was bitwise identical, 65.2 vs 66.8 us in
  // isolation and nothing end to end; a 16-wave / 256-query-row workgroup measured slower, 72.8 vs 66.4 us.  Both removed.)
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qb * (16 * NW);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const long bh = (long)b * H + h;
  const float scale = HD == 128 ? 0.08838834764831845f : 0.125f;  // 1/sqrt(HD)

  bf16x8_t qf[KS];
  {
    const bf16_t* qrow = Q + (bh * Tp + q0 + wave * 16 + lr) * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8_t*)(qrow + (ks * 4 + lg) * 8);
  }
  // Key validity (the key exists and is not masked) for the whole (padded) sequence goes to LDS once, coalesced, and the K
  // loop reads its four bytes per lane from there.  Read from global memory where it is used - under `key < T &&`, after the
  // S MFMAs - each of a tile's four mask bytes was a branch around a dependent global load with its own s_waitcnt vmcnt(0):
  // four L2 round trips in series in every K-tile.  (Sequences beyond MAXT keys keep the global reads.)
  constexpr int MAXT = 2048;
  __shared__ unsigned char Ms[MAXT];
  const bool mask_in_lds = Tp <= MAXT;   // uniform
  if (mask_in_lds)
    for (int i = tid; i < Tp; i += NW * 64) Ms[i] = (i < T && key_mask[(long)b * T + (i < T ? i : 0)] != 0) ? 1 : 0;
  // (published by the two barriers every K-tile starts with)
  float m_i[4], l_i[4];
  f32x4_t o[NF];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_i[r] = -INFINITY; l_i[r] = 0.f; }
#pragma unroll
  for (int n = 0; n < NF; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  char* Pw = Ps + wave * 2048;

  for (int kt = 0; kt < Tp; kt += 64) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < HD / (8 * NW); ++it) {
      const int idx = tid + 64 * NW * it;
      {
        const int row = idx / CH, c = idx % CH;
        const uint4 v = *(const uint4*)(K + (bh * Tp + kt + row) * HD + c * 8);
        *(uint4*)(Ks + row * (HD * 2) + ((c ^ (row & (CH - 1))) << 4)) = v;
      }
      {
        const int d = idx >> 3, c = idx & 7;
        const uint4 v = *(const uint4*)(Vt + (bh * HD + d) * Tp + kt + c * 8);
        *(uint4*)(Vs + d * 128 + ((c ^ ((d >> 1) & 7)) << 4)) = v;
      }
    }
    __syncthreads();
    // S = Q K^T : s[nb][r] = S[q = lg*4 + r][key = nb*16 + lr]
    f32x4_t s[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      s[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const int row = nb * 16 + lr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t kf = *(const bf16x8_t*)(Ks + row * (HD * 2) + (((ks * 4 + lg) ^ (row & (CH - 1))) << 4));
        s[nb] = SA_MFMA_16x16x32(qf[ks], kf, s[nb]);
      }
    }
    bool valid[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int key = kt + nb * 16 + lr;
      valid[nb] = mask_in_lds ? Ms[key] != 0 : (key < T && key_mask[(long)b * T + key] != 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = -INFINITY;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        s[nb][r] = valid[nb] ? s[nb][r] * scale : -INFINITY;
        mx = fmaxf(mx, s[nb][r]);
      }
      mx = row16_max(mx);
      const float m_new = fmaxf(m_i[r], mx);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = __expf(m_i[r] - m_safe);
      float rs = 0.f;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const float pv = __expf(s[nb][r] - m_safe);
        s[nb][r] = pv;
        rs += pv;
      }
      rs = row16_sum(rs);
      l_i[r] = l_i[r] * alpha + rs;
      m_i[r] = m_new;
#pragma unroll
      for (int n = 0; n < NF; ++n) o[n][r] *= alpha;
      // P -> LDS as an A operand image: row q = lg*4 + r, key = nb*16 + lr
      const int q = lg * 4 + r;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int c = nb * 2 + (lr >> 3);
        *(unsigned short*)(Pw + q * 128 + ((c ^ ((q >> 1) & 7)) << 4) + (lr & 7) * 2) = f2bf(s[nb][r]);
      }
    }
    __syncthreads();
    // O += P V : A = P[q = lr][key chunk], B = Vt[d = n*16 + lr][key chunk]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + lg;
      const bf16x8_t pf = *(const bf16x8_t*)(Pw + lr * 128 + ((c ^ ((lr >> 1) & 7)) << 4));
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int d = n * 16 + lr;
        const bf16x8_t vf = *(const bf16x8_t*)(Vs + d * 128 + ((c ^ ((d >> 1) & 7)) << 4));
        o[n] = SA_MFMA_16x16x32(pf, vf, o[n]);
      }
    }
  }
  // Output: a lane holds column n*16 + lr of rows lg*4 + r - stored from there, every element is a 2-byte store and an
  // instruction covers four 32-byte pieces.  The wave's 16 x HD tile goes through LDS instead (a private slice of the K / V^T
  // staging area, free after the loop) and leaves as 16-byte stores, 16 lanes per 256-byte row.
  const int D = H * HD;
  __syncthreads();   // every wave is through with Ks / Vs
  constexpr int SLICE = 16 * HD * 2;   // one wave's tile: Ks holds four of them, Vs the other four (NW = 8)
  static_assert(4 * SLICE == 64 * HD * 2 && 4 * SLICE <= HD * 128 && NW <= 8, "output staging fits the K / V^T staging areas");
  char* mine = wave < 4 ? Ks + wave * SLICE : Vs + (wave - 4) * SLICE;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float inv = 1.f / l_i[r];
    const int q = lg * 4 + r;
#pragma unroll
    for (int n = 0; n < NF; ++n)
      *(unsigned short*)(mine + q * (HD * 2) + (n * 16 + lr) * 2) = OUT_ALT ? f2alt(o[n][r] * inv) : f2bf(o[n][r] * inv);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private: only wave-level ordering is needed
  constexpr int CPRO = HD / 8;   // 16-byte chunks per output row
#pragma unroll
  for (int u = lane; u < 16 * CPRO; u += 64) {
    const int row = u / CPRO, c = u % CPRO;
    const int q = q0 + wave * 16 + row;
    if (q < T) *(uint4*)(out + ((long)b * T + q) * D + h * HD + c * 8) = *(const uint4*)(mine + row * (HD * 2) + c * 16);
  }
}

// (Round 5, GPU calls 9 - 11: a form of this kernel that reads the qkv GEMM's output rows directly - q/k RMSNorm + RoPE applied on load,
// V transposed on its way into LDS, no qkv_prep pass - was built, passed its parity tests on the simulator and on hardware, and was
// REMOVED: 38 us per launch at 4 clips against 18.5 + 12.5 for the two kernels (138.7 vs 95.3 ms per 32-clip step), 152.1 vs 156.7
// s-audio/s end to end at 4 clips, 231.3 vs 236.4 at 32; and a row's result depended on the batch it was evaluated in
// (tests/test_configs_gpu.py::test_sixty_four_rows_at_large_star, both of its V-transposition forms), which the two kernels do not
// show.  Not root-caused (the first suspect - 2-byte LDS scatter stores of V, 16-way bank conflicts - was replaced by a staging tile
// + column reads: same result).  Logs: profiles/r5_call9/ ... r5_call11/.)
// ---------------------------------------------------------------------------------------------------
// fp32 self-attention (parity path).  grid (ceil(T/32), H, B), 256 threads: thread = (query qi, lane-in-8 sub).
// ---------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void self_attn_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                            const float* __restrict__ Vt,
                                                            const unsigned char* __restrict__ key_mask,
                                                            float* __restrict__ out, int T, int Tp, int H) {
  __shared__ float Qs[32][HD + 1];
  __shared__ float KV[HD * 65];  // K tile as [64][HD + 1] or V^T tile as [HD][65]
  __shared__ float Ss[32][65];
  const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, qi = tid >> 3, sub = tid & 7;
  const long bh = (long)b * H + h;
  const float scale = HD == 128 ? 0.08838834764831845f : 0.125f;
  for (int idx = tid; idx < 32 * HD; idx += 256) {
    const int r = idx / HD, d = idx % HD;
    const int q = q0 + r;
    Qs[r][d] = q < Tp ? Q[(bh * Tp + q) * HD + d] : 0.f;
  }
  float m_i = -INFINITY, l_i = 0.f, o[HD / 8];
#pragma unroll
  for (int i = 0; i < HD / 8; ++i) o[i] = 0.f;
  for (int kt = 0; kt < Tp; kt += 64) {
    __syncthreads();
    for (int idx = tid; idx < 64 * HD; idx += 256) {
      const int r = idx / HD, d = idx % HD;
      KV[r * (HD + 1) + d] = K[(bh * Tp + kt + r) * HD + d];
    }
    __syncthreads();
    float sc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sc[u] = 0.f;
    for (int d = 0; d < HD; ++d) {
      const float qv = Qs[qi][d];
#pragma unroll
      for (int u = 0; u < 8; ++u) sc[u] = fmaf(qv, KV[(sub + 8 * u) * (HD + 1) + d], sc[u]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int key = kt + sub + 8 * u;
      const bool valid = key < T && key_mask[(long)b * T + key] != 0;
      sc[u] = valid ? sc[u] * scale : -INFINITY;
      mx = fmaxf(mx, sc[u]);
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float m_new = fmaxf(m_i, mx);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = expf(m_i - m_safe);
    float rs = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float pv = expf(sc[u] - m_safe);
      Ss[qi][sub + 8 * u] = pv;
      rs += pv;
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
    l_i = l_i * alpha + rs;
    m_i = m_new;
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) o[i] *= alpha;
    __syncthreads();  // scores written, K tile no longer needed
    for (int idx = tid; idx < HD * 64; idx += 256) {
      const int d = idx >> 6, kk = idx & 63;
      KV[d * 65 + kk] = Vt[(bh * HD + d) * Tp + kt + kk];
    }
    __syncthreads();
    for (int kk = 0; kk < 64; ++kk) {
      const float pv = Ss[qi][kk];
#pragma unroll
      for (int i = 0; i < HD / 8; ++i) o[i] = fmaf(pv, KV[(sub + 8 * i) * 65 + kk], o[i]);
    }
  }
  const int q = q0 + qi;
  if (q < T) {
    const float inv = 1.f / l_i;
    float* orow = out + ((long)b * T + q) * (H * HD) + h * HD;
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) orow[sub + 8 * i] = o[i] * inv;
  }
}

// ---------------------------------------------------------------------------------------------------
// Compensated MFMA self-attention of an fp32 context (SAMAUDIO_OPT_X3_CLASSES bit SAMAUDIO_X3_ATTENTION; reference
// transformer.py:153-160).  fp32 Q, K [B,H,Tp,HD] and V^T [B,H,HD,Tp] in, fp32 context rows out, as self_attn_f32_kernel - but
// both contractions run on the 16-bit MFMA over hi/lo-split operands (x = rn16(x) + rn16(x - rn16(x)), split on the way into
// LDS / registers):  S = Ql Kh + Qh Kl + Qh Kh,  O = Pl Vh + Ph Vl + Ph Vh  - each the fp32 product to ~2^-21 (IEEE half), small
// terms first, fp32 accumulation; softmax statistics in fp32 as in the 16-bit kernel.  The VALU kernel it replaces ran at 18 TF/s
// (810 ms of a 5.7 s step at 32 clips, profiles/r6_call1/).  Tiling = self_attn_bf16_kernel: NW waves x 16 query rows, keys in tiles
// of 64; LDS = hi + lo images of the K tile and of the V^T tile (4 x 16 KiB at HD 128) + per-wave P tiles.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8(const float4 a, const float4 b, uint4& hi, uint4& lo) {
#pragma clang fp contract(off)
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = pack_h16x2(v[2 * e], v[2 * e + 1]);
    l[e] = pack_h16x2(v[2 * e] - h16_lo(h[e]), v[2 * e + 1] - h16_hi(h[e]));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// OUT3: the context rows leave as the NEXT GEMM's compensated operand, out3 [B*T, 3 D] 16-bit = [lo | hi | hi] (wo on split operands),
// instead of fp32 rows
template <int NW, int HD, bool OUT3 = false>
__global__ __launch_bounds__(NW * 64) void self_attn_x3_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                               const float* __restrict__ Vt,
                                                               const unsigned char* __restrict__ key_mask,
                                                               float* __restrict__ out, bf16_t* __restrict__ out3, int T, int Tp, int H) {
  constexpr int CH = HD / 8;       // 8-element chunks per K row
  constexpr int KS = HD / 32;      // k-steps of S = Q K^T
  constexpr int NF = HD / 16;      // output fragments
  constexpr int KB = 64 * HD * 2;  // bytes of one K-tile image; a V^T-tile image is HD * 128 = the same
  __shared__ __attribute__((aligned(16))) char Ks[2 * KB];            // [hi | lo][key][HD d], chunk ^= key & (CH - 1)
  __shared__ __attribute__((aligned(16))) char Vs[2 * KB];            // [hi | lo][d][64 keys], chunk ^= (d >> 1) & 7
  __shared__ __attribute__((aligned(16))) char Ps[2 * NW * 16 * 128];  // [hi | lo] per wave [16 q][64 keys]
  constexpr int MAXT = 2048;
  __shared__ unsigned char Ms[MAXT];
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qb * (16 * NW);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const long bh = (long)b * H + h;
  const float scale = HD == 128 ? 0.08838834764831845f : 0.125f;

  bf16x8_t qh[KS], ql[KS];
  {
    int qr = q0 + wave * 16 + lr;
    qr = qr < Tp ? qr : Tp - 1;
    const float* qrow = Q + (bh * Tp + qr) * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float* src = qrow + (ks * 4 + lg) * 8;
      uint4 hi, lo;
      split8(*(const float4*)src, *(const float4*)(src + 4), hi, lo);
      qh[ks] = __builtin_bit_cast(bf16x8_t, hi);
      ql[ks] = __builtin_bit_cast(bf16x8_t, lo);
    }
  }
  const bool mask_in_lds = Tp <= MAXT;   // uniform
  if (mask_in_lds)
    for (int i = tid; i < Tp; i += NW * 64) Ms[i] = (i < T && key_mask[(long)b * T + (i < T ? i : 0)] != 0) ? 1 : 0;
  float m_i[4], l_i[4];
  f32x4_t o[NF];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_i[r] = -INFINITY; l_i[r] = 0.f; }
#pragma unroll
  for (int n = 0; n < NF; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  char* Pw = Ps + wave * 2048;
  constexpr int PL = NW * 2048;   // offset of the lo image of P

  // the fp32 K / V^T values of a key tile travel global -> registers -> (split) -> LDS; the NEXT tile's values are requested right
  // after this tile's went into LDS, so that their latency runs underneath this tile's MFMAs and softmax (round 6: the loads used to
  // sit between the two barriers at the top of the tile, exposed once per tile with one workgroup per CU)
  constexpr int NIT = HD / (8 * NW);
  float4 kreg[NIT][2], vreg[NIT][2];
  auto request = [&](int kt) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 64 * NW * it;
      const float* ksrc = K + (bh * Tp + kt + idx / CH) * HD + (idx % CH) * 8;
      kreg[it][0] = *(const float4*)ksrc;
      kreg[it][1] = *(const float4*)(ksrc + 4);
      const float* vsrc = Vt + (bh * HD + (idx >> 3)) * Tp + kt + (idx & 7) * 8;
      vreg[it][0] = *(const float4*)vsrc;
      vreg[it][1] = *(const float4*)(vsrc + 4);
    }
  };
  request(0);
  for (int kt = 0; kt < Tp; kt += 64) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 64 * NW * it;
      {
        const int row = idx / CH, c = idx % CH;
        uint4 hi, lo;
        split8(kreg[it][0], kreg[it][1], hi, lo);
        const int off = row * (HD * 2) + ((c ^ (row & (CH - 1))) << 4);
        *(uint4*)(Ks + off) = hi;
        *(uint4*)(Ks + KB + off) = lo;
      }
      {
        const int d = idx >> 3, c = idx & 7;
        uint4 hi, lo;
        split8(vreg[it][0], vreg[it][1], hi, lo);
        const int off = d * 128 + ((c ^ ((d >> 1) & 7)) << 4);
        *(uint4*)(Vs + off) = hi;
        *(uint4*)(Vs + KB + off) = lo;
      }
    }
    if (kt + 64 < Tp) request(kt + 64);
    __syncthreads();
    f32x4_t s[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      s[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const int row = nb * 16 + lr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = row * (HD * 2) + (((ks * 4 + lg) ^ (row & (CH - 1))) << 4);
        const bf16x8_t kh = *(const bf16x8_t*)(Ks + off), kl = *(const bf16x8_t*)(Ks + KB + off);
        s[nb] = SA_MFMA_16x16x32(ql[ks], kh, s[nb]);
        s[nb] = SA_MFMA_16x16x32(qh[ks], kl, s[nb]);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = row * (HD * 2) + (((ks * 4 + lg) ^ (row & (CH - 1))) << 4);
        s[nb] = SA_MFMA_16x16x32(qh[ks], *(const bf16x8_t*)(Ks + off), s[nb]);
      }
    }
    bool valid[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int key = kt + nb * 16 + lr;
      valid[nb] = mask_in_lds ? Ms[key] != 0 : (key < T && key_mask[(long)b * T + key] != 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = -INFINITY;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        s[nb][r] = valid[nb] ? s[nb][r] * scale : -INFINITY;
        mx = fmaxf(mx, s[nb][r]);
      }
      mx = row16_max(mx);
      const float m_new = fmaxf(m_i[r], mx);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = expf(m_i[r] - m_safe);
      float rs = 0.f;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const float pv = expf(s[nb][r] - m_safe);
        s[nb][r] = pv;
        rs += pv;
      }
      rs = row16_sum(rs);
      l_i[r] = l_i[r] * alpha + rs;
      m_i[r] = m_new;
#pragma unroll
      for (int n = 0; n < NF; ++n) o[n][r] *= alpha;
      const int q = lg * 4 + r;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int c = nb * 2 + (lr >> 3);
        const int off = q * 128 + ((c ^ ((q >> 1) & 7)) << 4) + (lr & 7) * 2;
        const unsigned short ph = f2bf(s[nb][r]);
        *(unsigned short*)(Pw + off) = ph;
        *(unsigned short*)(Pw + PL + off) = f2bf(s[nb][r] - bf2f(ph));
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + lg;
      const int poff = lr * 128 + ((c ^ ((lr >> 1) & 7)) << 4);
      const bf16x8_t ph = *(const bf16x8_t*)(Pw + poff), pl = *(const bf16x8_t*)(Pw + PL + poff);
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int d = n * 16 + lr;
        const int voff = d * 128 + ((c ^ ((d >> 1) & 7)) << 4);
        const bf16x8_t vh = *(const bf16x8_t*)(Vs + voff), vl = *(const bf16x8_t*)(Vs + KB + voff);
        o[n] = SA_MFMA_16x16x32(pl, vh, o[n]);
        o[n] = SA_MFMA_16x16x32(ph, vl, o[n]);
        o[n] = SA_MFMA_16x16x32(ph, vh, o[n]);
      }
    }
  }
  // fp32 context rows through LDS (a wave-private 16 x HD fp32 slice of the K / V^T staging areas): 16-byte stores, whole lines
  const int D = H * HD;
  __syncthreads();   // every wave is through with Ks / Vs
  constexpr int SLICE = 16 * HD * 4;
  static_assert(4 * SLICE <= 2 * KB && NW <= 8, "output staging fits the K / V^T staging areas");
  char* mine = wave < 4 ? Ks + wave * SLICE : Vs + (wave - 4) * SLICE;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float inv = 1.f / l_i[r];
    const int q = lg * 4 + r;
#pragma unroll
    for (int n = 0; n < NF; ++n) *(float*)(mine + q * (HD * 4) + (n * 16 + lr) * 4) = o[n][r] * inv;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private: only wave-level ordering is needed
  if constexpr (OUT3) {
    constexpr int CPR8 = HD / 8;   // 8-element chunks per output row: two 16-byte LDS reads, three 16-byte stores
#pragma unroll
    for (int u = lane; u < 16 * CPR8; u += 64) {
      const int row = u / CPR8, c = u % CPR8;
      const int q = q0 + wave * 16 + row;
      const float* src = (const float*)(mine + row * (HD * 4) + c * 32);
      uint4 hi, lo;
      split8(*(const float4*)src, *(const float4*)(src + 4), hi, lo);
      if (q < T) {
        bf16_t* dst = out3 + ((long)b * T + q) * (3L * D) + h * HD + c * 8;
        *(uint4*)dst = lo;
        *(uint4*)(dst + D) = hi;
        *(uint4*)(dst + 2L * D) = hi;
      }
    }
    return;
  }
  constexpr int CPRO = HD / 4;   // 16-byte chunks per output row
#pragma unroll
  for (int u = lane; u < 16 * CPRO; u += 64) {
    const int row = u / CPRO, c = u % CPRO;
    const int q = q0 + wave * 16 + row;
    if (q < T) *(uint4*)(out + ((long)b * T + q) * D + h * HD + c * 4) = *(const uint4*)(mine + row * (HD * 4) + c * 16);
  }
}

template <int HD>
static hipError_t launch_self_attention_t(const void* Q, const void* K, const void* Vt, const unsigned char* key_mask,
                                          void* out, bool bf16, int B, int T, int Tp, int H, hipStream_t st, bool out_alt = false) {
  if (bf16 && out_alt && Tp % 128 == 0)
    hipLaunchKernelGGL((self_attn_bf16_kernel<8, HD, true>), dim3(Tp / 128, H, B), dim3(512), 0, st, (const bf16_t*)Q, (const bf16_t*)K,
                       (const bf16_t*)Vt, key_mask, (bf16_t*)out, T, Tp, H);
  else if (bf16 && out_alt)
    hipLaunchKernelGGL((self_attn_bf16_kernel<4, HD, true>), dim3(Tp / 64, H, B), dim3(256), 0, st, (const bf16_t*)Q, (const bf16_t*)K,
                       (const bf16_t*)Vt, key_mask, (bf16_t*)out, T, Tp, H);
  else if (out_alt)
    return hipErrorInvalidValue;
  else if (bf16 && Tp % 128 == 0)
    hipLaunchKernelGGL((self_attn_bf16_kernel<8, HD>), dim3(Tp / 128, H, B), dim3(512), 0, st, (const bf16_t*)Q, (const bf16_t*)K,
                       (const bf16_t*)Vt, key_mask, (bf16_t*)out, T, Tp, H);
  else if (bf16)
    hipLaunchKernelGGL((self_attn_bf16_kernel<4, HD>), dim3(Tp / 64, H, B), dim3(256), 0, st, (const bf16_t*)Q, (const bf16_t*)K,
                       (const bf16_t*)Vt, key_mask, (bf16_t*)out, T, Tp, H);
  else
    hipLaunchKernelGGL(self_attn_f32_kernel<HD>, dim3((T + 31) / 32, H, B), dim3(256), 0, st, (const float*)Q,
                       (const float*)K, (const float*)Vt, key_mask, (float*)out, T, Tp, H);
  return hipGetLastError();
}

template <int NW, int HD>
static void launch_self_attention_x3_t(const float* Q, const float* K, const float* Vt, const unsigned char* key_mask, float* out,
                                       void* out3, dim3 grid, int T, int Tp, int H, hipStream_t st) {
  if (out3)
    hipLaunchKernelGGL((self_attn_x3_kernel<NW, HD, true>), grid, dim3(NW * 64), 0, st, Q, K, Vt, key_mask, out, (bf16_t*)out3, T, Tp, H);
  else
    hipLaunchKernelGGL((self_attn_x3_kernel<NW, HD, false>), grid, dim3(NW * 64), 0, st, Q, K, Vt, key_mask, out, (bf16_t*)nullptr, T, Tp, H);
}
hipError_t launch_self_attention_x3(const float* Q, const float* K, const float* Vt, const unsigned char* key_mask, float* out,
                                    int B, int T, int Tp, int H, int head_dim, hipStream_t st, void* out3) {
  if (Tp % 64 || (head_dim != 64 && head_dim != 128) || (!out && !out3)) return hipErrorInvalidValue;
  const bool wide = Tp % 128 == 0;
  const dim3 grid(wide ? Tp / 128 : Tp / 64, H, B);
  if (head_dim == 128 && wide) launch_self_attention_x3_t<8, 128>(Q, K, Vt, key_mask, out, out3, grid, T, Tp, H, st);
  else if (head_dim == 128) launch_self_attention_x3_t<4, 128>(Q, K, Vt, key_mask, out, out3, grid, T, Tp, H, st);
  else if (wide) launch_self_attention_x3_t<8, 64>(Q, K, Vt, key_mask, out, out3, grid, T, Tp, H, st);
  else launch_self_attention_x3_t<4, 64>(Q, K, Vt, key_mask, out, out3, grid, T, Tp, H, st);
  return hipGetLastError();
}

hipError_t launch_self_attention(const void* Q, const void* K, const void* Vt, const unsigned char* key_mask,
                                 void* out, bool bf16, int B, int T, int Tp, int H, hipStream_t st, bool out_alt) {
  return launch_self_attention_t<128>(Q, K, Vt, key_mask, out, bf16, B, T, Tp, H, st, out_alt);
}
hipError_t launch_self_attention_hd(const void* Q, const void* K, const void* Vt, const unsigned char* key_mask,
                                    void* out, bool bf16, int B, int T, int Tp, int H, int head_dim, hipStream_t st, bool out_alt) {
  if (head_dim == 64) return launch_self_attention_t<64>(Q, K, Vt, key_mask, out, bf16, B, T, Tp, H, st, out_alt);
  if (head_dim == 128) return launch_self_attention_t<128>(Q, K, Vt, key_mask, out, bf16, B, T, Tp, H, st, out_alt);
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------
// cross-attention onto the text memory (reference transformer.py:382-388 via :128-161: qk-norm on, no RoPE,
// no gate).  One wave per (row, head); lane holds elements (2*lane, 2*lane+1) of the 128-wide head.
// ---------------------------------------------------------------------------------------------------
template <typename TA, int HD>
__global__ __launch_bounds__(256) void cross_attn_kernel(const TA* __restrict__ q, const float* __restrict__ qw,
                                                         const TA* __restrict__ kv, long kv_ld,
                                                         const unsigned char* __restrict__ mask, TA* __restrict__ out,
                                                         long M, int T, int Lt, int H, float eps) {
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= M * H) return;
  const int lane = threadIdx.x & 63;
  const bool on = lane < HD / 2;   // HD = 64: half the wave sits out (zeros in the reductions)
  const long m = item / H;
  const int h = (int)(item % H);
  const long b = m / T;
  const int D = H * HD;
  float q0 = 0.f, q1 = 0.f;
  if (on) load2<TA>(q + m * D + h * HD + 2 * lane, q0, q1);
  const float inv = rsqrtf(wave_sum(q0 * q0 + q1 * q1) / (float)HD + eps);
  q0 *= inv * (on ? qw[2 * lane] : 0.f);
  q1 *= inv * (on ? qw[2 * lane + 1] : 0.f);
  const float scale = HD == 128 ? 0.08838834764831845f : 0.125f;   // head_dim^-0.5
  float mx = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f;
  for (int j = 0; j < Lt; ++j) {
    if (!mask[b * Lt + j]) continue;  // wave-uniform
    const TA* krow = kv + (b * Lt + j) * kv_ld + h * HD + 2 * lane;
    float k0 = 0.f, k1 = 0.f, v0 = 0.f, v1 = 0.f;
    if (on) {
      load2<TA>(krow, k0, k1);
      load2<TA>(krow + D, v0, v1);
    }
    const float s = wave_sum(q0 * k0 + q1 * k1) * scale;
    const float m_new = fmaxf(mx, s);
    const float a = expf(mx - m_new), p = expf(s - m_new);
    l = l * a + p;
    o0 = o0 * a + p * v0;
    o1 = o1 * a + p * v1;
    mx = m_new;
  }
  const float il = 1.f / l;
  if (on) store2<TA>(out + m * D + h * HD + 2 * lane, o0 * il, o1 * il);
}

// ---------------------------------------------------------------------------------------------------
// bf16 MFMA cross-attention for short memories (Lt <= 16 text tokens).  grid (ceil(T/64), H, B), 4 waves x 16 rows.
// Both contractions run transposed so that every lane keeps ONE query row:
//   S^T[token][row] = K[token][:] . Qn[row][:]   (A = K rows, B = normalised q)  -> lane: 4 tokens of its row
//   O^T[d][row]     = V^T[d][:]   . P[row][:]    (A = V^T from LDS, B = P)       -> lane: 4 consecutive d of its row
// so the q-norm statistics and the softmax are lane-local plus two 16/32-lane shuffles, and the output is written
// as 8-byte stores.  Traffic = q in + out (2*M*D*2 B); K/V of the (batch, head) are 4 KB from L2.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_attn_mfma_kernel(const bf16_t* __restrict__ q, const float* __restrict__ qw,
                                                              const bf16_t* __restrict__ kv, long kv_ld,
                                                              const unsigned char* __restrict__ mask,
                                                              bf16_t* __restrict__ out, int T, int Lt, int H, float eps) {
  __shared__ __attribute__((aligned(16))) unsigned short Vt[128 * 16];  // [d][token], tokens >= Lt are zero
  const int h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int D = H * 128;
  for (int idx = tid; idx < 128 * 16; idx += 256) {
    const int d = idx >> 4, j = idx & 15;
    Vt[idx] = j < Lt ? kv[((long)b * Lt + j) * kv_ld + D + h * 128 + d].v : (unsigned short)0;
  }
  // K fragments: rows = tokens
  bf16x8_t kf[4];
  // mask bytes are loaded unconditionally (clamped index) and up front: under `tok < Lt &&` each was a branch around a
  // dependent global load that had to land before the next instruction - one L2 round trip each, in series
  const unsigned char mask_r = mask[(long)b * Lt + (r < Lt ? r : 0)];
  unsigned char mask_e[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) mask_e[e] = mask[(long)b * Lt + (g * 4 + e < Lt ? g * 4 + e : 0)];
  const bool tok_ok = r < Lt && mask_r != 0;
  {
    const bf16_t* krow = kv + ((long)b * Lt + (r < Lt ? r : 0)) * kv_ld + h * 128 + g * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 v = *(const uint4*)(krow + ks * 32);
      if (!tok_ok) v = make_uint4(0u, 0u, 0u, 0u);
      kf[ks] = *(const bf16x8_t*)&v;
    }
  }
  // q rows: lane holds d = ks*32 + g*8 .. +8 of row r
  const int t = blockIdx.x * 64 + wave * 16 + r;
  const int tc = t < T ? t : T - 1;
  const long m = (long)b * T + tc;
  float qv[4][8];
  float ss = 0.f;
  {
    const bf16_t* qrow = q + m * D + h * 128 + g * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 v = *(const uint4*)(qrow + ks * 32);
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qv[ks][2 * e] = h16_lo(w4[e]);
        qv[ks][2 * e + 1] = h16_hi(w4[e]);
        ss += qv[ks][2 * e] * qv[ks][2 * e] + qv[ks][2 * e + 1] * qv[ks][2 * e + 1];
      }
    }
  }
  // q-norm weights of the lane's 32 channels: requested together with the operands and pinned there (an empty asm that
  // "uses" them) - left alone the compiler sinks each of these loads to its k-step, behind a vmcnt(0) of its own
  f32x4_t wq[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    wq[ks][0] = *(const f32x4_t*)(qw + ks * 32 + g * 8);
    wq[ks][1] = *(const f32x4_t*)(qw + ks * 32 + g * 8 + 4);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    asm volatile("" : "+v"(wq[ks][0]));
    asm volatile("" : "+v"(wq[ks][1]));
  }
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  const float inv = rsqrtf(ss / 128.f + eps);
  f32x4_t sT = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const f32x4_t w0 = wq[ks][0], w1 = wq[ks][1];
    const float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
    unsigned pk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      pk[e] = (unsigned)f2bf(qv[ks][2 * e] * inv * wv[2 * e]) | ((unsigned)f2bf(qv[ks][2 * e + 1] * inv * wv[2 * e + 1]) << 16);
    const uint4 pv = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    sT = SA_MFMA_16x16x32(kf[ks], *(const bf16x8_t*)&pv, sT);
  }
  // softmax over the tokens of row r: lane holds tokens g*4 + e
  const float scale = 0.08838834764831845f;
  float p[4], mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int tok = g * 4 + e;
    const bool ok = tok < Lt && mask_e[e] != 0;
    p[e] = ok ? sT[e] * scale : -INFINITY;
    mx = fmaxf(mx, p[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m_safe = mx == -INFINITY ? 0.f : mx;
  float l = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    p[e] = __expf(p[e] - m_safe);
    l += p[e];
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  // P as the B operand: lane (row r, group g) needs tokens 8g .. 8g+7 = groups 2g and 2g+1 of the same row
  unsigned ppk[4];
  {
    const int src_lo = ((2 * g) & 3) * 16 + r, src_hi = ((2 * g + 1) & 3) * 16 + r;
    float lo[4], hi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo[e] = __shfl(p[e], src_lo, 64);
      hi[e] = __shfl(p[e], src_hi, 64);
    }
    const bool live = g < 2;  // tokens 16..31 do not exist
    ppk[0] = live ? ((unsigned)f2bf(lo[0]) | ((unsigned)f2bf(lo[1]) << 16)) : 0u;
    ppk[1] = live ? ((unsigned)f2bf(lo[2]) | ((unsigned)f2bf(lo[3]) << 16)) : 0u;
    ppk[2] = live ? ((unsigned)f2bf(hi[0]) | ((unsigned)f2bf(hi[1]) << 16)) : 0u;
    ppk[3] = live ? ((unsigned)f2bf(hi[2]) | ((unsigned)f2bf(hi[3]) << 16)) : 0u;
  }
  const uint4 pu = make_uint4(ppk[0], ppk[1], ppk[2], ppk[3]);
  const bf16x8_t pB = *(const bf16x8_t*)&pu;
  __syncthreads();  // V^T staged
  const float il = 1.f / l;
  bf16_t* orow = out + m * D + h * 128 + g * 4;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    // lane groups 2, 3 stand for tokens 16..31, which do not exist: their operand must be an exact zero (reading
    // past the 16-token row would pick up the next row - or, for d = 127, LDS left behind by an earlier kernel,
    // and 0 * NaN from there poisons the whole tile: seen as box-dependent NaNs before this guard existed)
    bf16x8_t vf = *(const bf16x8_t*)(Vt + (n * 16 + r) * 16 + (g & 1) * 8);
    if (g >= 2) {
      const uint4 z = make_uint4(0u, 0u, 0u, 0u);
      vf = *(const bf16x8_t*)&z;
    }
    f32x4_t o = f32x4_t{0.f, 0.f, 0.f, 0.f};
    o = SA_MFMA_16x16x32(vf, pB, o);
    if (t < T) store4<bf16_t>(orow + n * 16, o[0] * il, o[1] * il, o[2] * il, o[3] * il);
  }
}

// ---------------------------------------------------------------------------------------------------
// Folded cross-attention output projection (bf16 mode, Lt <= 16).
//   reference: h += Wo . concat_h( P_h V_h )      (transformer.py:153-161, :382-388)
//   folded:    h += P . U,   U[(h, j), :] = Wo[:, h*128:(h+1)*128] . V[j, h, :]
// The contraction over D = H*128 channels of the c_wo GEMM becomes a contraction over H*Lt (176 at H = 22, Lt = 8)
// probabilities: 16x fewer MFMA flops for that GEMM, and the attention kernel writes [M, H*Lt] instead of [M, D].
//
// cross_attn_probs_kernel: same scores / softmax as cross_attn_mfma_kernel, output P[m][h*LtP + token] (normalised,
// bf16), LtP = 8 or 16.  grid (ceil(T/64), H, B).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_attn_probs_kernel(const bf16_t* __restrict__ q, const float* __restrict__ qw,
                                                               const bf16_t* __restrict__ kv, long kv_ld,
                                                               const unsigned char* __restrict__ mask,
                                                               bf16_t* __restrict__ P, int ldp, int T, int Lt, int LtP,
                                                               int H, float eps) {
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int D = H * 128;
  bf16x8_t kf[4];
  // mask bytes are loaded unconditionally (clamped index) and up front: under `tok < Lt &&` each was a branch around a
  // dependent global load that had to land before the next instruction - one L2 round trip each, in series
  const unsigned char mask_r = mask[(long)b * Lt + (r < Lt ? r : 0)];
  unsigned char mask_e[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) mask_e[e] = mask[(long)b * Lt + (g * 4 + e < Lt ? g * 4 + e : 0)];
  const bool tok_ok = r < Lt && mask_r != 0;
  {
    const bf16_t* krow = kv + ((long)b * Lt + (r < Lt ? r : 0)) * kv_ld + h * 128 + g * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 v = *(const uint4*)(krow + ks * 32);
      if (!tok_ok) v = make_uint4(0u, 0u, 0u, 0u);
      kf[ks] = *(const bf16x8_t*)&v;
    }
  }
  const int t = blockIdx.x * 64 + wave * 16 + r;
  const int tc = t < T ? t : T - 1;
  const long m = (long)b * T + tc;
  float qv[4][8];
  float ss = 0.f;
  {
    const bf16_t* qrow = q + m * D + h * 128 + g * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 v = *(const uint4*)(qrow + ks * 32);
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qv[ks][2 * e] = h16_lo(w4[e]);
        qv[ks][2 * e + 1] = h16_hi(w4[e]);
        ss += qv[ks][2 * e] * qv[ks][2 * e] + qv[ks][2 * e + 1] * qv[ks][2 * e + 1];
      }
    }
  }
  // q-norm weights of the lane's 32 channels: requested together with the operands and pinned there (an empty asm that
  // "uses" them) - left alone the compiler sinks each of these loads to its k-step, behind a vmcnt(0) of its own
  f32x4_t wq[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    wq[ks][0] = *(const f32x4_t*)(qw + ks * 32 + g * 8);
    wq[ks][1] = *(const f32x4_t*)(qw + ks * 32 + g * 8 + 4);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    asm volatile("" : "+v"(wq[ks][0]));
    asm volatile("" : "+v"(wq[ks][1]));
  }
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  const float inv = rsqrtf(ss / 128.f + eps);
  f32x4_t sT = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const f32x4_t w0 = wq[ks][0], w1 = wq[ks][1];
    const float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
    unsigned pk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      pk[e] = (unsigned)f2bf(qv[ks][2 * e] * inv * wv[2 * e]) | ((unsigned)f2bf(qv[ks][2 * e + 1] * inv * wv[2 * e + 1]) << 16);
    const uint4 pv = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    sT = SA_MFMA_16x16x32(kf[ks], *(const bf16x8_t*)&pv, sT);
  }
  const float scale = 0.08838834764831845f;
  float p[4], mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int tok = g * 4 + e;
    const bool ok = tok < Lt && mask_e[e] != 0;
    p[e] = ok ? sT[e] * scale : -INFINITY;
    mx = fmaxf(mx, p[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m_safe = mx == -INFINITY ? 0.f : mx;
  float l = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    p[e] = __expf(p[e] - m_safe);
    l += p[e];
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float il = 1.f / l;
  if (t < T && g * 4 < LtP) store4<bf16_t>(P + m * ldp + h * LtP + g * 4, p[0] * il, p[1] * il, p[2] * il, p[3] * il);
}

// cross_attn_fold_kernel: U^T[b][n][h*LtP + j] = sum_d Wo[n][h*128 + d] * V[b][j][h*128 + d]   (bf16, zero for
// j >= Lt and for the K padding), the per-batch weight operand ([N = D][K = KP], K contiguous) of the folded GEMM.
//
// Round 3 form.  What bounded the round-2 kernel (one wave = one head x 16 output channels, 8-byte stores straight from
// the accumulator layout: 49 us per launch at the benchmark shape, 0.9 TB/s) was not latency, partial lines or request
// rate taken one at a time but the SUM of its memory traffic: every (batch, head) slice of V - 2 KB -
// was fetched by each of the D/16 = 176 waves that needed it (PMC: 7.9 M of the launch's 9.9 M L2 requests), and every
// 16-byte piece of the output left as its own partial line (73.6 MB written for 34.6 MB of output).  Here
//   * a wave keeps the Wo fragments of 64 output channels of its head in registers (4 n-fragments x 4 k-steps), so a V
//     slice is fetched by D/64 = 44 waves: a quarter of the requests;
//   * the 8 waves of a workgroup take 8 CONSECUTIVE heads of the same 64 channels and 4 batch items per trip, stage
//     [item][channel][8 heads x LtP tokens] in LDS and the workgroup writes every (item, channel) row as ONE contiguous,
//     aligned run of 8 LtP elements = 128 B (LtP = 8) or 256 B (LtP = 16): whole cache lines only.
// grid (D/64, ceil(KP / (8 LtP)), batch splits).  Heads >= H of the last group write the zeros the K padding of U holds.
// MFMA orientation: rows = tokens (A = V), columns = n (B = Wo^T) -> a lane owns 4 consecutive tokens of one n; the sum
// over d runs through the same four MFMAs in the same order as before: bitwise the round-2 result.
// Round 5: ALL LAYERS OF AN EVALUATION IN ONE LAUNCH.  U depends on the layer's Wo and on the text memory only - not on the
// residual stream - so the 22 folds of an evaluation need not sit between the layers' GEMMs as 22 launches of 132 workgroups (12 us
// each at 4 clips, mostly latency): gridDim.z = layers x batch splits, layer l reads Wo_l (pointer table), the K | V columns
// [l * 2 D, (l + 1) * 2 D) of the stacked cross-attention projections and writes its own [B][D][KP] slice.  Same arithmetic per
// (layer, item): bitwise the per-layer launches.
struct FoldLayers {
  const bf16_t* wo[kMaxFoldLayers];
};
__global__ __launch_bounds__(512) void cross_attn_fold_kernel(const FoldLayers layers, const int zsplits, const bf16_t* __restrict__ kv_all,
                                                              long kv_ld, bf16_t* __restrict__ UT_all, int KP, int B, int Lt,
                                                              int LtP, int H) {
  __shared__ __attribute__((aligned(16))) unsigned short stage[4 * 64 * 8 * 16];  // [item][n][head][token <= 16]: 64 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int D = H * 128;
  const int layer = (int)blockIdx.z / zsplits, zsplit = (int)blockIdx.z % zsplits;
  const bf16_t* __restrict__ wo = layers.wo[layer];
  const bf16_t* __restrict__ kv = kv_all + (long)layer * 2 * D;
  bf16_t* __restrict__ UT = UT_all + (long)layer * B * D * KP;
  const int h = blockIdx.y * 8 + wave;
  const bool head_ok = h < H;
  const int hc = head_ok ? h : H - 1;
  const int n0 = blockIdx.x * 64;
  bf16x8_t wf[4][4];  // [n-fragment][k-step]
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      wf[s4][ks] = *(const bf16x8_t*)(wo + (long)(n0 + s4 * 16 + r) * D + hc * 128 + ks * 32 + g * 8);
  const int run = 8 * LtP;                        // elements of one (item, channel) run: this workgroup's heads
  const int k0 = blockIdx.y * run;                // first column of the run inside a U^T row
  const int segs = run / 8;                       // 16-byte segments per run
  const int bz = (((B + zsplits - 1) / zsplits) + 3) & ~3;
  const int b_end = (zsplit + 1) * bz < B ? (zsplit + 1) * bz : B;
  for (int b0 = zsplit * bz; b0 < b_end; b0 += 4) {
    uint4 v[4][4];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const int b = b0 + bb < B ? b0 + bb : B - 1;
      const bf16_t* vrow = kv + ((long)b * Lt + (r < Lt ? r : 0)) * kv_ld + D + hc * 128 + g * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        v[bb][ks] = *(const uint4*)(vrow + ks * 32);
        // rows of tokens that do not exist / heads beyond H must be exact zeros (never "multiplied away": 0 x NaN)
        if (r >= Lt || !head_ok) v[bb][ks] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        f32x4_t u = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) u = SA_MFMA_16x16x32(*(const bf16x8_t*)&v[bb][ks], wf[s4][ks], u);
        if (g * 4 < LtP) {  // lane: tokens g*4 .. g*4+3 of channel s4*16 + r
          ushort4 o;
          o.x = f2bf(u[0]); o.y = f2bf(u[1]); o.z = f2bf(u[2]); o.w = f2bf(u[3]);
          *(ushort4*)(stage + ((bb * 64 + s4 * 16 + r) * 8 + wave) * LtP + g * 4) = o;
        }
      }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * 64 * segs; idx += 512) {
      const int seg = idx % segs, n = (idx / segs) & 63, bb = idx / (segs * 64);
      if (b0 + bb < B && k0 + seg * 8 < KP)
        *(uint4*)(UT + ((long)(b0 + bb) * D + n0 + n) * KP + k0 + seg * 8) = *(const uint4*)(stage + (bb * 64 + n) * run + seg * 8);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// Folded cross-attention of an x3 context (SAMAUDIO_OPT_X3_CLASSES class CWO, Lt <= 16, head_dim 128): the same fold as above -
// h += P . U, U = Wo V per (layer, batch item) - with fp32 inputs and compensated operands on both sides of the K = H*LtP GEMM:
//   cross_attn_probs3_kernel: fp32 q (raw, q-norm applied here) and fp32 k -> softmax probabilities in fp32 (one wave per (row, head),
//     as cross_attn_kernel<float>), written as the GEMM's split activation row [P_lo | P_hi | P_hi] (3 KP 16-bit elements);
//   cross_attn_fold3_kernel: cross_attn_fold_kernel with fp32 Wo / V split in registers, U = V_l Wo_h + V_h Wo_l + V_h Wo_h in fp32,
//     written as the per-batch weight operand [U_hi | U_lo | U_hi] (rows of 3 KP).
// The D-wide c_wo GEMM over K' = 3 D of the unfolded x3 path becomes one over K' = 3 KP = 576: 7.6 % of the DiT's flops gone.
// ---------------------------------------------------------------------------------------------------
// LTP = token slots per head (8 | 16).  One wave = 16 rows of one clip x one head, the scores S = (q w) K^T on the 16-bit MFMA over
// split operands (q_l K_h + q_h K_l + q_h K_h, four k-steps of 32: 12 MFMAs), as every other contraction of the mode.  The form it
// replaces (rounds 6, first half: one wave per (row, head), fp32 products, nine butterfly reductions of six ds_bpermute steps each) was
// bound by the latency of its 54 cross-lane steps: 68 us per launch at 4 000 rows for 45 MB of q (profiles/r6_final3/).  Lane
// (lr, lg) holds the operands' k = 32 ks + 8 lg .. + 7 of row / token lr and, after the MFMAs, the scores of row lr and tokens
// 4 lg .. 4 lg + 3; the row statistic of the q-norm and the softmax need two exchanges across lg each.
template <int LTP>
__global__ __launch_bounds__(256) void cross_attn_probs3_kernel(const float* __restrict__ q, const float* __restrict__ qw,
                                                                const float* __restrict__ kv, long kv_ld,
                                                                const unsigned char* __restrict__ mask, bf16_t* __restrict__ P3, int KP,
                                                                long M, int T, int Lt, int H, float eps) {
#pragma clang fp contract(off)
  const int tblks = (T + 15) >> 4;                                  // row blocks per clip (a block never spans two clips: K differs)
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // (clip, row block, head)
  const long B = M / T;
  if (item >= B * tblks * H) return;
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
  const int h = (int)(item % H);
  const long bt = item / H;
  const long b = bt / tblks;
  const int t = (int)(bt % tblks) * 16 + lr;
  const bool row_ok = t < T;
  const long m = b * T + (row_ok ? t : T - 1);
  const int D = H * 128;
  const bool tok_ok = lr < Lt && mask[b * Lt + (lr < Lt ? lr : 0)] != 0;   // this lane's K row (token lr) takes part
  const float* qrow = q + m * D + h * 128 + lg * 8;
  const float* krow = kv + (b * Lt + (lr < Lt ? lr : 0)) * kv_ld + h * 128 + lg * 8;
  const float* wrow = qw + lg * 8;
  auto cl = [](float x) { return __builtin_amdgcn_fmed3f(x, -kH16Max, kH16Max); };
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float ss = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const float4 qa = *(const float4*)(qrow + ks * 32), qb = *(const float4*)(qrow + ks * 32 + 4);
    const float4 wa = *(const float4*)(wrow + ks * 32), wb = *(const float4*)(wrow + ks * 32 + 4);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 ka = tok_ok ? *(const float4*)(krow + ks * 32) : z, kb = tok_ok ? *(const float4*)(krow + ks * 32 + 4) : z;
    ss += qa.x * qa.x + qa.y * qa.y + qa.z * qa.z + qa.w * qa.w + qb.x * qb.x + qb.y * qb.y + qb.z * qb.z + qb.w * qb.w;
    uint4 qh, ql, kh, kl;
    split8(make_float4(cl(qa.x * wa.x), cl(qa.y * wa.y), cl(qa.z * wa.z), cl(qa.w * wa.w)),
           make_float4(cl(qb.x * wb.x), cl(qb.y * wb.y), cl(qb.z * wb.z), cl(qb.w * wb.w)), qh, ql);
    split8(make_float4(cl(ka.x), cl(ka.y), cl(ka.z), cl(ka.w)), make_float4(cl(kb.x), cl(kb.y), cl(kb.z), cl(kb.w)), kh, kl);
    const bf16x8_t Qh = __builtin_bit_cast(bf16x8_t, qh), Ql = __builtin_bit_cast(bf16x8_t, ql);
    const bf16x8_t Kh = __builtin_bit_cast(bf16x8_t, kh), Kl = __builtin_bit_cast(bf16x8_t, kl);
    acc = SA_MFMA_16x16x32(Kh, Ql, acc);   // (tokens as the first operand: a lane then owns 4 consecutive tokens of ITS row)
    acc = SA_MFMA_16x16x32(Kl, Qh, acc);
    acc = SA_MFMA_16x16x32(Kh, Qh, acc);
  }
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  const float inv = rsqrtf(ss / 128.f + eps);
  float sc[4], mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * lg + e;
    const bool live = j < Lt && mask[b * Lt + (j < Lt ? j : 0)] != 0;
    sc[e] = live ? acc[e] * inv * 0.08838834764831845f : -INFINITY;
    mx = fmaxf(mx, sc[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float ex[4], l = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ex[e] = sc[e] != -INFINITY ? expf(sc[e] - mx) : 0.f;
    l += ex[e];
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (row_ok && 4 * lg < LTP) {
    float pr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pr[e] = 4 * lg + e < Lt ? ex[e] / l : 0.f;   // (every token masked: 0 / 0 = NaN, as the reference's softmax of an all -inf row)
    const unsigned h0 = pack_h16x2(cl(pr[0]), cl(pr[1])), h1 = pack_h16x2(cl(pr[2]), cl(pr[3]));
    const unsigned l0 = pack_h16x2(pr[0] - h16_lo(h0), pr[1] - h16_hi(h0)), l1 = pack_h16x2(pr[2] - h16_lo(h1), pr[3] - h16_hi(h1));
    unsigned short* row = (unsigned short*)P3 + m * (3L * KP) + h * LTP + 4 * lg;
    *(uint2*)row = make_uint2(l0, l1);
    *(uint2*)(row + KP) = make_uint2(h0, h1);
    *(uint2*)(row + 2 * KP) = make_uint2(h0, h1);
  }
}

struct FoldLayers3 {
  const float* wo[kMaxFoldLayers];
};
__global__ __launch_bounds__(512) void cross_attn_fold3_kernel(const FoldLayers3 layers, const int zsplits, const float* __restrict__ kv_all,
                                                               long kv_ld, bf16_t* __restrict__ UT_all, int KP, int B, int Lt,
                                                               int LtP, int H) {
  __shared__ __attribute__((aligned(16))) unsigned short stage[2][4 * 64 * 8 * 16];  // [hi | lo][item][n][head][token <= 16]: 128 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int D = H * 128;
  const int layer = (int)blockIdx.z / zsplits, zsplit = (int)blockIdx.z % zsplits;
  const float* __restrict__ wo = layers.wo[layer];
  const float* __restrict__ kv = kv_all + (long)layer * 2 * D;
  bf16_t* __restrict__ UT = UT_all + (long)layer * B * D * 3 * KP;
  const int h = blockIdx.y * 8 + wave;
  const bool head_ok = h < H;
  const int hc = head_ok ? h : H - 1;
  const int n0 = blockIdx.x * 64;
  bf16x8_t wfh[4][4], wfl[4][4];  // [n-fragment][k-step], hi / lo halves of the fp32 Wo fragments
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float* src = wo + (long)(n0 + s4 * 16 + r) * D + hc * 128 + ks * 32 + g * 8;
      uint4 hi, lo;
      split8(*(const float4*)src, *(const float4*)(src + 4), hi, lo);
      wfh[s4][ks] = __builtin_bit_cast(bf16x8_t, hi);
      wfl[s4][ks] = __builtin_bit_cast(bf16x8_t, lo);
    }
  const int run = 8 * LtP;                        // elements of one (item, channel) run: this workgroup's heads
  const int k0 = blockIdx.y * run;                // first column of the run inside a third of a U^T row
  const int segs = run / 8;                       // 16-byte segments per run
  const int bz = (((B + zsplits - 1) / zsplits) + 3) & ~3;
  const int b_end = (zsplit + 1) * bz < B ? (zsplit + 1) * bz : B;
  for (int b0 = zsplit * bz; b0 < b_end; b0 += 4) {
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const int b = b0 + bb < B ? b0 + bb : B - 1;
      const float* vrow = kv + ((long)b * Lt + (r < Lt ? r : 0)) * kv_ld + D + hc * 128 + g * 8;
      bf16x8_t vh[4], vl[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        uint4 hi, lo;
        split8(*(const float4*)(vrow + ks * 32), *(const float4*)(vrow + ks * 32 + 4), hi, lo);
        // rows of tokens that do not exist / heads beyond H must be exact zeros (never "multiplied away": 0 x NaN)
        if (r >= Lt || !head_ok) hi = lo = make_uint4(0u, 0u, 0u, 0u);
        vh[ks] = __builtin_bit_cast(bf16x8_t, hi);
        vl[ks] = __builtin_bit_cast(bf16x8_t, lo);
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        f32x4_t u = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          u = SA_MFMA_16x16x32(vl[ks], wfh[s4][ks], u);
          u = SA_MFMA_16x16x32(vh[ks], wfl[s4][ks], u);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) u = SA_MFMA_16x16x32(vh[ks], wfh[s4][ks], u);
        if (g * 4 < LtP) {  // lane: tokens g*4 .. g*4+3 of channel s4*16 + r
          const unsigned h01 = pack_h16x2(u[0], u[1]), h23 = pack_h16x2(u[2], u[3]);
          const unsigned l01 = pack_h16x2(u[0] - h16_lo(h01), u[1] - h16_hi(h01)), l23 = pack_h16x2(u[2] - h16_lo(h23), u[3] - h16_hi(h23));
          const int at = ((bb * 64 + s4 * 16 + r) * 8 + wave) * LtP + g * 4;
          *(uint2*)(stage[0] + at) = make_uint2(h01, h23);
          *(uint2*)(stage[1] + at) = make_uint2(l01, l23);
        }
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * 64 * segs; idx += 512) {
      const int seg = idx % segs, n = (idx / segs) & 63, bb = idx / (segs * 64);
      if (b0 + bb < B && k0 + seg * 8 < KP) {
        bf16_t* dst = UT + ((long)(b0 + bb) * D + n0 + n) * (3L * KP) + k0 + seg * 8;   // [U_hi | U_lo | U_hi]
        const uint4 hi = *(const uint4*)(stage[0] + (bb * 64 + n) * run + seg * 8), lo = *(const uint4*)(stage[1] + (bb * 64 + n) * run + seg * 8);
        *(uint4*)dst = hi;
        *(uint4*)(dst + KP) = lo;
        *(uint4*)(dst + 2 * KP) = hi;
      }
    }
    __syncthreads();
  }
}

hipError_t launch_cross_attn_probs3(const float* q, const float* qw, const float* kv, long kv_ld, const unsigned char* mask, void* P3,
                                    int KP, int B, int T, int Lt, int LtP, int H, float eps, hipStream_t st) {
  if (Lt > 16 || (LtP != 8 && LtP != 16) || Lt > LtP || H * LtP > KP) return hipErrorInvalidValue;
  const long items = (long)B * ((T + 15) / 16) * H;   // one wave per (clip, 16-row block, head)
  if (LtP == 8)
    hipLaunchKernelGGL(cross_attn_probs3_kernel<8>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, q, qw, kv, kv_ld, mask, (bf16_t*)P3, KP,
                       (long)B * T, T, Lt, H, eps);
  else
    hipLaunchKernelGGL(cross_attn_probs3_kernel<16>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, q, qw, kv, kv_ld, mask, (bf16_t*)P3, KP,
                       (long)B * T, T, Lt, H, eps);
  return hipGetLastError();
}
hipError_t launch_cross_attn_fold3_layers(const float* const* wo, int n_layers, const float* kv_all, long kv_ld, void* UT3_all, int KP,
                                          int B, int Lt, int LtP, int H, hipStream_t st) {
  if (n_layers <= 0 || n_layers > kMaxFoldLayers || (LtP != 8 && LtP != 16) || Lt > LtP || H * LtP > KP) return hipErrorInvalidValue;
  FoldLayers3 layers;
  for (int l = 0; l < kMaxFoldLayers; ++l) layers.wo[l] = wo[l < n_layers ? l : 0];
  const int D = H * 128;
  const int groups = (KP + 8 * LtP - 1) / (8 * LtP);
  const int target = B <= 16 ? 1 : 384;   // (batch splits as launch_cross_attn_fold_layers)
  int zs = (target + (D / 64) * groups - 1) / ((D / 64) * groups);
  const int zmax = (B + 3) / 4;
  zs = zs < 1 ? 1 : zs > zmax ? zmax : zs;
  hipLaunchKernelGGL(cross_attn_fold3_kernel, dim3(D / 64, groups, zs * n_layers), dim3(512), 0, st, layers, zs, kv_all, kv_ld,
                     (bf16_t*)UT3_all, KP, B, Lt, LtP, H);
  return hipGetLastError();
}

hipError_t launch_cross_attn_probs(const void* q, const float* qw, const void* kv, long kv_ld, const unsigned char* mask,
                                   void* P, int ldp, int B, int T, int Lt, int LtP, int H, float eps, hipStream_t st) {
  hipLaunchKernelGGL(cross_attn_probs_kernel, dim3((T + 63) / 64, H, B), dim3(256), 0, st, (const bf16_t*)q, qw,
                     (const bf16_t*)kv, kv_ld, mask, (bf16_t*)P, ldp, T, Lt, LtP, H, eps);
  return hipGetLastError();
}

// wo[l]: the output projection of layer l; kv_all [B * Lt, kv_ld] with layer l's (k | v) in columns [l * 2 D, (l + 1) * 2 D);
// UT_all [n_layers][B][D][KP]
hipError_t launch_cross_attn_fold_layers(const void* const* wo, int n_layers, const void* kv_all, long kv_ld, void* UT_all, int KP,
                                         int B, int Lt, int LtP, int H, hipStream_t st) {
  if ((H * 128) % 64 || (LtP != 8 && LtP != 16) || KP % 8 || KP < H * LtP || n_layers < 1 || n_layers > kMaxFoldLayers)
    return hipErrorInvalidValue;
  const int groups = (KP + 8 * LtP - 1) / (8 * LtP);   // 8-head groups covering the padded row
  // batch splits, at least 4 items each (a trip handles 4).  Every split re-reads Wo, and the kernel is bound by the sum of its
  // L2 traffic, not by how many CUs hold a workgroup: up to 16 items no split at all (132 workgroups walking 4 trips: 23.3 vs
  // 27.1 us for 528 one-trip workgroups), beyond that ~384 workgroups (32 items: 35.6 vs 41.3 us; profiles/r3_call21/, where
  // requesting the V fragments one trip ahead also measured no better).
  const int target = B <= 16 ? 1 : 384;
  int zs = (target + (H * 128 / 64) * groups - 1) / ((H * 128 / 64) * groups);
  const int zmax = (B + 3) / 4;
  zs = zs < 1 ? 1 : zs > zmax ? zmax : zs;
  FoldLayers layers;
  for (int l = 0; l < kMaxFoldLayers; ++l) layers.wo[l] = (const bf16_t*)wo[l < n_layers ? l : 0];
  hipLaunchKernelGGL(cross_attn_fold_kernel, dim3(H * 128 / 64, groups, zs * n_layers), dim3(512), 0, st, layers, zs,
                     (const bf16_t*)kv_all, kv_ld, (bf16_t*)UT_all, KP, B, Lt, LtP, H);
  return hipGetLastError();
}
hipError_t launch_cross_attn_fold(const void* wo, const void* kv, long kv_ld, void* UT, int KP, int B, int Lt, int LtP,
                                  int H, hipStream_t st) {
  return launch_cross_attn_fold_layers(&wo, 1, kv, kv_ld, UT, KP, B, Lt, LtP, H, st);
}

hipError_t launch_cross_attention(const void* q, const float* qw, const void* kv, long kv_ld,
                                  const unsigned char* mask, void* out, bool bf16, int B, int T, int Lt, int H,
                                  float eps, hipStream_t st, int head_dim) {
  const long M = (long)B * T;
  if (head_dim != 64 && head_dim != 128) return hipErrorInvalidValue;
  if (bf16 && Lt <= 16 && head_dim == 128) {
    hipLaunchKernelGGL(cross_attn_mfma_kernel, dim3((T + 63) / 64, H, B), dim3(256), 0, st, (const bf16_t*)q, qw,
                       (const bf16_t*)kv, kv_ld, mask, (bf16_t*)out, T, Lt, H, eps);
    return hipGetLastError();
  }
  dim3 grid((unsigned)((M * H + 3) / 4)), block(256);
  if (bf16 && head_dim == 128)
    hipLaunchKernelGGL((cross_attn_kernel<bf16_t, 128>), grid, block, 0, st, (const bf16_t*)q, qw, (const bf16_t*)kv, kv_ld,
                       mask, (bf16_t*)out, M, T, Lt, H, eps);
  else if (bf16)
    hipLaunchKernelGGL((cross_attn_kernel<bf16_t, 64>), grid, block, 0, st, (const bf16_t*)q, qw, (const bf16_t*)kv, kv_ld,
                       mask, (bf16_t*)out, M, T, Lt, H, eps);
  else if (head_dim == 128)
    hipLaunchKernelGGL((cross_attn_kernel<float, 128>), grid, block, 0, st, (const float*)q, qw, (const float*)kv, kv_ld, mask,
                       (float*)out, M, T, Lt, H, eps);
  else
    hipLaunchKernelGGL((cross_attn_kernel<float, 64>), grid, block, 0, st, (const float*)q, qw, (const float*)kv, kv_ld, mask,
                       (float*)out, M, T, Lt, H, eps);
  return hipGetLastError();
}

}  // namespace sa
