// Kernels of the T5 prompt encoder that the GEMM / RMSNorm kernels do not already cover (SURVEY.md section 8 rows
// a3 / f4; reference sam_audio/model/text_encoder.py:19-37 -> transformers' T5EncoderModel, restated in
// oracle/t5_oracle.py).  The encoder runs once per separate() call on B x Lt <= a few hundred token rows, so these
// are small latency-bound kernels: one wave per row, fp32 arithmetic, no MFMA.
//   t5_embed        token ids -> rows of the shared embedding table (fp32 residual stream)
//   t5_attention    softmax(scale * q k^T + relative-position bias + key mask [+ sliding window]) v per (item, head, query
//                   row); T5 does NOT scale the scores by d_kv^-0.5 and adds the learned bias of the signed token
//                   distance; ModernBERT (the Judge's / PE-A-Frame's text tower, mbert.hip) scales, has no bias and
//                   limits every other layer to keys within +-64 tokens
//   mbert_rope      rotate-half rotary embedding of the q and k segments of fused q|k|v rows, in place
//   geglu           act(x[:, :F]) * x[:, F:]  (ModernBERT's gated feed-forward)
#include "kernels.h"

namespace sa {

// out[m, :] = table[ids[m], :]; ids outside the table are clamped (the host rejects them before the launch).
__global__ __launch_bounds__(256) void t5_embed_kernel(const long long* __restrict__ ids, const float* __restrict__ table,
                                                       float* __restrict__ out, int D, int vocab) {
  const long m = blockIdx.x;
  long long id = ids[m];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float4* src = (const float4*)(table + (long)id * D);
  float4* dst = (float4*)(out + m * D);
  for (int i = threadIdx.x; i < (D >> 2); i += 256) dst[i] = src[i];
}

hipError_t launch_t5_embed(const long long* ids, const float* table, float* out, long M, int D, int vocab,
                           hipStream_t st) {
  hipLaunchKernelGGL(t5_embed_kernel, dim3((unsigned)M), dim3(256), 0, st, ids, table, out, D, vocab);
  return hipGetLastError();
}

// grid (Lt, H, B), one wave.  qkv rows [B*Lt, 3*inner] (q | k | v, inner = H*dkv), mask [B, Lt] (1 = token),
// bias [H, 2*max_len - 1]: bias[h][(k - q) + max_len - 1] = relative_attention_bias[bucket(k - q)][h] (the bucket rule
// is evaluated once on the host at load time, sam_audio_amd/t5_encoder.py).  A masked key gets probability 0
// (transformers adds finfo.min, whose exp underflows to exactly 0); a row whose keys are ALL masked attends uniformly,
// which is what finfo.min + score rounds to there.  Lt <= 512, dkv <= 128.
template <typename TA>
__global__ __launch_bounds__(64) void t5_attention_kernel(const TA* __restrict__ qkv, const unsigned char* __restrict__ mask,
                                                          const float* __restrict__ bias, TA* __restrict__ out, int Lt,
                                                          int H, int dkv, int max_len, float scale, int window) {
  __shared__ float qs[128];
  __shared__ float ps[512];
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const long inner = (long)H * dkv, ld = 3 * inner;
  const TA* qrow = qkv + ((long)b * Lt + qi) * ld + (long)h * dkv;
  for (int d = lane; d < dkv; d += 64) qs[d] = Elem<TA>::load(qrow + d);
  __syncthreads();
  const float* brow = bias ? bias + (long)h * (2 * max_len - 1) + (max_len - 1) - qi : nullptr;
  float s[8];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int k = lane + 64 * c;
    s[c] = -INFINITY;
    const int dist = k > qi ? k - qi : qi - k;
    if (k < Lt && mask[(long)b * Lt + k] && (window <= 0 || dist <= window)) {
      const TA* krow = qkv + ((long)b * Lt + k) * ld + inner + (long)h * dkv;
      float acc = 0.f;
      for (int d = 0; d < dkv; ++d) acc = fmaf(qs[d], Elem<TA>::load(krow + d), acc);
      s[c] = acc * scale + (brow ? brow[k] : 0.f);
    }
    mx = fmaxf(mx, s[c]);
  }
  mx = wave_max(mx);
  float l = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int k = lane + 64 * c;
    if (mx == -INFINITY) s[c] = k < Lt ? 1.f : 0.f;
    else s[c] = s[c] == -INFINITY ? 0.f : expf(s[c] - mx);
    l += s[c];
  }
  l = wave_sum(l);
  const float il = 1.f / l;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int k = lane + 64 * c;
    if (k < Lt) ps[k] = s[c] * il;
  }
  __syncthreads();
  TA* orow = out + ((long)b * Lt + qi) * inner + (long)h * dkv;
  for (int d = lane; d < dkv; d += 64) {
    const TA* vcol = qkv + (long)b * Lt * ld + 2 * inner + (long)h * dkv + d;
    float acc = 0.f;
    for (int k = 0; k < Lt; ++k) acc = fmaf(ps[k], Elem<TA>::load(vcol + (long)k * ld), acc);
    Elem<TA>::store(orow + d, acc);
  }
}

hipError_t launch_t5_attention(const void* qkv, const unsigned char* mask, const float* bias, void* out, bool bf16, int B,
                               int Lt, int H, int dkv, int max_len, float scale, int window, hipStream_t st) {
  dim3 grid(Lt, H, B), block(64);
  if (bf16)
    hipLaunchKernelGGL(t5_attention_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)qkv, mask, bias, (bf16_t*)out, Lt, H,
                       dkv, max_len, scale, window);
  else
    hipLaunchKernelGGL(t5_attention_kernel<float>, grid, block, 0, st, (const float*)qkv, mask, bias, (float*)out, Lt, H, dkv,
                       max_len, scale, window);
  return hipGetLastError();
}

// rotate-half rotary embedding (transformers apply_rotary_pos_emb): x' = x * cos + rotate_half(x) * sin with
// rotate_half(x) = (-x[hd/2:], x[:hd/2]), on the q and k segments of q|k|v rows [B*Lt, 3*H*hd], in place; cos / sin
// [Lt_max, hd] (both halves of a row hold the same angles); position = token index inside the item.  grid (B*Lt), 256 thr.
template <typename TA>
__global__ __launch_bounds__(256) void mbert_rope_kernel(TA* __restrict__ qkv, const float* __restrict__ cs,
                                                         const float* __restrict__ sn, int Lt, int H, int hd) {
  const long m = blockIdx.x;
  const int t = (int)(m % Lt), half = hd >> 1;
  TA* row = qkv + m * 3L * H * hd;
  for (int idx = threadIdx.x; idx < 2 * H * half; idx += 256) {
    const int seg = idx / half, d = idx - seg * half;   // seg: q heads 0 .. H-1, then k heads H .. 2H-1 (contiguous in the row)
    TA* x = row + (long)seg * hd;
    const float x1 = Elem<TA>::load(x + d), x2 = Elem<TA>::load(x + d + half);
    const float c1 = cs[(long)t * hd + d], s1 = sn[(long)t * hd + d];
    const float c2 = cs[(long)t * hd + d + half], s2 = sn[(long)t * hd + d + half];
    Elem<TA>::store(x + d, x1 * c1 + (-x2) * s1);
    Elem<TA>::store(x + d + half, x2 * c2 + x1 * s2);
  }
}

hipError_t launch_mbert_rope(void* qkv, const float* cs, const float* sn, bool bf16, long M, int Lt, int H, int hd,
                             hipStream_t st) {
  if (bf16) hipLaunchKernelGGL(mbert_rope_kernel<bf16_t>, dim3((unsigned)M), dim3(256), 0, st, (bf16_t*)qkv, cs, sn, Lt, H, hd);
  else hipLaunchKernelGGL(mbert_rope_kernel<float>, dim3((unsigned)M), dim3(256), 0, st, (float*)qkv, cs, sn, Lt, H, hd);
  return hipGetLastError();
}

// out[m, f] = gelu(x[m, f]) * x[m, F + f]   (ModernBertMLP: input, gate = Wi(h).chunk(2); act(input) * gate; erf GELU)
template <typename TA>
__global__ __launch_bounds__(256) void geglu_kernel(const TA* __restrict__ x, TA* __restrict__ out, long M, int F) {
  const long total = M * F;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / F;
    const int f = (int)(i - m * F);
    const float a = Elem<TA>::load(x + m * 2 * F + f), g = Elem<TA>::load(x + m * 2 * F + F + f);
    Elem<TA>::store(out + i, gelu_f(a) * g);
  }
}

hipError_t launch_geglu(const void* x, void* out, bool bf16, long M, int F, hipStream_t st) {
  const long total = M * F;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (bf16) hipLaunchKernelGGL(geglu_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, M, F);
  else hipLaunchKernelGGL(geglu_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)out, M, F);
  return hipGetLastError();
}

}  // namespace sa
