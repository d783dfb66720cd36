// Kernels of the T5 prompt encoder that the GEMM / RMSNorm kernels do not already cover (SURVEY.md section 8 rows
// a3 / f4; reference sam_audio/model/text_encoder.py:19-37 -> transformers' T5EncoderModel, restated in
// oracle/t5_oracle.py).  The encoder runs once per separate() call on B x Lt <= a few hundred token rows, so these
// are small latency-bound kernels: one wave per row, fp32 arithmetic, no MFMA.
//   t5_embed        token ids -> rows of the shared embedding table (fp32 residual stream)
//   t5_attention    softmax(q k^T + relative-position bias + key mask) v per (item, head, query row); T5 does NOT
//                   scale the scores by d_kv^-0.5 and adds the learned bias of the signed token distance
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
                                                          int H, int dkv, int max_len) {
  __shared__ float qs[128];
  __shared__ float ps[512];
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const long inner = (long)H * dkv, ld = 3 * inner;
  const TA* qrow = qkv + ((long)b * Lt + qi) * ld + (long)h * dkv;
  for (int d = lane; d < dkv; d += 64) qs[d] = Elem<TA>::load(qrow + d);
  __syncthreads();
  const float* brow = bias + (long)h * (2 * max_len - 1) + (max_len - 1) - qi;
  float s[8];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int k = lane + 64 * c;
    s[c] = -INFINITY;
    if (k < Lt && mask[(long)b * Lt + k]) {
      const TA* krow = qkv + ((long)b * Lt + k) * ld + inner + (long)h * dkv;
      float acc = 0.f;
      for (int d = 0; d < dkv; ++d) acc = fmaf(qs[d], Elem<TA>::load(krow + d), acc);
      s[c] = acc + brow[k];
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
                               int Lt, int H, int dkv, int max_len, hipStream_t st) {
  dim3 grid(Lt, H, B), block(64);
  if (bf16)
    hipLaunchKernelGGL(t5_attention_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)qkv, mask, bias, (bf16_t*)out, Lt, H,
                       dkv, max_len);
  else
    hipLaunchKernelGGL(t5_attention_kernel<float>, grid, block, 0, st, (const float*)qkv, mask, bias, (float*)out, Lt, H, dkv,
                       max_len);
  return hipGetLastError();
}

}  // namespace sa
