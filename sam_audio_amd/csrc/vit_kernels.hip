// Streaming kernels of the PE-Core vision tower (reference
// sam_audio/model/vision_encoder.py:80-89 -> `pe.CLIP.encode_image`, architecture restated in oracle/vit_oracle.py).
// All of them are HBM-bound re-layouts around the GEMMs and the flash attention kernel (attention.hip):
//   patchify        frames [n,3,S,S] f32 -> im2col rows [n*G*G, Kp] (the k = stride = patch conv becomes one GEMM)
//   rope2d_split    fused q|k|v rows -> Q, K [n,H,Sp,hd] with the 2-D rotary embedding, V^T [n,H,hd,Sp]
//   pool_attention  one learned query per head over all tokens (attention pooling head)
//   l2_normalize    rows of the projected features
#include "kernels.h"

namespace sa {

// ------------------------------------------------------------------------------------------------
// im2col of a k = stride = P convolution: row (f, gy, gx), column k = c*P*P + py*P + px (the flattening of
// conv1.weight [W, 3, P, P]); columns >= 3*P*P are zero (K padded to the GEMM's slab).  grid (G*G, n), 256 threads.
// ------------------------------------------------------------------------------------------------
template <typename TA>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ frames, TA* __restrict__ out, int S,
                                                       int P, int G, int Kp) {
  const int patch = blockIdx.x, f = blockIdx.y;
  const int gy = patch / G, gx = patch - gy * G;
  const int kk = 3 * P * P;
  const float* src = frames + (long)f * 3 * S * S + (long)(gy * P) * S + gx * P;
  TA* dst = out + ((long)f * G * G + patch) * Kp;
  for (int k = threadIdx.x; k < Kp; k += 256) {
    float v = 0.f;
    if (k < kk) {
      const int c = k / (P * P), r = k - c * P * P;
      const int py = r / P, px = r - py * P;
      v = src[(long)c * S * S + (long)py * S + px];
    }
    Elem<TA>::store(dst + k, v);
  }
}

hipError_t launch_patchify(const float* frames, void* out, bool bf16, int n, int S, int P, int Kp, hipStream_t st) {
  const int G = S / P;
  dim3 grid(G * G, n), block(256);
  if (bf16) hipLaunchKernelGGL(patchify_kernel<bf16_t>, grid, block, 0, st, frames, (bf16_t*)out, S, P, G, Kp);
  else hipLaunchKernelGGL(patchify_kernel<float>, grid, block, 0, st, frames, (float*)out, S, P, G, Kp);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// q|k|v rows [n*T, 3*H*HD] (bias already added by the GEMM) -> Q, K [n,H,Tp,HD] (rows t >= T zero) with the rotary
// embedding on adjacent pairs: (x0, x1) -> (x0 c - x1 s, x0 s + x1 c), c / s = rc / rs[t][pair] (tables [T][HD/2];
// null = no rotation); V -> V^T [n,H,HD,Tp].  grid (Tp/64, H, n), 256 threads.  Generic element type (fp32 parity path).
// ------------------------------------------------------------------------------------------------
template <typename TA, int HD>
__global__ __launch_bounds__(256) void rope2d_split_kernel(const TA* __restrict__ qkv, const float* __restrict__ rc,
                                                           const float* __restrict__ rs, TA* __restrict__ Q,
                                                           TA* __restrict__ K, TA* __restrict__ Vt, int T, int Tp, int H) {
  const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int D = H * HD;
  const long ld = 3L * D;
  const long bh = (long)b * H + h;
  constexpr int HP = HD / 2;
  for (int idx = threadIdx.x; idx < 64 * HP; idx += 256) {
    const int tt = idx / HP, pr = idx - tt * HP;
    const int t = t0 + tt;
    float q0 = 0.f, q1 = 0.f, k0 = 0.f, k1 = 0.f;
    if (t < T) {
      const TA* row = qkv + ((long)b * T + t) * ld + h * HD + 2 * pr;
      load2<TA>(row, q0, q1);
      load2<TA>(row + D, k0, k1);
      if (rc) {
        const float c = rc[(long)t * HP + pr], s = rs[(long)t * HP + pr];
        const float a0 = q0 * c - q1 * s, a1 = q0 * s + q1 * c;
        const float b0 = k0 * c - k1 * s, b1 = k0 * s + k1 * c;
        q0 = a0; q1 = a1; k0 = b0; k1 = b1;
      }
    }
    store2<TA>(Q + (bh * Tp + t) * HD + 2 * pr, q0, q1);
    store2<TA>(K + (bh * Tp + t) * HD + 2 * pr, k0, k1);
  }
  __shared__ float tile[64][HD + 1];
  for (int idx = threadIdx.x; idx < 64 * HD; idx += 256) {
    const int tt = idx / HD, d = idx - tt * HD;
    const int t = t0 + tt;
    tile[tt][d] = t < T ? Elem<TA>::load(qkv + ((long)b * T + t) * ld + 2L * D + h * HD + d) : 0.f;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * HD; idx += 256) {
    const int d = idx >> 6, tt = idx & 63;
    Elem<TA>::store(Vt + (bh * HD + d) * Tp + t0 + tt, tile[tt][d]);
  }
}

// bf16 fast path: HD/8 lanes x 16 bytes per head row, so every global access is a 16-byte load / store; the V tile is
// transposed through LDS as 16-bit words and leaves as 16-byte rows of V^T (same scheme as qkv_prep_bf16_kernel).
template <int HD>
__global__ __launch_bounds__(256) void rope2d_split_bf16_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ rc,
                                                                const float* __restrict__ rs, bf16_t* __restrict__ Q,
                                                                bf16_t* __restrict__ K, bf16_t* __restrict__ Vt, int T,
                                                                int Tp, int H) {
  constexpr int CH = HD / 8;         // 16-byte chunks per head row
  constexpr int RPI = 256 / CH;      // rows per iteration
  constexpr int VS = HD + 8;         // LDS row stride in shorts (spreads banks)
  __shared__ __attribute__((aligned(16))) unsigned short vt[64 * VS];
  const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int D = H * HD;
  const long ld = 3L * D;
  const int tid = threadIdx.x;
  const int sub = tid % CH, rgrp = tid / CH;
  const long bh = (long)b * H + h;
  auto unpack = [](const uint4& v, float (&x)[8]) {
    const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[2 * e] = h16_lo(w4[e]);
      x[2 * e + 1] = h16_hi(w4[e]);
    }
  };
  auto pack = [](const float (&x)[8]) {
    // hardware conversion, as kernels.hip qkv_prep_bf16_kernel since round 4 (no SDWA rounding behind the packed-fp32 rotation)
    return make_uint4(pack_h16x2(x[0], x[1]), pack_h16x2(x[2], x[3]), pack_h16x2(x[4], x[5]), pack_h16x2(x[6], x[7]));
  };
#pragma unroll
  for (int it = 0; it < 64 / RPI; ++it) {
    const int tt = it * RPI + rgrp;
    const int t = t0 + tt;
    uint4 qo = make_uint4(0u, 0u, 0u, 0u), ko = qo, vv = qo;
    if (t < T) {
      const bf16_t* row = qkv + ((long)b * T + t) * ld + h * HD + sub * 8;
      qo = *(const uint4*)row;
      ko = *(const uint4*)(row + D);
      vv = *(const uint4*)(row + 2 * D);
      if (rc) {
        float q[8], k[8];
        unpack(qo, q);
        unpack(ko, k);
        const float4 c4 = *(const float4*)(rc + (long)t * (HD / 2) + sub * 4);
        const float4 s4 = *(const float4*)(rs + (long)t * (HD / 2) + sub * 4);
        const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = q[2 * e], a1 = q[2 * e + 1];
          q[2 * e] = a0 * cc[e] - a1 * sn[e];
          q[2 * e + 1] = a0 * sn[e] + a1 * cc[e];
          const float b0 = k[2 * e], b1 = k[2 * e + 1];
          k[2 * e] = b0 * cc[e] - b1 * sn[e];
          k[2 * e + 1] = b0 * sn[e] + b1 * cc[e];
        }
        qo = pack(q);
        ko = pack(k);
      }
    }
    *(uint4*)(Q + (bh * Tp + t) * HD + sub * 8) = qo;
    *(uint4*)(K + (bh * Tp + t) * HD + sub * 8) = ko;
    *(uint4*)(vt + tt * VS + sub * 8) = vv;
  }
  __syncthreads();
  // V^T rows: item -> (d, 8 consecutive t); HD d x 8 chunks
#pragma unroll
  for (int it = 0; it < HD * 8 / 256; ++it) {
    const int item = it * 256 + tid;
    const int d = item >> 3, c = item & 7;
    unsigned short x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = vt[(c * 8 + e) * VS + d];
    const uint4 o = make_uint4((unsigned)x[0] | ((unsigned)x[1] << 16), (unsigned)x[2] | ((unsigned)x[3] << 16),
                               (unsigned)x[4] | ((unsigned)x[5] << 16), (unsigned)x[6] | ((unsigned)x[7] << 16));
    *(uint4*)(Vt + (bh * HD + d) * Tp + t0 + c * 8) = o;
  }
}

template <int HD>
static hipError_t launch_rope2d_split_t(const void* qkv, const float* rc, const float* rs, void* Q, void* K, void* Vt,
                                        bool bf16, int n, int T, int Tp, int H, hipStream_t st) {
  dim3 grid(Tp / 64, H, n), block(256);
  if (bf16)
    hipLaunchKernelGGL(rope2d_split_bf16_kernel<HD>, grid, block, 0, st, (const bf16_t*)qkv, rc, rs, (bf16_t*)Q, (bf16_t*)K,
                       (bf16_t*)Vt, T, Tp, H);
  else
    hipLaunchKernelGGL((rope2d_split_kernel<float, HD>), grid, block, 0, st, (const float*)qkv, rc, rs, (float*)Q,
                       (float*)K, (float*)Vt, T, Tp, H);
  return hipGetLastError();
}

hipError_t launch_rope2d_split(const void* qkv, const float* rc, const float* rs, void* Q, void* K, void* Vt, bool bf16,
                               int n, int T, int Tp, int H, int head_dim, hipStream_t st) {
  if (head_dim == 64) return launch_rope2d_split_t<64>(qkv, rc, rs, Q, K, Vt, bf16, n, T, Tp, H, st);
  if (head_dim == 128) return launch_rope2d_split_t<128>(qkv, rc, rs, Q, K, Vt, bf16, n, T, Tp, H, st);
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// Attention pooling (AttentionPooling.forward -> nn.MultiheadAttention with ONE query): out[f, h*hd + :] =
// softmax_t(q_h . k[f, t, h] * hd^-0.5) . v[f, t, h];  q [H*hd] f32 is the same for every frame (probe through the
// q projection, precomputed at load); kv rows [n*T, 2*H*hd] = (k | v) from one GEMM.  One workgroup (4 waves) per
// (frame, head): a wave takes tokens w, w+4, ... with an online softmax, lane = 2 (hd = 128) or 1 (hd = 64) channels of
// the head; the four partial (m, l, o) states are merged through LDS.  grid (H, n), 256 threads.
// ------------------------------------------------------------------------------------------------
template <typename TA, int HD>
__global__ __launch_bounds__(256) void pool_attention_kernel(const float* __restrict__ q, const TA* __restrict__ kv,
                                                             TA* __restrict__ out, int T, int H) {
  constexpr int E = HD / 64;  // channels per lane
  const int h = blockIdx.x, f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = H * HD;
  const float scale = HD == 128 ? 0.08838834764831845f : 0.125f;
  float qv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) qv[e] = q[h * HD + lane * E + e] * scale;
  float m = -INFINITY, l = 0.f, o[E];
#pragma unroll
  for (int e = 0; e < E; ++e) o[e] = 0.f;
  for (int t = wave; t < T; t += 4) {
    const TA* krow = kv + ((long)f * T + t) * (2L * D) + h * HD + lane * E;
    float kk[E], vv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { kk[e] = Elem<TA>::load(krow + e); vv[e] = Elem<TA>::load(krow + D + e); }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += qv[e] * kk[e];
    s = wave_sum(s);
    const float m_new = fmaxf(m, s);
    const float a = expf(m - m_new), pv = expf(s - m_new);
    l = l * a + pv;
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = o[e] * a + pv * vv[e];
    m = m_new;
  }
  __shared__ float sm[4], sl[4], so[4][HD];
  if (lane == 0) { sm[wave] = m; sl[wave] = l; }
#pragma unroll
  for (int e = 0; e < E; ++e) so[wave][lane * E + e] = o[e];
  __syncthreads();
  if (wave == 0) {
    float mm = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float ll = 0.f, oo[E];
#pragma unroll
    for (int e = 0; e < E; ++e) oo[e] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float a = sm[w] == -INFINITY ? 0.f : expf(sm[w] - mm);  // a wave with no token (T < 4) contributes nothing
      ll += sl[w] * a;
#pragma unroll
      for (int e = 0; e < E; ++e) oo[e] += so[w][lane * E + e] * a;
    }
    const float inv = 1.f / ll;
    if constexpr (E == 2) {   // one pair through store2 (hardware conversion for 16-bit outputs: common.h)
      store2<TA>(out + (long)f * D + h * HD + lane * E, oo[0] * inv, oo[1] * inv);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) Elem<TA>::store(out + (long)f * D + h * HD + lane * E + e, oo[e] * inv);
    }
  }
}

hipError_t launch_pool_attention(const float* q, const void* kv, void* out, bool bf16, int n, int T, int H, int head_dim,
                                 hipStream_t st) {
  dim3 grid(H, n), block(256);
  if (head_dim == 128) {
    if (bf16) hipLaunchKernelGGL((pool_attention_kernel<bf16_t, 128>), grid, block, 0, st, q, (const bf16_t*)kv, (bf16_t*)out, T, H);
    else hipLaunchKernelGGL((pool_attention_kernel<float, 128>), grid, block, 0, st, q, (const float*)kv, (float*)out, T, H);
  } else if (head_dim == 64) {
    if (bf16) hipLaunchKernelGGL((pool_attention_kernel<bf16_t, 64>), grid, block, 0, st, q, (const bf16_t*)kv, (bf16_t*)out, T, H);
    else hipLaunchKernelGGL((pool_attention_kernel<float, 64>), grid, block, 0, st, q, (const float*)kv, (float*)out, T, H);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// x[r, :] /= max(||x[r, :]||_2, 1e-12)   (F.normalize, `encode_image(normalize=True)`); one wave per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_normalize_kernel(float* __restrict__ x, int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* xr = x + (long)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i] * xr[i];
  const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  for (int i = lane; i < D; i += 64) xr[i] *= inv;
}

hipError_t launch_l2_normalize(float* x, int rows, int D, hipStream_t st) {
  hipLaunchKernelGGL(l2_normalize_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, rows, D);
  return hipGetLastError();
}

// mean over the T tokens of each frame: out[f, :] = AT(mean_t x[f, t, :])  (pool_type "avg"); grid (n), 256 threads
template <typename TO>
__global__ __launch_bounds__(256) void token_mean_kernel(const float* __restrict__ x, long x_ld, float* __restrict__ out_f32,
                                                         TO* __restrict__ out_act, int T, int D, int t_lo) {
  const int f = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int t = t_lo; t < T; ++t) acc += x[((long)f * T + t) * x_ld + d];
    acc /= (float)(T - t_lo);
    if (out_f32) out_f32[(long)f * D + d] = acc;
    if (out_act) Elem<TO>::store(out_act + (long)f * D + d, acc);
  }
}

hipError_t launch_token_mean(const float* x, long x_ld, float* out_f32, void* out_act, bool bf16, int n, int T, int D,
                             hipStream_t st) {
  if (bf16) hipLaunchKernelGGL(token_mean_kernel<bf16_t>, dim3(n), dim3(256), 0, st, x, x_ld, out_f32, (bf16_t*)out_act, T, D, 0);
  else hipLaunchKernelGGL(token_mean_kernel<float>, dim3(n), dim3(256), 0, st, x, x_ld, out_f32, (float*)out_act, T, D, 0);
  return hipGetLastError();
}

}  // namespace sa
