// Stand-in for <hip/hip_runtime.h>  --  TEST INFRASTRUCTURE ONLY (oracle/simt/).
//
// With this directory first on the include path, the product's .hip sources (kernels AND host code) compile UNCHANGED
// as plain C++ for the host, and every `hipLaunchKernelGGL` runs the real kernel body on a functional SIMT simulator:
// one fiber per GPU thread, 64-lane wavefronts, workgroup barriers, LDS as `static` arrays, and the gfx950 builtins the
// kernels use (MFMA bf16 / f32, global_load_lds, DPP row rotations, shuffles, readfirstlane) emulated lane-exactly.
// It checks indexing, fragment layouts, swizzles, masks and epilogues - the FUNCTION of a kernel.  It does not model
// time, occupancy, bank conflicts, or the asynchrony of DMA / s_waitcnt (loads complete at issue), so it cannot see
// races: those and all performance questions stay with the MI355X runs.  Nothing under sam_audio_amd/ uses it.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------------------- host runtime
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotSupported = 801 };
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "simulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)std::malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// HIP graphs (SAMAUDIO_OPT_ODE_GRAPH): not simulated - capture is refused and the engine runs its launches eagerly
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
enum { hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned) { return hipErrorNotSupported; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
inline hipError_t hipGetDevice(int* dev) { *dev = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 16; return hipSuccess; }  // small "chip":
// persistent kernels then walk several tiles per workgroup even in small tests
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = std::malloc(8); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---------------------------------------------------------------------------------------------------- simulator
namespace simt {
constexpr int WAVE = 64;
constexpr int XBYTES = 64;  // bytes one lane can publish per collective

struct Wave {
  int live = 0, arrived = 0;
  unsigned gen = 0;
  int row_arrived[4] = {0, 0, 0, 0};   // DPP row operations synchronise the 16 lanes of one row only
  unsigned row_gen[4] = {0, 0, 0, 0};
  alignas(64) unsigned char xbuf[2][WAVE][XBYTES];
  alignas(64) unsigned char rbuf[2][WAVE][8];
  // global_load_lds instructions issued by this wave and not yet retired by one of its s_waitcnt vmcnt(N)
  // (only used in the "late" DMA mode, see dma_late())
  struct Dma { char* base; unsigned char data[WAVE][16]; bool visible = true; };   // visible: issued through the builtin (the compiler
                                                                                   // knows it; an inline-assembly DMA it does not)
  std::deque<Dma> pending;
  unsigned long dma_first = 0;  // sequence number of pending.front()
};
struct Block {
  int live = 0, arrived = 0;
  unsigned gen = 0;
};
struct Fiber {
  void* sp = nullptr;
  void* stack = nullptr;
  dim3 tid, bid, bdim, gdim;
  int lane = 0;
  unsigned xcount = 0, rcount = 0;
  unsigned long dma_seq = 0;  // global_load_lds instructions this lane has issued
  Wave* wave = nullptr;
  Block* block = nullptr;
  bool done = false;
  // wait condition: 0 none, 1 wave gen != wgen, 2 block gen != bgen, 3 row gen != rgen
  int waiting = 0;
  unsigned wgen = 0, bgen = 0, rgen = 0;
};
extern thread_local Fiber* cur;  // the workgroups of a launch run on several OS threads (OpenMP), one scheduler each
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void wave_sync();
void block_sync();
// publish n <= XBYTES bytes; returns the wave's exchange area [WAVE][XBYTES] once every live lane has published
const unsigned char* wave_publish(const void* data, size_t n);
// the same among the 16 lanes of the caller's DPP row (lanes 16k .. 16k+15), n <= 8; all 16 must take part
const unsigned char* row_publish(const void* data, size_t n);
// DMA model.  Default ("early"): a global_load_lds lands in LDS when it is issued.  SAMAUDIO_SIMT_DMA=late: it lands as
// LATE as the ISA allows - when the issuing wave's s_waitcnt vmcnt(N) retires it (oldest first, until N remain), or at
// that wave's __syncthreads() (which the compiler precedes with vmcnt(0)).  A kernel whose vmcnt counting lets a wave
// read a slab before it was retired then reads stale LDS and fails its parity test; "early" catches the opposite
// hazard (a slab overwritten while still being read).  Only DMA counts: other VMEM loads a wave has in flight would
// make the real vmcnt(N) retire fewer DMAs than this model assumes.
bool dma_late();
void wait_vmcnt(int n);   // wave-synchronous
void wait_vmcnt_if_visible();   // __syncthreads(): vmcnt(0) iff a compiler-visible DMA is pending
extern thread_local bool dma_asm;   // the next global_load_lds stands for an inline-assembly DMA (set by the rewritten asm lines)
void asm_stmt(const char* text);
}  // namespace simt

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  simt::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#ifdef SIMT_POISON
// poison build (libsamaudio_simt_poison.so): every LDS array lives in one linker section that the launcher fills with
// 0xFF bytes (NaN as fp32 and bf16) before EACH workgroup - the LDS of a real CU is not cleared between workgroups or
// kernels either, so a kernel that consumes LDS it never wrote (the stale-LDS NaN bug of DESIGN.md section 8) fails
// deterministically here.  One workgroup at a time in this build (the section is shared).
#define __shared__ static __attribute__((section("simt_lds")))
#else
#define __shared__ static thread_local   // one LDS image per OS thread = per concurrently simulated workgroup
#endif
#define threadIdx (simt::cur->tid)
#define blockIdx (simt::cur->bid)
#define blockDim (simt::cur->bdim)
#define gridDim (simt::cur->gdim)
inline void __syncthreads() {
  // hipcc emits s_waitcnt vmcnt(0) before the s_barrier ONLY for direct-to-LDS loads it can see (the builtin); a DMA issued as
  // inline assembly (gemm2.hip dma16a, gemm8.hip dma16s / dma16v) is retired by the kernel's explicit s_waitcnt vmcnt alone
  // (ADVICE round 3).  vmcnt retires in order, so one visible DMA in flight drains everything.
  if (simt::dma_late()) simt::wait_vmcnt_if_visible();
  simt::block_sync();
}

// ---------------------------------------------------------------------------------------------------- vector types
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct ushort2 { unsigned short x, y; };
struct ushort4 { unsigned short x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

// ---------------------------------------------------------------------------------------------------- device math
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline float __expf(float x) { return std::exp(x); }
inline float __sinf(float x) { return std::sin(x); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __cosf(float x) { return std::cos(x); }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }

// ---------------------------------------------------------------------------------------------------- wave collectives
inline float __shfl_xor(float v, int mask, int width = 64) {
  (void)width;
  const unsigned char* x = simt::wave_publish(&v, 4);
  float r;
  std::memcpy(&r, x + (size_t)((simt::cur->lane ^ mask) & 63) * simt::XBYTES, 4);
  return r;
}
inline float __shfl(float v, int src_lane, int width = 64) {
  (void)width;
  const unsigned char* x = simt::wave_publish(&v, 4);
  float r;
  std::memcpy(&r, x + (size_t)(src_lane & 63) * simt::XBYTES, 4);
  return r;
}
inline int __builtin_amdgcn_readfirstlane(int v) {
  const unsigned char* x = simt::wave_publish(&v, 4);
  int r;
  std::memcpy(&r, x, 4);  // kernels call it with every lane active
  return r;
}
// DPP: only the row rotations the kernels use (ctrl 0x121..0x12F = row_ror:1..15); the reductions built on them do not
// depend on the rotation direction
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  (void)old; (void)row_mask; (void)bank_mask; (void)bound_ctrl;
  const unsigned char* x = simt::row_publish(&src, 4);  // lanes of other rows may be masked off (EXEC)
  const int l = simt::cur->lane;
  if (ctrl < 0x121 || ctrl > 0x12F) { std::fprintf(stderr, "simt: unsupported DPP control 0x%x\n", ctrl); std::abort(); }
  const int n = ctrl - 0x120;
  const int from = (l & ~15) | (((l & 15) + 16 - n) & 15);  // row_ror:n - data moves n lanes to the right
  int r;
  std::memcpy(&r, x + (size_t)from * 8, 4);
  return r;
}
// v_med3_f32
inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
// v_permlane16_swap_b32: the odd 16-lane rows of `vdst` swap with the even rows of `src` (lane l of an odd row <-> lane
// l - 16); returns {new vdst, new src}.  Every lane of the wave takes part (as on the hardware: EXEC-masked lanes read
// garbage there).
struct simt_u32x2 {
  unsigned v[2];
  unsigned operator[](int i) const { return v[i]; }
};
inline simt_u32x2 __builtin_amdgcn_permlane16_swap(unsigned vdst, unsigned src, bool fi, bool bound_ctrl) {
  (void)fi; (void)bound_ctrl;
  const unsigned both[2] = {vdst, src};
  const unsigned char* x = simt::wave_publish(both, 8);
  const int l = simt::cur->lane;
  unsigned other[2];
  std::memcpy(other, x + (size_t)(l ^ 16) * simt::XBYTES, 8);
  simt_u32x2 r;
  if ((l >> 4) & 1) { r.v[0] = other[1]; r.v[1] = src; }   // odd row: vdst <- the even partner's src
  else { r.v[0] = vdst; r.v[1] = other[0]; }                // even row: src <- the odd partner's vdst
  return r;
}
inline void __builtin_amdgcn_s_waitcnt(int) { simt::wave_sync(); }   // lgkmcnt forms only (gemm8.hip)
inline void __builtin_amdgcn_s_barrier() { simt::block_sync(); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
inline void __builtin_amdgcn_s_setprio(int) {}

// global -> LDS DMA (global_load_lds_dwordx4): lane i's 16 bytes land at (wave-uniform LDS base) + 16*i; the base is the
// first lane's pointer (M0 is a scalar register).  Completes at issue in the simulation.
inline void __builtin_amdgcn_global_load_lds(const __attribute__((address_space(1))) void* g,
                                             __attribute__((address_space(3))) void* lds, int size, int off, int aux) {
  (void)aux;
  if (size != 16) { std::fprintf(stderr, "simt: global_load_lds size %d\n", size); std::abort(); }
  void* mine = (void*)lds;
  const unsigned char* x = simt::wave_publish(&mine, sizeof(void*));
  void* base;
  std::memcpy(&base, x, sizeof(void*));
  simt::Fiber* f = simt::cur;
  if (!simt::dma_late()) {
    std::memcpy((char*)base + off + 16 * f->lane, (const void*)g, 16);
    return;
  }
  simt::Wave* w = f->wave;
  const unsigned long seq = f->dma_seq++;
  while (w->dma_first + w->pending.size() <= seq) {
    w->pending.emplace_back();
    w->pending.back().base = (char*)base + off;
    w->pending.back().visible = !simt::dma_asm;
  }
  std::memcpy(w->pending[seq - w->dma_first].data[f->lane], (const void*)g, 16);
}

// ---------------------------------------------------------------------------------------------------- MFMA
namespace simt {
typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(16))) float v16f;
inline float bf(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; std::memcpy(&f, &u, 4); return f; }
struct AB16 { unsigned short a[8], b[8]; };
}  // namespace simt

// D[i][j] = C[i][j] + sum_k A[i][k] B[k][j], 16x16x32 bf16.  Lane l holds A[i = l&15][k = 8*(l>>4) .. +8],
// B[k = 8*(l>>4) .. +8][j = l&15]; C/D: col = l&15, row = 4*(l>>4) + reg.
inline simt::v4f __builtin_amdgcn_mfma_f32_16x16x32_bf16(simt::v8bf a, simt::v8bf b, simt::v4f c, int, int, int) {
  simt::AB16 m;
  std::memcpy(m.a, &a, 16);
  std::memcpy(m.b, &b, 16);
  const unsigned char* x = simt::wave_publish(&m, sizeof(m));
  const int l = simt::cur->lane, j = l & 15;
  float bcol[32];
  for (int kk = 0; kk < 4; ++kk) {
    const simt::AB16* pb = (const simt::AB16*)(x + (size_t)(j + 16 * kk) * simt::XBYTES);
    for (int e = 0; e < 8; ++e) bcol[8 * kk + e] = simt::bf(pb->b[e]);
  }
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l >> 4) + r;
    float s = 0.f;
    for (int kk = 0; kk < 4; ++kk) {
      const simt::AB16* pa = (const simt::AB16*)(x + (size_t)(i + 16 * kk) * simt::XBYTES);
      for (int e = 0; e < 8; ++e) s += simt::bf(pa->a[e]) * bcol[8 * kk + e];
    }
    c[r] += s;
  }
  return c;
}
// 32x32x16 bf16: lane l holds A[i = l&31][k = 8*(l>>5) .. +8], B[k = 8*(l>>5) .. +8][j = l&31];
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
inline simt::v16f __builtin_amdgcn_mfma_f32_32x32x16_bf16(simt::v8bf a, simt::v8bf b, simt::v16f c, int, int, int) {
  simt::AB16 m;
  std::memcpy(m.a, &a, 16);
  std::memcpy(m.b, &b, 16);
  const unsigned char* x = simt::wave_publish(&m, sizeof(m));
  const int l = simt::cur->lane, j = l & 31;
  float bcol[16];
  for (int kk = 0; kk < 2; ++kk) {
    const simt::AB16* pb = (const simt::AB16*)(x + (size_t)(j + 32 * kk) * simt::XBYTES);
    for (int e = 0; e < 8; ++e) bcol[8 * kk + e] = simt::bf(pb->b[e]);
  }
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float s = 0.f;
    for (int kk = 0; kk < 2; ++kk) {
      const simt::AB16* pa = (const simt::AB16*)(x + (size_t)(i + 32 * kk) * simt::XBYTES);
      for (int e = 0; e < 8; ++e) s += simt::bf(pa->a[e]) * bcol[8 * kk + e];
    }
    c[r] += s;
  }
  return c;
}
// 16x16x4 f32 (exact fp32, k-ordered fma chain): lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15].
inline simt::v4f __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, simt::v4f c, int, int, int) {
  float m[2] = {a, b};
  const unsigned char* x = simt::wave_publish(m, 8);
  const int l = simt::cur->lane, j = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l >> 4) + r;
    float s = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      std::memcpy(&av, x + (size_t)(i + 16 * k) * simt::XBYTES, 4);
      std::memcpy(&bv, x + (size_t)(j + 16 * k) * simt::XBYTES + 4, 4);
      s = std::fma(av, bv, s);
    }
    c[r] = s;
  }
  return c;
}

// ---------------------------------------------------------------------------------------------------- inline asm
// `asm volatile("s_waitcnt ..." ...)` statements have no host meaning; each becomes a wave-level synchronisation point
// (every lane of a wave executes them together on the GPU).  Must stay the LAST thing this header does: the standard
// headers above are already included and guarded.
#define asm
#define volatile(...) simt::asm_stmt(#__VA_ARGS__)
