// Per-CU pipe probe for gfx950 (standalone; build with tools/build_probe.sh, run on the GPU box).
//
// Question it answers: in the bf16 GEMM main loop, how do the three streams of one CU -
// v_mfma_f32_32x32x16_bf16 issue, global_load_lds_dwordx4 fill pieces (1 KiB per wave instruction) and
// ds_read_b128 fragment reads - interfere when they are issued (a) from the same wave, (b) from the two waves that
// share a SIMD, (c) from dedicated waves?  Every wave of a workgroup gets a ROLE = (M, D, R): per iteration it issues
// D fill pieces, R fragment reads and M MFMAs, free-running (no barriers), and reports its shader-clock cycles per
// iteration.  One workgroup per CU (128 KiB of LDS), 256 workgroups, every CU busy.
//
//   ./cu_probe            -> table of cycles / iteration per role for the built-in scenarios
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

struct Params {
  const char* src;       // global window the fill pieces read from
  long window;           // bytes (power of two)
  int shared;            // 1: all workgroups walk the same window (L2 hits), 0: one window per workgroup
  int iters;
  int barrier;           // 1: every iteration ends with an s_barrier over all waves (the real kernels' per-slab coupling)
  int role[12];          // per wave: index into the role table below, -1 = wave exits at once
  unsigned long long* out;  // [workgroup][12] cycles for `iters` iterations
  float* sink;
};

constexpr int LDS_BYTES = 128 * 1024;

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One role: per iteration D fill pieces (at most D outstanding from the previous iteration while the next is issued),
// R ds_read_b128, M MFMAs.  ILV: spread the pieces / reads between the MFMAs instead of issuing them first.
template <int M, int D, int R, bool ILV>
__device__ __forceinline__ void run_role(const Params& p, char* smem, int wave, int lane, unsigned long long* cyc, float* sink) {
  f32x16_t acc[4];  // 4 independent accumulators: 128 cycles between dependent MFMAs
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  bf16x8_t x, y;
#pragma unroll
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.001f * (lane + e)); y[e] = (__bf16)(0.002f * (lane - e)); }
  const long wmask = p.window - 1;
  const char* src = p.src + (p.shared ? 0 : (long)blockIdx.x * p.window);
  long off = ((long)blockIdx.x * 12 + wave) * 65536 + lane * 16;  // every wave streams through the window from its own start
  char* my_lds = smem + wave * 8192;                                // 8 KiB ring per wave
  const char* rd = smem + ((wave & 3) * 16384) + lane * 16;         // conflict-free b128 reads
  f32x4_t r[R > 0 ? R : 1];
  int slot = 0;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < p.iters; ++it) {
    if (D > 0) wait_vmcnt<D>();
    if (!ILV) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        dma16(src + (off & wmask), my_lds + ((slot + d) & 7) * 1024);
        off += 1024;
      }
#pragma unroll
      for (int i = 0; i < R; ++i) r[i] = *(const f32x4_t*)(rd + i * 1024);
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int m = 0; m < (M > 0 ? M : 1); ++m) {
        if (D > 0 && m < D) {
          dma16(src + (off & wmask), my_lds + ((slot + m) & 7) * 1024);
          off += 1024;
        }
        if (R > 0 && m < R) r[m] = *(const f32x4_t*)(rd + m * 1024);
        if (M > 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    slot = (slot + D) & 7;
    if (R > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < R; ++i) asm volatile("" ::"v"(r[i]));
    }
    if (p.barrier) __builtin_amdgcn_s_barrier();
  }
  wait_vmcnt<0>();
  const unsigned long long t1 = clock64();
  *cyc = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][7];
  if (s == 12345.678f) sink[0] = s;
}

// role table: {M, D, R, ILV}
//  0 MFMA only (16)            1 fill only (8 / iter)          2 fill only (16 / iter)      3 reads only (12)
//  4 MFMA 16 + reads 12        5 MFMA 16 + fill 4 + reads 12   6 MFMA 16 + fill 4           7 same as 5, interleaved
//  8 MFMA 16 + reads 12 interleaved                            9 fill only (4 / iter)
__global__ __launch_bounds__(768) void probe_kernel(const Params p) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < LDS_BYTES / 4; i += blockDim.x) ((float*)smem)[i] = 0.f;
  __syncthreads();
  const int role = p.role[wave];
  unsigned long long cyc = 0;
  switch (role) {
    case 0: run_role<16, 0, 0, false>(p, smem, wave, lane, &cyc, p.sink); break;
    case 1: run_role<0, 8, 0, false>(p, smem, wave, lane, &cyc, p.sink); break;
    case 2: run_role<0, 16, 0, false>(p, smem, wave, lane, &cyc, p.sink); break;
    case 3: run_role<0, 0, 12, false>(p, smem, wave, lane, &cyc, p.sink); break;
    case 4: run_role<16, 0, 12, false>(p, smem, wave, lane, &cyc, p.sink); break;
    case 5: run_role<16, 4, 12, false>(p, smem, wave, lane, &cyc, p.sink); break;
    case 6: run_role<16, 4, 0, false>(p, smem, wave, lane, &cyc, p.sink); break;
    case 7: run_role<16, 4, 12, true>(p, smem, wave, lane, &cyc, p.sink); break;
    case 8: run_role<16, 0, 12, true>(p, smem, wave, lane, &cyc, p.sink); break;
    case 9: run_role<0, 4, 0, false>(p, smem, wave, lane, &cyc, p.sink); break;
    default: __builtin_amdgcn_s_barrier(); break;  // idle wave: take part in the start barrier, then exit
  }
  if (lane == 0) p.out[(long)blockIdx.x * 12 + wave] = cyc;
}

struct RoleInfo { int M, D, R; const char* name; };
static const RoleInfo kRoles[] = {
    {16, 0, 0, "mfma16"},          {0, 8, 0, "fill8"},           {0, 16, 0, "fill16"},      {0, 0, 12, "read12"},
    {16, 0, 12, "mfma16+read12"},  {16, 4, 12, "mfma16+fill4+read12"}, {16, 4, 0, "mfma16+fill4"},
    {16, 4, 12, "mfma16+fill4+read12 ilv"}, {16, 0, 12, "mfma16+read12 ilv"}, {0, 4, 0, "fill4"}};

struct Scenario { const char* name; int nwaves; int role[12]; };

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 2000;
  int dev = 0, cus = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = cus;
  const long window = 2L << 20;  // 2 MiB
  char* src = nullptr;
  CHECK(hipMalloc(&src, window * grid));
  CHECK(hipMemset(src, 0, window * grid));
  unsigned long long* out = nullptr;
  CHECK(hipMalloc(&out, sizeof(unsigned long long) * 12 * grid));
  float* sink = nullptr;
  CHECK(hipMalloc(&sink, 64));
  std::vector<unsigned long long> host(12 * grid);

  // W(k) = waves 0..3 sit on SIMD 0..3 (one each), 4..7 second wave per SIMD, 8..11 third wave per SIMD
  const Scenario sc[] = {
      {"mfma x4 (1/SIMD)", 4, {0, 0, 0, 0}},
      {"mfma x8 (2/SIMD)", 8, {0, 0, 0, 0, 0, 0, 0, 0}},
      {"fill8 x1", 1, {1}},
      {"fill8 x4", 4, {1, 1, 1, 1}},
      {"fill8 x8", 8, {1, 1, 1, 1, 1, 1, 1, 1}},
      {"fill16 x4", 4, {2, 2, 2, 2}},
      {"read12 x4", 4, {3, 3, 3, 3}},
      {"read12 x8", 8, {3, 3, 3, 3, 3, 3, 3, 3}},
      {"(mfma+read) x8", 8, {4, 4, 4, 4, 4, 4, 4, 4}},
      {"(mfma+read ilv) x8", 8, {8, 8, 8, 8, 8, 8, 8, 8}},
      {"(mfma+fill4+read) x8   [8-wave kernels]", 8, {5, 5, 5, 5, 5, 5, 5, 5}},
      {"(mfma+fill4+read ilv) x8", 8, {7, 7, 7, 7, 7, 7, 7, 7}},
      {"(mfma+fill4) x8", 8, {6, 6, 6, 6, 6, 6, 6, 6}},
      {"mfma x8 + fill8 x4      [loader waves, no reads]", 12, {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1}},
      {"(mfma+read) x8 + fill8 x4   [gemm5 256x256]", 12, {4, 4, 4, 4, 4, 4, 4, 4, 1, 1, 1, 1}},
      {"(mfma+read ilv) x8 + fill8 x4", 12, {8, 8, 8, 8, 8, 8, 8, 8, 1, 1, 1, 1}},
      {"(mfma+read) x8 + fill16 x4", 12, {4, 4, 4, 4, 4, 4, 4, 4, 2, 2, 2, 2}},
      {"(mfma+read) x8 + fill4 x4", 12, {4, 4, 4, 4, 4, 4, 4, 4, 9, 9, 9, 9}},
      {"(mfma+read) x4 + fill8 x4   [1 compute wave/SIMD]", 8, {4, 4, 4, 4, 1, 1, 1, 1}},
  };
  for (int pass = 0; pass < 3; ++pass) {
    const int shared = pass != 2, barrier = pass == 1;
    printf("==== fill source: %s 2 MiB window(s), %s, %d workgroups, %d iterations ====\n",
           shared ? "one SHARED (L2-resident)" : "one PRIVATE per workgroup (mostly L2 misses)",
           barrier ? "s_barrier after every iteration" : "free-running waves", grid, iters);
    for (const Scenario& s : sc) {
      Params p;
      memset(&p, 0, sizeof(p));
      p.src = src; p.window = window; p.shared = shared; p.iters = iters; p.out = out; p.sink = sink;
      p.barrier = barrier;
      for (int w = 0; w < 12; ++w) p.role[w] = w < s.nwaves ? s.role[w] : -1;
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(64 * s.nwaves), 0, 0, p);  // warm-up
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(64 * s.nwaves), 0, 0, p);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipDeviceSynchronize());
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(host.data(), out, sizeof(unsigned long long) * 12 * grid, hipMemcpyDeviceToHost));
      printf("%-52s wall %8.3f ms\n", s.name, ms);
      // per distinct role: mean cycles / iteration over all waves with that role
      for (int r = 0; r < (int)(sizeof(kRoles) / sizeof(kRoles[0])); ++r) {
        double sum = 0;
        long n = 0;
        for (int g = 0; g < grid; ++g)
          for (int w = 0; w < s.nwaves; ++w)
            if (s.role[w] == r) { sum += (double)host[(long)g * 12 + w]; ++n; }
        if (!n) continue;
        const double cyc = sum / n / iters;
        const int waves = (int)(n / grid);
        const RoleInfo& ri = kRoles[r];
        printf("    %-26s x%-2d %9.1f cyc/iter", ri.name, waves, cyc);
        if (ri.M) printf("  | mfma pipe need %4d cyc/iter/wave, SIMD share %5.1f %%", ri.M * 32,
                         100.0 * ri.M * 32 * ((waves + 3) / 4) / cyc);
        if (ri.D) printf("  | %6.1f cyc/piece/CU, %6.1f B/cyc/CU", cyc / (ri.D * waves), 1024.0 * ri.D * waves / cyc);
        if (ri.R) printf("  | reads %6.1f B/cyc/CU", 1024.0 * ri.R * waves / cyc);
        printf("\n");
      }
      CHECK(hipEventDestroy(e0));
      CHECK(hipEventDestroy(e1));
    }
  }
  return 0;
}
