// Micro-benchmark (tools only): can a wave's LDS / VMEM / VALU instructions issue under its own MFMAs, and under another
// wave's MFMAs on the same SIMD?  One block per CU; every variant executes the same number of MFMAs per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_overlap.hip -o tools/ubench/mfma_overlap && tools/ubench/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)

// mode 0: MFMA only (1 wave/SIMD, 64 MFMAs per iteration)
// mode 1: same + one ds_read_b128 after every second MFMA (32 per iteration), same wave
// mode 2: same + one independent v_add after every MFMA, same wave
// mode 3: 2 waves/SIMD: waves 0-3 = mode 0, waves 4-7 issue only ds_read_b128 (32 per iteration)
// mode 4: 2 waves/SIMD, both MFMA only, 32 MFMAs per iteration each
// mode 5: 2 waves/SIMD, both: 32 MFMAs + 16 ds_read interleaved (the fine-grained interleave of the 8-wave GEMM)
// mode 6: 2 waves/SIMD, phases: wave group A does 32 MFMAs while group B does 16 ds_reads, barrier, swap
// mode 7: mode 1 with buffer_load ... lds (LDS-DMA) instead of ds_read (16 per iteration)
// mode 8: 1 wave/SIMD, MFMA + ds_read_b128 after EVERY MFMA (64 per iteration), swizzled fragment addresses
// mode 9: 1 wave/SIMD, the v7 stage without data dependences: 64 MFMAs + 32 swizzled ds_read_b128 + 16 LDS-DMA
// mode 10: mode 9 + s_waitcnt vmcnt(0) lgkmcnt(0) + block barrier per 64 MFMAs
// mode 11: mode 9 with 2 ds_reads after every MFMA (128 per iteration): LDS bandwidth probe
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, const char* src, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
  const unsigned laddr = (unsigned)(w * 8192 + lane * 16);
  u32x4 sink = {0, 0, 0, 0};
  int vsink = lane;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
  const bool mf = (MODE == 3) ? (w < 4) : true;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0 || MODE == 4) {
      constexpr int N = MODE == 0 ? 64 : 32;
#pragma unroll
      for (int n = 0; n < N; ++n) MFMA(acc[n & 7], a, b);
    } else if constexpr (MODE == 1 || MODE == 5) {
      constexpr int N = MODE == 1 ? 64 : 32;
#pragma unroll
      for (int n = 0; n < N; ++n) {
        MFMA(acc[n & 7], a, b);
        __builtin_amdgcn_sched_barrier(0);
        if (n & 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink) : "v"(laddr), "n"((n >> 1) * 1024 % 8192));
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int n = 0; n < 64; ++n) {
        MFMA(acc[n & 7], a, b);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(vsink) : "v"(lane));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (MODE == 3) {
      if (mf) {
#pragma unroll
        for (int n = 0; n < 64; ++n) MFMA(acc[n & 7], a, b);
      } else {
#pragma unroll
        for (int n = 0; n < 32; ++n) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink) : "v"(laddr), "n"(n * 1024 % 8192));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else if constexpr (MODE == 6) {
      const bool first = w < 4;
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        if (first == (ph == 0)) {
#pragma unroll
          for (int n = 0; n < 32; ++n) MFMA(acc[n & 7], a, b);
        } else {
#pragma unroll
          for (int n = 0; n < 16; ++n) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink) : "v"(laddr), "n"(n * 1024 % 8192));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
    } else if constexpr (MODE >= 8) {
      const unsigned fa = (unsigned)((w >> 1) * 16384 + (lane & 31) * 128 + ((((lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4));
#pragma unroll
      for (int n = 0; n < 64; ++n) {
        MFMA(acc[n & 7], a, b);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 8 || MODE == 11 || (n & 1)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink) : "v"(fa ^ (unsigned)(((n >> 4) & 3) << 5)), "n"((n & 7) * 4096));
        if (MODE == 11) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink) : "v"(fa ^ (unsigned)(((n >> 4) & 3) << 5)), "n"(32768 + (n & 7) * 4096));
        if (MODE >= 9 && (n & 3) == 3)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 65536 + w * 1024 + (n >> 2) * 4096), 16, lane * 16, (n >> 2) * 4096, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 10) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (MODE == 7) {
#pragma unroll
      for (int n = 0; n < 64; ++n) {
        MFMA(acc[n & 7], a, b);
        __builtin_amdgcn_sched_barrier(0);
        if ((n & 3) == 3)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + w * 16384 + (n >> 2) * 1024), 16, lane * 16, (n >> 2) * 4096, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sink[0] + (float)vsink;
}

template <int MODE>
void run(const char* name, int threads, float* out, const char* src) {
  const int iters = 2000, grid = 256;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, threads, 131072>>>(out, src, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, threads, 131072>>>(out, src, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // 64 MFMAs per SIMD per iteration in every mode: 256 CUs x 4 SIMDs x 64 x 32768 flop
  const double tf = 256.0 * 4 * 64 * 32768.0 * iters / (ms * 1e-3) / 1e12;
  printf("%-72s %8.3f ms  %7.0f TF/s  %6.1f ns per MFMA slot\n", name, ms, tf, ms * 1e6 / (iters * 64.0));
}

int main() {
  float* out; char* src;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&src, 1 << 21);
  hipMemset(src, 0, 1 << 21);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("0: 1 wave/SIMD, MFMA only", 256, out, src);
    run<8>("8: 1 wave/SIMD, MFMA + swizzled ds_read_b128 every MFMA", 256, out, src);
    run<11>("11: 1 wave/SIMD, MFMA + 2 ds_read_b128 every MFMA + DMA every 4th", 256, out, src);
    run<9>("9: 1 wave/SIMD, v7 stage mix: 64 MFMA + 32 ds_read + 16 LDS-DMA, no dependences", 256, out, src);
    run<10>("10: mode 9 + wait + block barrier per stage", 256, out, src);
    run<1>("1: 1 wave/SIMD, MFMA + ds_read_b128 every 2nd (same wave)", 256, out, src);
    run<2>("2: 1 wave/SIMD, MFMA + v_add every MFMA (same wave)", 256, out, src);
    run<7>("7: 1 wave/SIMD, MFMA + LDS-DMA every 4th (same wave)", 256, out, src);
    run<4>("4: 2 waves/SIMD, both MFMA only", 512, out, src);
    run<3>("3: 2 waves/SIMD, one MFMA only, the other ds_read only", 512, out, src);
    run<5>("5: 2 waves/SIMD, both MFMA + ds_read interleaved", 512, out, src);
    run<6>("6: 2 waves/SIMD, alternating phases (MFMA burst | reads), block barrier", 512, out, src);
  }
  return 0;
}
