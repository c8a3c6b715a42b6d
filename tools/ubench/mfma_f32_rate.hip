// Micro-benchmark (tools only; round 6): what does the f32-input MFMA (v_mfma_f32_32x32x2_f32) sustain on this part with REAL operands?
// conv_f32_kernel (omg_amd/csrc/gemm_f32.hip) measures 121-127 TF/s on every shape of the VAE up blocks (0.78-0.81 of the 157.3 TF/s peak);
// the guide's 155 TF/s is a register-only loop.  Modes: operands all zero | N(0,1); 1 or 2 waves per SIMD; MFMA only | + the kernel's 16
// ds_read_b128 per 64 MFMAs | + its 8 global_load_lds and one barrier per stage.  Rates only, results unchecked.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_rate.hip -o tools/ubench/mfma_f32_rate && tools/ubench/mfma_f32_rate [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int MIXB>
__global__ __launch_bounds__(256, 2) void k(float* out, const char* src, long src_bytes, int stages) {
  constexpr bool BUF = MIXB >= 1000;          // 1000 + MIX: the same mode with descriptor addressing (raw_ptr_buffer_load_lds, 32-bit lane offset)
  constexpr int MIX = MIXB % 1000;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
  auto ld = [&](unsigned long off, char* lds) {
    if constexpr (BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds, 16, (int)(unsigned)off, 0, 0, 0);
    else __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off), (lds_ptr_t)lds, 16, 0, 0);
  };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31, wm = w >> 1, wn = w & 1;
  for (int i = tid; i < 65536 / 16; i += 256) ((uint4*)smem)[i] = ((const uint4*)src)[(blockIdx.x * 4096 + i) & ((src_bytes >> 4) - 1)];
  __syncthreads();
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 af[4][2], bf[4][2];
  auto rd = [&](int buf, int j4) {
    const int kc = j4 * 2 + hi;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + l31, rb = wn * 64 + i * 32 + l31;
      af[j4][i] = *(const f32x4*)(smem + buf * 16384 + ra * 128 + ((kc ^ ((ra >> 1) & 7)) << 4));
      bf[j4][i] = *(const f32x4*)(smem + (2 + buf) * 16384 + rb * 128 + ((kc ^ ((rb >> 1) & 7)) << 4));
    }
  };
  if (MIX == 0) for (int j4 = 0; j4 < 4; ++j4) rd(0, j4);
  unsigned long goff = ((unsigned long)(blockIdx.x * 4 + w) * 65536 + lane * 16) & (src_bytes - 1);
  if constexpr (MIX == 3) {
    // the barrier moved INSIDE the stage: fragments of the next k-group are in registers before the MFMAs of the current one issue, the stage's
    // last group runs after the barrier on fragments read before it, and the next stage's first group is read right after the barrier
    auto dma = [&](int buf) {
#pragma unroll
      for (int d = 0; d < 8; ++d)
        ld((goff + d * 1024) & (src_bytes - 1), smem + ((d & 1) * 2 + buf) * 16384 + (w * 32 + (d >> 1) * 8) * 128);
      goff = (goff + 8192) & (src_bytes - 1);
    };
    auto mm = [&](int j4) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j4][j][e], af[j4][i][e], acc[i][j], 0, 0, 0);
    };
    dma(1);
    rd(0, 0);
    for (int s = 0; s < stages; ++s) {
      const int buf = s & 1;
      rd(buf, 1); __builtin_amdgcn_sched_barrier(0); mm(0); __builtin_amdgcn_sched_barrier(0);
      rd(buf, 2); __builtin_amdgcn_sched_barrier(0); mm(1); __builtin_amdgcn_sched_barrier(0);
      rd(buf, 3); __builtin_amdgcn_sched_barrier(0); mm(2); __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      rd(buf ^ 1, 0); __builtin_amdgcn_sched_barrier(0);
      dma(buf); __builtin_amdgcn_sched_barrier(0);
      mm(3); __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (MIX == 6 || MIX == 7) {
    // the pieces SPREAD over the stage: one global_load_lds after every 8th MFMA (MIX 6) / after every 8th MFMA but never first in a k-group (MIX 7)
    auto piece = [&](int buf, int d) {
      ld((goff + d * 1024) & (src_bytes - 1), smem + ((d & 1) * 2 + buf) * 16384 + (w * 32 + (d >> 1) * 8) * 128);
    };
    for (int s = 0; s < stages; ++s) {
      const int buf = s & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        rd(buf, j4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j4][j][e], af[j4][i][e], acc[i][j], 0, 0, 0);
              if (MIX == 7 && (e & 1) && i == 0 && j == 0) { __builtin_amdgcn_sched_barrier(0); piece(buf ^ 1, j4 * 2 + (e >> 1)); __builtin_amdgcn_sched_barrier(0); }
            }
          if (MIX == 6 && (e & 1)) { __builtin_amdgcn_sched_barrier(0); piece(buf ^ 1, j4 * 2 + (e >> 1)); __builtin_amdgcn_sched_barrier(0); }
        }
      }
      goff = (goff + 8192) & (src_bytes - 1);
    }
  } else if constexpr (MIX >= 100) {
    // piece d right behind MFMA number P0 + d * STEP of the stage (MIX = 100 + 10 * STEP + P0)
    constexpr int STEP = (MIX - 100) / 10, P0 = (MIX - 100) % 10;
    auto piece = [&](int buf, int d) {
      ld((goff + d * 1024) & (src_bytes - 1), smem + ((d & 1) * 2 + buf) * 16384 + (w * 32 + (d >> 1) * 8) * 128);
    };
    for (int s = 0; s < stages; ++s) {
      const int buf = s & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        rd(buf, j4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j4][j][e], af[j4][i][e], acc[i][j], 0, 0, 0);
              const int m = j4 * 16 + e * 4 + i * 2 + j;
              if (m >= P0 && (m - P0) % STEP == 0 && (m - P0) / STEP < 8) { __builtin_amdgcn_sched_barrier(0); piece(buf ^ 1, (m - P0) / STEP); __builtin_amdgcn_sched_barrier(0); }
            }
      }
      goff = (goff + 8192) & (src_bytes - 1);
    }
  } else
  for (int s = 0; s < stages; ++s) {
    const int buf = s & 1;
    if (MIX == 2 || MIX == 4 || MIX == 5) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (MIX != 5) __syncthreads();
      if (MIX != 4)
#pragma unroll
      for (int d = 0; d < 8; ++d)
        ld((goff + d * 1024) & (src_bytes - 1), smem + ((d & 1) * 2 + (buf ^ 1)) * 16384 + (w * 32 + (d >> 1) * 8) * 128);
      goff = (goff + 8192) & (src_bytes - 1);
    }
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      if (MIX >= 1) rd(MIX == 5 ? 0 : buf, j4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j4][j][e], af[j4][i][e], acc[i][j], 0, 0, 0);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  out[(blockIdx.x * 256 + tid) & 65535] = sum;
}

template <int MIX>
void run(const char* name, int grid, float* out, const char* src, long src_bytes, double secs) {
  const int stages = 2000, lds = 65536;
  hipFuncSetAttribute((const void*)k<MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  k<MIX><<<grid, 256, lds>>>(out, src, src_bytes, 50);
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  double el = 0;
  while (el < secs) {
    for (int r = 0; r < 4; ++r) k<MIX><<<grid, 256, lds>>>(out, src, src_bytes, stages);
    hipDeviceSynchronize();
    n += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double tf = (double)grid * 2.0 * 128 * 128 * 32 * (double)stages * n / el / 1e12;   // a stage of a block: 128 x 128 x 32 MACs
  printf("PHASE %-72s %6.1f TF/s over %.1f s  (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", name, tf, el,
         el / ((double)stages * n * 64.0 * (grid / 256.0)) * 2.4e9);
  fflush(stdout);
  struct timespec ts = {1, 0};
  nanosleep(&ts, nullptr);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  const long src_bytes = 1L << 28, hot = 1L << 22;      // `hot`: the DMA source wraps inside 4 MiB (an L2 hit, as the convolution's 9-fold pixel reuse is)
  float* out; char *zsrc, *rsrc;
  hipMalloc(&out, 65536 * 4);
  hipMalloc(&zsrc, src_bytes);
  hipMalloc(&rsrc, src_bytes);
  hipMemset(zsrc, 0, src_bytes);
  {
    std::vector<float> h(16 << 20);
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) v = nd(g);
    for (long o = 0; o < src_bytes; o += (long)h.size() * 4) hipMemcpy(rsrc + o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  }
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("zeros,  MFMA only, 1 wave per SIMD", 256, out, zsrc, src_bytes, secs);
    run<0>("zeros,  MFMA only, 2 waves per SIMD", 512, out, zsrc, src_bytes, secs);
    run<0>("N(0,1), MFMA only, 1 wave per SIMD", 256, out, rsrc, src_bytes, secs);
    run<0>("N(0,1), MFMA only, 2 waves per SIMD", 512, out, rsrc, src_bytes, secs);
    run<1>("N(0,1), + 16 ds_read_b128 per 64 MFMAs, 2 waves per SIMD", 512, out, rsrc, src_bytes, secs);
    run<2>("N(0,1), + reads + 8 global_load_lds + vmcnt(0) + barrier per stage, 2 waves", 512, out, rsrc, src_bytes, secs);
    run<2>("zeros,  + reads + 8 global_load_lds + vmcnt(0) + barrier per stage, 2 waves", 512, out, zsrc, src_bytes, secs);
    run<3>("N(0,1), same work, barrier inside the stage (fragments prefetched across it), 2 waves", 512, out, rsrc, src_bytes, secs);
    run<1002>("N(0,1), L2, DESCRIPTOR addressing: reads + 8 buffer_load lds as a burst + vmcnt(0) + barrier, 2 waves", 512, out, rsrc, hot, secs);
    run<1180>("N(0,1), L2, DESCRIPTOR addressing: piece d behind MFMA 0 + 8 d, 2 waves", 512, out, rsrc, hot, secs);
    run<180>("N(0,1), L2: piece d behind MFMA 0 + 8 d, 2 waves", 512, out, rsrc, hot, secs);
    run<110>("N(0,1), L2: piece d behind MFMA 0 + 1 d, 2 waves", 512, out, rsrc, hot, secs);
    run<120>("N(0,1), L2: piece d behind MFMA 0 + 2 d, 2 waves", 512, out, rsrc, hot, secs);
    run<141>("N(0,1), L2: piece d behind MFMA 1 + 4 d, 2 waves", 512, out, rsrc, hot, secs);
    run<142>("N(0,1), L2: piece d behind MFMA 2 + 4 d, 2 waves", 512, out, rsrc, hot, secs);
    run<160>("N(0,1), L2: piece d behind MFMA 0 + 6 d, 2 waves", 512, out, rsrc, hot, secs);
    run<174>("N(0,1), L2: piece d behind MFMA 4 + 7 d, 2 waves", 512, out, rsrc, hot, secs);
    run<141>("N(0,1), L2: piece d behind MFMA 1 + 4 d, 1 wave per SIMD", 256, out, rsrc, hot, secs);
    run<6>("N(0,1), L2-resident source: the 8 pieces spread, one after every 8th MFMA, 2 waves", 512, out, rsrc, hot, secs);
    run<7>("N(0,1), L2-resident source: the 8 pieces spread, each right behind an MFMA of a k-group, 2 waves", 512, out, rsrc, hot, secs);
    run<6>("N(0,1), L2-resident source: the 8 pieces spread, one after every 8th MFMA, 1 wave per SIMD", 256, out, rsrc, hot, secs);
    run<4>("N(0,1), reads + barrier per stage, NO DMA, 2 waves", 512, out, rsrc, hot, secs);
    run<5>("N(0,1), L2-resident source: reads + 8 global_load_lds + vmcnt(0), NO barrier, 2 waves", 512, out, rsrc, hot, secs);
    run<2>("N(0,1), L2-resident source: reads + DMA + vmcnt(0) + barrier, 1 wave per SIMD", 256, out, rsrc, hot, secs);
    run<2>("N(0,1), L2-resident source: reads + 8 global_load_lds + vmcnt(0) + barrier per stage, 2 waves", 512, out, rsrc, hot, secs);
    run<3>("N(0,1), L2-resident source: barrier inside the stage, 2 waves", 512, out, rsrc, hot, secs);
  }
  return 0;
}
