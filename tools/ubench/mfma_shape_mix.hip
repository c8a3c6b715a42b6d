// Micro-benchmark (tools only; round 6): is the 16 x 16 x 32 MFMA more frugal than the 32 x 32 x 16 one for THIS kernel's K-loop
// instruction mix — same tile (256 x 256 x 64 on four waves of 128 x 128, 256 accumulators), same LDS image and XOR swizzle, same 32 ds_read_b128
// and 16 LDS-DMA instructions and one barrier per 64-wide stage, random N(0, 1) operands so that the data paths toggle as in the benchmark — when
// the part sits at its power / current limit?  DESIGN.md §8 item 1 named "the library's 16x16x32 loop is ~9 % more frugal per FLOP" as the one GEMM
// experiment left; this prices the MFMA shape alone before a K loop, an accumulator layout and four epilogue forms are rewritten around it.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shape_mix.hip -o tools/ubench/mfma_shape_mix
//   tools/ubench/mfma_shape_mix [seconds per mode]        (tools/mfma_shape_probe.py runs it with rocm-smi polled beside it)
// Results are NOT checked: there is no product to be right about, only rates; every mode executes the same FLOPs per stage.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// SHAPE 32: 4 x 4 accumulator tiles of 32 x 32, per 16-wide k-step 4 A + 4 W fragments -> 16 MFMAs; 4 k-steps per stage
// SHAPE 16: 8 x 8 accumulator tiles of 16 x 16, per 32-wide k-step 8 A + 8 W fragments -> 64 MFMAs; 2 k-steps per stage
// MIX 0: MFMAs only (fragments read once)    MIX 1: the stage's 32 fragment reads feed its MFMAs    MIX 2: + 16 LDS-DMA of 1 KiB per wave + 1 barrier
template <int SHAPE, int MIX>
__global__ __launch_bounds__(256, 1) void k(float* out, const char* src, long src_bytes, int stages) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  // stage image: A [256 rows][128 B] at 0, W [256 rows][128 B] at 32 KB; a second image at 64 KB; the DMA lands in the third region (128 KB ..)
  for (int i = tid; i < 131072 / 16; i += 256) ((uint4*)smem)[i] = ((const uint4*)src)[(blockIdx.x * 8192 + i) & ((src_bytes >> 4) - 1)];
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(src_bytes < 0x7fffff00 ? src_bytes : 0x7fffff00), 0x00020000);
  unsigned goff = (unsigned)((blockIdx.x * 4 + w) * 65536 + lane * 16);
  float sum = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int hi = lane >> 5, l31 = lane & 31;
    f16x8 fa[4], fw[4];
    auto rd = [&](int buf, int ks) {
      const int sw = ((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *(const f16x8*)(smem + buf * 65536 + (wm * 128 + i * 32 + l31) * 128 + sw);
        fw[i] = *(const f16x8*)(smem + buf * 65536 + 32768 + (wn * 128 + i * 32 + l31) * 128 + sw);
      }
    };
    if (MIX == 0) rd(0, 0);
    for (int s = 0; s < stages; ++s) {
      const int buf = s & 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (MIX >= 1) rd(buf, ks);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fa[i], acc[i][j], 0, 0, 0);
        if (MIX == 2) {
#pragma unroll
          for (int d = 0; d < 4; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + 131072 + w * 1024 + ((ks * 4 + d) & 7) * 4096), 16, goff, (ks * 4 + d) * 1024, 0, 0);
        }
      }
      if (MIX == 2) {
        goff = (goff + 16384u) & 0x3fffffffu;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  } else {
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, l15 = lane & 15;
    f16x8 fa[8], fw[8];
    auto rd = [&](int buf, int ks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = i * 16 + l15;
        const int sw = ((ks * 4 + g) ^ ((row >> 1) & 7)) << 4;
        fa[i] = *(const f16x8*)(smem + buf * 65536 + (wm * 128 + row) * 128 + sw);
        fw[i] = *(const f16x8*)(smem + buf * 65536 + 32768 + (wn * 128 + row) * 128 + sw);
      }
    };
    if (MIX == 0) rd(0, 0);
    for (int s = 0; s < stages; ++s) {
      const int buf = s & 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (MIX >= 1) rd(buf, ks);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j], fa[i], acc[i][j], 0, 0, 0);
        if (MIX == 2) {
#pragma unroll
          for (int d = 0; d < 8; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + 131072 + w * 1024 + ((ks * 8 + d) & 7) * 4096), 16, goff, (ks * 8 + d) * 1024, 0, 0);
        }
      }
      if (MIX == 2) {
        goff = (goff + 16384u) & 0x3fffffffu;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  }
  out[blockIdx.x * 256 + tid] = sum;
}

template <int SHAPE, int MIX>
void run(const char* name, float* out, const char* src, long src_bytes, double secs) {
  const int stages = 4000, grid = 256, lds = 163840;
  hipFuncSetAttribute((const void*)k<SHAPE, MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  k<SHAPE, MIX><<<grid, 256, lds>>>(out, src, src_bytes, 50);
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  double el = 0;
  while (el < secs) {
    for (int r = 0; r < 4; ++r) k<SHAPE, MIX><<<grid, 256, lds>>>(out, src, src_bytes, stages);
    hipDeviceSynchronize();
    n += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  // per stage and CU: 256 x 256 x 64 MACs
  const double tf = 256.0 * 2.0 * 256 * 256 * 64 * (double)stages * n / el / 1e12;
  printf("PHASE %-58s %7.0f TF/s sustained over %.1f s  (%.1f cycles per 32x32x16-equivalent MFMA slot at 2.4 GHz)\n", name, tf, el,
         el / ((double)stages * n * 64.0) * 2.4e9);
  fflush(stdout);
  struct timespec ts = {1, 0};
  nanosleep(&ts, nullptr);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  const long src_bytes = 1L << 30;
  float* out; char* src;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&src, src_bytes);
  {   // N(0, 1) fp16 operands, 64 MB of them tiled over the buffer
    std::vector<_Float16> h(32 << 20);
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) v = (_Float16)nd(g);
    for (long o = 0; o < src_bytes; o += (long)h.size() * 2) hipMemcpy(src + o, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  }
  for (int rep = 0; rep < 2; ++rep) {
    run<32, 0>("32x32x16 MFMA only", out, src, src_bytes, secs);
    run<16, 0>("16x16x32 MFMA only", out, src, src_bytes, secs);
    run<32, 1>("32x32x16 + 32 fragment reads per stage", out, src, src_bytes, secs);
    run<16, 1>("16x16x32 + 32 fragment reads per stage", out, src, src_bytes, secs);
    run<32, 2>("32x32x16 + reads + 16 LDS-DMA + barrier per stage", out, src, src_bytes, secs);
    run<16, 2>("16x16x32 + reads + 16 LDS-DMA + barrier per stage", out, src, src_bytes, secs);
  }
  return 0;
}
