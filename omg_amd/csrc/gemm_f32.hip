// gemm_f32.hip — fp32 implicit-GEMM convolution on the f32-input MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// The reference decodes the VAE in fp32 ("it overflows in float16": upcast_vae, /root/reference src/pipelines/lora_pipeline.py
// :639-652; with torch 2's SDPA processor diffusers keeps post_quant_conv / conv_in / the mid block in fp16 and runs the UP BLOCKS,
// conv_norm_out and conv_out in fp32).  The up blocks are 3x3 convolutions at 128..1024 pixels with 512 / 256 / 128 channels:
// fp32 operands in HBM, exact fp32 products accumulated in fp32 — bit for bit an fmaf chain (cdna guide §3), at the fp32 vector
// rate of 157 TFLOP/s (1/16 of the bf16 MFMA rate).
//
//   Y[M = B*Hout*Wout, N = Cout] = X (im2col, K = taps x Cin) · W[N, K]^T + bias (+ residual)
//
// Tile 128 x 128 on four waves (64 x 64 each = 2 x 2 MFMA tiles), K stage = 32 floats (128-byte rows, the same LDS image as
// the 16-bit kernels: chunk ^= (row >> 1) & 7, filled by LDS-DMA with the swizzle on the source address), double buffered;
// out-of-image taps are lane offsets beyond the descriptor's range, which the buffer unit returns as zeros.
// MFMA operand mapping: the instruction wants ONE float per lane (A[i = lane & 31][k = lane >> 5]); a lane reads a whole
// 16-byte chunk (4 consecutive k) with one ds_read_b128 — lane half h takes chunk 2 j + h — and issues four MFMAs, the e-th
// using element e of both operands: that MFMA contracts k = 8 j + e (h = 0) and 8 j + 4 + e (h = 1); over e and j every k of
// the stage is used exactly once.  The product order differs from a plain k = 0, 1, 2, ... loop; fp32 addition is not
// associative, so the result equals an fp32 reference to rounding (1e-6 relative), not bitwise.
// The accumulators hold the TRANSPOSED tile (mfma(W, A)): a lane owns one output row and runs of 4 consecutive channels = one
// 16-byte store.  64 MFMAs x 64 cycles per wave and stage against 16 ds_read_b128: MFMA-bound by a wide margin — what kept rounds 2-5 at 124 TF/s
// (0.79 of the peak) was the ISSUE cost of the stage's eight LDS-DMA pieces (comment above the kernel); round 6: 141-143 TF/s = 0.90-0.91.
#include "common.h"

namespace {

constexpr int FBM = 128, FBN = 128, FBK = 32;
constexpr int FTILE = FBM * FBK * 4;           // 16 KiB per operand tile

struct F32P {
  int M, N, K;
  const char* A; const char* W;
  const char* bias; const char* residual; char* C;
  int Hin, Win, Cin, Hout, Wout, ksize, upsample;
  int tiles_m, tiles_n;
};

typedef __attribute__((address_space(3))) void* f32_lds_ptr_t;

// 16 bytes per lane, global -> LDS, through a buffer descriptor (a non-template wrapper: see dma16 in gemm_epilogue.h)
__device__ __forceinline__ void f32_dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (f32_lds_ptr_t)lds, 16, (int)voff, soff, 0, 0);
}

// an offset no descriptor of this file reaches (num_records <= F32_OOB): the buffer unit turns the load into zeros — the padding taps
constexpr unsigned F32_OOB = 0xfff00000u;

#ifndef OMG_F32_P0
#define OMG_F32_P0 0
#endif
#ifndef OMG_F32_STEP
#define OMG_F32_STEP 8
#endif

// The K loop's eight LDS-DMA pieces per wave and stage are NOT issued as a burst behind the barrier: piece d goes right behind MFMA number
// P0 + STEP d of the stage (tools/ubench/mfma_f32_rate.hip: a burst costs this instruction mix 70 cycles of MFMA issue per piece, the spread
// placement 39; the 64-cycle MFMA ahead of a piece covers part of its issue), and its address is a descriptor + a 32-bit lane offset that
// changes once per TAP + a scalar offset per stage — no per-stage 64-bit address arithmetic between the MFMAs.
template <int P0, int STEP>
__global__ __launch_bounds__(256, 2) void conv_f32_kernel(F32P p) {
  static_assert(P0 >= 0 && STEP >= 1 && P0 + 7 * STEP < 64, "eight pieces inside the stage's 64 MFMAs");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;       // N fastest: consecutive blocks of an XCD share the A panel
  const int m0 = tm * FBM, n0 = tn * FBN;

  // descriptors: W whole; X from the image of the tile's first row on (a tile reaches into the next image at most when Hout * Wout >= 128;
  // the launcher checks that every lane offset stays below F32_OOB), so that 32-bit offsets do for inputs beyond 4 GB
  const int hw = p.Hout * p.Wout;
  const int b0 = m0 / hw;
  const long img_bytes = (long)p.Hin * p.Win * p.Cin * 4;
  const long a_left = ((long)(p.M / hw) - b0) * img_bytes;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)b0 * img_bytes), 0,
                                                                      (int)(unsigned)(a_left < (long)F32_OOB ? a_left : (long)F32_OOB), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)((long)p.N * p.K * 4), 0x00020000);

  const int prow = lane >> 3, ppos = lane & 7;
  int cb[4], cy[4], cx[4], sch[4];
  unsigned voffW[4], voffA[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = w * 32 + i * 8 + prow;
    sch[i] = (ppos ^ ((r >> 1) & 7)) * 16;
    int gm = m0 + r; if (gm > p.M - 1) gm = p.M - 1;
    int gn = n0 + r; if (gn > p.N - 1) gn = p.N - 1;
    voffW[i] = (unsigned)gn * (unsigned)p.K * 4u + (unsigned)sch[i];
    const int b = gm / hw; const int rem = gm - b * hw;
    cb[i] = b - b0; cy[i] = rem / p.Wout; cx[i] = rem - cy[i] * p.Wout;
  }
  const int cpt = p.Cin / FBK;                  // stages per tap
  const int pad = p.ksize == 3 ? 1 : 0;
  const int Hl = p.upsample ? p.Hin * 2 : p.Hin, Wl = p.upsample ? p.Win * 2 : p.Win;
  const int nk = p.K / FBK;

  // the lane offsets of the four A pieces for one tap: once per tap and tile, not once per stage
  auto set_tap = [&](int tap) {
    const int ty = p.ksize == 3 ? (tap >= 6 ? 2 : tap >= 3 ? 1 : 0) : 0;
    const int dy = ty - pad, dx = tap - ty * p.ksize - pad;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = cy[i] + dy, ix = cx[i] + dx;
      const bool ok = ((unsigned)iy < (unsigned)Hl) & ((unsigned)ix < (unsigned)Wl);
      if (p.upsample) { iy >>= 1; ix >>= 1; }
      const unsigned pix = (unsigned)((cb[i] * p.Hin + iy) * p.Win + ix);
      voffA[i] = ok ? pix * (unsigned)p.Cin * 4u + (unsigned)sch[i] : F32_OOB;
    }
  };
  // piece d of a stage: row block d >> 1 of A (even d) / of W (odd d), 8 rows x 128 bytes per wave
  auto piece = [&](int nbuf, int d, int soffA, int soffW) {
    const int i = d >> 1;
    if (d & 1) f32_dma16(rsW, smem + (2 + nbuf) * FTILE + (w * 32 + i * 8) * 128, voffW[i], soffW);
    else f32_dma16(rsA, smem + nbuf * FTILE + (w * 32 + i * 8) * 128, voffA[i], soffA);
  };

  const int wm = w >> 1, wn = w & 1;
  f32x16 acc[2][2];                              // acc[i][j]: output rows wm*64 + 32 i + l31, channels wn*64 + 32 j + ...
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // (tap, c): the stage being ISSUED — one ahead of the stage being computed
  int tap = 0, c = 0;
  set_tap(0);
#pragma unroll
  for (int d = 0; d < 8; ++d) piece(0, d, 0, 0);
  if (++c == cpt) { c = 0; tap = 1; if (nk > cpt) set_tap(1); }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // stage kt + 1 into the other buffer; behind the last stage the pieces are still issued (their sources are in range or range-checked, their
    // buffer is never read): eight uniform branches per stage would cost more than one stage of traffic per tile
    const int soffA = c * (FBK * 4), soffW = (kt + 1) * (FBK * 4);
    const char* a_base = smem + buf * FTILE;
    const char* b_base = smem + (2 + buf) * FTILE;
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const int kc = j4 * 2 + hi;
      f32x4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31;
        af[i] = *(const f32x4*)(a_base + ra * 128 + ((kc ^ ((ra >> 1) & 7)) << 4));
        const int rb = wn * 64 + i * 32 + l31;
        bf[i] = *(const f32x4*)(b_base + rb * 128 + ((kc ^ ((rb >> 1) & 7)) << 4));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);
            const int m = j4 * 16 + e * 4 + i * 2 + j;
            if (m >= P0 && (m - P0) % STEP == 0 && (m - P0) / STEP < 8) {
              __builtin_amdgcn_sched_barrier(0);
              piece(buf ^ 1, (m - P0) / STEP, soffA, soffW);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
    }
    if (++c == cpt) { c = 0; ++tap; if (kt + 2 < nk) set_tap(tap); }
  }

  // ---- epilogue: lane (l31, hi) of acc[i][j] holds row m = wm0 + 32 i + l31, channels wn0 + 32 j + 8 g + 4 hi + {0..3} (g = r >> 2)
  const int wm0 = m0 + wm * 64, wn0 = n0 + wn * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int gm = wm0 + i * 32 + l31;
    if (gm >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = wn0 + j * 32 + g * 8 + hi * 4;
        if (c >= p.N) continue;                   // N % 4 == 0: a run of 4 is inside or outside as a whole
        f32x4 v = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
        if (p.bias) v += *(const f32x4*)(p.bias + (long)c * 4);
        if (p.residual) v += *(const f32x4*)(p.residual + ((long)gm * p.N + c) * 4);
        *(f32x4*)(p.C + ((long)gm * p.N + c) * 4) = v;
      }
  }
}

}  // namespace

extern "C" int omg_conv2d_f32(const omg_conv2d_f32_args* a, void* stream) {
  OMG_REQUIRE(a != nullptr, "omg_conv2d_f32: null args");
  OMG_REQUIRE(a->ksize == 1 || a->ksize == 3, "omg_conv2d_f32: ksize must be 1 or 3");
  OMG_REQUIRE(a->Cin > 0 && a->Cin % 32 == 0 && a->Cout % 4 == 0, "omg_conv2d_f32: Cin % 32, Cout % 4");
  OMG_REQUIRE(a->X && a->W && a->Y, "omg_conv2d_f32: null operand");
  const int Hl = a->upsample ? 2 * a->Hin : a->Hin, Wl = a->upsample ? 2 * a->Win : a->Win;
  OMG_REQUIRE(a->Hout == Hl && a->Wout == Wl, "omg_conv2d_f32: stride 1, 'same' padding only");
  F32P p{};
  p.M = a->B * a->Hout * a->Wout; p.N = a->Cout; p.K = a->ksize * a->ksize * a->Cin;
  if (p.M == 0) return OMG_OK;
  p.A = (const char*)a->X; p.W = (const char*)a->W; p.bias = (const char*)a->bias; p.residual = (const char*)a->residual; p.C = (char*)a->Y;
  p.Hin = a->Hin; p.Win = a->Win; p.Cin = a->Cin; p.Hout = a->Hout; p.Wout = a->Wout; p.ksize = a->ksize; p.upsample = a->upsample;
  p.tiles_m = (p.M + FBM - 1) / FBM; p.tiles_n = (p.N + FBN - 1) / FBN;
  {   // 32-bit lane offsets from the first image a tile touches (kernel header): a tile spans at most FBM / (Hout * Wout) + 2 images
    const long img_bytes = (long)a->Hin * a->Win * a->Cin * 4, span = FBM / ((long)a->Hout * a->Wout) + 2;
    OMG_REQUIRE(span * img_bytes < (long)F32_OOB && (long)p.N * p.K * 4 < 0x7fffffffL, "omg_conv2d_f32: an image / the weight beyond the 32-bit offsets of the kernel");
  }
  static bool attr = false;
  if (!attr) { attr = true; (void)hipFuncSetAttribute((const void*)conv_f32_kernel<OMG_F32_P0, OMG_F32_STEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * FTILE); }
  OMG_LAUNCH((conv_f32_kernel<OMG_F32_P0, OMG_F32_STEP>), dim3(p.tiles_m * p.tiles_n), dim3(256), 4 * FTILE, (hipStream_t)stream, p);
  return omg_check_launch("conv2d_f32");
}
