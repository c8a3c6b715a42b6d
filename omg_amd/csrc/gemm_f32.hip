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
// the 16-bit kernels: chunk ^= (row >> 1) & 7, filled by LDS-DMA with the swizzle on the source address), double buffered.
// MFMA operand mapping: the instruction wants ONE float per lane (A[i = lane & 31][k = lane >> 5]); a lane reads a whole
// 16-byte chunk (4 consecutive k) with one ds_read_b128 — lane half h takes chunk 2 j + h — and issues four MFMAs, the e-th
// using element e of both operands: that MFMA contracts k = 8 j + e (h = 0) and 8 j + 4 + e (h = 1); over e and j every k of
// the stage is used exactly once.  The product order differs from a plain k = 0, 1, 2, ... loop; fp32 addition is not
// associative, so the result equals an fp32 reference to rounding (1e-6 relative), not bitwise.
// The accumulators hold the TRANSPOSED tile (mfma(W, A)): a lane owns one output row and runs of 4 consecutive channels = one
// 16-byte store.  64 MFMAs x 64 cycles per wave and stage against 16 ds_read_b128: MFMA-bound by a wide margin.
#include "common.h"

// out-of-image taps source 16 zero bytes from here (a device symbol cannot be shared across translation units without -fgpu-rdc)
__device__ __attribute__((aligned(256))) unsigned char omg_zero_page_f32[256];

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
typedef const __attribute__((address_space(1))) void* f32_gbl_ptr_t;

__global__ __launch_bounds__(256, 2) void conv_f32_kernel(F32P p) {
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

  const int prow = lane >> 3, ppos = lane & 7;
  int cb[4], cy[4], cx[4], wrow[4], sch[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = w * 32 + i * 8 + prow;
    sch[i] = (ppos ^ ((r >> 1) & 7)) * 16;
    int gm = m0 + r; if (gm > p.M - 1) gm = p.M - 1;
    int gn = n0 + r; if (gn > p.N - 1) gn = p.N - 1;
    wrow[i] = gn;
    const int hw = p.Hout * p.Wout;
    const int b = gm / hw; const int rem = gm - b * hw;
    cb[i] = b; cy[i] = rem / p.Wout; cx[i] = rem - cy[i] * p.Wout;
  }
  const int cpt = p.Cin / FBK;                  // stages per tap
  const int pad = p.ksize == 3 ? 1 : 0;
  const int Hl = p.upsample ? p.Hin * 2 : p.Hin, Wl = p.upsample ? p.Win * 2 : p.Win;
  const char* zero = (const char*)omg_zero_page_f32;
  const int nk = p.K / FBK;

  auto issue = [&](int kt, int buf) {
    const int tap = kt / cpt, c0 = (kt - tap * cpt) * FBK;
    const int dy = tap / p.ksize - pad, dx = tap - (tap / p.ksize) * p.ksize - pad;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = cy[i] + dy, ix = cx[i] + dx;
      const bool ok = (iy >= 0) && (iy < Hl) && (ix >= 0) && (ix < Wl);
      if (p.upsample) { iy >>= 1; ix >>= 1; }
      const long pix = ((long)cb[i] * p.Hin + iy) * p.Win + ix;
      const char* asrc = ok ? p.A + (pix * p.Cin + c0) * 4 + sch[i] : zero;
      __builtin_amdgcn_global_load_lds((f32_gbl_ptr_t)asrc, (f32_lds_ptr_t)(smem + buf * FTILE + (w * 32 + i * 8) * 128), 16, 0, 0);
      const char* wsrc = p.W + ((long)wrow[i] * p.K + (long)kt * FBK) * 4 + sch[i];
      __builtin_amdgcn_global_load_lds((f32_gbl_ptr_t)wsrc, (f32_lds_ptr_t)(smem + (2 + buf) * FTILE + (w * 32 + i * 8) * 128), 16, 0, 0);
    }
  };

  const int wm = w >> 1, wn = w & 1;
  f32x16 acc[2][2];                              // acc[i][j]: output rows wm*64 + 32 i + l31, channels wn*64 + 32 j + ...
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
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
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);
    }
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
  static bool attr = false;
  if (!attr) { attr = true; (void)hipFuncSetAttribute((const void*)conv_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * FTILE); }
  OMG_LAUNCH(conv_f32_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), 4 * FTILE, (hipStream_t)stream, p);
  return omg_check_launch("conv2d_f32");
}
