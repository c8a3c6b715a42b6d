// tools/exp/conv_out_v2.h — EXPERIMENT for round 5 (built by `make -C omg_amd/csrc EXP=1` only; never part of the product library).
// WRITTEN IN ROUND 4 WITH NO GPU TIME LEFT: compiled for gfx950, NOT RUN.
//
// conv_out (misc.hip: the UNet's last convolution, 320 -> 4 channels, fp32 NCHW output) takes 1.86 ms per call at the benchmark's batch
// (64 x 128 x 128 x 320: 0.45 % of the step, 50 calls) for 671 MB of input — 14x its memory time.  One wave per output pixel, and every
// pixel re-reads the lane's slice of ALL the weights (4 x 2880 x 2 B = 23 KB per pixel and wave through the vector L1: 24 GB per launch).
// conv_out_kernel2 keeps that slice in registers: iteration n of a lane handles i = lane + 64 n of the 9 * Cin / 8 (tap, 8-channel vector)
// pairs, which does not depend on the pixel, so the 4 x NIT 16-byte weight vectors are loaded once per wave (96 VGPRs at Cin = 320) and a wave
// walks 256 pixels instead of 32.  Per pixel the same loads of the input, the same products added to the same accumulators in the same order,
// the same cross-lane reduction: torch.equal with conv_out_kernel (tests/test_kernels_gpu.py).  16-bit storage, Cout <= 4, 9 * Cin / 8 <= 64 * NIT.
// Included inside misc.hip.
template <typename T, int NIT>
__global__ __launch_bounds__(256) void conv_out_kernel2(const char* X, int B, int H, int W, int Cin, const char* Wt,
                                                         const T* bias, int Cout, float* Y) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long npix = (long)B * H * W;
  const int nvec = Cin / 8;
  const int K = 9 * Cin;
  u32x4 wreg[NIT][4];
  int dy[NIT], dx[NIT], cb[NIT];
  bool live[NIT];
#pragma unroll
  for (int n = 0; n < NIT; ++n) {
    const int i = lane + 64 * n;
    live[n] = i < 9 * nvec;
    const int tap = live[n] ? i / nvec : 0, vec = live[n] ? i - tap * nvec : 0;
    dy[n] = tap / 3 - 1; dx[n] = tap % 3 - 1; cb[n] = vec * 8 * (int)sizeof(T);
#pragma unroll
    for (int co = 0; co < 4; ++co)
      wreg[n][co] = (live[n] && co < Cout) ? *(const u32x4*)(Wt + ((long)co * K + tap * Cin + vec * 8) * (long)sizeof(T)) : u32x4{0u, 0u, 0u, 0u};
  }
  for (long pix = (long)blockIdx.x * 4 + w; pix < npix; pix += (long)gridDim.x * 4) {
    const int b = (int)(pix / (H * W)); const int rem = (int)(pix - (long)b * H * W);
    const int y = rem / W, x = rem - y * W;
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = 0.f;
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
      const int iy = y + dy[n], ix = x + dx[n];
      if (!live[n] || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      float f[8];
      load8<T>(X + (((long)b * H + iy) * W + ix) * Cin * (long)sizeof(T) + cb[n], f);
#pragma unroll
      for (int co = 0; co < 4; ++co) {
        if (co < Cout) {
          float wv[8];
          unpack8<T>(wreg[n][co], wv);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[co] += f[e] * wv[e];
        }
      }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc[co] += __shfl_xor(acc[co], off);
    }
    if (lane < Cout) {
      float v = 0.f;
#pragma unroll
      for (int co = 0; co < 4; ++co) if (co == lane) v = acc[co];
      Y[(((long)b * Cout + lane) * H + y) * W + x] = v + (bias ? (float)bias[lane] : 0.f);
    }
  }
}
