// norm.hip — GroupNorm (NHWC, optional fused SiLU, optional fused channel concat) and
// LayerNorm for gfx950.  Both are HBM-bound streaming kernels: 16-byte loads per lane,
// fp32 statistics, deterministic reduction order (no float atomics), so the same
// input gives bit-identical output on every run.
//
// GroupNorm is three launches: `gn_stats` writes per-(sample, pixel-chunk, group) partial
// (sum, sumsq); `gn_finalize` folds the partials in a fixed order (in fp64) into
// mean/rstd; `gn_apply` streams y = silu?(x*scale + shift).  The second read of x is served
// by L2 / Infinity Cache for every tensor on the SDXL path (<= 42 MB).
#include "common.h"

namespace {

constexpr int GN_MAX_CHUNKS = 1024;    // partial-sum chunks per sample (64 until round 2: a 1024x1024x128 VAE map then ran on 128 workgroups)

struct GnP {
  const char* X1; int C1; const char* X2; int C2;
  int B, HW, G, cpg, nvec, tpp, vpt, pr, nchunk, ppc, b0;
  float eps; const char* gamma; const char* beta; int silu;
  float* ws; char* Y;
  char* Q; unsigned char* S; long P;       // MX8 output: e4m3 bytes [B*HW][Cq] + scale bytes [(Cq/128)][P][4], P = B*HW
  int Cq;                                  // C rounded up to a multiple of 128: the pad channels are written as zeros (scale byte 0)
};

template <typename T>
OMG_DEV void gn_load(const GnP& p, int b, int pix, int vec, float (&f)[8]) {
  const int c = vec * 8;
  const char* src = (c < p.C1) ? p.X1 + (((long)b * p.HW + pix) * p.C1 + c) * (long)sizeof(T)
                               : p.X2 + (((long)b * p.HW + pix) * p.C2 + (c - p.C1)) * (long)sizeof(T);
  load8<T>(src, f);
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(GnP p) {
  extern __shared__ float lds[];            // [pr][C] sums, then [pr][C] sumsq
  const int C = p.C1 + p.C2;
  const int tid = threadIdx.x;
  const int prow = tid / p.tpp, tv = tid - prow * p.tpp;
  const int chunk = blockIdx.x, b = p.b0 + blockIdx.y;
  const int pix0 = chunk * p.ppc;
  const int pix1 = min(p.HW, pix0 + p.ppc);
  float s[2][8], q[2][8];
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[v][e] = 0.f; q[v][e] = 0.f; }
  if (prow < p.pr) {
    // four pixels per trip: four independent 16-byte loads in flight per lane (one load per trip left the kernel latency-bound
    // at 0.8-2 TB/s); the accumulation order over pixels is unchanged
    for (int pix = pix0 + prow; pix < pix1; pix += 4 * p.pr) {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int vec = tv + v * p.tpp;
        if (v < p.vpt && vec < p.nvec) {
          float f[4][8];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int px = pix + u * p.pr;
            if (px < pix1) gn_load<T>(p, b, px, vec, f[u]);
            else {
#pragma unroll
              for (int e = 0; e < 8; ++e) f[u][e] = 0.f;
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[v][e] += f[u][e]; q[v][e] += f[u][e] * f[u][e]; }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int vec = tv + v * p.tpp;
      if (v < p.vpt && vec < p.nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          lds[prow * C + vec * 8 + e] = s[v][e];
          lds[(p.pr + prow) * C + vec * 8 + e] = q[v][e];
        }
      }
    }
  }
  __syncthreads();
  if (tid < p.G) {
    float ss = 0.f, qq = 0.f;
    for (int r = 0; r < p.pr; ++r)
      for (int c = tid * p.cpg; c < (tid + 1) * p.cpg; ++c) { ss += lds[r * C + c]; qq += lds[(p.pr + r) * C + c]; }
    float* out = p.ws + (((long)b * GN_MAX_CHUNKS + chunk) * p.G + tid) * 2;
    out[0] = ss; out[1] = qq;
  }
}

// One wave per (group, sample): folds the chunk partials in a FIXED order (lane l takes chunks l, l + 64, ...; then a butterfly
// whose pairing does not depend on the data) in fp64 and leaves (mean, rstd) behind the partials in the workspace.  Until round 2
// every block of gn_apply did this fold itself, 32 threads walking all the partials one dependent load after the other: with up to
// 1024 chunks per sample that serial prologue was 80 % of the apply pass on the UNet's largest maps (4.6 ms for 4 GB of traffic).
__global__ __launch_bounds__(64) void gn_finalize_kernel(GnP p) {
  const int g = blockIdx.x, b = p.b0 + blockIdx.y;
  const int lane = threadIdx.x;
  double ss = 0.0, qq = 0.0;
  for (int ch = lane; ch < p.nchunk; ch += 64) {
    const float* in = p.ws + (((long)b * GN_MAX_CHUNKS + ch) * p.G + g) * 2;
    ss += (double)in[0]; qq += (double)in[1];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    ss += __shfl_xor(ss, o);
    qq += __shfl_xor(qq, o);
  }
  if (lane == 0) {
    const double n = (double)p.HW * p.cpg;
    const double mean = ss / n;
    double var = qq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    float* mr = p.ws + (long)p.B * GN_MAX_CHUNKS * p.G * 2 + ((long)b * p.G + g) * 2;
    mr[0] = (float)mean;
    mr[1] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
}

// MX8 = true: the consumer is the MX-fp8 convolution (gemm_mx8.hip).  y is rounded to T exactly as the 16-bit kernel stores it,
// then quantised per 32 consecutive channels (4 neighbouring lanes: two DPP exchanges for the amax) with the rule of
// quant_mx8_kernel; the scale bytes go to S[c / 128][pixel][(c / 32) % 4] — the conv kernel fetches one dword per (output row,
// tap, 128-channel stage) with a per-lane LDS-DMA.
template <typename T, bool MX8 = false>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnP p) {
  __shared__ float mean_s[64], rstd_s[64];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = p.b0 + blockIdx.y;
  const int C = p.C1 + p.C2;
  if (tid < p.G) {      // mean / rstd of the sample's groups: folded once by gn_finalize_kernel
    const float* mr = p.ws + (long)p.B * GN_MAX_CHUNKS * p.G * 2 + ((long)b * p.G + tid) * 2;
    mean_s[tid] = mr[0];
    rstd_s[tid] = mr[1];
  }
  __syncthreads();
  const int prow = tid / p.tpp, tv = tid - prow * p.tpp;
  if (prow >= p.pr) return;
  float sc[2][8], sh[2][8];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int vec = tv + v * p.tpp;
    if (v < p.vpt && vec < p.nvec) {
      float ga[8], be[8];
      load8<T>(p.gamma + (long)vec * 8 * sizeof(T), ga);
      load8<T>(p.beta + (long)vec * 8 * sizeof(T), be);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (vec * 8 + e) / p.cpg;
        const float a = rstd_s[g] * ga[e];
        sc[v][e] = a; sh[v][e] = be[e] - mean_s[g] * a;
      }
    }
  }
  const int pix0 = chunk * p.ppc;
  const int pix1 = min(p.HW, pix0 + p.ppc);
  for (int pix = pix0 + prow; pix < pix1; pix += 4 * p.pr) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int vec = tv + v * p.tpp;
      if (v < p.vpt && vec < p.nvec) {
        float f[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int px = pix + u * p.pr;
          if (px < pix1) gn_load<T>(p, b, px, vec, f[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int px = pix + u * p.pr;
          if (px < pix1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float y = f[u][e] * sc[v][e] + sh[v][e];
              if (p.silu) y = silu_fast(y);
              f[u][e] = y;
            }
            if constexpr (!MX8) store8<T>(p.Y + (((long)b * p.HW + px) * C + vec * 8) * (long)sizeof(T), f[u]);
          }
        }
        if constexpr (MX8) {
          // all four lanes of a 32-channel block share px (same prow) and take the same branches: the exchanges are safe
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int px = pix + u * p.pr;
            const bool ok = px < pix1;
            float r[8];
            {
              const u32x4 pk = pack8<T>(f[u]);      // the value the 16-bit path would have stored
              unpack8<T>(pk, r);
            }
            float amax = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = __builtin_fmaxf(amax, __builtin_fabsf(r[e]));
            if (!ok) amax = 0.f;
            amax = __builtin_fmaxf(amax, __shfl_xor(amax, 1));
            amax = __builtin_fmaxf(amax, __shfl_xor(amax, 2));
            const unsigned be = mx8_scale_exp(amax);
            const float inv = mx8_inv_scale(be);
            const int sc_mode = (C & 127) == 0 ? 2 : (C & 63) == 0 ? 1 : 0;
            unsigned sc4 = be;                                   // every lane of the wave takes part in the exchanges
            if (sc_mode >= 1) sc4 |= __shfl_down(be, 4) << 8;
            if (sc_mode == 2) { sc4 |= __shfl_down(be, 8) << 16; sc4 |= __shfl_down(be, 12) << 24; }
            if (ok) {
              const long gp = (long)b * p.HW + px;
              u32x2 o = {mx8_pack4(r[0] * inv, r[1] * inv, r[2] * inv, r[3] * inv), mx8_pack4(r[4] * inv, r[5] * inv, r[6] * inv, r[7] * inv)};
              *(u32x2*)(p.Q + gp * p.Cq + vec * 8) = o;
              // scale bytes: one store per 128 channels (C % 128 == 0: the four block exponents sit in lanes vec, +4, +8, +12 of the
              // same pixel and wave), per 64 channels (C % 64 == 0), else per block — single-byte stores cost a transaction each
              if (sc_mode == 2) { if ((vec & 15) == 0) *(unsigned*)(p.S + (((long)(vec >> 4) * p.P + gp) << 2)) = sc4; }
              else if (sc_mode == 1) { if ((vec & 7) == 0) *(unsigned short*)(p.S + (((long)(vec >> 4) * p.P + gp) << 2) + ((vec >> 2) & 3)) = (unsigned short)sc4; }
              else if ((vec & 3) == 0) p.S[(((long)(vec >> 4) * p.P + gp) << 2) + ((vec >> 2) & 3)] = (unsigned char)be;
              const int cpad = C + vec * 8;      // the first (Cq - C) / 8 lanes of the pixel also clear its pad channels
              if (cpad < p.Cq) {
                *(u32x2*)(p.Q + gp * p.Cq + cpad) = u32x2{0u, 0u};
                if ((cpad & 31) == 0) p.S[(((long)(cpad >> 7) * p.P + gp) << 2) + ((cpad >> 5) & 3)] = 0;
              }
            }
          }
        }
      }
    }
  }
}

// LayerNorm: one wave per row, row held in registers, exact two-pass variance.
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_kernel(const char* X, long ldx, int M, int C, float eps, const char* gamma,
                                                 const char* beta, char* Y, long ldy) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= M) return;
  const int nvec = C >> 3;
  float x[NV][8];
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int vec = lane + v * 64;
    if (vec < nvec) {
      unpack8<T>(*(const u32x4*)(X + ((long)row * ldx + vec * 8) * 2), x[v]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[v][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[v][e] = 0.f;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int vec = lane + v * 64;
    if (vec < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = x[v][e] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int vec = lane + v * 64;
    if (vec < nvec) {
      float ga[8], be[8], y[8];
      unpack8<T>(*(const u32x4*)(gamma + (long)vec * 16), ga);
      unpack8<T>(*(const u32x4*)(beta + (long)vec * 16), be);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (x[v][e] - mean) * rstd * ga[e] + be[e];
      *(u32x4*)(Y + ((long)row * ldy + vec * 8) * 2) = pack8<T>(y);
    }
  }
}

// LayerNorm whose output goes straight to an MX-fp8 GEMM (omg_gemm_mx8): the normalised row never exists in 16 bits.  Same
// statistics and affine arithmetic as ln_kernel; a lane's 8 outputs are a quarter of a 32-block (amax over 4 adjacent lanes),
// 16 adjacent lanes are one 128-wide stage = one scale dword S[stage][row].  C % 128 == 0.
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_mx8_kernel(const char* X, long ldx, int M, int C, float eps, const char* gamma,
                                                     const char* beta, char* Q, long ldq, unsigned* S, int s_ld) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= M) return;
  const int nvec = C >> 3;
  float x[NV][8];
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int vec = lane + v * 64;
    if (vec < nvec) {
      unpack8<T>(*(const u32x4*)(X + ((long)row * ldx + vec * 8) * 2), x[v]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[v][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[v][e] = 0.f;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int vec = lane + v * 64;
    if (vec < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = x[v][e] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int vec = lane + v * 64;             // nvec is a multiple of 16: a group of 16 lanes is valid or invalid as a whole
    float y[8];
    float amax = 0.f;
    if (vec < nvec) {
      float ga[8], be[8];
      unpack8<T>(*(const u32x4*)(gamma + (long)vec * 16), ga);
      unpack8<T>(*(const u32x4*)(beta + (long)vec * 16), be);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // the value the 16-bit LayerNorm would have stored: the quantiser then sees exactly what omg_quant_mx8 would see
        y[e] = (float)(T)((x[v][e] - mean) * rstd * ga[e] + be[e]);
        amax = __builtin_fmaxf(amax, __builtin_fabsf(y[e]));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = 0.f;
    }
    amax = __builtin_fmaxf(amax, __shfl_xor(amax, 1));
    amax = __builtin_fmaxf(amax, __shfl_xor(amax, 2));
    const unsigned be_ = mx8_scale_exp(amax);
    const float inv = mx8_inv_scale(be_);
    unsigned sc = be_;
    sc |= __shfl_down(be_, 4) << 8;
    sc |= __shfl_down(be_, 8) << 16;
    sc |= __shfl_down(be_, 12) << 24;
    if (vec < nvec) {
      *(u32x2*)(Q + (long)row * ldq + vec * 8) = u32x2{mx8_pack4(y[0] * inv, y[1] * inv, y[2] * inv, y[3] * inv),
                                                        mx8_pack4(y[4] * inv, y[5] * inv, y[6] * inv, y[7] * inv)};
      if ((lane & 15) == 0) S[(long)(vec >> 4) * s_ld + row] = sc;
    }
  }
}

}  // namespace

extern "C" int omg_layernorm_mx8(int dtype, const void* X, int64_t ldx, int M, int C, float eps, const void* gamma, const void* beta,
                                 void* Q, int64_t ldq, void* scales, int s_ld, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_layernorm_mx8: dtype");
  OMG_REQUIRE(X && gamma && beta && Q && scales, "omg_layernorm_mx8: null operand");
  OMG_REQUIRE(C % 128 == 0 && C <= 2048 && ldx % 8 == 0 && ldq % 16 == 0 && s_ld >= M && s_ld % 4 == 0, "omg_layernorm_mx8: C % 128, C <= 2048, ldx % 8, ldq % 16, s_ld");
  if (M == 0) return OMG_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((M + 3) / 4);
  const int nv = (C / 8 + 63) / 64;
#define LNQ_LAUNCH(TT, NVV) OMG_LAUNCH((ln_mx8_kernel<TT, NVV>), grid, dim3(256), 0, s, (const char*)X, (long)ldx, M, C, eps, (const char*)gamma, (const char*)beta, (char*)Q, (long)ldq, (unsigned*)scales, s_ld)
  if (dtype == OMG_F16) {
    switch (nv) { case 1: LNQ_LAUNCH(f16, 1); break; case 2: LNQ_LAUNCH(f16, 2); break; case 3: LNQ_LAUNCH(f16, 3); break; default: LNQ_LAUNCH(f16, 4); }
  } else {
    switch (nv) { case 1: LNQ_LAUNCH(bf16, 1); break; case 2: LNQ_LAUNCH(bf16, 2); break; case 3: LNQ_LAUNCH(bf16, 3); break; default: LNQ_LAUNCH(bf16, 4); }
  }
#undef LNQ_LAUNCH
  return omg_check_launch("layernorm_mx8");
}

extern "C" int64_t omg_groupnorm_ws_floats(int B, int groups, int HW) {
  (void)HW;
  return (int64_t)B * GN_MAX_CHUNKS * groups * 2 + (int64_t)B * groups * 2;      // chunk partials + (mean, rstd) per (sample, group)
}

namespace {
int gn_run(int dtype, const void* X1, int C1, const void* X2, int C2, int B, int HW, int groups,
           float eps, const void* gamma, const void* beta, int silu, float* workspace, void* Y, void* Q, void* S,
           void* stream) {
  const bool mx8 = Q != nullptr;
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16 || (dtype == OMG_F32 && !mx8), "omg_groupnorm: dtype");
  OMG_REQUIRE(X1 && gamma && beta && workspace && (Y || mx8) && (C2 == 0 || X2), "omg_groupnorm: null operand");
  if (mx8) OMG_REQUIRE(S != nullptr && (C1 + C2) % 32 == 0, "omg_groupnorm_mx8: scales, C % 32 == 0");
  const int C = C1 + C2;
  OMG_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && C % groups == 0 && groups <= 64 && groups > 0, "omg_groupnorm: channels/groups");
  OMG_REQUIRE(C / 8 <= 512, "omg_groupnorm: C <= 4096");
  if (B == 0 || HW == 0) return OMG_OK;
  GnP p{};
  p.X1 = (const char*)X1; p.C1 = C1; p.X2 = (const char*)X2; p.C2 = C2;
  p.B = B; p.HW = HW; p.G = groups; p.cpg = C / groups; p.nvec = C / 8;
  p.vpt = p.nvec <= 256 ? 1 : 2;
  p.tpp = p.vpt == 1 ? p.nvec : (p.nvec + 1) / 2;
  p.pr = 256 / p.tpp;
  if (mx8) {   // the 32-channel amax and the scale-byte gathers exchange between NEIGHBOURING lanes of one pixel: a pixel's lanes must
               // start on a multiple of the exchange width (always true for one vector per lane; C > 2048 needs C % 256 / 128 / 64 == 0)
    const int need = (C & 127) == 0 ? 16 : (C & 63) == 0 ? 8 : 4;
    OMG_REQUIRE(p.tpp % need == 0, "omg_groupnorm_mx8: C > 2048 needs C/16 to be a multiple of the scale-exchange width");
  }
  long elems = (long)HW * C;
  int nchunk = (int)((elems + 32767) / 32768);
  if (nchunk > GN_MAX_CHUNKS) nchunk = GN_MAX_CHUNKS;
  if (nchunk > HW) nchunk = HW;
  if (nchunk < 1) nchunk = 1;
  p.ppc = (HW + nchunk - 1) / nchunk;
  nchunk = (HW + p.ppc - 1) / p.ppc;
  p.nchunk = nchunk;
  p.eps = eps; p.gamma = (const char*)gamma; p.beta = (const char*)beta; p.silu = silu;
  p.ws = workspace; p.Y = (char*)Y;
  p.Q = (char*)Q; p.S = (unsigned char*)S; p.P = (long)B * HW; p.Cq = (C + 127) / 128 * 128;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)2 * p.pr * C * sizeof(float);
  // one pair of launches for the whole batch.  Tried and rejected (round 2): statistics + apply on groups of samples sized to the
  // 256 MiB Infinity Cache so that the apply pass re-reads x on-die — 16 % SLOWER at B = 64 (smaller grids, 2 us per extra boundary).
  long sub = B;
  for (int b0 = 0; b0 < B; b0 += (int)sub) {
    p.b0 = b0;
    const int nb = B - b0 < (int)sub ? B - b0 : (int)sub;
    dim3 grid(nchunk, nb);
    if (dtype == OMG_F16) {
      OMG_LAUNCH(gn_stats_kernel<f16>, grid, dim3(256), lds, s, p);
      OMG_LAUNCH(gn_finalize_kernel, dim3(p.G, nb), dim3(64), 0, s, p);
      if (mx8) OMG_LAUNCH((gn_apply_kernel<f16, true>), grid, dim3(256), 0, s, p);
      else OMG_LAUNCH((gn_apply_kernel<f16, false>), grid, dim3(256), 0, s, p);
    } else if (dtype == OMG_BF16) {
      OMG_LAUNCH(gn_stats_kernel<bf16>, grid, dim3(256), lds, s, p);
      OMG_LAUNCH(gn_finalize_kernel, dim3(p.G, nb), dim3(64), 0, s, p);
      if (mx8) OMG_LAUNCH((gn_apply_kernel<bf16, true>), grid, dim3(256), 0, s, p);
      else OMG_LAUNCH((gn_apply_kernel<bf16, false>), grid, dim3(256), 0, s, p);
    } else {          // fp32 storage: the up-blocks of the upcast VAE decode (lora_pipeline.py:639-652)
      OMG_LAUNCH(gn_stats_kernel<float>, grid, dim3(256), lds, s, p);
      OMG_LAUNCH(gn_finalize_kernel, dim3(p.G, nb), dim3(64), 0, s, p);
      OMG_LAUNCH((gn_apply_kernel<float, false>), grid, dim3(256), 0, s, p);
    }
  }
  return omg_check_launch("groupnorm");
}
}  // namespace

extern "C" int omg_groupnorm(int dtype, const void* X1, int C1, const void* X2, int C2, int B, int HW, int groups,
                             float eps, const void* gamma, const void* beta, int silu, float* workspace, void* Y,
                             void* stream) {
  OMG_REQUIRE(Y != nullptr, "omg_groupnorm: null output");
  return gn_run(dtype, X1, C1, X2, C2, B, HW, groups, eps, gamma, beta, silu, workspace, Y, nullptr, nullptr, stream);
}

extern "C" int omg_groupnorm_mx8(int dtype, const void* X1, int C1, const void* X2, int C2, int B, int HW, int groups,
                                 float eps, const void* gamma, const void* beta, int silu, float* workspace, void* Q,
                                 void* scales, void* stream) {
  OMG_REQUIRE(Q != nullptr && scales != nullptr, "omg_groupnorm_mx8: null output");
  return gn_run(dtype, X1, C1, X2, C2, B, HW, groups, eps, gamma, beta, silu, workspace, nullptr, Q, scales, stream);
}

extern "C" int omg_layernorm(int dtype, const void* X, int64_t ldx, int M, int C, float eps, const void* gamma,
                             const void* beta, void* Y, int64_t ldy, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_layernorm: dtype");
  OMG_REQUIRE(X && gamma && beta && Y, "omg_layernorm: null operand");
  OMG_REQUIRE(C % 8 == 0 && C <= 2048 && ldx % 8 == 0 && ldy % 8 == 0, "omg_layernorm: C % 8, C <= 2048");
  if (M == 0) return OMG_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((M + 3) / 4);
  const int nv = (C / 8 + 63) / 64;
#define LN_LAUNCH(TT, NVV) OMG_LAUNCH((ln_kernel<TT, NVV>), grid, dim3(256), 0, s, (const char*)X, (long)ldx, M, C, eps, (const char*)gamma, (const char*)beta, (char*)Y, (long)ldy)
  if (dtype == OMG_F16) {
    switch (nv) { case 1: LN_LAUNCH(f16, 1); break; case 2: LN_LAUNCH(f16, 2); break; case 3: LN_LAUNCH(f16, 3); break; default: LN_LAUNCH(f16, 4); }
  } else {
    switch (nv) { case 1: LN_LAUNCH(bf16, 1); break; case 2: LN_LAUNCH(bf16, 2); break; case 3: LN_LAUNCH(bf16, 3); break; default: LN_LAUNCH(bf16, 4); }
  }
#undef LN_LAUNCH
  return omg_check_launch("layernorm");
}
