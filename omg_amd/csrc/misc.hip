// misc.hip — boundary convolutions (NCHW latents <-> NHWC features), the time/text
// conditioning elementwise pieces, and the fused region-fusion + CFG + scheduler step.
#include "common.h"
#include <string.h>

// per calling thread: entry points are re-entrant across host threads / streams (B5), so is their diagnostic
static thread_local char g_err[256] = "";
void omg_set_error(const char* msg) { strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
extern "C" const char* omg_last_error(void) { return g_err; }
extern "C" int omg_abi_version(void) { return OMG_ABI_VERSION; }

namespace {

// ---------------------------------------------------------------- conv_in
// conv_in (Cin = 4, 3x3) as im2col + the MFMA GEMM: patch[pix][(ky*3+kx)*Cin + ci], zero padded to KP (64)
// columns so that it is one BK slice of omg_gemm.  One thread per pixel: 36 coalesced NCHW reads
// (neighbouring threads = neighbouring x), 128 B written as 8 x 16 B.
template <typename T, typename TI>
__global__ __launch_bounds__(256) void im2col_in_kernel(const TI* X, int B, int Cin, int H, int W, int KP, char* out) {
  const long npix = (long)B * H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
    const int b = (int)(pix / (H * W)); const int rem = (int)(pix - (long)b * H * W);
    const int y = rem / W, x = rem - y * W;
    for (int c8 = 0; c8 < KP / 8; ++c8) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = c8 * 8 + e;
        const int tap = k / Cin, ci = k - tap * Cin;
        const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
        const bool ok = (tap < 9) && iy >= 0 && iy < H && ix >= 0 && ix < W;
        f[e] = ok ? (float)X[(((long)b * Cin + ci) * H + iy) * W + ix] : 0.f;
      }
      *(u32x4*)(out + (pix * KP + c8 * 8) * 2) = pack8<T>(f);
    }
  }
}

// --------------------------------------------------------------- conv_out
// out[b,co,y,x] = bias[co] + sum_{ky,kx,c} w[co][ky][kx][c] * in[b,y+ky-1,x+kx-1,c]
// one wave per output pixel; lanes stride over (tap, 8-channel vec); Cout <= 8.
template <typename T>
__global__ __launch_bounds__(256) void conv_out_kernel(const char* X, int B, int H, int W, int Cin, const char* Wt,
                                                        const T* bias, int Cout, float* Y) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long npix = (long)B * H * W;
  const int nvec = Cin / 8;
  const int K = 9 * Cin;
  for (long pix = (long)blockIdx.x * 4 + w; pix < npix; pix += (long)gridDim.x * 4) {
    const int b = (int)(pix / (H * W)); const int rem = (int)(pix - (long)b * H * W);
    const int y = rem / W, x = rem - y * W;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int i = lane; i < 9 * nvec; i += 64) {
      const int tap = i / nvec, vec = i - tap * nvec;
      const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      float f[8];
      load8<T>(X + ((((long)b * H + iy) * W + ix) * Cin + vec * 8) * (long)sizeof(T), f);
#pragma unroll
      for (int co = 0; co < 8; ++co) {
        if (co < Cout) {
          float wv[8];
          load8<T>(Wt + ((long)co * K + tap * Cin + vec * 8) * (long)sizeof(T), wv);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[co] += f[e] * wv[e];
        }
      }
    }
#pragma unroll
    for (int co = 0; co < 8; ++co) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc[co] += __shfl_xor(acc[co], off);
    }
    if (lane < Cout) {
      float v = 0.f;
#pragma unroll
      for (int co = 0; co < 8; ++co) if (co == lane) v = acc[co];
      Y[(((long)b * Cout + lane) * H + y) * W + x] = v + (bias ? (float)bias[lane] : 0.f);
    }
  }
}

// conv_out, round 5: ONE LANE PER OUTPUT PIXEL for the 16-bit UNet (Cout <= 4).  The kernel above gives a wave to every pixel and re-reads the lane's slice of
// all the weights through the vector L1 for each of them (23 KB per pixel, 24 GB per launch: 1.9 ms for 671 MB of input, 14x its memory time — 0.5 % of the
// benchmark step; round 5's first attempt, a per-wave pixel walk with the weight slice in registers, was twice as slow and is gone).  Here the weights are
// WAVE-UNIFORM — every lane of a wave needs the same [co][tap][8 channels] vector at the same time — so they come through the scalar cache (s_load_dwordx4)
// and feed v_dot2c_f32_{f16,bf16} as its SGPR operand: per (tap, 8 channels) one 16-byte vector load of the lane's own pixel and 16 dot2 instructions.  A lane
// sweeps the 640 contiguous bytes of a neighbour pixel over 40 iterations (every 128-byte line is used whole before it leaves the L1); out-of-image taps load
// the lane's own pixel and are zeroed by a select, so that control flow — and with it the scalar weight loads — stays uniform.  fp32 accumulation of exact
// products in a fixed order: deterministic, batch-invariant; not bitwise the old kernel's summation order (tests compare with F.conv2d to 1e-4).
template <typename T> struct Dot2;
template <> struct Dot2<f16> {
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  static OMG_DEV float run(unsigned a, unsigned b, float c) { return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false); }
};
template <> struct Dot2<bf16> {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  static OMG_DEV float run(unsigned a, unsigned b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false); }
};
template <typename T, int CO>
__global__ __launch_bounds__(256) void conv_out_pixel_kernel(const char* X, int B, int H, int W, int Cin, const char* Wt, const T* bias, float* Y) {
  const long npix = (long)B * H * W;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = pix < npix;
  const long pc = live ? pix : npix - 1;                 // dead lanes of the last block compute the last pixel and store nothing
  const int b = (int)(pc / (H * W)); const int rem = (int)(pc - (long)b * H * W);
  const int y = rem / W, x = rem - y * W;
  const int nvec = Cin / 8;                              // a multiple of 4 (the launcher checks Cin % 32 == 0)
  const long K2 = 9L * Cin * 2;                          // bytes per output channel of the weights [co][ky][kx][ci]
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = 0.f;
  for (int tap = 0; tap < 9; ++tap) {
    const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
    const bool ok = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
    const u32x4* px = (const u32x4*)(X + ((((long)b * H + (ok ? iy : y)) * W + (ok ? ix : x)) * Cin) * 2);
    const char* wt = Wt + (long)tap * Cin * 2;
    for (int v0 = 0; v0 < nvec; v0 += 4) {               // four 16-byte loads in flight per lane: 64 contiguous bytes of the pixel's channel vector
      u32x4 xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = px[v0 + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok) xv[u] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          const u32x4 wv = *(const u32x4*)(wt + co * K2 + (v0 + u) * 16);      // the same address in every lane: scalar cache
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[co] = Dot2<T>::run(xv[u][j], wv[j], acc[co]);
        }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int co = 0; co < CO; ++co) Y[(((long)b * CO + co) * H + y) * W + x] = acc[co] + (bias ? (float)bias[co] : 0.f);
  }
}

// ------------------------------------------------- timestep embedding etc.
// diffusers get_timestep_embedding(t, dim, flip_sin_to_cos=True, downscale_freq_shift=0):
// out[i, j] = cos(t_i * f_j) for j < dim/2 ; sin(t_i * f_{j-dim/2}) otherwise; f_j = exp(-ln(10000) * j / (dim/2))
template <typename T>
__global__ void timestep_embedding_kernel(const float* t, int n, int dim, T* out, long ldo) {
  const int i = blockIdx.x;
  const int half = dim / 2;
  for (int j = threadIdx.x; j < dim; j += blockDim.x) {
    const int jj = j < half ? j : j - half;
    const float freq = expf(-9.210340371976184f * (float)jj / (float)half);
    const float a = t[i] * freq;
    out[(long)i * ldo + j] = (T)(j < half ? cosf(a) : sinf(a));
  }
}

template <typename T>
__global__ void silu_kernel(const T* x, T* y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (T)silu_f((float)x[i]);
}

template <typename T>
__global__ void add_inplace_kernel(char* y, const char* a, long nvec) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    float f[8], g[8];
    unpack8<T>(*(const u32x4*)(y + i * 16), f);
    unpack8<T>(*(const u32x4*)(a + i * 16), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += g[e];
    *(u32x4*)(y + i * 16) = pack8<T>(f);
  }
}

template <typename T>
__global__ void copy2d_kernel(const T* src, long lds_, T* dst, long ldd, long rows, long cols) {
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols, c = i - r * cols;
    dst[r * ldd + c] = src[r * lds_ + c];
  }
}

// out[0:n] = table[(*step_idx) * n : ...]  (2-byte elements): selects the per-step conditioning row block on the
// device so that a captured step graph needs no host-side argument (hipGraph replay freezes kernel arguments).
__global__ void gather_step_kernel(const uint16_t* table, const int* step_idx, uint16_t* out, long n) {
  const long base = (long)(*step_idx) * n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = table[base + i];
}

// ------------------------------------------------------ fused step kernel
struct StepP {
  int C, H, W, Hm, Wm, n_concepts, fuse;
  float gs;
  const float* noise;
  const float* region[OMG_MAX_CONCEPTS];
  const float* masks[OMG_MAX_CONCEPTS];
  const float* coef; int* step_idx; int advance;
  float* latents; int out_dtype; void* mi_next; float* fused_out;
};

__global__ __launch_bounds__(256) void step_kernel(StepP p) {
  const int HW = p.H * p.W;
  const int n = p.C * HW;                 // elements per sample
  const int step = *p.step_idx;
  const float cx = p.coef[step * 4 + 0], ce = p.coef[step * 4 + 1], cin = p.coef[step * 4 + 2];
  // nearest resize F.interpolate(mode='nearest'): src = floor(dst * in/out)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = i / HW; const int pix = i - c * HW;
    const int y = pix / p.W, x = pix - y * p.W;
    float unc0 = p.noise[0 * n + i], unc1 = p.noise[1 * n + i];
    float cnd0 = p.noise[2 * n + i], cnd1 = p.noise[3 * n + i];
    if (p.fuse) {
      const int my = (int)(((long)y * p.Hm) / p.H), mx = (int)(((long)x * p.Wm) / p.W);
      bool any = false;
      float add_u = 0.f, add_c = 0.f;
      for (int k = 0; k < p.n_concepts; ++k) {
        if (p.masks[k] == nullptr) continue;
        const bool on = p.masks[k][(long)my * p.Wm + mx] == 1.0f;
        if (on) {
          any = true;
          if (p.region[k]) { add_u += p.region[k][i]; add_c += p.region[k][n + i]; }
        }
      }
      // new = edit * [union == 0] + sum_c region_c * [mask_c == 1]   (overlaps sum)
      unc1 = (any ? 0.f : unc1) + add_u;
      cnd1 = (any ? 0.f : cnd1) + add_c;
      if (p.fused_out) { p.fused_out[i] = unc1; p.fused_out[n + i] = cnd1; }
    }
    // explicit, identical fma sequences for both samples: the base and the edited sample must stay
    // bit-identical whenever their inputs are (stage 1)
    const float e0 = __fmaf_rn(p.gs, __fsub_rn(cnd0, unc0), unc0);
    const float e1 = __fmaf_rn(p.gs, __fsub_rn(cnd1, unc1), unc1);
    const float l0 = __fmaf_rn(ce, e0, __fmul_rn(cx, p.latents[i]));
    const float l1 = __fmaf_rn(ce, e1, __fmul_rn(cx, p.latents[n + i]));
    p.latents[i] = l0; p.latents[n + i] = l1;
    if (p.mi_next) {
      const float a0 = __fmul_rn(l0, cin), a1 = __fmul_rn(l1, cin);
      if (p.out_dtype == OMG_F16) {
        f16* o = (f16*)p.mi_next;
        o[i] = (f16)a0; o[n + i] = (f16)a1; o[2 * n + i] = (f16)a0; o[3 * n + i] = (f16)a1;
      } else if (p.out_dtype == OMG_BF16) {
        bf16* o = (bf16*)p.mi_next;
        o[i] = (bf16)a0; o[n + i] = (bf16)a1; o[2 * n + i] = (bf16)a0; o[3 * n + i] = (bf16)a1;
      } else {
        float* o = (float*)p.mi_next;
        o[i] = a0; o[n + i] = a1; o[2 * n + i] = a0; o[3 * n + i] = a1;
      }
    }
  }
}

// bump the device step counter after every block of step_kernel has read it: separate 1-thread launch
__global__ void step_advance_kernel(int* step_idx) { *step_idx = *step_idx + 1; }

__global__ void scale_model_input_kernel(const float* latents, const float* cin, int n, int out_dtype, void* out) {
  const float c = *cin;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += gridDim.x * blockDim.x) {
    const float v = latents[i] * c;
    if (out_dtype == OMG_F16) { ((f16*)out)[i] = (f16)v; ((f16*)out)[2 * n + i] = (f16)v; }
    else if (out_dtype == OMG_BF16) { ((bf16*)out)[i] = (bf16)v; ((bf16*)out)[2 * n + i] = (bf16)v; }
    else { ((float*)out)[i] = v; ((float*)out)[2 * n + i] = v; }
  }
}

}  // namespace

extern "C" int omg_conv_in(int dtype, const void* X, int x_is_f32, int B, int Cin, int H, int W, const void* Wt,
                           const void* bias, int Cout, void* workspace, void* Y, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_conv_in: dtype");
  OMG_REQUIRE(X && Wt && Y && workspace && Cout % 8 == 0 && Cin > 0 && 9 * Cin <= 64, "omg_conv_in: args (9*Cin <= 64)");
  const long npix = (long)B * H * W;
  if (npix == 0) return OMG_OK;
  const int KP = 64;
  long blocks = (npix + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OMG_F16) {
    if (x_is_f32) OMG_LAUNCH((im2col_in_kernel<f16, float>), dim3(blocks), dim3(256), 0, s, (const float*)X, B, Cin, H, W, KP, (char*)workspace);
    else OMG_LAUNCH((im2col_in_kernel<f16, f16>), dim3(blocks), dim3(256), 0, s, (const f16*)X, B, Cin, H, W, KP, (char*)workspace);
  } else {
    if (x_is_f32) OMG_LAUNCH((im2col_in_kernel<bf16, float>), dim3(blocks), dim3(256), 0, s, (const float*)X, B, Cin, H, W, KP, (char*)workspace);
    else OMG_LAUNCH((im2col_in_kernel<bf16, bf16>), dim3(blocks), dim3(256), 0, s, (const bf16*)X, B, Cin, H, W, KP, (char*)workspace);
  }
  int rc = omg_check_launch("im2col_in");
  if (rc) return rc;
  omg_gemm_args g{};
  g.dtype = dtype; g.M = (int)npix; g.N = Cout; g.K = KP;
  g.A = workspace; g.lda = KP; g.W = Wt; g.ldw = KP;
  g.groups = 1; g.rows_per_group = (int)npix; g.bias = bias; g.act = OMG_ACT_NONE; g.out_scale = 1.0f;
  g.C = Y; g.ldc = Cout;
  return omg_gemm(&g, stream);
}

// 16-bit -> fp32 copy of a contiguous tensor (the VAE decoder's `sample.to(upscale_dtype)` between mid block and up blocks)
template <typename T>
__global__ __launch_bounds__(256) void cast_f32_kernel(const char* X, float* Y, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float f[8];
    load8<T>(X + i * 16, f);
    store8<float>((char*)(Y + i * 8), f);
  }
}

extern "C" int omg_cast_f32(int dtype, const void* X, float* Y, int64_t n, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_cast_f32: dtype");
  OMG_REQUIRE(X && Y && n % 8 == 0, "omg_cast_f32: args");
  if (n == 0) return OMG_OK;
  long blocks = (n / 8 + 255) / 256; if (blocks > 16384) blocks = 16384;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OMG_F16) OMG_LAUNCH(cast_f32_kernel<f16>, dim3(blocks), dim3(256), 0, s, (const char*)X, Y, (long)(n / 8));
  else OMG_LAUNCH(cast_f32_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const char*)X, Y, (long)(n / 8));
  return omg_check_launch("cast_f32");
}

extern "C" int omg_conv_out(int dtype, const void* X, int B, int H, int W, int Cin, const void* Wt, const void* bias,
                            int Cout, float* Y, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16 || dtype == OMG_F32, "omg_conv_out: dtype");
  OMG_REQUIRE(X && Wt && Y && Cin % 8 == 0 && Cout >= 1 && Cout <= 8, "omg_conv_out: args");
  const long npix = (long)B * H * W;
  if (npix == 0) return OMG_OK;
  long blocks = (npix + 3) / 4; if (blocks > 8192) blocks = 8192;
  hipStream_t s = (hipStream_t)stream;
  // (the pixel kernel reads X and Wt in 16-byte vectors / s_load_dwordx16: unaligned bases take the per-wave kernel — ADVICE r5)
  if (dtype != OMG_F32 && Cout == 4 && Cin % 32 == 0 && (((uintptr_t)X | (uintptr_t)Wt) & 15) == 0) {      // the 16-bit UNet's last convolution: one lane per pixel, weights through the scalar cache
    const unsigned pb = (unsigned)((npix + 255) / 256);
    if (dtype == OMG_F16) OMG_LAUNCH((conv_out_pixel_kernel<f16, 4>), dim3(pb), dim3(256), 0, s, (const char*)X, B, H, W, Cin, (const char*)Wt, (const f16*)bias, Y);
    else OMG_LAUNCH((conv_out_pixel_kernel<bf16, 4>), dim3(pb), dim3(256), 0, s, (const char*)X, B, H, W, Cin, (const char*)Wt, (const bf16*)bias, Y);
    return omg_check_launch("conv_out_pixel");
  }
  if (dtype == OMG_F16) OMG_LAUNCH(conv_out_kernel<f16>, dim3(blocks), dim3(256), 0, s, (const char*)X, B, H, W, Cin, (const char*)Wt, (const f16*)bias, Cout, Y);
  else if (dtype == OMG_BF16) OMG_LAUNCH(conv_out_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const char*)X, B, H, W, Cin, (const char*)Wt, (const bf16*)bias, Cout, Y);
  else OMG_LAUNCH(conv_out_kernel<float>, dim3(blocks), dim3(256), 0, s, (const char*)X, B, H, W, Cin, (const char*)Wt, (const float*)bias, Cout, Y);
  return omg_check_launch("conv_out");
}

extern "C" int omg_timestep_embedding(int dtype, const float* t, int n, int dim, void* out, int64_t ldo, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_timestep_embedding: dtype");
  OMG_REQUIRE(t && out && dim % 2 == 0 && n >= 0, "omg_timestep_embedding: args");
  if (n == 0) return OMG_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OMG_F16) OMG_LAUNCH(timestep_embedding_kernel<f16>, dim3(n), dim3(128), 0, s, t, n, dim, (f16*)out, (long)ldo);
  else OMG_LAUNCH(timestep_embedding_kernel<bf16>, dim3(n), dim3(128), 0, s, t, n, dim, (bf16*)out, (long)ldo);
  return omg_check_launch("timestep_embedding");
}

// ---------------------------------------------------------------- VAE decode helpers (row N1)
// one 256-thread block per row; the row (<= 16384 columns for a 128x128 latent) is read twice (max, then exp + sum) and
// written once: HBM / L2 bound, 3 x cols x 2 bytes per row
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* X, long cols, long ld, float scale) {
  __shared__ float red[8];
  T* row = X + (long)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float m = -3.0e38f;
  for (long c = tid; c < cols; c += 256) m = fmaxf(m, (float)row[c]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
  const float sl2 = scale * 1.4426950408889634f, ml2 = m * 1.4426950408889634f;
  float sum = 0.f;
  for (long c = tid; c < cols; c += 256) sum += __builtin_amdgcn_exp2f((float)row[c] * sl2 - ml2);
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) red[4 + w] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (long c = tid; c < cols; c += 256) row[c] = (T)(__builtin_amdgcn_exp2f((float)row[c] * sl2 - ml2) * inv);
}

__global__ __launch_bounds__(256) void channel_mix_kernel(const float* X, const float* Wm, const float* bias, int Cin, int Cout,
                                                          long HW, float* Y) {
  const int b = blockIdx.y;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    float x[8];
    for (int c = 0; c < Cin; ++c) x[c] = X[((long)b * Cin + c) * HW + p];
    for (int o = 0; o < Cout; ++o) {
      float a = bias ? bias[o] : 0.f;
      for (int c = 0; c < Cin; ++c) a = __fmaf_rn(Wm[o * Cin + c], x[c], a);
      Y[((long)b * Cout + o) * HW + p] = a;
    }
  }
}

extern "C" int omg_softmax_rows(int dtype, void* X, int64_t rows, int64_t cols, int64_t ld, float scale, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_softmax_rows: dtype");
  OMG_REQUIRE(X && rows >= 0 && cols > 0 && ld >= cols && rows < 2147483647L, "omg_softmax_rows: args");
  if (rows == 0) return OMG_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OMG_F16) OMG_LAUNCH(softmax_rows_kernel<f16>, dim3((unsigned)rows), dim3(256), 0, s, (f16*)X, (long)cols, (long)ld, scale);
  else OMG_LAUNCH(softmax_rows_kernel<bf16>, dim3((unsigned)rows), dim3(256), 0, s, (bf16*)X, (long)cols, (long)ld, scale);
  return omg_check_launch("softmax_rows");
}

extern "C" int omg_channel_mix(const float* X, const float* Wm, const float* bias, int B, int Cin, int Cout, int64_t HW,
                               float* Y, void* stream) {
  OMG_REQUIRE(X && Wm && Y && B >= 0 && HW > 0, "omg_channel_mix: args");
  OMG_REQUIRE(Cin >= 1 && Cin <= 8 && Cout >= 1 && Cout <= 8, "omg_channel_mix: 1 <= Cin, Cout <= 8");
  if (B == 0) return OMG_OK;
  long blocks = (HW + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  OMG_LAUNCH(channel_mix_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, X, Wm, bias, Cin, Cout, (long)HW, Y);
  return omg_check_launch("channel_mix");
}

extern "C" int omg_silu(int dtype, const void* x, void* y, int64_t n, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_silu: dtype");
  OMG_REQUIRE(x && y, "omg_silu: null");
  if (n == 0) return OMG_OK;
  long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OMG_F16) OMG_LAUNCH(silu_kernel<f16>, dim3(blocks), dim3(256), 0, s, (const f16*)x, (f16*)y, (long)n);
  else OMG_LAUNCH(silu_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)x, (bf16*)y, (long)n);
  return omg_check_launch("silu");
}

extern "C" int omg_add_inplace(int dtype, void* y, const void* a, int64_t n, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_add_inplace: dtype");
  OMG_REQUIRE(y && a && n % 8 == 0, "omg_add_inplace: n % 8");
  if (n == 0) return OMG_OK;
  const long nvec = n / 8;
  long blocks = (nvec + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OMG_F16) OMG_LAUNCH(add_inplace_kernel<f16>, dim3(blocks), dim3(256), 0, s, (char*)y, (const char*)a, nvec);
  else OMG_LAUNCH(add_inplace_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (char*)y, (const char*)a, nvec);
  return omg_check_launch("add_inplace");
}

extern "C" int omg_copy2d(int dtype, const void* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows, int64_t cols, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_copy2d: dtype");
  OMG_REQUIRE(src && dst, "omg_copy2d: null");
  const long n = rows * cols;
  if (n == 0) return OMG_OK;
  long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipStream_t s = (hipStream_t)stream;
  // f16 and bf16 are both 2-byte copies
  OMG_LAUNCH(copy2d_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)src, (long)lds_, (uint16_t*)dst, (long)ldd, (long)rows, (long)cols);
  return omg_check_launch("copy2d");
}

extern "C" int omg_gather_step(int dtype, const void* table, const int32_t* step_idx, void* out, int64_t n_per_step, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_gather_step: dtype");
  OMG_REQUIRE(table && step_idx && out && n_per_step > 0, "omg_gather_step: args");
  long blocks = (n_per_step + 255) / 256; if (blocks > 1024) blocks = 1024;
  OMG_LAUNCH(gather_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)table, (const int*)step_idx, (uint16_t*)out, (long)n_per_step);
  return omg_check_launch("gather_step");
}

extern "C" int omg_fuse_cfg_step(const omg_step_args* a, void* stream) {
  OMG_REQUIRE(a != nullptr, "omg_fuse_cfg_step: null args");
  OMG_REQUIRE(a->noise_pred && a->coef && a->step_idx && a->latents, "omg_fuse_cfg_step: null operand");
  OMG_REQUIRE(a->n_concepts >= 0 && a->n_concepts <= OMG_MAX_CONCEPTS, "omg_fuse_cfg_step: n_concepts");
  OMG_REQUIRE(a->C > 0 && a->H > 0 && a->W > 0, "omg_fuse_cfg_step: shape");
  if (a->fuse) OMG_REQUIRE(a->Hm > 0 && a->Wm > 0, "omg_fuse_cfg_step: mask shape");
  StepP p{};
  p.C = a->C; p.H = a->H; p.W = a->W; p.Hm = a->Hm; p.Wm = a->Wm; p.n_concepts = a->n_concepts; p.fuse = a->fuse;
  p.gs = a->guidance_scale; p.noise = a->noise_pred;
  for (int k = 0; k < OMG_MAX_CONCEPTS; ++k) { p.region[k] = k < a->n_concepts ? a->region_pred[k] : nullptr; p.masks[k] = k < a->n_concepts ? a->masks[k] : nullptr; }
  p.coef = a->coef; p.step_idx = a->step_idx; p.advance = a->advance;
  p.latents = a->latents; p.out_dtype = a->out_dtype; p.mi_next = a->model_input_next; p.fused_out = a->fused_noise_out;
  const int n = a->C * a->H * a->W;
  int blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
  hipStream_t s = (hipStream_t)stream;
  OMG_LAUNCH(step_kernel, dim3(blocks), dim3(256), 0, s, p);
  if (a->advance) OMG_LAUNCH(step_advance_kernel, dim3(1), dim3(1), 0, s, a->step_idx);
  return omg_check_launch("fuse_cfg_step");
}

extern "C" int omg_scale_model_input(int dtype, const float* latents, const float* coef_cin, int n_per_sample, void* out, void* stream) {
  OMG_REQUIRE(latents && coef_cin && out, "omg_scale_model_input: null");
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16 || dtype == 2, "omg_scale_model_input: dtype");
  int blocks = (2 * n_per_sample + 255) / 256; if (blocks > 1024) blocks = 1024;
  OMG_LAUNCH(scale_model_input_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, latents, coef_cin, n_per_sample, dtype, out);
  return omg_check_launch("scale_model_input");
}
