// litemla.hip — the pieces of EfficientViT's lightweight multi-scale linear attention (LiteMLA) that are not GEMMs, for gfx950.
// Reference: src/efficientvit/models/nn/ops.py:335-455 (the segmentation hand-off between the two stages).
//
//   omg_dwconv2d         the depthwise s x s convolution of the multi-scale aggregation (ops.py:372-380), NHWC, stride 1,
//                        "same" padding, fp32 accumulation.  (The 1x1 convolution with 3*heads groups that follows it, the qkv
//                        and proj 1x1 convolutions are omg_gemm launches — omg_amd/litemla.py.)
//   omg_relu_linear_att  relu_linear_att (ops.py:405-441) in fp32, as the reference computes it: per (sample, group of 3*dim
//                        channels)  kv = relu(K)^T [V | 1]  over all tokens, then per token  out = relu(q) kv,
//                        out[:dim] / (out[dim] + eps).  Two launches: partial kv per 128-token chunk into a workspace, then the
//                        apply pass sums the partials in chunk order (deterministic: no float atomics) and streams the tokens.
// Both are HBM-bound streaming kernels (a 64x64-token map with 1536 qkv channels is 25 MB); 16-byte accesses per lane.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void dwconv_kernel(const char* X, long ldx, const char* Wt, const char* bias, char* Y, long ldy,
                                                     int B, int H, int Wd, int C, int k) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // over B * H * W * (C / 8)
  const int nv = C >> 3;
  if (idx >= (long)B * H * Wd * nv) return;
  const int v = (int)(idx % nv);
  const long pix = idx / nv;
  const int x = (int)(pix % Wd), y = (int)((pix / Wd) % H);
  const long b = pix / ((long)Wd * H);
  float acc[8];
  if (bias != nullptr) load8<T>(bias + (long)v * 8 * sizeof(T), acc);
  else {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  }
  const int r = k >> 1;
  for (int dy = -r; dy <= r; ++dy) {
    const int yy = y + dy;
    if ((unsigned)yy >= (unsigned)H) continue;
    for (int dx = -r; dx <= r; ++dx) {
      const int xx = x + dx;
      if ((unsigned)xx >= (unsigned)Wd) continue;
      float xv[8], wv[8];
      load8<T>(X + (((b * H + yy) * Wd + xx) * ldx + v * 8) * (long)sizeof(T), xv);
      load8<T>(Wt + ((long)((dy + r) * k + dx + r) * C + v * 8) * (long)sizeof(T), wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(xv[e], wv[e], acc[e]);
    }
  }
  store8<T>(Y + (pix * ldy + v * 8) * (long)sizeof(T), acc);
}

constexpr int RLA_TCH = 128;        // tokens per partial-sum chunk

// grid (chunks, G, B).  LDS: relu(k) [TCH][DIM] and [v | 1] [TCH][DIM + 1]; thread o accumulates kv[i][j], o = i (DIM + 1) + j.
template <typename T, int DIM>
__global__ __launch_bounds__(256) void rla_kv_kernel(const char* QKV, long ld, int HW, int G, float* ws, int nch) {
  __shared__ float ks[RLA_TCH][DIM];
  __shared__ float vs[RLA_TCH][DIM + 1];
  const int ch = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const int t0 = ch * RLA_TCH;
  constexpr int VPT = 2 * DIM / 8;                   // 16-byte vectors of (k, v) per token
  for (int i = tid; i < RLA_TCH * VPT; i += 256) {
    const int t = i / VPT, vv = i - t * VPT;
    const int tok = t0 + t;
    float f[8];
    if (tok < HW) load8<T>(QKV + (((long)b * HW + tok) * ld + (long)g * 3 * DIM + DIM + vv * 8) * (long)sizeof(T), f);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = 0.f;
    }
    if (vv < DIM / 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) ks[t][vv * 8 + e] = __builtin_fmaxf(f[e], 0.f);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) vs[t][(vv - DIM / 8) * 8 + e] = f[e];
    }
    if (vv == 0) vs[t][DIM] = tok < HW ? 1.f : 0.f;
  }
  __syncthreads();
  float* out = ws + (((long)b * G + g) * nch + ch) * (DIM * (DIM + 1));
  for (int o = tid; o < DIM * (DIM + 1); o += 256) {
    const int i = o / (DIM + 1), j = o - i * (DIM + 1);
    float acc = 0.f;
#pragma unroll 8
    for (int t = 0; t < RLA_TCH; ++t) acc = __builtin_fmaf(ks[t][i], vs[t][j], acc);
    out[o] = acc;
  }
}

// grid (token blocks, G, B); a thread = one token x one 8-wide slice of the output
template <typename T, int DIM>
__global__ __launch_bounds__(256) void rla_apply_kernel(const char* QKV, long ld, int HW, int G, const float* ws, int nch, float eps,
                                                        char* OUT, long ldo) {
  __shared__ float kv[DIM][DIM + 1];
  constexpr int S = DIM / 8;                         // slices per token
  constexpr int TPB = 256 / S;                       // tokens per block
  const int g = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const float* part = ws + ((long)b * G + g) * nch * (DIM * (DIM + 1));
  for (int o = tid; o < DIM * (DIM + 1); o += 256) {
    float a = 0.f;
    for (int c = 0; c < nch; ++c) a += part[(long)c * (DIM * (DIM + 1)) + o];          // chunk order: deterministic
    kv[o / (DIM + 1)][o % (DIM + 1)] = a;
  }
  __syncthreads();
  const int tok = blockIdx.x * TPB + tid / S, sl = tid % S;
  if (tok >= HW) return;
  float q[DIM];
  const char* qp = QKV + (((long)b * HW + tok) * ld + (long)g * 3 * DIM) * (long)sizeof(T);
#pragma unroll
  for (int vq = 0; vq < S; ++vq) {
    float f[8];
    load8<T>(qp + vq * 8 * sizeof(T), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) q[vq * 8 + e] = __builtin_fmaxf(f[e], 0.f);
  }
  float den = 0.f, o8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o8[e] = 0.f;
#pragma unroll
  for (int i = 0; i < DIM; ++i) {
    den = __builtin_fmaf(q[i], kv[i][DIM], den);
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = __builtin_fmaf(q[i], kv[i][sl * 8 + e], o8[e]);
  }
  const float inv = 1.0f / (den + eps);
#pragma unroll
  for (int e = 0; e < 8; ++e) o8[e] *= inv;
  store8<T>(OUT + (((long)b * HW + tok) * ldo + (long)g * DIM + sl * 8) * (long)sizeof(T), o8);
}

template <typename T, int DIM>
int rla_launch(const void* QKV, long ld, int B, int HW, int G, float eps, float* ws, void* OUT, long ldo, hipStream_t s) {
  const int nch = (HW + RLA_TCH - 1) / RLA_TCH;
  OMG_LAUNCH((rla_kv_kernel<T, DIM>), dim3(nch, G, B), dim3(256), 0, s, (const char*)QKV, ld, HW, G, ws, nch);
  constexpr int TPB = 256 / (DIM / 8);
  OMG_LAUNCH((rla_apply_kernel<T, DIM>), dim3((HW + TPB - 1) / TPB, G, B), dim3(256), 0, s, (const char*)QKV, ld, HW, G, (const float*)ws, nch, eps,
             (char*)OUT, ldo);
  return omg_check_launch("relu_linear_att");
}

}  // namespace

extern "C" int omg_dwconv2d(int dtype, const void* X, int64_t ldx, int B, int H, int W, int C, int ksize, const void* Wt, const void* bias,
                            void* Y, int64_t ldy, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_dwconv2d: dtype");
  OMG_REQUIRE(X && Wt && Y, "omg_dwconv2d: null operand");
  OMG_REQUIRE(C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "omg_dwconv2d: C, ldx, ldy multiples of 8");
  OMG_REQUIRE(ksize >= 1 && ksize <= 9 && (ksize & 1), "omg_dwconv2d: odd kernel size <= 9");
  const long total = (long)B * H * W * (C / 8);
  if (total == 0) return OMG_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == OMG_F16) OMG_LAUNCH(dwconv_kernel<f16>, grid, dim3(256), 0, s, (const char*)X, (long)ldx, (const char*)Wt, (const char*)bias, (char*)Y, (long)ldy, B, H, W, C, ksize);
  else OMG_LAUNCH(dwconv_kernel<bf16>, grid, dim3(256), 0, s, (const char*)X, (long)ldx, (const char*)Wt, (const char*)bias, (char*)Y, (long)ldy, B, H, W, C, ksize);
  return omg_check_launch("dwconv2d");
}

extern "C" int64_t omg_relu_linear_att_ws_floats(int B, int groups, int dim, int HW) {
  return (int64_t)B * groups * ((HW + RLA_TCH - 1) / RLA_TCH) * dim * (dim + 1);
}

extern "C" int omg_relu_linear_att(int dtype, const void* QKV, int64_t ld, int B, int HW, int groups, int dim, float eps, float* workspace,
                                   void* OUT, int64_t ldo, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_relu_linear_att: dtype");
  OMG_REQUIRE(QKV && workspace && OUT, "omg_relu_linear_att: null operand");
  OMG_REQUIRE(dim == 8 || dim == 16 || dim == 32, "omg_relu_linear_att: dim must be 8, 16 or 32");
  OMG_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && ld >= (int64_t)groups * 3 * dim && ldo >= (int64_t)groups * dim, "omg_relu_linear_att: ld, ldo");
  OMG_REQUIRE(groups > 0 && groups <= 65535 && B <= 65535, "omg_relu_linear_att: grid limits");
  if (B == 0 || HW == 0) return OMG_OK;
  hipStream_t s = (hipStream_t)stream;
#define RLA(TT, DD) return rla_launch<TT, DD>(QKV, (long)ld, B, HW, groups, eps, workspace, OUT, (long)ldo, s)
  if (dtype == OMG_F16) { if (dim == 8) RLA(f16, 8); if (dim == 16) RLA(f16, 16); RLA(f16, 32); }
  if (dim == 8) RLA(bf16, 8);
  if (dim == 16) RLA(bf16, 16);
  RLA(bf16, 32);
#undef RLA
}
