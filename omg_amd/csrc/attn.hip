// attn.hip — flash-style attention forward for gfx950, head_dim 64, with
// prompt-to-prompt probability borrowing (see include/omg_hip.h, omg_attn_fwd).
//
// One workgroup = 4 wave64 = 128 query rows of one (batch, head); each wave owns 32
// query rows.  Everything is computed in the "swapped" orientation so that a query
// row is lane-local:
//     S^T[key][q] = K · Q^T      (v_mfma_f32_32x32x16, A = K tile from LDS, B = Q in VGPRs)
//     O^T[d][q]   = V^T · P^T    (A = V^T tile from LDS, B = P straight from the S^T registers)
// With the 32x32 C/D layout (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) a lane
// holds, for its query q = lane&31, sixteen keys per 32-key tile; the online-softmax
// max/sum are in-register reductions plus ONE exchange with lane^32, and the rescale
// of O^T is a lane-local multiply.  The MFMA contraction does not care about the
// ORDER of keys inside a 16-key k-step, so P is fed to the second MFMA in exactly the
// order the first one produced it, and the matching key permutation is applied to the
// V^T operand's LDS read (two 8-byte reads) — no cross-lane shuffles of P at all.
//
// V is consumed K-major (V^T[d][key]); omg_transpose_v produces that image once per
// layer (HBM-bound, ~4 % of the layer's attention time at 64x64 tokens).
#include "common.h"

namespace {

constexpr int QB = 128;      // query rows per workgroup
constexpr int KVB = 64;      // keys per tile
constexpr int TILE = KVB * 64 * 2;   // 8 KiB: K tile [64 keys][64 d], V^T tile [64 d][64 keys]

struct AttnP {
  int B, heads, Nq, Nkv, Nkv_pad;
  const char* Q; long ldq, q_bs;
  const char* K; long ldk, k_bs;
  const char* Vt;
  const int* qk_src;
  float scale_log2e;
  int accumulate; float out_scale;
  char* O; long ldo, o_bs;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnP p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // K[2], Vt[2]
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int bq = p.qk_src ? p.qk_src[b] : b;       // batch supplying Q and K
  const int q0 = blockIdx.x * QB + w * 32;
  int q = q0 + l31;
  const bool qvalid = q < p.Nq;
  if (!qvalid) q = p.Nq - 1;

  // Q fragments: B operand, lane (n = q, k-octet = hi) for each 16-wide d step
  V8 qf[4];
  {
    const char* qp = p.Q + ((long)bq * p.q_bs + (long)q * p.ldq + h * 64) * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const V8*)(qp + (ks * 16 + hi * 8) * 2);
  }

  // staging coordinates: thread -> (row, 16-B chunk), two passes of 32 rows
  const int srow = tid >> 3, schunk = tid & 7;
  const char* kbase = p.K + ((long)bq * p.k_bs + h * 64) * 2;
  const char* vbase = p.Vt + ((long)(b * p.heads + h) * 64) * (long)p.Nkv_pad * 2;
  u32x4 hk[2], hv[2];
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = ps * 32 + srow;
      const int key = kv0 + row;
      u32x4 z = {0u, 0u, 0u, 0u};
      hk[ps] = (key < p.Nkv) ? *(const u32x4*)(kbase + ((long)key * p.ldk + schunk * 8) * 2) : z;
      hv[ps] = *(const u32x4*)(vbase + ((long)row * p.Nkv_pad + kv0 + schunk * 8) * 2);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = ps * 32 + srow;
      const int off = row * 128 + ((schunk ^ (row & 7)) << 4);
      *(u32x4*)(smem + buf * TILE + off) = hk[ps];
      *(u32x4*)(smem + (2 + buf) * TILE + off) = hv[ps];
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int ntiles = (p.Nkv + KVB - 1) / KVB;
  load_tile(0);
  store_tile(0);
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    const int kv0 = t * KVB;
    __syncthreads();
    if (t + 1 < ntiles) load_tile(kv0 + KVB);
    const char* kt = smem + buf * TILE;
    const char* vt = smem + (2 + buf) * TILE;

    // ---- S^T = K · Q^T : two 32-key tiles
    f32x16 s[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[i][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kc = ks * 2 + hi;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + l31;
        V8 kf = *(const V8*)(kt + row * 128 + ((kc ^ (row & 7)) << 4));
        s[i] = Vec<T>::mfma32(kf, qf[ks], s[i]);
      }
    }
    // ---- mask the key tail (last tile only)
    if (kv0 + KVB > p.Nkv) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= p.Nkv) s[i][r] = -1e30f;
        }
    }
    // ---- online softmax (row = this lane's query; the other 16+16 keys live in lane^32)
    float mt = s[0][0];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[i][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2e);
    m_run = m_new;
    const float mb = m_new * p.scale_log2e;
    float psum = 0.f;
    V8 pf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[i][r] * p.scale_log2e - mb);
        psum += e;
        pf[i][r >> 3][r & 7] = (T)e;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

    // ---- O^T += V^T · P^T : A = V^T (row d, keys in P's order), B = P
#pragma unroll
    for (int i = 0; i < 2; ++i)        // 32-key tile
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) { // 16-key k-step
        const int c0 = i * 4 + k2 * 2;  // 16-B chunk holding keys [base+4hi .. +3]; chunk+1 holds base+8+4hi
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int row = dt * 32 + l31;
          const char* rp = vt + row * 128 + hi * 8;
          V4 lo = *(const V4*)(rp + ((c0 ^ (row & 7)) << 4));
          V4 hi4 = *(const V4*)(rp + (((c0 + 1) ^ (row & 7)) << 4));
          V8 vf = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
          o[dt] = Vec<T>::mfma32(vf, pf[i][k2], o[dt]);
        }
      }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
  }

  // ---- finish: combine the two half-rows' sums, normalise, store
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = p.out_scale / l_tot;
  if (qvalid) {
    char* op = p.O + ((long)b * p.o_bs + (long)q * p.ldo + h * 64) * 2;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * hi;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[dt][g * 4 + e] * inv;
        V4* dst = (V4*)(op + d * 2);
        if (p.accumulate) {
          V4 old = *dst;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)old[e];
        }
        V4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = (T)v[e];
        *dst = out;
      }
  }
}

// V[B, Nkv, heads*64] -> Vt[B, heads, 64, Nkv_pad]; grid (Nkv_pad/64, heads, B)
template <typename T>
__global__ __launch_bounds__(256) void transpose_v_kernel(const char* V, long ldv, long v_bs, int heads, int Nkv, int Nkv_pad, char* Vt) {
  __shared__ T tile[64][66];
  const int tid = threadIdx.x;
  const int kv0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  // load 64 keys x 64 d: thread -> (key = tid/8 + 32*ps, 8 d)
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int key = ps * 32 + (tid >> 3), c = tid & 7;
    float f[8];
    if (kv0 + key < Nkv) {
      unpack8<T>(*(const u32x4*)(V + ((long)b * v_bs + (long)(kv0 + key) * ldv + h * 64 + c * 8) * 2), f);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[key][c * 8 + e] = (T)f[e];
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int d = ps * 32 + (tid >> 3), c = tid & 7;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)tile[c * 8 + e][d];
    *(u32x4*)(Vt + (((long)(b * heads + h) * 64 + d) * Nkv_pad + kv0 + c * 8) * 2) = pack8<T>(f);
  }
}

// ---------------- protocol mode: materialised probabilities ----------------
// P[bh, q, :] = softmax(scale * Q K^T): one wave per query row, keys strided over lanes.
// Used only when a controller needs the explicit tensor (non-identity mapper); not a hot path.
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_kernel(AttnP p, char* P) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
  if (q >= p.Nq) return;
  const int bq = p.qk_src ? p.qk_src[b] : b;
  float qv[64];
  const char* qp = p.Q + ((long)bq * p.q_bs + (long)q * p.ldq + h * 64) * 2;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float f[8];
    unpack8<T>(*(const u32x4*)(qp + c * 16), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[c * 8 + e] = f[e];
  }
  char* prow = P + (((long)(b * p.heads + h) * p.Nq + q) * (long)p.Nkv) * 2;
  const float sc = p.scale_log2e;
  // pass 1: max
  float mx = -1e30f;
  for (int k = lane; k < p.Nkv; k += 64) {
    const char* kp = p.K + ((long)bq * p.k_bs + (long)k * p.ldk + h * 64) * 2;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float f[8];
      unpack8<T>(*(const u32x4*)(kp + c * 16), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += f[e] * qv[c * 8 + e];
    }
    mx = fmaxf(mx, acc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float sum = 0.f;
  for (int k = lane; k < p.Nkv; k += 64) {
    const char* kp = p.K + ((long)bq * p.k_bs + (long)k * p.ldk + h * 64) * 2;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float f[8];
      unpack8<T>(*(const u32x4*)(kp + c * 16), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += f[e] * qv[c * 8 + e];
    }
    sum += exp2f((acc - mx) * sc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float inv = 1.f / sum;
  for (int k = lane; k < p.Nkv; k += 64) {
    const char* kp = p.K + ((long)bq * p.k_bs + (long)k * p.ldk + h * 64) * 2;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float f[8];
      unpack8<T>(*(const u32x4*)(kp + c * 16), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += f[e] * qv[c * 8 + e];
    }
    ((T*)prow)[k] = (T)(exp2f((acc - mx) * sc) * inv);
  }
}

// O[b,q,h*64+d] = sum_k P[bh,q,k] V[b,k,h*64+d]: one wave per query row, lane = d
template <typename T>
__global__ __launch_bounds__(256) void attn_apply_probs_kernel(const char* P, const char* V, long ldv, long v_bs,
                                                               int heads, int Nq, int Nkv, char* O, long ldo, long o_bs) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
  if (q >= Nq) return;
  const T* prow = (const T*)(P + (((long)(b * heads + h) * Nq + q) * (long)Nkv) * 2);
  const T* vp = (const T*)(V + ((long)b * v_bs + h * 64 + lane) * 2);
  float acc = 0.f;
  for (int k = 0; k < Nkv; ++k) acc += (float)prow[k] * (float)vp[(long)k * ldv];
  ((T*)(O + ((long)b * o_bs + (long)q * ldo + h * 64) * 2))[lane] = (T)acc;
}

AttnP make_params(const omg_attn_args* a) {
  AttnP p{};
  p.B = a->B; p.heads = a->heads; p.Nq = a->Nq; p.Nkv = a->Nkv; p.Nkv_pad = a->Nkv_pad;
  p.Q = (const char*)a->Q; p.ldq = a->ldq; p.q_bs = a->q_bstride;
  p.K = (const char*)a->K; p.ldk = a->ldk; p.k_bs = a->k_bstride;
  p.Vt = (const char*)a->Vt; p.qk_src = a->qk_src;
  p.scale_log2e = a->scale * 1.4426950408889634f;
  p.accumulate = a->accumulate; p.out_scale = a->out_scale;
  p.O = (char*)a->O; p.ldo = a->ldo; p.o_bs = a->o_bstride;
  return p;
}

}  // namespace

extern "C" int omg_attn_fwd(const omg_attn_args* a, void* stream) {
  OMG_REQUIRE(a != nullptr, "omg_attn_fwd: null args");
  OMG_REQUIRE(a->dtype == OMG_F16 || a->dtype == OMG_BF16, "omg_attn_fwd: dtype");
  OMG_REQUIRE(a->B > 0 && a->heads > 0 && a->Nq > 0 && a->Nkv > 0, "omg_attn_fwd: shape");
  OMG_REQUIRE(a->Nkv_pad % 64 == 0 && a->Nkv_pad >= a->Nkv, "omg_attn_fwd: Nkv_pad");
  OMG_REQUIRE(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldo % 4 == 0, "omg_attn_fwd: strides");
  OMG_REQUIRE(a->Q && a->K && a->Vt && a->O, "omg_attn_fwd: null operand");
  AttnP p = make_params(a);
  dim3 grid((a->Nq + QB - 1) / QB, a->heads, a->B);
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == OMG_F16) OMG_LAUNCH(attn_fwd_kernel<f16>, grid, dim3(256), 0, s, p);
  else OMG_LAUNCH(attn_fwd_kernel<bf16>, grid, dim3(256), 0, s, p);
  return omg_check_launch("attn_fwd");
}

extern "C" int omg_transpose_v(int dtype, const void* V, int64_t ldv, int64_t v_bstride, int B, int heads, int Nkv,
                               int Nkv_pad, void* Vt, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_transpose_v: dtype");
  OMG_REQUIRE(V && Vt && Nkv_pad % 64 == 0 && Nkv_pad >= Nkv && ldv % 8 == 0, "omg_transpose_v: args");
  dim3 grid(Nkv_pad / 64, heads, B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OMG_F16) OMG_LAUNCH(transpose_v_kernel<f16>, grid, dim3(256), 0, s, (const char*)V, (long)ldv, (long)v_bstride, heads, Nkv, Nkv_pad, (char*)Vt);
  else OMG_LAUNCH(transpose_v_kernel<bf16>, grid, dim3(256), 0, s, (const char*)V, (long)ldv, (long)v_bstride, heads, Nkv, Nkv_pad, (char*)Vt);
  return omg_check_launch("transpose_v");
}

extern "C" int omg_attn_probs(const omg_attn_args* a, void* P, void* stream) {
  OMG_REQUIRE(a && P && a->Q && a->K, "omg_attn_probs: null");
  OMG_REQUIRE(a->dtype == OMG_F16 || a->dtype == OMG_BF16, "omg_attn_probs: dtype");
  OMG_REQUIRE(a->ldq % 8 == 0 && a->ldk % 8 == 0, "omg_attn_probs: strides");
  AttnP p = make_params(a);
  dim3 grid((a->Nq + 3) / 4, a->heads, a->B);
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == OMG_F16) OMG_LAUNCH(attn_probs_kernel<f16>, grid, dim3(256), 0, s, p, (char*)P);
  else OMG_LAUNCH(attn_probs_kernel<bf16>, grid, dim3(256), 0, s, p, (char*)P);
  return omg_check_launch("attn_probs");
}

extern "C" int omg_attn_apply_probs(int dtype, const void* P, const void* V, int64_t ldv, int64_t v_bstride, int B,
                                    int heads, int Nq, int Nkv, void* O, int64_t ldo, int64_t o_bstride, void* stream) {
  OMG_REQUIRE(P && V && O, "omg_attn_apply_probs: null");
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_attn_apply_probs: dtype");
  dim3 grid((Nq + 3) / 4, heads, B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OMG_F16) OMG_LAUNCH(attn_apply_probs_kernel<f16>, grid, dim3(256), 0, s, (const char*)P, (const char*)V, (long)ldv, (long)v_bstride, heads, Nq, Nkv, (char*)O, (long)ldo, (long)o_bstride);
  else OMG_LAUNCH(attn_apply_probs_kernel<bf16>, grid, dim3(256), 0, s, (const char*)P, (const char*)V, (long)ldv, (long)v_bstride, heads, Nq, Nkv, (char*)O, (long)ldo, (long)o_bstride);
  return omg_check_launch("attn_apply_probs");
}
