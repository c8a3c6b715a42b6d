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
// layer (HBM-bound, ~4 % of the layer's attention time at 64x64 tokens), with the keys of every 16-key group stored in the
// order [0-3, 8-11 | 4-7, 12-15] so that a lane's operand is one ds_read_b128.  Both LDS tiles use rows of 128 B with
// chunk ^= (row >> 1) & 7: any 16 rows distinct mod 16 then cover the 16 slots of the 256-byte bank row exactly once
// (conflict-free ds_read_b128; the first version XOR-ed row & 7 and read V in 8-byte pieces: 2-way / 4-way conflicts).
//
// Three kernels share this arithmetic (dispatch at the bottom of the file):
//   attn_fwd_kernel7   Nkv > 128 (self-attention; attn_v7.h): 64 query rows per wave, K / row-major V tiles by LDS-DMA, V^T fragments by the transposing
//                      LDS read, softmax denominator on the 16 x 16 x 32 MFMA, no score masking;
//   attn_fwd_kernel6   Nkv <= 128 (cross-attention over 77 / 93 / 16 tokens): K / V^T resident per (sample, head) over strips of
//                      512 query rows, O through a wave-private LDS transpose; bitwise equal to kernel2;
//   attn_fwd_kernel2   the per-128-row form of the latter: fallback when O is not 16-byte aligned, and the A/B baseline.
#include "common.h"
#include <type_traits>

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

// ------------------------------------------------------------------------------------------------
// v2: the same tiling with the softmax's VALU work roughly halved (at head_dim 64 the 16 MFMAs of a 64-key tile are 512 cycles
// per wave while the v1 softmax issued ~240 VALU instructions, ~1100 cycles: the kernel was VALU-bound at 0.25 of the MFMA peak).
//   * Q is pre-multiplied by scale * log2(e) once, and the score accumulators START at -m (the row's reference maximum) instead
//     of zero: the MFMA chain delivers S' = (K Q'^T) - m directly and P = exp2(S') needs no per-element multiply-subtract.  The
//     16-register tuple of -m is only rewritten when m moves.
//   * m is a REFERENCE maximum, not the running maximum: it is raised (and O, l rescaled, S' shifted) only when some row of the
//     wave sees a score more than ATTN_THR above it (wave-uniform branch; P <= 2^ATTN_THR = 256 stays far inside fp16 / bf16 range,
//     l and O accumulate in fp32).  The first tile always takes the branch, so m >= the first tile's maximum and later, smaller
//     tiles behave exactly as in the running-maximum scheme.  Order at a raise: O *= a, l *= a, S' -= d — every quantity still
//     at the old reference is rescaled exactly once BEFORE this tile's P is exponentiated (cdna guide T13).
//   * P is converted pairwise (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32, round to nearest even).
constexpr float ATTN_THR = 8.0f;

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel2(AttnP p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // K[2], Vt[2]
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  typedef T T2 __attribute__((ext_vector_type(2)));
  typedef float F2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int bq = p.qk_src ? p.qk_src[b] : b;
  const int q0 = blockIdx.x * QB + w * 32;
  int q = q0 + l31;
  const bool qvalid = q < p.Nq;
  if (!qvalid) q = p.Nq - 1;

  V8 qf[4];
  {
    const char* qp = p.Q + ((long)bq * p.q_bs + (long)q * p.ldq + h * 64) * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const V8 raw = *(const V8*)(qp + (ks * 16 + hi * 8) * 2);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[ks][e] = (T)((float)raw[e] * p.scale_log2e);
    }
  }

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
      const int off = row * 128 + ((schunk ^ ((row >> 1) & 7)) << 4);
      *(u32x4*)(smem + buf * TILE + off) = hk[ps];
      *(u32x4*)(smem + (2 + buf) * TILE + off) = hv[ps];
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_ref = 0.f, l_run = 0.f;       // scores and m_ref are in log2 units (Q carries scale * log2 e)
  f32x16 negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;

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

    // ---- S' = K · Q'^T - m_ref
    f32x16 s[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 32 + l31;
      s[i] = Vec<T>::mfma32(*(const V8*)(kt + row * 128 + ((hi ^ ((row >> 1) & 7)) << 4)), qf[0], negm);
    }
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) {
      const int kc = ks * 2 + hi;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + l31;
        V8 kf = *(const V8*)(kt + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
        s[i] = Vec<T>::mfma32(kf, qf[ks], s[i]);
      }
    }
    if (kv0 + KVB > p.Nkv) {      // key tail (last tile only)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= p.Nkv) s[i][r] = -1e30f;
        }
    }
    // ---- tile maximum relative to the reference (pairs of fmaxf fuse into v_max3_f32)
    float mt = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[0][r]), r + 1 < 16 ? s[0][r + 1] : s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[1][r]), s[1][r + 1]);
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    if (t == 0 || __builtin_amdgcn_ballot_w64(mt > ATTN_THR) != 0) {
      // raise (first tile: set) the reference: everything at the old reference is rescaled once, S' moves to the new one
      const float d = t == 0 ? mt : fmaxf(mt, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-d);
      m_ref += d;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[i][r] *= alpha; s[i][r] -= d; }
#pragma unroll
      for (int r = 0; r < 16; ++r) negm[r] = -m_ref;
    }
    float psum = 0.f;
    V8 pf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float e0 = __builtin_amdgcn_exp2f(s[i][r]);
        const float e1 = __builtin_amdgcn_exp2f(s[i][r + 1]);
        psum += e0 + e1;
        const T2 pk = __builtin_convertvector(F2{e0, e1}, T2);
        pf[i][r >> 3][r & 7] = pk[0];
        pf[i][r >> 3][(r & 7) + 1] = pk[1];
      }
    l_run += psum;

    // ---- O^T += V^T · P^T
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int c0 = i * 4 + k2 * 2 + hi;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int row = dt * 32 + l31;
          const V8 vf = *(const V8*)(vt + row * 128 + ((c0 ^ ((row >> 1) & 7)) << 4));
          o[dt] = Vec<T>::mfma32(vf, pf[i][k2], o[dt]);
        }
      }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
  }

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

// ------------------------------------------------------------------------------------------------
// v6 (Nkv <= 128: cross-attention over the 77 text tokens / 93 with image-prompt tokens / 16 IP tokens): v2's arithmetic, bit for
// bit, restructured for a launch that is bound by memory traffic and latency, not by the matrix pipe (two key tiles per query row).
//   * K and V^T of a (sample, head) are staged ONCE per workgroup and stay in LDS for a strip of `q_chunk` query rows: v2 staged
//     them per 128 rows — as many bytes from L2 as Q and O together — and paid their latency plus two block barriers per block;
//   * after that single barrier the four waves run independently over the strip's 128-row steps; the next step's Q rows are
//     requested before the current step computes;
//   * O leaves as whole 128-byte row segments: the normalised tile goes through 4 KB of wave-private LDS (16-byte chunk ^ (row & 7))
//     and is stored as 16 bytes per lane, 8 rows per instruction — v2 wrote 8 bytes per lane, 16 instructions per tile.
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel6(AttnP p, int q_chunk) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE + 4 * 4096];   // K[2], Vt[2], O staging per wave
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  typedef T T2 __attribute__((ext_vector_type(2)));
  typedef float F2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int bq = p.qk_src ? p.qk_src[b] : b;
  const int ntiles = (p.Nkv + KVB - 1) / KVB;       // 1 or 2

  {  // ---- K / V^T of the (sample, head): every tile, once
    const int srow = tid >> 3, schunk = tid & 7;
    const char* kbase = p.K + ((long)bq * p.k_bs + h * 64) * 2;
    const char* vbase = p.Vt + ((long)(b * p.heads + h) * 64) * (long)p.Nkv_pad * 2;
    for (int t = 0; t < ntiles; ++t) {
      const int kv0 = t * KVB;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int row = ps * 32 + srow;
        const int key = kv0 + row;
        u32x4 z = {0u, 0u, 0u, 0u};
        const u32x4 hk = (key < p.Nkv) ? *(const u32x4*)(kbase + ((long)key * p.ldk + schunk * 8) * 2) : z;
        const u32x4 hv = *(const u32x4*)(vbase + ((long)row * p.Nkv_pad + kv0 + schunk * 8) * 2);
        const int off = row * 128 + ((schunk ^ ((row >> 1) & 7)) << 4);
        *(u32x4*)(smem + t * TILE + off) = hk;
        *(u32x4*)(smem + (2 + t) * TILE + off) = hv;
      }
    }
  }
  __syncthreads();

  const int q_begin = blockIdx.x * q_chunk;
  const int q_end = q_begin + q_chunk < p.Nq ? q_begin + q_chunk : p.Nq;
  char* ost = smem + 4 * TILE + w * 4096;
  auto load_q = [&](int q0, V8* raw) {
    int q = q0 + l31;
    if (q > p.Nq - 1) q = p.Nq - 1;
    const char* qp = p.Q + ((long)bq * p.q_bs + (long)q * p.ldq + h * 64) * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) raw[ks] = *(const V8*)(qp + (ks * 16 + hi * 8) * 2);
  };
  V8 raw[4];
  int q0 = q_begin + w * 32;
  if (q0 < q_end) load_q(q0, raw);
  for (; q0 < q_end; q0 += QB) {
    V8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[ks][e] = (T)((float)raw[ks][e] * p.scale_log2e);
    if (q0 + QB < q_end) load_q(q0 + QB, raw);       // in flight under this step's MFMAs

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_ref = 0.f, l_run = 0.f;
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    for (int t = 0; t < ntiles; ++t) {
      const int kv0 = t * KVB;
      const char* kt = smem + t * TILE;
      const char* vt = smem + (2 + t) * TILE;
      // ---- S' = K · Q'^T - m_ref
      f32x16 s[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + l31;
        s[i] = Vec<T>::mfma32(*(const V8*)(kt + row * 128 + ((hi ^ ((row >> 1) & 7)) << 4)), qf[0], negm);
      }
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) {
        const int kc = ks * 2 + hi;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = i * 32 + l31;
          V8 kf = *(const V8*)(kt + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
          s[i] = Vec<T>::mfma32(kf, qf[ks], s[i]);
        }
      }
      if (kv0 + KVB > p.Nkv) {      // key tail (last tile only)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= p.Nkv) s[i][r] = -1e30f;
          }
      }
      // ---- tile maximum relative to the reference (pairs of fmaxf fuse into v_max3_f32)
      float mt = s[0][0];
#pragma unroll
      for (int r = 1; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[0][r]), r + 1 < 16 ? s[0][r + 1] : s[0][r]);
#pragma unroll
      for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[1][r]), s[1][r + 1]);
      mt = fmaxf(mt, __shfl_xor(mt, 32));
      if (t == 0 || __builtin_amdgcn_ballot_w64(mt > ATTN_THR) != 0) {
        // raise (first tile: set) the reference: everything at the old reference is rescaled once, S' moves to the new one
        const float d = t == 0 ? mt : fmaxf(mt, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        m_ref += d;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[i][r] *= alpha; s[i][r] -= d; }
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_ref;
      }
      float psum = 0.f;
      // block (i, k2) = the tile's keys 16 (2 i + k2) .. + 15 = registers 8 k2 .. 8 k2 + 7 of s[i]: its probabilities, then O^T += V^T · P^T for it.
      // Round 6: a block without a real key is skipped.  77 text tokens are one whole tile + 13 keys of the second: 3 of its 4 blocks get no
      // exponentials, no conversions and no P·V MFMAs — 80 exponentials per lane and 32-row step instead of 128, in a loop that is bound by them.
      // A uniform branch (Nkv is a kernel argument); what is skipped were exact zeros (2^-1e30, 0 · V): the same output (a sum that is exactly -0
      // may come out +0).
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          if (kv0 + 16 * (2 * i + k2) >= p.Nkv) continue;      // no real key: its probabilities are exact zeros
          V8 pf;
#pragma unroll
          for (int r = 8 * k2; r < 8 * k2 + 8; r += 2) {
            const float e0 = __builtin_amdgcn_exp2f(s[i][r]);
            const float e1 = __builtin_amdgcn_exp2f(s[i][r + 1]);
            psum += e0 + e1;
            const T2 pk = __builtin_convertvector(F2{e0, e1}, T2);
            pf[r & 7] = pk[0];
            pf[(r & 7) + 1] = pk[1];
          }
          const int c0 = i * 4 + k2 * 2 + hi;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int row = dt * 32 + l31;
            const V8 vf = *(const V8*)(vt + row * 128 + ((c0 ^ ((row >> 1) & 7)) << 4));
            o[dt] = Vec<T>::mfma32(vf, pf, o[dt]);
          }
        }
      l_run += psum;
    }

    // ---- O: normalise, 16 bits, transpose through the wave's LDS strip, row-contiguous 16-byte stores
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = p.out_scale / l_tot;
    const int q = q0 + l31;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[dt][g * 4 + e] * inv;
        if (p.accumulate) {                          // the IP-Adapter branch adds into the text branch's output (same order as v2)
          if (q < p.Nq) {
            const V4 old = *(const V4*)(p.O + ((long)b * p.o_bs + (long)q * p.ldo + h * 64 + dt * 32 + 8 * g + 4 * hi) * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)old[e];
          }
        }
        V4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = (T)v[e];
        *(V4*)(ost + l31 * 128 + (((dt * 4 + g) ^ (l31 & 7)) << 4) + hi * 8) = out;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // one wave, in-order LDS: the tile is written before it is read back
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), ch = lane & 7;
      const u32x4 val = *(const u32x4*)(ost + row * 128 + ((ch ^ (row & 7)) << 4));
      const int qr = q0 + row;
      if (qr < p.Nq) *(u32x4*)(p.O + ((long)b * p.o_bs + (long)qr * p.ldo + h * 64 + ch * 8) * 2) = val;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // ... and read back before the next step overwrites it
  }
}

// ------------------------------------------------------------------------------------------------
// The self-attention kernel's structure (attn_fwd_kernel7 in attn_v7.h; until round 6 also attn_fwd_kernel3 here, the same loop on a K-major V^T
// image — deleted: nothing in the product produced that image for more than 128 keys any more): v2's arithmetic with the LDS traffic per MFMA
// halved and no register staging.
//   * a wave owns TWO 32-row query blocks (64 rows): every K / V^T fragment read from LDS feeds two MFMAs instead of one (v1 / v2
//     read one ds_read_b128 per MFMA; with 8-12 waves per CU the LDS port was as busy as the matrix pipe), and the softmax VALU
//     work of one query block can be scheduled under the MFMAs of the other;
//   * 256 query rows per workgroup (4 waves), so a staged K / V^T tile serves twice as many rows;
//   * tiles go global -> LDS by LDS-DMA (global_load_lds, 1 KiB per wave instruction, swizzle applied to the per-lane SOURCE
//     address): no staging registers, no ds_write instructions; the DMA of tile t+1 is issued right after the barrier that
//     releases its buffer and lands under the compute of tile t.
typedef __attribute__((address_space(3))) void* attn_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* attn_gbl_ptr_t;

#include "attn_v7.h"       // attn_fwd_kernel7: the self-attention kernel — V read row-major through ds_read_b64_tr_b16, XCD-aware block order

// V[B, Nkv, heads*64] -> Vt[B, heads, 64, Nkv_pad]; grid (Nkv_pad/64, heads, B)
template <typename T>
__global__ __launch_bounds__(256) void transpose_v_kernel(const char* V, long ldv, long v_bs, int heads, int Nkv, int Nkv_pad, char* Vt, int mfma_order) {
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
    // key order inside each group of 16: [0-3, 8-11 | 4-7, 12-15] — the 8 keys one lane half of the P·V MFMA consumes per k-step
    // (its P registers hold keys 4 hi + {0..3} and 8 + 4 hi + {0..3}) are then ONE 16-byte chunk of the row
    float f[8];
    const int kb = mfma_order ? (c >> 1) * 16 + (c & 1) * 4 : c * 8;
    const int kstep = mfma_order ? 8 : 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)tile[kb + (e & 3) + (e >> 2) * kstep][d];
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

int g_attn_variant = 0;      // 0 = heuristic (above 128 keys: v7 on row-major V, v2 on a V^T image; v6 up to 128; v2 when O is not 16-byte aligned), 2 / 6 / 7 force one; tools / A-B tests only

}  // namespace

static int g_attn_qchunk = 512;     // v6: query rows per workgroup strip (tools: bits 8.. of the variant word, in units of 128)

static int g_attn_natural_order = 0; // v7, tools: bit 16 of the variant word = blocks in launch order instead of the XCD-aware order
extern "C" void omg_debug_set_attn_variant(int v) {
  g_attn_variant = v & 0xff;
  g_attn_qchunk = ((v >> 8) & 0xff) ? ((v >> 8) & 0xff) * 128 : 512;
  g_attn_natural_order = (v >> 16) & 1;
}
extern "C" int omg_attn_fwd(const omg_attn_args* a, void* stream) {
  OMG_REQUIRE(a != nullptr, "omg_attn_fwd: null args");
  OMG_REQUIRE(a->dtype == OMG_F16 || a->dtype == OMG_BF16, "omg_attn_fwd: dtype");
  OMG_REQUIRE(a->B > 0 && a->heads > 0 && a->Nq > 0 && a->Nkv > 0, "omg_attn_fwd: shape");
  OMG_REQUIRE(a->Vt == nullptr || (a->Nkv_pad % 64 == 0 && a->Nkv_pad >= a->Nkv), "omg_attn_fwd: Nkv_pad");
  OMG_REQUIRE(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldo % 4 == 0, "omg_attn_fwd: strides");
  OMG_REQUIRE(a->Q && a->K && (a->Vt || a->V) && a->O, "omg_attn_fwd: null operand");
  AttnP p = make_params(a);
  dim3 grid((a->Nq + QB - 1) / QB, a->heads, a->B);
  hipStream_t s = (hipStream_t)stream;
  // row-major V (omg_attn_args.V): the self-attention kernel reads it in place — no omg_transpose_v pass.  Up to 128 keys the resident-K/V kernel
  // (v6) and v2 want the V^T image: a caller that passes only V there gets an error, not a silent fallback.
  if (a->V != nullptr && (a->Nkv > 128 || a->Vt == nullptr)) {
    OMG_REQUIRE(g_attn_variant == 0 || g_attn_variant == 7, "omg_attn_fwd: the forced variant reads a V^T image, the caller passed row-major V (tools: pass Vt)");
    // attn_fwd_kernel7 addresses a (sample, head) slice of K / V with 32-bit offsets behind a buffer descriptor
    OMG_REQUIRE(((long)a->Nkv * a->ldk + 64) * 2 < 0x7fffffffL && ((long)a->Nkv * a->ldv + 64) * 2 < 0x7fffffffL, "omg_attn_fwd: a K / V slice beyond 2 GB");
    OMG_REQUIRE(a->ldv % 8 == 0 && a->v_bstride % 8 == 0, "omg_attn_fwd: V strides must be multiples of 8 elements");
    OMG_REQUIRE(a->Nkv > 128 || g_attn_variant == 7, "omg_attn_fwd: row-major V needs more than 128 keys (pass Vt from omg_transpose_v below that)");
    dim3 grid7((a->Nq + 255) / 256, a->heads, a->B);
    const int xcd_order = g_attn_natural_order ? 0 : 1;
    if (a->dtype == OMG_F16) OMG_LAUNCH(attn_fwd_kernel7<f16>, grid7, dim3(256), 0, s, p, (const char*)a->V, (long)a->ldv, (long)a->v_bstride, xcd_order);
    else OMG_LAUNCH(attn_fwd_kernel7<bf16>, grid7, dim3(256), 0, s, p, (const char*)a->V, (long)a->ldv, (long)a->v_bstride, xcd_order);
    return omg_check_launch("attn_fwd_v7");
  }
  OMG_REQUIRE(a->Vt != nullptr, "omg_attn_fwd: Vt (omg_transpose_v) is required up to 128 keys, and above them when V is not given row-major");
  OMG_REQUIRE(g_attn_variant != 7, "omg_attn_fwd: variant 7 reads row-major V (omg_attn_args.V); the caller passed only Vt");
  // a V^T image with more than 128 keys (no product path produces one: value_operand hands the self-attention its row-major V; unaligned views and
  // tools only) runs the per-128-row kernel v2 — correct for any key count, not tuned for long rows
  if ((g_attn_variant == 6 || g_attn_variant == 0) && a->Nkv <= 2 * KVB && a->ldo % 8 == 0 && a->o_bstride % 8 == 0 &&
             ((uintptr_t)a->O & 15) == 0) {   // v6: K / V^T resident per (sample, head), strips of 512 query rows, 16-byte O stores
    const int q_chunk = g_attn_qchunk;
    dim3 grid6((a->Nq + q_chunk - 1) / q_chunk, a->heads, a->B);
    if (a->dtype == OMG_F16) OMG_LAUNCH(attn_fwd_kernel6<f16>, grid6, dim3(256), 0, s, p, q_chunk);
    else OMG_LAUNCH(attn_fwd_kernel6<bf16>, grid6, dim3(256), 0, s, p, q_chunk);
  } else {
    if (a->dtype == OMG_F16) OMG_LAUNCH(attn_fwd_kernel2<f16>, grid, dim3(256), 0, s, p);
    else OMG_LAUNCH(attn_fwd_kernel2<bf16>, grid, dim3(256), 0, s, p);
  }
  return omg_check_launch("attn_fwd");
}

extern "C" int omg_transpose_v(int dtype, const void* V, int64_t ldv, int64_t v_bstride, int B, int heads, int Nkv,
                               int Nkv_pad, void* Vt, int mfma_key_order, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_transpose_v: dtype");
  OMG_REQUIRE(V && Vt && Nkv_pad % 64 == 0 && Nkv_pad >= Nkv && ldv % 8 == 0, "omg_transpose_v: args");
  dim3 grid(Nkv_pad / 64, heads, B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OMG_F16) OMG_LAUNCH(transpose_v_kernel<f16>, grid, dim3(256), 0, s, (const char*)V, (long)ldv, (long)v_bstride, heads, Nkv, Nkv_pad, (char*)Vt, mfma_key_order);
  else OMG_LAUNCH(transpose_v_kernel<bf16>, grid, dim3(256), 0, s, (const char*)V, (long)ldv, (long)v_bstride, heads, Nkv, Nkv_pad, (char*)Vt, mfma_key_order);
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
