// gemm_epilogue.h — pieces shared by the MFMA GEMM translation units (gemm.hip: fp16 / bf16 operands; gemm_mx8.hip: MX-fp8
// operands): the launch parameter block, the buffer-descriptor LDS-DMA wrapper and the register-direct epilogue of the kernels
// that accumulate the transposed tile.  Everything lives in an anonymous namespace (one copy per translation unit).
#pragma once
#include "common.h"

namespace {

struct GemmP {
  int M, N, K;
  const char* A; long lda;
  const char* W; long ldw;
  const char* A2; long lda2;
  const char* W2; long ldw2;
  int K2, a2_col_block;
  int tile_groups, rows_per_group;   // tile_groups > 1: M tiles never straddle a group
  const int* group_adapter;
  long w_adapter_stride, w2_adapter_stride;
  const char* bias;
  const char* group_bias; long ldgb;
  const char* residual; long ldr;
  int act; float out_scale;
  char* C; long ldc;
  // conv geometry (CONV only)
  int Hin, Win, C1, C2, Hout, Wout, ksize, stride, upsample;
  const char* X2;
  int tiles_m, tiles_n;              // tiles_m is per group
  int dbg;                           // ablation bits (tools only): 1 = no DMA in loop, 2 = no wait/barrier, 4 = no ds_read
  // MX-fp8 operands (gemm_mx8.hip): stage-major scale dwords S[K/128][ld]
  const char* SA; const char* SW;
  int sa_ld, sw_ld;
  long sw_adapter_stride;
  // MX-fp8 OUTPUT (GEGLU epilogue of gemm_mx8.hip only): C is then the e4m3 byte matrix (ldc in bytes) and QS its stage-major
  // scale dwords [n_out / 128][qs_ld] — the operand format of the next omg_gemm_mx8
  char* QS; int qs_ld;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
// Every GEMM/conv variant produces the same bits for the same problem (so that batching requests, which changes the
// tile choice, does not change results): accumulators start at bias (+ the per-sample bias when a sample's rows are a
// multiple of 256, i.e. no tile of any variant straddles samples), the K loop adds the products in the same order, and
// the epilogue is [+ per-row group bias if it was not folded] -> SiLU (hardware reciprocal) -> fma(v, out_scale, residual).
OMG_DEV bool fold_group_bias(const GemmP& p) { return p.group_bias != nullptr && p.rows_per_group % 256 == 0; }

template <int N> OMG_DEV void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// ------------------------------------------------------------------------------------------------
// Register-direct epilogue of the kernels that accumulate the TRANSPOSED tile (v6, v7: mfma(W fragment, A fragment)).
// Register r of lane (l31, hi) of acc[i][j] is C[wm0 + 32 i + l31][wn0 + 32 j + 8 (r >> 2) + 4 hi + (r & 3)]: a lane owns
// ONE output row per i and runs of 4 consecutive columns.  v_permlane32_swap trades the second run of a pair with the
// partner lane (hi ^ 1), after which a lane holds 8 consecutive columns = one 16-byte store; loads of per-column /
// per-element operands (bias, per-sample bias, residual) go through the same exchange the other way round.
// No LDS, no barriers: the staging version cost 10-15 us per 256x256 tile (25-35 % of a K=1280 GEMM) — it serialised
// 32 LDS round trips per wave and waited on vmcnt(0) (which also counts the stores in flight) at every residual load.
template <typename T>
OMG_DEV void swap_runs(unsigned (&q)[4]) {   // q[0..1] = run 0 (4 halves), q[2..3] = run 1
  // v_permlane32_swap v_a, v_b exchanges lanes 32-63 of v_a with lanes 0-31 of v_b (both operands are written).
  // Kept as inline asm with wait states on both sides, tied to the four registers by data dependence, so that neither the
  // producer (v_cvt_pk_f16_f32) nor the consumer (buffer_store) can be scheduled against an instruction that writes both of
  // its operands.  (The corruption first suspected here turned out to be the store-data hazard described in store_runs.)
  asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 4"
               : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
}
// All epilogue memory operations go through buffer descriptors with 32-bit offsets: an offset at or beyond num_records
// makes a load return zeros and a store vanish, so row / column predicates are a v_cndmask on the offset instead of an
// exec-mask branch around every access, and there is no 64-bit address arithmetic per access.
constexpr int EPI_OOB = 0x7f000000;
// The descriptor words go through readfirstlane: built from kernel arguments they ARE wave-uniform, but hipcc kept them in VGPRs
// inside the epilogue and wrapped EVERY buffer load / store in a waterfall loop (4 v_readfirstlane + compare + saveexec + branch:
// ~900 loops in a v7 kernel, the 32 stores of a wave serialised one loop after the other — cdna guide T20).
OMG_DEV __amdgpu_buffer_rsrc_t epi_rsrc(const char* base, long bytes) {
  const unsigned long long a = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int n = __builtin_amdgcn_readfirstlane(base != nullptr ? (int)(bytes < 0x7effff00L ? bytes : 0x7effff00L) : 0);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}
// 16 bytes at byte offset `off` (this lane's 8 consecutive columns) -> the 8 values in accumulator order (runs 0, 1)
template <typename T>
OMG_DEV void decode_runs(const u32x4 raw, float (&f)[8]) {
  unsigned q[4] = {raw[0], raw[1], raw[2], raw[3]};
  swap_runs<T>(q);
  u32x4 sw = {q[0], q[1], q[2], q[3]};
  unpack8<T>(sw, f);
}
template <typename T>
OMG_DEV void load_runs(__amdgpu_buffer_rsrc_t rs, int off, int soff, float (&f)[8]) {
  decode_runs<T>(__builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0), f);
}
template <typename T>
OMG_DEV void store_runs(__amdgpu_buffer_rsrc_t rs, int off, int soff, const float (&f)[8]) {
  u32x4 pk = pack8<T>(f);
  unsigned q[4] = {pk[0], pk[1], pk[2], pk[3]};
  swap_runs<T>(q);
  u32x4 sw = {q[0], q[1], q[2], q[3]};
  // The constant goes into the VGPR offset, never into an SGPR soffset: with `buffer_store_dwordx4 ..., s1 offen` the VALU
  // instruction right behind the store overwrote dword 2 of the store data in the last lanes of each row before the
  // store had read it (observed: the next row block's row index in the output; tools/debug_gemm_small.py).  hipcc adds
  // the wait state only when soffset is an immediate.
  __builtin_amdgcn_raw_buffer_store_b128(sw, rs, off + soff, 0, 0);
}
// tools only (dbg 2048 / 4096): the same 16 bytes per lane written in a row-contiguous pattern (wrong placement, every byte of
// the tile still written once) and / or with the non-temporal bit — prices a transposed, streaming epilogue before it is built
template <typename T>
OMG_DEV void store_runs_dbg(__amdgpu_buffer_rsrc_t rs, int off, const float (&f)[8], bool nt, int pol) {
  u32x4 pk = pack8<T>(f);
  switch (pol) {
    case 1: __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 1); break;      // sc0
    case 2: __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 16); break;     // sc1
    case 3: __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 17); break;     // sc0 sc1
    case 4: __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 3); break;      // nt sc0
    case 5: __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 18); break;     // nt sc1
    case 6: __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 19); break;     // nt sc0 sc1
    default:
      if (nt) __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 2);
      else __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 0);
  }
}

// Accumulators start at the bias instead of zero (one add per output saved in the epilogue, where a wave has no
// partner to hide VALU latency behind).  Same register <-> element map as epilogue_direct.
template <typename T, int MT, int NT>
OMG_DEV bool acc_init_bias(const GemmP& p, f32x16 (&acc)[MT][NT], int lane, int m0, int wn0) {
  const int hi = lane >> 5;
  // the per-sample bias (conv + time embedding) is folded in as well under the rule shared by all variants
  const bool fold_gb = fold_group_bias(p);
  const __amdgpu_buffer_rsrc_t rsB = epi_rsrc(p.bias, (long)p.N * 2);
  const __amdgpu_buffer_rsrc_t rsG = epi_rsrc(fold_gb ? p.group_bias + (long)(m0 / p.rows_per_group) * p.ldgb * 2 : nullptr, (long)p.N * 2);
  // All 2 * 2 * NT loads are issued before the first one is decoded.  Written load -> decode -> load -> decode, every load sat behind an
  // `s_waitcnt vmcnt(0)` + the volatile lane swap of the previous one (round 3, found in the ISA): sixteen L2 round trips in series at the
  // top of EVERY tile — and the first wait also covered the 16 LDS-DMA of stage 0 — i.e. most of the 3-4 us "prologue" the per-CU
  // timeline showed.
  u32x4 rb[NT][2], rg[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int c = wn0 + j * 32 + pr * 16 + hi * 8;      // c >= N is beyond num_records: zeros
      rb[j][pr] = __builtin_amdgcn_raw_buffer_load_b128(rsB, c * 2, 0, 0);
      rg[j][pr] = __builtin_amdgcn_raw_buffer_load_b128(rsG, c * 2, 0, 0);
    }
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      float f[8], g[8];
      decode_runs<T>(rb[j][pr], f);
      decode_runs<T>(rg[j][pr], g);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][j][pr * 8 + e] = f[e] + g[e];
    }
  return p.group_bias != nullptr && !fold_gb;     // true: the epilogue still has to add the per-row group bias
}

// State shared by the epilogue bodies: descriptors, the lane's column offsets and column predicates.
template <int NT>
struct EpiCtx {
  __amdgpu_buffer_rsrc_t rsC, rsR, rsG;
  int hi, l31, wm0, m_end;
  int lane_col;                 // byte offset of the lane's unit (j = 0, pr = 0) inside a row
  int wn0, n_out;               // first output column of the wave tile / width of the output matrix
  char* xl;                     // XE: this wave's 8 KB of LDS for the row-block transposition
  const char* rl;               // residual tile of this wave staged in LDS by LDS-DMA (res_stage_dma), or nullptr
  int voob[NT][2];              // 0 where the unit's columns are inside the matrix, EPI_OOB where they are not
};


// ------------------------------------------------------------------------------------------------
// XE (v7 256x256 only): row-contiguous, non-temporal stores.  The register-direct store above writes 32 rows x 32 bytes per
// instruction; with the non-temporal bit those partial lines go to DRAM one by one (3-4x slower), without it the C tile's
// 128 KB per CU and round (4 MB per XCD = the whole L2) evicts the A / W lines the next tile's K loop wants (K loop
// 36.8 -> 33.4 us on the GEGLU-sized GEMM once the stores stream, tools/gemm_timeline.py bits 2048 | 4096).  So a row block
// (32 rows x ROWB bytes of one wave) is exchanged through 8 KB of wave-private LDS: a lane writes its two 8-byte runs of
// each unit (no v_permlane32_swap needed), reads back 16 bytes of a row whose ROWB / 16 pieces sit in neighbouring lanes,
// and stores 1 KB per instruction as whole 128- / 256-byte row segments with `nt`.  Same values, same packing, same bits.
// (The residual stays on its register-direct loads: fetching it as row segments and handing it round through the same LDS
// image was measured 3-7 % slower on the N = 1280 / 640 projections.)
// LDS image: row r at r * ROWB; 16-byte piece c of row r at piece (c ^ swz(r)), 8-byte halves exchanged when r >= 16
// (32 lanes x 8 bytes of a ds_write_b64 then cover all 64 banks; the ds_read_b128 of 16 lanes cover 256 contiguous bytes).
template <typename T, int ROWB>
OMG_DEV void xe_put(const EpiCtx<4>& cx, int piece, const float (&v)[8]) {
  const u32x4 pk = pack8<T>(v);                     // pk[0..1] = run 0 (columns +0..3), pk[2..3] = run 1 (columns +8..11), both + 4 hi
  const int sw = ROWB == 256 ? (cx.l31 & 15) : ((cx.l31 >> 1) & 7);
  char* b = cx.xl + cx.l31 * ROWB + ((cx.hi ^ (cx.l31 >> 4)) << 3);
  const int c0 = piece ^ sw;
  const u32x2 r0 = {pk[0], pk[1]}, r1 = {pk[2], pk[3]};
  *(u32x2*)(b + (c0 << 4)) = r0;
  *(u32x2*)(b + ((c0 ^ 1) << 4)) = r1;
}
template <typename T, int ROWB>
OMG_DEV void xe_flush(const GemmP& p, const EpiCtx<4>& cx, int i, int col0) {
  constexpr int LPR = ROWB / 16;                    // lanes per row: 16 (256-byte rows) or 8 (GEGLU: 128-byte rows)
  constexpr int RPI = 64 / LPR;                     // rows per store instruction
  constexpr int NU = 32 / RPI;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS serves a wave in order; this is the compiler's fence
  const int lane = cx.hi * 32 + cx.l31;
  const int rq = lane / LPR, c = lane % LPR;
  const bool col_ok = col0 + 8 * c < cx.n_out && !(p.dbg & 1024);
  const int colb = (col0 + 8 * c) * 2;
  u32x4 d[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int rr = u * RPI + rq;
    const int sw = ROWB == 256 ? (rr & 15) : ((rr >> 1) & 7);
    const u32x4 x = *(const u32x4*)(cx.xl + rr * ROWB + ((c ^ sw) << 4));
    if (u * RPI >= 16) d[u] = u32x4{x[2], x[3], x[0], x[1]}; else d[u] = x;     // rows >= 16 hold their halves exchanged
  }
  if constexpr (ROWB == 128) {
    if (p.QS != nullptr) {
      // MX-fp8 output: the lane's 8 columns are a quarter of a 32-wide block (pieces 0-3: block 0 of the wave's 64 columns,
      // 4-7: block 1).  The 16-bit values just packed are what the unfused path would have stored and omg_quant_mx8 read.
      const __amdgpu_buffer_rsrc_t rsQ = epi_rsrc(p.QS, (long)(cx.n_out >> 7) * p.qs_ld * 4);
      const int sbase = ((col0 >> 7) * p.qs_ld) * 4 + ((col0 >> 5) & 3);          // the wave's first block inside its stage dword
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int gm = cx.wm0 + i * 32 + u * RPI + rq;
        float f[8];
        unpack8<T>(d[u], f);
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = __builtin_fmaxf(amax, __builtin_fabsf(f[e]));
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, 1));
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, 2));
        const unsigned be = mx8_scale_exp(amax);
        const float inv = mx8_inv_scale(be);
        const u32x2 o = {mx8_pack4(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv), mx8_pack4(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv)};
        const bool ok = gm < cx.m_end && col_ok;
        __builtin_amdgcn_raw_buffer_store_b64(o, cx.rsC, ok ? gm * (int)p.ldc + col0 + 8 * c : EPI_OOB, 0, 0);
        const unsigned both = be | (__shfl_down(be, 4) << 8);                     // lane c = 0: blocks 0 and 1 of the wave's row
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)both, rsQ, (ok && c == 0) ? sbase + gm * 4 : EPI_OOB, 0, 0);
      }
      asm volatile("" ::: "memory");
      return;
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int gm = cx.wm0 + i * 32 + u * RPI + rq;
    const int off = (gm < cx.m_end && col_ok) ? gm * (int)p.ldc * 2 + colb : EPI_OOB;
    __builtin_amdgcn_raw_buffer_store_b128(d[u], cx.rsC, off, 0, 2);            // nt: stream past L2
  }
  asm volatile("" ::: "memory");
}
// SiLU / per-row group bias / residual decided at run time inside the unit loop: the rare combinations
template <typename T, int MT, int NT, bool RS, bool GENERIC, bool XE = false, bool RL = false>
OMG_DEV void epilogue_rows(const GemmP& p, f32x16 (&acc)[MT][NT], const EpiCtx<NT>& cx, bool has_gb) {
  const float osc = p.out_scale;
  const bool has_rs = GENERIC ? p.residual != nullptr : RS;
  if constexpr (RL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the staged residual tile has landed (own DMAs, own LDS slice)
  u32x4 rraw[2][NT][2];         // residual of row block i: the lane's NT*2 16-byte units, fetched one row block ahead
#define OMG_FETCH_RES(i_, buf_)                                                                            \
  do {                                                                                                     \
    const int gm_ = cx.wm0 + (i_) * 32 + cx.l31;                                                           \
    const int ro_ = gm_ < cx.m_end ? gm_ * (int)p.ldr * 2 + cx.lane_col : EPI_OOB;                         \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                         \
      _Pragma("unroll") for (int pr = 0; pr < 2; ++pr)                                                     \
        rraw[buf_][j][pr] = __builtin_amdgcn_raw_buffer_load_b128(cx.rsR, ro_ | cx.voob[j][pr], (j * 32 + pr * 16) * 2, 0); \
  } while (0)
  if (has_rs && !RL) OMG_FETCH_RES(0, 0);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int gm = cx.wm0 + i * 32 + cx.l31;
    const bool row_ok = gm < cx.m_end;
    if (has_rs && !RL && i + 1 < MT) OMG_FETCH_RES(i + 1, (i + 1) & 1);
    const int ro = row_ok ? gm * (int)p.ldc * 2 + cx.lane_col : EPI_OOB;
    const int go = GENERIC && has_gb && row_ok ? ((gm / p.rows_per_group) * (int)p.ldgb) * 2 + cx.lane_col : EPI_OOB;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[i][j][pr * 8 + e];
        if constexpr (GENERIC) {
          if (has_gb) {    // wave tile straddles samples (tiny feature maps only): per-row group bias, latency not hidden
            float f[8];
            load_runs<T>(cx.rsG, go | cx.voob[j][pr], (j * 32 + pr * 16) * 2, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += f[e];
          }
          if (p.act == OMG_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_fast(v[e]);
          }
        }
        if (has_rs) {
          // RL: the lane's 16 bytes (8 columns) of row l31 from the LDS image res_stage_dma wrote: 16-byte piece c of row r at position c ^ (r & 15)
          const u32x4 rr = RL ? *(const u32x4*)(cx.rl + i * 8192 + cx.l31 * 256 + (((4 * j + 2 * pr + cx.hi) ^ (cx.l31 & 15)) << 4))
                              : rraw[i & 1][j][pr];
          unsigned q[4] = {rr[0], rr[1], rr[2], rr[3]};
          swap_runs<T>(q);
          u32x4 sw = {q[0], q[1], q[2], q[3]};
          float rf[8];
          unpack8<T>(sw, rf);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], osc, rf[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= osc;
        }
        if constexpr (XE) {
          xe_put<T, 256>(cx, 4 * j + 2 * pr, v);
        } else if (p.dbg & (2048 | 4096 | 0xe000)) {
          const int lane_ = cx.hi * 32 + cx.l31, u_ = j * 2 + pr;
          const int off_ = (p.dbg & 2048) ? (cx.wm0 + i * 32 + u_ * 4 + (lane_ >> 4)) * (int)p.ldc * 2 + (cx.lane_col - cx.hi * 16) + (lane_ & 15) * 16
                                          : (ro | cx.voob[j][pr]) + (j * 32 + pr * 16) * 2;
          store_runs_dbg<T>(cx.rsC, off_, v, (p.dbg & 4096) != 0, (p.dbg >> 13) & 7);
        } else
        store_runs<T>(cx.rsC, ro | cx.voob[j][pr], (j * 32 + pr * 16) * 2, v);
        __builtin_amdgcn_sched_barrier(0);   // keeps the scheduler from hoisting every accumulator read to the top (spills)
      }
    if constexpr (XE) xe_flush<T, 256>(p, cx, i, cx.wn0);
  }
#undef OMG_FETCH_RES
}

template <typename T, int MT, int NT, bool XE = false>
OMG_DEV void epilogue_geglu(const GemmP& p, f32x16 (&acc)[MT][NT], const EpiCtx<NT>& cx, int lane_col_g) {
  const float osc = p.out_scale;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int gm = cx.wm0 + i * 32 + cx.l31;
    const int ro = gm < cx.m_end ? gm * (int)p.ldc * 2 + lane_col_g : EPI_OOB;
#pragma unroll
    for (int b = 0; b < NT / 2; ++b)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {       // the gate's polynomial on pairs: four v_pk_fma_f32 per two elements (gelu.h)
          const omg_f32x2 gl = gelu_f2(omg_f32x2{acc[i][2 * b + 1][pr * 8 + e], acc[i][2 * b + 1][pr * 8 + e + 1]});
          o[e] = acc[i][2 * b][pr * 8 + e] * gl[0] * osc;
          o[e + 1] = acc[i][2 * b][pr * 8 + e + 1] * gl[1] * osc;
        }
        if constexpr (XE) xe_put<T, 128>(cx, 4 * b + 2 * pr, o);
        else store_runs<T>(cx.rsC, ro | cx.voob[2 * b][pr], (b * 32 + pr * 16) * 2, o);
        __builtin_amdgcn_sched_barrier(0);
      }
    if constexpr (XE) xe_flush<T, 128>(p, cx, i, cx.wn0 >> 1);
  }
}

// EF = the ONE epilogue form compiled into the kernel: 0 = all of them behind run-time branches (small tiles, v6); 1 = bias only,
// 2 = + residual staged in LDS (res_stage_dma), 3 = GEGLU, 4 = per-row group bias / SiLU / residual decided per unit at run time,
// 5 = + residual by register-direct loads (kernels without a free LDS slice for the staging).
// One form per kernel: with all forms behind run-time branches the 256-accumulator kernels spill inside the epilogue, and a scratch reload
// there waits on vmcnt — i.e. on every store in flight (the 20 % residual penalty of round 2 was mostly that).
template <typename T, int MT, int NT, bool XE = false, int EF = 0>
OMG_DEV void epilogue_direct(const GemmP& p, f32x16 (&acc)[MT][NT], int lane, int wm0, int wn0, int m_end, bool has_gb, char* xl = nullptr,
                             const char* res_lds = nullptr) {
  static_assert(!XE || NT == 4, "the transposed epilogue is written for 128-column wave tiles");
  const bool geglu = EF == 0 ? p.act == OMG_ACT_GEGLU : EF == 3;
  const int n_out = geglu ? p.N / 2 : p.N;
  EpiCtx<NT> cx;
  cx.hi = lane >> 5; cx.l31 = lane & 31; cx.wm0 = wm0; cx.m_end = m_end;
  cx.rsC = epi_rsrc(p.C, p.QS != nullptr ? (long)(p.M - 1) * p.ldc + n_out : ((long)(p.M - 1) * p.ldc + n_out) * 2);
  cx.rsR = epi_rsrc(p.residual, ((long)(p.M - 1) * p.ldr + p.N) * 2);
  cx.rsG = epi_rsrc(has_gb ? p.group_bias : nullptr, 0x7effff00L);
  // the buffer bound only protects the end of the matrix, not the end of a row: per-unit column predicate, as an
  // offset bit pattern that is OR-ed in (row offsets stay far below EPI_OOB's bits)
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) cx.voob[j][pr] = (wn0 + j * 32 + pr * 16 + cx.hi * 8 < p.N && !(p.dbg & 1024)) ? 0 : EPI_OOB;   // dbg 1024 (tools): drop all stores
  cx.lane_col = (wn0 + cx.hi * 8) * 2;
  cx.wn0 = wn0; cx.n_out = n_out; cx.xl = xl; cx.rl = res_lds;
  if constexpr (EF == 1) {
    epilogue_rows<T, MT, NT, false, false, XE>(p, acc, cx, false);
  } else if constexpr (EF == 2) {
    epilogue_rows<T, MT, NT, true, false, XE, true>(p, acc, cx, false);
  } else if constexpr (EF == 3) {
    epilogue_geglu<T, MT, NT, XE>(p, acc, cx, ((wn0 >> 1) + cx.hi * 8) * 2);
  } else if constexpr (EF == 4) {
    epilogue_rows<T, MT, NT, false, true, XE>(p, acc, cx, has_gb);
  } else if constexpr (EF == 5) {
    epilogue_rows<T, MT, NT, true, false, XE>(p, acc, cx, false);
  } else {
    if (geglu) {
      epilogue_geglu<T, MT, NT, XE>(p, acc, cx, ((wn0 >> 1) + cx.hi * 8) * 2);   // GEGLU output is half as wide
    } else if (has_gb || p.act == OMG_ACT_SILU) {
      epilogue_rows<T, MT, NT, false, true, XE>(p, acc, cx, has_gb);
    } else if (p.residual != nullptr) {
      epilogue_rows<T, MT, NT, true, false, XE>(p, acc, cx, false);
    } else {
      epilogue_rows<T, MT, NT, false, false, XE>(p, acc, cx, false);
    }
  }
}

// 16 bytes per lane, global -> LDS, through a buffer descriptor.  A non-template wrapper on purpose: with value-dependent
// arguments the builtin's checks are deferred to instantiation time, where the host pass of hipcc silently drops the kernel.
OMG_DEV void dma16(__amdgpu_buffer_rsrc_t rs, char* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds, 16, voff, soff, 0, 0);
}

// The residual sub-tile of one wave (128 rows x 128 columns, 32 KB) into LDS by LDS-DMA: 32 instructions of 4 rows x 256 bytes, every one
// of them whole 256-byte row segments (the register-direct loads of the epilogue touch 32 rows x 32 bytes per instruction: 64 cache-line
// look-ups each; a K = 1280 projection with a residual ran 20 % below the same GEMM without one).  Image: row r of row block i at
// i * 8192 + r * 256, its 16-byte piece c at position c ^ (r & 15) — the permutation is applied on the source side (the DMA destination is
// lane-linear), and a ds_read_b128 of 16 lanes then covers all 64 banks.  Rows / columns outside the matrix read zeros (never stored).
OMG_DEV void res_stage_dma(const GemmP& p, char* lds, int lane, int wm0, int wn0, int m_end) {
  const __amdgpu_buffer_rsrc_t rsR = epi_rsrc(p.residual, ((long)(p.M - 1) * p.ldr + p.N) * 2);
  const int rq = lane >> 4, pos = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int rr = u * 4 + rq;
      const int gm = wm0 + i * 32 + rr;
      const int col = wn0 + ((pos ^ (rr & 15)) << 3);
      const int off = (gm < m_end && col < p.N) ? (gm * (int)p.ldr + col) * 2 : EPI_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsR, (lds_ptr_t)(lds + i * 8192 + u * 1024), 16, off, 0, 0, 0);
    }
}

// 4 bytes per lane (256 contiguous bytes of LDS per instruction): per-row scale dwords of the MX-fp8 convolution
OMG_DEV void dma4(__amdgpu_buffer_rsrc_t rs, char* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds, 4, voff, soff, 0, 0);
}

}  // namespace
