// gemm_mx8.hip — block-scaled fp8 (OCP MX: e4m3 elements, one E8M0 scale per 32 consecutive K elements) GEMM for gfx950.
//
//   C[M,N] = epi( dequant(Aq, SA)[M,K] · dequant(Wq, SW)[N,K]^T )        dequant(q, s)[r][k] = q[r][k] * 2^(s[r][k/32] - 127)
//
// The contraction runs on v_mfma_scale_f32_32x32x64_f8f6f4 — the only large-K low-precision MFMA of CDNA4 and the only way to
// the 5 PFLOP/s fp8 rate (MI355X_MICROARCH.md; a non-scaled fp8 MFMA runs at the bf16 rate).  The instruction takes 64 K
// elements per issue and applies 2^(sa + sb - 254) to each 32-wide block's partial dot product before the fp32 accumulation.
// Operand layout, MEASURED on the part (tools/mx8_probe.py: one non-zero byte walked through K against per-block scales):
// lane (row i = lane & 31, half h = lane >> 5) holds in registers 0-3 the bytes k = 16 h .. 16 h + 15 and in registers 4-7 the
// bytes k = 32 + 16 h .. 32 + 16 h + 15 of the k-step — i.e. block 0 (k 0..31) is registers 0-3 of BOTH lane halves and block 1
// registers 4-7 — while the scale of block b of row i is the selected byte of the scale register of lane i + 32 b.
//
// Kernel = the structure of gemm_kernel_v7 (gemm.hip), at the same BYTES per stage: 256x256 output tile on four waves
// (128x128 per wave, 256 accumulator registers), double-buffered LDS stages of 128 bytes of K per row (= 128 fp8 elements,
// twice v7's 64 halves) filled by LDS-DMA through buffer descriptors into the same XOR-swizzled 128-byte-row image, the K
// loop software-pipelined inside the wave with one block barrier per stage.  A stage is 2 k-steps of 16 MFMAs x 64 cycles
// (v7: 4 x 16 x 32): per unit time the LDS reads, DMA instructions and bytes moved are the same as v7's, the FLOPs double.
//
// Scales travel with the stage: they are stored STAGE-MAJOR as one dword per (row, stage) — the four E8M0 bytes of the
// row's four 32-blocks inside that 128-wide stage — S[K/128][rows] (uint32), so the 256 rows of a tile are 1 KiB contiguous
// for a given stage = ONE LDS-DMA instruction per operand per stage (wave 0: A scales, wave 1: W scales), and a lane fetches
// its row's dword with a ds_read_b32.  After `>> 8*hi` byte 0 / byte 2 of that dword are the scales lane half hi has to supply
// for k-step 0 / 1 (blocks hi and 2 + hi of the stage; op_sel picks the byte).  The quantiser (quant_mx8_kernel, below) writes this layout; weights are quantised once.
//
// Epilogue, tile -> CU mapping, per-sample weight slots, bias folding: shared with v7 (gemm_epilogue.h).  The output and the
// bias / residual operands stay fp16 / bf16.
#include "common.h"
#include "gemm_epilogue.h"

typedef int i32x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int MXK = 128;                       // K bytes (= fp8 elements) per row per stage
constexpr int MX_TILE = 256 * MXK;             // one operand tile of a stage: 32 KiB
constexpr int MX_SC = 2 * MX_TILE;             // scale dwords behind the two operand tiles: [256 A rows][256 W rows]
constexpr int MX_STAGE = 2 * MX_TILE + 2048;   // 67,584 B per stage, 135,168 B for the double buffer

OMG_DEV f32x16 mfma_mx8(i32x8 a, i32x8 b, f32x16 c, int sa, int sb, int ks) {
  // cbsz = blgp = 0: both operands fp8 e4m3.  op_sel (byte of the scale register): 0 for k-step 0, 2 for k-step 1.
  return ks == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb)
                 : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, sa, 2, sb);
}

// D1 = number of the 16 operand DMA instructions of stage kt+2 issued during the LAST k-step of stage kt (the rest follow in
// k-step 0 of stage kt+1).  The buffer they fill is released by the barrier of stage kt, and they must have landed by the
// barrier of stage kt+1: a DMA issued in the last k-step has two k-steps (~2000 cycles) to land, one issued in k-step 0 one.
//
// CONV = true: the A operand is an NHWC MX-fp8 feature map (bytes [B*H*W][C], C % 128 == 0) read as the implicit GEMM of a 3x3,
// stride 1, pad 1 convolution: stage kt covers channels 128 (kt % (C/128)) .. + 127 of tap kt / (C/128), row m of the tile
// reads pixel (y + dy, x + dx) of its sample, and taps outside the map are the descriptor's out-of-range zeros.  A-operand
// scales are stored per pixel, SA[C/128][B*H*W] dwords (omg_groupnorm_mx8 writes them): every wave fetches the dwords of 64 of
// the tile's rows with ONE 4-byte-per-lane LDS-DMA per stage at the tap-shifted pixel (out of range: byte 0 = 2^-127, times
// zeros).  Weights [Cout][9 C] and their scales are those of the Linear path (omg_quant_mx8 of the packed conv weight).
template <typename T, int D1, bool CONV = false, bool XE = true, int EF = 0>
__global__ __launch_bounds__(256, 1) void gemm_mx8_kernel(GemmP p) {
  constexpr int MT = 4, NT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  // ---- tile mapping: identical to v7 (XCD-aware bijective remap, 8 M-tiles x all N-tiles per group)
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_per_group = p.tiles_m * p.tiles_n;
  const int grp = bid / tiles_per_group;
  const int t_in = bid - grp * tiles_per_group;
  int tm, tn;
  {
    const int per_group = 8 * p.tiles_n;
    const int gid = t_in / per_group;
    const int first_m = gid * 8;
    const int gsz = (p.tiles_m - first_m) < 8 ? (p.tiles_m - first_m) : 8;
    const int r = t_in - gid * per_group;
    tm = first_m + (r % gsz);
    tn = r / gsz;
  }
  const int m_base = (p.tile_groups > 1) ? grp * p.rows_per_group : 0;
  const int m_end = (p.tile_groups > 1) ? m_base + p.rows_per_group : p.M;
  const int m0 = m_base + tm * 256;
  const int n0 = tn * 256;
  int adapter = 0;
  if (p.group_adapter != nullptr) adapter = p.group_adapter[grp];
  if (p.w_adapter_stride != 0 && adapter < 0) return;
  const char* Wp = p.W + (p.w_adapter_stride != 0 ? (long)adapter * p.w_adapter_stride : 0);
  const char* SWp = p.SW + (p.w_adapter_stride != 0 ? (long)adapter * p.sw_adapter_stride * 4 : 0);
  const int nk = p.K / MXK;
  const int cpt = CONV ? p.C1 / MXK : 1;                   // stages per tap

  // ---- descriptors: exact sizes, so rows past the end and nothing else read as zero
  const long a_bytes = CONV ? (long)p.M * p.C1 : (long)(p.M - 1) * p.lda + p.K;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)a_bytes, 0x00020000);
  const long w_bytes = (long)(p.N - 1) * p.ldw + p.K;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, (int)w_bytes, 0x00020000);
  // scale arrays: [nk][ld] dwords.  Wave 0 stages the A scales, wave 1 the W scales (1 KiB each per stage).
  // (CONV: wave 1 stages the W scales this way; the A scales are fetched per row by every wave, below.)
  const bool sA = !CONV && w == 0;
  const bool sc_wave = CONV ? w == 1 : w < 2;
  // with weight slots the W-scale base is shifted by the slot's columns: the descriptor ends where the tensor ends, and a
  // lane's rows are bounded by the slot's N rows (not by the row count of all slots)
  const long s_shift = (!sA && p.w_adapter_stride != 0) ? (long)adapter * p.sw_adapter_stride * 4 : 0;
  const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(sA ? p.SA : SWp), 0,
      (int)((long)nk * (sA ? p.sa_ld : p.sw_ld) * 4 - s_shift), 0x00020000);
  const int s_step = (sA ? p.sa_ld : p.sw_ld) * 4;                     // bytes between stages
  const int s_rows = sA ? p.sa_ld : (p.w_adapter_stride != 0 ? (p.N + 3) & ~3 : p.sw_ld);
  // a lane carries 4 consecutive rows' dwords; rows past the array's row count must not wrap into the next stage's rows
  const int s_row0 = (sA ? m0 : n0) + lane * 4;
  const int voffS = s_row0 + 3 < s_rows ? s_row0 * 4 : 0x7ffffff0;
  const int ldoS = MX_SC + (sA ? 0 : 1024);
  // CONV A scales: lane -> row m0 + 64 w + lane of the tile
  const int hw = CONV ? p.Hout * p.Wout : 1;
  const __amdgpu_buffer_rsrc_t rsSA = __builtin_amdgcn_make_buffer_rsrc((void*)p.SA, 0, CONV ? (int)((long)cpt * p.M * 4) : 0, 0x00020000);
  int s_pix = 0, s_yx = 0x7fff0000;
  if constexpr (CONV) {
    const int gm = m0 + w * 64 + lane;
    if (gm < m_end) {
      const int rem = gm % hw;
      const int y = rem / p.Wout;
      s_pix = gm * 4; s_yx = (y << 16) | (rem - y * p.Wout);
    }
  }

  // ---- operand DMA: one instruction moves 8 rows x 128 B; wave w owns row blocks w, w+4, ..., w+28 of A and of W.
  // Row block i of a wave is 32 rows further: one VGPR offset per operand + an SGPR step (rows past the matrix end are out of
  // the descriptor's range and read as zeros; rows of the next group only feed accumulator rows the epilogue never stores).
  const int prow = lane >> 3, ppos = lane & 7;
  const int dchunk = (ppos ^ ((w & 1) * 4 + (prow >> 1))) * 16;       // ((row >> 1) & 7) with row = (w + 4i) * 8 + prow
  const int voffA0 = CONV ? (m0 + w * 8 + prow) * p.C1 + dchunk : (int)((long)(m0 + w * 8 + prow) * p.lda) + dchunk;
  const int voffW0 = (int)((long)(n0 + w * 8 + prow) * p.ldw) + dchunk;
  const int stepA = CONV ? 32 * p.C1 : (int)(32 * p.lda), stepW = (int)(32 * p.ldw);
  const int ldo = w * 1024;
  int a_yx[8];                   // CONV: (y << 16) | x of the lane's row in each of the wave's 8 A row blocks; rows past the end never pass the bounds test
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    a_yx[d] = 0x7fff0000;
    if constexpr (CONV) {
      const int gm = m0 + (w + 4 * d) * 8 + prow;
      if (gm < m_end) {
        const int rem = gm % hw;
        const int y = rem / p.Wout;
        a_yx[d] = (y << 16) | (rem - y * p.Wout);
      }
    }
  }

  const int wm = w >> 1, wn = w & 1;
  f32x16 acc[MT][NT];
  const bool gb_epi = acc_init_bias<T, MT, NT>(p, acc, lane, m0, n0 + wn * 128);

  // ---- fragment addressing.  k-step ks, lane half hi -> registers 0-3 from the 16-byte chunk 4 ks + hi (block 2 ks), registers
  // 4-7 from chunk 4 ks + 2 + hi (block 2 ks + 1), at their swizzled positions (chunk ^ ((row >> 1) & 7); tile bases are
  // multiples of 16 rows, so the term is the lane's).  The lane's scale byte is that of block 2 ks + hi.
  const int sw = (l31 >> 1) & 7;
  int foff[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) foff[ks][h] = l31 * 128 + (((ks * 4 + h * 2 + hi) ^ sw) << 4);
  const int aB = wm * (128 * 128), bB = MX_TILE + wn * (128 * 128);
  const int sAoff = MX_SC + (wm * 128 + l31) * 4, sWoff = MX_SC + 1024 + (wn * 128 + l31) * 4;
  const int sshift = hi * 8;

  int koff = 0, kst = 0;
  int tdy = 0, tdx = 0, tapoff = 0, stapoff = 0;           // CONV: the stage's tap and its byte offsets into the map / the scale plane
#define MX_PREP(kt_)                                                                                       \
  do {                                                                                                     \
    koff = (kt_) * MXK; kst = (kt_);                                                                       \
    if constexpr (CONV) {                                                                                  \
      const int tap_ = (kt_) / cpt; const int cb_ = (kt_) - tap_ * cpt;                                    \
      tdy = tap_ / 3 - 1; tdx = tap_ - (tap_ / 3) * 3 - 1;                                                 \
      tapoff = (tdy * p.Wout + tdx) * p.C1 + cb_ * MXK;                                                    \
      stapoff = (cb_ * p.M + tdy * p.Wout + tdx) * 4;                                                      \
    }                                                                                                      \
  } while (0)
#define MX_INSIDE(yx_) ((unsigned)(((yx_) >> 16) + tdy) < (unsigned)p.Hout && (unsigned)(((yx_) & 0xffff) + tdx) < (unsigned)p.Wout)
#define MX_DMA(d_, nb_)                                                                                    \
  do {                                                                                                     \
    if ((d_) < 8) {                                                                                        \
      if constexpr (CONV) dma16(rsA, (nb_) + ldo + ((d_) & 7) * 4096,                                      \
                                MX_INSIDE(a_yx[(d_) & 7]) ? voffA0 + ((d_) & 7) * stepA + tapoff : 0x7ffffff0, 0); \
      else dma16(rsA, (nb_) + ldo + ((d_) & 7) * 4096, voffA0, koff + ((d_) & 7) * stepA);                 \
    } else dma16(rsW, (nb_) + MX_TILE + ldo + ((d_) & 7) * 4096, voffW0, koff + ((d_) & 7) * stepW);       \
  } while (0)
#define MX_DMAS(nb_)                                                                                       \
  do {                                                                                                     \
    if (sc_wave) dma16(rsS, (nb_) + ldoS, voffS, kst * s_step);                                            \
    if constexpr (CONV) dma4(rsSA, (nb_) + MX_SC + w * 256, MX_INSIDE(s_yx) ? s_pix + stapoff : 0x7ffffff0, 0); \
  } while (0)
  i32x8 af[2][MT], bf[2][NT];
  int sa[MT], sb[NT];            // scale dwords of the current stage (byte 0: k-step 0, byte 2: k-step 1)
  int san[MT], sbn[NT];          // ... of the next stage, read during the last k-step and moved over at the stage boundary
  // fragment r_ of k-step ks_ (r_ < 4: W fragment r_, else A fragment r_ - 4): two 16-byte reads
#define MX_RD1(f_, sb_, ks_, r_)                                                                           \
  do {                                                                                                     \
    const bool isA_ = (r_) >= NT;                                                                          \
    const int idx_ = isA_ ? (r_) - NT : (r_);                                                              \
    const char* q_ = (sb_) + (isA_ ? aB : bB) + idx_ * 4096;                                               \
    const u32x4 lo_ = *(const u32x4*)(q_ + foff[ks_][0]);                                                  \
    const u32x4 hi_ = *(const u32x4*)(q_ + foff[ks_][1]);                                                  \
    const i32x8 v_ = {(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
    if (isA_) af[f_][idx_] = v_; else bf[f_][idx_] = v_;                                                   \
  } while (0)
#define MX_RDS(da_, db__, sb_)                                                                             \
  do {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) da_[i] = (int)(*(const unsigned*)((sb_) + sAoff + i * 128) >> sshift); \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) db__[j] = (int)(*(const unsigned*)((sb_) + sWoff + j * 128) >> sshift); \
  } while (0)
#define MX_MM1(f_, ks_, n_)                                                                                \
  acc[(n_) >> 2][(n_) & 3] = mfma_mx8(bf[f_][(n_) & 3], af[f_][(n_) >> 2], acc[(n_) >> 2][(n_) & 3], sb[(n_) & 3], sa[(n_) >> 2], ks_)
  // One k-step: 16 MFMAs from fragment set f_; between them the 8 fragments (16 ds_read_b128) of the NEXT k-step into set
  // 1-f_ (RD_), the scale dwords of the next stage (RDS_), and DMA instructions d0_ .. d0_+dn_-1 (+ the scale DMA, DSC_).
#define MX_KSTEP(f_, ks_, RD_, rb_, rks_, RDS_, DMA_, d0_, dn_, db_, DSC_)                                 \
  do {                                                                                                     \
    _Pragma("unroll") for (int n_ = 0; n_ < 16; ++n_) {                                                    \
      MX_MM1(f_, ks_, n_);                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if ((RD_) && n_ < 8) MX_RD1(1 - (f_), rb_, rks_, n_);                                                \
      if ((RDS_) && n_ == 8) MX_RDS(san, sbn, rb_);                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if ((DMA_) && n_ < (dn_)) MX_DMA((d0_) + n_, db_);                                                   \
      if ((DSC_) && n_ == 15) MX_DMAS(db_);                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
    }                                                                                                      \
  } while (0)

  // ---- prologue: stage 0 completely; the first D1 instructions (+ scales) of stage 1; first fragments and scales
  MX_PREP(0);
#pragma unroll
  for (int d = 0; d < 16; ++d) MX_DMA(d, smem);
  MX_DMAS(smem);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  MX_PREP(1);
  if (nk > 1) {
#pragma unroll
    for (int d = 0; d < D1; ++d) MX_DMA(d, smem + MX_STAGE);
    MX_DMAS(smem + MX_STAGE);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) MX_RD1(0, smem, 0, r);
  MX_RDS(sa, sb, smem);

  // One stage.  HAS1_/HAS2_ (stage kt+1 / kt+2 exist) are literal so that the steady-state body is one basic block.
#define MX_STAGE_BODY(HAS1_, HAS2_)                                                                        \
  do {                                                                                                     \
    const char* cur = smem + (kt & 1) * MX_STAGE;                                                          \
    char* nxt = smem + ((kt + 1) & 1) * MX_STAGE;                                                          \
    /* k-step 0: fragments of k-step 1 of this stage; the remaining 16 - D1 DMAs of stage kt+1 */          \
    MX_KSTEP(0, 0, true, cur, 1, false, HAS1_, D1, 16 - D1, nxt, false);                                   \
    /* stage kt+1 has landed (this wave's part) and this wave's reads of `cur` are complete: join the block */ \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                                                          \
    if (HAS2_) MX_PREP(kt + 2);                                                                            \
    /* k-step 1: first fragments + scales of stage kt+1; the first D1 DMAs (+ scales) of stage kt+2 into the released buffer */ \
    MX_KSTEP(1, 1, HAS1_, nxt, 0, HAS1_, HAS2_, 0, D1, (char*)cur, HAS2_);                                 \
    if (HAS1_) {                                                                                           \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) sa[i] = san[i];                                       \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) sb[j] = sbn[j];                                       \
    }                                                                                                      \
  } while (0)
  int kt = 0;
  for (; kt < nk - 2; ++kt) MX_STAGE_BODY(true, true);
  if (kt < nk - 1) { MX_STAGE_BODY(true, false); ++kt; }
  MX_STAGE_BODY(false, false);
#undef MX_STAGE_BODY
#undef MX_KSTEP
#undef MX_MM1
#undef MX_RDS
#undef MX_RD1
#undef MX_DMAS
#undef MX_DMA
#undef MX_INSIDE
#undef MX_PREP
  // Transposed streaming epilogue (gemm_epilogue.h, XE): the 8 KB per wave it needs are taken from the stage buffer the LAST
  // stage does not use — released for every wave by that stage's barrier, never read or filled again.
  epilogue_direct<T, MT, NT, XE, EF>(p, acc, lane, m0 + wm * 128, n0 + wn * 128, m_end, gb_epi, smem + (nk & 1) * MX_STAGE + w * 8192);
}

// ------------------------------------------------------------------------------------------------
// gemm_mx8_kernel_p (round 6) — the Linear form of the kernel above as a PERSISTENT tile walk with the next tile's stage 0
// in flight under the current tile's last k-step and epilogue.  The MX-fp8 GEMM is the kernel furthest below its roofline and the only one that is not
// power-capped: a K = 1280 tile is ten 128-wide stages — 9.3 us of matrix pipe at 2.2 GHz — inside ~18 us, the rest being the latency of its first stage
// (2 - 3 us), its epilogue (4 - 5 us) and the dispatch gap.  gemm_kernel_v12 (gemm_v12.h) showed on the 16-bit kernel that loads issued IN FRONT of the
// epilogue's stores do not queue behind them; the same move here, on the two-stage K loop as it is (the five-buffer ring does not fit beside 2 KB of scale
// images per stage, and the loop is not what this kernel lacks):
//   * grid = the CU count (a multiple of 8, so that a block's tiles stay on one XCD's share of the XCD-aware order); block b walks tiles b, b + grid, ...;
//   * behind the barrier of a tile's LAST stage every wave has finished reading that stage's buffer: the 16 operand DMAs + the scale DMA of the NEXT
//     tile's stage 0 go into it, one per MFMA slot of the last k-step — the slots that carry stage kt + 2's DMAs in the steady state and are empty there.
//     They always go into buffer 0 and the epilogue's XE staging region is always buffer 1's operand area (both free behind that barrier whatever the
//     stage count), so the buffers keep their roles from tile to tile and the K loop its immediate LDS offsets;
//   * the next tile's coordinates are decoded at the TOP of a tile (while stage 0 is still landing) and held in SGPRs through the K loop; its descriptors
//     and the lane's source offsets are written — unconditionally (gemm_v12.h's lesson: a conditional write keeps the old value alive round the loop) —
//     after the current tile's last DMA has been issued; the epilogue works from its own copies of the tile's coordinates and from a lane id taken
//     afresh from the exec-mask count, so that LICM cannot hoist its ~50 lane constants in front of the tile loop.
// Same loads, same MFMA order per accumulator, same epilogue code: bitwise identical to the one-tile-per-block form (tests/test_mx8_gpu.py forces both).
struct MxTile { int m0, n0, m_end, grp; };
OMG_DEV MxTile mx_decode_tile(const GemmP& p, int vb, int ntiles) {
  int bid = vb;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_per_group = p.tiles_m * p.tiles_n;
  const int grp = bid / tiles_per_group;
  const int t_in = bid - grp * tiles_per_group;
  const int per_group = 8 * p.tiles_n;
  const int gid = t_in / per_group;
  const int first_m = gid * 8;
  const int gsz = (p.tiles_m - first_m) < 8 ? (p.tiles_m - first_m) : 8;
  const int r = t_in - gid * per_group;
  const int tm = first_m + (r % gsz);
  const int tn = r / gsz;
  const int m_base = (p.tile_groups > 1) ? grp * p.rows_per_group : 0;
  MxTile t;
  t.m_end = (p.tile_groups > 1) ? m_base + p.rows_per_group : p.M;
  t.m0 = m_base + tm * 256;
  t.n0 = tn * 256;
  t.grp = grp;
  return t;
}
// the first virtual block id >= vb (in steps of `step`) whose tile is computed (weight slot >= 0), or >= ntiles
OMG_DEV int mx_next_tile(const GemmP& p, int vb, int step, int ntiles) {
  if (p.w_adapter_stride == 0 || p.group_adapter == nullptr) return vb;
  while (vb < ntiles) {
    const MxTile t = mx_decode_tile(p, vb, ntiles);
    if (p.group_adapter[t.grp] >= 0) break;
    vb += step;
  }
  return vb;
}
OMG_DEV int mx_fresh_lane() {
  int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return l;
}

template <typename T, int D1>
__global__ __launch_bounds__(256, 1) void gemm_mx8_kernel_p(GemmP p) {
  constexpr int MT = 4, NT = 4;
  constexpr bool XE = true;
  constexpr int EF = 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;
  const int ntiles = p.tile_groups * p.tiles_m * p.tiles_n;
  const int step = (int)gridDim.x;
  const int nk = p.K / MXK;
  int vb = mx_next_tile(p, (int)blockIdx.x, step, ntiles);
  if (vb >= ntiles) return;

  const long a_bytes = (long)(p.M - 1) * p.lda + p.K;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)a_bytes, 0x00020000);
  const long w_bytes = (long)(p.N - 1) * p.ldw + p.K;
  __amdgpu_buffer_rsrc_t rsW, rsS;
  const bool sA = w == 0;               // wave 0 stages the A scales, wave 1 the W scales (1 KiB each per stage)
  const bool sc_wave = w < 2;
  const int s_step = (sA ? p.sa_ld : p.sw_ld) * 4;
  const int ldoS = MX_SC + (sA ? 0 : 1024);
  const int prow = lane >> 3, ppos = lane & 7;
  const int dchunk = (ppos ^ ((w & 1) * 4 + (prow >> 1))) * 16;
  const int stepA = (int)(32 * p.lda), stepW = (int)(32 * p.ldw);
  const int ldo = w * 1024;
  int m0, n0, m_end;
  int voffA0, voffW0, voffS;
  // the tile at (m0_, n0_) of weight-slot group grp_: descriptors and the lane's source offsets
#define MXP_TILE(m0_, n0_, mend_, grp_)                                                                    \
  do {                                                                                                     \
    m0 = (m0_); n0 = (n0_); m_end = (mend_);                                                               \
    int adapter_ = 0;                                                                                      \
    if (p.group_adapter != nullptr) adapter_ = p.group_adapter[grp_];                                      \
    const char* Wp_ = p.W + (p.w_adapter_stride != 0 ? (long)adapter_ * p.w_adapter_stride : 0);           \
    const char* SWp_ = p.SW + (p.w_adapter_stride != 0 ? (long)adapter_ * p.sw_adapter_stride * 4 : 0);    \
    rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp_, 0, (int)w_bytes, 0x00020000);                      \
    const long s_shift_ = (!sA && p.w_adapter_stride != 0) ? (long)adapter_ * p.sw_adapter_stride * 4 : 0; \
    rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(sA ? p.SA : SWp_), 0, (int)((long)nk * (sA ? p.sa_ld : p.sw_ld) * 4 - s_shift_), 0x00020000); \
    const int s_rows_ = sA ? p.sa_ld : (p.w_adapter_stride != 0 ? (p.N + 3) & ~3 : p.sw_ld);              \
    const int s_row0_ = (sA ? m0 : n0) + lane * 4;                                                         \
    voffS = s_row0_ + 3 < s_rows_ ? s_row0_ * 4 : 0x7ffffff0;                                              \
    voffA0 = (int)((long)(m0 + w * 8 + prow) * p.lda) + dchunk;                                            \
    voffW0 = (int)((long)(n0 + w * 8 + prow) * p.ldw) + dchunk;                                            \
  } while (0)
  {
    const MxTile t = mx_decode_tile(p, vb, ntiles);
    MXP_TILE(t.m0, t.n0, t.m_end, t.grp);
  }

  const int wm = w >> 1, wn = w & 1;
  const int sw = (l31 >> 1) & 7;
  int foff[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) foff[ks][h] = l31 * 128 + (((ks * 4 + h * 2 + hi) ^ sw) << 4);
  const int aB = wm * (128 * 128), bB = MX_TILE + wn * (128 * 128);
  const int sAoff = MX_SC + (wm * 128 + l31) * 4, sWoff = MX_SC + 1024 + (wn * 128 + l31) * 4;
  const int sshift = hi * 8;

  int koff = 0, kst = 0;
#define MX_PREP(kt_) do { koff = (kt_) * MXK; kst = (kt_); } while (0)
#define MX_DMA(d_, nb_)                                                                                    \
  do {                                                                                                     \
    if ((d_) < 8) dma16(rsA, (nb_) + ldo + ((d_) & 7) * 4096, voffA0, koff + ((d_) & 7) * stepA);          \
    else dma16(rsW, (nb_) + MX_TILE + ldo + ((d_) & 7) * 4096, voffW0, koff + ((d_) & 7) * stepW);         \
  } while (0)
#define MX_DMAS(nb_) do { if (sc_wave) dma16(rsS, (nb_) + ldoS, voffS, kst * s_step); } while (0)
  i32x8 af[2][MT], bf[2][NT];
  int sa[MT], sb[NT];
  int san[MT], sbn[NT];
#define MX_RD1(f_, sb_, ks_, r_)                                                                           \
  do {                                                                                                     \
    const bool isA_ = (r_) >= NT;                                                                          \
    const int idx_ = isA_ ? (r_) - NT : (r_);                                                              \
    const char* q_ = (sb_) + (isA_ ? aB : bB) + idx_ * 4096;                                               \
    const u32x4 lo_ = *(const u32x4*)(q_ + foff[ks_][0]);                                                  \
    const u32x4 hi_ = *(const u32x4*)(q_ + foff[ks_][1]);                                                  \
    const i32x8 v_ = {(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
    if (isA_) af[f_][idx_] = v_; else bf[f_][idx_] = v_;                                                   \
  } while (0)
#define MX_RDS(da_, db__, sb_)                                                                             \
  do {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) da_[i] = (int)(*(const unsigned*)((sb_) + sAoff + i * 128) >> sshift); \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) db__[j] = (int)(*(const unsigned*)((sb_) + sWoff + j * 128) >> sshift); \
  } while (0)
#define MX_MM1(f_, ks_, n_)                                                                                \
  acc[(n_) >> 2][(n_) & 3] = mfma_mx8(bf[f_][(n_) & 3], af[f_][(n_) >> 2], acc[(n_) >> 2][(n_) & 3], sb[(n_) & 3], sa[(n_) >> 2], ks_)
#define MX_KSTEP(f_, ks_, RD_, rb_, rks_, RDS_, DMA_, d0_, dn_, db_, DSC_)                                 \
  do {                                                                                                     \
    _Pragma("unroll") for (int n_ = 0; n_ < 16; ++n_) {                                                    \
      MX_MM1(f_, ks_, n_);                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if ((RD_) && n_ < 8) MX_RD1(1 - (f_), rb_, rks_, n_);                                                \
      if ((RDS_) && n_ == 8) MX_RDS(san, sbn, rb_);                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if ((DMA_) && n_ < (dn_)) MX_DMA((d0_) + n_, db_);                                                   \
      if ((DSC_) && n_ == 15) MX_DMAS(db_);                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
    }                                                                                                      \
  } while (0)

  // ---- the first tile's stage 0.  EVERY tile's stage 0 lives in buffer 0 (no parity: with run-time buffer roles the fragment reads lose their
  // immediate offsets — the first version of this kernel swapped the buffers from tile to tile and its K loop ran 9 % slower at 8192^3).  Buffer 0 is free
  // behind the last stage's barrier whatever the stage count: an even count's last stage computes from buffer 1 (buffer 0 was released a stage earlier), an
  // odd count's from buffer 0 itself (released by this barrier).  The epilogue's XE staging region is buffer 1's operand area, equally free there.
  MX_PREP(0);
#pragma unroll
  for (int d = 0; d < 16; ++d) MX_DMA(d, smem);
  MX_DMAS(smem);
  const bool gb_epi = p.group_bias != nullptr && !fold_group_bias(p);
  for (;;) {
    // the tile after this one: decoded here, while stage 0 is landing, and carried through the K loop in SGPRs
    const int vbn = mx_next_tile(p, vb + step, step, ntiles);
    const bool has_next = vbn < ntiles;
    const MxTile tnx = mx_decode_tile(p, has_next ? vbn : vb, ntiles);
    const int e_m0 = m0, e_n0 = n0, e_mend = m_end;
    // the accumulators are DEFINED here, per tile (no loop-carried phi on 256 registers: with acc initialised at the loop's bottom hipcc shuffled it
    // between the AGPR and the VGPR half and spilled it — 371 VGPRs, scratch traffic INSIDE the K loop); the bias loads' latency meets the wait for
    // stage 0 below
    f32x16 acc[MT][NT];
    (void)acc_init_bias<T, MT, NT>(p, acc, mx_fresh_lane(), m0, n0 + wn * 128);
    char* const b0 = smem;                               // stages 0, 2, 4, ... of every tile
    char* const b1 = smem + MX_STAGE;                    // stages 1, 3, 5, ...
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    MX_PREP(1);
    if (nk > 1) {
#pragma unroll
      for (int d = 0; d < D1; ++d) MX_DMA(d, b1);
      MX_DMAS(b1);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) MX_RD1(0, b0, 0, r);
    MX_RDS(sa, sb, b0);

#define MX_STAGE_BODY(HAS1_, HAS2_)                                                                        \
    do {                                                                                                   \
      const char* cur = smem + (kt & 1) * MX_STAGE;                                                        \
      char* nxt = smem + ((kt + 1) & 1) * MX_STAGE;                                                        \
      MX_KSTEP(0, 0, true, cur, 1, false, HAS1_, D1, 16 - D1, nxt, false);                                 \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                          \
      __builtin_amdgcn_s_barrier();                                                                        \
      if (HAS2_) MX_PREP(kt + 2);                                                                          \
      MX_KSTEP(1, 1, HAS1_, nxt, 0, HAS1_, HAS2_, 0, D1, (char*)cur, HAS2_);                               \
      if (HAS1_) {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) sa[i] = san[i];                                     \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) sb[j] = sbn[j];                                     \
      }                                                                                                    \
    } while (0)
    int kt = 0;
    for (; kt < nk - 2; ++kt) MX_STAGE_BODY(true, true);
    if (kt < nk - 1) { MX_STAGE_BODY(true, false); ++kt; }
    {   // the last stage: no load of this tile is left to issue
      const char* cur = smem + (kt & 1) * MX_STAGE;
      MX_KSTEP(0, 0, true, cur, 1, false, false, 0, 0, (char*)cur, false);
      // from here on the DMA state belongs to the NEXT tile (without one: this tile again, out of range, so that nothing is written conditionally)
      MXP_TILE(tnx.m0, tnx.n0, tnx.m_end, tnx.grp);
      if (!has_next) { voffA0 = 0x7ffffff0; voffW0 = 0x7ffffff0; voffS = 0x7ffffff0; }
      MX_PREP(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // no wave reads a stage buffer any more: buffer 0 takes the next tile's stage 0
      MX_KSTEP(1, 1, false, cur, 0, false, true, 0, 16, b0, true);
    }
#undef MX_STAGE_BODY
    // XE staging region of the epilogue: buffer 1's operand area
    char* const xe = b1 + w * 8192;
    const int lane_e = mx_fresh_lane();
    epilogue_direct<T, MT, NT, XE, EF>(p, acc, lane_e, e_m0 + wm * 128, e_n0 + wn * 128, e_mend, gb_epi, xe);
    if (!has_next) break;
    vb = vbn;
  }
#undef MX_KSTEP
#undef MX_MM1
#undef MX_RDS
#undef MX_RD1
#undef MX_DMAS
#undef MX_DMA
#undef MX_PREP
#undef MXP_TILE
}

// ------------------------------------------------------------------------------------------------
// Quantiser: X[M, K] (fp16 / bf16, row stride ldx) -> Q[M, K] e4m3 bytes (row stride ldq) + S[K/128][s_ld] scale dwords.
// Scale of a 32-block: the smallest power of two 2^e with amax / 2^e <= 448 (the e4m3 maximum), stored as E8M0 e + 127 —
// no element saturates, the largest keeps all three mantissa bits.  A block of zeros gets byte 0 (2^-127).
// A workgroup takes 32 rows; per iteration one 128-wide stage: thread t -> row t / 8, elements 16 (t % 8) .. + 15, so two
// neighbouring lanes share a 32-block (one DPP exchange for the amax), a row's 128 output bytes are one whole cache line and
// the 32 rows' scale dwords of the stage are one more.
template <typename T>
__global__ __launch_bounds__(256) void quant_mx8_kernel(const char* x, long ldx, int M, int K, char* q, long ldq, unsigned* s, int s_ld) {
  const int t = threadIdx.x;
  const int row = blockIdx.x * 32 + (t >> 3);
  const int part = t & 7;
  if (row >= M) return;
  const char* xr = x + (long)row * ldx * 2 + part * 32;
  char* qr = q + (long)row * ldq + part * 16;
  const int nst = K / 128;
  for (int st = 0; st < nst; ++st) {
    const u32x4 r0 = *(const u32x4*)(xr + (long)st * 256);
    const u32x4 r1 = *(const u32x4*)(xr + (long)st * 256 + 16);
    float f[16];
    {
      float a[8], b[8];
      unpack8<T>(r0, a);
      unpack8<T>(r1, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) { f[e] = a[e]; f[8 + e] = b[e]; }
    }
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) amax = __builtin_fmaxf(amax, __builtin_fabsf(f[e]));
    amax = __builtin_fmaxf(amax, __shfl_xor(amax, 1));
    const unsigned be = mx8_scale_exp(amax);
    const float inv = mx8_inv_scale(be);
    unsigned o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = mx8_pack4(f[4 * d] * inv, f[4 * d + 1] * inv, f[4 * d + 2] * inv, f[4 * d + 3] * inv);
    *(u32x4*)(qr + (long)st * 128) = u32x4{o[0], o[1], o[2], o[3]};
    // the stage's four scale bytes sit in lanes part = 0, 2, 4, 6 of the row: gather them into lane part = 0
    unsigned sc = be;
    sc |= __shfl_down(be, 2) << 8;
    sc |= __shfl_down(be, 4) << 16;
    sc |= __shfl_down(be, 6) << 24;
    if (part == 0) s[(long)st * s_ld + row] = sc;
  }
}

int g_mx_d1 = 12;      // measured best of 8 / 12 / 16 on the UNet's shapes (profiles/r02_mx8_bench.log)
int mx_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}
int g_mx_dbg = 0;      // 64: register-direct epilogue instead of the transposed streaming one (tools / A-B only)

template <typename T, bool CONV = false>
int launch_mx8(GemmP p, hipStream_t s, int mrows) {
  constexpr int lds = 2 * MX_STAGE;
  p.tiles_m = (mrows + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int grid = p.tile_groups * p.tiles_m * p.tiles_n;
  if (grid <= 0) return OMG_OK;
  p.dbg = g_mx_dbg;
#define MX_LAUNCH(D1_, XE_, EF_)                                                                           \
  do {                                                                                                     \
    static bool attr = false;                                                                              \
    if (!attr) { attr = true; (void)hipFuncSetAttribute((const void*)gemm_mx8_kernel<T, D1_, CONV, XE_, EF_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); } \
    OMG_LAUNCH((gemm_mx8_kernel<T, D1_, CONV, XE_, EF_>), dim3(grid), dim3(256), lds, s, p);               \
  } while (0)
  if constexpr (!CONV) {
    // Linear problems with FEW column tiles and a SHORT K loop (N <= 1280, K <= 1280: the attention projections — 11 % of the fp8 step) run the persistent
    // tile walk gemm_mx8_kernel_p; everything else keeps one tile per block.  Measured in situ, twice, interleaved on one box
    // (profiles/r06_by_shape_fp8_{one_tile,persistent}*.txt): 65536 x 1280 x 1280 1228 -> 1340 and 1267 -> 1374 TF/s (+ 8 ... 9 %), 262144 x 640 x 640 667 -> 771 and
    // 685 -> 804 (+ 16 ... 17 %), 32768 x 1280 x 1280 + 1 ... 2 %; but GEGLU 65536 x 10240 x 1280 - 8 ... 10 %, x 3840 x 1280 - 2 ... 5 %, K = 5120 - 3 %, 8192^3 - 8 %: with many
    // column tiles / long K the hand-over's vmcnt(0) in front of the next tile's first barrier waits for the previous epilogue's stores to drain (CDNA4's
    // vmcnt counts stores), which a FRESH workgroup overlaps with its prologue — at fp8 rates a tile is half as long as the 16-bit kernel's and the same
    // store volume weighs twice.  Tools: dbg bit 128 = one tile per block everywhere, bit 512 = the walk everywhere.
    const bool small = p.tiles_n <= 5 && p.K <= 1280 && p.act != OMG_ACT_GEGLU;
    if (!(g_mx_dbg & (64 | 128)) && g_mx_d1 >= 12 && g_mx_d1 < 16 && (small || (g_mx_dbg & 512))) {
      static bool attr_p = false;
      if (!attr_p) { attr_p = true; (void)hipFuncSetAttribute((const void*)gemm_mx8_kernel_p<T, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
      int cus = mx_num_cus() & ~7;
      if (g_mx_dbg & 256) cus = 8;        // tests only: eight blocks, so that a small problem makes every block walk several tiles
      const int pgrid = (grid > cus && cus > 0) ? cus : grid;
      OMG_LAUNCH((gemm_mx8_kernel_p<T, 12>), dim3(pgrid), dim3(256), lds, s, p);
      return omg_check_launch("gemm_mx8_p");
    }
  }
  if (g_mx_dbg & 64) MX_LAUNCH(12, false, 0);
  else if (g_mx_d1 >= 16) MX_LAUNCH(16, true, 0);
  else if (g_mx_d1 >= 12) MX_LAUNCH(12, true, 0);      // (one kernel per epilogue form, as gemm.hip does for v7, was measured 15-30 % SLOWER
                                                       // here: the specialised builds split the K loop into several basic blocks)
  else MX_LAUNCH(8, true, 0);
#undef MX_LAUNCH
  return omg_check_launch("gemm_mx8");
}

}  // namespace

extern "C" void omg_debug_set_mx8_split(int d1) { g_mx_d1 = d1 & 0xff; g_mx_dbg = d1 >> 8; }

extern "C" int omg_quant_mx8(int dtype, const void* x, long ldx, int M, int K, void* q, long ldq, void* scales, int s_ld, void* stream) {
  OMG_REQUIRE(dtype == OMG_F16 || dtype == OMG_BF16, "omg_quant_mx8: dtype");
  OMG_REQUIRE(x && q && scales, "omg_quant_mx8: null operand");
  OMG_REQUIRE(M >= 0 && K > 0 && K % 128 == 0, "omg_quant_mx8: K must be a multiple of 128 (one LDS stage of the MX GEMM)");
  OMG_REQUIRE(ldx % 8 == 0 && ldq % 16 == 0 && s_ld >= M && s_ld % 4 == 0, "omg_quant_mx8: ldx % 8, ldq % 16, s_ld >= M, s_ld % 4");
  if (M == 0) return OMG_OK;
  hipStream_t s = (hipStream_t)stream;
  const int grid = (M + 31) / 32;
  if (dtype == OMG_F16) OMG_LAUNCH((quant_mx8_kernel<f16>), dim3(grid), dim3(256), 0, s, (const char*)x, ldx, M, K, (char*)q, ldq, (unsigned*)scales, s_ld);
  else OMG_LAUNCH((quant_mx8_kernel<bf16>), dim3(grid), dim3(256), 0, s, (const char*)x, ldx, M, K, (char*)q, ldq, (unsigned*)scales, s_ld);
  return omg_check_launch("quant_mx8");
}

extern "C" int omg_gemm_mx8(const omg_gemm_mx8_args* a, void* stream) {
  OMG_REQUIRE(a != nullptr, "omg_gemm_mx8: null args");
  OMG_REQUIRE(a->dtype == OMG_F16 || a->dtype == OMG_BF16, "omg_gemm_mx8: output dtype");
  OMG_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "omg_gemm_mx8: shape");
  if (a->M == 0) return OMG_OK;
  OMG_REQUIRE(a->K % 128 == 0 && a->N % 8 == 0, "omg_gemm_mx8: K must be a multiple of 128, N of 8");
  OMG_REQUIRE(a->lda % 16 == 0 && a->ldw % 16 == 0 && a->ldc % 8 == 0, "omg_gemm_mx8: lda, ldw multiples of 16; ldc of 8");
  const long c_elt = a->c_scale != nullptr ? 1 : 2;
  OMG_REQUIRE(a->A && a->W && a->C && a->a_scale && a->w_scale, "omg_gemm_mx8: null operand");
  OMG_REQUIRE(a->sa_ld >= a->M && a->sa_ld % 4 == 0 && a->sw_ld % 4 == 0, "omg_gemm_mx8: scale row counts");
  OMG_REQUIRE(a->groups >= 1 && (long)a->groups * a->rows_per_group == a->M, "omg_gemm_mx8: M != groups*rows_per_group");
  if (a->act == OMG_ACT_GEGLU) OMG_REQUIRE(a->N % 64 == 0 && !a->residual, "omg_gemm_mx8: GEGLU needs N % 64 == 0, no residual");
  if (a->residual) OMG_REQUIRE(a->ldr % 8 == 0, "omg_gemm_mx8: ldr");
  const long lim = 0x7fff0000L;
  OMG_REQUIRE((long)a->M * a->lda < lim && (long)a->N * a->ldw < lim && (long)a->M * a->ldc * c_elt < lim && (long)a->M * a->ldr * 2 < lim,
              "omg_gemm_mx8: an operand exceeds the 2 GiB buffer-descriptor range");
  GemmP p{};
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = (const char*)a->A; p.lda = a->lda; p.W = (const char*)a->W; p.ldw = a->ldw;
  p.SA = (const char*)a->a_scale; p.sa_ld = a->sa_ld; p.SW = (const char*)a->w_scale; p.sw_ld = a->sw_ld;
  const bool per_group = (a->group_adapter != nullptr) && a->groups > 1;
  p.tile_groups = per_group ? a->groups : 1;
  p.rows_per_group = a->rows_per_group;
  p.group_adapter = (const int*)a->group_adapter;
  p.w_adapter_stride = a->w_adapter_stride; p.sw_adapter_stride = a->sw_adapter_stride;
  OMG_REQUIRE(a->sw_ld >= (a->w_adapter_stride != 0 ? 0 : a->N), "omg_gemm_mx8: sw_ld < N");
  p.bias = (const char*)a->bias; p.residual = (const char*)a->residual; p.ldr = a->ldr;
  p.act = a->act; p.out_scale = a->out_scale; p.C = (char*)a->C; p.ldc = a->ldc;
  if (a->c_scale != nullptr) {
    OMG_REQUIRE(a->act == OMG_ACT_GEGLU && a->N % 256 == 0 && a->ldc % 16 == 0 && a->sc_ld >= a->M && a->sc_ld % 4 == 0 && !(g_mx_dbg & 64),
                "omg_gemm_mx8: MX-fp8 output needs the GEGLU epilogue, N % 256 == 0, ldc % 16 == 0, sc_ld >= M");
    p.QS = (char*)a->c_scale; p.qs_ld = a->sc_ld;
  }
  const int mrows = per_group ? a->rows_per_group : a->M;
  hipStream_t s = (hipStream_t)stream;
  return a->dtype == OMG_F16 ? launch_mx8<f16>(p, s, mrows) : launch_mx8<bf16>(p, s, mrows);
}

extern "C" int omg_conv2d_mx8(const omg_conv2d_mx8_args* a, void* stream) {
  OMG_REQUIRE(a != nullptr, "omg_conv2d_mx8: null args");
  OMG_REQUIRE(a->dtype == OMG_F16 || a->dtype == OMG_BF16, "omg_conv2d_mx8: output dtype");
  OMG_REQUIRE(a->B >= 0 && a->H > 0 && a->W > 0 && a->H < 32768 && a->W < 32768, "omg_conv2d_mx8: shape");
  OMG_REQUIRE(a->Cin > 0 && a->Cin % 128 == 0 && a->Cout > 0 && a->Cout % 8 == 0, "omg_conv2d_mx8: Cin % 128 == 0, Cout % 8 == 0");
  OMG_REQUIRE(a->X && a->x_scale && a->Wq && a->w_scale && a->Y, "omg_conv2d_mx8: null operand");
  OMG_REQUIRE(a->sw_ld >= a->Cout && a->sw_ld % 4 == 0, "omg_conv2d_mx8: sw_ld");
  const long M = (long)a->B * a->H * a->W;
  if (M == 0) return OMG_OK;
  const long K = 9L * a->Cin;
  const long lim = 0x7fff0000L;
  OMG_REQUIRE(M * a->Cin < lim && (long)a->Cout * K < lim && M * a->Cout * 2 < lim && (long)(a->Cin / 128) * M * 4 < lim,
              "omg_conv2d_mx8: an operand exceeds the 2 GiB buffer-descriptor range");
  GemmP p{};
  p.M = (int)M; p.N = a->Cout; p.K = (int)K;
  p.A = (const char*)a->X; p.lda = a->Cin; p.W = (const char*)a->Wq; p.ldw = K;
  p.SA = (const char*)a->x_scale; p.sa_ld = (int)M; p.SW = (const char*)a->w_scale; p.sw_ld = a->sw_ld;
  p.C1 = a->Cin; p.Hin = p.Hout = a->H; p.Win = p.Wout = a->W; p.ksize = 3; p.stride = 1;
  p.tile_groups = 1; p.rows_per_group = a->H * a->W;
  p.bias = (const char*)a->bias;
  p.group_bias = (const char*)a->group_bias; p.ldgb = a->ldgb;
  if (a->group_bias) OMG_REQUIRE(a->ldgb >= a->Cout, "omg_conv2d_mx8: ldgb");
  p.residual = (const char*)a->residual; p.ldr = a->Cout;
  p.act = a->act; p.out_scale = a->out_scale; p.C = (char*)a->Y; p.ldc = a->Cout;
  OMG_REQUIRE(a->act == OMG_ACT_NONE || a->act == OMG_ACT_SILU, "omg_conv2d_mx8: act");
  hipStream_t s = (hipStream_t)stream;
  return a->dtype == OMG_F16 ? launch_mx8<f16, true>(p, s, (int)M) : launch_mx8<bf16, true>(p, s, (int)M);
}
