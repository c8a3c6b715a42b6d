// gemm.hip — MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   C[M,N] = epi( A[M,K] · W[N,K]^T (+ A2[M,K2] · W2[N,K2]^T) )
//
// Both operands are K-contiguous (nn.Linear weights are [N,K]; conv weights are
// repacked to [Cout][ky][kx][Cin] so that K = (tap, cin) and one 64-wide K slice
// never straddles a tap).  Activations are NHWC, so the conv A-operand row for
// output pixel (b,y,x) and tap (dy,dx) is 128 contiguous bytes of the input.
//
// Kernels in this file (chosen per launch by choose_variant; every one produces the same bits for the same problem):
//   v1 (gemm_kernel)      128x128x64 tile, 4 waves of 64x64, two blocks per CU — small launches, narrow outputs, and everything
//                         the large-tile kernels do not take (the LoRA second K-segment, K % 64 != 0, operands >= 2 GiB).
//   v6 (gemm_kernel_v6)   256x{256,128}x64 on eight waves, LDS-DMA issued between MFMA quartets through buffer descriptors; 256x128 in use.
//   v7 (gemm_kernel_v7)   256x256 (and 128x320 for the convs of width 320 k) on FOUR waves, 128x128 per wave, K loop
//                         software-pipelined inside the wave, one barrier per stage; in use for the 128x320 conv tile.
//   v11 (gemm_v11.h)      round 4: v7's 256x256 tile with a table-driven K loop on five rotating half-stage LDS buffers, LDS-DMA spread
//                         over the whole stage — the workhorse (>= 70 % of the time).
// (Round 3 removed v5 — the eight-wave predecessor of v6 — and v8, a persistent form of v7 that spilled: DESIGN.md §5.)
// Common to all: v_mfma_f32_32x32x16, tiles staged global->LDS with LDS-DMA (1 KiB per wave instruction) into a
// double-buffered, XOR-swizzled image — the DMA destination is lane-linear, so the swizzle is applied to the per-lane
// SOURCE chunk and again on the ds_read_b128 (cdna guide §5.4 rule 21); out-of-image taps and the K tail read zeros
// (a global zero page in v1, out-of-range buffer offsets in v6 / v7).  v1 stages the fp32 accumulators through LDS for
// row-coalesced stores; v6 / v7 accumulate the transposed tile and store 16 bytes per lane straight from registers
// (epilogue_direct).  Measurements and the rejected alternatives: DESIGN.md §5.
#include "common.h"
#include "gemm_epilogue.h"

__device__ __attribute__((aligned(256))) unsigned char omg_zero_page[256];

__device__ long long omg_dbg_ts[8192][6];   // tools only (dbg bit 16): per-block timestamps: start, stage 0 landed, loop end, epilogue issued, HW_ID, XCC_ID

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_LD = 68;                     // fp32 epilogue staging row stride (64 + 4 pad)
constexpr int LDS_BYTES_MAIN = 4 * TILE_BYTES;   // A[2] + B[2]
constexpr int LDS_BYTES_EPI = 4 * 64 * STAGE_LD * 4;
constexpr int LDS_BYTES = LDS_BYTES_EPI > LDS_BYTES_MAIN ? LDS_BYTES_EPI : LDS_BYTES_MAIN;


template <bool GLDS>
OMG_DEV void stage16(const char* src, char* lds_wave_base, int lane, u32x4& hold) {
  if constexpr (GLDS) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
  } else {
    hold = *(const u32x4*)src;
  }
}

// non-transposed accumulators: acc[i][j][r] is column col0 + 32 j + l31
template <typename T, int MT, int NT>
OMG_DEV void acc_init_cols(const GemmP& p, f32x16 (&acc)[MT][NT], int l31, int col0, int m0) {
  const bool fold = fold_group_bias(p);
  const long grow = fold ? (long)(m0 / p.rows_per_group) * p.ldgb : 0;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int c = col0 + j * 32 + l31;
    float b = 0.f;
    if (c < p.N) {
      float g = 0.f;
      if (p.bias) b = (float)((const T*)p.bias)[c];
      if (fold) g = (float)((const T*)p.group_bias)[grow + c];
      b = b + g;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
  }
}

template <typename T>
OMG_DEV void gemm_epilogue(const GemmP& p, f32x16 (&acc)[2][2], char* smem, int w, int lane, int m0, int n0, int m_end) {
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = w >> 1, wn = w & 1;
  // ---------------- epilogue: acc -> LDS (fp32) -> coalesced 16-B rows
  __syncthreads();
  float* stage = (float*)smem + w * (64 * STAGE_LD);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        stage[row * STAGE_LD + j * 32 + l31] = acc[i][j][r];
      }
  __syncthreads();

  const int wm0 = m0 + wm * 64;
  const int wn0 = n0 + wn * 64;
  if (p.act == OMG_ACT_GEGLU) {
    // wave tile columns = [32 value | 32 gate]; 4 lanes per row, 16 rows per pass
    const int ocol0 = (wn0 >> 1);
    const int sub = lane & 3, rsub = lane >> 2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 16 + rsub;
      const int gm = wm0 + row;
      const int gc = wn0 + sub * 8;          // packed column of the value half
      if (gm < m_end && gc < p.N) {
        float v[8], g[8];
        const float* sp = stage + row * STAGE_LD + sub * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = sp[e]; g[e] = sp[32 + e]; }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[e] * gelu_f(g[e]) * p.out_scale;
        *(u32x4*)(p.C + ((long)gm * p.ldc + ocol0 + sub * 8) * 2) = pack8<T>(o);
      }
    }
    return;
  }
  {
    const int sub = lane & 7, rsub = lane >> 3;
    const int gc = wn0 + sub * 8;
    const bool gb_rows = p.group_bias != nullptr && !fold_group_bias(p);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + rsub;
      const int gm = wm0 + row;
      if (gm < m_end && gc < p.N) {
        float v[8];
        const float* sp = stage + row * STAGE_LD + sub * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sp[e];
        if (gb_rows) {
          float gb[8];
          const int g = gm / p.rows_per_group;
          unpack8<T>(*(const u32x4*)(p.group_bias + ((long)g * p.ldgb + gc) * 2), gb);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += gb[e];
        }
        if (p.act == OMG_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = silu_fast(v[e]);
        }
        if (p.residual) {
          float rv[8];
          unpack8<T>(*(const u32x4*)(p.residual + ((long)gm * p.ldr + gc) * 2), rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], p.out_scale, rv[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
        }
        *(u32x4*)(p.C + ((long)gm * p.ldc + gc) * 2) = pack8<T>(v);
      }
    }
  }
}

template <typename T, bool CONV, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  // ---- tile mapping: XCD-aware bijective remap, then M-fastest within the chunk
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_per_group = p.tiles_m * p.tiles_n;
  const int grp = bid / tiles_per_group;
  const int t_in = bid - grp * tiles_per_group;
  // grouped ordering: 8 M-tiles x all N-tiles per group, M fastest inside the group, so that the ~32 consecutive
  // tiles an XCD receives touch ~8 A panels + <= 4..5 W panels instead of 32 + 1 (tall-skinny GEMMs re-read A per N tile)
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
  const int m0 = m_base + tm * BM;
  const int n0 = tn * BN;

  int adapter = 0;
  if (p.group_adapter != nullptr) adapter = p.group_adapter[grp];
  const bool seg2 = (p.K2 > 0) && (adapter >= 0);
  if (p.w_adapter_stride != 0 && adapter < 0) return;   // LoRA-down GEMM for a sample without adapter
  const char* Wp = p.W + (p.w_adapter_stride != 0 ? (long)adapter * p.w_adapter_stride * 2 : 0);
  const char* W2p = p.W2 + (seg2 ? (long)adapter * p.w2_adapter_stride * 2 : 0);
  const int a2off = (p.a2_col_block > 0) ? (n0 / p.a2_col_block) * p.K2 : 0;

  const int nk1 = (p.K + BK - 1) / BK;
  const int nk2 = seg2 ? (p.K2 + BK - 1) / BK : 0;
  const int nk = nk1 + nk2;

  // ---- per-lane staging coordinates: 4 A rows + 4 W rows per k-tile
  const int prow = lane >> 3;          // row within the 8-row DMA piece
  const int ppos = lane & 7;           // 16-B position within the 128-B row
  int arow[4];                         // global A row (clamped)
  int wrow[4];                         // global W row (clamped)
  int cb[4], cy[4], cx[4];             // conv: decoded output pixel
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = w * 32 + i * 8 + prow;
    int gm = m0 + r; if (gm > m_end - 1) gm = m_end - 1;
    int gn = n0 + r; if (gn > p.N - 1) gn = p.N - 1;
    arow[i] = gm; wrow[i] = gn;
    if constexpr (CONV) {
      const int hw = p.Hout * p.Wout;
      const int b = gm / hw; const int rem = gm - b * hw;
      cb[i] = b; cy[i] = rem / p.Wout; cx[i] = rem - cy[i] * p.Wout;
    }
  }
  const int Ctot = p.C1 + p.C2;
  const int cpt = CONV ? Ctot / BK : 1;   // k-tiles per tap
  const int pad = CONV ? (p.ksize == 3 ? 1 : 0) : 0;
  const char* zero = (const char*)omg_zero_page;

  u32x4 hold[8];
  char* ldsA = smem;
  char* ldsB = smem + 2 * TILE_BYTES;

  auto issue = [&](int kt, int buf) {
    const bool s2 = kt >= nk1;
    const int k0 = (s2 ? kt - nk1 : kt) * BK;
    const int Kseg = s2 ? p.K2 : p.K;
    // conv tap decode (wave-uniform)
    int dy = 0, dx = 0, c0 = k0;
    const char* xsrc = p.A; int xC = p.C1;
    if constexpr (CONV) {
      const int tap = kt / cpt; const int cc = kt - tap * cpt;
      dy = tap / p.ksize; dx = tap - dy * p.ksize;
      c0 = cc * BK;
      if (c0 >= p.C1) { xsrc = p.X2; xC = p.C2; c0 -= p.C1; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = w * 32 + i * 8 + prow;
      const int c = ppos ^ ((r >> 1) & 7);                    // source chunk for this LDS position
      const bool kvalid = (k0 + c * 8) < Kseg;
      // A operand
      const char* asrc;
      if constexpr (CONV) {
        int iy = cy[i] * p.stride + dy - pad;
        int ix = cx[i] * p.stride + dx - pad;
        const int Hl = p.upsample ? p.Hin * 2 : p.Hin;
        const int Wl = p.upsample ? p.Win * 2 : p.Win;
        const bool ok = (iy >= 0) && (iy < Hl) && (ix >= 0) && (ix < Wl);
        if (p.upsample) { iy >>= 1; ix >>= 1; }
        const long pix = ((long)cb[i] * p.Hin + iy) * p.Win + ix;
        asrc = ok ? xsrc + (pix * xC + c0 + c * 8) * 2 : zero;
      } else {
        if (!s2) asrc = kvalid ? p.A + ((long)arow[i] * p.lda + k0 + c * 8) * 2 : zero;
        else     asrc = kvalid ? p.A2 + ((long)arow[i] * p.lda2 + a2off + k0 + c * 8) * 2 : zero;
      }
      stage16<GLDS>(asrc, ldsA + buf * TILE_BYTES + (w * 32 + i * 8) * 128, lane, hold[i]);
      // W operand
      const char* wsrc;
      if (!s2) wsrc = kvalid ? Wp + ((long)wrow[i] * p.ldw + (CONV ? kt * BK : k0) + c * 8) * 2 : zero;
      else     wsrc = kvalid ? W2p + ((long)wrow[i] * p.ldw2 + k0 + c * 8) * 2 : zero;
      stage16<GLDS>(wsrc, ldsB + buf * TILE_BYTES + (w * 32 + i * 8) * 128, lane, hold[4 + i]);
    }
  };
  auto commit = [&](int buf) {   // register-staged fallback: write the held chunks
    if constexpr (!GLDS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = (w * 32 + i * 8) * 128 + lane * 16;
        *(u32x4*)(ldsA + buf * TILE_BYTES + off) = hold[i];
        *(u32x4*)(ldsB + buf * TILE_BYTES + off) = hold[4 + i];
      }
    }
  };

  const int wm = w >> 1, wn = w & 1;
  f32x16 acc[2][2];
  acc_init_cols<T, 2, 2>(p, acc, lane & 31, n0 + wn * 64, m0);

  using V8 = typename Vec<T>::v8;

  issue(0, 0);
  commit(0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
    const char* a_base = ldsA + buf * TILE_BYTES;
    const char* b_base = ldsB + buf * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      V8 af[2], bf[2];
      const int kc = ks * 2 + hi;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31;
        af[i] = *(const V8*)(a_base + ra * 128 + ((kc ^ ((ra >> 1) & 7)) << 4));
        const int rb = wn * 64 + i * 32 + l31;
        bf[i] = *(const V8*)(b_base + rb * 128 + ((kc ^ ((rb >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Vec<T>::mfma32(af[i], bf[j], acc[i][j]);
    }
    if constexpr (!GLDS) { if (kt + 1 < nk) commit(buf ^ 1); }
  }

  gemm_epilogue<T>(p, acc, smem, w, lane, m0, n0, m_end);
}


// ------------------------------------------------------------------------------------------------
// Large-tile kernels (v6, v7): 256-row block tiles, LDS stages of BK = 64 filled by LDS-DMA, raw s_barrier + explicit
// s_waitcnt instead of __syncthreads.  LDS image per stage: rows of 128 B (8 chunks of 16 B), chunk ^= (row >> 1) & 7,
// applied on the DMA source side and on the fragment read.
// conv A-operand byte offset of one output pixel's tap inside the (logical, possibly 2x-upsampled) input image;
// out-of-image taps get an offset beyond the buffer's num_records, which the buffer load turns into zeros
OMG_DEV int conv_voff(int b, int oy, int ox, int ch_bytes, int stride, int dy, int dx, int Hl, int Wl, int ups, int Hin, int Win,
                      int pix_bytes, int c0_bytes) {
  // branch-free on purpose: with `&&` hipcc put two exec-mask regions (s_and_saveexec + s_cbranch_execz) round the address of
  // EVERY A-operand DMA inside the MFMA loop; one unsigned compare per axis and a select instead
  int iy = oy * stride + dy, ix = ox * stride + dx;
  const bool ok = ((unsigned)iy < (unsigned)Hl) & ((unsigned)ix < (unsigned)Wl);
  iy >>= ups; ix >>= ups;
  const int pix = (b * Hin + iy) * Win + ix;
  const int off = pix * pix_bytes + c0_bytes + ch_bytes;
  return ok ? off : 0x7ffffff0;
}

// ------------------------------------------------------------------------------------------------
// v6: v5's structure (256 x {256,128} x 64 tile, double buffer, two-group stagger) with the stage's LDS-DMA instructions
// spread one by one between pairs/quartets of MFMAs instead of issued as a burst, and buffer-descriptor addressing
// (SGPR descriptor + 32-bit per-lane offset + SGPR stage offset: no per-stage 64-bit address VALU; out-of-image conv taps
// and the K tail are simply out-of-range offsets, which the buffer unit returns as zeros).  Phase timing of v5
// (tools/gemm_phases.py): a wave spent ~100-170 cycles BLOCKED on each global_load_lds when the waves of a group issued
// their 8 together (the texture addresser queues), ~1100 cycles per stage against 1024 cycles of MFMA.
//   MT = 4: 256x256, waves 2(M) x 4(N), wave tile 128x64, 4+4 DMA / wave / stage, one DMA per 4 MFMAs
//   MT = 2: 256x128, waves 4(M) x 2(N), wave tile  64x64, 4+2 DMA / wave / stage, one DMA per 2 MFMAs (3 of 4 slots)
// LoRA second K-segment is not handled here (v5/v1 do that).
template <typename T, bool CONV, int MT>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v6(GemmP p) {
  constexpr int BM_ = 256, BN_ = MT == 4 ? 256 : 128, BKc = 64, NW = 8, NT = 2;
  constexpr int WN_ = MT == 4 ? 4 : 2;
  constexpr int BI = BN_ / 8 / NW;                 // W DMA instructions per wave per stage (4 or 2)
  constexpr int ND = 4 + BI;                       // all DMA instructions per wave per stage (8 or 6)
  constexpr int HB = ND / 2;                       // per MFMA block (4 or 3)
  constexpr int A_BYTES = BM_ * BKc * 2;
  constexpr int STAGE_BYTES = (BM_ + BN_) * BKc * 2;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;

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
  const int m0 = m_base + tm * BM_;
  const int n0 = tn * BN_;
  int adapter = 0;
  if (p.group_adapter != nullptr) adapter = p.group_adapter[grp];
  if (p.w_adapter_stride != 0 && adapter < 0) return;
  const char* Wp = p.W + (p.w_adapter_stride != 0 ? (long)adapter * p.w_adapter_stride * 2 : 0);
  const int nk = (p.K + BKc - 1) / BKc;

  const int Ctot = p.C1 + p.C2;
  // num_records: exact byte sizes, so that any offset at or beyond them reads as zero
  const long a_bytes = CONV ? (long)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.C1 * 2 : ((long)(p.M - 1) * p.lda + p.K) * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(a_bytes < 0x7fffff00 ? a_bytes : 0x7fffff00), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(CONV && p.X2 ? p.X2 : p.A), 0,
      CONV ? (int)((long)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.C2 * 2) : 0, 0x00020000);
  const long w_bytes = ((long)(p.N - 1) * p.ldw + p.K) * 2;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, (int)w_bytes, 0x00020000);

  const int prow = lane >> 3, ppos = lane & 7;
  int voffA[4], voffW[BI], ldoA[4], ldoW[BI];
  int cb[4], cy[4], cx[4], cch[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (w + i * NW) * 8 + prow;
    const int c = ppos ^ ((r >> 1) & 7);
    int gm = m0 + r; if (gm > m_end - 1) gm = m_end - 1;
    ldoA[i] = (w + i * NW) * 1024;
    if constexpr (CONV) {
      const int hw = p.Hout * p.Wout;
      const int b = gm / hw; const int rem = gm - b * hw;
      cb[i] = b; cy[i] = rem / p.Wout; cx[i] = rem - cy[i] * p.Wout; cch[i] = c * 16;
      voffA[i] = 0;
    } else {
      voffA[i] = (int)(((long)gm * p.lda + c * 8) * 2);
    }
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int r = (w + i * NW) * 8 + prow;
    const int c = ppos ^ ((r >> 1) & 7);
    int gn = n0 + r; if (gn > p.N - 1) gn = p.N - 1;
    voffW[i] = (int)(((long)gn * p.ldw + c * 8) * 2);
    ldoW[i] = A_BYTES + (w + i * NW) * 1024;
  }

  const int wm = w / WN_, wn = w % WN_;
  f32x16 acc[MT][NT];
  const bool gb_epi = acc_init_bias<T, MT, NT>(p, acc, lane, m0, n0 + wn * 64);
  using V8 = typename Vec<T>::v8;
  int aro[MT][4], bro[NT][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int kc = ks * 2 + hi;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int ra = wm * (MT * 32) + i * 32 + l31;
      aro[i][ks] = ra * 128 + ((kc ^ ((ra >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int rb = wn * 64 + j * 32 + l31;
      bro[j][ks] = A_BYTES + rb * 128 + ((kc ^ ((rb >> 1) & 7)) << 4);
    }
  }

  // per-stage uniform state of the NEXT stage's loads (set by `prep`)
  int koff = 0;                         // byte offset of the stage inside a W row (and an A row of the plain GEMM)
  int tap_dy = 0, tap_dx = 0, c0b = 0, xCb = 0;   // conv: tap, byte offset of the channel slice, bytes per pixel
  bool x2 = false;
  const int cpt = CONV ? Ctot / BKc : 1;
  const int pad = CONV ? (p.ksize == 3 ? 1 : 0) : 0;
  const int Hl = CONV ? (p.upsample ? p.Hin * 2 : p.Hin) : 0;
  const int Wl = CONV ? (p.upsample ? p.Win * 2 : p.Win) : 0;
#define OMG_PREP(kt_)                                                                                      \
  do {                                                                                                     \
    koff = (kt_) * (BKc * 2);                                                                              \
    if constexpr (CONV) {                                                                                  \
      const int tap = (kt_) / cpt; const int cc = (kt_) - tap * cpt;                                       \
      tap_dy = tap / p.ksize - pad; tap_dx = tap - (tap / p.ksize) * p.ksize - pad;                        \
      int c0 = cc * BKc;                                                                                   \
      x2 = c0 >= p.C1;                                                                                     \
      if (x2) c0 -= p.C1;                                                                                  \
      c0b = c0 * 2; xCb = (x2 ? p.C2 : p.C1) * 2;                                                          \
    }                                                                                                      \
  } while (0)
  // byte offset of A row block i's 16 bytes for the prepared stage (i is a compile-time constant at every use)
#define OMG_AVO(i_) (CONV ? conv_voff(cb[i_], cy[i_], cx[i_], cch[i_], p.stride, tap_dy, tap_dx, Hl, Wl, p.upsample, p.Hin, p.Win, xCb, c0b) : voffA[i_])

  const bool late = w >= 4;
  V8 af[2][MT], bf[2][NT];
  // DMA instruction d of the stage: d < 4 -> A row block d, else W row block d-4
#define OMG_ISA(d_) ((d_) < 4)
#define OMG_AI(d_) ((d_) < 4 ? (d_) : 0)
#define OMG_WI(d_) ((d_) >= 4 ? (d_) - 4 : 0)
#define OMG_VO(d_) (OMG_ISA(d_) ? OMG_AVO(OMG_AI(d_)) : voffW[OMG_WI(d_)])
#define OMG_LO(d_) (OMG_ISA(d_) ? ldoA[OMG_AI(d_)] : ldoW[OMG_WI(d_)])
#define OMG_RS(d_) (OMG_ISA(d_) ? ((CONV && x2) ? rsA2 : rsA) : rsW)
#define OMG_SO(d_) ((OMG_ISA(d_) && CONV) ? 0 : koff)
#define OMG_DMA(d_, nb_)                                                                                     \
  dma16(OMG_RS(d_), (nb_) + OMG_LO(d_), OMG_VO(d_), OMG_SO(d_))
  // one instruction shared by both groups: the early group issues d = q, the late group d = HB + q
#define OMG_DMA2(q_, nb_)                                                                                    \
  dma16(late ? OMG_RS(HB + (q_)) : OMG_RS(q_), (nb_) + (late ? OMG_LO(HB + (q_)) : OMG_LO(q_)),              \
        late ? OMG_VO(HB + (q_)) : OMG_VO(q_), late ? OMG_SO(HB + (q_)) : OMG_SO(q_))
  // MFMA slot q (0..3) of a half: MT*NT*2/4 MFMAs
#define OMG_Q(q_)                                                                                            \
  do {                                                                                                       \
    if constexpr (MT == 4) {                                                                                 \
      _Pragma("unroll") for (int i = ((q_) & 1) * 2; i < ((q_) & 1) * 2 + 2; ++i)                            \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                       \
          acc[i][j] = Vec<T>::mfma32(bf[(q_) >> 1][j], af[(q_) >> 1][i], acc[i][j]);                         \
    } else {                                                                                                 \
      _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                         \
        acc[(q_) & 1][j] = Vec<T>::mfma32(bf[(q_) >> 1][j], af[(q_) >> 1][(q_) & 1], acc[(q_) & 1][j]);      \
    }                                                                                                        \
  } while (0)
#define OMG_RD(sb_, half_)                                                                                   \
  do {                                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                       \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) bf[ks][j] = *(const V8*)((sb_) + bro[j][(half_) * 2 + ks]); \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) af[ks][i] = *(const V8*)((sb_) + aro[i][(half_) * 2 + ks]); \
    }                                                                                                        \
  } while (0)

  // prologue: stage 0 as a burst
  OMG_PREP(0);
  OMG_DMA(0, smem); OMG_DMA(1, smem); OMG_DMA(2, smem); OMG_DMA(3, smem);
  OMG_DMA(4, smem); OMG_DMA(5, smem);
  if constexpr (ND == 8) { OMG_DMA(6, smem); OMG_DMA(7, smem); }

  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const char* sb = smem + (kt & 1) * STAGE_BYTES;
    char* nb = smem + ((kt + 1) & 1) * STAGE_BYTES;
    const bool nxt = kt + 1 < nk;
    OMG_PREP(kt + 1);
    if (late) {      // block 0 of the late group: previous stage's second half + DMA 0..HB-1
      if (kt > 0) {
        OMG_Q(0); if (nxt) OMG_DMA(0, nb);
        OMG_Q(1); if (nxt) OMG_DMA(1, nb);
        OMG_Q(2); if (nxt) OMG_DMA(2, nb);
        OMG_Q(3); if constexpr (HB == 4) { if (nxt) OMG_DMA(3, nb); }
      } else if (nxt) {
        OMG_DMA(0, nb); OMG_DMA(1, nb); OMG_DMA(2, nb);
        if constexpr (HB == 4) OMG_DMA(3, nb);
      }
    }
    OMG_RD(sb, 0);
    // first half of stage kt: block 0 of the early group (DMA q) == block 1 of the late group (DMA HB+q)
    OMG_Q(0); if (nxt) OMG_DMA2(0, nb);
    OMG_Q(1); if (nxt) OMG_DMA2(1, nb);
    OMG_Q(2); if (nxt) OMG_DMA2(2, nb);
    OMG_Q(3); if constexpr (HB == 4) { if (nxt) OMG_DMA2(3, nb); }
    OMG_RD(sb, 1);
    if (!late) {
      OMG_Q(0); if (nxt) OMG_DMA(HB + 0, nb);
      OMG_Q(1); if (nxt) OMG_DMA(HB + 1, nb);
      OMG_Q(2); if (nxt) OMG_DMA(HB + 2, nb);
      OMG_Q(3); if constexpr (HB == 4) { if (nxt) OMG_DMA(HB + 3, nb); }
    }
  }
  if (late) { OMG_Q(0); OMG_Q(1); OMG_Q(2); OMG_Q(3); }
#undef OMG_DMA
#undef OMG_DMA2
#undef OMG_Q
#undef OMG_RD
#undef OMG_ISA
#undef OMG_AI
#undef OMG_WI
#undef OMG_VO
#undef OMG_LO
#undef OMG_RS
#undef OMG_SO
#undef OMG_PREP
#undef OMG_AVO
  if (p.dbg & 32) {   // tools only: time the tile without its epilogue (the sum keeps the MFMAs alive)
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (sacc == 1.2345e-30f) p.C[0] = 1;
    return;
  }
  epilogue_direct<T, MT, NT>(p, acc, lane, m0 + wm * (MT * 32), n0 + wn * 64, m_end, gb_epi);
}

// ------------------------------------------------------------------------------------------------
// v7: 256x256x64 tile on FOUR waves (2 x 2), each owning 128x128 of the output (16 accumulator tiles = 256 registers;
// one wave per SIMD with the whole 512-entry register file).  Per 16-wide k-step a wave reads 4 A + 4 B fragments for 16
// MFMAs (0.5 ds_read_b128 per MFMA; the 8-wave kernels need 0.75) so the CU's LDS read traffic per stage drops from
// 192 KB to 128 KB: with the 64 KB of DMA writes, LDS time goes from ~100 % of the MFMA time of a stage to ~75 %.
// With a single wave per SIMD nothing hides a stall, so the K loop is software-pipelined inside the wave:
//   * fragments of k-step s+1 are read while the MFMAs of k-step s issue (two fragment register sets);
//   * the stage's 16 LDS-DMA instructions are spread over two k-steps, one per two MFMAs;
//   * one block barrier per stage, placed before the LAST k-step: by then every wave has issued (and waited for) its
//     reads of the current buffer, and the next stage has had two k-steps (~1000 cycles) to land, so right after the
//     barrier the wave starts reading the next stage's first fragments AND refilling the current buffer with stage kt+2
//     while the last k-step's MFMAs run.
// ABL (tools only, results are wrong): 1 = no LDS-DMA in the K loop, 2 = no fragment reads in the K loop, 4 = no wait / barrier
// (MT, NT) = (4, 4): 256x256 tile.  (2, 5): 128x320 tile, wave tile 64x160 — for the layers whose width is 320 * k and
// pads badly to 256 (N = 320: 512, N = 640: 768) or leaves a ragged last round (N = 1280 at M = 32768).
template <typename T, bool CONV, int ABL = 0, int MT = 4, int NT = 4, bool XE = false, int EF = 0>
__global__ __launch_bounds__(256, 1) void gemm_kernel_v7(GemmP p) {
  constexpr int BM_ = MT * 64, BN_ = NT * 64, BKc = 64;
  constexpr int AB = MT * 2, WB = NT * 2;          // A / W row blocks (8 rows each) per wave per stage
  constexpr int NMM = MT * NT;                     // MFMAs per k-step
  constexpr int SLOTS = NMM / 2;                   // pairs of MFMAs per k-step
  constexpr int NRD = MT + NT;                     // fragment reads per k-step
  static_assert(NMM % 2 == 0, "even number of MFMAs per k-step");
  constexpr int A_BYTES = BM_ * BKc * 2;
  constexpr int STAGE_BYTES = (BM_ + BN_) * BKc * 2;

  const bool ts_on = (p.dbg & 16) && blockIdx.x < 8192 && threadIdx.x == 0;
  long long ts0 = 0, ts1 = 0, ts2 = 0;
  // tools only (dbg bits 16..23 = S in units of 0.25 us): de-phase the CUs.  Every block of the launch's first round waits a
  // pseudo-random share of S before it starts; a CU takes its next block when the previous one ends, so the offsets persist and
  // the per-tile bursts (first-stage fetch, C-tile stores) of the 256 CUs no longer hit the memory system at the same instant.
  if (((p.dbg >> 16) & 0xff) != 0 && blockIdx.x < 256) {
    const long long wait = (long long)((blockIdx.x * 97) & 255) * ((p.dbg >> 16) & 0xff) * 25 / 256;      // 10 ns ticks
    const long long t_end = __builtin_amdgcn_s_memrealtime() + wait;
    while (__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(2);
  }
  if (p.dbg & 16) ts0 = __builtin_amdgcn_s_memrealtime();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;

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
  const int m0 = m_base + tm * BM_;
  const int n0 = tn * BN_;
  int adapter = 0;
  if (p.group_adapter != nullptr) adapter = p.group_adapter[grp];
  if (p.w_adapter_stride != 0 && adapter < 0) return;
  const char* Wp = p.W + (p.w_adapter_stride != 0 ? (long)adapter * p.w_adapter_stride * 2 : 0);
  const int nk = (p.K + BKc - 1) / BKc;

  const int Ctot = p.C1 + p.C2;
  const long a_bytes = CONV ? (long)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.C1 * 2 : ((long)(p.M - 1) * p.lda + p.K) * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(a_bytes < 0x7fffff00 ? a_bytes : 0x7fffff00), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(CONV && p.X2 ? p.X2 : p.A), 0,
      CONV ? (int)((long)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.C2 * 2) : 0, 0x00020000);
  const long w_bytes = ((long)(p.N - 1) * p.ldw + p.K) * 2;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, (int)w_bytes, 0x00020000);

  // DMA: one instruction moves 8 rows x 128 B; wave w owns row blocks w, w+4, ..., w+28 of A and of W
  const int prow = lane >> 3, ppos = lane & 7;
  int voffA[AB], voffW[WB];
  int cb[AB], cy[AB], cx[AB];
  const int dchunk = (ppos ^ ((w & 1) * 4 + (prow >> 1))) * 16;   // ((row >> 1) & 7) with row = (w + 4i) * 8 + prow
#pragma unroll
  for (int i = 0; i < AB; ++i) {
    const int r = (w + i * 4) * 8 + prow;
    int gm = m0 + r; if (gm > m_end - 1) gm = m_end - 1;
    if constexpr (CONV) {
      const int hw = p.Hout * p.Wout;
      const int b = gm / hw; const int rem = gm - b * hw;
      cb[i] = b; cy[i] = rem / p.Wout; cx[i] = rem - cy[i] * p.Wout;
      voffA[i] = 0;
    } else {
      cb[i] = cy[i] = cx[i] = 0;
      voffA[i] = (int)((long)gm * p.lda * 2) + dchunk;
    }
  }
#pragma unroll
  for (int i = 0; i < WB; ++i) {
    const int r = (w + i * 4) * 8 + prow;
    int gn = n0 + r; if (gn > p.N - 1) gn = p.N - 1;
    voffW[i] = (int)((long)gn * p.ldw * 2) + dchunk;
  }
  const int ldo = w * 1024;

  const int wm = w >> 1, wn = w & 1;
  f32x16 acc[MT][NT];
  using V8 = typename Vec<T>::v8;
  // fragment i of k-step ks sits at aoff[ks] + i * 4096 (32 rows further: same swizzle term)
  int aoff[4], boff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int sw = ((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4;
    aoff[ks] = (wm * (MT * 32) + l31) * 128 + sw;               // (MT*32) >> 1 and (NT*32) >> 1 are multiples of 8: same swizzle term
    boff[ks] = A_BYTES + (wn * (NT * 32) + l31) * 128 + sw;
  }

  int koff = 0;
  int tap_dy = 0, tap_dx = 0, c0b = 0, xCb = 0;
  bool x2 = false;
  const int cpt = CONV ? Ctot / BKc : 1;
  const int pad = CONV ? (p.ksize == 3 ? 1 : 0) : 0;
  const int Hl = CONV ? (p.upsample ? p.Hin * 2 : p.Hin) : 0;
  const int Wl = CONV ? (p.upsample ? p.Win * 2 : p.Win) : 0;
#define OMG_PREP(kt_)                                                                                      \
  do {                                                                                                     \
    koff = (kt_) * (BKc * 2);                                                                              \
    if constexpr (CONV) {                                                                                  \
      const int tap = (kt_) / cpt; const int cc = (kt_) - tap * cpt;                                       \
      tap_dy = tap / p.ksize - pad; tap_dx = tap - (tap / p.ksize) * p.ksize - pad;                        \
      int c0 = cc * BKc;                                                                                   \
      x2 = c0 >= p.C1;                                                                                     \
      if (x2) c0 -= p.C1;                                                                                  \
      c0b = c0 * 2; xCb = (x2 ? p.C2 : p.C1) * 2;                                                          \
    }                                                                                                      \
  } while (0)
  // DMA instruction d of the prepared stage: d < AB -> A row block w + 4d, else W row block w + 4(d-AB); d < AB + WB
#define OMG_DMA(d_, nb_)                                                                                   \
  do {                                                                                                     \
    if ((d_) < AB) {                                                                                       \
      const int i_ = (d_) < AB ? (d_) : 0;                                                                 \
      if (CONV) dma16(x2 ? rsA2 : rsA, (nb_) + ldo + i_ * 4096,                                            \
                      conv_voff(cb[i_], cy[i_], cx[i_], dchunk, p.stride, tap_dy, tap_dx, Hl, Wl, p.upsample, p.Hin, p.Win, xCb, c0b), 0); \
      else dma16(rsA, (nb_) + ldo + i_ * 4096, voffA[i_], koff);                                           \
    } else {                                                                                               \
      const int i_ = (d_) - AB < WB ? (d_) - AB : 0;                                                       \
      dma16(rsW, (nb_) + A_BYTES + ldo + i_ * 4096, voffW[i_], koff);                                      \
    }                                                                                                      \
  } while (0)
#define OMG_DMAN(first_, n_, nb_)                                                                          \
  do { _Pragma("unroll") for (int d_ = 0; d_ < (n_); ++d_) OMG_DMA((first_) + d_, nb_); } while (0)
#define OMG_RD(f_, sb_, ks_)                                                                               \
  do {                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) bf[f_][j] = *(const V8*)((sb_) + boff[ks_] + j * 4096); \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) af[f_][i] = *(const V8*)((sb_) + aoff[ks_] + i * 4096); \
  } while (0)
#define OMG_MM(f_)                                                                                         \
  do {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                         \
      _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                       \
        acc[i][j] = Vec<T>::mfma32(af[f_][i], bf[f_][j], acc[i][j]);                                       \
  } while (0)
  // one fragment read (r_ < 4: W fragment r_, else A fragment r_-4) / one MFMA (n_ = 4 i + j) / one k-step written in
  // the issue order wanted: MFMA, read, MFMA, DMA, ... — the compiler keeps LDS-DMA instructions where the source has them
#define OMG_RD1(f_, sb_, ks_, r_)                                                                          \
  do {                                                                                                     \
    /* read order = order of first use by the MFMAs (n = NT i + j): W0, A0, W1 .. W(NT-1), A1 .. A(MT-1) */ \
    const bool isA_ = (r_) == 1 || (r_) > NT;                                                              \
    const int idx_ = (r_) <= 1 ? 0 : (r_) <= NT ? (r_) - 1 : (r_) - NT;                                    \
    if (!isA_) bf[f_][idx_] = *(const V8*)((sb_) + boff[ks_] + idx_ * 4096);                               \
    else af[f_][idx_] = *(const V8*)((sb_) + aoff[ks_] + idx_ * 4096);                                     \
  } while (0)
#define OMG_MM1(f_, n_) acc[(n_) / NT][(n_) % NT] = Vec<T>::mfma32(bf[f_][(n_) % NT], af[f_][(n_) / NT], acc[(n_) / NT][(n_) % NT])
  // k-step computing with fragment set f_ while set 1-f_ is refilled from (rb_, rks_) (RD_ = 1: one read per slot,
  // 2: two per slot in the first four slots, so that they are back before the loop's first MFMA needs them), and 8 DMAs
  // (instructions d0_..d0_+7 of the prepared stage into db_) are issued when DMA_
  // slot s_ of a k-step: MFMA 2s, reads, MFMA 2s+1, DMAs.  RD_ = 1: reads spread over the slots (ceil(NRD / SLOTS) per
  // slot), 2: two per slot from the first slot on.  DMA_: instructions d0_ .. d0_ + dn_ - 1, ceil(dn_ / SLOTS) per slot.
#define OMG_KSTEP(f_, RD_, rb_, rks_, DMA_, d0_, dn_, db_)                                                 \
  do {                                                                                                     \
    constexpr int RPS_ = (RD_) == 2 ? 2 : (NRD + SLOTS - 1) / SLOTS;                                       \
    constexpr int DPS_ = ((dn_) + SLOTS - 1) / SLOTS;                                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < SLOTS; ++s_) {                                                 \
      OMG_MM1(f_, 2 * s_);                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if ((RD_) != 0 && !(ABL & 2)) {                                                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < RPS_; ++q_)                                                \
          if (s_ * RPS_ + q_ < NRD) OMG_RD1(1 - (f_), rb_, rks_, s_ * RPS_ + q_);                          \
      }                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      OMG_MM1(f_, 2 * s_ + 1);                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if ((DMA_) && !(ABL & 1)) {                                                                          \
        _Pragma("unroll") for (int q_ = 0; q_ < DPS_; ++q_)                                                \
          if (s_ * DPS_ + q_ < (dn_)) OMG_DMA((d0_) + s_ * DPS_ + q_, db_);                                \
      }                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
    }                                                                                                      \
  } while (0)

  V8 af[2][MT], bf[2][NT];
  // prologue: stage 0 completely, the A half of stage 1, the first fragments
  OMG_PREP(0);
  OMG_DMAN(0, AB + WB, smem);
  // the bias loads ride behind the first stage's DMA (their latency used to sit in front of it: ~1 us of every tile)
  const bool gb_epi = acc_init_bias<T, MT, NT>(p, acc, lane, m0, n0 + wn * (NT * 32));
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (p.dbg & 16) ts1 = __builtin_amdgcn_s_memrealtime();
  OMG_PREP(1);
  if (nk > 1) OMG_DMAN(0, AB, smem + STAGE_BYTES);
  OMG_RD(0, smem, 0);

  // One stage.  HAS1_/HAS2_ (stage kt+1 / kt+2 exist) are literal so that the steady-state body is one basic block —
  // the interleave requests below only work inside a block; the last two stages are peeled copies.
#define OMG_STAGE(HAS1_, HAS2_)                                                                            \
  do {                                                                                                     \
    const char* cur = smem + (kt & 1) * STAGE_BYTES;                                                       \
    char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;                                                       \
    /* k-step 0 (+ the W half of stage kt+1), k-steps 1, 2 */                                              \
    OMG_KSTEP(0, 1, cur, 1, HAS1_, AB, WB, nxt);                                                            \
    OMG_KSTEP(1, 1, cur, 2, false, 0, 0, nxt);                                                             \
    OMG_KSTEP(0, 1, cur, 3, false, 0, 0, nxt);                                                             \
    /* stage kt+1 has landed (this wave's part), this wave's reads of `cur` are complete: join the block */ \
    if (!(ABL & 4)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } \
    /* */                                                                                                  \
    /* k-step 3 (+ first fragments of stage kt+1, + the A half of stage kt+2 into the buffer just released) */ \
    if (HAS2_) OMG_PREP(kt + 2);                                                                           \
    OMG_KSTEP(1, (HAS1_) ? 2 : 0, nxt, 0, HAS2_, 0, AB, (char*)cur);                                                     \
  } while (0)
  int kt = 0;
  for (; kt < nk - 2; ++kt) OMG_STAGE(true, true);
  if (kt < nk - 1) { OMG_STAGE(true, false); ++kt; }
  OMG_STAGE(false, false);
#undef OMG_STAGE
#undef OMG_PREP
#undef OMG_DMA
#undef OMG_DMAN
#undef OMG_RD
#undef OMG_MM
#undef OMG_RD1
#undef OMG_MM1
#undef OMG_KSTEP
  if (p.dbg & 32) {   // tools only: time the tile without its epilogue (the sum keeps the MFMAs alive)
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (sacc == 1.2345e-30f) p.C[0] = 1;
    return;
  }
  if (p.dbg & 16) ts2 = __builtin_amdgcn_s_memrealtime();
  // XE: 8 KB per wave behind the two stage buffers (other waves may still be reading the last stage)
  if constexpr (EF == 2) {
    // residual: staged through the stage buffers, which nobody reads after the loop's last barrier (this wave's 32 KB slice)
    res_stage_dma(p, smem + w * 32768, lane, m0 + wm * 128, n0 + wn * 128, m_end);
    epilogue_direct<T, MT, NT, XE, EF>(p, acc, lane, m0 + wm * (MT * 32), n0 + wn * (NT * 32), m_end, gb_epi, smem + 2 * STAGE_BYTES + w * 8192,
                                       smem + w * 32768);
  } else
  epilogue_direct<T, MT, NT, XE, EF>(p, acc, lane, m0 + wm * (MT * 32), n0 + wn * (NT * 32), m_end, gb_epi, smem + 2 * STAGE_BYTES + w * 8192);
  if (ts_on) {
    long long* t = omg_dbg_ts[blockIdx.x];
    t[0] = ts0; t[1] = ts1; t[2] = ts2; t[3] = __builtin_amdgcn_s_memrealtime();
    t[4] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all 32 bits
    t[5] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
  }
}

bool g_use_glds = true;
int g_dbg = 0;
int g_variant = 0;   // 0 = heuristic; 1 = 128x128 v1; 13/14 = v6 256x256 / 256x128; 15 = v7 256x256; 24 = v7 128x320; 25 = v12 256x256 (ring K loop, persistent
                     // walk, transposed streaming epilogue); 28 = v13 256x320

template <typename T, bool CONV, int MT>
int launch_v6(GemmP p, hipStream_t s, int mrows) {
  constexpr int BN_ = MT == 4 ? 256 : 128;
  constexpr int ring = 2 * (256 + BN_) * 64 * 2;
  constexpr int epi = 8 * 32 * STAGE_LD * 4;
  constexpr int lds = ring > epi ? ring : epi;
  static bool attr = false;
  if (!attr) {
    attr = true;
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v6<T, CONV, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  p.tiles_m = (mrows + 255) / 256;
  p.tiles_n = (p.N + BN_ - 1) / BN_;
  p.dbg = g_dbg;
  const int grid = p.tile_groups * p.tiles_m * p.tiles_n;
  if (grid <= 0) return OMG_OK;
  OMG_LAUNCH((gemm_kernel_v6<T, CONV, MT>), dim3(grid), dim3(512), lds, s, p);
  return omg_check_launch("gemm_v6");
}

template <typename T, bool CONV, int ABL = 0, int MT = 4, int NT = 4, bool XE = false, int EF = 0>
int launch_v7(GemmP p, hipStream_t s, int mrows) {
  constexpr int BM_ = MT * 64, BN_ = NT * 64;
  constexpr int lds = 2 * (BM_ + BN_) * 64 * 2 + (XE ? 4 * 8192 : 0);
  static bool attr = false;
  if (!attr) {
    attr = true;
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v7<T, CONV, ABL, MT, NT, XE, EF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  p.tiles_m = (mrows + BM_ - 1) / BM_;
  p.tiles_n = (p.N + BN_ - 1) / BN_;
  p.dbg = g_dbg;
  const int grid = p.tile_groups * p.tiles_m * p.tiles_n;
  if (grid <= 0) return OMG_OK;
  OMG_LAUNCH((gemm_kernel_v7<T, CONV, ABL, MT, NT, XE, EF>), dim3(grid), dim3(256), lds, s, p);
  return omg_check_launch("gemm_v7");
}

int num_cus();
#include "gemm_v12.h"      // the 256 x 256 four-wave tile: table-driven ring K loop, persistent tile walk (variant 25)
#include "gemm_v13.h"      // the 256 x 320 four-wave tile with class-pinned inline-asm MFMAs (variant 28)

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

// Tile choice from measured rates (profiles/r01_microbench_*.log, r05_exp_v13_*.log): a 256x256 tile wins whenever it can put >= ~120 tiles on
// the 256 CUs; below that the 256x128 kernel if IT reaches ~120 tiles, else the 128x128 kernel (2 blocks per CU).  Narrow outputs (N <= 128:
// LoRA-down, ControlNet conditioning embedding) never take BN = 256.  The 256x320 tile (28) where its rounds x relative tile time undercut the
// rest: it removes the padding of N = 640 / 960 / 1920 (256-wide: 768 / 1024 / 2048), replaces the 128x320 tile on the N = 320 / 640
// convolutions, and wins whenever 320-wide tiles fill fewer rounds (N = 1280 at M = 32768: 2 rounds instead of 3).  GEGLU stays on 256x256
// (a 160-wide wave tile cannot hold whole [32 value | 32 gate] blocks).
int choose_variant(int mrows, int groups, int N, bool conv, bool geglu = false) {
  if (g_variant != 0) return g_variant;
  const long t256 = (long)groups * ((mrows + 255) / 256) * ((N + 255) / 256);
  const long t256x128 = (long)groups * ((mrows + 255) / 256) * ((N + 127) / 128);
  const long t128x320 = (long)groups * ((mrows + 127) / 128) * ((N + 319) / 320);
  const long t256x320 = (long)groups * ((mrows + 255) / 256) * ((N + 319) / 320);
  // rounds of 256 one-block CUs x time per tile relative to a 256x256 tile (tools/shape_sweep.py, profiles/r01_shape_sweep_*.log,
  // r05_exp_v13_ab_*.log): 256x128 on eight waves 0.58; 128x320 on four waves 0.70 (its area is 0.625); 256x320 1.28 on the Linear layers
  // (area 1.25; the 256x256 kernel walks tiles persistently, this one does not) and 1.20 on the implicit-GEMM convolutions (K >= 2880 hides it).
  const double c256 = (N > 128 && t256 >= 120) ? (double)((t256 + 255) / 256) : 1e30;
  const double c128 = t256x128 >= 120 ? (double)((t256x128 + 255) / 256) * 0.58 * 1.06 : 1e30;
  // 128x320 only for the implicit-GEMM conv (long K): on the Linear layers' K = 640..5120 its per-tile overhead loses
  const double c320 = (conv && !geglu && N % 320 == 0 && t128x320 >= 120) ? (double)((t128x320 + 255) / 256) * 0.70 : 1e30;
  const double c13 = (!geglu && N % 320 == 0 && t256x320 >= 120) ? (double)((t256x320 + 255) / 256) * (conv ? 1.20 : 1.28) : 1e30;
  if (c13 < c256 && c13 < c128 && c13 < c320) return 28;
  if (c320 < c256 && c320 < c128) return 24;
  if (c256 <= c128) return c256 < 1e30 ? 25 : 1;   // 256x256 on four waves (v12); 15 = v7 / 13 = the eight-wave v6 of the same tile
  return 14;                                       // 256x128 on eight waves (v6)
}

template <typename T, bool CONV>
int launch(const GemmP& p, hipStream_t s) {
  const int mrows = p.tile_groups > 1 ? p.rows_per_group : p.M;
  if (g_use_glds) {
    int v = choose_variant(mrows, p.tile_groups, p.N, CONV, p.act == OMG_ACT_GEGLU);
    // v6 (interleaved DMA) handles everything except the LoRA second K-segment and > 2 GiB operands
    const long lim = 0x7fff0000L;
    const long a_sz = CONV ? (long)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * (p.C1 > p.C2 ? p.C1 : p.C2) * 2 : (long)p.M * p.lda * 2;
    const long c_sz = (long)p.M * (p.ldc > p.ldr ? p.ldc : p.ldr) * 2;
    const bool v6ok = p.K2 == 0 && p.A2 == nullptr && p.K % 64 == 0 && a_sz < lim && (long)p.N * p.ldw * 2 < lim && c_sz < lim &&
                      (p.group_bias == nullptr || (long)(p.M / (p.rows_per_group > 0 ? p.rows_per_group : 1) + 1) * p.ldgb * 2 < lim);
    if ((v == 24 || v == 28) && p.act == OMG_ACT_GEGLU) v = 25;      // forced variants: a 160-wide wave tile cannot hold whole [32 value | 32 gate] blocks
    // what the large-tile kernels do not handle (!v6ok) falls through to the 128x128 kernel below
    if (v == 24 && v6ok) return launch_v7<T, CONV, 0, 2, 5>(p, s, mrows);
    if (v == 15 && v6ok) return launch_v7<T, CONV>(p, s, mrows);
    // 25: the 256x256 tile as the heuristic uses it — gemm_kernel_v12 (five rotating half-stage buffers, persistent walk), one kernel per epilogue form
    if (v == 25 && v6ok) return launch_v12_form<T, CONV>(p, s, mrows);
    // 28: the 256x320 tile — gemm_kernel_v13, transposed streaming epilogue for the aligned 128-column groups
    if (v == 28 && v6ok) return launch_v13_form<T, CONV>(p, s, mrows);
#ifdef OMG_ABLATION_BUILDS   // make ABLATE=1: seven more instantiations of v7 for tools/gemm_ablate.py (3 minutes of compile time)
    if constexpr (!CONV && sizeof(T) == 2 && Vec<T>::is_f16) {      // ablation builds of v7 (tools/gemm_ablate.py), fp16 plain GEMM only
      if (v >= 17 && v <= 23 && v6ok) {
        switch (v - 16) {
          case 1: return launch_v7<T, CONV, 1>(p, s, mrows);
          case 2: return launch_v7<T, CONV, 2>(p, s, mrows);
          case 3: return launch_v7<T, CONV, 3>(p, s, mrows);
          case 4: return launch_v7<T, CONV, 4>(p, s, mrows);
          case 5: return launch_v7<T, CONV, 5>(p, s, mrows);
          case 6: return launch_v7<T, CONV, 6>(p, s, mrows);
          default: return launch_v7<T, CONV, 7>(p, s, mrows);
        }
      }
    }
#endif
    if ((v == 13 || v == 14) && v6ok) return v == 13 ? launch_v6<T, CONV, 4>(p, s, mrows) : launch_v6<T, CONV, 2>(p, s, mrows);
  }
  const int grid = p.tile_groups * p.tiles_m * p.tiles_n;
  if (grid <= 0) return OMG_OK;
  if (g_use_glds) {
    OMG_LAUNCH((gemm_kernel<T, CONV, true>), dim3(grid), dim3(256), LDS_BYTES, s, p);
  } else {
    OMG_LAUNCH((gemm_kernel<T, CONV, false>), dim3(grid), dim3(256), LDS_BYTES, s, p);
  }
  return omg_check_launch("gemm");
}

bool g_attr_set = false;
void ensure_attrs() {
  if (g_attr_set) return;
  g_attr_set = true;
  // > 64 KiB of dynamic LDS needs an explicit opt-in
  (void)hipFuncSetAttribute((const void*)gemm_kernel<f16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  (void)hipFuncSetAttribute((const void*)gemm_kernel<f16, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  (void)hipFuncSetAttribute((const void*)gemm_kernel<f16, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  (void)hipFuncSetAttribute((const void*)gemm_kernel<f16, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
#ifndef OMG_DEV_F16_ONLY
  (void)hipFuncSetAttribute((const void*)gemm_kernel<bf16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  (void)hipFuncSetAttribute((const void*)gemm_kernel<bf16, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  (void)hipFuncSetAttribute((const void*)gemm_kernel<bf16, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  (void)hipFuncSetAttribute((const void*)gemm_kernel<bf16, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
#endif
}

}  // namespace

extern "C" int omg_debug_read_ts(long long* out, int blocks) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(omg_dbg_ts), sizeof(long long) * 6 * (blocks < 8192 ? blocks : 8192));
}
extern "C" void omg_debug_set_glds(int on) { g_use_glds = on != 0; }
extern "C" void omg_debug_set_gemm_variant(int v) { g_variant = v & 0xff; g_dbg = v >> 8; }
// host-only: the tile variant the cost model picks for (rows per group, groups, N, conv) — lets the CPU tests pin the table
extern "C" int omg_debug_choose_variant(int mrows, int groups, int N, int conv) {
  const int saved = g_variant;
  g_variant = 0;
  const int v = choose_variant(mrows, groups, N, (conv & 1) != 0, (conv & 2) != 0);      // bit 1 of `conv`: a GEGLU epilogue
  g_variant = saved;
  return v;
}

extern "C" int omg_gemm(const omg_gemm_args* a, void* stream) {
  OMG_REQUIRE(a != nullptr, "omg_gemm: null args");
  OMG_REQUIRE(a->dtype == OMG_F16 || a->dtype == OMG_BF16, "omg_gemm: dtype");
  OMG_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "omg_gemm: shape");
  if (a->M == 0) return OMG_OK;
  OMG_REQUIRE(a->K % 8 == 0 && a->N % 8 == 0 && a->K2 % 8 == 0, "omg_gemm: K, K2, N must be multiples of 8");
  OMG_REQUIRE(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->ldc % 8 == 0, "omg_gemm: leading dims must be multiples of 8");
  OMG_REQUIRE(a->A && a->W && a->C, "omg_gemm: null operand");
  OMG_REQUIRE(a->groups >= 1 && (long)a->groups * a->rows_per_group == a->M, "omg_gemm: M != groups*rows_per_group");
  if (a->K2 > 0) OMG_REQUIRE(a->A2 && a->W2 && a->lda2 % 8 == 0 && a->ldw2 % 8 == 0, "omg_gemm: LoRA segment operands");
  if (a->act == OMG_ACT_GEGLU) OMG_REQUIRE(a->N % 64 == 0 && !a->residual && !a->group_bias, "omg_gemm: GEGLU needs N % 64 == 0, no residual");
  if (a->residual) OMG_REQUIRE(a->ldr % 8 == 0, "omg_gemm: ldr");
  if (a->group_bias) OMG_REQUIRE(a->ldgb % 8 == 0, "omg_gemm: ldgb");
  ensure_attrs();
  GemmP p{};
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = (const char*)a->A; p.lda = a->lda; p.W = (const char*)a->W; p.ldw = a->ldw;
  p.A2 = (const char*)a->A2; p.lda2 = a->lda2; p.W2 = (const char*)a->W2; p.ldw2 = a->ldw2;
  p.K2 = a->K2; p.a2_col_block = a->a2_col_block;
  const bool per_group = (a->group_adapter != nullptr) && a->groups > 1;
  p.tile_groups = per_group ? a->groups : 1;
  p.rows_per_group = a->rows_per_group;
  p.group_adapter = a->group_adapter;
  p.w_adapter_stride = a->w_adapter_stride; p.w2_adapter_stride = a->w2_adapter_stride;
  p.bias = (const char*)a->bias; p.group_bias = (const char*)a->group_bias; p.ldgb = a->ldgb;
  p.residual = (const char*)a->residual; p.ldr = a->ldr;
  p.act = a->act; p.out_scale = a->out_scale; p.C = (char*)a->C; p.ldc = a->ldc;
  const int mrows = per_group ? a->rows_per_group : a->M;
  p.tiles_m = (mrows + BM - 1) / BM;
  p.tiles_n = (a->N + BN - 1) / BN;
  hipStream_t s = (hipStream_t)stream;
#ifdef OMG_DEV_F16_ONLY
  OMG_REQUIRE(a->dtype == OMG_F16, "omg_gemm: this is a DEV=1 build (fp16 kernels only)");
  return launch<f16, false>(p, s);
#else
  return a->dtype == OMG_F16 ? launch<f16, false>(p, s) : launch<bf16, false>(p, s);
#endif
}

extern "C" int omg_conv2d(const omg_conv2d_args* a, void* stream) {
  OMG_REQUIRE(a != nullptr, "omg_conv2d: null args");
  OMG_REQUIRE(a->dtype == OMG_F16 || a->dtype == OMG_BF16, "omg_conv2d: dtype");
  OMG_REQUIRE(a->ksize == 1 || a->ksize == 3, "omg_conv2d: ksize must be 1 or 3");
  OMG_REQUIRE(a->stride == 1 || a->stride == 2, "omg_conv2d: stride must be 1 or 2");
  OMG_REQUIRE(a->C1 > 0 && a->C1 % 64 == 0 && a->C2 % 64 == 0 && a->C2 >= 0, "omg_conv2d: channels must be multiples of 64");
  OMG_REQUIRE(a->Cout % 8 == 0, "omg_conv2d: Cout % 8");
  OMG_REQUIRE(a->X1 && a->W && a->Y && (a->C2 == 0 || a->X2), "omg_conv2d: null operand");
  OMG_REQUIRE(!(a->upsample && a->stride != 1), "omg_conv2d: upsample with stride");
  const int Hl = a->upsample ? 2 * a->Hin : a->Hin, Wl = a->upsample ? 2 * a->Win : a->Win;
  const int pad = a->ksize == 3 ? 1 : 0;
  OMG_REQUIRE(a->Hout == (Hl + 2 * pad - a->ksize) / a->stride + 1 && a->Wout == (Wl + 2 * pad - a->ksize) / a->stride + 1,
              "omg_conv2d: output size mismatch");
  ensure_attrs();
  GemmP p{};
  const int Ctot = a->C1 + a->C2;
  p.M = a->B * a->Hout * a->Wout; p.N = a->Cout; p.K = a->ksize * a->ksize * Ctot;
  if (p.M == 0) return OMG_OK;
  p.A = (const char*)a->X1; p.X2 = (const char*)a->X2;
  p.W = (const char*)a->W; p.ldw = p.K;
  p.K2 = 0; p.tile_groups = 1; p.rows_per_group = a->Hout * a->Wout;
  p.bias = (const char*)a->bias; p.group_bias = (const char*)a->group_bias; p.ldgb = a->ldgb;
  p.residual = (const char*)a->residual; p.ldr = a->Cout;
  OMG_REQUIRE(a->act == OMG_ACT_NONE || a->act == OMG_ACT_SILU, "omg_conv2d: act");
  p.act = a->act; p.out_scale = a->out_scale; p.C = (char*)a->Y; p.ldc = a->Cout;
  p.Hin = a->Hin; p.Win = a->Win; p.C1 = a->C1; p.C2 = a->C2; p.Hout = a->Hout; p.Wout = a->Wout;
  p.ksize = a->ksize; p.stride = a->stride; p.upsample = a->upsample;
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  if (a->group_bias) OMG_REQUIRE(a->ldgb % 8 == 0, "omg_conv2d: ldgb");
  hipStream_t s = (hipStream_t)stream;
#ifdef OMG_DEV_F16_ONLY
  OMG_REQUIRE(a->dtype == OMG_F16, "omg_conv2d: this is a DEV=1 build (fp16 kernels only)");
  return launch<f16, true>(p, s);
#else
  return a->dtype == OMG_F16 ? launch<f16, true>(p, s) : launch<bf16, true>(p, s);
#endif
}
