// gemm_v13.h — the product kernel of the 256 x 320 x 64 tile (variant 28), FOUR waves (2 x 2 waves of 128 x 160 — 20 accumulator tiles = 320
// registers per lane).  Written in round 4, first run and landed in round 5 (profiles/r05_exp_v13_*.log: bitwise equal with every other variant;
// +15 ... 17 % on the N = 320 / 640 convolutions against the 128 x 320 tile, +3 ... 23 % on the 256-tile convolutions, +8 ... 18 % on the
// N = 640 / 1920 Linears, behind the 256 x 256 ring kernel where N = 1280 fills whole rounds of both; the whole benchmark +3.1 %).
//
// Why this tile:
//   * 320 = the width unit of the SDXL UNet (320 / 640 / 960 / 1280 / 1920 / 2560 / 3840): every N of the workload except the GEGLU projection is a whole
//     number of 320-wide tiles, where the 256-wide tile pads N = 640 to 768 (+20 % work) and N = 320 / 640 convolutions run the 128 x 320 tile of v7
//     (0.7 fragment reads per MFMA, 56 KB of LDS-DMA per 5.2 MFLOP);
//   * per MFMA it needs LESS of everything the 256 x 256 tile is short of: 0.45 fragment reads (256^2: 0.5), 72 KB of LDS-DMA per 10.5 MFLOP (64 KB per
//     8.4), 11 % fewer operand bytes from L2 per FLOP — and 25 % more work behind every tile's 7 - 9 us of prologue + epilogue + dispatch gap.
// Round 1 tried this tile as gemm_kernel_v7<.., 4, 5> and dropped it: with the MFMA builtin hipcc keeps all 320 accumulators in one allocation class
// and moves 1000 - 1300 of them between the AGPR and the VGPR half EVERY STAGE, plus 500 bytes of scratch (the same instantiation, compiled again in
// round 4: 1280 v_accvgpr_* + 88 scratch accesses in the steady-state stage).  Here the MFMAs are inline asm with the allocation class of each accumulator
// written in the constraint — acc[i][0..3] "+a" (256 AGPRs), acc[i][4] "+v" (64 VGPRs) — and the steady-state stage of the Linear kernel is 80 MFMAs,
// 36 ds_read_b128, 18 LDS-DMA, 11 VALU, no accumulator move, no scratch (tests/test_codeobj.py checks that on every EXP build).
// What inline-asm MFMAs cost: the hazard recogniser does not see them.  Nothing but MFMAs touches an accumulator between the two fences below
// (acc_fence: an asm that names all 20 accumulators and carries the wait states), so the only hazards are VALU write -> first MFMA (fence in front of
// the loop) and last MFMA -> first epilogue read (fence behind it; without it hipcc hoisted v_accvgpr_read one instruction behind the MFMA that
// produces the value).
//
// K loop: gemm_kernel_v7's (two 72 KB stages, fragments one k-step ahead in two register sets, one barrier per stage in front of its last k-step,
// the W half of stage kt + 1 issued over k-step 0 and the A half of stage kt + 2 over k-step 3, one LDS-DMA per two MFMAs).  The five-buffer ring of
// gemm_v11.h does not fit: 3 x 32 KB + 2 x 40 KB = 176 KB.
// Column ownership: wave column wn owns the 128 columns [128 wn, 128 wn + 128) (j = 0..3) and the 32 columns [256 + 32 wn, + 32) (j = 4), so that
// both waves' 128-column groups start on a 256-byte boundary of the output row: XE (variant 28) sends them through the transposed streaming epilogue
// of gemm_epilogue.h unchanged (whole 256-byte row segments, nt) and stores the odd 32 columns register-direct without nt (the two waves' halves of that
// 128-byte line meet in L2).  (Storing everything register-direct, as the 128 x 320 tile does — variant 27 of the experiment — was 1 ... 12 % slower.)
// Epilogue forms: 1 bias only, 4 per-row group bias / SiLU / residual decided per unit, 5 residual by register-direct loads.  No GEGLU (a 160-wide wave
// tile cannot hold whole [32 value | 32 gate] blocks): the launcher returns such problems to the 256 x 256 kernel.
// Values: the same loads, the same MFMA order per accumulator, the same epilogue arithmetic as every other variant — bitwise identical by construction.
// This header is included inside gemm.hip's anonymous namespace.

template <typename T> struct MfmaAsm;
template <> struct MfmaAsm<f16> {
  static OMG_DEV void a(f32x16& c, f16x8 x, f16x8 y) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(x), "v"(y)); }
  static OMG_DEV void v(f32x16& c, f16x8 x, f16x8 y) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y)); }
};
template <> struct MfmaAsm<bf16> {
  static OMG_DEV void a(f32x16& c, bf16x8 x, bf16x8 y) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(x), "v"(y)); }
  static OMG_DEV void v(f32x16& c, bf16x8 x, bf16x8 y) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y)); }
};
// All 20 accumulators pass through these four statements (an asm takes at most 30 operands; "+" counts twice): whatever reads or writes an
// accumulator on the other side is ordered against every MFMA, and the first statement carries the wait states (s_nop 15 = 16 of them; a 16-pass
// MFMA result needs 18 before a VALU / v_accvgpr read — the MFMAs in front of the last one have long retired, the pipe is in order).
OMG_DEV void acc_fence(f32x16 (&acc)[4][5]) {
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+v"(acc[0][4]));
  asm volatile("" : "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]), "+v"(acc[1][4]));
  asm volatile("" : "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+v"(acc[2][4]));
  asm volatile("" : "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3]), "+v"(acc[3][4]));
}

// first output column of column block j of the wave: j < 4 -> the aligned 128-column group, j = 4 -> the odd 32 columns
#define OMG_V13_COL(j_, wn0_, wn4_) ((j_) < 4 ? (wn0_) + (j_) * 32 : (wn4_))

// acc_init_bias (gemm_epilogue.h) for the column ownership above: accumulators start at bias (+ the folded per-sample bias)
template <typename T>
OMG_DEV bool acc_init_bias13(const GemmP& p, f32x16 (&acc)[4][5], int lane, int m0, int wn0, int wn4) {
  const int hi = lane >> 5;
  const bool fold_gb = fold_group_bias(p);
  const __amdgpu_buffer_rsrc_t rsB = epi_rsrc(p.bias, (long)p.N * 2);
  const __amdgpu_buffer_rsrc_t rsG = epi_rsrc(fold_gb ? p.group_bias + (long)(m0 / p.rows_per_group) * p.ldgb * 2 : nullptr, (long)p.N * 2);
  u32x4 rb[5][2], rg[5][2];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int c = OMG_V13_COL(j, wn0, wn4) + pr * 16 + hi * 8;      // c >= N is beyond num_records: zeros
      rb[j][pr] = __builtin_amdgcn_raw_buffer_load_b128(rsB, c * 2, 0, 0);
      rg[j][pr] = __builtin_amdgcn_raw_buffer_load_b128(rsG, c * 2, 0, 0);
    }
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      float f[8], g[8];
      decode_runs<T>(rb[j][pr], f);
      decode_runs<T>(rg[j][pr], g);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][j][pr * 8 + e] = f[e] + g[e];
    }
  return p.group_bias != nullptr && !fold_gb;
}

// epilogue_rows (gemm_epilogue.h) for the 128 x 160 wave tile.  RS / GENERIC as there; the units of j < 4 go through xe_put / xe_flush (the
// EpiCtx<4> view cx4 is all those two read), the units of j = 4 are stored register-direct.  Same arithmetic, same packing, same bits.
template <typename T, bool RS, bool GENERIC>
OMG_DEV void epilogue_rows13(const GemmP& p, f32x16 (&acc)[4][5], const EpiCtx<5>& cx, const EpiCtx<4>& cx4, int lane_col4, bool has_gb) {
  const float osc = p.out_scale;
  const bool has_rs = GENERIC ? p.residual != nullptr : RS;
  u32x4 rraw[2][5][2];          // residual of row block i: the lane's 10 16-byte units, fetched one row block ahead
#define OMG_V13_LCOL(j_) ((j_) < 4 ? cx.lane_col : lane_col4)
#define OMG_V13_SOFF(j_, pr_) ((((j_) < 4 ? (j_) * 32 : 0) + (pr_) * 16) * 2)
#define OMG_FETCH_RES13(i_, buf_)                                                                          \
  do {                                                                                                     \
    const int gm_ = cx.wm0 + (i_) * 32 + cx.l31;                                                           \
    const int rrow_ = gm_ < cx.m_end ? gm_ * (int)p.ldr * 2 : EPI_OOB;                                     \
    _Pragma("unroll") for (int j = 0; j < 5; ++j)                                                          \
      _Pragma("unroll") for (int pr = 0; pr < 2; ++pr)                                                     \
        rraw[buf_][j][pr] = __builtin_amdgcn_raw_buffer_load_b128(cx.rsR, (rrow_ + OMG_V13_LCOL(j)) | cx.voob[j][pr], OMG_V13_SOFF(j, pr), 0); \
  } while (0)
  if (has_rs) OMG_FETCH_RES13(0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = cx.wm0 + i * 32 + cx.l31;
    const bool row_ok = gm < cx.m_end;
    if (has_rs && i + 1 < 4) OMG_FETCH_RES13(i + 1, (i + 1) & 1);
    const int crow = row_ok ? gm * (int)p.ldc * 2 : EPI_OOB;
    const int grow = GENERIC && has_gb && row_ok ? ((gm / p.rows_per_group) * (int)p.ldgb) * 2 : EPI_OOB;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[i][j][pr * 8 + e];
        if constexpr (GENERIC) {
          if (has_gb) {
            float f[8];
            load_runs<T>(cx.rsG, (grow + OMG_V13_LCOL(j)) | cx.voob[j][pr], OMG_V13_SOFF(j, pr), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += f[e];
          }
          if (p.act == OMG_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_fast(v[e]);
          }
        }
        if (has_rs) {
          const u32x4 rr = rraw[i & 1][j][pr];
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
        if (j < 4) xe_put<T, 256>(cx4, 4 * j + 2 * pr, v);
        else store_runs<T>(cx.rsC, (crow + OMG_V13_LCOL(j)) | cx.voob[j][pr], OMG_V13_SOFF(j, pr), v);
        __builtin_amdgcn_sched_barrier(0);
      }
    xe_flush<T, 256>(p, cx4, i, cx4.wn0);
  }
#undef OMG_FETCH_RES13
#undef OMG_V13_SOFF
#undef OMG_V13_LCOL
}

template <typename T, int EF>
OMG_DEV void epilogue13(const GemmP& p, f32x16 (&acc)[4][5], int lane, int wm0, int wn0, int wn4, int m_end, bool has_gb, char* xl) {
  static_assert(EF == 1 || EF == 4 || EF == 5, "forms of the 256 x 320 tile: bias only, generic, residual");
  EpiCtx<5> cx;
  cx.hi = lane >> 5; cx.l31 = lane & 31; cx.wm0 = wm0; cx.m_end = m_end;
  cx.rsC = epi_rsrc(p.C, ((long)(p.M - 1) * p.ldc + p.N) * 2);
  cx.rsR = epi_rsrc(p.residual, ((long)(p.M - 1) * p.ldr + p.N) * 2);
  cx.rsG = epi_rsrc(has_gb ? p.group_bias : nullptr, 0x7effff00L);
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) cx.voob[j][pr] = (OMG_V13_COL(j, wn0, wn4) + pr * 16 + cx.hi * 8 < p.N && !(p.dbg & 1024)) ? 0 : EPI_OOB;
  cx.lane_col = (wn0 + cx.hi * 8) * 2;
  cx.wn0 = wn0; cx.n_out = p.N; cx.xl = xl; cx.rl = nullptr;
  const int lane_col4 = (wn4 + cx.hi * 8) * 2;
  EpiCtx<4> cx4;                 // what xe_put / xe_flush read
  cx4.rsC = cx.rsC; cx4.rsR = cx.rsR; cx4.rsG = cx.rsG;
  cx4.hi = cx.hi; cx4.l31 = cx.l31; cx4.wm0 = wm0; cx4.m_end = m_end; cx4.lane_col = cx.lane_col; cx4.wn0 = wn0; cx4.n_out = p.N; cx4.xl = xl; cx4.rl = nullptr;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) cx4.voob[j][pr] = cx.voob[j][pr];
  if constexpr (EF == 1) epilogue_rows13<T, false, false>(p, acc, cx, cx4, lane_col4, false);
  else if constexpr (EF == 5) epilogue_rows13<T, true, false>(p, acc, cx, cx4, lane_col4, false);
  else epilogue_rows13<T, false, true>(p, acc, cx, cx4, lane_col4, has_gb);
}

template <typename T, bool CONV, int EF>
__global__ __launch_bounds__(256, 1) void gemm_kernel_v13(GemmP p) {
  constexpr int MT = 4, NT = 5;
  constexpr int BM_ = MT * 64, BN_ = NT * 64, BKc = 64;
  constexpr int AB = MT * 2, WB = NT * 2;          // A / W row blocks (8 rows each) per wave per stage
  constexpr int NMM = MT * NT;                     // MFMAs per k-step
  constexpr int SLOTS = NMM / 2;                   // pairs of MFMAs per k-step
  constexpr int NRD = MT + NT;                     // fragment reads per k-step
  constexpr int A_BYTES = BM_ * BKc * 2;
  constexpr int STAGE_BYTES = (BM_ + BN_) * BKc * 2;
  static_assert(2 * STAGE_BYTES <= 160 * 1024, "two stages in LDS");

  const bool ts_on = (p.dbg & 16) && blockIdx.x < 8192 && threadIdx.x == 0;      // tools/gemm_timeline.py: per-block time stamps
  long long ts0 = 0, ts1 = 0, ts2 = 0;
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

  // DMA: one instruction moves 8 rows x 128 B; wave w owns row blocks w, w+4, ... of A (8 of 32) and of W (10 of 40)
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
  const int wn0 = n0 + wn * 128, wn4 = n0 + 256 + wn * 32;       // first column of the aligned group / of the odd block (header)
  f32x16 acc[MT][NT];
  using V8 = typename Vec<T>::v8;
  // A fragment i of k-step ks at aoff[ks] + i * 4096; W fragment j < 4 at boff[ks] + j * 4096 (tile rows 128 wn + 32 j), j = 4 at boff4[ks]
  // (tile rows 256 + 32 wn); every first row is a multiple of 32, so the swizzle term is the same
  int aoff[4], boff[4], boff4[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int sw = ((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4;
    aoff[ks] = (wm * (MT * 32) + l31) * 128 + sw;
    boff[ks] = A_BYTES + (wn * 128 + l31) * 128 + sw;
    boff4[ks] = A_BYTES + (256 + wn * 32 + l31) * 128 + sw;
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
#define OMG_WFRAG(sb_, ks_, j_) (*(const V8*)((sb_) + ((j_) < 4 ? boff[ks_] + (j_) * 4096 : boff4[ks_])))
#define OMG_RD(f_, sb_, ks_)                                                                               \
  do {                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) bf[f_][j] = OMG_WFRAG(sb_, ks_, j);                     \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) af[f_][i] = *(const V8*)((sb_) + aoff[ks_] + i * 4096); \
  } while (0)
  // read order = order of first use by the MFMAs (n = NT i + j): W0, A0, W1 .. W4, A1 .. A3
#define OMG_RD1(f_, sb_, ks_, r_)                                                                          \
  do {                                                                                                     \
    const bool isA_ = (r_) == 1 || (r_) > NT;                                                              \
    const int idx_ = (r_) <= 1 ? 0 : (r_) <= NT ? (r_) - 1 : (r_) - NT;                                    \
    if (!isA_) bf[f_][idx_] = OMG_WFRAG(sb_, ks_, idx_);                                                   \
    else af[f_][idx_] = *(const V8*)((sb_) + aoff[ks_] + idx_ * 4096);                                     \
  } while (0)
  // acc[i][j] += W fragment j x A fragment i (the transposed tile), allocation class by j (header)
#define OMG_MM1(f_, n_)                                                                                    \
  do {                                                                                                     \
    if ((n_) % NT == 4) MfmaAsm<T>::v(acc[(n_) / NT][4], bf[f_][4], af[f_][(n_) / NT]);                    \
    else MfmaAsm<T>::a(acc[(n_) / NT][(n_) % NT], bf[f_][(n_) % NT], af[f_][(n_) / NT]);                   \
  } while (0)
  // slot s_ of a k-step: MFMA 2s, reads, MFMA 2s+1, DMAs.  RD_ = 1: one read per slot, 2: two per slot from the first slot on.
#define OMG_KSTEP(f_, RD_, rb_, rks_, DMA_, d0_, dn_, db_)                                                 \
  do {                                                                                                     \
    constexpr int RPS_ = (RD_) == 2 ? 2 : (NRD + SLOTS - 1) / SLOTS;                                       \
    constexpr int DPS_ = ((dn_) + SLOTS - 1) / SLOTS;                                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < SLOTS; ++s_) {                                                 \
      OMG_MM1(f_, 2 * s_);                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if ((RD_) != 0) {                                                                                    \
        _Pragma("unroll") for (int q_ = 0; q_ < RPS_; ++q_)                                                \
          if (s_ * RPS_ + q_ < NRD) OMG_RD1(1 - (f_), rb_, rks_, s_ * RPS_ + q_);                          \
      }                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      OMG_MM1(f_, 2 * s_ + 1);                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (DMA_) {                                                                                          \
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
  const bool gb_epi = acc_init_bias13<T>(p, acc, lane, m0, wn0, wn4);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (p.dbg & 16) ts1 = __builtin_amdgcn_s_memrealtime();
  OMG_PREP(1);
  if (nk > 1) OMG_DMAN(0, AB, smem + STAGE_BYTES);
  OMG_RD(0, smem, 0);
  acc_fence(acc);            // the VALU writes of the initialisation are behind every MFMA's wait states

#define OMG_STAGE(HAS1_, HAS2_)                                                                            \
  do {                                                                                                     \
    const char* cur = smem + (kt & 1) * STAGE_BYTES;                                                       \
    char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;                                                       \
    OMG_KSTEP(0, 1, cur, 1, HAS1_, AB, WB, nxt);                                                           \
    OMG_KSTEP(1, 1, cur, 2, false, 0, 0, nxt);                                                             \
    OMG_KSTEP(0, 1, cur, 3, false, 0, 0, nxt);                                                             \
    /* stage kt+1 has landed (this wave's part), this wave's reads of `cur` are complete: join the block */ \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();              \
    if (HAS2_) OMG_PREP(kt + 2);                                                                           \
    OMG_KSTEP(1, (HAS1_) ? 2 : 0, nxt, 0, HAS2_, 0, AB, (char*)cur);                                       \
  } while (0)
  int kt = 0;
  for (; kt < nk - 2; ++kt) OMG_STAGE(true, true);
  if (kt < nk - 1) { OMG_STAGE(true, false); ++kt; }
  OMG_STAGE(false, false);
#undef OMG_STAGE
#undef OMG_PREP
#undef OMG_DMA
#undef OMG_DMAN
#undef OMG_WFRAG
#undef OMG_RD
#undef OMG_RD1
#undef OMG_MM1
#undef OMG_KSTEP
  acc_fence(acc);            // every MFMA has retired before anything reads an accumulator
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
  // XE: 8 KB per wave at the start of LDS — behind the last stage's barrier no wave reads a stage buffer any more (that stage issues no fragment
  // reads behind its barrier: its last k-step computes from registers)
  epilogue13<T, EF>(p, acc, lane, m0 + wm * (MT * 32), wn0, wn4, m_end, gb_epi, smem + w * 8192);
  if (ts_on) {
    long long* t = omg_dbg_ts[blockIdx.x];
    t[0] = ts0; t[1] = ts1; t[2] = ts2; t[3] = __builtin_amdgcn_s_memrealtime();
    t[4] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all 32 bits
    t[5] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
  }
}
#undef OMG_V13_COL

template <typename T, bool CONV, int EF>
int launch_v13(GemmP p, hipStream_t s, int mrows) {
  constexpr int lds = 2 * (256 + 320) * 64 * 2;
  static bool attr = false;
  if (!attr) {
    attr = true;
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v13<T, CONV, EF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  p.tiles_m = (mrows + 255) / 256;
  p.tiles_n = (p.N + 319) / 320;
  p.dbg = g_dbg;
  const int grid = p.tile_groups * p.tiles_m * p.tiles_n;
  if (grid <= 0) return OMG_OK;
  OMG_LAUNCH((gemm_kernel_v13<T, CONV, EF>), dim3(grid), dim3(256), lds, s, p);
  return omg_check_launch("gemm_v13");
}
// GEGLU problems are not this tile's (header): the caller sends them to the 256 x 256 kernel
template <typename T, bool CONV>
int launch_v13_form(const GemmP& p, hipStream_t s, int mrows) {
  const bool gb_rows = p.group_bias != nullptr && p.rows_per_group % 256 != 0;       // == !fold_group_bias
  if (gb_rows || p.act == OMG_ACT_SILU) return launch_v13<T, CONV, 4>(p, s, mrows);
  if (p.residual != nullptr) return launch_v13<T, CONV, 5>(p, s, mrows);
  return launch_v13<T, CONV, 1>(p, s, mrows);
}
