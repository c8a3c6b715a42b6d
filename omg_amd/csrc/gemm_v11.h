// gemm_v11.h — the 256 x 256 x 64 four-wave GEMM / implicit-GEMM convolution with a TABLE-DRIVEN K loop (round 4).  Tile, wave layout
// (2 x 2 waves of 128 x 128, 256 accumulators in AGPRs), LDS image, XOR swizzle, MFMA order per accumulator and every epilogue are
// gemm_kernel_v7's — results are torch.equal with every other variant (tests/test_kernels_gpu.py) — what differs is WHEN things are issued:
//   * one fragment register set per k-step of a stage (4 x 8 fragments = 128 VGPRs), filled by inline-asm ds_read_b128 from a per-k-step
//     base address + immediate offsets.  hipcc does not track these reads, so the only LDS waits in the loop are the ones placed here:
//     one counted `s_waitcnt lgkmcnt(N)` in front of each k-step (N = the number of younger reads in flight, derived from the table) and one
//     `lgkmcnt(0)` in front of the stage barrier (v7: 23 compiler-placed partial waits per stage);
//   * a stage is 32 SLOTS of two MFMAs; a table gives, per fragment read, per LDS-DMA instruction and for the barrier, the slot it sits in.
//     The stage bodies are straight-line macro code GENERATED from the tables by tools/gen_ksched.py into gemm_v11_sched.inc (a first version
//     unrolled the tables with nested constexpr lambdas: hipcc did not finish one translation unit in 30 minutes);
//   * SCH >= 5 ("ring"): the 160 KB of LDS are FIVE 32 KB half-stage buffers — A current, W current, A next, W next, spare — whose roles rotate
//     every stage (three A buffers cycle, two W buffers swap; five scalar moves), instead of two 64 KB stages.  The A half of stage kt + 2
//     goes into the spare from the first slot of stage kt on, the W half into the current W buffer behind the barrier: the 16 LDS-DMA
//     instructions of a stage are spread over the WHOLE stage (one per four MFMAs) and have up to a stage more time to land.  The fifth
//     buffer is the XE staging region of the epilogue, which nobody needs inside the loop.
// Measured (profiles/r04_ksched_*.log, interleaved A/B on one box, all variants bitwise equal): reading fragments two k-steps ahead (schedules
// 0, 1, 2, 4) is worth -2 ... +1 % against v7 — the K loop is NOT bound by LDS read latency; two DMAs per slot (3) cost 5 - 8 %; the ring
// (5, 6, 8, 9: placements inside it do not matter) gains +6 ... 8 % at K = 5120, +3 ... 4 % on the K = 1280 projections with a residual,
// +0.5 ... 2 % elsewhere, 0 on the convolutions: it pays where the A operand streams from HBM.  Schedule 5 is the product kernel of the
// 256 x 256 tile (variant 25); the others are built by `make EXP=1` (variants 35 + SCH) for tools/ksched_ab.py.
// This header is included inside gemm.hip's anonymous namespace.
template <typename T, bool CONV, int EF, int SCH>
__global__ __launch_bounds__(256, 1) void gemm_kernel_v11(GemmP p) {
  constexpr int MT = 4, NT = 4;
  constexpr bool XE = true;
  constexpr int BM_ = MT * 64, BN_ = NT * 64, BKc = 64;
  constexpr int AB = MT * 2, WB = NT * 2;          // A / W row blocks (8 rows each) per wave per stage
  constexpr int A_BYTES = BM_ * BKc * 2;
  constexpr int STAGE_BYTES = (BM_ + BN_) * BKc * 2;
  static_assert(STAGE_BYTES == 65536, "the buffer toggle is one address bit");
  // SCH >= 5: five 32 KB half-stage buffers R0..R4 (A cur, W cur, A next, W next, spare; R4 = the XE region, free inside the loop) whose roles
  // rotate every stage, instead of two 64 KB stages: the A half of stage kt + 2 can go into the spare BEFORE the stage's barrier
  constexpr bool RING = SCH >= 5;
  constexpr int HALF = 32768;

#ifdef OMG_EXP_KSCHED
  // SCH == 10 (make EXP=1, variant 31; NOT RUN): schedule 5 with a shorter way to the first LDS-DMA.  The emitted prologue of the product kernel
  // reaches its first DMA after SIX serialised scalar-cache round trips (the launch parameters are fetched field by field, each batch behind the
  // branch that needs it) and — with per-sample weight slots — one vector-memory round trip for the group's adapter id, all of it per tile.
  // Here every launch parameter the way to the first DMA needs is requested in ONE batch (the asm only makes them live at this point), and the
  // adapter id comes through the scalar cache (below).
  if constexpr (SCH == 10)
    asm volatile("" ::"s"(p.M), "s"(p.N), "s"(p.K), "s"(p.A), "s"(p.lda), "s"(p.W), "s"(p.ldw), "s"(p.tile_groups), "s"(p.rows_per_group),
                 "s"(p.group_adapter), "s"(p.w_adapter_stride), "s"(p.tiles_m), "s"(p.tiles_n), "s"(p.dbg), "s"(p.bias), "s"(p.group_bias), "s"(p.ldgb));
  if constexpr (SCH == 10 && CONV)
    asm volatile("" ::"s"(p.Hin), "s"(p.Win), "s"(p.C1), "s"(p.C2), "s"(p.Hout), "s"(p.Wout), "s"(p.ksize), "s"(p.stride), "s"(p.upsample), "s"(p.X2));
#endif
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
#ifdef OMG_EXP_KSCHED
  if constexpr (SCH == 10) {      // the adapter id of the tile's group through the scalar cache (the table is written before the launch, never by it)
    if (p.group_adapter != nullptr) adapter = *(const __attribute__((address_space(4))) int*)(p.group_adapter + grp);
  } else
#endif
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
      const int i_ = (d_) >= AB ? (d_) - AB : 0;                                                           \
      dma16(rsW, (nb_) + A_BYTES + ldo + i_ * 4096, voffW[i_], koff);                                      \
    }                                                                                                      \
  } while (0)
#define OMG_DMAN(first_, n_, nb_)                                                                          \
  do { _Pragma("unroll") for (int d_ = 0; d_ < (n_); ++d_) OMG_DMA((first_) + d_, nb_); } while (0)
  // ---- fragment sets: one per k-step of a stage (4 x (4 W + 4 A fragments) = 128 VGPRs), read by inline-asm ds_read_b128 — the compiler
  // does not track them, so the only LDS waits in the loop are the ones the generated stage body places
  V8 fw[4][NT], fa[4][MT];
  unsigned la[4], lw[4];       // LDS byte address of fragment 0 of k-step ks in buffer 0; fragment i is + i * 4096, buffer 1 is ^ STAGE_BYTES
  {
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned sw = (unsigned)(((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
      la[ks] = lds0 + (unsigned)((wm * (MT * 32) + l31) * 128) + sw;       // (MT*32) >> 1 and (NT*32) >> 1 are multiples of 8: same swizzle term
      lw[ks] = lds0 + (unsigned)((RING ? 0 : A_BYTES) + (wn * (NT * 32) + l31) * 128) + sw;
    }
  }
  int sa_cur = 0, sw_cur = HALF, sa_nxt = 2 * HALF, sw_nxt = 3 * HALF, s_sp = 4 * HALF;      // RING: byte offsets of the five roles
  (void)sa_cur; (void)sw_cur; (void)sa_nxt; (void)sw_nxt; (void)s_sp;
  // RING: DMA instruction d of the prepared stage kt + 2: its A half into the spare, its W half into the current W buffer (behind the barrier)
#define OMG_DMAR(d_)                                                                                       \
  do {                                                                                                     \
    if ((d_) < AB) {                                                                                       \
      const int i_ = (d_) < AB ? (d_) : 0;                                                                 \
      if (CONV) dma16(x2 ? rsA2 : rsA, smem + s_sp + ldo + i_ * 4096,                                      \
                      conv_voff(cb[i_], cy[i_], cx[i_], dchunk, p.stride, tap_dy, tap_dx, Hl, Wl, p.upsample, p.Hin, p.Win, xCb, c0b), 0); \
      else dma16(rsA, smem + s_sp + ldo + i_ * 4096, voffA[i_], koff);                                     \
    } else {                                                                                               \
      const int i_ = (d_) >= AB ? (d_) - AB : 0;                                                           \
      dma16(rsW, smem + sw_cur + ldo + i_ * 4096, voffW[i_], koff);                                        \
    }                                                                                                      \
  } while (0)
#define OMG_SB __builtin_amdgcn_sched_barrier(0)
  // read r_ of set ks_ (order of first use by the MFMAs n = NT i + j: W0, A0, W1 .. W3, A1 .. A3) from the buffer selected by tog_
#define OMG_XRD1(ks_, r_, tog_)                                                                            \
  do {                                                                                                     \
    constexpr bool isA_ = (r_) == 1 || (r_) > NT;                                                          \
    constexpr int idx_ = (r_) <= 1 ? 0 : (r_) <= NT ? (r_) - 1 : (r_) - NT;                                \
    unsigned ad_;                                                                                          \
    if constexpr (RING) ad_ = (isA_ ? la[ks_] : lw[ks_]) + (unsigned)((tog_) == tnxt ? (isA_ ? sa_nxt : sw_nxt) : (isA_ ? sa_cur : sw_cur)); \
    else ad_ = (isA_ ? la[ks_] : lw[ks_]) ^ (tog_);                                                        \
    if constexpr (isA_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[ks_][idx_]) : "v"(ad_), "n"(idx_ * 4096)); \
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[ks_][idx_]) : "v"(ad_), "n"(idx_ * 4096));  \
  } while (0)
#define OMG_XMM1(ks_, n_) acc[(n_) / NT][(n_) % NT] = Vec<T>::mfma32(fw[ks_][(n_) % NT], fa[ks_][(n_) / NT], acc[(n_) / NT][(n_) % NT])
  // the fragments of k-step ks_ are back when at most left_ younger LDS reads are still in flight (LDS returns in order); the "+v"
  // operands make every MFMA of the k-step depend on the wait
#define OMG_XWAIT(ks_, left_)                                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #left_ ")"                                                             \
               : "+v"(fw[ks_][0]), "+v"(fw[ks_][1]), "+v"(fw[ks_][2]), "+v"(fw[ks_][3]),                   \
                 "+v"(fa[ks_][0]), "+v"(fa[ks_][1]), "+v"(fa[ks_][2]), "+v"(fa[ks_][3]))
#include "gemm_v11_sched.inc"
#ifdef OMG_EXP_KSCHED
#define OMG_KS_DISPATCH(what_)                                                                             \
  do {                                                                                                     \
    if constexpr (SCH == 0) { what_(0); } else if constexpr (SCH == 1) { what_(1); } else if constexpr (SCH == 2) { what_(2); } \
    else if constexpr (SCH == 3) { what_(3); } else if constexpr (SCH == 4) { what_(4); } else if constexpr (SCH == 5) { what_(5); } \
    else if constexpr (SCH == 6) { what_(6); } else if constexpr (SCH == 7) { what_(7); } else if constexpr (SCH == 8) { what_(8); } \
    else if constexpr (SCH == 10) { what_(5); } else { what_(9); } \
  } while (0)
#else
  static_assert(SCH == 5, "the product build carries schedule 5 only (make EXP=1 for the others)");
#define OMG_KS_DISPATCH(what_) do { what_(5); } while (0)
#endif

  // prologue: stage 0 completely, stage 1 completely, the reads a previous stage would have issued for stage 0
  OMG_PREP(0);
  OMG_DMAN(0, AB + WB, smem);
  const bool gb_epi = acc_init_bias<T, MT, NT>(p, acc, lane, m0, n0 + wn * (NT * 32));
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (p.dbg & 16) ts1 = __builtin_amdgcn_s_memrealtime();
  OMG_PREP(1);
  if (nk > 1) OMG_DMAN(0, AB + WB, smem + STAGE_BYTES);
#define OMG_KS_PRO(i_) OMG_KS_PROLOGUE_##i_()
  { const unsigned tnxt = RING ? 1u : 0xffffffffu; (void)tnxt; OMG_KS_DISPATCH(OMG_KS_PRO); }      // the prologue reads use tag 0 = the current roles / buffer 0

  int kt = 0;
  // two stages: tcur / tnxt are the buffer-select address bit; RING: tags (0 = current, 1 = next roles)
#define OMG_KS_VARS const unsigned tcur = RING ? 0u : (unsigned)(kt & 1) * (unsigned)STAGE_BYTES; const unsigned tnxt = RING ? 1u : tcur ^ (unsigned)STAGE_BYTES; \
  char* curb = smem + (kt & 1) * STAGE_BYTES; (void)tnxt; (void)curb
#define OMG_KS_ROT do { if constexpr (RING) { const int t_ = sa_cur; sa_cur = sa_nxt; sa_nxt = s_sp; s_sp = t_; const int u_ = sw_cur; sw_cur = sw_nxt; sw_nxt = u_; } } while (0)
#define OMG_KS_TT(i_) OMG_KS_STAGE_##i_(true, true)
#define OMG_KS_TF(i_) OMG_KS_STAGE_##i_(true, false)
#define OMG_KS_FF(i_) OMG_KS_STAGE_##i_(false, false)
  for (; kt < nk - 2; ++kt) { OMG_KS_VARS; OMG_KS_DISPATCH(OMG_KS_TT); OMG_KS_ROT; }
  if (kt < nk - 1) { OMG_KS_VARS; OMG_KS_DISPATCH(OMG_KS_TF); OMG_KS_ROT; ++kt; }
  { OMG_KS_VARS; OMG_KS_DISPATCH(OMG_KS_FF); }
#undef OMG_KS_TT
#undef OMG_KS_TF
#undef OMG_KS_FF
#undef OMG_KS_VARS
#undef OMG_KS_ROT
#undef OMG_DMAR
#undef OMG_KS_PRO
#undef OMG_KS_DISPATCH
#undef OMG_XWAIT
#undef OMG_XMM1
#undef OMG_XRD1
#undef OMG_SB
#undef OMG_PREP
#undef OMG_DMA
#undef OMG_DMAN
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
  // XE: 8 KB per wave in the fifth 32 KB region (nobody reads a stage buffer after the last stage's barrier)
  if constexpr (EF == 2) {
    // residual: staged through the stage buffers, which nobody reads after the last stage's barrier (that stage issues no reads behind it)
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

template <typename T, bool CONV, int EF, int SCH>
int launch_v11(GemmP p, hipStream_t s, int mrows) {
  constexpr int lds = 2 * (256 + 256) * 64 * 2 + 4 * 8192;
  static bool attr = false;
  if (!attr) {
    attr = true;
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<T, CONV, EF, SCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  p.tiles_m = (mrows + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  p.dbg = g_dbg;
  const int grid = p.tile_groups * p.tiles_m * p.tiles_n;
  if (grid <= 0) return OMG_OK;
  OMG_LAUNCH((gemm_kernel_v11<T, CONV, EF, SCH>), dim3(grid), dim3(256), lds, s, p);
  return omg_check_launch("gemm_v11");
}
template <typename T, bool CONV, int SCH>
int launch_v11_form(const GemmP& p, hipStream_t s, int mrows) {
  const bool gb_rows = p.group_bias != nullptr && p.rows_per_group % 256 != 0;
  if (p.act == OMG_ACT_GEGLU) return launch_v11<T, CONV, 3, SCH>(p, s, mrows);
  if (gb_rows || p.act == OMG_ACT_SILU) return launch_v11<T, CONV, 4, SCH>(p, s, mrows);
  if (p.residual != nullptr) return launch_v11<T, CONV, 2, SCH>(p, s, mrows);
  return launch_v11<T, CONV, 1, SCH>(p, s, mrows);
}
