// gemm_v10_exp.h — EXPERIMENT, not part of the product build (compiled only with `make EXP=1`, reached only through
// omg_debug_set_gemm_variant(35)): gemm_kernel_v7's 256 x 256 x 64 tile, prologue, LDS image and epilogues with the K loop's latency
// discipline changed to what the vendor's hand-written kernel does (profiles/r03_hipblaslt_main_loop_vs_v7.txt):
//   * one fragment register set per k-step of a stage (128 VGPRs), every fragment read TWO k-steps (>= 1000 cycles) before its first use,
//     from a per-k-step base address + immediate offsets, by inline-asm ds_read_b128 — so hipcc places no partial waits of its own;
//   * four s_waitcnt lgkmcnt per stage (v7: 23), one barrier per stage placed before k-step 2 (v7: before k-step 3);
//   * all 16 LDS-DMA instructions of stage kt+2 issued in k-steps 2 / 3 of stage kt, right after the barrier that releases the buffer.
// The MFMAs, their order per accumulator and every epilogue are v7's: results must be torch.equal with every other variant.
// STATUS: written and compiled in round 3 without GPU time left; the emitted loop was checked by disassembly only
// (instruction counts in the commit message).  NEVER RUN: validate with tests/test_kernels_gpu.py's variant list (add 35) before measuring.
// This header is included inside gemm.hip's anonymous namespace.

template <typename T, bool CONV, int EF>
__global__ __launch_bounds__(256, 1) void gemm_kernel_v10(GemmP p) {
  constexpr int MT = 4, NT = 4, ABL = 0;
  constexpr bool XE = true;
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
  // ---- fragment sets: one per k-step of a stage (4 x (4 W + 4 A fragments) = 128 VGPRs), read TWO k-steps before use by inline-asm
  // ds_read_b128 — the compiler does not track them, so the only LDS waits in the loop are the four written below
  V8 fw[4][NT], fa[4][MT];
  unsigned la[4], lw[4];                     // LDS byte address of fragment 0 of k-step ks in buffer 0; fragment i is + i * 4096, buffer 1 is ^ STAGE_BYTES
  {
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { la[ks] = lds0 + (unsigned)aoff[ks]; lw[ks] = lds0 + (unsigned)boff[ks]; }
  }
  static_assert(STAGE_BYTES == 65536, "the buffer toggle is one address bit");
#define OMG_XRD1(ks_, r_, tog_)                                                                             \
  do {                                                                                                     \
    /* read order = order of first use by the MFMAs (n = NT i + j): W0, A0, W1 .. W(NT-1), A1 .. A(MT-1) */ \
    constexpr bool isA_ = (r_) == 1 || (r_) > NT;                                                          \
    constexpr int idx_ = (r_) <= 1 ? 0 : (r_) <= NT ? (r_) - 1 : (r_) - NT;                                \
    if constexpr (isA_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[ks_][idx_]) : "v"(la[ks_] ^ (tog_)), "n"(idx_ * 4096)); \
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[ks_][idx_]) : "v"(lw[ks_] ^ (tog_)), "n"(idx_ * 4096));             \
  } while (0)
#define OMG_XMM1(ks_, n_) acc[(n_) / NT][(n_) % NT] = Vec<T>::mfma32(fw[ks_][(n_) % NT], fa[ks_][(n_) / NT], acc[(n_) / NT][(n_) % NT])
  // the fragments of k-step ks_ are back when at most `left_` younger LDS reads are still in flight (LDS returns in order); the "+v"
  // operands make every MFMA of the k-step depend on the wait
#define OMG_XWAIT(ks_, left_)                                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #left_ ")"                                                             \
               : "+v"(fw[ks_][0]), "+v"(fw[ks_][1]), "+v"(fw[ks_][2]), "+v"(fw[ks_][3]),                   \
                 "+v"(fa[ks_][0]), "+v"(fa[ks_][1]), "+v"(fa[ks_][2]), "+v"(fa[ks_][3]))
  // k-step ks_: 16 MFMAs; slot s (two MFMAs) carries one fragment read of k-step rks_ (RD_) from the buffer selected by rtog_ and one
  // LDS-DMA of the prepared stage (DMA_: instructions d0_ .. d0_ + 7 into db_)
#define OMG_XKSTEP(ks_, RD_, rks_, rtog_, DMA_, d0_, db_)                                                  \
  do {                                                                                                     \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) {                                                     \
      OMG_XMM1(ks_, 2 * s_);                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (RD_) {                                                                                           \
        switch (s_) {                                                                                      \
          case 0: OMG_XRD1(rks_, 0, rtog_); break; case 1: OMG_XRD1(rks_, 1, rtog_); break;                \
          case 2: OMG_XRD1(rks_, 2, rtog_); break; case 3: OMG_XRD1(rks_, 3, rtog_); break;                \
          case 4: OMG_XRD1(rks_, 4, rtog_); break; case 5: OMG_XRD1(rks_, 5, rtog_); break;                \
          case 6: OMG_XRD1(rks_, 6, rtog_); break; default: OMG_XRD1(rks_, 7, rtog_); break;               \
        }                                                                                                  \
      }                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      OMG_XMM1(ks_, 2 * s_ + 1);                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (DMA_) OMG_DMA((d0_) + s_, db_);                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
    }                                                                                                      \
  } while (0)

  // prologue: stage 0 completely, stage 1 completely, the fragments of k-steps 0 and 1
  OMG_PREP(0);
  OMG_DMAN(0, AB + WB, smem);
  const bool gb_epi = acc_init_bias<T, MT, NT>(p, acc, lane, m0, n0 + wn * (NT * 32));
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (p.dbg & 16) ts1 = __builtin_amdgcn_s_memrealtime();
  OMG_PREP(1);
  if (nk > 1) OMG_DMAN(0, AB + WB, smem + STAGE_BYTES);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    switch (r) { case 0: OMG_XRD1(0, 0, 0u); break; case 1: OMG_XRD1(0, 1, 0u); break; case 2: OMG_XRD1(0, 2, 0u); break; case 3: OMG_XRD1(0, 3, 0u); break;
                 case 4: OMG_XRD1(0, 4, 0u); break; case 5: OMG_XRD1(0, 5, 0u); break; case 6: OMG_XRD1(0, 6, 0u); break; default: OMG_XRD1(0, 7, 0u); break; }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    switch (r) { case 0: OMG_XRD1(1, 0, 0u); break; case 1: OMG_XRD1(1, 1, 0u); break; case 2: OMG_XRD1(1, 2, 0u); break; case 3: OMG_XRD1(1, 3, 0u); break;
                 case 4: OMG_XRD1(1, 4, 0u); break; case 5: OMG_XRD1(1, 5, 0u); break; case 6: OMG_XRD1(1, 6, 0u); break; default: OMG_XRD1(1, 7, 0u); break; }
  }

  // One stage.  Reads run two k-steps ahead: k-steps 0 / 1 fetch k-steps 2 / 3 of the current buffer; then the block joins (stage kt+1
  // has landed, nobody reads `cur` any more); k-steps 2 / 3 fetch k-steps 0 / 1 of the next buffer and carry the 16 LDS-DMA
  // instructions of stage kt+2 into the buffer just released.  Four LDS waits and one barrier per stage.
#define OMG_XSTAGE(HAS1_, HAS2_)                                                                           \
  do {                                                                                                     \
    const unsigned tcur = (unsigned)(kt & 1) * (unsigned)STAGE_BYTES;                                      \
    const unsigned tnxt = tcur ^ (unsigned)STAGE_BYTES;                                                    \
    char* curb = smem + (kt & 1) * STAGE_BYTES;                                                            \
    OMG_XWAIT(0, 8);                                                                                       \
    OMG_XKSTEP(0, true, 2, tcur, false, 0, curb);                                                          \
    OMG_XWAIT(1, 8);                                                                                       \
    OMG_XKSTEP(1, true, 3, tcur, false, 0, curb);                                                          \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
    OMG_XWAIT(2, 0);                                                                                       \
    OMG_XWAIT(3, 0);                                                                                       \
    __builtin_amdgcn_s_barrier();                                                                          \
    if (HAS2_) OMG_PREP(kt + 2);                                                                           \
    OMG_XKSTEP(2, HAS1_, 0, tnxt, HAS2_, 0, curb);                                                         \
    OMG_XKSTEP(3, HAS1_, 1, tnxt, HAS2_, 8, curb);                                                         \
  } while (0)
  int kt = 0;
  for (; kt < nk - 2; ++kt) OMG_XSTAGE(true, true);
  if (kt < nk - 1) { OMG_XSTAGE(true, false); ++kt; }
  OMG_XSTAGE(false, false);
#undef OMG_XSTAGE
#undef OMG_XKSTEP
#undef OMG_XWAIT
#undef OMG_XMM1
#undef OMG_XRD1
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


template <typename T, bool CONV, int EF>
int launch_v10(GemmP p, hipStream_t s, int mrows) {
  constexpr int lds = 2 * (256 + 256) * 64 * 2 + 4 * 8192;
  static bool attr = false;
  if (!attr) {
    attr = true;
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v10<T, CONV, EF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  p.tiles_m = (mrows + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  p.dbg = g_dbg;
  const int grid = p.tile_groups * p.tiles_m * p.tiles_n;
  if (grid <= 0) return OMG_OK;
  OMG_LAUNCH((gemm_kernel_v10<T, CONV, EF>), dim3(grid), dim3(256), lds, s, p);
  return omg_check_launch("gemm_v10");
}
