// gemm_v12.h — the product kernel of the 256 x 256 x 64 tile (variant 25): four waves (2 x 2 of 128 x 128, 256 accumulators in AGPRs), a
// TABLE-DRIVEN ring K loop, a persistent tile walk with the next tile's first two stages in flight under the last stage's MFMAs.
//
// K loop (round 4, was gemm_v11.h): tile, LDS image, XOR swizzle, MFMA order per accumulator and every epilogue are gemm_kernel_v7's — results are
// torch.equal with every other variant (tests/test_kernels_gpu.py) — what differs is WHEN things are issued:
//   * one fragment register set per k-step of a stage (4 x 8 fragments = 128 VGPRs), filled by inline-asm ds_read_b128 from a per-k-step base
//     address + immediate offsets.  hipcc does not track these reads, so the only LDS waits in the loop are the ones placed here: one counted
//     `s_waitcnt lgkmcnt(N)` in front of each k-step (N = the number of younger reads in flight, derived from the table) and one `lgkmcnt(0)` in
//     front of the stage barrier (v7: 23 compiler-placed partial waits per stage);
//   * a stage is 32 SLOTS of two MFMAs; a table gives, per fragment read, per LDS-DMA instruction and for the barrier, the slot it sits in.  The
//     stage bodies are straight-line macro code GENERATED from the table by tools/gen_ksched.py into gemm_v12_sched.inc;
//   * the 160 KB of LDS are FIVE 32 KB half-stage buffers — A current, W current, A next, W next, spare — whose roles rotate every stage (three A
//     buffers cycle, two W buffers swap; five scalar moves).  The A half of stage kt + 2 goes into the spare from the first slot of stage kt on,
//     the W half into the current W buffer behind the barrier: the 16 LDS-DMA instructions of a stage are spread over the WHOLE stage (one per
//     four MFMAs).  The fifth buffer is the XE staging region of the epilogue, which nobody needs inside the loop.
//   Round 4's A/B of ten such tables (profiles/r04_ksched_*.log): the ring gains +6 ... 8 % at K = 5120, +3 ... 4 % on the K = 1280 projections
//   with a residual, 0 on the convolutions; placements inside it do not matter.  Only the winning table (schedule 5) is carried.
// Tile walk (round 5, profiles/r05_exp_v12_ab_*.log — interleaved A/B against the one-tile-per-block form, every form bitwise equal): of a
// 38 - 48 us tile at K = 1280, 7 - 9 us lay OUTSIDE the K loop (prologue, epilogue, dispatch gap).  The ring leaves ALL five LDS buffers free from
// the barrier of a tile's last stage on — 40 MFMAs (~1 us) before the epilogue starts — and that window is used:
//   EF == 2 (residual)   the residual tile of THIS tile's epilogue (res_stage_dma: 32 LDS-DMA instructions per wave) is put in flight behind that
//                        barrier instead of behind the loop (+3 ... 5 %); one tile per block (the residual image owns four of the five buffers,
//                        and carrying the tile walk's scalar state through the 256-VGPR epilogue spilled SGPRs to scratch);
//   EF != 2              persistent: a block walks tiles vb = blockIdx.x, + gridDim.x, ... (grid = the CU count; the XCD-aware tile order is kept
//                        because 256 is a multiple of 8), and the NEXT tile's stages 0 and 1 (32 LDS-DMA instructions per wave) are issued in the
//                        window, i.e. IN FRONT of the epilogue's first store in program order — round 3's persistent kernel issued them behind
//                        the stores and found them queued behind those in the CU's memory pipe.  +2 ... 8 % on the Linear shapes, +0.3 % on the
//                        convolutions.  (A counted vmcnt wait in front of the next tile's first barrier, variant 48 of the experiment, was
//                        within noise of this form and is not carried.)
// Values: the same loads, the same MFMA order per accumulator, the same epilogue code as every other variant — bitwise identical by construction.
// This header is included inside gemm.hip's anonymous namespace.
#include "gemm_v12_sched.inc"

template <typename T, int NT>
OMG_DEV void bias_issue(const GemmP& p, int lane, int m0, int wn0, u32x4 (&rb)[NT][2], u32x4 (&rg)[NT][2]) {
  // the loads of acc_init_bias (gemm_epilogue.h), split from their decode so that they can be issued a tile ahead
  const int hi = lane >> 5;
  const bool fold_gb = fold_group_bias(p);
  const __amdgpu_buffer_rsrc_t rsB = epi_rsrc(p.bias, (long)p.N * 2);
  const __amdgpu_buffer_rsrc_t rsG = epi_rsrc(fold_gb ? p.group_bias + (long)(m0 / p.rows_per_group) * p.ldgb * 2 : nullptr, (long)p.N * 2);
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int c = wn0 + j * 32 + pr * 16 + hi * 8;
      rb[j][pr] = __builtin_amdgcn_raw_buffer_load_b128(rsB, c * 2, 0, 0);
      rg[j][pr] = __builtin_amdgcn_raw_buffer_load_b128(rsG, c * 2, 0, 0);
    }
}
template <typename T, int MT, int NT>
OMG_DEV void bias_apply(const u32x4 (&rb)[NT][2], const u32x4 (&rg)[NT][2], f32x16 (&acc)[MT][NT]) {
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
}

constexpr bool omg_is_next(unsigned tag) { return tag == 1u; }      // stage tags of the generated bodies: 0 = current roles, 1 = next
// the lane id from the exec-mask count instead of a VGPR kept alive since the kernel's first instruction; the empty asm keeps what is derived
// from it where it is written (inside the persistent tile loop everything lane-derived is loop-invariant and would be hoisted in front of it)
OMG_DEV int fresh_lane() {
  int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return l;
}
struct TileV12 { int m0, n0, m_end, grp; };
// virtual block id -> tile, exactly v11's mapping with gridDim.x = the number of tiles
OMG_DEV TileV12 decode_tile_v12(const GemmP& p, int vb, int ntiles) {
  int bid = vb;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_per_group = p.tiles_m * p.tiles_n;
  const int grp = bid / tiles_per_group;
  const int t_in = bid - grp * tiles_per_group;
  // G row tiles x all column tiles form a walk group: the 32 CUs of an XCD run G x (32 / G) tiles at a time and fetch G + 32 / G operand panels
  // for them — 8 x 4 and 4 x 8 both 12 panels per 32 tiles = 0.375 per tile, which IS the measured fabric read volume (2.52 GB per launch of
  // 65536 x 10240 x 1280 against 0.19 GB of operands; round 5 measured G = 4: the same 2.517 GB, the same time; G = 2: 18 panels, 1 ... 3 % slower —
  // profiles/r05_walk_group_ab.log, r05_pmc_traffic_fp16*.json).  The traffic is the price of 32 concurrent tiles per L2, not of the walk order.
  constexpr int G = 8;
  const int per_group = G * p.tiles_n;
  const int gid = t_in / per_group;
  const int first_m = gid * G;
  const int gsz = (p.tiles_m - first_m) < G ? (p.tiles_m - first_m) : G;
  const int r = t_in - gid * per_group;
  const int tm = first_m + (r % gsz);
  const int tn = r / gsz;
  const int m_base = (p.tile_groups > 1) ? grp * p.rows_per_group : 0;
  TileV12 t;
  t.m_end = (p.tile_groups > 1) ? m_base + p.rows_per_group : p.M;
  t.m0 = m_base + tm * 256;
  t.n0 = tn * 256;
  t.grp = grp;
  return t;
}
// the first virtual block id >= vb (in steps of `step`) whose tile is computed (adapter >= 0), or >= ntiles
OMG_DEV int next_tile_v12(const GemmP& p, int vb, int step, int ntiles) {
  if (p.w_adapter_stride == 0 || p.group_adapter == nullptr) return vb;
  while (vb < ntiles) {
    const TileV12 t = decode_tile_v12(p, vb, ntiles);
    if (p.group_adapter[t.grp] >= 0) break;
    vb += step;
  }
  return vb;
}

template <typename T, bool CONV, int EF>
__global__ __launch_bounds__(256, 1) void gemm_kernel_v12(GemmP p) {
  constexpr int MT = 4, NT = 4;
  constexpr bool XE = true;
  constexpr int BM_ = MT * 64, BN_ = NT * 64, BKc = 64;
  constexpr int AB = MT * 2, WB = NT * 2;
  constexpr int A_BYTES = BM_ * BKc * 2;
  constexpr int STAGE_BYTES = (BM_ + BN_) * BKc * 2;
  constexpr int HALF = 32768;
  static_assert(A_BYTES == HALF && STAGE_BYTES == 2 * HALF, "the ring's canonical roles are the two-stage layout");
  constexpr bool EARLY_RES = EF == 2;      // the residual tile in flight behind the last stage's barrier; one tile per block (header)
  constexpr bool PERSIST = EF != 2;        // tile walk + the next tile's stages 0 and 1 issued in the same window
  constexpr bool PREFETCH = PERSIST;
  static_assert(OMG_KS_LAST_TAILS >= 32, "32 prefetch DMAs / 32 residual DMAs");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = p.tile_groups * p.tiles_m * p.tiles_n;
  const int step = PERSIST ? (int)gridDim.x : ntiles;
  const int nk = (p.K + BKc - 1) / BKc;

  const int Ctot = p.C1 + p.C2;
  const long a_bytes = CONV ? (long)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.C1 * 2 : ((long)(p.M - 1) * p.lda + p.K) * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(a_bytes < 0x7fffff00 ? a_bytes : 0x7fffff00), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(CONV && p.X2 ? p.X2 : p.A), 0,
      CONV ? (int)((long)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.C2 * 2) : 0, 0x00020000);
  const long w_bytes = ((long)(p.N - 1) * p.ldw + p.K) * 2;
  __amdgpu_buffer_rsrc_t rsW;

  const int prow = lane >> 3, ppos = lane & 7;
  int voffA[AB], voffW[WB];
  int cb[AB], cy[AB], cx[AB];
  const int dchunk = (ppos ^ ((w & 1) * 4 + (prow >> 1))) * 16;
  const int ldo = w * 1024;
  const int wm = w >> 1, wn = w & 1;
  f32x16 acc[MT][NT];
  using V8 = typename Vec<T>::v8;

  // ---- the tile being computed (m0 .. also feed its epilogue) and, from the head of its last stage on, the next one
  int vb = next_tile_v12(p, (int)blockIdx.x, step, ntiles);
  if (vb >= ntiles) return;
  int m0, n0, m_end;
#define OMG_TILE_SCALARS(t_)                                                                               \
  do {                                                                                                     \
    m0 = (t_).m0; n0 = (t_).n0; m_end = (t_).m_end;                                                        \
    int adapter_ = 0;                                                                                      \
    if (p.group_adapter != nullptr) adapter_ = p.group_adapter[(t_).grp];                                  \
    const char* Wp_ = p.W + (p.w_adapter_stride != 0 ? (long)adapter_ * p.w_adapter_stride * 2 : 0);       \
    rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp_, 0, (int)w_bytes, 0x00020000);                      \
  } while (0)
  // row block i_ of the A operand / of the W operand of the tile at (m0, n0): the lane's source offsets of gemm_kernel_v11
#define OMG_ADDR_A(i_)                                                                                     \
  do {                                                                                                     \
    const int r_ = (w + (i_) * 4) * 8 + prow;                                                              \
    int gm_ = m0 + r_; if (gm_ > m_end - 1) gm_ = m_end - 1;                                               \
    if constexpr (CONV) {                                                                                  \
      const int hw_ = p.Hout * p.Wout;                                                                     \
      const int b_ = gm_ / hw_; const int rem_ = gm_ - b_ * hw_;                                           \
      cb[i_] = b_; cy[i_] = rem_ / p.Wout; cx[i_] = rem_ - cy[i_] * p.Wout;                                \
      voffA[i_] = 0;                                                                                       \
    } else {                                                                                               \
      cb[i_] = cy[i_] = cx[i_] = 0;                                                                        \
      voffA[i_] = (int)((long)gm_ * p.lda * 2) + dchunk;                                                   \
    }                                                                                                      \
  } while (0)
#define OMG_ADDR_W(i_)                                                                                     \
  do {                                                                                                     \
    const int r_ = (w + (i_) * 4) * 8 + prow;                                                              \
    int gn_ = n0 + r_; if (gn_ > p.N - 1) gn_ = p.N - 1;                                                   \
    voffW[i_] = (int)((long)gn_ * p.ldw * 2) + dchunk;                                                     \
  } while (0)
  {
    const TileV12 t = decode_tile_v12(p, vb, ntiles);
    OMG_TILE_SCALARS(t);
  }
#pragma unroll
  for (int i = 0; i < AB; ++i) OMG_ADDR_A(i);
#pragma unroll
  for (int i = 0; i < WB; ++i) OMG_ADDR_W(i);

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
  V8 fw[4][NT], fa[4][MT];
  unsigned la[4], lw[4];
#define OMG_LA_LW(lane_)                                                                                   \
  do {                                                                                                     \
    const unsigned lds0_ = (unsigned)(unsigned long)(lds_ptr_t)smem;                                       \
    const int hi_ = (lane_) >> 5, l31_ = (lane_) & 31;                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                     \
      const unsigned sw_ = (unsigned)(((ks * 2 + hi_) ^ ((l31_ >> 1) & 7)) << 4);                          \
      la[ks] = lds0_ + (unsigned)((wm * (MT * 32) + l31_) * 128) + sw_;                                    \
      lw[ks] = lds0_ + (unsigned)((wn * (NT * 32) + l31_) * 128) + sw_;                                    \
    }                                                                                                      \
  } while (0)
  OMG_LA_LW(lane);
  int sa_cur, sw_cur, sa_nxt, sw_nxt, s_sp;
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
#define OMG_XRD1(ks_, r_, tog_)                                                                            \
  do {                                                                                                     \
    constexpr bool isA_ = (r_) == 1 || (r_) > NT;                                                          \
    constexpr int idx_ = (r_) <= 1 ? 0 : (r_) <= NT ? (r_) - 1 : (r_) - NT;                                \
    const unsigned ad_ = (isA_ ? la[ks_] : lw[ks_]) + (unsigned)(omg_is_next(tog_) ? (isA_ ? sa_nxt : sw_nxt) : (isA_ ? sa_cur : sw_cur)); \
    if constexpr (isA_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[ks_][idx_]) : "v"(ad_), "n"(idx_ * 4096)); \
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[ks_][idx_]) : "v"(ad_), "n"(idx_ * 4096));  \
  } while (0)
#define OMG_XMM1(ks_, n_) acc[(n_) / NT][(n_) % NT] = Vec<T>::mfma32(fw[ks_][(n_) % NT], fa[ks_][(n_) / NT], acc[(n_) / NT][(n_) % NT])
// the last stage's MFMA, anchored: an empty asm that "modifies" the accumulator keeps the MFMA in its slot.  Inside a K-loop stage the fragment
// reads of the NEXT stage (asm volatile, overwriting the set an MFMA reads) do that; the last stage has none, and in the persistent form without
// hooks hipcc moved all 64 MFMAs behind the stage's last wait (seen in the ISA: 16 reads, the barrier, then 64 MFMAs back to back)
#define OMG_XMML(ks_, n_) do { OMG_XMM1(ks_, n_); asm volatile("" : "+a"(acc[(n_) / NT][(n_) % NT])); } while (0)
#define OMG_XWAIT(ks_, left_)                                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #left_ ")"                                                             \
               : "+v"(fw[ks_][0]), "+v"(fw[ks_][1]), "+v"(fw[ks_][2]), "+v"(fw[ks_][3]),                   \
                 "+v"(fa[ks_][0]), "+v"(fa[ks_][1]), "+v"(fa[ks_][2]), "+v"(fa[ks_][3]))

  // ---- hooks of the last stage (gemm_v12_sched.inc).  HEAD (24, in front of the barrier, under the MFMAs of k-steps 0 and 1): the next tile's
  // scalars and the lane's 16 source offsets — voffA / voffW / cb / cy / cx are dead from the last DMA of the tile on, the epilogue keeps its
  // own copies (e_m0, e_n0, e_mend).  TAIL (40, behind the barrier): 0..31 the residual tile (EF == 2) or the next tile's stages 0 and 1.
  int e_m0 = 0, e_n0 = 0, e_mend = 0;
  int vbn = 0;
  int lane_l = 0; (void)lane_l;
  bool has_next = false;
  u32x4 rb[NT][2], rg[NT][2];
#define OMG_HEAD(n_)                                                                                       \
  do {                                                                                                     \
    if constexpr (PREFETCH) {      /* unconditional on purpose (without a next tile: this tile again) — conditional writes would merge with */ \
      if constexpr ((n_) == 0) {   /* the old values and keep them alive round the whole loop */           \
        const TileV12 t_ = decode_tile_v12(p, has_next ? vbn : vb, ntiles);                                \
        OMG_TILE_SCALARS(t_);                                                                              \
      } else if constexpr ((n_) >= 1 && (n_) <= AB) {                                                      \
        OMG_ADDR_A((n_) >= 1 && (n_) <= AB ? (n_) - 1 : 0);                                                \
      } else if constexpr ((n_) > AB && (n_) <= AB + WB) {                                                 \
        OMG_ADDR_W((n_) > AB && (n_) <= AB + WB ? (n_) - 1 - AB : 0);                                      \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)
#define OMG_TAIL(n_)                                                                                       \
  do {                                                                                                     \
    if constexpr (EARLY_RES) {                                                                             \
      if constexpr ((n_) == 0) lane_l = fresh_lane();                                /* the address arithmetic stays HERE: computed from `lane` */ \
      if constexpr ((n_) < 32) {                                                     /* it is loop-invariant and would sit in registers through the K loop */ \
        constexpr int i_ = (n_) >> 3, u_ = (n_) & 7;                                                       \
        const int rr_ = u_ * 4 + (lane_l >> 4);                                                            \
        const int gm_ = e_m0 + wm * 128 + i_ * 32 + rr_;                                                   \
        const int col_ = e_n0 + wn * 128 + (((lane_l & 15) ^ (rr_ & 15)) << 3);                            \
        const int off_ = (gm_ < e_mend && col_ < p.N) ? (gm_ * (int)p.ldr + col_) * 2 : EPI_OOB;           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsRes, (lds_ptr_t)(smem + w * 32768 + i_ * 8192 + u_ * 1024), 16, off_, 0, 0, 0); \
      }                                                                                                    \
    }                                                                                                      \
    if constexpr (PREFETCH) {                                                                              \
      if constexpr ((n_) == 0) OMG_PREP(0);                                                                \
      if constexpr ((n_) == AB + WB) OMG_PREP(1);                                                          \
      if constexpr ((n_) < AB + WB) {                                                                      \
        if (has_next) OMG_DMA((n_) < AB + WB ? (n_) : 0, smem);                                            \
      } else if constexpr ((n_) < 2 * (AB + WB)) {                                                         \
        if (has_next && nk > 1) OMG_DMA((n_) >= AB + WB && (n_) < 2 * (AB + WB) ? (n_) - (AB + WB) : 0, smem + STAGE_BYTES); \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)
  const __amdgpu_buffer_rsrc_t rsRes = epi_rsrc(EARLY_RES ? p.residual : nullptr, ((long)(p.M - 1) * p.ldr + p.N) * 2);
  const bool fold_gb = fold_group_bias(p);
  const bool gb_epi = p.group_bias != nullptr && !fold_gb;
  (void)rsRes;

  // ---- the first tile's stage 0 and bias
  OMG_PREP(0);
  OMG_DMAN(0, AB + WB, smem);
  bias_issue<T, NT>(p, lane, m0, n0 + wn * (NT * 32), rb, rg);
  bias_apply<T, MT, NT>(rb, rg, acc);
  bool first = true;
  for (;;) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (!PREFETCH || first) {
      OMG_PREP(1);
      if (nk > 1) OMG_DMAN(0, AB + WB, smem + STAGE_BYTES);
    }
    sa_cur = 0; sw_cur = HALF; sa_nxt = 2 * HALF; sw_nxt = 3 * HALF; s_sp = 4 * HALF;
    { const unsigned tnxt = 1u; (void)tnxt; OMG_KS_PROLOGUE_5(); }
    // the tile after this one, known before the last stage starts
    e_m0 = m0; e_n0 = n0; e_mend = m_end;
    if constexpr (PERSIST) { vbn = next_tile_v12(p, vb + step, step, ntiles); has_next = vbn < ntiles; }

    int kt = 0;
#define OMG_KS_VARS const unsigned tcur = 0u; const unsigned tnxt = 1u; (void)tnxt; (void)tcur
#define OMG_KS_ROT do { const int t_ = sa_cur; sa_cur = sa_nxt; sa_nxt = s_sp; s_sp = t_; const int u_ = sw_cur; sw_cur = sw_nxt; sw_nxt = u_; } while (0)
    for (; kt < nk - 2; ++kt) { OMG_KS_VARS; OMG_KS_STAGE_5(true, true); OMG_KS_ROT; }
    if (kt < nk - 1) { OMG_KS_VARS; OMG_KS_STAGE_5(true, false); OMG_KS_ROT; ++kt; }
    { OMG_KS_VARS; OMG_KS_LAST_5(); }

    // the epilogue's lane constants (column predicates, XE / residual LDS addresses) are tile-invariant: inside the persistent loop LICM would
    // hoist them in front of it and carry ~50 registers through the K loop (seen as scratch spills) — recompute them per tile instead
    const int lane_e = PERSIST ? fresh_lane() : lane;
    if constexpr (EF == 2) {
      if constexpr (!EARLY_RES) res_stage_dma(p, smem + w * 32768, lane_e, e_m0 + wm * 128, e_n0 + wn * 128, e_mend);
      epilogue_direct<T, MT, NT, XE, EF>(p, acc, lane_e, e_m0 + wm * (MT * 32), e_n0 + wn * (NT * 32), e_mend, gb_epi, smem + 2 * STAGE_BYTES + w * 8192,
                                         smem + w * 32768);
    } else
    epilogue_direct<T, MT, NT, XE, EF>(p, acc, lane_e, e_m0 + wm * (MT * 32), e_n0 + wn * (NT * 32), e_mend, gb_epi, smem + 2 * STAGE_BYTES + w * 8192);
    if (!PERSIST || !has_next) break;
    vb = vbn;
    first = false;
    { const int lane_n = fresh_lane(); OMG_LA_LW(lane_n); }      // not carried through the epilogue
    if constexpr (!PREFETCH) {
      const TileV12 t = decode_tile_v12(p, vb, ntiles);
      OMG_TILE_SCALARS(t);
#pragma unroll
      for (int i = 0; i < AB; ++i) OMG_ADDR_A(i);
#pragma unroll
      for (int i = 0; i < WB; ++i) OMG_ADDR_W(i);
      OMG_PREP(0);
      OMG_DMAN(0, AB + WB, smem);
    }
    bias_issue<T, NT>(p, lane, m0, n0 + wn * (NT * 32), rb, rg);
    // at the loop's BOTTOM on purpose: decoded at the top, the loads would meet the first tile's (just issued) on the loop header and the
    // compiler's wait would be vmcnt(0) on both paths — here it sees the loads and the epilogue's stores behind them in one straight line
    bias_apply<T, MT, NT>(rb, rg, acc);
  }
#undef OMG_KS_VARS
#undef OMG_KS_ROT
#undef OMG_HEAD
#undef OMG_TAIL
#undef OMG_DMAR
#undef OMG_XWAIT
#undef OMG_XMM1
#undef OMG_XMML
#undef OMG_XRD1
#undef OMG_SB
#undef OMG_PREP
#undef OMG_DMA
#undef OMG_DMAN
#undef OMG_ADDR_A
#undef OMG_LA_LW
#undef OMG_ADDR_W
#undef OMG_TILE_SCALARS
}

template <typename T, bool CONV, int EF>
int launch_v12(GemmP p, hipStream_t s, int mrows) {
  constexpr int lds = 2 * (256 + 256) * 64 * 2 + 4 * 8192;
  static bool attr = false;
  if (!attr) {
    attr = true;
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v12<T, CONV, EF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  p.tiles_m = (mrows + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  p.dbg = g_dbg;
  const int ntiles = p.tile_groups * p.tiles_m * p.tiles_n;
  if (ntiles <= 0) return OMG_OK;
  int grid = ntiles;
  if (EF != 2) {      // persistent: one block per CU; a multiple of 8 keeps a block's tiles on one XCD's share of the tile order
    int cus = num_cus() & ~7;
    if (g_dbg & 0x10000) cus = 8;       // tests only: eight blocks, so that a small problem makes every block walk several tiles
    if (grid > cus && cus > 0) grid = cus;
  }
  OMG_LAUNCH((gemm_kernel_v12<T, CONV, EF>), dim3(grid), dim3(256), lds, s, p);
  return omg_check_launch("gemm_v12");
}
template <typename T, bool CONV>
int launch_v12_form(const GemmP& p, hipStream_t s, int mrows) {
  const bool gb_rows = p.group_bias != nullptr && p.rows_per_group % 256 != 0;       // == !fold_group_bias
  if (p.act == OMG_ACT_GEGLU) return launch_v12<T, CONV, 3>(p, s, mrows);
  if (gb_rows || p.act == OMG_ACT_SILU) return launch_v12<T, CONV, 4>(p, s, mrows);
  // (round 5, profiles/r05_res_form_ab_*.log: sending residual launches to the PERSISTENT generic form 4 — residual by register-direct loads —
  // instead of this one-tile-per-block LDS-staged form was 1 ... 4 % slower on seven of nine shapes: the staged residual wins over the tile walk)
  if (p.residual != nullptr) return launch_v12<T, CONV, 2>(p, s, mrows);
  return launch_v12<T, CONV, 1>(p, s, mrows);
}
