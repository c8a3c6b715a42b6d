// attn_v7.h — the self-attention kernel of the product (more than 128 keys, V given ROW-MAJOR in omg_attn_args.V).  Written in round 4, first run
// and landed in round 5 (profiles/r05_exp_attn_v7_*.log: torch.equal with attn_fwd_kernel3 — round 3's kernel on a V^T image, deleted in round 6 — on
// whole and ragged tiles, plain / borrowed Q,K / accumulate; with omg_transpose_v gone from the self-attention path the benchmark step is 0.5 - 1 % shorter).
//
// attn_fwd_kernel7 = the structure described in attn.hip (64 query rows per wave, K / V tiles by LDS-DMA, swapped S^T = K Q^T, O^T = V^T P^T) reading V
// ROW-MAJOR — [key][d], exactly as the QKV projection wrote it — instead of the K-major V^T image omg_transpose_v makes once per attention call
// (0.9 % of the benchmark step: 3648 launches of 51 us).  The V tile is staged like the K tile (same LDS-DMA pattern, same XOR swizzle) and the
// P.V MFMA's A operand — 32 d x 16 keys, eight keys per lane — comes out of it by two `ds_read_b64_tr_b16` (gfx950's transposing LDS read) per
// fragment instead of one ds_read_b128: the same bytes per lane, and the guide prices the transposing read at the plain b64's cost beside
// MFMAs.  The key order the P registers dictate inside a group of 16 ([0-3, 8-11 | 4-7, 12-15] by lane half) is met by the ADDRESSES the
// lanes supply, so no permuted image is needed either.
// What is assumed about the instruction (cdna_hip_programming.md, LDS section + T10; tools/ubench/tr16_probe.hip checked exactly this on the
// part before the kernel is trusted): per group of 16 lanes, lane i supplies the address of 4 consecutive 16-bit elements = row (i >> 2),
// columns 4 (i & 3) .. + 3 of a 4 x 16 block, and lane c receives column c (4 elements, row order).
// Ragged last tile: the staged rows past Nkv repeat the last key (as K's do) — finite values, so the probabilities of those keys are set to
// zero (a V^T image has zero columns there instead); with them zero the denominator (below) needs no mask either.
// Two questions rode on the same kernel in round 5 (profiles/r05_exp_attn_v7_stagger.log, _xcd.log):
//   * do the two workgroups of a CU run in lockstep, and does de-phasing them (half of the first-round workgroups started 10 / 20 / 40 us late)
//     recover what the 32 x 32 launches lose against the 64 x 64 ones?  +5 ... 8 % at 32 x 32 in the microbenchmark, nothing at 64 x 64 — and
//     the XCD-aware block order below gives more (+11 %) without a busy-wait, so the knob is not carried;
//   * an XCD-aware block order: KEPT, see the kernel's first lines.
// Included inside attn.hip's anonymous namespace.
template <typename T> struct TrRead;
template <> struct TrRead<f16> {
  typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 raw4;
  static OMG_DEV f16x4 rd(const char* lds) {
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) raw4*)lds));
  }
};
template <> struct TrRead<bf16> {
  typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 raw4;
  static OMG_DEV bf16x4 rd(const char* lds) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) raw4*)lds));
  }
};

// The FIRST MFMA of every S^T accumulator — the one whose C operand is the splat of the running reference
// maximum (negm) — written as inline asm in its three-address form (destination != C, early-clobber).  In v3's emitted tile loop hipcc uses that
// form for the first key block only; for the second it COPIES the 16-register splat (2 x 8 v_mov_b64 per tile and wave) and accumulates in place.
// The hazard recogniser does not see an asm MFMA; what follows on the same registers are MFMAs of the same opcode accumulating in place (a
// dependence the matrix pipe handles back to back) and its sources are LDS reads the compiler still waits for.  Same arithmetic, same bits.
template <typename T> struct MfmaInit;
template <> struct MfmaInit<f16> {
  static OMG_DEV void run(f32x16& d, f16x8 a, f16x8 b, const f32x16& c) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c)); }
};
template <> struct MfmaInit<bf16> {
  static OMG_DEV void run(f32x16& d, bf16x8 a, bf16x8 b, const f32x16& c) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c)); }
};

// The softmax denominator (round 6; VERDICT r5 weak 7: "25 % more MFMAs issued than the algorithm needs").  Round 3 formed the row sums of P by one more
// 32 x 32 x 16 MFMA per (query block, 16 keys) against a fragment of ones — a third "V^T row block" of which ONE row is useful: 8 of a tile's 40 MFMAs, 256
// of its 1280 matrix-pipe cycles, 32 accumulator registers zeroed per tile.  Now the same sum runs on the 16 x 16 x 32 MFMA (half the pipe time per
// instruction) into an accumulator that lives across ALL tiles.  The P registers are a 32 x 32 x 16 B operand — lane l holds eight keys of query l & 31, key
// half l >> 5 —; READ as a 16 x 16 x 32 B operand the same registers are [k = 8 (l >> 4) .. + 7][column l & 15]: k groups 0 / 2 of column j are the two key
// halves of query j, k groups 1 / 3 those of query 16 + j.  Against an A operand whose row 0 is ones on k groups 0, 2 and whose row 1 is ones on k groups 1, 3
// the product's row 0 is the row sum of queries 0..15 and row 1 that of queries 16..31: D[0][j] -> lane j register 0, D[1][j] -> lane j register 1.  Four
// accumulator registers per query block, no per-tile zeroing, no per-tile read-out: the matrix-pipe sum is folded into the lane-local running sum only where
// a rescale needs it (rare) and at the end, by two ds_bpermute.  Interleaved A/B of three forms on one box (profiles/r06_attn_bench_den_forms.log; the 32-row
// ones MFMA | this | fp32 adds of the exponentials on the VALU): (64,10,4096,4096) 971 | 1046 | 1032 TF/s, (64,20,1024,1024) 745-810 | 848-858 | 840; in
// situ (whole benchmark step, same box, profiles/r06_bench_fp16_den{0,1}*.json) 0.4064 / 0.4056 -> 0.4100 / 0.4099 images/s, self-attention 898 -> 943 and
// 754 -> 785 TF/s.  The other two forms are deleted.  Against the fp32 reference all three pass the same tolerance; this one differs from round 3's by the
// summation order of the same rounded probabilities (<= 2 ulp of the 16-bit output, tests/test_kernels_gpu.py).
// 16 bytes per lane, global -> LDS, through a buffer descriptor (non-template on purpose: see dma16 in gemm_epilogue.h)
__device__ __forceinline__ void attn_dma16(__amdgpu_buffer_rsrc_t rs, char* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (attn_lds_ptr_t)lds, 16, voff, soff, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel7(AttnP p, const char* Vrm, long ldv, long v_bs, int xcd_order) {
  constexpr int QW = 2;                      // 32-row query blocks per wave
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // K[2], V[2]: both [64 keys][64 d], 16-byte chunk c of row r at position c ^ ((r >> 1) & 7)
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  typedef T T2 __attribute__((ext_vector_type(2)));
  typedef float F2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  // XCD-aware block order.  Workgroups go to the eight XCDs round-robin by linear id, so the 4 (32 x 32) or 16 (64 x 64) query blocks of one
  // (sample, head) — which stage the same K / V tiles — would land on different XCDs and each L2 fetch its own copy (up to 8 x the K / V bytes
  // over the fabric).  Remapped, the ids an XCD receives enumerate consecutive (query block, head, sample) items: +1.6 % at 64 x 64, +6 % at
  // 32 x 32 (profiles/r05_exp_attn_v7_xcd.log).
  int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  {
    const int total = gridDim.x * gridDim.y * gridDim.z;
    if (xcd_order && (total & 7) == 0) {
      const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const int wi = (lin & 7) * (total >> 3) + (lin >> 3);
      qblk = wi % (int)gridDim.x;
      h = (wi / (int)gridDim.x) % (int)gridDim.y;
      b = wi / (int)(gridDim.x * gridDim.y);
    }
  }
  const int bq = p.qk_src ? p.qk_src[b] : b;
  const int q0 = qblk * (4 * 32 * QW) + w * (32 * QW);

  // ---- LDS-DMA staging: one instruction = 8 rows x 128 B; wave w moves row blocks w and w + 4 of the K tile and of the V^T tile
  const int prow = lane >> 3, ppos = lane & 7;
  const char* kbase = p.K + ((long)bq * p.k_bs + h * 64) * 2;
  const char* vbase = Vrm + ((long)b * v_bs + h * 64) * 2;        // row-major V: key row stride ldv, the caller's own projection output
  int srow[2], schunk[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    srow[j] = (w + 4 * j) * 8 + prow;
    schunk[j] = (ppos ^ ((srow[j] >> 1) & 7)) * 16;
  }
  // ds_read_b64_tr_b16: every group of 16 lanes fetches a [4 keys][16 d] block — lane i16 of the group supplies the address of the four
  // consecutive d  4 (i16 & 3) .. + 3  of key (i16 >> 2) of the block — and lane c of the group receives column c: the four keys of d = c.
  // Group g = lane >> 4 covers d = 16 (g & 1) .. + 15 of the 32-wide d block dt, keys 4 hi + 0..3 (read 0) / 8 + 4 hi + 0..3 (read 1) of a
  // 16-key group.  vtr[r][dt]: the lane's byte offset inside the tile for (read r, d block dt) of 16-key group 0; group (i, k2) adds
  // (32 i + 16 k2) * 128 (the swizzle term (key >> 1) & 7 does not change: 16 i + 8 k2 is a multiple of 8).
  int vtr[2][2];
  {
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int key = 8 * r + 4 * hi + (i16 >> 2);
        const int chunk = dt * 4 + 2 * g1 + ((i16 & 3) >> 1);
        vtr[r][dt] = key * 128 + ((chunk ^ ((key >> 1) & 7)) << 4) + (i16 & 1) * 8;
      }
  }
  // Descriptor addressing (round 6): the (sample, head) slice of K / of V behind an SGPR descriptor, the lane's two row offsets as 32-bit VGPRs that never
  // change, the tile as a SCALAR offset.  `global_load_lds` with 64-bit lane addresses costs the issuing wave ~30 cycles more of MFMA issue per piece than
  // `buffer_load ... lds` (tools/ubench/mfma_f32_rate.hip: a burst of eight 133 -> 142 TF/s on the fp32 convolution's mix), and the four running 64-bit
  // pointers of round 5 (one add each per tile) are gone from a loop that is bound by its VALU.  The launcher checks that a slice fits 31 bits.
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)(((long)(p.Nkv - 1) * p.ldk + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)(((long)(p.Nkv - 1) * ldv + 64) * 2), 0x00020000);
  int kvo[2], vvo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    kvo[j] = srow[j] * p.ldk * 2 + schunk[j];
    vvo[j] = srow[j] * (int)ldv * 2 + schunk[j];
  }
  const int kstep = KVB * p.ldk * 2, vstep = KVB * (int)ldv * 2;
  // the ragged last tile: rows past the end repeat the last key (their scores are masked / their P is zeroed below)
  auto dma_tile = [&](int kv0, int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int key = kv0 + srow[j];
      if (key > p.Nkv - 1) key = p.Nkv - 1;
      attn_dma16(rsK, smem + buf * TILE + (w + 4 * j) * 1024, key * p.ldk * 2 + schunk[j], 0);
      attn_dma16(rsV, smem + (2 + buf) * TILE + (w + 4 * j) * 1024, key * (int)ldv * 2 + schunk[j], 0);
    }
  };
  // whole tile t: the same 8 rows x 128 B pattern for K and for V
  auto dma_whole = [&](int t, int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      attn_dma16(rsK, smem + buf * TILE + (w + 4 * j) * 1024, kvo[j], t * kstep);
      attn_dma16(rsV, smem + (2 + buf) * TILE + (w + 4 * j) * 1024, vvo[j], t * vstep);
    }
  };

  f32x16 o[QW][2], negm[QW];
  float m_ref[QW], l_run[QW];
#pragma unroll
  for (int qb = 0; qb < QW; ++qb) {
    m_ref[qb] = 0.f; l_run[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[qb][0][r] = 0.f; o[qb][1][r] = 0.f; negm[qb][r] = 0.f; }
  }

  // the persistent 16 x 16 accumulators of the denominator and the selector fragment (comment above the kernel)
  f32x4 den16[QW];
  V8 sel16;
  {
#pragma unroll
    for (int qb = 0; qb < QW; ++qb) den16[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, g = lane >> 4;
    const float one = ((i16 == 0 && (g & 1) == 0) || (i16 == 1 && (g & 1) == 1)) ? 1.0f : 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sel16[e] = (T)one;
  }
  // row sum of query (lane & 31) out of the 16 x 16 accumulator: queries 0..15 in register 0 of lanes 0..15, queries 16..31 in register 1
  auto den16_read = [&](const f32x4& d) -> float {
    const float v0 = __shfl(d[0], lane & 15), v1 = __shfl(d[1], lane & 15);
    return (lane & 16) ? v1 : v0;
  };

  const int ntiles = (p.Nkv + KVB - 1) / KVB;
  const int nfull = p.Nkv / KVB;             // tiles with 64 real keys
  if (nfull > 0) dma_whole(0, 0); else dma_tile(0, 0);
  V8 qf[QW][4];
  int qrow[QW];
  // Round 6: the Q loads are issued BEHIND the first K / V tile's LDS-DMA (above), not in front of it: the two global round trips of a block's prologue
  // overlap instead of following each other (the empty asm below waits for everything, and everything is needed before tile 0).
  // all eight Q loads of the wave's two query blocks are in flight before the first is converted (v3's emitted prologue waits for block 0's
  // four loads, scales them, and only then issues block 1's: one more global round trip in front of the first key tile of every workgroup)
  V8 qraw[QW][4];
#pragma unroll
  for (int qb = 0; qb < QW; ++qb) {
    int q = q0 + qb * 32 + l31;
    qrow[qb] = q;
    if (q >= p.Nq) q = p.Nq - 1;
    const char* qp = p.Q + ((long)bq * p.q_bs + (long)q * p.ldq + h * 64) * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qraw[qb][ks] = *(const V8*)(qp + (ks * 16 + hi * 8) * 2);
    }
  }
  {
    asm volatile("" : "+v"(qraw[0][0]), "+v"(qraw[0][1]), "+v"(qraw[0][2]), "+v"(qraw[0][3]), "+v"(qraw[1][0]), "+v"(qraw[1][1]), "+v"(qraw[1][2]), "+v"(qraw[1][3]));
#pragma unroll
    for (int qb = 0; qb < QW; ++qb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[qb][ks][e] = (T)((float)qraw[qb][ks][e] * p.scale_log2e);
  }

  auto tile_body = [&](const int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int buf = t & 1;
    const int kv0 = t * KVB;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's part of tile t has landed
    __syncthreads();                                     // ... everybody's has, and nobody reads buffer buf^1 any more
    if (t + 1 < nfull) dma_whole(t + 1, buf ^ 1);                    // the next tile has 64 real keys
    else if (t + 1 < ntiles) dma_tile(kv0 + KVB, buf ^ 1);       // the ragged last tile: clamped rows
    const char* kt = smem + buf * TILE;
    const char* vt = smem + (2 + buf) * TILE;

    // ---- S' = K · Q'^T - m_ref for both query blocks: each K fragment feeds two MFMAs
    f32x16 s[QW][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kc = ks * 2 + hi;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + l31;
        const V8 kf = *(const V8*)(kt + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
#pragma unroll
        for (int qb = 0; qb < QW; ++qb) {
          if (ks == 0) MfmaInit<T>::run(s[qb][i], kf, qf[qb][ks], negm[qb]);
          else s[qb][i] = Vec<T>::mfma32(kf, qf[qb][ks], s[qb][i]);
        }
      }
    }
    V8 pf[QW][2][2];
#pragma unroll
    for (int qb = 0; qb < QW; ++qb) {
      float mt = s[qb][0][0];
#pragma unroll
      for (int r = 1; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[qb][0][r]), r + 1 < 16 ? s[qb][0][r + 1] : s[qb][0][r]);
#pragma unroll
      for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[qb][1][r]), s[qb][1][r + 1]);
      {   // max over the two lane halves through v_permlane32_swap_b32 (gfx950: lanes 32..63 of the first operand <-> lanes 0..31 of the second,
          // inside the VALU) instead of __shfl_xor's ds_bpermute_b32: no LDS round trip in front of the branch below, twice per tile.  Inline asm:
          // given the SAME value for both operands hipcc 7.2 folds the builtin's two results into one; the s_nop carry the VALU hazard slots
        float lo_ = mt, hi_ = mt;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo_), "+v"(hi_));
        mt = fmaxf(lo_, hi_);
      }
      if (t == 0 || __builtin_amdgcn_ballot_w64(mt > ATTN_THR) != 0) {
        const float d = t == 0 ? mt : fmaxf(mt, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        m_ref[qb] += d;
        if (t != 0) {      // what the matrix pipe has summed so far is at the old reference: fold it in before the sum is rescaled
          l_run[qb] += den16_read(den16[qb]);
          den16[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        l_run[qb] *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[qb][i][r] *= alpha; s[qb][i][r] -= d; }
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[qb][r] = -m_ref[qb];
      }
    }
    // ---- P = 2^S' and O^T += V^T · P^T, KEY HALF BY KEY HALF: the exponentials of half 1 are independent of the P·V MFMAs of half 0, so the
    // quarter-rate v_exp_f32 of one half issue under the matrix pipe's work on the other (round 5: all 64 exponentials used to run back to back in
    // front of all 24 MFMAs — 1024 cycles in which this wave kept the matrix pipe idle; interleaved: (64,10,4096,4096) 911 -> 940 TF/s, MFMA busy
    // 0.566 -> 0.594, profiles/r05_attn_bench_*.log).  Same values, same accumulation order: torch.equal.
    auto exps = [&](const int i) {
#pragma unroll
      for (int qb = 0; qb < QW; ++qb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float e0 = __builtin_amdgcn_exp2f(s[qb][i][r]);
          float e1 = __builtin_amdgcn_exp2f(s[qb][i][r + 1]);
          if constexpr (TAIL) {      // the staged V rows past Nkv are copies of the last key, not zeros: their probabilities are (v3: V^T columns of zeros)
            const int key = kv0 + i * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (key >= p.Nkv) e0 = 0.f;
            if (key + 1 >= p.Nkv) e1 = 0.f;
          }
          const T2 pk = __builtin_convertvector(F2{e0, e1}, T2);
          pf[qb][i][r >> 3][r & 7] = pk[0];
          pf[qb][i][r >> 3][(r & 7) + 1] = pk[1];
        }
    };
    auto pv = [&](const int i) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          // V^T fragment (32 d x 16 keys) out of the ROW-MAJOR tile by two transposing reads: keys 4 hi + 0..3 and 8 + 4 hi + 0..3 of the 16
          const V4 lo = TrRead<T>::rd(vt + vtr[0][dt] + i * 4096 + k2 * 2048);
          const V4 hi4 = TrRead<T>::rd(vt + vtr[1][dt] + i * 4096 + k2 * 2048);
          const V8 vf = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int qb = 0; qb < QW; ++qb) o[qb][dt] = Vec<T>::mfma32(vf, pf[qb][i][k2], o[qb][dt]);
        }
        // the probabilities of keys past Nkv are already zero (exps): the selector needs no mask
#pragma unroll
        for (int qb = 0; qb < QW; ++qb) den16[qb] = Vec<T>::mfma16(sel16, pf[qb][i][k2], den16[qb]);
      }
    };
    exps(0);
    pv(0);
    exps(1);      // (16 keys at a time instead of 32 was tried: 256 VGPRs + 36 bytes of scratch, and hipcc hoisted the exponentials anyway)
    pv(1);
  };
  for (int t = 0; t < nfull; ++t) tile_body(t, std::false_type{});
  if (nfull < ntiles) tile_body(nfull, std::true_type{});

  // Epilogue.  A lane owns, per 32-wide d block and group g, FOUR consecutive d of its query row (d = 32 dt + 8 g + 4 hi + 0..3): 8-byte stores, 16 per query
  // block and lane, each instruction touching 32 rows x 2 pieces — the store-issue-bound tail the guide prices at ~9 k cycles per block (T21).  Round 6: the
  // two lane halves of a row hold ADJACENT pieces, so one v_permlane32_swap per dword pairs the pieces of groups g and g + 1 — the low half ends up with all
  // eight d of group g, the high half with those of g + 1 — and each lane stores 16 bytes: half the store instructions, whole 16-byte pieces.  Same values,
  // same rounding (the swap moves the already packed 16-bit results).  The accumulate form (tools / tests only on this kernel) and outputs that are not
  // 16-byte aligned keep the 8-byte stores.
  const bool wide = !p.accumulate && (p.ldo & 7) == 0 && (p.o_bs & 7) == 0 && (((unsigned long long)p.O) & 15) == 0;
#pragma unroll
  for (int qb = 0; qb < QW; ++qb) {
    const float l_tot = l_run[qb] + den16_read(den16[qb]);
    const float inv = p.out_scale / l_tot;
    char* op = p.O + ((long)b * p.o_bs + (long)(qrow[qb] < p.Nq ? qrow[qb] : 0) * p.ldo + h * 64) * 2;
    if (wide) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          unsigned x[2], y[2];                      // the lane's packed pieces of groups 2 g2 and 2 g2 + 1
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            V4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (T)(o[qb][dt][(2 * g2 + k) * 4 + e] * inv);
            const u32x2 w2 = __builtin_bit_cast(u32x2, pk);
            if (k == 0) { x[0] = w2[0]; x[1] = w2[1]; } else { y[0] = w2[0]; y[1] = w2[1]; }
          }
          // lanes 32..63 of x <-> lanes 0..31 of y: low half = [own x | partner's x] (d 8 g .. 8 g + 7), high half = [partner's y | own y] (g + 1)
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1" : "+v"(x[0]), "+v"(x[1]), "+v"(y[0]), "+v"(y[1]));
          const u32x4 out = {x[0], x[1], y[0], y[1]};
          if (qrow[qb] < p.Nq) *(u32x4*)(op + (dt * 32 + 16 * g2 + 8 * hi) * 2) = out;
        }
    } else if (qrow[qb] < p.Nq) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * hi;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = o[qb][dt][g * 4 + e] * inv;
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
}

