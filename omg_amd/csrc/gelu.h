// gelu.h — the exact (erf) GELU of the GEGLU epilogues (diffusers GEGLU: F.gelu(gate), approximate='none') through ONE transcendental.
// Written in round 4, first run and landed in round 5 (profiles/r05_third_gelu2_*.log: every tile variant bitwise equal, the gate function within
// one fp16 ulp of the float64 erf form, the GEGLU launches of the benchmark +2.3 ... 5.1 %).  It replaced erf by Abramowitz & Stegun 7.1.26
// (v_rcp_f32 + five FMAs + v_exp_f32 + seven more instructions per gate value: the emitted GEGLU epilogue held 128 v_rcp, 128 v_exp and ~1650 other
// VALU instructions per lane — the largest epilogue of the step, VALU work with nothing to hide behind).  Transcendentals are quarter rate.
// This form:   erfc(t) = 2^q(t),  q = t (c1 + c2 t + ... + c7 t^6)  (weighted minimax fit of log2 erfc on [0, 4.25], |error| of erfc <= 7.8e-8 —
// A&S 7.1.26: 1.5e-7 — no constant term: q(0) = 0 exactly), and with h = 2^(q - 1) = erfc / 2
//     gelu(x) = x Phi(x) = max(x, 0) - |x| h          (x >= 0: x - x erfc/2;  x < 0: x erfc(|t|)/2 = -|x| h)
// the sign handling, the 1 + erf and the 0.5 x are gone as well.  The coefficients below are in a = |x| (the 1/sqrt 2 of t folded in), a is
// clamped to 4.25 sqrt 2 where erfc = 3e-9: clamp, eight FMAs, exp2, max, FMA.
// Measured on the host in emulated fp32 over [-12, 12] + N(0, 2) (tests/test_gelu.py): max |gelu - exact| 3.9e-7 (the erf_as form: 4.7e-7); 5.7 % of
// the fp16-rounded outputs differ from the erf_as form's by one ulp, each as close to the correctly rounded value as the other.
// The polynomial is written on PAIRS so that its eight FMAs are four v_pk_fma_f32 (left to the SLP vectoriser the scalar form stayed scalar:
// 872 v_fmaak / v_fma per lane instead of ~512 packed); per element the emitted code is then one v_med3 (|x| as a source modifier), four packed
// FMAs, v_exp_f32, v_max, half a packed FMA.  The packed and the scalar FMA round identically: gelu_f (the small-tile kernels) and gelu_f2 agree bit for bit
// (tests/test_kernels_gpu.py::test_gemm_variants_are_bitwise_identical).
// Non-finite gates (pinned by tests/test_kernels_gpu.py::test_geglu_gate_of_non_finite_values): +inf -> +inf, -inf -> -0 (6.01 * 1.5e-9 below zero),
// +-65504 saturate to x / 0.  NaN gates are NOT guaranteed to propagate: v_med3 returns min3 for a NaN operand (a = 0) and the integer max takes a NaN whose
// sign bit is set (x86's default NaN) to 0, so such a gate comes out 0; the test prints what a NaN gate gives on the part beside a plain GEMM with the same
// NaN bias.  Accepted: fmaxf instead of the integer max canonicalises its operand first — one more VALU instruction per element on the largest epilogue of
// the step — to keep a value that only a corrupt input can carry; overflow in the ff projection shows up as +-inf, which does propagate.
typedef float omg_f32x2 __attribute__((ext_vector_type(2)));
OMG_DEV omg_f32x2 gelu_f2(omg_f32x2 x) {
  const omg_f32x2 a = {__builtin_amdgcn_fmed3f(__builtin_fabsf(x[0]), 0.0f, 6.0104076400856545f),
                       __builtin_amdgcn_fmed3f(__builtin_fabsf(x[1]), 0.0f, 6.0104076400856545f)};      // min(|x|, 4.25 sqrt 2)
#define OMG_S2(c_) (omg_f32x2{(c_), (c_)})
  omg_f32x2 p = OMG_S2(8.857517595988839e-06f);
  p = __builtin_elementwise_fma(p, a, OMG_S2(-5.7689967245371745e-05f));
  p = __builtin_elementwise_fma(p, a, OMG_S2(-0.00040700129195864497f));
  p = __builtin_elementwise_fma(p, a, OMG_S2(0.007363154687953159f));
  p = __builtin_elementwise_fma(p, a, OMG_S2(-0.052666626891519484f));
  p = __builtin_elementwise_fma(p, a, OMG_S2(-0.4591643175619639f));
  p = __builtin_elementwise_fma(p, a, OMG_S2(-1.1511088393548352f));
  const omg_f32x2 q1 = __builtin_elementwise_fma(p, a, OMG_S2(-1.0f));
#undef OMG_S2
  const omg_f32x2 h = {__builtin_amdgcn_exp2f(q1[0]), __builtin_amdgcn_exp2f(q1[1])};                    // erfc(|x| / sqrt 2) / 2
  // max(x, 0) on the BITS: a negative float is a negative integer (v_max_i32: one instruction; fmaxf canonicalises its operand first: two)
  // through scalar copies: __builtin_bit_cast applied to the vector ELEMENT x[1] read element 0 (hipcc 7.2: both maxima came out as max(x[0], 0) and
  // the final v_pk_fma_f32 used r[0] for both halves — found in round 5 as 37 % wrong outputs of every large-tile GEGLU launch, a per-variant diff tool, profiles/r05_second_gelu2_diag.log)
  const float x0 = x[0], x1 = x[1];
  const int b0 = __builtin_bit_cast(int, x0), b1 = __builtin_bit_cast(int, x1);
  const int m0 = b0 > 0 ? b0 : 0, m1 = b1 > 0 ? b1 : 0;
  const omg_f32x2 r = {__builtin_bit_cast(float, m0), __builtin_bit_cast(float, m1)};
  return __builtin_elementwise_fma(-a, h, r);
}
OMG_DEV float gelu_f(float x) { return gelu_f2(omg_f32x2{x, x})[0]; }
