// common.h — shared device/host helpers for libomg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/omg_hip.h"

typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define OMG_DEV __device__ __forceinline__

template <typename T> struct Vec;
template <> struct Vec<f16> {
  using v8 = f16x8; using v4 = f16x4;
  static constexpr bool is_f16 = true;
  static OMG_DEV f32x16 mfma32(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static OMG_DEV f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Vec<bf16> {
  using v8 = bf16x8; using v4 = bf16x4;
  static constexpr bool is_f16 = false;
  static OMG_DEV f32x16 mfma32(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static OMG_DEV f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

template <typename T> OMG_DEV float to_f32(T x) { return (float)x; }
template <typename T> OMG_DEV T from_f32(float x) { return (T)x; }

// 16-byte vector <-> 8 floats
template <typename T> OMG_DEV void unpack8(u32x4 raw, float (&f)[8]) {
  typename Vec<T>::v8 v = __builtin_bit_cast(typename Vec<T>::v8, raw);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T> OMG_DEV u32x4 pack8(const float (&f)[8]) {
  typename Vec<T>::v8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (T)f[i];
  return __builtin_bit_cast(u32x4, v);
}

// 8 consecutive elements <-> 8 floats for 16-bit storage (one 16-byte access) and for fp32 storage (two)
template <typename T> OMG_DEV void load8(const char* p, float (&f)[8]) { unpack8<T>(*(const u32x4*)p, f); }
template <> OMG_DEV void load8<float>(const char* p, float (&f)[8]) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
}
template <typename T> OMG_DEV void store8(char* p, const float (&f)[8]) { *(u32x4*)p = pack8<T>(f); }
template <> OMG_DEV void store8<float>(char* p, const float (&f)[8]) {
  *(f32x4*)p = f32x4{f[0], f[1], f[2], f[3]};
  *(f32x4*)(p + 16) = f32x4{f[4], f[5], f[6], f[7]};
}

OMG_DEV float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// x * sigmoid(x) with the hardware reciprocal (1 ulp) — the IEEE division of silu_f is ten VALU instructions (GEMM epilogues)
OMG_DEV float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
#include "gelu.h"         // gelu_f / gelu_f2: the exact (erf) GELU of the GEGLU epilogues through one transcendental

// ---- MX-fp8 quantisation helpers (gemm_mx8.hip, norm.hip): E8M0 scale of a 32-block and the e4m3 element cast.
// Scale = the smallest power of two 2^e with amax / 2^e <= 448 (the e4m3 maximum): no element saturates.  Returns e + 127.
OMG_DEV unsigned mx8_scale_exp(float amax) {
  const float v = amax * (1.0f / 448.0f);
  const unsigned bits = __builtin_bit_cast(unsigned, v);
  int e = (int)(bits >> 23) - 127 + ((bits & 0x7fffffu) != 0 ? 1 : 0);
  e = e < -127 ? -127 : (e > 127 ? 127 : e);
  return (unsigned)(e + 127);
}
// 2^-(be - 127): be = 0 -> 2^127 (a block of zeros), be = 254 -> 2^-127 as the exponent field 0 would be 0.0, never reached by
// fp16 / bf16 inputs (|x| / 448 < 2^127)
OMG_DEV float mx8_inv_scale(unsigned be) { return __builtin_bit_cast(float, (254u - be) << 23); }
OMG_DEV unsigned mx8_pack4(float a, float b, float c, float d) {     // four e4m3 bytes, round to nearest even (v_cvt_pk_fp8_f32: OCP on gfx950)
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (unsigned)r;
}

// 256 zero bytes any lane may source a padded (out-of-image / beyond-K) 16-byte chunk from
extern __device__ __attribute__((aligned(256))) unsigned char omg_zero_page[256];

// host side
void omg_set_error(const char* msg);
static inline int omg_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { omg_set_error(hipGetErrorString(e)); (void)what; return OMG_ELAUNCH; }
  return OMG_OK;
}
// clear any stale sticky error left by an unrelated runtime call, then launch
#define OMG_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define OMG_REQUIRE(cond, msg) do { if (!(cond)) { omg_set_error(msg); return OMG_EINVAL; } } while (0)
