// Element-wise helpers shared by the split-precision GEMM kernels (csrc/gemm_x3.hip, csrc/mlp_fused.hip): the exact-erf
// GELU (models/layers/octformer_layers.py:53-59 uses torch.nn.GELU()) and the fp32 -> bf16 (hi, lo) split.
#pragma once
#include "hfl_common.h"

__device__ __forceinline__ uint32_t x3_bf16_rne(float v) {
  uint32_t u = __float_as_uint(v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

// exact (erf) GELU, erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, below the 2^-17 relative error of the
// split that follows): one v_rcp, one v_exp, 7 fma -- a third of the library erff's instruction count
__device__ __forceinline__ float x3_gelu(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float erf_abs = fmaf(-poly * t, e, 1.0f);               // erf(|v| / sqrt 2)
  const float erf_v = copysignf(erf_abs, v);
  return 0.5f * v * (1.0f + erf_v);
}

// ---- packed (two values per instruction) forms for the epilogues: the fc1 epilogue was 2500 VALU instructions per wave
// (PMC: 43 M per launch, 70 us of every SIMD against 43 us of MFMA) in its scalar form
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 x3_erf_abs2(f32x2 v, f32x2& e_out) {          // erf(|v| / sqrt 2), e_out = exp(-v^2 / 2)
  const f32x2 z = __builtin_elementwise_abs(v) * 0.70710678118654752440f;
  const f32x2 den = z * 0.3275911f + 1.0f;
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2 poly = t * 1.061405429f + (-1.453152027f);
  poly = poly * t + 1.421413741f;
  poly = poly * t + (-0.284496736f);
  poly = poly * t + 0.254829592f;
  const f32x2 a = z * z * (-1.4426950408889634f);
  const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  e_out = e;
  return 1.0f - poly * t * e;
}
__device__ __forceinline__ f32x2 x3_gelu2(f32x2 v) {
  f32x2 e;
  const f32x2 erf_abs = x3_erf_abs2(v, e);
  const f32x2 erf_v = {copysignf(erf_abs[0], v[0]), copysignf(erf_abs[1], v[1])};
  return v * 0.5f * (erf_v + 1.0f);
}
__device__ __forceinline__ f32x2 x3_gelu_grad2(f32x2 v) {
  f32x2 e;
  const f32x2 erf_abs = x3_erf_abs2(v, e);
  const f32x2 erf_v = {copysignf(erf_abs[0], v[0]), copysignf(erf_abs[1], v[1])};
  return v * 0.3989422804014327f * e + (erf_v + 1.0f) * 0.5f;
}
// two floats -> packed bf16 hi pair and lo pair (v_cvt_pk_bf16_f32: round to nearest even, as x3_bf16_rne)
__device__ __forceinline__ void x3_split_pair(f32x2 v, uint32_t& hi, uint32_t& lo) {
  const x3_bf16x2 h = __builtin_convertvector(v, x3_bf16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const x3_bf16x2 l = __builtin_convertvector(r, x3_bf16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// the same split with scalar subtractions (for code that runs between MFMAs: no v_pk_*_f32 there)
__device__ __forceinline__ void x3_split_pair_scalar(float v0, float v1, uint32_t& hi, uint32_t& lo) {
  const x3_bf16x2 h = __builtin_convertvector((f32x2){v0, v1}, x3_bf16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  const float r0 = v0 - __uint_as_float(hi << 16), r1 = v1 - __uint_as_float(hi & 0xffff0000u);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r0, r1}, x3_bf16x2));
}

// d/dv gelu(v) = Phi(v) + v phi(v), same erf approximation
__device__ __forceinline__ float x3_gelu_grad(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);         // exp(-v^2 / 2)
  const float erf_v = copysignf(fmaf(-poly * t, e, 1.0f), v);
  return fmaf(v * 0.3989422804014327f, e, 0.5f * (1.0f + erf_v));
}

