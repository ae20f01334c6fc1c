// Row helpers of the transformer heads' LayerNorm kernels (rowops.hip, linear_ln.hip): one 512-wide row spread over a wave, 8
// consecutive elements per lane, fp32 statistics (two-pass in registers: mean, then centred variance), fixed summation order.
// Shared as source so that both translation units run the same instruction sequence per row.
#pragma once
#include <hip/hip_fp16.h>
#include "fp_common.h"

#ifndef FP_HALF8_DEFINED
#define FP_HALF8_DEFINED
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#endif
typedef float float4_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void load8f(const float* p, float f[8]) {
  const float4_ a = *reinterpret_cast<const float4_*>(p), b = *reinterpret_cast<const float4_*>(p + 4);
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}

__device__ __forceinline__ void store8f(float* p, const float f[8]) {
  *reinterpret_cast<float4_*>(p) = float4_{f[0], f[1], f[2], f[3]};
  *reinterpret_cast<float4_*>(p + 4) = float4_{f[4], f[5], f[6], f[7]};
}

// f[8] (one row of 512 spread over the wave) -> (f - mean) * rstd, in place
__device__ __forceinline__ void ln_row(float eps, float f[8]) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += f[e];
  const float mean = wave_sum(s) * (1.0f / 512.0f);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { f[e] -= mean; q = fmaf(f[e], f[e], q); }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / 512.0f) + eps);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] *= rstd;
}

// the same for R independent rows held by one wave, the R reduction chains interleaved (per row: exactly ln_row)
template <int R>
__device__ __forceinline__ void ln_rows(float eps, float f[R][8]) {
  float s[R], q[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    s[u] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s[u] += f[u][e];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int u = 0; u < R; ++u) s[u] += __shfl_xor(s[u], o, 64);
  }
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const float mean = s[u] * (1.0f / 512.0f);
    q[u] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { f[u][e] -= mean; q[u] = fmaf(f[u][e], f[u][e], q[u]); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int u = 0; u < R; ++u) q[u] += __shfl_xor(q[u], o, 64);
  }
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const float rstd = rsqrtf(q[u] * (1.0f / 512.0f) + eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[u][e] *= rstd;
  }
}

// residual-stream value of row `row`, elements [8 lane, 8 lane + 8): x32 if given, else fp32(tok16) + pe[row % S]
__device__ __forceinline__ void resid_row(const float* x32, const _Float16* tok16, const float* pe, int S, size_t row, int lane,
                                          float f[8]) {
  if (x32) {
    load8f(x32 + row * 512 + lane * 8, f);
  } else {
    const half8 t = *reinterpret_cast<const half8*>(tok16 + row * 512 + lane * 8);
    float pv[8];
    load8f(pe + (size_t)(row % (size_t)S) * 512 + lane * 8, pv);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)t[e] + pv[e];
  }
}
