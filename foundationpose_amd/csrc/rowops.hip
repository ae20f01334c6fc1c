// Row-wise ops of the transformer heads (refine_network.py:56-70 nn.TransformerEncoderLayer post-norm LayerNorms and
// the `.mean(dim=1)` over the 400 tokens, refine_network.py:90-91 / score_network.py:74), gfx950.  Pure HBM streams:
//   fp_layernorm_f16_fwd : y = LN(x) * gamma + beta, one wave per 512-wide row, 16 bytes per lane in and out, fp32
//                          statistics (two-pass in registers: mean, then centred variance), wave reduction by DPP/shuffle
//   fp_colmean_f16_fwd   : out[g] = mean over the rows of group g of (LN(x) or x) -- the token mean fused with the last
//                          LayerNorm, so the normalised (N,400,512) tensor is never written; one 1024-thread
//                          workgroup per group, fixed summation order (deterministic)
#include <hip/hip_fp16.h>
#include "fp_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// row of 512 halves held as 8 per lane -> normalised values (fp32) in f[8]
__device__ __forceinline__ void ln_row(const half8 x, float eps, float f[8]) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { f[e] = (float)x[e]; s += f[e]; }
  const float mean = wave_sum(s) * (1.0f / 512.0f);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { f[e] -= mean; q = fmaf(f[e], f[e], q); }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / 512.0f) + eps);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] *= rstd;
}

__global__ __launch_bounds__(256) void k_layernorm512(const _Float16* __restrict__ X, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, _Float16* __restrict__ Y,
                                                      int M) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const half8 x = *reinterpret_cast<const half8*>(X + (size_t)row * 512 + lane * 8);
  float f[8];
  ln_row(x, eps, f);
  half8 y;
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = (_Float16)fmaf(f[e], gamma[lane * 8 + e], beta[lane * 8 + e]);
  *reinterpret_cast<half8*>(Y + (size_t)row * 512 + lane * 8) = y;
}

template <bool LN>
__global__ __launch_bounds__(1024) void k_colmean512(const _Float16* __restrict__ X, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, float* __restrict__ out,
                                                     int rows_per_group) {
  __shared__ float part[16][512];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = blockIdx.x;
  const _Float16* Xg = X + (size_t)g * rows_per_group * 512;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = wid; r < rows_per_group; r += 16) {
    const half8 x = *reinterpret_cast<const half8*>(Xg + (size_t)r * 512 + lane * 8);
    float f[8];
    if (LN) {
      ln_row(x, eps, f);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (float)x[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[wid][lane * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 512) {
    const int c = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += part[w][c];
    s *= 1.0f / (float)rows_per_group;
    if (LN) s = fmaf(s, gamma[c], beta[c]);   // mean(LN(x)*gamma + beta) = mean(LN(x))*gamma + beta
    out[(size_t)g * 512 + c] = s;
  }
}

extern "C" int fp_layernorm_f16_fwd(const void* x, const float* gamma, const float* beta, float eps, void* y, int M,
                                    int D, void* stream) {
  FP_REQUIRE(M >= 0, "fp_layernorm_f16_fwd: M < 0");
  if (M == 0) return FP_OK;
  FP_REQUIRE(x && gamma && beta && y, "fp_layernorm_f16_fwd: NULL tensor");
  FP_REQUIRE(D == 512, "fp_layernorm_f16_fwd: D=%d unsupported (d_model of both networks is 512)", D);
  hipLaunchKernelGGL(k_layernorm512, dim3(fp_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, gamma, beta,
                     eps, (_Float16*)y, M);
  FP_CHECK_LAUNCH("fp_layernorm_f16_fwd");
  return FP_OK;
}

extern "C" int fp_colmean_f16_fwd(const void* x, const float* gamma, const float* beta, float eps, float* out, int groups,
                                  int rows_per_group, int D, void* stream) {
  FP_REQUIRE(groups >= 0, "fp_colmean_f16_fwd: groups < 0");
  if (groups == 0) return FP_OK;
  FP_REQUIRE(x && out && rows_per_group > 0, "fp_colmean_f16_fwd: bad arguments");
  FP_REQUIRE(D == 512, "fp_colmean_f16_fwd: D=%d unsupported (d_model of both networks is 512)", D);
  FP_REQUIRE((gamma == nullptr) == (beta == nullptr), "fp_colmean_f16_fwd: gamma and beta go together");
  if (gamma)
    hipLaunchKernelGGL(k_colmean512<true>, dim3(groups), dim3(1024), 0, (hipStream_t)stream, (const _Float16*)x, gamma, beta,
                       eps, out, rows_per_group);
  else
    hipLaunchKernelGGL(k_colmean512<false>, dim3(groups), dim3(1024), 0, (hipStream_t)stream, (const _Float16*)x, gamma, beta,
                       eps, out, rows_per_group);
  FP_CHECK_LAUNCH("fp_colmean_f16_fwd");
  return FP_OK;
}
