// Row-wise ops of the transformer heads (refine_network.py:56-70 nn.TransformerEncoderLayer, post-norm; the
// `.mean(dim=1)` over the 400 tokens, refine_network.py:90-91 / score_network.py:74; PositionalEmbedding,
// network_modules.py:133-137; the N-row Linear layers that follow the token mean), gfx950.  Pure HBM streams.
//
// Arithmetic policy = the reference's autocast(fp16) op sequence (oracle/nets_amp.py): the residual stream of the
// encoder layer and both LayerNorms are fp32 (their inputs are fp32: fp16 tokens + the fp32 positional table), every
// Linear consumes the fp16 rounding of that stream and produces fp16.  So the stream exists twice in HBM: fp32 for the
// next residual add, fp16 as the next GEMM operand.
//   fp_add_pe_f16_fwd        : x16 = fp16(fp32(tok16) + pe)                       -- the in_proj operand
//   fp_layernorm_res_fwd     : z = resid + fp32(branch16); y = LN(z)*gamma + beta -> y32 and y16;  resid = x32 or
//                              fp32(tok16) + pe (the layer input is never materialised in fp32)
//   fp_colmean_f16_fwd       : out[g] = mean over the rows of group g of LN(resid32 + fp32(x16)) or of x16 -- the token
//                              mean fused with the last LayerNorm, so the normalised (N,400,512) tensor is never written
//   fp_rows_linear_fwd       : y = x32 @ w16^T + b for a few hundred rows (the heads after the token mean)
// One wave per 512-wide row, 8 elements per lane, fp32 statistics (two-pass in registers: mean, then centred variance),
// fixed summation order (deterministic).
#include <hip/hip_fp16.h>
#include "fp_common.h"

#include "rowops_ln.h"

__global__ __launch_bounds__(256) void k_add_pe512(const _Float16* __restrict__ tok, const float* __restrict__ pe,
                                                   _Float16* __restrict__ out, int M, int S) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float f[8];
  resid_row(nullptr, tok, pe, S, (size_t)row, lane, f);
  half8 y;
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = (_Float16)f[e];
  *reinterpret_cast<half8*>(out + (size_t)row * 512 + lane * 8) = y;
}

__global__ __launch_bounds__(256) void k_layernorm_res512(const float* __restrict__ x32, const _Float16* __restrict__ tok16,
                                                          const float* __restrict__ pe, int S,
                                                          const _Float16* __restrict__ branch, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float* __restrict__ y32,
                                                          _Float16* __restrict__ y16, int M) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float f[8], gm[8], bt[8];
  resid_row(x32, tok16, pe, S, (size_t)row, lane, f);
  const half8 b = *reinterpret_cast<const half8*>(branch + (size_t)row * 512 + lane * 8);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] += (float)b[e];
  ln_row(eps, f);
  load8f(gamma + lane * 8, gm);
  load8f(beta + lane * 8, bt);
  half8 h;
#pragma unroll
  for (int e = 0; e < 8; ++e) { f[e] = fmaf(f[e], gm[e], bt[e]); h[e] = (_Float16)f[e]; }
  if (y32) store8f(y32 + (size_t)row * 512 + lane * 8, f);
  if (y16) *reinterpret_cast<half8*>(y16 + (size_t)row * 512 + lane * 8) = h;
}

template <bool LN>
__global__ __launch_bounds__(1024) void k_colmean512(const _Float16* __restrict__ X, const float* __restrict__ R32,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                     float* __restrict__ out, int rows_per_group) {
  __shared__ float part[16][512];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = blockIdx.x;
  const size_t base = (size_t)g * rows_per_group * 512;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // A wave's rows (wid, wid + 16, ...) are taken CM_ROWS at a time: their loads go out together and their LayerNorm
  // reductions -- twelve dependent cross-lane exchanges per row -- are interleaved; each row's arithmetic and the order in
  // which rows enter `acc` are unchanged.  With one row in flight per wave the kernel ran at the latency of a load plus that
  // chain (16 waves x 3 KB per CU and round trip: 3.1 TB/s on the 126 CUs a sub-batch's 126 groups occupy).
  constexpr int CM_ROWS = 4;
  for (int r0 = wid; r0 < rows_per_group; r0 += 16 * CM_ROWS) {
    float f[CM_ROWS][8];
    {
      half8 x[CM_ROWS];
      float rr[CM_ROWS][8];
#pragma unroll
      for (int u = 0; u < CM_ROWS; ++u) {
        const int r = min(r0 + 16 * u, rows_per_group - 1);   // past the end: a valid row, normalised but never added
        x[u] = *reinterpret_cast<const half8*>(X + base + (size_t)r * 512 + lane * 8);
        if (LN && R32) load8f(R32 + base + (size_t)r * 512 + lane * 8, rr[u]);
      }
#pragma unroll
      for (int u = 0; u < CM_ROWS; ++u) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          f[u][e] = (float)x[u][e];
          if (LN && R32) f[u][e] += rr[u][e];
        }
      }
    }
    if (LN) ln_rows<CM_ROWS>(eps, f);
#pragma unroll
    for (int u = 0; u < CM_ROWS; ++u) {
      const bool live = r0 + 16 * u < rows_per_group;   // wave-uniform
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = live ? acc[e] + f[u][e] : acc[e];
    }
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

// y[m][n] = sum_k x[m][k] * w[n][k] + b[n]: a workgroup takes 4 rows of x (fp32, in LDS) and a chunk of RL_COLS outputs,
// wave w the outputs chunk + w, w+4, ...; a lane owns 8 consecutive k of every 512-wide k block, so a weight row is
// read as whole 1 KiB lines.
#define RL_ROWS 4
#define RL_COLS 32
template <bool X16, bool Y16>
__global__ __launch_bounds__(256) void k_rows_linear(const void* __restrict__ Xv, const _Float16* __restrict__ Wt,
                                                     const float* __restrict__ bias, void* __restrict__ Yv, int M, int K, int N,
                                                     int round_f16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rl[];
  float* xs = reinterpret_cast<float*>(smem_rl);          // [RL_ROWS][K]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int m0 = blockIdx.x * RL_ROWS;
  for (int i = threadIdx.x; i < RL_ROWS * K; i += 256) {
    const int r = i / K, k = i - r * K;
    float v = 0.f;
    if (m0 + r < M)
      v = X16 ? (float)reinterpret_cast<const _Float16*>(Xv)[(size_t)(m0 + r) * K + k]
              : reinterpret_cast<const float*>(Xv)[(size_t)(m0 + r) * K + k];
    xs[i] = v;
  }
  __syncthreads();
  const int n_end = min(N, (int)(blockIdx.y + 1) * RL_COLS);
  for (int n = blockIdx.y * RL_COLS + wid; n < n_end; n += 4) {
    float acc[RL_ROWS] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 512) {
      const int k = k0 + lane * 8;
      if (k < K) {
        const half8 w = *reinterpret_cast<const half8*>(Wt + (size_t)n * K + k);
#pragma unroll
        for (int r = 0; r < RL_ROWS; ++r) {
          float xv[8];
          load8f(xs + r * K + k, xv);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[r] = fmaf(xv[e], (float)w[e], acc[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RL_ROWS; ++r) acc[r] = wave_sum(acc[r]);
    if (lane < RL_ROWS && m0 + lane < M) {
      float v = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : (lane == 2 ? acc[2] : acc[3]));
      v += bias ? bias[n] : 0.f;
      if (Y16) reinterpret_cast<_Float16*>(Yv)[(size_t)(m0 + lane) * N + n] = (_Float16)v;
      else reinterpret_cast<float*>(Yv)[(size_t)(m0 + lane) * N + n] = round_f16 ? (float)(_Float16)v : v;
    }
  }
}

// dst[c][r][0..C) = src[r][0..C) for c < copies: `rows` rows of C fp16 values at a row stride, 16 bytes per lane, every source
// vector read once and stored `copies / gridDim.y` times by the same lane (torch.cat((a, b), 1) of refine_network.py:84 when all b
// are one image: engine._HipEncoder shared_b)
__global__ __launch_bounds__(256) void k_replicate_rows(const uint4* __restrict__ src, uint4* __restrict__ dst, int copies, int rows,
                                                        int vec_per_row, int src_stride_v, int dst_stride_v, size_t copy_stride_v) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * vec_per_row) return;
  const int r = idx / vec_per_row, v = idx - r * vec_per_row;
  const uint4 val = src[(size_t)r * src_stride_v + v];
  uint4* d = dst + (size_t)r * dst_stride_v + v;
  for (int c = blockIdx.y; c < copies; c += gridDim.y) d[(size_t)c * copy_stride_v] = val;
}

extern "C" int fp_replicate_rows_f16(const void* src, void* dst, int copies, int rows, int channels, int src_row_stride,
                                     int dst_row_stride, long long dst_copy_stride, void* stream) {
  FP_REQUIRE(copies >= 0 && rows >= 0, "fp_replicate_rows_f16: negative size");
  if (copies == 0 || rows == 0) return FP_OK;
  FP_REQUIRE(src && dst, "fp_replicate_rows_f16: NULL tensor");
  FP_REQUIRE(channels > 0 && channels % 8 == 0 && src_row_stride % 8 == 0 && dst_row_stride % 8 == 0 && dst_copy_stride % 8 == 0,
             "fp_replicate_rows_f16: channels and strides must be multiples of 8 fp16 values (16-byte vectors)");
  FP_REQUIRE(src_row_stride >= channels && dst_row_stride >= channels && dst_copy_stride > 0, "fp_replicate_rows_f16: bad strides");
  FP_REQUIRE((((size_t)src | (size_t)dst) & 15) == 0, "fp_replicate_rows_f16: tensors must be 16-byte aligned");
  FP_REQUIRE((long long)rows * (channels / 8) < (1ll << 31), "fp_replicate_rows_f16: rows * channels too large");
  const int vpr = channels / 8;
  const dim3 grid(fp_cdiv(rows * vpr, 256), copies < 16 ? copies : 16), block(256);
  hipLaunchKernelGGL(k_replicate_rows, grid, block, 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, copies, rows, vpr,
                     src_row_stride / 8, dst_row_stride / 8, (size_t)(dst_copy_stride / 8));
  FP_CHECK_LAUNCH("fp_replicate_rows_f16");
  return FP_OK;
}

extern "C" int fp_add_pe_f16_fwd(const void* tok, const float* pe, void* out, int M, int S, int D, void* stream) {
  FP_REQUIRE(M >= 0, "fp_add_pe_f16_fwd: M < 0");
  if (M == 0) return FP_OK;
  FP_REQUIRE(tok && pe && out && S > 0, "fp_add_pe_f16_fwd: bad arguments");
  FP_REQUIRE(D == 512, "fp_add_pe_f16_fwd: D=%d unsupported (d_model of both networks is 512)", D);
  hipLaunchKernelGGL(k_add_pe512, dim3(fp_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)tok, pe,
                     (_Float16*)out, M, S);
  FP_CHECK_LAUNCH("fp_add_pe_f16_fwd");
  return FP_OK;
}

extern "C" int fp_layernorm_res_fwd(const float* x32, const void* tok16, const float* pe, int S, const void* branch16,
                                    const float* gamma, const float* beta, float eps, float* y32, void* y16, int M, int D,
                                    void* stream) {
  FP_REQUIRE(M >= 0, "fp_layernorm_res_fwd: M < 0");
  if (M == 0) return FP_OK;
  FP_REQUIRE(branch16 && gamma && beta && (y32 || y16), "fp_layernorm_res_fwd: NULL tensor");
  FP_REQUIRE((x32 != nullptr) != (tok16 != nullptr), "fp_layernorm_res_fwd: give the residual as x32 OR as tok16 (+ pe)");
  FP_REQUIRE(x32 || (pe && S > 0), "fp_layernorm_res_fwd: tok16 needs the positional table and its period");
  FP_REQUIRE(D == 512, "fp_layernorm_res_fwd: D=%d unsupported (d_model of both networks is 512)", D);
  hipLaunchKernelGGL(k_layernorm_res512, dim3(fp_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x32, (const _Float16*)tok16,
                     pe, S, (const _Float16*)branch16, gamma, beta, eps, y32, (_Float16*)y16, M);
  FP_CHECK_LAUNCH("fp_layernorm_res_fwd");
  return FP_OK;
}

extern "C" int fp_colmean_f16_fwd(const void* x, const float* resid32, const float* gamma, const float* beta, float eps,
                                  float* out, int groups, int rows_per_group, int D, void* stream) {
  FP_REQUIRE(groups >= 0, "fp_colmean_f16_fwd: groups < 0");
  if (groups == 0) return FP_OK;
  FP_REQUIRE(x && out && rows_per_group > 0, "fp_colmean_f16_fwd: bad arguments");
  FP_REQUIRE(D == 512, "fp_colmean_f16_fwd: D=%d unsupported (d_model of both networks is 512)", D);
  FP_REQUIRE((gamma == nullptr) == (beta == nullptr), "fp_colmean_f16_fwd: gamma and beta go together");
  FP_REQUIRE(gamma || !resid32, "fp_colmean_f16_fwd: a residual only makes sense with the LayerNorm");
  if (gamma)
    hipLaunchKernelGGL(k_colmean512<true>, dim3(groups), dim3(1024), 0, (hipStream_t)stream, (const _Float16*)x, resid32, gamma,
                       beta, eps, out, rows_per_group);
  else
    hipLaunchKernelGGL(k_colmean512<false>, dim3(groups), dim3(1024), 0, (hipStream_t)stream, (const _Float16*)x, resid32, gamma,
                       beta, eps, out, rows_per_group);
  FP_CHECK_LAUNCH("fp_colmean_f16_fwd");
  return FP_OK;
}

extern "C" int fp_rows_linear_fwd(const void* x, const void* w, const float* bias, void* y, int M, int K, int N,
                                  int flags, void* stream) {
  FP_REQUIRE(M >= 0 && N >= 0, "fp_rows_linear_fwd: negative size");
  if (M == 0 || N == 0) return FP_OK;
  FP_REQUIRE(x && w && y, "fp_rows_linear_fwd: NULL tensor");
  FP_REQUIRE(K > 0 && K % 8 == 0 && K <= 2048, "fp_rows_linear_fwd: K=%d must be a multiple of 8 (<= 2048)", K);
  FP_REQUIRE((((size_t)x | (size_t)w) & 15) == 0, "fp_rows_linear_fwd: tensors must be 16-byte aligned");
  FP_REQUIRE((flags & ~(FP_ROWS_ROUND_F16 | FP_ROWS_X_F16 | FP_ROWS_Y_F16)) == 0, "fp_rows_linear_fwd: unknown flags 0x%x", flags);
  const dim3 grid(fp_cdiv(M, RL_ROWS), fp_cdiv(N, RL_COLS)), block(256);
  const size_t lds = (size_t)RL_ROWS * K * sizeof(float);
  const int rnd = (flags & FP_ROWS_ROUND_F16) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  const _Float16* wt = (const _Float16*)w;
  if (flags & FP_ROWS_X_F16) {
    if (flags & FP_ROWS_Y_F16) hipLaunchKernelGGL((k_rows_linear<true, true>), grid, block, lds, st, x, wt, bias, y, M, K, N, rnd);
    else hipLaunchKernelGGL((k_rows_linear<true, false>), grid, block, lds, st, x, wt, bias, y, M, K, N, rnd);
  } else {
    if (flags & FP_ROWS_Y_F16) hipLaunchKernelGGL((k_rows_linear<false, true>), grid, block, lds, st, x, wt, bias, y, M, K, N, rnd);
    else hipLaunchKernelGGL((k_rows_linear<false, false>), grid, block, lds, st, x, wt, bias, y, M, K, N, rnd);
  }
  FP_CHECK_LAUNCH("fp_rows_linear_fwd");
  return FP_OK;
}
