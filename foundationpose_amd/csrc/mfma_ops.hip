// MFMA kernels of the network stage (gfx950 / CDNA4, wave64):
//   fp_conv7x7s2_bn_relu_fwd : the "patch-embed" conv (refine_network.py:38, score_network.py:37) as an
//                              implicit GEMM  [64 ch] x [K = C_in*49 -> 304] x [pixels], fused BN(scale,shift)+ReLU
//   fp_linear_f16_fwd        : y = x @ w^T + b (QKV in_proj 512 -> 1536 and the other 512-wide projections),
//                              128x128x32 tiles, v_mfma_f32_16x16x32_f16, fp32 accumulate
#include <hip/hip_fp16.h>
#include "fp_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4_ __attribute__((ext_vector_type(4)));
typedef float float16_ __attribute__((ext_vector_type(16)));

// =====================================================================================================
// fp_linear_f16_fwd
// =====================================================================================================
#define LIN_BM 128
#define LIN_BN 128
#define LIN_BK 32
#define LIN_LDS_STRIDE 40  // halves per LDS row: 32 + 8 pad (80 B) => conflict-free ds_read_b128 per 16-lane group

__global__ __launch_bounds__(256) void k_linear_f16(const _Float16* __restrict__ X, const _Float16* __restrict__ Wt,
                                                    const float* __restrict__ bias, _Float16* __restrict__ Y, int M,
                                                    int K, int Nout, int relu) {
  __shared__ __attribute__((aligned(16))) _Float16 sA[LIN_BM * LIN_LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) _Float16 sB[LIN_BN * LIN_LDS_STRIDE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;  // 2x2 waves, each 64x64
  // XCD-aware tile order: consecutive tiles along M on the same XCD share the same weight panel in L2
  const int tiles_m = (M + LIN_BM - 1) / LIN_BM;
  const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
  const int row_base = bm * LIN_BM, col_base = bn * LIN_BN;

  float4_ acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4_){0.f, 0.f, 0.f, 0.f};

  // staging assignment: thread -> (row, 8-half column chunk); two passes of 64 rows
  const int lr = tid >> 2, lc = (tid & 3) * 8;
  const int fr = lane & 15, fk = (lane >> 4) * 8;

  for (int k0 = 0; k0 < K; k0 += LIN_BK) {
    half8 ra[2], rb[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int r = lr + ps * 64;
      int gr = row_base + r;
      gr = gr < M ? gr : M - 1;
      ra[ps] = *reinterpret_cast<const half8*>(X + (size_t)gr * K + k0 + lc);
      rb[ps] = *reinterpret_cast<const half8*>(Wt + (size_t)(col_base + r) * K + k0 + lc);
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int r = lr + ps * 64;
      *reinterpret_cast<half8*>(&sA[r * LIN_LDS_STRIDE + lc]) = ra[ps];
      *reinterpret_cast<half8*>(&sB[r * LIN_LDS_STRIDE + lc]) = rb[ps];
    }
    __syncthreads();
    half8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      fa[i] = *reinterpret_cast<const half8*>(&sA[(wr * 64 + i * 16 + fr) * LIN_LDS_STRIDE + fk]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fb[j] = *reinterpret_cast<const half8*>(&sB[(wc * 64 + j * 16 + fr) * LIN_LDS_STRIDE + fk]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  }
  // epilogue: D[row = (lane>>4)*4 + reg][col = lane&15]
  const int ocol = lane & 15, orow = (lane >> 4) * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = col_base + wc * 64 + j * 16 + ocol;
    const float bv = bias ? bias[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gr = row_base + wr * 64 + i * 16 + orow + r;
        if (gr < M) {
          float v = acc[i][j][r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          Y[(size_t)gr * Nout + c] = (_Float16)v;
        }
      }
    }
  }
}

extern "C" int fp_linear_f16_fwd(const void* x, const void* w, const float* bias, void* y, int M, int K, int Nout,
                                 int relu, void* stream) {
  FP_REQUIRE(M >= 0, "fp_linear_f16_fwd: M < 0");
  if (M == 0) return FP_OK;
  FP_REQUIRE(x && w && y, "fp_linear_f16_fwd: NULL tensor");
  FP_REQUIRE(K > 0 && K % LIN_BK == 0, "fp_linear_f16_fwd: K=%d must be a multiple of %d", K, LIN_BK);
  FP_REQUIRE(Nout > 0 && Nout % LIN_BN == 0, "fp_linear_f16_fwd: Nout=%d must be a multiple of %d", Nout, LIN_BN);
  const int tiles_m = fp_cdiv(M, LIN_BM), tiles_n = Nout / LIN_BN;
  hipLaunchKernelGGL(k_linear_f16, dim3(tiles_m * tiles_n), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x,
                     (const _Float16*)w, bias, (_Float16*)y, M, K, Nout, relu);
  FP_CHECK_LAUNCH("fp_linear_f16_fwd");
  return FP_OK;
}

// =====================================================================================================
// fp_conv7x7s2_bn_relu_fwd
// =====================================================================================================
// GEMM view: D[ch (64)][pixel] = sum_k Wp[ch][k] * patch[k][pixel], k = c*49 + ky*7 + kx, padded 294 -> 304.
// MFMA 32x32x16 f16: A = weights (rows = channels), B = input patches (cols = 32 consecutive output pixels of one
// output row), so for a fixed accumulator register the 32 lanes of a half-wave hold 32 consecutive pixels of one
// channel => 64-byte contiguous NCHW stores.
// Workgroup: 256 threads = 4 waves; tile = 4 output rows x 32 output cols; wave w owns output row w.
// LDS: input patch [6][13][72] halves (rows 2*4+5, cols 2*32+5 -> 72) + weights [19][64][16] halves.
#define CV_CIN 6
#define CV_KREAL (CV_CIN * 49)
#define CV_KSTEPS 19
#define CV_KPAD (CV_KSTEPS * 16)
#define CV_TR 4
#define CV_TC 32
#define CV_PH (2 * CV_TR + 5)
#define CV_PW 72
#define CV_PATCH (CV_CIN * CV_PH * CV_PW)

__global__ __launch_bounds__(256) void k_conv7x7s2(const _Float16* __restrict__ X, const _Float16* __restrict__ Wg,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   _Float16* __restrict__ Y, int Hin, int Win, int Hout, int Wout,
                                                   int tiles_x, int tiles_y, int nhwc) {
  __shared__ __attribute__((aligned(16))) _Float16 sW[CV_KSTEPS * 64 * 16];
  __shared__ __attribute__((aligned(16))) _Float16 sP[CV_PATCH];
  __shared__ unsigned short sOff[CV_KPAD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int oy0 = ty * CV_TR, ox0 = tx * CV_TC;

  // weights -> LDS in fragment order [kstep][ch][16], zero padded in k
  for (int e = tid; e < CV_KSTEPS * 64 * 16; e += 256) {
    const int kk = e & 15, ch = (e >> 4) & 63, ks = e >> 10;
    const int k = ks * 16 + kk;
    sW[e] = (k < CV_KREAL) ? Wg[ch * CV_KREAL + k] : (_Float16)0.f;
  }
  // k -> patch offset table
  for (int k = tid; k < CV_KPAD; k += 256) {
    int off = 0;
    if (k < CV_KREAL) {
      const int c = k / 49, r = k - c * 49, ky = r / 7, kx = r - ky * 7;
      off = (c * CV_PH + ky) * CV_PW + kx;
    }
    sOff[k] = (unsigned short)off;
  }
  // input patch (zero padded borders): rows iy = 2*oy0 - 3 + pr, cols ix = 2*ox0 - 3 + pc
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
  const _Float16* Xb = X + (size_t)b * CV_CIN * Hin * Win;
  for (int e = tid; e < CV_PATCH; e += 256) {
    const int pc = e % CV_PW, t = e / CV_PW, pr = t % CV_PH, c = t / CV_PH;
    const int iy = iy0 + pr, ix = ix0 + pc;
    _Float16 v = (_Float16)0.f;
    if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) v = Xb[((size_t)c * Hin + iy) * Win + ix];
    sP[e] = v;
  }
  __syncthreads();

  // wave `wid` computes output row oy0 + wid, 32 pixels, 64 channels (two 32-channel fragments)
  const int pix = lane & 31, kh = (lane >> 5) * 8;
  const int pbase = (2 * wid) * CV_PW + 2 * pix;
  float16_ acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  for (int ks = 0; ks < CV_KSTEPS; ++ks) {
    half8 fb;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ks * 16 + kh + e;
      const _Float16 v = sP[pbase + sOff[k]];
      fb[e] = (k < CV_KREAL) ? v : (_Float16)0.f;
    }
    const half8 fa0 = *reinterpret_cast<const half8*>(&sW[(ks * 64 + pix) * 16 + kh]);
    const half8 fa1 = *reinterpret_cast<const half8*>(&sW[(ks * 64 + 32 + pix) * 16 + kh]);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb, acc1, 0, 0, 0);
  }
  // D[row = channel][col = pixel]: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
  const int oy = oy0 + wid, ox = ox0 + pix;
  if (oy < Hout && ox < Wout) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = (half ? acc1[r] : acc0[r]);
        v = fmaxf(fmaf(v, scale[ch], shift[ch]), 0.f);
        size_t o;
        if (nhwc == 2) o = (((size_t)b * (Hout + 2) + oy + 1) * (Wout + 2) + ox + 1) * 64 + ch;  // NHWC with a 1-px zero border
        else if (nhwc) o = (((size_t)b * Hout + oy) * Wout + ox) * 64 + ch;
        else o = (((size_t)b * 64 + ch) * Hout + oy) * Wout + ox;
        Y[o] = (_Float16)v;
      }
    }
  }
}

int fp_conv1_nhwc_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int B, int Hin,
                         int Win, int pad, hipStream_t stream);   // conv1.hip

extern "C" int fp_conv7x7s2_bn_relu_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y,
                                        int B, int Hin, int Win, int channels_last_out, void* stream) {
  FP_REQUIRE(B >= 0, "fp_conv7x7s2_bn_relu_fwd: B < 0");
  if (B == 0) return FP_OK;
  FP_REQUIRE(x && w && scale && shift && y, "fp_conv7x7s2_bn_relu_fwd: NULL tensor");
  FP_REQUIRE(Hin > 0 && Win > 0 && Hin % 2 == 0 && Win % 2 == 0, "fp_conv7x7s2_bn_relu_fwd: odd input size");
  FP_REQUIRE(B <= 65535, "fp_conv7x7s2_bn_relu_fwd: B=%d exceeds the grid limit; chunk the batch", B);
  FP_REQUIRE(channels_last_out >= 0 && channels_last_out <= 2, "fp_conv7x7s2_bn_relu_fwd: unknown output layout %d", channels_last_out);
  if (channels_last_out != 0 && Win <= 256 && (Win % 8) == 0)   // NHWC outputs: the streaming kernel of conv1.hip
    return fp_conv1_nhwc_launch(x, w, scale, shift, y, B, Hin, Win, channels_last_out == 2 ? 1 : 0, (hipStream_t)stream);
  const int Hout = Hin / 2, Wout = Win / 2;
  const int tiles_x = fp_cdiv(Wout, CV_TC), tiles_y = fp_cdiv(Hout, CV_TR);
  hipLaunchKernelGGL(k_conv7x7s2, dim3(tiles_x * tiles_y, B), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x,
                     (const _Float16*)w, scale, shift, (_Float16*)y, Hin, Win, Hout, Wout, tiles_x, tiles_y,
                     channels_last_out);
  FP_CHECK_LAUNCH("fp_conv7x7s2_bn_relu_fwd");
  return FP_OK;
}
