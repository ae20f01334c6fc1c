// Observed-frame crop kernel (B side): replaces the kornia warp_perspective call sites
// (predict_pose_refine.py:63,72 ; predict_score.py:89-90), the dataset normalisation
// (h5_dataset.py:79-114 refine, :137-170 score incl. the depth -> frame -> xyz -> crop chain) and the
// channel concat (predict_pose_refine.py:188).  The warp is always scale+translate, so the source
// coordinates are affine in (i, j); one lane per output pixel, consecutive lanes = consecutive i, which makes
// both the frame reads (a few adjacent texels per wave, L2 resident: the frame is 3.7 MB) and the planar
// NCHW stores coalesced.  Compiled with -ffp-contract=off (definition shared with oracle/fp_oracle.c).
#include <hip/hip_fp16.h>
#include "fp_common.h"

struct __attribute__((aligned(4))) f3 { float x, y, z; };   // 12-byte texel, dword aligned: one global_load_dwordx3

__device__ __forceinline__ int nn_index(float x) { return (int)rintf(x); }  // half-to-even like grid_sample nearest

// One output pixel: returns the 6 network channels (rgb/255 bilinear, xyz nearest + normalisation).
template <int MODE>
__device__ __forceinline__ void warp_pixel(const float* __restrict__ rgb, const float* __restrict__ xyz_map,
                                           const float* __restrict__ depthf, float sx, float tx, float sy, float ty,
                                           float i00, float i02, float i11, float i12, float cW, float cH, const fp_k9& K,
                                           float t0, float t1, float t2, float inv_r, bool normalize, int H, int W, int oh,
                                           int ow, int i, int j, float a[6]) {
  const float xs = fmaf((float)i, i00, i02), ys = fmaf((float)j, i11, i12);
  const float ix = fmaf(xs, cW, -0.5f), iy = fmaf(ys, cH, -0.5f);
  // ---- rgb, bilinear with zero padding (tap order nw, ne, sw, se as torch grid_sample)
  {
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = ((float)x1 - ix) * ((float)y1 - iy);
    const float wne = (ix - (float)x0) * ((float)y1 - iy);
    const float wsw = ((float)x1 - ix) * (iy - (float)y0);
    const float wse = (ix - (float)x0) * (iy - (float)y0);
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
    const bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    // one 12-byte load per tap (the frame is AoS rgb): the kernel is bound by the number of gather instructions the
    // texture addresser has to walk, not by bytes -- 5 wide loads per pixel instead of 15 dword loads
    f3 tnw = {0.f, 0.f, 0.f}, tne = tnw, tsw = tnw, tse = tnw;
    if (vx0 && vy0) tnw = *reinterpret_cast<const f3*>(rgb + ((size_t)y0 * W + x0) * 3);
    if (vx1 && vy0) tne = *reinterpret_cast<const f3*>(rgb + ((size_t)y0 * W + x1) * 3);
    if (vx0 && vy1) tsw = *reinterpret_cast<const f3*>(rgb + ((size_t)y1 * W + x0) * 3);
    if (vx1 && vy1) tse = *reinterpret_cast<const f3*>(rgb + ((size_t)y1 * W + x1) * 3);
    const float nw3[3] = {tnw.x, tnw.y, tnw.z}, ne3[3] = {tne.x, tne.y, tne.z}, sw3[3] = {tsw.x, tsw.y, tsw.z}, se3[3] = {tse.x, tse.y, tse.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;                             // same order and the same skipped taps as before: an absent tap adds nothing
      if (vx0 && vy0) acc += nw3[c] * wnw;
      if (vx1 && vy0) acc += ne3[c] * wne;
      if (vx0 && vy1) acc += sw3[c] * wsw;
      if (vx1 && vy1) acc += se3[c] * wse;
      a[c] = acc * (1.0f / 255.0f);  // torch GPU `/255.0` = mul by f32 reciprocal
    }
  }
  // ---- xyz, nearest
  float pt[3] = {0.f, 0.f, 0.f};
  const int qx = nn_index(ix), qy = nn_index(iy);
  const bool q_in = qx >= 0 && qx < W && qy >= 0 && qy < H;
  if (MODE == FP_MODE_REFINE) {
    if (q_in) {
      const f3 s = *reinterpret_cast<const f3*>(xyz_map + ((size_t)qy * W + qx) * 3);
      pt[0] = s.x; pt[1] = s.y; pt[2] = s.z;
    }
  } else if (q_in) {
    const float cSw = (float)ow / (float)(ow - 1), cSh = (float)oh / (float)(oh - 1);
    // integer window edge => s*(q - left): exactly 0 on the edge, so the -0.5 tie of hop 2 is deterministic
    const float lfx = rintf(i02), lfy = rintf(i12);
    const float ccx = (fabsf(i02 - lfx) <= 1e-3f) ? sx * ((float)qx - lfx) : fmaf(sx, (float)qx, tx);
    const float ccy = (fabsf(i12 - lfy) <= 1e-3f) ? sy * ((float)qy - lfy) : fmaf(sy, (float)qy, ty);
    const int px = nn_index(fmaf(ccx, cSw, -0.5f)), py = nn_index(fmaf(ccy, cSh, -0.5f));
    float z = 0.f;
    if (px >= 0 && px < ow && py >= 0 && py < oh) {
      const float xs2 = fmaf((float)px, i00, i02), ys2 = fmaf((float)py, i11, i12);
      const int rx = nn_index(fmaf(xs2, cW, -0.5f)), ry = nn_index(fmaf(ys2, cH, -0.5f));
      if (rx >= 0 && rx < W && ry >= 0 && ry < H) z = depthf[(size_t)ry * W + rx];
    }
    if (!(z < 0.001f)) {
      pt[0] = (((float)qx - K.v[2]) * z) / K.v[0];
      pt[1] = (((float)qy - K.v[5]) * z) / K.v[4];
      pt[2] = z;
    }
  }
  const float thr = (MODE == FP_MODE_SCORE) ? 0.1f : 0.001f;
  const bool invalid = pt[2] < thr;
  const float d[3] = {pt[0] - t0, pt[1] - t1, pt[2] - t2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float val = d[c];
    if (normalize) {
      val = val * inv_r;
      if (invalid || fabsf(val) >= 2.0f) val = 0.f;
    }
    a[3 + c] = val;
  }
}

// One lane per output pixel, consecutive lanes = consecutive pixels of a row (coalesced plane stores).  Measured on
// MI355X: giving a lane 8 pixels (16-byte stores) is 4x SLOWER (0.34 ms vs 0.08 ms at N=252) -- the kernel is bound by
// the gather of the 12-byte AoS frame texels through the vector L1 (about 23 cache lines per wave-load), not by its
// stores, and fewer, fatter lanes only remove the parallelism that hides that latency.
// Per-hypothesis constants of the warp (the inverse of the scale + translate crop transform): four IEEE divisions that
// every lane of a workgroup used to repeat (~45 of its 319 VALU instructions per wave, profiles/r03_stage_counters_sq.json:
// the kernel is VALU-issue-bound, 57 % of its wave cycles are issue stalls); now one lane computes them -- the same
// instructions, so the same bits -- and the workgroup reads them from LDS.  The frame constants W/(W-1), H/(H-1) and the
// row / column of a pixel (an emulated integer division) come from the host: float division there is the same IEEE
// operation, the integer division a multiply-shift.
struct WarpConst { float cW, cH; unsigned mul_ow, shr_ow; };

template <int MODE>
__global__ __launch_bounds__(256) void k_warp(const float* __restrict__ rgb, const float* __restrict__ xyz_map,
                                              const float* __restrict__ depthf, const float* __restrict__ tfs,
                                              fp_k9 K, const float* __restrict__ poses, float inv_r, int flags,
                                              int H, int W, int oh, int ow, void* __restrict__ Bout, WarpConst wc) {
  __shared__ float inv_tf[4];
  const int n = blockIdx.y;
  const float* tf = tfs + (size_t)n * 9;
  const float sx = tf[0], tx = tf[2], sy = tf[4], ty = tf[5];
  if (threadIdx.x == 0) {
    inv_tf[0] = 1.0f / sx;
    inv_tf[1] = 1.0f / sy;
    inv_tf[2] = (-tx) / sx;
    inv_tf[3] = (-ty) / sy;
  }
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int npx = oh * ow;
  if (p >= npx) return;
  const int j = wc.mul_ow ? (int)(__umulhi((unsigned)p, wc.mul_ow) >> wc.shr_ow) : p;
  const int i = p - j * ow;
  const float i00 = inv_tf[0], i11 = inv_tf[1], i02 = inv_tf[2], i12 = inv_tf[3];
  const float cW = wc.cW, cH = wc.cH;
  const float* P = poses + (size_t)n * 16;
  float a[6];
  warp_pixel<MODE>(rgb, xyz_map, depthf, sx, tx, sy, ty, i00, i02, i11, i12, cW, cH, K, P[3], P[7], P[11], inv_r,
                   (flags & FP_FLAG_NORMALIZE_XYZ) != 0, H, W, oh, ow, i, j, a);
  const size_t o = (size_t)n * 6 * npx + p;
  if (flags & FP_FLAG_OUT_F16) {
    __half* B = reinterpret_cast<__half*>(Bout);
#pragma unroll
    for (int c = 0; c < 6; ++c) B[o + (size_t)c * npx] = __float2half_rn(a[c]);
  } else {
    float* B = reinterpret_cast<float*>(Bout);
#pragma unroll
    for (int c = 0; c < 6; ++c) B[o + (size_t)c * npx] = a[c];
  }
}

extern "C" int fp_warp_crops(const float* rgb, const float* xyz_map, const float* depth, const float* tf_to_crops,
                             const float* K9, const float* poses, float mesh_diameter, int flags, int mode, int H,
                             int W, int N, int oh, int ow, void* B, void* stream) {
  FP_REQUIRE(N >= 0, "fp_warp_crops: N < 0");
  if (N == 0) return FP_OK;
  FP_REQUIRE(rgb && tf_to_crops && K9 && poses && B, "fp_warp_crops: NULL tensor");
  FP_REQUIRE(H > 1 && W > 1 && oh > 1 && ow > 1, "fp_warp_crops: degenerate sizes");
  FP_REQUIRE(N <= 65535, "fp_warp_crops: N=%d exceeds the grid limit; chunk the batch", N);
  FP_REQUIRE(mode == FP_MODE_REFINE || mode == FP_MODE_SCORE, "fp_warp_crops: unknown mode %d", mode);
  FP_REQUIRE(mode != FP_MODE_REFINE || xyz_map, "fp_warp_crops: REFINE mode needs xyz_map");
  FP_REQUIRE(mode != FP_MODE_SCORE || depth, "fp_warp_crops: SCORE mode needs depth");
  fp_k9 K;
  for (int i = 0; i < 9; ++i) K.v[i] = K9[i];
  const float inv_r = 1.0f / (mesh_diameter * 0.5f);
  dim3 grid(fp_cdiv(oh * ow, 256), N), block(256);
  WarpConst wc;
  wc.cW = (float)W / (float)(W - 1);
  wc.cH = (float)H / (float)(H - 1);
  {   // p / ow for p < oh * ow <= 2^20 as umulhi(p, mul) >> shr (exact: ceil(2^(31+lg) / ow) with lg = ceil(log2 ow)); 0 = divisor 1
    wc.mul_ow = 0; wc.shr_ow = 0;
    if (ow > 1) {
      int lg = 0;
      while ((1u << lg) < (unsigned)ow) ++lg;
      const int sh = 31 + lg;
      wc.mul_ow = (unsigned)(((1ull << sh) + (unsigned)ow - 1) / (unsigned)ow);
      wc.shr_ow = (unsigned)(sh - 32);
    }
  }
  if (mode == FP_MODE_REFINE)
    hipLaunchKernelGGL(k_warp<FP_MODE_REFINE>, grid, block, 0, (hipStream_t)stream, rgb, xyz_map, depth, tf_to_crops,
                       K, poses, inv_r, flags, H, W, oh, ow, B, wc);
  else
    hipLaunchKernelGGL(k_warp<FP_MODE_SCORE>, grid, block, 0, (hipStream_t)stream, rgb, xyz_map, depth, tf_to_crops,
                       K, poses, inv_r, flags, H, W, oh, ow, B, wc);
  FP_CHECK_LAUNCH("fp_warp_crops");
  return FP_OK;
}
