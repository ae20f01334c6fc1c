// Per-frame and per-hypothesis small ops: depth erosion / bilateral filter / back-projection,
// crop windows, pose update.  All HBM-trivial; one thread per pixel or per pose.
// Compiled with -ffp-contract=off so the operation order matches the definition in DESIGN.md.
#include "fp_common.h"

// ---------------------------------------------------------------- a1 (Utils.py:359-384)
__global__ void k_erode(const float* __restrict__ depth, float* __restrict__ out, int H, int W, int radius,
                        float diff_thres, float ratio_thres, float zfar) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y * blockDim.y + threadIdx.y;
  if (w >= W || h >= H) return;
  const float d0 = depth[h * W + w];
  float bad = 0.f, total = 0.f;
  for (int u = w - radius; u <= w + radius; ++u) {
    if (u < 0 || u >= W) continue;
    for (int v = h - radius; v <= h + radius; ++v) {
      if (v < 0 || v >= H) continue;
      const float cur = depth[v * W + u];
      total += 1.0f;
      if (cur < 0.001f || cur >= zfar || fabsf(cur - d0) > diff_thres) bad += 1.0f;
    }
  }
  out[h * W + w] = (bad / total > ratio_thres) ? 0.0f : d0;
}

// ---------------------------------------------------------------- a2 (Utils.py:304-343)
__global__ void k_bilateral(const float* __restrict__ depth, float* __restrict__ out, int H, int W, int radius,
                            float zfar, float sigmaD, float sigmaR) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y * blockDim.y + threadIdx.y;
  if (w >= W || h >= H) return;
  float res = 0.f, mean = 0.f;
  int nvalid = 0;
  for (int u = w - radius; u <= w + radius; ++u) {
    if (u < 0 || u >= W) continue;
    for (int v = h - radius; v <= h + radius; ++v) {
      if (v < 0 || v >= H) continue;
      const float cur = depth[v * W + u];
      if (cur >= 0.001f && cur < zfar) { nvalid++; mean += cur; }
    }
  }
  if (nvalid > 0) {
    mean /= (float)nvalid;
    const float dC = depth[h * W + w];
    const float two_sd2 = 2.0f * sigmaD * sigmaD, two_sr2 = 2.0f * sigmaR * sigmaR;
    float sw = 0.f, s = 0.f;
    for (int u = w - radius; u <= w + radius; ++u) {
      if (u < 0 || u >= W) continue;
      for (int v = h - radius; v <= h + radius; ++v) {
        if (v < 0 || v >= H) continue;
        const float cur = depth[v * W + u];
        if (cur >= 0.001f && cur < zfar && fabsf(cur - mean) < 0.01f) {
          const float a = -(float)((u - w) * (u - w) + (h - v) * (h - v)) / two_sd2;
          const float b = (dC - cur) * (dC - cur) / two_sr2;
          const float wt = expf(a - b);
          sw += wt;
          s += wt * cur;
        }
      }
    }
    if (sw > 0.f) res = s / sw;
  }
  out[h * W + w] = res;
}

// ---------------------------------------------------------------- a3 (Utils.py:399-438)
__global__ void k_depth_to_xyz(const float* __restrict__ depth, fp_k9d K, float zfar, int f64_internal,
                               float* __restrict__ xyz, int H, int W) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y * blockDim.y + threadIdx.y;
  if (u >= W || v >= H) return;
  const float z = depth[v * W + u];
  float x = 0.f, y = 0.f, zz = 0.f;
  if (f64_internal) {
    if (!(z < 0.001f)) {
      const double zd = (double)z;
      x = (float)(((double)u - K.v[2]) * zd / K.v[0]);
      y = (float)(((double)v - K.v[5]) * zd / K.v[4]);
      zz = z;
    }
  } else {
    if (!(z < 0.001f || z > zfar)) {
      const float fx = (float)K.v[0], fy = (float)K.v[4], cx = (float)K.v[2], cy = (float)K.v[5];
      x = (((float)u - cx) * z) / fx;
      y = (((float)v - cy) * z) / fy;
      zz = z;
    }
  }
  float* o = xyz + (size_t)(v * W + u) * 3;
  o[0] = x; o[1] = y; o[2] = zz;
}

// ---------------------------------------------------------------- a5+a6 (Utils.py:577-621, float64 internals)
__global__ void k_crop_windows(const float* __restrict__ poses, fp_k9d K, double radius, int out_w, int out_h,
                               int N, float* __restrict__ tfs, float* __restrict__ bbox) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* P = poses + (size_t)n * 16;
  const double tx = (double)P[3], ty = (double)P[7], tz = (double)P[11];
  double u0 = 0.0, v0 = 0.0, rad = 0.0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double ox = (k == 1) ? radius : ((k == 2) ? -radius : 0.0);
    const double oy = (k == 3) ? radius : ((k == 4) ? -radius : 0.0);
    const double x = tx + ox, y = ty + oy, z = tz;
    const double px = (K.v[0] * x + K.v[1] * y) + K.v[2] * z;
    const double py = (K.v[3] * x + K.v[4] * y) + K.v[5] * z;
    const double pz = (K.v[6] * x + K.v[7] * y) + K.v[8] * z;
    const double u = px / pz, v = py / pz;
    if (k == 0) { u0 = u; v0 = v; }
    const double a = fabs(u - u0), b = fabs(v - v0);
    rad = fmax(rad, fmax(a, b));
  }
  const double left = nearbyint(u0 - rad), right = nearbyint(u0 + rad);
  const double top = nearbyint(v0 - rad), bottom = nearbyint(v0 + rad);
  const float sx = (float)((double)out_w / (right - left));
  const float sy = (float)((double)out_h / (bottom - top));
  const float ntx = (float)(-left), nty = (float)(-top);
  float* tf = tfs + (size_t)n * 9;
  const float t02 = sx * ntx, t12 = sy * nty;
  tf[0] = sx;  tf[1] = 0.f; tf[2] = t02;
  tf[3] = 0.f; tf[4] = sy;  tf[5] = t12;
  tf[6] = 0.f; tf[7] = 0.f; tf[8] = 1.f;
  const float i00 = 1.0f / sx, i11 = 1.0f / sy;
  const float i02 = (-t02) / sx, i12 = (-t12) / sy;
  float* bb = bbox + (size_t)n * 4;
  bb[0] = i02;
  bb[1] = i12;
  bb[2] = (i00 * (float)(out_w - 1)) + i02;
  bb[3] = (i11 * (float)(out_h - 1)) + i12;
}

// ---------------------------------------------------------------- a13 (predict_pose_refine.py:195-234)
struct fp_f3 { float v[3]; };

__global__ void k_pose_update(const float* __restrict__ trans, const float* __restrict__ rot,
                              const float* __restrict__ poses_in, int rot_rep, int normalize_xyz, fp_f3 tn,
                              float rot_normalizer, float mesh_diameter, int N, float* __restrict__ poses_out,
                              float* __restrict__ trans_delta_out, float* __restrict__ rot_delta_out, int trans_rep, fp_k9 K,
                              const float* __restrict__ tf_to_crops, float input_w) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float dt[3];
  if (trans_rep == FP_TRANS_DEEPIM) {
    // predict_pose_refine.py:201-215: (trans.x, trans.y) = shift of the projected object centre in crop pixels / crop
    // width, trans.z = new depth / current depth.  tf_to_crops = [[sx,0,tx],[0,sy,ty],[0,0,1]], K upper triangular:
    // both inverses in closed form.
    const float* A = poses_in + (size_t)n * 16;
    const float* tf = tf_to_crops + (size_t)n * 9;
    const float tx = A[3], ty = A[7], tz = A[11];
    const float u = (K.v[0] * tx + K.v[1] * ty + K.v[2] * tz) / tz, v = (K.v[4] * ty + K.v[5] * tz) / tz;
    const float uc = tf[0] * u + tf[1] * v + tf[2], vc = tf[3] * u + tf[4] * v + tf[5];
    const float z_pred = trans[n * 3 + 2] * tz;
    const float ucp = uc + trans[n * 3] * input_w, vcp = vc + trans[n * 3 + 1] * input_w;
    const float vp = (vcp - tf[5]) / tf[4];
    const float up = ((ucp - tf[2]) - tf[1] * vp) / tf[0];
    const float yn = (vp - K.v[5]) / K.v[4];
    const float xn = ((up - K.v[2]) - K.v[1] * yn) / K.v[0];
    dt[0] = xn * z_pred - tx; dt[1] = yn * z_pred - ty; dt[2] = z_pred - tz;
    if (normalize_xyz) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dt[c] *= (mesh_diameter / 2.0f);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = trans[n * 3 + c];
      // 'tracknet' (:195-199) squashes the raw output only when the crops are NOT normalised; any other trans_rep is the
      // plain `else` (:217-218): the raw output.  With normalize_xyz the two coincide (:232-233).
      if (!normalize_xyz) v = (trans_rep == FP_TRANS_RAW) ? v : tanhf(v) * tn.v[c];
      else v = v * (mesh_diameter / 2.0f);
      dt[c] = v;
    }
  }
  float R[9];
  if (rot_rep == FP_ROT_AXIS_ANGLE) {
    float w[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) w[c] = tanhf(rot[n * 3 + c]) * rot_normalizer;
    const float n2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    const float th = sqrtf(fmaxf(n2, 1e-4f));
    const float ith = 1.0f / th;
    const float f1 = ith * sinf(th);
    const float f2 = (ith * ith) * (1.0f - cosf(th));
    const float Kx[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float k2 = (Kx[r * 3] * Kx[c] + Kx[r * 3 + 1] * Kx[3 + c]) + Kx[r * 3 + 2] * Kx[6 + c];
        R[r * 3 + c] = (f1 * Kx[r * 3 + c] + f2 * k2) + ((r == c) ? 1.0f : 0.0f);
      }
  } else {
    const float* d = rot + (size_t)n * 6;
    const float a1[3] = {d[0], d[1], d[2]}, a2[3] = {d[3], d[4], d[5]};
    const float l1 = fmaxf(sqrtf((a1[0] * a1[0] + a1[1] * a1[1]) + a1[2] * a1[2]), 1e-12f);
    const float b1[3] = {a1[0] / l1, a1[1] / l1, a1[2] / l1};
    const float dp = (b1[0] * a2[0] + b1[1] * a2[1]) + b1[2] * a2[2];
    const float u2[3] = {a2[0] - dp * b1[0], a2[1] - dp * b1[1], a2[2] - dp * b1[2]};
    const float l2 = fmaxf(sqrtf((u2[0] * u2[0] + u2[1] * u2[1]) + u2[2] * u2[2]), 1e-12f);
    const float b2[3] = {u2[0] / l2, u2[1] / l2, u2[2] / l2};
    const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
#pragma unroll
    for (int c = 0; c < 3; ++c) { R[c] = b1[c]; R[3 + c] = b2[c]; R[6 + c] = b3[c]; }
  }
  const float* A = poses_in + (size_t)n * 16;
  float* O = poses_out + (size_t)n * 16;
  float o[16];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      o[r * 4 + c] = (R[0 * 3 + r] * A[0 * 4 + c] + R[1 * 3 + r] * A[1 * 4 + c]) + R[2 * 3 + r] * A[2 * 4 + c];
    o[r * 4 + 3] = A[r * 4 + 3] + dt[r];
  }
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) O[k] = o[k];
  // what the reference keeps as last_trans_update / last_rot_update (predict_pose_refine.py:238-239): the metric
  // translation delta and the applied rotation so3_exp_map(w)^T
  if (trans_delta_out) {
#pragma unroll
    for (int c = 0; c < 3; ++c) trans_delta_out[n * 3 + c] = dt[c];
  }
  if (rot_delta_out) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) rot_delta_out[(size_t)n * 9 + r * 3 + c] = R[c * 3 + r];
  }
}

// ---------------------------------------------------------------- C ABI
extern "C" int fp_depth_erode(const float* depth, float* out, int H, int W, int radius, float diff_thres,
                              float ratio_thres, float zfar, void* stream) {
  FP_REQUIRE(depth && out && H > 0 && W > 0 && radius >= 0, "fp_depth_erode: bad arguments");
  dim3 b(64, 4), g(fp_cdiv(W, 64), fp_cdiv(H, 4));
  hipLaunchKernelGGL(k_erode, g, b, 0, (hipStream_t)stream, depth, out, H, W, radius, diff_thres, ratio_thres, zfar);
  FP_CHECK_LAUNCH("fp_depth_erode");
  return FP_OK;
}

extern "C" int fp_depth_bilateral(const float* depth, float* out, int H, int W, int radius, float zfar,
                                  float sigmaD, float sigmaR, void* stream) {
  FP_REQUIRE(depth && out && H > 0 && W > 0 && radius >= 0, "fp_depth_bilateral: bad arguments");
  dim3 b(64, 4), g(fp_cdiv(W, 64), fp_cdiv(H, 4));
  hipLaunchKernelGGL(k_bilateral, g, b, 0, (hipStream_t)stream, depth, out, H, W, radius, zfar, sigmaD, sigmaR);
  FP_CHECK_LAUNCH("fp_depth_bilateral");
  return FP_OK;
}

extern "C" int fp_depth_to_xyz(const float* depth, const double* K, float zfar, int f64_internal, float* xyz,
                               int H, int W, void* stream) {
  FP_REQUIRE(depth && K && xyz && H > 0 && W > 0, "fp_depth_to_xyz: bad arguments");
  fp_k9d Kd;
  for (int i = 0; i < 9; ++i) Kd.v[i] = K[i];
  dim3 b(64, 4), g(fp_cdiv(W, 64), fp_cdiv(H, 4));
  hipLaunchKernelGGL(k_depth_to_xyz, g, b, 0, (hipStream_t)stream, depth, Kd, zfar, f64_internal, xyz, H, W);
  FP_CHECK_LAUNCH("fp_depth_to_xyz");
  return FP_OK;
}

extern "C" int fp_crop_windows(const float* poses, const double* K, double mesh_diameter, double crop_ratio,
                               int out_w, int out_h, int N, float* tf_to_crops, float* bbox2d, void* stream) {
  FP_REQUIRE(N >= 0, "fp_crop_windows: N < 0");
  if (N == 0) return FP_OK;
  FP_REQUIRE(poses && K && tf_to_crops && bbox2d && out_w > 1 && out_h > 1, "fp_crop_windows: bad arguments");
  fp_k9d Kd;
  for (int i = 0; i < 9; ++i) Kd.v[i] = K[i];
  const double radius = mesh_diameter * crop_ratio / 2.0;
  hipLaunchKernelGGL(k_crop_windows, dim3(fp_cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, poses, Kd, radius,
                     out_w, out_h, N, tf_to_crops, bbox2d);
  FP_CHECK_LAUNCH("fp_crop_windows");
  return FP_OK;
}

extern "C" int fp_pose_update(const float* trans, const float* rot, const float* poses_in, int rot_rep,
                              int normalize_xyz, const float* trans_normalizer, float rot_normalizer,
                              float mesh_diameter, int N, float* poses_out, float* trans_delta_out, float* rot_delta_out,
                              int trans_rep, const float* K9, const float* tf_to_crops, float input_w, void* stream) {
  FP_REQUIRE(N >= 0, "fp_pose_update: N < 0");
  if (N == 0) return FP_OK;
  FP_REQUIRE(trans && rot && poses_in && poses_out, "fp_pose_update: NULL tensor");
  FP_REQUIRE(rot_rep == FP_ROT_AXIS_ANGLE || rot_rep == FP_ROT_6D, "fp_pose_update: unknown rot_rep %d", rot_rep);
  FP_REQUIRE(trans_rep == FP_TRANS_TRACKNET || trans_rep == FP_TRANS_DEEPIM || trans_rep == FP_TRANS_RAW, "fp_pose_update: unknown trans_rep %d", trans_rep);
  FP_REQUIRE(trans_rep != FP_TRANS_DEEPIM || (K9 && tf_to_crops && input_w > 0.f),
             "fp_pose_update: trans_rep deepim needs K, tf_to_crops and the crop width");
  fp_k9 Kk = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}};
  if (K9) for (int i = 0; i < 9; ++i) Kk.v[i] = K9[i];
  fp_f3 tn = {{1.f, 1.f, 1.f}};
  if (trans_normalizer) { tn.v[0] = trans_normalizer[0]; tn.v[1] = trans_normalizer[1]; tn.v[2] = trans_normalizer[2]; }
  hipLaunchKernelGGL(k_pose_update, dim3(fp_cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, trans, rot, poses_in,
                     rot_rep, normalize_xyz, tn, rot_normalizer, mesh_diameter, N, poses_out, trans_delta_out, rot_delta_out,
                     trans_rep, Kk, tf_to_crops, input_w);
  FP_CHECK_LAUNCH("fp_pose_update");
  return FP_OK;
}
