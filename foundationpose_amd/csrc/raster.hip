// Fused pose-hypothesis rasteriser for gfx950 (replaces nvdiffrast_render, Utils.py:133-219, plus the
// A-side of make_crop_data_batch / transform_batch / concat -- see include/fp_amd.h).
//
// One workgroup = (hypothesis n, strip of SH output rows).  Phases, all inside one launch:
//   1. vertex pass   : every vertex of the mesh is transformed with pose n, projected into the crop,
//                      snapped to 1/16 px and cached in LDS as {int16 x, int16 y, f32 1/z} (8 B / vertex)
//   2. raster pass   : triangle-parallel (one lane per triangle, the typical ~4 px triangle makes a
//                      pixel-parallel scan of the bin ~10x more work); integer edge functions with a
//                      top-left tie rule give exact coverage, the depth key
//                      (round(z_cam * 2^20) << 32 | tri_id) goes into the LDS strip z-buffer with a
//                      64-bit ds_min -- deterministic winner, independent of arrival order
//   3. resolve pass  : pixel-parallel, consecutive lanes = consecutive pixels of a row (coalesced stores);
//                      perspective-correct barycentrics of the winner, attribute interpolation,
//                      bilinear texture, Lambert shading, (x - t)/radius normalisation, masks, and the
//                      network tensor A[n, 0:6] is written directly (fp16 or fp32), so no intermediate
//                      image ever reaches HBM.
// Compiled with -ffp-contract=off: the float expression order below is the definition shared with the
// CPU oracle (oracle/fp_oracle.c) and is what makes zbuf / tri_id bit-exact across CPU and GPU.
#include <hip/hip_fp16.h>
#include "fp_common.h"

#define FP_SUBPIX 16.0f
#define FP_GUARD_LO (-8192.0f)
#define FP_GUARD_HI (24575.0f)
#define FP_ZNEAR 0.001f
#define FP_ZMAXF 4095.0f
#define FP_ZSCALEF 1048576.0f
#define FP_VTX_INVALID 0x80008000u
#define FP_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull
#define FP_RASTER_THREADS 256

struct HypConst {
  float P[12];             // rows of [R|t]
  float umin, vmin, ax, ay;
  float fx, sk, cx, fy, cy;
};

struct RenderOut {
  void* A;
  float* color;
  float* depth;
  float* xyz;
  float* normal;
  uint32_t* zbuf;
  int32_t* tri_id;
};

struct VtxRec {            // LDS / workspace vertex cache entry
  uint32_t xy;             // int16 x | int16 y << 16, snapped to 1/16 px; FP_VTX_INVALID if culled
  float iw;                // 1 / z_cam
};

__device__ __forceinline__ HypConst load_hyp(const float* __restrict__ poses, const float* __restrict__ bbox2d,
                                             const fp_k9& K, int n, int H, int W, int oh, int ow) {
  HypConst h;
  const float* P = poses + (size_t)n * 16;
#pragma unroll
  for (int i = 0; i < 12; ++i) h.P[i] = P[i];
  float umin = 0.f, vmin = 0.f, umax = (float)W, vmax = (float)H;
  if (bbox2d) {
    umin = bbox2d[n * 4 + 0]; vmin = bbox2d[n * 4 + 1]; umax = bbox2d[n * 4 + 2]; vmax = bbox2d[n * 4 + 3];
  }
  h.umin = umin; h.vmin = vmin;
  h.ax = (float)ow / (umax - umin);
  h.ay = (float)oh / (vmax - vmin);
  h.fx = K.v[0]; h.sk = K.v[1]; h.cx = K.v[2]; h.fy = K.v[4]; h.cy = K.v[5];
  return h;
}

__device__ __forceinline__ void cam_point(const HypConst& h, float vx, float vy, float vz, float& xc, float& yc,
                                          float& zc) {
  xc = fmaf(h.P[0], vx, fmaf(h.P[1], vy, fmaf(h.P[2], vz, h.P[3])));
  yc = fmaf(h.P[4], vx, fmaf(h.P[5], vy, fmaf(h.P[6], vz, h.P[7])));
  zc = fmaf(h.P[8], vx, fmaf(h.P[9], vy, fmaf(h.P[10], vz, h.P[11])));
}

// unsnapped crop-pixel position of a camera-space point (same expression as project_vertex / the oracle)
__device__ __forceinline__ void crop_xy(const HypConst& h, float xc, float yc, float zc, float& X, float& Y) {
  const float iw = 1.0f / zc;
  const float skyc = h.sk * yc;
  const float pu = fmaf(h.fx, xc, skyc);
  const float pv = h.fy * yc;
  const float u = fmaf(pu, iw, h.cx);
  const float v = fmaf(pv, iw, h.cy);
  X = (u - h.umin) * h.ax;
  Y = (v - h.vmin) * h.ay;
}

__device__ __forceinline__ VtxRec project_vertex(const HypConst& h, float vx, float vy, float vz) {
  float xc, yc, zc;
  cam_point(h, vx, vy, vz, xc, yc, zc);
  bool ok = zc > FP_ZNEAR;
  const float iw = 1.0f / zc;
  const float skyc = h.sk * yc;
  const float pu = fmaf(h.fx, xc, skyc);
  const float pv = h.fy * yc;
  const float u = fmaf(pu, iw, h.cx);
  const float v = fmaf(pv, iw, h.cy);
  const float X = (u - h.umin) * h.ax;
  const float Y = (v - h.vmin) * h.ay;
  const float xs = rintf(X * FP_SUBPIX), ys = rintf(Y * FP_SUBPIX);
  ok = ok && (xs >= FP_GUARD_LO) && (xs <= FP_GUARD_HI) && (ys >= FP_GUARD_LO) && (ys <= FP_GUARD_HI);
  VtxRec r;
  r.iw = iw;
  if (ok) {
    const int xi = (int)xs, yi = (int)ys;
    r.xy = ((uint32_t)xi & 0xFFFFu) | ((uint32_t)yi << 16);
  } else {
    r.xy = FP_VTX_INVALID;
  }
  return r;
}

__device__ __forceinline__ int vx_of(uint32_t xy) { return (int)(int16_t)(xy & 0xFFFFu); }
__device__ __forceinline__ int vy_of(uint32_t xy) { return (int)(int16_t)(xy >> 16); }
__device__ __forceinline__ bool edge_owner(int dx, int dy) { return (dy > 0) || (dy == 0 && dx < 0); }

struct TriSetup {
  int x0, y0, x1, y1, x2, y2;
  int area2;
  int b0, b1, b2;
  int s1, s2;  // face slots of weights 1 and 2 (1,2 or swapped 2,1); slot of weight 0 is always 0
};

// vertex records r0,r1,r2 are in face order.  Returns false when the triangle is skipped.
__device__ __forceinline__ bool tri_setup(const VtxRec& r0, const VtxRec& r1, const VtxRec& r2, TriSetup& t) {
  if (r0.xy == FP_VTX_INVALID || r1.xy == FP_VTX_INVALID || r2.xy == FP_VTX_INVALID) return false;
  int x0 = vx_of(r0.xy), y0 = vy_of(r0.xy);
  int x1 = vx_of(r1.xy), y1 = vy_of(r1.xy);
  int x2 = vx_of(r2.xy), y2 = vy_of(r2.xy);
  int area2 = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
  if (area2 == 0) return false;
  t.s1 = 1; t.s2 = 2;
  if (area2 < 0) {
    int tx = x1, ty = y1;
    x1 = x2; y1 = y2; x2 = tx; y2 = ty;
    t.s1 = 2; t.s2 = 1;
    area2 = -area2;
  }
  t.x0 = x0; t.y0 = y0; t.x1 = x1; t.y1 = y1; t.x2 = x2; t.y2 = y2;
  t.area2 = area2;
  t.b0 = edge_owner(x2 - x1, y2 - y1) ? 0 : -1;
  t.b1 = edge_owner(x0 - x2, y0 - y2) ? 0 : -1;
  t.b2 = edge_owner(x1 - x0, y1 - y0) ? 0 : -1;
  return true;
}

__device__ __forceinline__ void tri_weights(const TriSetup& t, int px, int py, int& w0, int& w1, int& w2) {
  w0 = (t.x2 - t.x1) * (py - t.y1) - (t.y2 - t.y1) * (px - t.x1);
  w1 = (t.x0 - t.x2) * (py - t.y2) - (t.y0 - t.y2) * (px - t.x2);
  w2 = (t.x1 - t.x0) * (py - t.y0) - (t.y1 - t.y0) * (px - t.x0);
}

__device__ __forceinline__ float lerpf(float a, float b, float c) { return fmaf(c, b - a, a); }
__device__ __forceinline__ int wrapi(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }
__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

__device__ __forceinline__ void tex_fetch(const float* __restrict__ tex, int Ht, int Wt, float u, float v,
                                          float out[3]) {
  const float uu = fmaf(u, (float)Wt, -0.5f), vv = fmaf(v, (float)Ht, -0.5f);
  const float fu0 = floorf(uu), fv0 = floorf(vv);
  const float fu = uu - fu0, fv = vv - fv0;
  const int i0 = wrapi((int)fu0, Wt), i1 = wrapi((int)fu0 + 1, Wt);
  const int j0 = wrapi((int)fv0, Ht), j1 = wrapi((int)fv0 + 1, Ht);
  const float* t00 = tex + ((size_t)j0 * Wt + i0) * 3;
  const float* t10 = tex + ((size_t)j0 * Wt + i1) * 3;
  const float* t01 = tex + ((size_t)j1 * Wt + i0) * 3;
  const float* t11 = tex + ((size_t)j1 * Wt + i1) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a = lerpf(t00[c], t10[c], fu);
    const float b = lerpf(t01[c], t11[c], fu);
    out[c] = lerpf(a, b, fv);
  }
}

// Stand-alone vertex pass for meshes whose vertex cache does not fit in LDS (workspace path).
__global__ __launch_bounds__(256) void k_vertex_pass(fp_mesh m, const float* __restrict__ poses,
                                                     const float* __restrict__ bbox2d, fp_k9 K, int H, int W,
                                                     int oh, int ow, VtxRec* __restrict__ ws) {
  const int n = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= m.V) return;
  const HypConst h = load_hyp(poses, bbox2d, K, n, H, W, oh, ow);
  ws[(size_t)n * m.V + v] = project_vertex(h, m.pos[v * 3], m.pos[v * 3 + 1], m.pos[v * 3 + 2]);
}

template <bool VLDS>
__global__ __launch_bounds__(FP_RASTER_THREADS) void k_render(
    fp_mesh m, const float* __restrict__ poses, const float* __restrict__ bbox2d, fp_k9 K, int H, int W, int oh,
    int ow, int SH, float w_ambient, float w_diffuse, float inv_r, float xyz_thr, int flags, RenderOut out,
    const VtxRec* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = blockIdx.y;
  const int row0 = blockIdx.x * SH;
  const int rows = min(SH, oh - row0);
  const int npix = rows * ow;
  unsigned long long* zb = reinterpret_cast<unsigned long long*>(smem);
  VtxRec* vc = reinterpret_cast<VtxRec*>(smem + (size_t)SH * ow * sizeof(unsigned long long));
  const int tid = threadIdx.x;
  const HypConst h = load_hyp(poses, bbox2d, K, n, H, W, oh, ow);

  // ---- phase 1: vertex cache + z-buffer clear
  if (VLDS) {
    for (int v = tid; v < m.V; v += FP_RASTER_THREADS)
      vc[v] = project_vertex(h, m.pos[v * 3], m.pos[v * 3 + 1], m.pos[v * 3 + 2]);
  }
  const VtxRec* vsrc = VLDS ? vc : (ws + (size_t)n * m.V);
  for (int p = tid; p < npix; p += FP_RASTER_THREADS) zb[p] = FP_KEY_EMPTY;
  __syncthreads();

  // ---- phase 2: triangle-parallel raster into the LDS strip
  const int ylo = row0 * 16 + 8;                 // sub-pixel y of the first / last pixel centre of this strip
  const int yhi = (row0 + rows - 1) * 16 + 8;
  for (int t = tid; t < m.T; t += FP_RASTER_THREADS) {
    const int f0 = m.faces[t * 3], f1 = m.faces[t * 3 + 1], f2 = m.faces[t * 3 + 2];
    const VtxRec r0 = vsrc[f0], r1 = vsrc[f1], r2 = vsrc[f2];
    TriSetup tr;
    if (!tri_setup(r0, r1, r2, tr)) continue;
    const int miny = min(tr.y0, min(tr.y1, tr.y2)), maxy = max(tr.y0, max(tr.y1, tr.y2));
    if (maxy < ylo || miny > yhi) continue;
    const int minx = min(tr.x0, min(tr.x1, tr.x2)), maxx = max(tr.x0, max(tr.x1, tr.x2));
    int i0 = (minx - 8 + 15) >> 4, i1 = (maxx - 8) >> 4;
    int j0 = (miny - 8 + 15) >> 4, j1 = (maxy - 8) >> 4;
    i0 = max(i0, 0); i1 = min(i1, ow - 1);
    j0 = max(j0, row0); j1 = min(j1, row0 + rows - 1);
    if (i0 > i1 || j0 > j1) continue;
    const float iw0 = r0.iw;
    const float iw1 = (tr.s1 == 1) ? r1.iw : r2.iw;
    const float iw2 = (tr.s1 == 1) ? r2.iw : r1.iw;
    const float fE = (float)tr.area2;
    for (int j = j0; j <= j1; ++j) {
      for (int i = i0; i <= i1; ++i) {
        int w0, w1, w2;
        tri_weights(tr, 16 * i + 8, 16 * j + 8, w0, w1, w2);
        if (((w0 + tr.b0) | (w1 + tr.b1) | (w2 + tr.b2)) < 0) continue;
        const float g0 = (float)w0, g1 = (float)w1, g2 = (float)w2;
        const float S = fmaf(g2, iw2, fmaf(g1, iw1, g0 * iw0));
        const float z = fE / S;
        const float zc = fminf(z, FP_ZMAXF);
        const uint32_t zq = (uint32_t)rintf(zc * FP_ZSCALEF);
        const unsigned long long key = ((unsigned long long)zq << 32) | (uint32_t)t;
        atomicMin(&zb[(j - row0) * ow + i], key);
      }
    }
  }
  __syncthreads();

  // ---- phase 3: resolve + shade + write
  const size_t plane = (size_t)oh * ow;
  const float t0 = h.P[3], t1 = h.P[7], t2 = h.P[11];
  const bool normalize = (flags & FP_FLAG_NORMALIZE_XYZ) != 0;
  const bool out_f16 = (flags & FP_FLAG_OUT_F16) != 0;
  for (int p = tid; p < npix; p += FP_RASTER_THREADS) {
    const int jl = p / ow, i = p - jl * ow, j = row0 + jl;
    const unsigned long long key = zb[p];
    const bool covered = key != FP_KEY_EMPTY;
    float col[3] = {0.f, 0.f, 0.f}, pt[3] = {0.f, 0.f, 0.f}, nm[3] = {0.f, 0.f, 0.f};
    int tid_out = -1;
    if (covered) {
      const int t = (int)(uint32_t)(key & 0xFFFFFFFFull);
      tid_out = t;
      const int f[3] = {m.faces[t * 3], m.faces[t * 3 + 1], m.faces[t * 3 + 2]};
      // nvdiffrast's per-pixel pass: perspective-correct barycentrics of the winner from its
      // UNSNAPPED vertices in face order, p_k = z_k * (X_k - pixel centre), a0 = p1 x p2, ..., clamped (u, v), 1-u-v
      const int fa0 = f[0], fa1 = f[1], fa2 = f[2];
      float q0[3], q1[3], q2[3], X0, Y0, X1, Y1, X2, Y2;
      cam_point(h, m.pos[fa0 * 3], m.pos[fa0 * 3 + 1], m.pos[fa0 * 3 + 2], q0[0], q0[1], q0[2]);
      cam_point(h, m.pos[fa1 * 3], m.pos[fa1 * 3 + 1], m.pos[fa1 * 3 + 2], q1[0], q1[1], q1[2]);
      cam_point(h, m.pos[fa2 * 3], m.pos[fa2 * 3 + 1], m.pos[fa2 * 3 + 2], q2[0], q2[1], q2[2]);
      crop_xy(h, q0[0], q0[1], q0[2], X0, Y0);
      crop_xy(h, q1[0], q1[1], q1[2], X1, Y1);
      crop_xy(h, q2[0], q2[1], q2[2], X2, Y2);
      const float fxp = (float)i + 0.5f, fyp = (float)j + 0.5f;
      const float p0x = (X0 - fxp) * q0[2], p0y = (Y0 - fyp) * q0[2];
      const float p1x = (X1 - fxp) * q1[2], p1y = (Y1 - fyp) * q1[2];
      const float p2x = (X2 - fxp) * q2[2], p2y = (Y2 - fyp) * q2[2];
      const float m0a = p1x * p2y, m0b = p1y * p2x, m1a = p2x * p0y, m1b = p2y * p0x, m2a = p0x * p1y, m2b = p0y * p1x;
      const float a0 = m0a - m0b, a1 = m1a - m1b, a2 = m2a - m2b;
      const float iwb = 1.0f / ((a0 + a1) + a2);
      const float b0 = clamp01(a0 * iwb), b1 = clamp01(a1 * iwb);
      const float b2 = (1.0f - b0) - b1;
#pragma unroll
      for (int c = 0; c < 3; ++c) pt[c] = fmaf(b2, q2[c], fmaf(b1, q1[c], b0 * q0[c]));
      float base[3];
      if (m.tex) {
        const int32_t* fu = (m.uv_idx ? m.uv_idx : m.faces) + (size_t)t * 3;
        const int ua = fu[0], ub = fu[1], uc = fu[2];
        float tu = fmaf(b2, m.uv[uc * 2], fmaf(b1, m.uv[ub * 2], b0 * m.uv[ua * 2]));
        float tv = fmaf(b2, m.uv[uc * 2 + 1], fmaf(b1, m.uv[ub * 2 + 1], b0 * m.uv[ua * 2 + 1]));
        tu = tu - floorf(tu);
        tv = tv - floorf(tv);
        tex_fetch(m.tex, m.Ht, m.Wt, tu, tv, base);
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          base[c] = fmaf(b2, m.vcol[fa2 * 3 + c], fmaf(b1, m.vcol[fa1 * 3 + c], b0 * m.vcol[fa0 * 3 + c]));
      }
      const int fav[3] = {fa0, fa1, fa2};
      const float bb[3] = {b0, b1, b2};
      float nk[3][3], dk[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float* vn = m.nrm + (size_t)fav[k] * 3;
        nk[k][0] = fmaf(h.P[2], vn[2], fmaf(h.P[1], vn[1], h.P[0] * vn[0]));
        nk[k][1] = fmaf(h.P[6], vn[2], fmaf(h.P[5], vn[1], h.P[4] * vn[0]));
        nk[k][2] = fmaf(h.P[10], vn[2], fmaf(h.P[9], vn[1], h.P[8] * vn[0]));
        const float len = sqrtf(fmaf(nk[k][2], nk[k][2], fmaf(nk[k][1], nk[k][1], nk[k][0] * nk[k][0])));
        dk[k] = clamp01((-nk[k][2]) / fmaxf(len, 1e-12f));
      }
      float nsum[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) nsum[c] = fmaf(bb[2], nk[2][c], fmaf(bb[1], nk[1][c], bb[0] * nk[0][c]));
      const float dsum = fmaf(bb[2], dk[2], fmaf(bb[1], dk[1], bb[0] * dk[0]));
      const float nl = sqrtf(fmaf(nsum[2], nsum[2], fmaf(nsum[1], nsum[1], nsum[0] * nsum[0])));
      const float inl = fmaxf(nl, 1e-12f);
#pragma unroll
      for (int c = 0; c < 3; ++c) nm[c] = nsum[c] / inl;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float amb = base[c] * w_ambient;
        const float dif = (dsum * base[c]) * w_diffuse;
        col[c] = clamp01(amb + dif);
      }
    }
    const size_t o = (size_t)n * plane + (size_t)j * ow + i;
    if (out.zbuf) out.zbuf[o] = covered ? (uint32_t)(key >> 32) : FP_ZBUF_EMPTY;
    if (out.tri_id) out.tri_id[o] = tid_out;
    if (out.depth) out.depth[o] = pt[2];
    if (out.color) { out.color[o * 3] = col[0]; out.color[o * 3 + 1] = col[1]; out.color[o * 3 + 2] = col[2]; }
    if (out.xyz) { out.xyz[o * 3] = pt[0]; out.xyz[o * 3 + 1] = pt[1]; out.xyz[o * 3 + 2] = pt[2]; }
    if (out.normal) { out.normal[o * 3] = nm[0]; out.normal[o * 3 + 1] = nm[1]; out.normal[o * 3 + 2] = nm[2]; }
    if (out.A) {
      float a[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) a[c] = (col[c] * 255.0f) * (1.0f / 255.0f);  // torch GPU `/255.0` = mul by f32 reciprocal
      const bool invalid = pt[2] < xyz_thr;
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
      const size_t ao = (size_t)n * 6 * plane + (size_t)j * ow + i;
      if (out_f16) {
        __half* A = reinterpret_cast<__half*>(out.A);
#pragma unroll
        for (int c = 0; c < 6; ++c) A[ao + (size_t)c * plane] = __float2half_rn(a[c]);
      } else {
        float* A = reinterpret_cast<float*>(out.A);
#pragma unroll
        for (int c = 0; c < 6; ++c) A[ao + (size_t)c * plane] = a[c];
      }
    }
  }
}

// ---------------------------------------------------------------- host side
#define FP_LDS_BUDGET (160 * 1024)
#define FP_ZB_TARGET (40 * 160 * 8)   // 50 KiB strip z-buffer => 2-3 workgroups per CU

static int strip_rows(int oh, int ow) {
  int sh = FP_ZB_TARGET / (ow * 8);
  if (sh < 1) sh = 1;
  if (sh > oh) sh = oh;
  return sh;
}

static bool vertex_cache_in_lds(int V, int oh, int ow) {
  const size_t zb = (size_t)strip_rows(oh, ow) * ow * 8;
  return zb + (size_t)V * sizeof(VtxRec) <= 80 * 1024;  // keep two workgroups per CU resident
}

extern "C" size_t fp_workspace_bytes(int N, int V, int T, int oh, int ow) {
  (void)T;
  if (N <= 0 || V <= 0) return 0;
  if (vertex_cache_in_lds(V, oh, ow)) return 0;
  return (size_t)N * V * sizeof(VtxRec);
}

extern "C" int fp_render_crops(const fp_mesh* mesh, const float* poses, const float* bbox2d, const float* K9, int H,
                               int W, int N, int oh, int ow, float w_ambient, float w_diffuse, float mesh_diameter,
                               float xyz_thr, int flags, void* A, float* color, float* depth, float* xyz,
                               float* normal, uint32_t* zbuf, int32_t* tri_id, void* workspace,
                               size_t workspace_bytes, void* stream) {
  FP_REQUIRE(N >= 0, "fp_render_crops: N < 0");
  if (N == 0) return FP_OK;
  FP_REQUIRE(mesh && poses && K9, "fp_render_crops: NULL mesh/poses/K");
  FP_REQUIRE(oh > 0 && ow > 0 && oh <= 1024 && ow <= 1024, "fp_render_crops: output size %dx%d unsupported (max 1024)", oh, ow);
  FP_REQUIRE(bbox2d || (oh == H && ow == W), "fp_render_crops: full-frame render needs oh==H and ow==W");
  FP_REQUIRE(N <= 65535, "fp_render_crops: N=%d exceeds the grid limit; chunk the batch", N);
  fp_k9 K;
  for (int i = 0; i < 9; ++i) K.v[i] = K9[i];
  const int SH = strip_rows(oh, ow);
  const int nstrips = fp_cdiv(oh, SH);
  const bool vlds = vertex_cache_in_lds(mesh->V, oh, ow);
  const float inv_r = 1.0f / (mesh_diameter * 0.5f);
  RenderOut out = {A, color, depth, xyz, normal, zbuf, tri_id};
  hipStream_t st = (hipStream_t)stream;
  size_t lds = (size_t)SH * ow * 8;
  if (vlds) {
    lds += (size_t)mesh->V * sizeof(VtxRec);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FP_LDS_BUDGET);
      attr_set = true;
    }
    hipLaunchKernelGGL(k_render<true>, dim3(nstrips, N), dim3(FP_RASTER_THREADS), lds, st, *mesh, poses, bbox2d, K, H,
                       W, oh, ow, SH, w_ambient, w_diffuse, inv_r, xyz_thr, flags, out, (const VtxRec*)nullptr);
  } else {
    const size_t need = (size_t)N * mesh->V * sizeof(VtxRec);
    if (!workspace || workspace_bytes < need) {
      fp_set_error("fp_render_crops: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
      return FP_ERR_WORKSPACE;
    }
    static bool attr_set2 = false;
    if (!attr_set2) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FP_LDS_BUDGET);
      attr_set2 = true;
    }
    hipLaunchKernelGGL(k_vertex_pass, dim3(fp_cdiv(mesh->V, 256), N), dim3(256), 0, st, *mesh, poses, bbox2d, K, H, W,
                       oh, ow, (VtxRec*)workspace);
    FP_CHECK_LAUNCH("fp_render_crops(vertex pass)");
    hipLaunchKernelGGL(k_render<false>, dim3(nstrips, N), dim3(FP_RASTER_THREADS), lds, st, *mesh, poses, bbox2d, K, H,
                       W, oh, ow, SH, w_ambient, w_diffuse, inv_r, xyz_thr, flags, out, (const VtxRec*)workspace);
  }
  FP_CHECK_LAUNCH("fp_render_crops");
  return FP_OK;
}
