/*
 * fp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the image-space arithmetic of the FoundationPose
 * render-and-compare hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (foundationpose_amd/) never does.
 *
 * PARITY STATUS: pinned against golden vectors minted by the reference's own Python for this path
 * (tests/golden/pipeline_golden.npz, pipeline_golden_wide.npz; tests/test_oracle_pipeline_golden*.py).  Still "parity unpinned": the
 * internals of nvdiffrast (rasterize/interpolate/texture), kornia (warp_perspective) and pytorch3d, which are
 * absent from /root/reference and from this container and are restated from their published semantics
 * (SURVEY.md App. B); NVIDIA Warp's two depth kernels are restated from their in-tree source (Utils.py:304-395).
 * The integer z-buffer is DEFINED here.
 *
 * Reference call sites restated (paths relative to /root/reference):
 *   fpo_erode_depth        Utils.py:359-395   (erode_depth_kernel)
 *   fpo_bilateral_depth    Utils.py:304-356   (bilateral_filter_depth_kernel)
 *   fpo_depth_to_xyz       Utils.py:399-417 (f64 internals) / :420-438 (f32)
 *   fpo_crop_windows       Utils.py:577-621 + predict_pose_refine.py:44-45
 *   fpo_render_crops       Utils.py:133-219 (nvdiffrast_render) +
 *                          predict_pose_refine.py:54-56, h5_dataset.py:79-114
 *   fpo_warp_crops         predict_pose_refine.py:63,72; predict_score.py:89-90;
 *                          h5_dataset.py:79-114 (refine) / :137-170 (score)
 *   fpo_pose_update        predict_pose_refine.py:195-234, Utils.py:848-855,
 *                          pytorch3d so3_exp_map / rotation_6d_to_matrix [3P]
 *   fpo_cluster_poses      mycpp/src/app/pybind_api.cpp:24-68, mycpp/src/Utils.cpp:21-26
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * All float expressions are written with explicit fmaf()/separate statements so
 * that the HIP kernels can mirror the operation order exactly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FPO_SUBPIX 16            /* 4 sub-pixel bits, as nvdiffrast's CudaRaster */
#define FPO_GUARD_LO (-8192)     /* snapped coordinate guard band: [-512 px, 1536 px) */
#define FPO_GUARD_HI (24575)
#define FPO_ZNEAR 0.001f
#define FPO_ZSCALE 1048576.0f    /* 2^20 fixed-point steps per metre */
#define FPO_ZMAX 4095.0f
#define FPO_EMPTY 0xFFFFFFFFFFFFFFFFull

#define FPO_FLAG_NORMALIZE_XYZ 1
#define FPO_MODE_REFINE 0
#define FPO_MODE_SCORE 1

void fpo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int fpo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ a1 */
void fpo_erode_depth(const float* depth, float* out, int H, int W, int radius,
                     float depth_diff_thres, float ratio_thres, float zfar) {
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; ++h) {
    for (int w = 0; w < W; ++w) {
      float d_ori = depth[h * W + w];
      float bad = 0.f, total = 0.f;
      for (int u = w - radius; u <= w + radius; ++u) {
        if (u < 0 || u >= W) continue;
        for (int v = h - radius; v <= h + radius; ++v) {
          if (v < 0 || v >= H) continue;
          float cur = depth[v * W + u];
          total += 1.0f;
          if (cur < 0.001f || cur >= zfar || fabsf(cur - d_ori) > depth_diff_thres) bad += 1.0f;
        }
      }
      /* the reference kernel does not early-return on an invalid centre (App. D.12) */
      out[h * W + w] = (bad / total > ratio_thres) ? 0.0f : d_ori;
    }
  }
}

/* ------------------------------------------------------------------ a2 */
void fpo_bilateral_depth(const float* depth, float* out, int H, int W, int radius,
                         float zfar, float sigmaD, float sigmaR) {
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; ++h) {
    for (int w = 0; w < W; ++w) {
      float res = 0.0f;
      float mean = 0.0f;
      int nvalid = 0;
      for (int u = w - radius; u <= w + radius; ++u) {
        if (u < 0 || u >= W) continue;
        for (int v = h - radius; v <= h + radius; ++v) {
          if (v < 0 || v >= H) continue;
          float cur = depth[v * W + u];
          if (cur >= 0.001f && cur < zfar) { nvalid++; mean += cur; }
        }
      }
      if (nvalid > 0) {
        mean /= (float)nvalid;
        float dC = depth[h * W + w];
        float sw = 0.f, s = 0.f;
        float two_sd2 = 2.0f * sigmaD * sigmaD;
        float two_sr2 = 2.0f * sigmaR * sigmaR;
        for (int u = w - radius; u <= w + radius; ++u) {
          if (u < 0 || u >= W) continue;
          for (int v = h - radius; v <= h + radius; ++v) {
            if (v < 0 || v >= H) continue;
            float cur = depth[v * W + u];
            if (cur >= 0.001f && cur < zfar && fabsf(cur - mean) < 0.01f) {
              float a = -(float)((u - w) * (u - w) + (h - v) * (h - v)) / two_sd2;
              float b = (dC - cur) * (dC - cur) / two_sr2;
              float wt = expf(a - b);
              sw += wt;
              s += wt * cur;
            }
          }
        }
        if (sw > 0.f) res = s / sw;
      }
      out[h * W + w] = res;
    }
  }
}

/* ------------------------------------------------------------------ a3 */
/* f64_internal=1: numpy variant (register), K promoted to f64, no zfar test;
 * f64_internal=0: torch batch variant (track_one / scorer), f32, zfar test. */
void fpo_depth_to_xyz(const float* depth, const double* K, float zfar, int f64_internal,
                      float* xyz, int H, int W) {
  const float fxf = (float)K[0], fyf = (float)K[4], cxf = (float)K[2], cyf = (float)K[5];
#pragma omp parallel for schedule(static)
  for (int v = 0; v < H; ++v) {
    for (int u = 0; u < W; ++u) {
      float z = depth[v * W + u];
      float* o = xyz + (size_t)(v * W + u) * 3;
      if (f64_internal) {
        if (z < 0.001f) { o[0] = o[1] = o[2] = 0.f; continue; }
        double zd = (double)z;
        o[0] = (float)(((double)u - K[2]) * zd / K[0]);
        o[1] = (float)(((double)v - K[5]) * zd / K[4]);
        o[2] = z;
      } else {
        if (z < 0.001f || z > zfar) { o[0] = o[1] = o[2] = 0.f; continue; }
        o[0] = (((float)u - cxf) * z) / fxf;
        o[1] = (((float)v - cyf) * z) / fyf;
        o[2] = z;
      }
    }
  }
}

/* ------------------------------------------------------------ a5 + a6 */
static double rne_d(double x) { return nearbyint(x); } /* default rounding mode = half-even */

void fpo_crop_windows(const float* poses, const double* K, double mesh_diameter,
                      double crop_ratio, int out_w, int out_h, int N,
                      float* tf_to_crops /*N,9*/, float* bbox2d /*N,4*/) {
  const double radius = mesh_diameter * crop_ratio / 2.0;
  const double off[5][3] = {{0, 0, 0}, {radius, 0, 0}, {-radius, 0, 0}, {0, radius, 0}, {0, -radius, 0}};
  for (int n = 0; n < N; ++n) {
    const float* P = poses + (size_t)n * 16;
    double tx = (double)P[3], ty = (double)P[7], tz = (double)P[11];
    double uu[5], vv[5];
    for (int k = 0; k < 5; ++k) {
      double x = tx + off[k][0], y = ty + off[k][1], z = tz + off[k][2];
      double px = (K[0] * x + K[1] * y) + K[2] * z;
      double py = (K[3] * x + K[4] * y) + K[5] * z;
      double pz = (K[6] * x + K[7] * y) + K[8] * z;
      uu[k] = px / pz;
      vv[k] = py / pz;
    }
    double rad = 0.0;
    for (int k = 0; k < 5; ++k) {
      double a = fabs(uu[k] - uu[0]), b = fabs(vv[k] - vv[0]);
      if (a > rad) rad = a;
      if (b > rad) rad = b;
    }
    double left = rne_d(uu[0] - rad), right = rne_d(uu[0] + rad);
    double top = rne_d(vv[0] - rad), bottom = rne_d(vv[0] + rad);
    float sx = (float)((double)out_w / (right - left));
    float sy = (float)((double)out_h / (bottom - top));
    float ntx = (float)(-left), nty = (float)(-top);
    float* tf = tf_to_crops + (size_t)n * 9;
    tf[0] = sx;  tf[1] = 0.f; tf[2] = sx * ntx;
    tf[3] = 0.f; tf[4] = sy;  tf[5] = sy * nty;
    tf[6] = 0.f; tf[7] = 0.f; tf[8] = 1.f;
    /* closed-form inverse of the scale+translate tf, applied to (0,0),(ow-1,oh-1) */
    float i00 = 1.0f / sx, i11 = 1.0f / sy;
    float i02 = (-tf[2]) / sx, i12 = (-tf[5]) / sy;
    float* bb = bbox2d + (size_t)n * 4;
    bb[0] = i02;
    bb[1] = i12;
    bb[2] = (i00 * (float)(out_w - 1)) + i02;
    bb[3] = (i11 * (float)(out_h - 1)) + i12;
  }
}

/* ------------------------------------------------------------------ a8 */
typedef struct {
  int32_t xi, yi;       /* crop-pixel position snapped to 1/16 px (coverage + integer depth key) */
  float X, Y;           /* unsnapped crop-pixel position (per-pixel barycentrics, nvdiffrast semantics B.1) */
  float iw;
  float xc, yc, zc;
  int valid;
} fpo_vtx;

/* tie rule for a pixel centre exactly on an edge (vertices oriented to positive area in THIS, y-down, crop space): the
 * top-left rule of a rasteriser working in nvdiffrast's y-up window space -- the reference flips the rows AFTER
 * rasterising (Utils.py:216-218) -- seen from here: an edge owns its points if it runs downwards in window space
 * (dy > 0 in both spaces, because the orientation fix reverses the edge when the rows are flipped), and a horizontal
 * edge if it runs towards -x there = towards +x here. */
static inline int edge_owner(int dx, int dy) { return (dy > 0) || (dy == 0 && dx > 0); }

typedef struct {
  int a0, a1, a2;       /* vertex slots after orientation fix (0..2 into the face) */
  int32_t x0, y0, x1, y1, x2, y2;
  int32_t area2;
  int b0, b1, b2;       /* tie-rule bias per barycentric weight */
} fpo_tri;

/* returns 0 if the triangle is skipped */
static int tri_setup(const fpo_vtx* v, const int* f, fpo_tri* t) {
  const fpo_vtx *p0 = &v[f[0]], *p1 = &v[f[1]], *p2 = &v[f[2]];
  if (!(p0->valid && p1->valid && p2->valid)) return 0;
  int32_t area2 = (p1->xi - p0->xi) * (p2->yi - p0->yi) - (p1->yi - p0->yi) * (p2->xi - p0->xi);
  if (area2 == 0) return 0;
  t->a0 = 0; t->a1 = 1; t->a2 = 2;
  if (area2 < 0) { const fpo_vtx* tmp = p1; p1 = p2; p2 = tmp; t->a1 = 2; t->a2 = 1; area2 = -area2; }
  t->x0 = p0->xi; t->y0 = p0->yi; t->x1 = p1->xi; t->y1 = p1->yi; t->x2 = p2->xi; t->y2 = p2->yi;
  t->area2 = area2;
  /* weight k is the edge function of the edge opposite vertex k */
  t->b0 = edge_owner(t->x2 - t->x1, t->y2 - t->y1) ? 0 : -1;
  t->b1 = edge_owner(t->x0 - t->x2, t->y0 - t->y2) ? 0 : -1;
  t->b2 = edge_owner(t->x1 - t->x0, t->y1 - t->y0) ? 0 : -1;
  return 1;
}

static inline void tri_weights(const fpo_tri* t, int32_t px, int32_t py, int32_t* w0, int32_t* w1, int32_t* w2) {
  *w0 = (t->x2 - t->x1) * (py - t->y1) - (t->y2 - t->y1) * (px - t->x1);
  *w1 = (t->x0 - t->x2) * (py - t->y2) - (t->y0 - t->y2) * (px - t->x2);
  *w2 = (t->x1 - t->x0) * (py - t->y0) - (t->y1 - t->y0) * (px - t->x0);
}

static inline int floordiv16(int32_t a) { return a >> 4; } /* arithmetic shift = floor */

static inline float lerpf(float a, float b, float c) { return fmaf(c, b - a, a); }

static inline int wrapi(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }

static void tex_fetch(const float* tex, int Ht, int Wt, float u, float v, float out[3]) {
  float uu = fmaf(u, (float)Wt, -0.5f), vv = fmaf(v, (float)Ht, -0.5f);
  float fu0 = floorf(uu), fv0 = floorf(vv);
  float fu = uu - fu0, fv = vv - fv0;
  int i0 = wrapi((int)fu0, Wt), i1 = wrapi((int)fu0 + 1, Wt);
  int j0 = wrapi((int)fv0, Ht), j1 = wrapi((int)fv0 + 1, Ht);
  const float* t00 = tex + ((size_t)j0 * Wt + i0) * 3;
  const float* t10 = tex + ((size_t)j0 * Wt + i1) * 3;
  const float* t01 = tex + ((size_t)j1 * Wt + i0) * 3;
  const float* t11 = tex + ((size_t)j1 * Wt + i1) * 3;
  for (int c = 0; c < 3; ++c) {
    float a = lerpf(t00[c], t10[c], fu);
    float b = lerpf(t01[c], t11[c], fu);
    out[c] = lerpf(a, b, fv);
  }
}

static inline float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

/*
 * Renders N hypotheses of one mesh into per-hypothesis (oh,ow) crops.
 * bbox2d (N,4) = (umin,vmin,umax,vmax) image window mapped onto the crop (null => full frame, needs oh=H, ow=W).
 * Outputs (any may be NULL):
 *   A      (N,6,oh,ow) f32 : network-ready [rgb, xyz-normalised]        (a8 + a10 + a11)
 *   color  (N,oh,ow,3), depth (N,oh,ow), xyz (N,oh,ow,3), normal (N,oh,ow,3)   nvdiffrast_render outputs
 *   zbuf   (N,oh,ow) u32 fixed-point depth (0xFFFFFFFF empty), tri_id (N,oh,ow) i32 (-1 empty)
 */
void fpo_render_crops(const float* pos, const float* nrm, const int* faces, const float* uv,
                      const int* uv_idx, const float* tex, int Ht, int Wt, const float* vcol, int V, int T,
                      const float* poses, const float* bbox2d, const float* K9, int H, int W, int N,
                      int oh, int ow, float w_ambient, float w_diffuse, float mesh_diameter,
                      float xyz_thr, int flags,
                      float* A, float* color, float* depth, float* xyz, float* normal,
                      uint32_t* zbuf, int32_t* tri_id) {
  const float fx = K9[0], sk = K9[1], cx = K9[2], fy = K9[4], cy = K9[5];
  const size_t npx = (size_t)oh * ow;
  const int normalize = flags & FPO_FLAG_NORMALIZE_XYZ;
  const float inv_r = 1.0f / (mesh_diameter * 0.5f);
  const float inv255 = 1.0f / 255.0f;
#pragma omp parallel
  {
    fpo_vtx* vt = (fpo_vtx*)malloc(sizeof(fpo_vtx) * (size_t)V);
    uint64_t* zb = (uint64_t*)malloc(sizeof(uint64_t) * npx);
#pragma omp for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
      const float* P = poses + (size_t)n * 16;
      float umin = 0.f, vmin = 0.f, umax = (float)W, vmax = (float)H;
      if (bbox2d) { umin = bbox2d[n * 4 + 0]; vmin = bbox2d[n * 4 + 1]; umax = bbox2d[n * 4 + 2]; vmax = bbox2d[n * 4 + 3]; }
      const float ax = (float)ow / (umax - umin);
      const float ay = (float)oh / (vmax - vmin);
      /* ---- vertex pass */
      for (int i = 0; i < V; ++i) {
        const float vx = pos[i * 3], vy = pos[i * 3 + 1], vz = pos[i * 3 + 2];
        fpo_vtx* o = &vt[i];
        o->xc = fmaf(P[0], vx, fmaf(P[1], vy, fmaf(P[2], vz, P[3])));
        o->yc = fmaf(P[4], vx, fmaf(P[5], vy, fmaf(P[6], vz, P[7])));
        o->zc = fmaf(P[8], vx, fmaf(P[9], vy, fmaf(P[10], vz, P[11])));
        int ok = (o->zc > FPO_ZNEAR);
        float iw = 1.0f / o->zc;
        float skyc = sk * o->yc;
        float pu = fmaf(fx, o->xc, skyc);
        float pv = fy * o->yc;
        float u = fmaf(pu, iw, cx);
        float v = fmaf(pv, iw, cy);
        float X = (u - umin) * ax;
        float Y = (v - vmin) * ay;
        float xs = rintf(X * (float)FPO_SUBPIX), ys = rintf(Y * (float)FPO_SUBPIX);
        ok = ok && (xs >= (float)FPO_GUARD_LO) && (xs <= (float)FPO_GUARD_HI) &&
             (ys >= (float)FPO_GUARD_LO) && (ys <= (float)FPO_GUARD_HI);
        o->valid = ok;
        o->X = X;
        o->Y = Y;
        o->iw = iw;
        o->xi = ok ? (int32_t)xs : 0;
        o->yi = ok ? (int32_t)ys : 0;
      }
      /* ---- raster pass: integer coverage + fixed-point depth keys */
      for (size_t p = 0; p < npx; ++p) zb[p] = FPO_EMPTY;
      for (int t = 0; t < T; ++t) {
        fpo_tri tr;
        const int* f = faces + (size_t)t * 3;
        if (!tri_setup(vt, f, &tr)) continue;
        int32_t minx = tr.x0 < tr.x1 ? tr.x0 : tr.x1; if (tr.x2 < minx) minx = tr.x2;
        int32_t maxx = tr.x0 > tr.x1 ? tr.x0 : tr.x1; if (tr.x2 > maxx) maxx = tr.x2;
        int32_t miny = tr.y0 < tr.y1 ? tr.y0 : tr.y1; if (tr.y2 < miny) miny = tr.y2;
        int32_t maxy = tr.y0 > tr.y1 ? tr.y0 : tr.y1; if (tr.y2 > maxy) maxy = tr.y2;
        int i0 = floordiv16(minx - 8 + 15), i1 = floordiv16(maxx - 8);
        int j0 = floordiv16(miny - 8 + 15), j1 = floordiv16(maxy - 8);
        if (i0 < 0) i0 = 0;
        if (j0 < 0) j0 = 0;
        if (i1 > ow - 1) i1 = ow - 1;
        if (j1 > oh - 1) j1 = oh - 1;
        const int fa[3] = {f[tr.a0], f[tr.a1], f[tr.a2]};
        const float iw0 = vt[fa[0]].iw, iw1 = vt[fa[1]].iw, iw2 = vt[fa[2]].iw;
        const float fE = (float)tr.area2;
        for (int j = j0; j <= j1; ++j) {
          for (int i = i0; i <= i1; ++i) {
            int32_t w0, w1, w2;
            tri_weights(&tr, 16 * i + 8, 16 * j + 8, &w0, &w1, &w2);
            if (((w0 + tr.b0) | (w1 + tr.b1) | (w2 + tr.b2)) < 0) continue;
            float f0 = (float)w0, f1 = (float)w1, f2 = (float)w2;
            float S = fmaf(f2, iw2, fmaf(f1, iw1, f0 * iw0));
            float z = fE / S;
            float zc = fminf(z, FPO_ZMAX);
            uint32_t zq = (uint32_t)rintf(zc * FPO_ZSCALE);
            uint64_t key = ((uint64_t)zq << 32) | (uint32_t)t;
            size_t p = (size_t)j * ow + i;
            if (key < zb[p]) zb[p] = key;
          }
        }
      }
      /* ---- resolve pass: shade the winner of every pixel */
      const float t0 = P[3], t1 = P[7], t2 = P[11];
      for (int j = 0; j < oh; ++j) {
        for (int i = 0; i < ow; ++i) {
          size_t p = (size_t)j * ow + i;
          uint64_t key = zb[p];
          float col[3] = {0, 0, 0}, pt[3] = {0, 0, 0}, nm[3] = {0, 0, 0};
          int covered = (key != FPO_EMPTY);
          int32_t tid = -1;
          if (covered) {
            tid = (int32_t)(uint32_t)(key & 0xFFFFFFFFu);
            const int* f = faces + (size_t)tid * 3;
            /* nvdiffrast's per-pixel pass: perspective-correct barycentrics of the winner from
             * its UNSNAPPED vertices, p_k = w_k * (ndc_k - pixel centre) (here in crop-pixel units, which leaves the
             * ratios unchanged), a0 = p1 x p2, ..., u = a0/(a0+a1+a2), v = a1/(...), both clamped to [0,1];
             * dr.interpolate uses (u, v, 1-u-v) for the face's vertices in face order. */
            const int slot[3] = {0, 1, 2};
            const int fa[3] = {f[0], f[1], f[2]};
            const fpo_vtx *q0 = &vt[fa[0]], *q1 = &vt[fa[1]], *q2 = &vt[fa[2]];
            const float fxp = (float)i + 0.5f, fyp = (float)j + 0.5f;
            const float p0x = (q0->X - fxp) * q0->zc, p0y = (q0->Y - fyp) * q0->zc;
            const float p1x = (q1->X - fxp) * q1->zc, p1y = (q1->Y - fyp) * q1->zc;
            const float p2x = (q2->X - fxp) * q2->zc, p2y = (q2->Y - fyp) * q2->zc;
            const float m0a = p1x * p2y, m0b = p1y * p2x, m1a = p2x * p0y, m1b = p2y * p0x, m2a = p0x * p1y, m2b = p0y * p1x;
            const float a0 = m0a - m0b, a1 = m1a - m1b, a2 = m2a - m2b;
            const float iwb = 1.0f / ((a0 + a1) + a2);
            const float b0 = clamp01(a0 * iwb), b1 = clamp01(a1 * iwb);
            const float b2 = (1.0f - b0) - b1;
            pt[0] = fmaf(b2, q2->xc, fmaf(b1, q1->xc, b0 * q0->xc));
            pt[1] = fmaf(b2, q2->yc, fmaf(b1, q1->yc, b0 * q0->yc));
            pt[2] = fmaf(b2, q2->zc, fmaf(b1, q1->zc, b0 * q0->zc));
            /* base colour */
            float base[3];
            if (tex) {
              const int* fu = (uv_idx ? uv_idx : faces) + (size_t)tid * 3;
              const float *ua = uv + (size_t)fu[slot[0]] * 2, *ub = uv + (size_t)fu[slot[1]] * 2, *uc = uv + (size_t)fu[slot[2]] * 2;
              float tu = fmaf(b2, uc[0], fmaf(b1, ub[0], b0 * ua[0]));
              float tv = fmaf(b2, uc[1], fmaf(b1, ub[1], b0 * ua[1]));
              tu = tu - floorf(tu);
              tv = tv - floorf(tv);
              tex_fetch(tex, Ht, Wt, tu, tv, base);
            } else {
              const float *ca = vcol + (size_t)fa[0] * 3, *cb = vcol + (size_t)fa[1] * 3, *cc = vcol + (size_t)fa[2] * 3;
              for (int c = 0; c < 3; ++c) base[c] = fmaf(b2, cc[c], fmaf(b1, cb[c], b0 * ca[c]));
            }
            /* normals + Lambert term per vertex, interpolated */
            float nsum[3] = {0, 0, 0};
            float dsum = 0.f;
            const float bb[3] = {b0, b1, b2};
            float nk[3][3], dk[3];
            for (int k = 0; k < 3; ++k) {
              const float* vn = nrm + (size_t)fa[k] * 3;
              nk[k][0] = fmaf(P[2], vn[2], fmaf(P[1], vn[1], P[0] * vn[0]));
              nk[k][1] = fmaf(P[6], vn[2], fmaf(P[5], vn[1], P[4] * vn[0]));
              nk[k][2] = fmaf(P[10], vn[2], fmaf(P[9], vn[1], P[8] * vn[0]));
              float len = sqrtf(fmaf(nk[k][2], nk[k][2], fmaf(nk[k][1], nk[k][1], nk[k][0] * nk[k][0])));
              dk[k] = clamp01((-nk[k][2]) / fmaxf(len, 1e-12f));
            }
            for (int c = 0; c < 3; ++c) nsum[c] = fmaf(bb[2], nk[2][c], fmaf(bb[1], nk[1][c], bb[0] * nk[0][c]));
            dsum = fmaf(bb[2], dk[2], fmaf(bb[1], dk[1], bb[0] * dk[0]));
            float nl = sqrtf(fmaf(nsum[2], nsum[2], fmaf(nsum[1], nsum[1], nsum[0] * nsum[0])));
            float inl = fmaxf(nl, 1e-12f);
            for (int c = 0; c < 3; ++c) nm[c] = nsum[c] / inl;
            for (int c = 0; c < 3; ++c) {
              float amb = base[c] * w_ambient;
              float dif = (dsum * base[c]) * w_diffuse;
              col[c] = clamp01(amb + dif);
            }
          }
          size_t o = (size_t)n * npx + p;
          if (zbuf) zbuf[o] = covered ? (uint32_t)(key >> 32) : 0xFFFFFFFFu;
          if (tri_id) tri_id[o] = tid;
          if (depth) depth[o] = pt[2];
          if (color) { color[o * 3] = col[0]; color[o * 3 + 1] = col[1]; color[o * 3 + 2] = col[2]; }
          if (xyz) { xyz[o * 3] = pt[0]; xyz[o * 3 + 1] = pt[1]; xyz[o * 3 + 2] = pt[2]; }
          if (normal) { normal[o * 3] = nm[0]; normal[o * 3 + 1] = nm[1]; normal[o * 3 + 2] = nm[2]; }
          if (A) {
            float* a = A + (size_t)n * 6 * npx + p;
            /* torch's GPU `tensor / 255.0` multiplies by the f32 reciprocal (BinaryDivTrueKernel scalar fast path) */
            for (int c = 0; c < 3; ++c) a[(size_t)c * npx] = (col[c] * 255.0f) * inv255;
            int invalid = pt[2] < xyz_thr;
            float d[3] = {pt[0] - t0, pt[1] - t1, pt[2] - t2};
            for (int c = 0; c < 3; ++c) {
              float val = d[c];
              if (normalize) {
                val = val * inv_r;
                if (invalid || fabsf(val) >= 2.0f) val = 0.f;
              }
              a[(size_t)(3 + c) * npx] = val;
            }
          }
        }
      }
    }
    free(vt);
    free(zb);
  }
}

/* ------------------------------------------------------- a9 + a10 + a11 (B side) */
static inline int nn_index(float x) { return (int)nearbyintf(x); }

void fpo_warp_crops(const float* rgb /*H,W,3 0..255*/, const float* xyz_map /*H,W,3 or NULL*/,
                    const float* depthf /*H,W or NULL*/, const float* tf_to_crops /*N,9*/,
                    const float* K9, const float* poses /*N,16*/, float mesh_diameter, int flags,
                    int mode, int H, int W, int N, int oh, int ow, float* B /*N,6,oh,ow*/) {
  const float fx = K9[0], cx = K9[2], fy = K9[4], cy = K9[5];
  const size_t npx = (size_t)oh * ow;
  const int normalize = flags & FPO_FLAG_NORMALIZE_XYZ;
  const float inv_r = 1.0f / (mesh_diameter * 0.5f);
  const float cW = (float)W / (float)(W - 1), cH = (float)H / (float)(H - 1);
  const float cSw = (float)ow / (float)(ow - 1), cSh = (float)oh / (float)(oh - 1);
  const float thr = (mode == FPO_MODE_SCORE) ? 0.1f : 0.001f;
  const float inv255 = 1.0f / 255.0f;
#pragma omp parallel for schedule(dynamic, 1)
  for (int n = 0; n < N; ++n) {
    const float* tf = tf_to_crops + (size_t)n * 9;
    const float sx = tf[0], tx = tf[2], sy = tf[4], ty = tf[5];
    const float i00 = 1.0f / sx, i11 = 1.0f / sy;
    const float i02 = (-tx) / sx, i12 = (-ty) / sy;
    /* the crop window's left/top edge is an integer (Utils.py:586-589 round()); when it is, frame -> crop
     * coordinates are evaluated as s*(q - left): exactly 0 on the window edge, so the half-integer tie of the
     * scorer's hop 2 (sample coordinate -0.5 for q = left) resolves the same way for every window. */
    const float lfx = rintf(i02), lfy = rintf(i12);
    const int alx = fabsf(i02 - lfx) <= 1e-3f, aly = fabsf(i12 - lfy) <= 1e-3f;
    const float* P = poses + (size_t)n * 16;
    const float t0 = P[3], t1 = P[7], t2 = P[11];
    for (int j = 0; j < oh; ++j) {
      for (int i = 0; i < ow; ++i) {
        float xs = fmaf((float)i, i00, i02), ys = fmaf((float)j, i11, i12);
        float ix = fmaf(xs, cW, -0.5f), iy = fmaf(ys, cH, -0.5f);
        float* b = B + (size_t)n * 6 * npx + (size_t)j * ow + i;
        /* ---- rgb: bilinear, zeros padding (torch grid_sample tap order nw,ne,sw,se) */
        float fx0 = floorf(ix), fy0 = floorf(iy);
        int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        float wnw = ((float)x1 - ix) * ((float)y1 - iy);
        float wne = (ix - (float)x0) * ((float)y1 - iy);
        float wsw = ((float)x1 - ix) * (iy - (float)y0);
        float wse = (ix - (float)x0) * (iy - (float)y0);
        for (int c = 0; c < 3; ++c) {
          float acc = 0.f;
          if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) acc += rgb[((size_t)y0 * W + x0) * 3 + c] * wnw;
          if (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H) acc += rgb[((size_t)y0 * W + x1) * 3 + c] * wne;
          if (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H) acc += rgb[((size_t)y1 * W + x0) * 3 + c] * wsw;
          if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) acc += rgb[((size_t)y1 * W + x1) * 3 + c] * wse;
          b[(size_t)c * npx] = acc * inv255;
        }
        /* ---- xyz: nearest */
        float pt[3] = {0, 0, 0};
        int qx = nn_index(ix), qy = nn_index(iy);
        int q_in = (qx >= 0 && qx < W && qy >= 0 && qy < H);
        if (mode == FPO_MODE_REFINE) {
          if (q_in) { const float* s = xyz_map + ((size_t)qy * W + qx) * 3; pt[0] = s[0]; pt[1] = s[1]; pt[2] = s[2]; }
        } else if (q_in) {
          /* scorer: depth crop -> full frame -> back-projection -> crop (h5_dataset.py:159-161) */
          float ccx = alx ? sx * ((float)qx - lfx) : fmaf(sx, (float)qx, tx);
          float ccy = aly ? sy * ((float)qy - lfy) : fmaf(sy, (float)qy, ty);
          int px = nn_index(fmaf(ccx, cSw, -0.5f)), py = nn_index(fmaf(ccy, cSh, -0.5f));
          float z = 0.f;
          if (px >= 0 && px < ow && py >= 0 && py < oh) {
            float xs2 = fmaf((float)px, i00, i02), ys2 = fmaf((float)py, i11, i12);
            int rx = nn_index(fmaf(xs2, cW, -0.5f)), ry = nn_index(fmaf(ys2, cH, -0.5f));
            if (rx >= 0 && rx < W && ry >= 0 && ry < H) z = depthf[(size_t)ry * W + rx];
          }
          if (!(z < 0.001f)) {
            pt[0] = (((float)qx - cx) * z) / fx;
            pt[1] = (((float)qy - cy) * z) / fy;
            pt[2] = z;
          }
        }
        int invalid = pt[2] < thr;
        float d[3] = {pt[0] - t0, pt[1] - t1, pt[2] - t2};
        for (int c = 0; c < 3; ++c) {
          float val = d[c];
          if (normalize) {
            val = val * inv_r;
            if (invalid || fabsf(val) >= 2.0f) val = 0.f;
          }
          b[(size_t)(3 + c) * npx] = val;
        }
      }
    }
  }
}

/* ----------------------------------------------------------------- a13 */
#define FPO_ROT_AXIS_ANGLE 0
#define FPO_ROT_6D 1

void fpo_pose_update(const float* trans /*N,3*/, const float* rot /*N,3|6*/, const float* poses_in,
                     int rot_rep, int normalize_xyz, const float* trans_normalizer /*3*/,
                     float rot_normalizer, float mesh_diameter, int N, float* poses_out) {
  for (int n = 0; n < N; ++n) {
    float dt[3];
    for (int c = 0; c < 3; ++c) {
      float v = trans[n * 3 + c];
      if (!normalize_xyz) v = tanhf(v) * trans_normalizer[c];
      else v = v * (mesh_diameter / 2.0f);
      dt[c] = v;
    }
    float R[9]; /* rotation BEFORE the reference's transpose */
    if (rot_rep == FPO_ROT_AXIS_ANGLE) {
      float w[3];
      for (int c = 0; c < 3; ++c) w[c] = tanhf(rot[n * 3 + c]) * rot_normalizer;
      float n2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
      float th = sqrtf(fmaxf(n2, 1e-4f));
      float ith = 1.0f / th;
      float f1 = ith * sinf(th);
      float f2 = (ith * ith) * (1.0f - cosf(th));
      float Kx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
      float K2[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          K2[r * 3 + c] = (Kx[r * 3] * Kx[c] + Kx[r * 3 + 1] * Kx[3 + c]) + Kx[r * 3 + 2] * Kx[6 + c];
      for (int k = 0; k < 9; ++k) R[k] = (f1 * Kx[k] + f2 * K2[k]) + ((k % 4 == 0) ? 1.0f : 0.0f);
    } else {
      const float* d = rot + (size_t)n * 6;
      float a1[3] = {d[0], d[1], d[2]}, a2[3] = {d[3], d[4], d[5]};
      float l1 = fmaxf(sqrtf((a1[0] * a1[0] + a1[1] * a1[1]) + a1[2] * a1[2]), 1e-12f);
      float b1[3] = {a1[0] / l1, a1[1] / l1, a1[2] / l1};
      float dp = (b1[0] * a2[0] + b1[1] * a2[1]) + b1[2] * a2[2];
      float u2[3] = {a2[0] - dp * b1[0], a2[1] - dp * b1[1], a2[2] - dp * b1[2]};
      float l2 = fmaxf(sqrtf((u2[0] * u2[0] + u2[1] * u2[1]) + u2[2] * u2[2]), 1e-12f);
      float b2[3] = {u2[0] / l2, u2[1] / l2, u2[2] / l2};
      float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
      for (int c = 0; c < 3; ++c) { R[c] = b1[c]; R[3 + c] = b2[c]; R[6 + c] = b3[c]; }
    }
    /* reference uses the transpose: dR[r][c] = R[c][r]; R' = dR @ R_A ; t' = t + dt */
    const float* A = poses_in + (size_t)n * 16;
    float* O = poses_out + (size_t)n * 16;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) {
        O[r * 4 + c] = (R[0 * 3 + r] * A[0 * 4 + c] + R[1 * 3 + r] * A[1 * 4 + c]) + R[2 * 3 + r] * A[2 * 4 + c];
      }
      O[r * 4 + 3] = A[r * 4 + 3] + dt[r];
    }
    O[12] = 0.f; O[13] = 0.f; O[14] = 0.f; O[15] = 1.f;
  }
}

/* ------------------------------------------------------------------ a4 */
/* returns the number kept; keep_idx (>= N ints) receives the indices of kept poses */
int fpo_cluster_poses(float angle_diff_deg, float dist_diff, const float* poses /*N,16*/, int N,
                      const float* sym /*S,16*/, int S, int* keep_idx) {
  const float rad_thres = (float)(angle_diff_deg / 180.0 * M_PI);
  int nk = 0;
  if (N <= 0) return 0;
  keep_idx[nk++] = 0;
  for (int i = 1; i < N; ++i) {
    const float* cur = poses + (size_t)i * 16;
    int isnew = 1;
    for (int k = 0; k < nk && isnew; ++k) {
      const float* cl = poses + (size_t)keep_idx[k] * 16;
      float dx = cl[3] - cur[3], dy = cl[7] - cur[7], dz = cl[11] - cur[11];
      float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
      if (dist >= dist_diff) continue;
      for (int s = 0; s < S; ++s) {
        const float* tf = sym + (size_t)s * 16;
        float R1[9];
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c)
            R1[r * 3 + c] = ((cur[r * 4] * tf[c] + cur[r * 4 + 1] * tf[4 + c]) + cur[r * 4 + 2] * tf[8 + c]) + cur[r * 4 + 3] * tf[12 + c];
        /* trace(R1 * R2^T) */
        float tr = 0.f;
        for (int r = 0; r < 3; ++r) tr += (R1[r * 3] * cl[r * 4] + R1[r * 3 + 1] * cl[r * 4 + 1]) + R1[r * 3 + 2] * cl[r * 4 + 2];
        float cs = (tr - 1.0f) / 2.0f;
        cs = fmaxf(fminf(cs, 1.0f), -1.0f);
        if (acosf(cs) < rad_thres) { isnew = 0; break; }
      }
    }
    if (isnew) keep_idx[nk++] = i;
  }
  return nk;
}
