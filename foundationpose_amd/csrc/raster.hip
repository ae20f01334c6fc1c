// Fused pose-hypothesis rasteriser for gfx950 (replaces nvdiffrast_render, Utils.py:133-219, plus the
// A-side of make_crop_data_batch / transform_batch / concat -- see include/fp_amd.h).
//
// fp_render_crops = three launches (k_vertex, k_bin, k_raster -- see the pipeline comment below): per-vertex work once
// per hypothesis, triangles binned to 16-row strips, strip z-buffer in LDS merged with a 64-bit ds_min on the key
// (round(z_cam * 2^20) << 32 | tri_id) -- deterministic winner, independent of arrival order -- and a pixel-parallel
// resolve that writes the network tensor A[n, 0:6] directly (fp16 or fp32), so no intermediate image reaches HBM.
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
// phase-skip bits for scripts/raster_phases.py exist only in libfp_amd_profile.so (-DFP_PROFILE_BUILD); the product
// kernel has no such branches and fp_render_crops rejects flag bits it does not define
#ifdef FP_PROFILE_BUILD
#define FP_PROF_FLAG(flags, bit) (((flags) & (bit)) != 0)
#define FP_RENDER_FLAG_MASK (FP_FLAG_NORMALIZE_XYZ | FP_FLAG_OUT_F16 | 0xF0000)
#else
#define FP_PROF_FLAG(flags, bit) false
#define FP_RENDER_FLAG_MASK (FP_FLAG_NORMALIZE_XYZ | FP_FLAG_OUT_F16)
#endif

/* tie rule for a pixel centre exactly on an edge (vertices oriented to positive area in THIS, y-down, crop space): the
 * top-left rule of a rasteriser working in nvdiffrast's y-up window space -- the reference flips the rows AFTER
 * rasterising (Utils.py:216-218) -- seen from here: an edge owns its points if it runs downwards in window space
 * (dy > 0 in both spaces, because the orientation fix reverses the edge when the rows are flipped), and a horizontal
 * edge if it runs towards -x there = towards +x here. */
__device__ __forceinline__ bool edge_owner(int dx, int dy) { return (dy > 0) || (dy == 0 && dx > 0); }

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
// i mod n into [0, n).  The texel indices of tex_fetch are floor(u * n - 0.5) and that + 1 with u in [0, 1], i.e. -1 .. n: one
// conditional add / subtract covers them; the emulated integer modulo (~35 VALU instructions, four per shaded pixel: a third
// of the resolve phase's instructions, profiles/r03_stage_counters_sq.json) is only taken for anything further out.
__device__ __forceinline__ int wrapi(int i, int n) {
  if ((unsigned)i < (unsigned)n) return i;
  if (i >= -n && i < 0) return i + n;
  if (i >= n && i < 2 * n) return i - n;
  int r = i % n;
  return r < 0 ? r + n : r;
}
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

// ---------------------------------------------------------------------------------------------------------------
// Pipeline of one fp_render_crops call (three launches on the caller's stream, scratch in the caller's workspace):
//   k_vertex : grid (V/256, N)  one lane per (hypothesis, vertex): camera-space position, unsnapped and snapped crop
//              position, 1/z, Lambert term of the vertex normal  -> VtxRec (8 B, raster) + VtxAttr (24 B, resolve)
//   k_bin    : grid (T/256, N)  one lane per (hypothesis, triangle): strips of FP_STRIP_ROWS rows it can touch
//              -> per-(hypothesis, strip) triangle lists (wave-ballot compaction, one atomicAdd per wave and strip)
//   k_raster : grid (strips, N) one workgroup per (hypothesis, strip): LDS strip z-buffer
//              phase 1  lanes walk the strip's triangle list: integer edge functions, 64-bit ds_min of the depth key
//              phase 2  lanes = pixels (coalesced): barycentrics, interpolation, texture, shading, network tensor A
// Per-vertex work is done once per hypothesis (not once per strip and not three times per pixel), a strip only ever
// looks at the triangles binned to it, and 16-row strips (20 KiB of LDS) keep 6-7 workgroups resident per CU.
#define FP_STRIP_ROWS 16
#define FP_MAX_STRIPS 64          // oh <= 1024
#define FP_BIG_CELLS 24           // clipped bounding boxes above this many pixels are rasterised cooperatively
#define FP_BIG_MAX 96             // queue capacity (entries of 56 B)

struct VtxAttr {            // per (hypothesis, vertex), resolve-side
  float xc, yc, zc;        // camera-space position
  float X, Y;              // unsnapped crop-pixel position
  float dk;                // clip(normalize(R n) . (0,0,-1), 0, 1)  (Utils.py:203-206)
};

struct BigTri {             // a finished triangle setup parked in LDS (oriented: area2 > 0)
  int x0, y0, x1, y1, x2, y2;
  float iw0, iw1, iw2;
  int t, i0, i1, j0, j1;
};

struct RenderWs {
  VtxRec* vr;              // [N][V]
  VtxAttr* va;             // [N][V]
  int* counts;             // [N][strips]
  unsigned short* lists16; // [N][strips][T]   (T <= 65535)
  int* lists32;            // same with 32-bit ids for larger meshes
};

__global__ __launch_bounds__(256) void k_vertex(fp_mesh m, const float* __restrict__ poses,
                                                const float* __restrict__ bbox2d, fp_k9 K, int H, int W, int oh, int ow,
                                                int nstrips, RenderWs ws) {
  const int n = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < nstrips) ws.counts[n * nstrips + threadIdx.x] = 0;
  if (v >= m.V) return;
  const HypConst h = load_hyp(poses, bbox2d, K, n, H, W, oh, ow);
  const float vx = m.pos[v * 3], vy = m.pos[v * 3 + 1], vz = m.pos[v * 3 + 2];
  VtxAttr a;
  cam_point(h, vx, vy, vz, a.xc, a.yc, a.zc);
  crop_xy(h, a.xc, a.yc, a.zc, a.X, a.Y);
  const float* vn = m.nrm + (size_t)v * 3;
  const float n0 = fmaf(h.P[2], vn[2], fmaf(h.P[1], vn[1], h.P[0] * vn[0]));
  const float n1 = fmaf(h.P[6], vn[2], fmaf(h.P[5], vn[1], h.P[4] * vn[0]));
  const float n2 = fmaf(h.P[10], vn[2], fmaf(h.P[9], vn[1], h.P[8] * vn[0]));
  const float len = sqrtf(fmaf(n2, n2, fmaf(n1, n1, n0 * n0)));
  a.dk = clamp01((-n2) / fmaxf(len, 1e-12f));
  const size_t o = (size_t)n * m.V + v;
  ws.va[o] = a;
  ws.vr[o] = project_vertex(h, vx, vy, vz);
}

__global__ __launch_bounds__(256) void k_bin(fp_mesh m, int oh, int ow, int nstrips, RenderWs ws) {
  const int n = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  int s0 = 1, s1 = 0;   // empty range
  if (t < m.T) {
    const VtxRec* vr = ws.vr + (size_t)n * m.V;
    const VtxRec r0 = vr[m.faces[t * 3]], r1 = vr[m.faces[t * 3 + 1]], r2 = vr[m.faces[t * 3 + 2]];
    TriSetup tr;
    if (tri_setup(r0, r1, r2, tr)) {
      const int miny = min(tr.y0, min(tr.y1, tr.y2)), maxy = max(tr.y0, max(tr.y1, tr.y2));
      const int minx = min(tr.x0, min(tr.x1, tr.x2)), maxx = max(tr.x0, max(tr.x1, tr.x2));
      const int i0 = max((minx - 8 + 15) >> 4, 0), i1 = min((maxx - 8) >> 4, ow - 1);
      const int j0 = max((miny - 8 + 15) >> 4, 0), j1 = min((maxy - 8) >> 4, oh - 1);
      if (i0 <= i1 && j0 <= j1) { s0 = j0 / FP_STRIP_ROWS; s1 = j1 / FP_STRIP_ROWS; }
    }
  }
  const int lane = threadIdx.x & 63;
  // strips any lane of this wave touches
  int lo = s0 <= s1 ? s0 : FP_MAX_STRIPS, hi = s0 <= s1 ? s1 : -1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
  // lane s keeps the ballot of strip s, so the (up to 64) atomicAdds of a wave are all in flight together
  unsigned long long mymask = 0ull;
  for (int s = lo; s <= hi; ++s) {
    const unsigned long long mask = __ballot(s >= s0 && s <= s1);
    if (lane == s) mymask = mask;
  }
  int mybase = 0;
  if (mymask != 0ull) mybase = atomicAdd(&ws.counts[n * nstrips + lane], __popcll(mymask));
  for (int s = lo; s <= hi; ++s) {
    const unsigned long long mask = __shfl(mymask, s, 64);
    const int base = __shfl(mybase, s, 64);
    if (s >= s0 && s <= s1) {
      const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
      const size_t li = ((size_t)n * nstrips + s) * m.T + pos;
      if (ws.lists16) ws.lists16[li] = (unsigned short)t;
      else ws.lists32[li] = t;
    }
  }
}

__global__ __launch_bounds__(FP_RASTER_THREADS) void k_raster(
    fp_mesh m, const float* __restrict__ poses, const float* __restrict__ bbox2d, fp_k9 K, int H, int W, int oh,
    int ow, int nstrips, float w_ambient, float w_diffuse, float inv_r, float xyz_thr, int flags, RenderOut out,
    RenderWs ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = blockIdx.y, strip = blockIdx.x;
  const int row0 = strip * FP_STRIP_ROWS;
  const int rows = min(FP_STRIP_ROWS, oh - row0);
  const int npix = rows * ow;
  unsigned long long* zb = reinterpret_cast<unsigned long long*>(smem);
  int* nbig = reinterpret_cast<int*>(smem + (size_t)FP_STRIP_ROWS * ow * sizeof(unsigned long long));
  BigTri* big = reinterpret_cast<BigTri*>(nbig + 4);
  const int tid = threadIdx.x;
  if (tid == 0) *nbig = 0;
  const VtxRec* vr = ws.vr + (size_t)n * m.V;
  const VtxAttr* va = ws.va + (size_t)n * m.V;

  for (int p = tid; p < npix; p += FP_RASTER_THREADS) zb[p] = FP_KEY_EMPTY;
  __syncthreads();

  // ---- phase 1: the strip's triangle list -> LDS z-buffer.  One lane per triangle; a triangle whose clipped bounding
  // box exceeds FP_BIG_CELLS pixels (the fan triangles of a cap span 50 x 16 of them) would stall its whole wave in
  // the per-lane pixel loop, so it is queued and rasterised afterwards by all lanes together (pixel-parallel).
  const int cnt = FP_PROF_FLAG(flags, 0x20000) ? 0 : ws.counts[n * nstrips + strip];   // 0x20000: profiling aid, skip phase 1
  const size_t lbase = ((size_t)n * nstrips + strip) * m.T;
  // edge functions are affine in the pixel index: w_k(i+1, j) = w_k(i, j) + 16*dwx_k, so a row costs three integer adds
  // per cell after one evaluation at its first cell (`first`/`step` stride the cells of a row-major walk over lanes)
  auto raster_cells = [&](const TriSetup& tr, float iw0, float iw1, float iw2, int t, int i0, int i1, int j0, int j1,
                          int first, int step) {
    if (FP_PROF_FLAG(flags, 0x40000)) return;   // profiling aid: setup only
    const float fE = (float)tr.area2;
    const int dx0 = -16 * (tr.y2 - tr.y1), dx1 = -16 * (tr.y0 - tr.y2), dx2 = -16 * (tr.y1 - tr.y0);
    auto cell = [&](int i, int j, int w0, int w1, int w2) {
      if (((w0 + tr.b0) | (w1 + tr.b1) | (w2 + tr.b2)) < 0) return;
      const float g0 = (float)w0, g1 = (float)w1, g2 = (float)w2;
      const float S = fmaf(g2, iw2, fmaf(g1, iw1, g0 * iw0));
      const float z = fE / S;
      const float zc = fminf(z, FP_ZMAXF);
      const uint32_t zq = (uint32_t)rintf(zc * FP_ZSCALEF);
      const unsigned long long key = ((unsigned long long)zq << 32) | (uint32_t)t;
      if (FP_PROF_FLAG(flags, 0x80000)) { asm volatile("" ::"v"((uint32_t)key), "v"((uint32_t)(key >> 32))); }   // profiling aid: no LDS atomic
      else atomicMin(&zb[(j - row0) * ow + i], key);
    };
    if (step == 1) {            // one lane owns the whole box: incremental walk
      for (int j = j0; j <= j1; ++j) {
        int w0, w1, w2;
        tri_weights(tr, 16 * i0 + 8, 16 * j + 8, w0, w1, w2);
        for (int i = i0; i <= i1; ++i) {
          cell(i, j, w0, w1, w2);
          w0 += dx0; w1 += dx1; w2 += dx2;
        }
      }
    } else {                    // the lanes of a wave stride the cells of the box
      const int bw = i1 - i0 + 1;
      const int cells = bw * (j1 - j0 + 1);
      // cell (jj, ii) of c = first + k * step without a division per cell
      const int sj = step / bw, si = step - sj * bw;
      int jj = first / bw, ii = first - jj * bw;
      for (int c = first; c < cells; c += step, ii += si, jj += sj) {
        if (ii >= bw) { ii -= bw; ++jj; }
        const int i = i0 + ii, j = j0 + jj;
        int w0, w1, w2;
        tri_weights(tr, 16 * i + 8, 16 * j + 8, w0, w1, w2);
        cell(i, j, w0, w1, w2);
      }
    }
  };
  auto setup = [&](int t, TriSetup& tr, float& iw0, float& iw1, float& iw2, int& i0, int& i1, int& j0, int& j1) -> bool {
    const VtxRec r0 = vr[m.faces[t * 3]], r1 = vr[m.faces[t * 3 + 1]], r2 = vr[m.faces[t * 3 + 2]];
    if (!tri_setup(r0, r1, r2, tr)) return false;
    const int miny = min(tr.y0, min(tr.y1, tr.y2)), maxy = max(tr.y0, max(tr.y1, tr.y2));
    const int minx = min(tr.x0, min(tr.x1, tr.x2)), maxx = max(tr.x0, max(tr.x1, tr.x2));
    i0 = max((minx - 8 + 15) >> 4, 0); i1 = min((maxx - 8) >> 4, ow - 1);
    j0 = max((miny - 8 + 15) >> 4, row0); j1 = min((maxy - 8) >> 4, row0 + rows - 1);
    iw0 = r0.iw;
    iw1 = (tr.s1 == 1) ? r1.iw : r2.iw;
    iw2 = (tr.s1 == 1) ? r2.iw : r1.iw;
    return i0 <= i1 && j0 <= j1;
  };
  for (int e = tid; e < cnt; e += FP_RASTER_THREADS) {
    const int t = ws.lists16 ? (int)ws.lists16[lbase + e] : ws.lists32[lbase + e];
    TriSetup tr;
    float iw0, iw1, iw2;
    int i0, i1, j0, j1;
    if (!setup(t, tr, iw0, iw1, iw2, i0, i1, j0, j1)) continue;
    if ((i1 - i0 + 1) * (j1 - j0 + 1) > FP_BIG_CELLS) {
      const int slot = atomicAdd(nbig, 1);
      if (slot < FP_BIG_MAX) {   // park the finished setup in LDS; queue full: fall through to the per-lane loop
        BigTri& b = big[slot];
        b.x0 = tr.x0; b.y0 = tr.y0; b.x1 = tr.x1; b.y1 = tr.y1; b.x2 = tr.x2; b.y2 = tr.y2;
        b.iw0 = iw0; b.iw1 = iw1; b.iw2 = iw2; b.t = t; b.i0 = i0; b.i1 = i1; b.j0 = j0; b.j1 = j1;
        continue;
      }
    }
    raster_cells(tr, iw0, iw1, iw2, t, i0, i1, j0, j1, 0, 1);
  }
  __syncthreads();
  {
    // one wave per parked triangle, its 64 lanes over the clipped bounding box
    const int nb = min(*nbig, FP_BIG_MAX);
    const int lane = tid & 63;
    for (int q = tid >> 6; q < nb; q += FP_RASTER_THREADS / 64) {
      const BigTri b = big[q];
      TriSetup tr;
      tr.x0 = b.x0; tr.y0 = b.y0; tr.x1 = b.x1; tr.y1 = b.y1; tr.x2 = b.x2; tr.y2 = b.y2;
      tr.area2 = (b.x1 - b.x0) * (b.y2 - b.y0) - (b.y1 - b.y0) * (b.x2 - b.x0);   // already oriented: > 0
      tr.b0 = edge_owner(tr.x2 - tr.x1, tr.y2 - tr.y1) ? 0 : -1;
      tr.b1 = edge_owner(tr.x0 - tr.x2, tr.y0 - tr.y2) ? 0 : -1;
      tr.b2 = edge_owner(tr.x1 - tr.x0, tr.y1 - tr.y0) ? 0 : -1;
      tr.s1 = 1; tr.s2 = 2;
      raster_cells(tr, b.iw0, b.iw1, b.iw2, b.t, b.i0, b.i1, b.j0, b.j1, lane, 64);
    }
  }
  __syncthreads();

  // ---- phase 2: resolve + shade + write
  const HypConst h = load_hyp(poses, bbox2d, K, n, H, W, oh, ow);
  const size_t plane = (size_t)oh * ow;
  const float t0 = h.P[3], t1 = h.P[7], t2 = h.P[11];
  const bool normalize = (flags & FP_FLAG_NORMALIZE_XYZ) != 0;
  const bool out_f16 = (flags & FP_FLAG_OUT_F16) != 0;
  // pixel (jl, i) of p = tid + k * THREADS without a division per iteration
  const int step_j = FP_RASTER_THREADS / ow, step_i = FP_RASTER_THREADS - step_j * ow;
  int jl = tid / ow, i = tid - jl * ow;
  for (int p = tid; p < npix; p += FP_RASTER_THREADS, i += step_i, jl += step_j) {
    if (i >= ow) { i -= ow; ++jl; }
    const int j = row0 + jl;
    const unsigned long long key = zb[p];
    const bool covered = key != FP_KEY_EMPTY;
    float col[3] = {0.f, 0.f, 0.f}, pt[3] = {0.f, 0.f, 0.f}, nm[3] = {0.f, 0.f, 0.f};
    int tid_out = -1;
    if (covered && !FP_PROF_FLAG(flags, 0x10000)) {   // 0x10000: profiling aid, skip the shading gathers
      const int t = (int)(uint32_t)(key & 0xFFFFFFFFull);
      tid_out = t;
      const int fa0 = m.faces[t * 3], fa1 = m.faces[t * 3 + 1], fa2 = m.faces[t * 3 + 2];
      const VtxAttr A0 = va[fa0], A1 = va[fa1], A2 = va[fa2];
      // nvdiffrast's per-pixel pass: perspective-correct barycentrics of the winner from its
      // UNSNAPPED vertices in face order, p_k = z_k * (X_k - pixel centre), a0 = p1 x p2, ..., clamped (u, v), 1-u-v
      const float fxp = (float)i + 0.5f, fyp = (float)j + 0.5f;
      const float p0x = (A0.X - fxp) * A0.zc, p0y = (A0.Y - fyp) * A0.zc;
      const float p1x = (A1.X - fxp) * A1.zc, p1y = (A1.Y - fyp) * A1.zc;
      const float p2x = (A2.X - fxp) * A2.zc, p2y = (A2.Y - fyp) * A2.zc;
      const float m0a = p1x * p2y, m0b = p1y * p2x, m1a = p2x * p0y, m1b = p2y * p0x, m2a = p0x * p1y, m2b = p0y * p1x;
      const float a0 = m0a - m0b, a1 = m1a - m1b, a2 = m2a - m2b;
      const float iwb = 1.0f / ((a0 + a1) + a2);
      const float b0 = clamp01(a0 * iwb), b1 = clamp01(a1 * iwb);
      const float b2 = (1.0f - b0) - b1;
      pt[0] = fmaf(b2, A2.xc, fmaf(b1, A1.xc, b0 * A0.xc));
      pt[1] = fmaf(b2, A2.yc, fmaf(b1, A1.yc, b0 * A0.yc));
      pt[2] = fmaf(b2, A2.zc, fmaf(b1, A1.zc, b0 * A0.zc));
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
      const float dsum = fmaf(b2, A2.dk, fmaf(b1, A1.dk, b0 * A0.dk));
      if (out.normal) {   // only the nvdiffrast_render shim asks for normals: redo the per-vertex transform here
        const int fav[3] = {fa0, fa1, fa2};
        const float bb[3] = {b0, b1, b2};
        float nk[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float* vn = m.nrm + (size_t)fav[k] * 3;
          nk[k][0] = fmaf(h.P[2], vn[2], fmaf(h.P[1], vn[1], h.P[0] * vn[0]));
          nk[k][1] = fmaf(h.P[6], vn[2], fmaf(h.P[5], vn[1], h.P[4] * vn[0]));
          nk[k][2] = fmaf(h.P[10], vn[2], fmaf(h.P[9], vn[1], h.P[8] * vn[0]));
        }
        float nsum[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) nsum[c] = fmaf(bb[2], nk[2][c], fmaf(bb[1], nk[1][c], bb[0] * nk[0][c]));
        const float nl = sqrtf(fmaf(nsum[2], nsum[2], fmaf(nsum[1], nsum[1], nsum[0] * nsum[0])));
        const float inl = fmaxf(nl, 1e-12f);
#pragma unroll
        for (int c = 0; c < 3; ++c) nm[c] = nsum[c] / inl;
      }
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
static inline size_t ws_align(size_t x) { return (x + 255) & ~(size_t)255; }

struct WsLayout { size_t vr, va, counts, lists, total; int nstrips; bool ids16; };

static WsLayout ws_layout(int N, int V, int T, int oh) {
  WsLayout L;
  L.nstrips = fp_cdiv(oh, FP_STRIP_ROWS);
  L.ids16 = T <= 65535;
  size_t o = 0;
  L.vr = o; o = ws_align(o + (size_t)N * V * sizeof(VtxRec));
  L.va = o; o = ws_align(o + (size_t)N * V * sizeof(VtxAttr));
  L.counts = o; o = ws_align(o + (size_t)N * L.nstrips * sizeof(int));
  L.lists = o; o = ws_align(o + (size_t)N * L.nstrips * T * (L.ids16 ? 2 : 4));
  L.total = o;
  return L;
}

extern "C" size_t fp_workspace_bytes(int N, int V, int T, int oh, int ow) {
  (void)ow;
  if (N <= 0 || V <= 0 || T <= 0 || oh <= 0) return 0;
  return ws_layout(N, V, T, oh).total;
}

extern "C" int fp_render_crops(const fp_mesh* mesh, const float* poses, const float* bbox2d, const float* K9, int H,
                               int W, int N, int oh, int ow, float w_ambient, float w_diffuse, float mesh_diameter,
                               float xyz_thr, int flags, void* A, float* color, float* depth, float* xyz,
                               float* normal, uint32_t* zbuf, int32_t* tri_id, void* workspace,
                               size_t workspace_bytes, void* stream) {
  FP_REQUIRE(N >= 0, "fp_render_crops: N < 0");
  FP_REQUIRE((flags & ~FP_RENDER_FLAG_MASK) == 0, "fp_render_crops: unknown flag bits 0x%x", flags & ~FP_RENDER_FLAG_MASK);
  if (N == 0) return FP_OK;
  FP_REQUIRE(mesh && poses && K9, "fp_render_crops: NULL mesh/poses/K");
  FP_REQUIRE(oh > 0 && ow > 0 && oh <= 1024 && ow <= 1024, "fp_render_crops: output size %dx%d unsupported (max 1024)", oh, ow);
  FP_REQUIRE(bbox2d || (oh == H && ow == W), "fp_render_crops: full-frame render needs oh==H and ow==W");
  FP_REQUIRE(N <= 65535, "fp_render_crops: N=%d exceeds the grid limit; chunk the batch", N);
  const WsLayout L = ws_layout(N, mesh->V, mesh->T, oh);
  if (!workspace || workspace_bytes < L.total) {
    fp_set_error("fp_render_crops: workspace too small (%zu < %zu bytes, see fp_workspace_bytes)", workspace_bytes, L.total);
    return FP_ERR_WORKSPACE;
  }
  fp_k9 K;
  for (int i = 0; i < 9; ++i) K.v[i] = K9[i];
  unsigned char* w8 = (unsigned char*)workspace;
  RenderWs ws;
  ws.vr = (VtxRec*)(w8 + L.vr); ws.va = (VtxAttr*)(w8 + L.va); ws.counts = (int*)(w8 + L.counts);
  ws.lists16 = L.ids16 ? (unsigned short*)(w8 + L.lists) : nullptr;
  ws.lists32 = L.ids16 ? nullptr : (int*)(w8 + L.lists);
  const float inv_r = 1.0f / (mesh_diameter * 0.5f);
  RenderOut out = {A, color, depth, xyz, normal, zbuf, tri_id};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_vertex, dim3(fp_cdiv(mesh->V, 256), N), dim3(256), 0, st, *mesh, poses, bbox2d, K, H, W, oh, ow,
                     L.nstrips, ws);
  FP_CHECK_LAUNCH("fp_render_crops(vertex)");
  hipLaunchKernelGGL(k_bin, dim3(fp_cdiv(mesh->T, 256), N), dim3(256), 0, st, *mesh, oh, ow, L.nstrips, ws);
  FP_CHECK_LAUNCH("fp_render_crops(bin)");
  const size_t lds = (size_t)FP_STRIP_ROWS * ow * sizeof(unsigned long long) + 16 + FP_BIG_MAX * sizeof(BigTri);
  FP_SET_MAX_LDS(k_raster, 160 * 1024);
  hipLaunchKernelGGL(k_raster, dim3(L.nstrips, N), dim3(FP_RASTER_THREADS), lds, st, *mesh, poses, bbox2d, K, H, W, oh, ow,
                     L.nstrips, w_ambient, w_diffuse, inv_r, xyz_thr, flags, out, ws);
  FP_CHECK_LAUNCH("fp_render_crops");
  return FP_OK;
}
