// fp_igemm_f16_fwd -- the network stage's workhorse on gfx950: one MFMA kernel for
//   * every 3x3 convolution of the two encoders (refine_network.py:37-50, score_network.py:36-49;
//     network_modules.py:37-50 ConvBNReLU, :73-111 ResnetBasicBlock) as an implicit GEMM over NHWC activations, with
//     the eval-mode BatchNorm folded into (weights, bias) and bias + residual add + ReLU fused in the epilogue, and
//   * every 512-wide projection of the transformer heads (QKV in_proj, out_proj, FFN linear1/linear2;
//     refine_network.py:56-70, score_network.py:52-53) as the 1-tap special case.
//
// GEMM view: D[m][n] = sum_k A[m][k] * W[n][k],  m = output pixel (b, oy, ox), n = output channel,
// k = (tap, ci) with tap = ky*3+kx.  Activations live in HBM as NHWC with a zero border of `pad` pixels
// ((B, H+2, W+2, C) for the 3x3 layers), so a tap is a constant element offset added to a per-row base address and
// no boundary test exists anywhere in the main loop.
//
// Structure (cdna_hip_programming.md section 5, "step 3" + 2-phase pipeline):
//   * workgroup tile 128 (pixels) x 128 (channels) x 64 (k), 256 threads = 4 waves in 2x2, each wave 64x64 as
//     2x2 v_mfma_f32_32x32x16_f16 tiles (fp32 accumulate, 64 accumulator registers)
//   * operands staged HBM -> LDS by global_load_lds_dwordx4 (no VGPR round trip), two LDS stages (64 KiB), the
//     loads of k-step t+1 are in flight while k-step t is multiplied; one barrier per k-step
//   * LDS rows are 128 B (64 halves); the 16-byte chunk index is XOR-swizzled with (row>>1)&7 on the SOURCE address
//     (the LDS-DMA destination is lane-linear) and on the ds_read_b128 address: conflict-free for the 32x32x16
//     fragment read (16 distinct 16-B slots per ds_read_b128 lane group)
//   * MFMA orientation D[n][m]: a lane ends up with 4 consecutive channels of one pixel per accumulator quad, which
//     go through an XOR-swizzled LDS tile so that HBM sees full 256-B row segments (16 B per lane), with bias,
//     residual and ReLU applied on the way out
//   * XCD-aware tile order: the channel tiles of one pixel tile run back to back on the same XCD (shared A rows in
//     that XCD's L2); weights (<= 4.7 MB per layer) stay resident in every L2.
#include <hip/hip_fp16.h>
#include "fp_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16_ __attribute__((ext_vector_type(16)));

#define IG_BM 128
#define IG_BN 128
#define IG_BK 64
#define IG_THREADS 256
#define IG_STAGE_BYTES (2 * IG_BM * IG_BK * 2)      // A tile + W tile of one stage = 32 KiB
#define IG_LDS_BYTES (2 * IG_STAGE_BYTES)            // 64 KiB

struct IgemmGeom {      // row m -> element offset of pixel (b, y*stride + pad_off, x*stride + pad_off) in a padded NHWC buffer
  int HoWo, Wo;         // output pixels per image / per row (1,1 for a plain GEMM)
  int Hp, Wp;           // padded height / width of the buffer
  int stride;           // spatial stride applied to (oy, ox)
  int off;              // border offset added to the pixel position (0 for the conv input: tap (0,0) = top-left pad)
  int cstride;          // channels per pixel in the buffer
  int coff;             // first channel
  int bsplit;           // image b -> (b % bsplit), channel group (b / bsplit) * cgroup  (0 = off)
  int cgroup;
};

struct IgemmParams {
  const _Float16* A;
  const _Float16* Wt;   // [N][taps*Cin]
  const float* bias;    // [N] or null
  const _Float16* R;    // residual or null
  _Float16* Y;
  int M, N, Cin, taps;
  int relu;
  IgemmGeom in, out, res;
};

__device__ __forceinline__ long long ig_row_off(const IgemmGeom& g, int m) {
  const int b = m / g.HoWo;
  const int r = m - b * g.HoWo;
  const int oy = r / g.Wo;
  const int ox = r - oy * g.Wo;
  int bb = b, cg = 0;
  if (g.bsplit > 0) { cg = b / g.bsplit; bb = b - cg * g.bsplit; }
  return (((long long)bb * g.Hp + (oy * g.stride + g.off)) * g.Wp + (ox * g.stride + g.off)) * g.cstride + g.coff +
         (long long)cg * g.cgroup;
}

__global__ __launch_bounds__(IG_THREADS, 2) void k_igemm_f16(IgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;   // wave position: pixels (m) x channels (n)

  // ---- XCD-aware tile order (bijective for any grid size)
  const int tiles_n = p.N / IG_BN;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int q = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
  const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int m0 = bm * IG_BM, n0 = bn * IG_BN;
  const int Ktot = p.taps * p.Cin;

  // ---- per-thread staging sources: wave w loads rows 32w..32w+31 of both tiles, 4 LDS-DMA instructions each
  const _Float16* asrc[4];
  const _Float16* wsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = wid * 32 + j * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);   // logical chunk that lands in physical chunk (lane & 7)
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    asrc[j] = p.A + ig_row_off(p.in, m) + c * 8;
    wsrc[j] = p.Wt + (size_t)(n0 + row) * Ktot + c * 8;
  }
  const int cpt = p.Cin / IG_BK;            // k-steps per tap
  const int nk = p.taps * cpt;

  auto stage = [&](int ks, int buf) {
    const int tap = ks / cpt;
    const int ci0 = (ks - tap * cpt) * IG_BK;
    const int ky = tap / 3, kx = tap - ky * 3;
    const long long aoff = ((long long)ky * p.in.Wp + kx) * p.in.cstride + ci0;   // 0 + ci0 for a plain GEMM (taps = 1)
    const int woff = ks * IG_BK;
    unsigned char* sa = smem + buf * IG_STAGE_BYTES + wid * 4096;
    unsigned char* sw = sa + IG_BM * IG_BK * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + aoff),
                                       (__attribute__((address_space(3))) void*)(sa + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + woff),
                                       (__attribute__((address_space(3))) void*)(sw + j * 1024), 16, 0, 0);
    }
  };

  float16_ acc[2][2];   // [channel tile i][pixel tile j]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read addressing: lane reads row (lane & 31), logical chunk 2*kk + (lane >> 5)
  const int frow = lane & 31, fhalf = lane >> 5;
  int a_rowb[2], w_rowb[2], a_sw[2], w_sw[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ra = wm * 64 + t * 32 + frow, rw = wn * 64 + t * 32 + frow;
    a_rowb[t] = ra * 128; a_sw[t] = (ra >> 1) & 7;
    w_rowb[t] = rw * 128; w_sw[t] = (rw >> 1) & 7;
  }

  stage(0, 0);
  __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): the LDS-DMA of stage 0 has landed
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) stage(ks + 1, buf ^ 1);
    const unsigned char* sa = smem + buf * IG_STAGE_BYTES;
    const unsigned char* sw = sa + IG_BM * IG_BK * 2;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int c = 2 * kk + fhalf;
      half8 fa[2], fw[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[t] = *reinterpret_cast<const half8*>(sa + a_rowb[t] + ((c ^ a_sw[t]) << 4));
        fw[t] = *reinterpret_cast<const half8*>(sw + w_rowb[t] + ((c ^ w_sw[t]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }

  // ---- epilogue: accumulators (+bias) -> half -> swizzled LDS tile E[m][n] -> 16-B coalesced row stores
  // D[i = channel][j = pixel]: lane holds pixel (lane & 31), channels 8g + 4*(lane>>5) + {0..3}, g = reg >> 2
  unsigned char* E = smem;   // 128 rows x 256 B, chunk index XORed with (m & 15)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);   // first of 4 consecutive channels (tile-local)
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = p.bias[n0 + nl + e];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ml = wm * 64 + j * 32 + (lane & 31);
        half4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (_Float16)(acc[i][j][g * 4 + e] + bv[e]);
        const int chunk = (nl >> 3) ^ (ml & 15);
        *reinterpret_cast<half4*>(E + ml * 256 + (chunk << 4) + ((nl & 4) << 1)) = v;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int qd = tid + it * IG_THREADS;
    const int ml = qd >> 4, ch = qd & 15;
    const int m = m0 + ml;
    if (m >= p.M) continue;
    half8 v = *reinterpret_cast<const half8*>(E + ml * 256 + ((ch ^ (ml & 15)) << 4));
    const int n = n0 + ch * 8;
    if (p.R) {
      const half8 rv = *reinterpret_cast<const half8*>(p.R + ig_row_off(p.res, m) + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)v[e] + (float)rv[e]);
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > (_Float16)0.f ? v[e] : (_Float16)0.f;
    }
    *reinterpret_cast<half8*>(p.Y + ig_row_off(p.out, m) + n) = v;
  }
}

static int ig_check_geom(const fp_igemm_geom* g, const char* what) {
  FP_REQUIRE(g->pixels_per_image > 0 && g->width > 0 && g->padded_h > 0 && g->padded_w > 0 && g->cstride > 0 &&
                 g->stride > 0 && g->offset >= 0 && g->coff >= 0 && g->bsplit >= 0,
             "fp_igemm_f16_fwd: bad %s geometry", what);
  FP_REQUIRE((g->cstride % 8) == 0 && (g->coff % 8) == 0 && (g->cgroup % 8) == 0,
             "fp_igemm_f16_fwd: %s channel stride/offset must be multiples of 8 (16-byte rows)", what);
  return FP_OK;
}

static IgemmGeom ig_geom(const fp_igemm_geom* g) {
  IgemmGeom o;
  o.HoWo = g->pixels_per_image; o.Wo = g->width; o.Hp = g->padded_h; o.Wp = g->padded_w; o.stride = g->stride;
  o.off = g->offset; o.cstride = g->cstride; o.coff = g->coff; o.bsplit = g->bsplit; o.cgroup = g->cgroup;
  return o;
}

extern "C" int fp_igemm_f16_fwd(const void* x, const fp_igemm_geom* x_geom, const void* w, const float* bias,
                                const void* residual, const fp_igemm_geom* r_geom, void* y, const fp_igemm_geom* y_geom,
                                int M, int N, int Cin, int taps, int relu, void* stream) {
  FP_REQUIRE(M >= 0, "fp_igemm_f16_fwd: M < 0");
  if (M == 0) return FP_OK;
  FP_REQUIRE(x && w && y && x_geom && y_geom, "fp_igemm_f16_fwd: NULL tensor / geometry");
  FP_REQUIRE(taps == 1 || taps == 9, "fp_igemm_f16_fwd: taps must be 1 (GEMM) or 9 (3x3 conv), got %d", taps);
  FP_REQUIRE(N > 0 && N % IG_BN == 0, "fp_igemm_f16_fwd: N=%d must be a multiple of %d", N, IG_BN);
  FP_REQUIRE(Cin > 0 && Cin % IG_BK == 0, "fp_igemm_f16_fwd: Cin=%d must be a multiple of %d", Cin, IG_BK);
  FP_REQUIRE(!residual || r_geom, "fp_igemm_f16_fwd: residual without geometry");
  if (int e = ig_check_geom(x_geom, "input")) return e;
  if (int e = ig_check_geom(y_geom, "output")) return e;
  if (residual) if (int e = ig_check_geom(r_geom, "residual")) return e;
  IgemmParams p;
  p.A = (const _Float16*)x; p.Wt = (const _Float16*)w; p.bias = bias; p.R = (const _Float16*)residual; p.Y = (_Float16*)y;
  p.M = M; p.N = N; p.Cin = Cin; p.taps = taps; p.relu = relu;
  p.in = ig_geom(x_geom); p.out = ig_geom(y_geom); p.res = residual ? ig_geom(r_geom) : ig_geom(y_geom);
  const long long tiles = (long long)fp_cdiv(M, IG_BM) * (N / IG_BN);
  FP_REQUIRE(tiles < (1ll << 31), "fp_igemm_f16_fwd: too many tiles");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_igemm_f16), hipFuncAttributeMaxDynamicSharedMemorySize, IG_LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_igemm_f16, dim3((unsigned)tiles), dim3(IG_THREADS), IG_LDS_BYTES, (hipStream_t)stream, p);
  FP_CHECK_LAUNCH("fp_igemm_f16_fwd");
  return FP_OK;
}
