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
#include <stdlib.h>
#include <string.h>
#include "igemm_common.h"
#include "igemm_epilogue.h"

int fp_igemm_pp_launch(const IgemmParams& p, int variant, hipStream_t stream);   // igemm_pp.hip
int fp_conv3x3_sw_launch(const IgemmParams& p, hipStream_t stream);               // conv_sw.hip (shifted-window 3x3)
bool fp_conv3x3_sw_applicable(const IgemmParams& p);
int fp_conv3x3_sw_tile_rows(const IgemmParams& p);

// Workgroup tile BM (pixels) x BN (channels) x 64 (k); every wave owns (32*TM) x 64 outputs as TM x 2
// v_mfma_f32_32x32x16_f16 tiles; NST LDS stages (prefetch distance NST-1 k-steps, counted vmcnt + raw s_barrier).
// Measured at the bench shapes: the throughput follows the operand bytes per flop, not the schedule --
// 4x2 MFMA tiles per wave (256x256 workgroup tile) halve the staged bytes per MFMA against 2x2 (128x128), and the
// 128x128 variant makes up for it with two co-resident workgroups that fill each other's barrier and epilogue gaps.
// SPLITK (round 5, fp_igemm_f16_splitk_fwd): grid.y = number of k ranges; workgroup (tile, split) multiplies k-steps
// [split * nk / nsplit, (split + 1) * nk / nsplit) of its tile and leaves the fp32 accumulators in the slab, in fragment order (a
// wave store = one contiguous KiB); k_splitk_epilogue adds the ranges in order and runs the epilogue arithmetic.
template <int BM, int BN, int TM, int NST, int BK, bool SPLITK = false>
__global__ __launch_bounds__((BM / (32 * TM)) * (BN / 64) * 64, ((BM / (32 * TM)) * (BN / 64) == 4 && TM == 4) ? 2 : 1) void k_igemm_f16(IgemmParams p) {
  constexpr int NWN = BN / 64;
  constexpr int NW = (BM / (32 * TM)) * NWN;
  constexpr int THREADS = NW * 64;
  constexpr int ROWB = BK * 2;                     // bytes per LDS row (one pixel / one output channel, BK halves)
  constexpr int CPK = BK / 8;                      // 16-byte chunks per row: 8 (BK=64) or 4 (BK=32)
  constexpr int RPI = 1024 / ROWB;                 // rows per LDS-DMA instruction (64 lanes x 16 B)
  constexpr int KK = BK / 16;                      // MFMA k-substeps per stage
  constexpr int A_BYTES = BM * ROWB;
  constexpr int STAGE_BYTES = A_BYTES + BN * ROWB;
  constexpr int AI = BM / RPI / NW;                // A-tile LDS-DMA instructions per wave and stage
  constexpr int WI = BN / RPI / NW;                // W-tile ...
  static_assert(AI >= 1 && WI >= 1, "tile too small for this wave count");
  // XOR swizzle of the chunk index: 16 rows of a ds_read_b128 lane group must land on 16 distinct 16-byte slots of
  // the 256-byte bank row, which holds 2 rows at BK=64 and 4 rows at BK=32
  auto swz = [](int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  constexpr int LPS = AI + WI;
  constexpr int LDS_MAIN = ig_lds_main<BM, BN>(NST * STAGE_BYTES);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it in an SGPR
  const int wm = wid / NWN, wn = wid - wm * NWN;   // wave position: pixels (m) x channels (n)
  float* bias_lds = reinterpret_cast<float*>(smem + LDS_MAIN);

  // ---- XCD-aware tile order (bijective for any grid size)
  const int tiles_n = p.N / BN;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int q = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
  const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int Ktot = p.taps * p.Cin;
  if constexpr (!SPLITK) ig_bias_to_lds(p, n0, bias_lds, wid, lane);

  // ---- per-thread staging sources: wave w loads rows [8*AI*w, +8*AI) of the A tile and [8*WI*w, +8*WI) of the W tile.
  // Kept as 32-bit byte offsets from the (uniform) tensor bases, so that the per-k-step part of every address is a
  // scalar and the LDS-DMA takes the saddr + voffset form: no vector ALU work per load (the first version spent more
  // issue slots on 64-bit address adds and on an integer division per k-step than on MFMAs).
  unsigned aoff32[AI], woff32[WI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int row = wid * (AI * RPI) + j * RPI + lane / CPK;
    const int c = (lane % CPK) ^ swz(row);   // logical chunk that lands in physical chunk (lane % CPK)
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    aoff32[j] = (unsigned)((ig_row_off(p.in, m) + c * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const int row = wid * (WI * RPI) + j * RPI + lane / CPK;
    const int c = (lane % CPK) ^ swz(row);
    woff32[j] = (unsigned)((((size_t)(n0 + row) * Ktot) + c * 8) * 2);
  }
  int nk = p.taps * (p.Cin / BK);
  // running state of the NEXT k-step to stage: (ky, kx, ci0) advance without divisions; all scalar
  int st_ci0 = 0, st_kx = 0, st_ky = 0, st_k = 0;
  const int inWp = p.in.Wp, inCs = p.in.cstride, Cin = p.Cin;
  if constexpr (SPLITK) {
    const int split = blockIdx.y, cpt = Cin / BK;
    const int ks0 = (int)(((long long)split * nk) / p.nsplit), ks1 = (int)(((long long)(split + 1) * nk) / p.nsplit);
    const int tap = ks0 / cpt;
    st_k = ks0; st_ci0 = (ks0 - tap * cpt) * BK; st_ky = tap / 3; st_kx = tap - 3 * st_ky;
    nk = ks1 - ks0;
  }
  // buffer descriptors (wave-uniform): LDS-DMA as `buffer_load_dwordx4 voff, rsrc, soff offen lds` -- per-lane byte
  // offset in a VGPR computed once, per-k-step offset in an SGPR, LDS destination in M0: no vector ALU work per load
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Wt), 0, 0x7FFFFFFF, 0x00020000);

  auto stage = [&](int buf) {
    const int asoff = (((st_ky * inWp + st_kx) * inCs) + st_ci0) * 2;   // bytes; just ci0 for a plain GEMM (taps = 1)
    const int wsoff = st_k * (BK * 2);
    unsigned char* sa = smem + buf * STAGE_BYTES + wid * (AI * 1024);
    unsigned char* sw = smem + buf * STAGE_BYTES + A_BYTES + wid * (WI * 1024);
#pragma unroll
    for (int j = 0; j < AI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(sa + j * 1024), 16,
                                               (int)aoff32[j], asoff, 0, 0);
#pragma unroll
    for (int j = 0; j < WI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(sw + j * 1024), 16,
                                               (int)woff32[j], wsoff, 0, 0);
    ++st_k;
    st_ci0 += BK;
    if (st_ci0 == Cin) { st_ci0 = 0; if (++st_kx == 3) { st_kx = 0; ++st_ky; } }
  };

  float16_ acc[2][TM];   // [channel tile i][pixel tile j]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read addressing: lane reads row (lane & 31), logical chunk 2*kk + (lane >> 5); the swizzled byte offsets
  // of the 4 k-substeps are computed once, a read costs one add of the (scalar) stage base
  const int frow = lane & 31, fhalf = lane >> 5;
  int a_off[TM][KK], w_off[2][KK];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int ra = wm * (32 * TM) + t * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) a_off[t][kk] = ra * ROWB + (((2 * kk + fhalf) ^ swz(ra)) << 4);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rw = wn * 64 + t * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) w_off[t][kk] = A_BYTES + rw * ROWB + (((2 * kk + fhalf) ^ swz(rw)) << 4);
  }

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) stage(s);
  int buf = 0, nbuf = NST - 1;
  for (int ks = 0; ks < nk; ++ks) {
    // stage ks must have landed; the NST-2 stages issued after it may stay in flight (loads retire in order)
    if (NST > 2 && ks + NST - 2 < nk) {
      constexpr int INFLIGHT = LPS * (NST - 2);
      static_assert(NST <= 2 || INFLIGHT == 4 || INFLIGHT == 6 || INFLIGHT == 8 || INFLIGHT == 12, "add the counted wait");
      if (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (INFLIGHT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();     // everyone's part of stage ks is visible; everyone is done reading stage ks-1
    if (ks + NST - 1 < nk) stage(nbuf);
    const unsigned char* sb = smem + buf * STAGE_BYTES;
    // fragment double buffer: the ds_read_b128 of k-substep kk+1 are issued before the MFMAs of kk
    half8 fa[2][TM], fw[2][2];
    auto load_frags = [&](int kk, int slot) {
#pragma unroll
      for (int t = 0; t < TM; ++t) fa[slot][t] = *reinterpret_cast<const half8*>(sb + a_off[t][kk]);
#pragma unroll
      for (int t = 0; t < 2; ++t) fw[slot][t] = *reinterpret_cast<const half8*>(sb + w_off[t][kk]);
    };
    load_frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (kk < KK - 1) load_frags(kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above the MFMAs (hipcc otherwise sinks it below them)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk & 1][i], fa[kk & 1][j], acc[i][j], 0, 0, 0);
    }
    buf = (buf + 1 == NST) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == NST) ? 0 : nbuf + 1;
  }
  if constexpr (SPLITK) {
    typedef float float4_ __attribute__((ext_vector_type(4)));
    float4_* dst = reinterpret_cast<float4_*>(p.slab) + ((((size_t)blockIdx.y * nwg + tile) * NW + wid) * (8 * TM)) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const float4_ v = {acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
          dst[((i * 4 + g) * TM + j) * 64] = v;
        }
    return;
  }
  __syncthreads();   // all fragment reads done before the staging buffers become the transpose tile

  ig_epilogue<BM, BN, TM, THREADS, 0>(p, acc, smem, m0, n0, wm, wn, tid, lane, bias_lds);
}

template <int BM, int BN, int TM, int NST, int BK>
static int ig_launch(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDS = ig_lds_main<BM, BN>(NST * (BM + BN) * BK * 2) + IG_BIAS_LDS;
  constexpr int THREADS = (BM / (32 * TM)) * (BN / 64) * 64;
  static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  const long long tiles = (long long)fp_cdiv(p.M, BM) * (p.N / BN);
  FP_REQUIRE(tiles < (1ll << 31), "fp_igemm_f16_fwd: too many tiles");
  FP_SET_MAX_LDS((k_igemm_f16<BM, BN, TM, NST, BK>), LDS);
  hipLaunchKernelGGL((k_igemm_f16<BM, BN, TM, NST, BK>), dim3((unsigned)tiles), dim3(THREADS), LDS, stream, p);
  FP_CHECK_LAUNCH("fp_igemm_f16_fwd");
  return FP_OK;
}

// The second half of a split-K product: one thread per accumulator quad (4 consecutive channels of one pixel) adds the k ranges IN
// RANGE ORDER (deterministic; another fp32 summation order than the unsplit kernel's) and applies the epilogue of
// igemm_epilogue.h to its four values -- the same operations in the same order per element: bias / conv rounding / BatchNorm,
// residual add as an IEEE half add, ReLU, the optional positional second output.
template <int BM, int BN, int TM>
__global__ __launch_bounds__(256) void k_splitk_epilogue(IgemmParams p, int ntiles) {
  constexpr int NWN = BN / 64, NW = (BM / (32 * TM)) * NWN;
  typedef float float4_ __attribute__((ext_vector_type(4)));
  typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = (int)(t & 63);
  size_t q = t >> 6;
  const int j = (int)(q % TM); q /= TM;
  const int g = (int)(q & 3); q >>= 2;
  const int i = (int)(q & 1); q >>= 1;
  const int wid = (int)(q % NW);
  const int tile = (int)(q / NW);
  if (tile >= ntiles) return;
  const int tiles_n = p.N / BN;
  const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int wm = wid / NWN, wn = wid - wm * NWN;
  const int m = bm * BM + wm * (32 * TM) + j * 32 + (lane & 31);
  const int n = bn * BN + wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
  if (m >= p.M) return;
  const float4_* src = reinterpret_cast<const float4_*>(p.slab) + t;
  const size_t per_split = (size_t)ntiles * NW * (8 * TM) * 64;
  float4_ a = src[0];
  for (int s = 1; s < p.nsplit; ++s) {
    const float4_ b = src[(size_t)s * per_split];
    a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
  }
  float4_ bv = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bv = *reinterpret_cast<const float4_*>(p.bias + n);
  half4 v;
  if (p.round_acc) {
    const half2_ b01 = {(_Float16)bv[0], (_Float16)bv[1]}, b23 = {(_Float16)bv[2], (_Float16)bv[3]};
    half2_ t01 = {(_Float16)a[0], (_Float16)a[1]};
    half2_ t23 = {(_Float16)a[2], (_Float16)a[3]};
    t01 = t01 + b01;
    t23 = t23 + b23;
    if (p.bn_scale) {
      const float4_ sc = *reinterpret_cast<const float4_*>(p.bn_scale + n), sh = *reinterpret_cast<const float4_*>(p.bn_shift + n);
      v[0] = (_Float16)fmaf((float)t01[0], sc[0], sh[0]);
      v[1] = (_Float16)fmaf((float)t01[1], sc[1], sh[1]);
      v[2] = (_Float16)fmaf((float)t23[0], sc[2], sh[2]);
      v[3] = (_Float16)fmaf((float)t23[1], sc[3], sh[3]);
    } else {
      v[0] = t01[0]; v[1] = t01[1]; v[2] = t23[0]; v[3] = t23[1];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (_Float16)(a[e] + bv[e]);
  }
  if (p.R) v = v + *reinterpret_cast<const half4*>(p.R + ig_row_off(p.res, m) + n);
  if (p.relu) {
    const half4 zero = {0, 0, 0, 0};
    v = __builtin_elementwise_max(v, zero);
  }
  *reinterpret_cast<half4*>(p.Y + ig_row_off(p.out, m) + n) = v;
  if (p.Ype) {
    const float4_ e0 = *reinterpret_cast<const float4_*>(p.pe + (size_t)(m % p.pe_period) * p.N + n);
    half4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (_Float16)((float)v[e] + e0[e]);
    *reinterpret_cast<half4*>(p.Ype + (size_t)m * p.N + n) = w;
  }
}

// split-K always runs the 128 x 128 x 64 tile (4 waves, two stages): it exists for launches of a few dozen tiles
constexpr int SK_BM = 128, SK_BN = 128, SK_TM = 2, SK_NST = 2, SK_BK = 64;
static size_t ig_splitk_bytes(int M, int N, int splits) {
  return (size_t)splits * fp_cdiv(M, SK_BM) * (N / SK_BN) * (size_t)(SK_BM * SK_BN * 4);
}
static int ig_launch_splitk(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDS = ig_lds_main<SK_BM, SK_BN>(SK_NST * (SK_BM + SK_BN) * SK_BK * 2) + IG_BIAS_LDS;
  const long long tiles = (long long)fp_cdiv(p.M, SK_BM) * (p.N / SK_BN);
  FP_REQUIRE(tiles * p.nsplit < (1ll << 31) && p.nsplit <= 65535, "fp_igemm_f16_splitk_fwd: too many tiles");
  FP_SET_MAX_LDS((k_igemm_f16<SK_BM, SK_BN, SK_TM, SK_NST, SK_BK, true>), LDS);
  hipLaunchKernelGGL((k_igemm_f16<SK_BM, SK_BN, SK_TM, SK_NST, SK_BK, true>), dim3((unsigned)tiles, (unsigned)p.nsplit), dim3(256), LDS, stream, p);
  FP_CHECK_LAUNCH("fp_igemm_f16_splitk_fwd(partial products)");
  const long long threads = tiles * 4 * (8 * SK_TM) * 64;
  hipLaunchKernelGGL((k_splitk_epilogue<SK_BM, SK_BN, SK_TM>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, p, (int)tiles);
  FP_CHECK_LAUNCH("fp_igemm_f16_splitk_fwd(reduce + epilogue)");
  return FP_OK;
}

static int ig_check_geom(const fp_igemm_geom* g, const char* what, const char* who) {
  FP_REQUIRE(g->pixels_per_image > 0 && g->width > 0 && g->padded_h > 0 && g->padded_w > 0 && g->cstride > 0 &&
                 g->stride > 0 && g->offset >= 0 && g->coff >= 0 && g->bsplit >= 0,
             "%s: bad %s geometry", who, what);
  FP_REQUIRE((g->cstride % 8) == 0 && (g->coff % 8) == 0 && (g->cgroup % 8) == 0,
             "%s: %s channel stride/offset must be multiples of 8 (16-byte rows)", who, what);
  return FP_OK;
}

static IgemmGeom ig_geom(const fp_igemm_geom* g) {
  IgemmGeom o;
  o.HoWo = g->pixels_per_image; o.Wo = g->width; o.Hp = g->padded_h; o.Wp = g->padded_w; o.stride = g->stride;
  o.off = g->offset; o.cstride = g->cstride; o.coff = g->coff; o.bsplit = g->bsplit; o.cgroup = g->cgroup;
  ig_fastdiv_init(o.HoWo, &o.mulP, &o.shrP);
  ig_fastdiv_init(o.Wo, &o.mulW, &o.shrW);
  ig_fastdiv_init(o.bsplit, &o.mulB, &o.shrB);
  return o;
}

static int ig_dispatch(IgemmParams& p, hipStream_t stream) {
  int sel = 0;
#ifdef FP_PROFILE_BUILD
  // profiling builds only (make profile): FP_IGEMM_TILE = 128x128 | 256x128 | 256x256 | pp256x256 | pp256x128 | generic
  // forces one schedule; the release library has no environment switches on this path
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("FP_IGEMM_TILE");
    forced = 0;
    if (e) {
      if (!strcmp(e, "128x128")) forced = 1;
      else if (!strcmp(e, "256x128")) forced = 2;
      else if (!strcmp(e, "256x256")) forced = 3;
      else if (!strcmp(e, "ls256x128")) forced = 4;   // 4 waves x (128 x 64), 3 stages of BK=32: two workgroups per CU
      else if (!strcmp(e, "pp256x256")) forced = 7;
      else if (!strcmp(e, "pp256x128")) forced = 9;
      else if (!strcmp(e, "generic")) forced = 100;   // default selection without the shifted-window kernel
    }
  }
  sel = forced;
#endif
  const bool sw = sel == 0 && fp_conv3x3_sw_applicable(p);
  if (sel >= 100) sel = 0;
  // measured at the bench shapes, full batch (M = 252 x 400) and sub-batch (126 x 400), scripts/bench_igemm.py with
  // FP_N=252|126 and FP_IGEMM_TILE forced (profiles/r02_g_igemm_tiles.log): the 256x256 ping-pong kernel wins for wide
  // outputs (QKV projection, 0.74 against 0.72 PF/s), for the stride-2 conv with a long reduction (256->512: 0.92-0.98
  // against 0.89-0.90) and for the 512-wide Linear layers below ~75 k rows (0.63 against 0.58; above, two 128x128
  // workgroups per CU win 0.67 against 0.62); 128x128 elsewhere (64->128 stride 2: 0.54 against 0.47)
  if (!sw && sel == 0) {
    const bool wide = (p.N % 256) == 0;
    const bool pp = wide && (p.M >= 150000 || p.N >= 1024 || (p.taps == 9 && p.Cin >= 256) || (p.taps == 1 && p.M < 75000 && p.M >= 4096));
    sel = pp ? 7 : 1;
    // round 3 (profiles/r03_igemm_ab.log): the lock-step 256x128 tile with 4 waves of 128x64 and two workgroups per CU beats
    // both for the stem's stride-2 conv (64->128: 0.59-0.61 against 0.53-0.55 PF/s) and for the QKV projection (0.73-0.74
    // against 0.69-0.71); it loses on the 256->512 stride-2 conv and the 512-wide Linear layers at sub-batch size
    if ((p.taps == 9 && p.in.stride == 2 && p.Cin <= 64) || (p.taps == 1 && p.N >= 1024)) sel = 4;
  }
  if (sel == 3 && (p.N % 256) != 0) sel = 2;
  if (sel == 7 && (p.N % 256) != 0) sel = 9;
  if (sw) return fp_conv3x3_sw_launch(p, stream);
  if (sel >= 7) return fp_igemm_pp_launch(p, sel - 7, stream);
#ifdef FP_PROFILE_BUILD
  if (sel == 2) return ig_launch<256, 128, 2, 3, 64>(p, stream);
  if (sel == 3) return ig_launch<256, 256, 4, 2, 64>(p, stream);
#endif
  if (sel == 4) return ig_launch<256, 128, 4, 3, 32>(p, stream);
  return ig_launch<128, 128, 2, 2, 64>(p, stream);
}


static int ig_build_params(const void* x, const fp_igemm_geom* x_geom, const void* w, void* y, const fp_igemm_geom* y_geom,
                           int M, int N, int Cin, int taps, const fp_igemm_epilogue* ep, IgemmParams& p, const char* who) {
  FP_REQUIRE(x && w && y && x_geom && y_geom, "%s: NULL tensor / geometry", who);
  FP_REQUIRE(taps == 1 || taps == 9, "%s: taps must be 1 (GEMM) or 9 (3x3 conv), got %d", who, taps);
  FP_REQUIRE(N > 0 && N % 128 == 0, "%s: N=%d must be a multiple of 128", who, N);
  FP_REQUIRE(Cin > 0 && Cin % 64 == 0, "%s: Cin=%d must be a multiple of 64", who, Cin);
  static const fp_igemm_epilogue no_epilogue = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr};
  const fp_igemm_epilogue& e = ep ? *ep : no_epilogue;
  FP_REQUIRE(!e.residual || e.r_geom, "%s: residual without geometry", who);
  FP_REQUIRE((e.bn_scale == nullptr) == (e.bn_shift == nullptr), "%s: bn_scale and bn_shift go together", who);
  FP_REQUIRE(!e.bn_scale || (e.flags & FP_IGEMM_ROUND_ACC), "%s: BatchNorm needs FP_IGEMM_ROUND_ACC (conv semantics)", who);
  FP_REQUIRE((e.flags & ~(FP_IGEMM_RELU | FP_IGEMM_ROUND_ACC | FP_IGEMM_HAS_W_TILES)) == 0, "%s: unknown flags 0x%x", who, e.flags);
  FP_REQUIRE((e.pe == nullptr) == (e.y_pe == nullptr) && (!e.pe || e.pe_period > 0), "%s: pe, pe_period and y_pe go together", who);
  FP_REQUIRE((((size_t)x | (size_t)w | (size_t)y | (size_t)e.residual | (size_t)e.bias | (size_t)e.bn_scale | (size_t)e.bn_shift |
               (size_t)e.pe | (size_t)e.y_pe) & 15) == 0, "%s: tensors must be 16-byte aligned", who);
  if (int err = ig_check_geom(x_geom, "input", who)) return err;
  if (int err = ig_check_geom(y_geom, "output", who)) return err;
  if (e.residual) if (int err = ig_check_geom(e.r_geom, "residual", who)) return err;
  // w_tiles is a trailing member added in ABI 212: it is READ only when the caller says the struct has it (a native caller compiled
  // against an older header passes a shorter struct and can never set the bit)
  const void* w_tiles = (e.flags & FP_IGEMM_HAS_W_TILES) ? e.w_tiles : nullptr;
  FP_REQUIRE(!w_tiles || (taps == 9 && (((size_t)w_tiles) & 15) == 0 && w_tiles != w), "%s: w_tiles is for 3x3 convolutions, 16-byte aligned, not w itself", who);
  p.A = (const _Float16*)x; p.Wt = (const _Float16*)w; p.Wpk = (const _Float16*)w_tiles; p.bias = e.bias; p.bn_scale = e.bn_scale; p.bn_shift = e.bn_shift;
  p.R = (const _Float16*)e.residual; p.Y = (_Float16*)y;
  p.pe = e.pe; p.Ype = (_Float16*)e.y_pe; p.pe_period = e.pe_period;
  p.M = M; p.N = N; p.Cin = Cin; p.taps = taps; p.relu = (e.flags & FP_IGEMM_RELU) ? 1 : 0;
  p.round_acc = (e.flags & FP_IGEMM_ROUND_ACC) ? 1 : 0;
  p.in = ig_geom(x_geom); p.out = ig_geom(y_geom); p.res = e.residual ? ig_geom(e.r_geom) : ig_geom(y_geom);
  p.slab = nullptr; p.nsplit = 0;
  return FP_OK;
}

extern "C" int fp_igemm_f16_fwd(const void* x, const fp_igemm_geom* x_geom, const void* w, void* y, const fp_igemm_geom* y_geom,
                                int M, int N, int Cin, int taps, const fp_igemm_epilogue* ep, void* stream) {
  FP_REQUIRE(M >= 0, "fp_igemm_f16_fwd: M < 0");
  if (M == 0) return FP_OK;
  IgemmParams p;
  if (int err = ig_build_params(x, x_geom, w, y, y_geom, M, N, Cin, taps, ep, p, "fp_igemm_f16_fwd")) return err;
  return ig_dispatch(p, (hipStream_t)stream);
}

extern "C" size_t fp_igemm_splitk_workspace_bytes(int M, int N, int splits) {
  if (M <= 0 || N <= 0 || N % SK_BN != 0 || splits <= 0) return 0;
  return ig_splitk_bytes(M, N, splits);
}

extern "C" int fp_igemm_f16_splitk_fwd(const void* x, const fp_igemm_geom* x_geom, const void* w, void* y, const fp_igemm_geom* y_geom,
                                       int M, int N, int Cin, int taps, const fp_igemm_epilogue* ep, int splits, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  FP_REQUIRE(M >= 0, "fp_igemm_f16_splitk_fwd: M < 0");
  if (M == 0) return FP_OK;
  IgemmParams p;
  if (int err = ig_build_params(x, x_geom, w, y, y_geom, M, N, Cin, taps, ep, p, "fp_igemm_f16_splitk_fwd")) return err;
  FP_REQUIRE(!p.Wpk, "fp_igemm_f16_splitk_fwd: w_tiles is the operand layout of the shifted-window kernel; the split-K kernels do not take it");
  const int nk = taps * (Cin / SK_BK);
  FP_REQUIRE(splits >= 1 && splits <= nk, "fp_igemm_f16_splitk_fwd: splits=%d must be in [1, %d] (k-steps of 64)", splits, nk);
  const size_t need = ig_splitk_bytes(M, N, splits);
  if (!workspace || workspace_bytes < need || ((size_t)workspace & 15)) {
    fp_set_error("fp_igemm_f16_splitk_fwd: workspace too small or unaligned (%zu < %zu bytes, see fp_igemm_splitk_workspace_bytes)", workspace_bytes, need);
    return FP_ERR_WORKSPACE;
  }
  p.slab = (float*)workspace; p.nsplit = splits;
  return ig_launch_splitk(p, (hipStream_t)stream);
}
