// k_conv_sw -- shifted-window 3x3 convolution (stride 1, pad 1) on the ping-pong schedule of igemm_pp.hip: the kernel
// behind fp_igemm_f16_fwd for the twelve ResnetBasicBlock convolutions of each encoder (network_modules.py:73-111;
// refine_network.py:40-49, score_network.py:39-48), where input and output share one padded pixel grid.
//
// Why: as an implicit GEMM with k = (tap, ci) every workgroup fetches each of its activation rows nine times, once per
// tap (measured on the 256->256 layer: 1.33 GB moved for 0.43 GB of tensors, and an LDS-DMA issue rate that keeps the
// memory cluster of the ping-pong loop longer than its MFMA cluster).  Here the k order is (channel chunk,
// tap): per 32-channel chunk ONE patch of the padded input -- the tile's 256 output pixels plus a halo of Wp+1 pixels
// on either side, as one contiguous run of the padded NHWC grid -- is staged in LDS, and the nine taps read it at a row
// shift of ky*Wp + kx.  Rows of the GEMM stay the true output pixels (no border work, same epilogue): a lane's fragment
// row for tap (ky,kx) is its pixel's position in the patch + the shift.
//
// Operand traffic per k-step drops from (256 + BN) rows to BN rows + 1/9 patch; LDS-DMA instructions per wave and
// k-step from 4 to 2.4 (BN = 256) and from 3 to 1.4 (BN = 128).
//
// LDS: two patch buffers of 512 rows x 64 B (double buffered across chunks) + a ring of 4 weight stages of BN x 64 B;
// 64-byte rows XOR-swizzled as in igemm_pp.hip (chunk ^ ((row >> 2) & 3), a bijection of row mod 16, so 32 consecutive
// patch rows at ANY offset are conflict-free for ds_read_b128).  Schedule: two groups of 4 waves one cluster apart,
// a k-step = memory cluster (12 fragment reads + this step's LDS-DMA) | barrier | 16 MFMAs | barrier; weights are
// prefetched three k-steps ahead, the next chunk's patch is requested one piece per wave at taps 0-3 of the current chunk.
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "igemm_common.h"
#include "igemm_epilogue.h"

#ifdef FP_PROFILE_BUILD
// profiling build only: 100 MHz wall-clock time per phase, summed over the workgroups of a launch (scripts/dbg_conv_sw.py)
__device__ unsigned long long sw_dbg[8];
extern "C" int fp_dbg_conv_sw(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, ~0ull, 0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(sw_dbg), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sw_dbg), 8 * sizeof(unsigned long long));
}
// the epilogue's own phase timers (igemm_epilogue.h, this translation unit's copy): out[0..7]
extern "C" int fp_dbg_conv_sw_epilogue(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(ig_epi_dbg), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ig_epi_dbg), 8 * sizeof(unsigned long long));
}
#define SW_CLK(t) const unsigned long long t = wall_clock64()
#else
#define SW_CLK(t)
#endif

#ifndef SW_DEFAULT_VARIANT
#define SW_DEFAULT_VARIANT 0
#endif

namespace {

constexpr int SW_BK = 32, SW_NSTW = 4;
// patch rows per buffer (LDS-DMA instructions of 16 rows, a multiple of 8 of them): the tile's BM output pixels + halo,
// see fp_conv3x3_sw_applicable
constexpr int sw_prows(int BM) { return BM == 256 ? 512 : 768; }

template <int N>
__device__ __forceinline__ void sw_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// padded-grid index of tap (0,0) of output pixel m: (b * Hp + oy) * Wp + ox
__device__ __forceinline__ int sw_q(const IgemmGeom& g, int m) {
  const int b = ig_fastdiv(m, g.mulP, g.shrP);
  const int r = m - b * g.HoWo;
  const int oy = ig_fastdiv(r, g.mulW, g.shrW);
  const int ox = r - oy * g.Wo;
  return (b * g.Hp + oy) * g.Wp + ox;
}

// Tile shapes: 512 x 128 (the product's, every layer since round 4: see SW_FORCE_512 below) and 256 x 256 (N % 256 == 0); each wave owns
// 128 x 64 outputs = 16 MFMAs per k-step either way.
template <int BM, int BN, int TM>
__global__ __launch_bounds__(512, 1) void k_conv_sw(IgemmParams p) {
  constexpr int BK = SW_BK, NW = 8, THREADS = 512;
  constexpr int NWN = BN / 64;
  static_assert((BM / (32 * TM)) * NWN == NW, "8 waves");
  constexpr int ROWB = BK * 2;                      // 64-byte LDS rows
  constexpr int W_BYTES = BN * ROWB;
  constexpr int SW_PROWS = sw_prows(BM);
  constexpr int SW_PATCH_BYTES = SW_PROWS * ROWB;
  constexpr int WI = BN / 16 / NW;                  // weight LDS-DMA instructions per wave and k-step (16 rows each)
  constexpr int PI = SW_PROWS / 16 / NW;            // patch instructions per wave and chunk: one at each of taps 0..PI-1
  static_assert(WI >= 1 && PI >= 1 && PI <= 7 && SW_PROWS == PI * 16 * NW, "tile shape");
  constexpr int STAGES_BYTES = 2 * SW_PATCH_BYTES + SW_NSTW * W_BYTES;
  constexpr int LDS_MAIN = ig_lds_main<BM, BN>(STAGES_BYTES);
  auto swz = [](int row) { return (row >> 2) & 3; };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  SW_CLK(t_start);
  unsigned char* const patch = smem;                              // 2 x SW_PATCH_BYTES
  unsigned char* const wring = smem + 2 * SW_PATCH_BYTES;         // SW_NSTW x W_BYTES
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;
  const int wm = wid / NWN, wn = wid - wm * NWN;
  float* bias_lds = reinterpret_cast<float*>(smem + LDS_MAIN);

  // XCD-aware tile order (as igemm.hip): neighbouring pixel tiles (overlapping halos) and the channel tiles of one pixel
  // tile run on the same XCD
  const int tiles_n = p.N / BN;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (qd8 + 1) : r8 * (qd8 + 1) + (xcd - r8) * qd8) + loc;
  const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int Cin = p.Cin, Ktot = 9 * Cin, Wp = p.in.Wp;
  const int ncc = Cin / BK;
  ig_bias_to_lds(p, n0, bias_lds, wid, lane);

  // ---- DMA sources.  Patch row R <-> padded pixel q0 + R (clamped into the tensor: clamped rows are never read by a
  // row that is stored).  Wave w owns patch instructions 4w..4w+3 (16 rows each).
  const int q0 = sw_q(p.in, m0);
  const int qmax = (p.M / p.in.HoWo) * p.in.Hp * Wp - 1;
  unsigned poff32[PI], woff32[WI];
#pragma unroll
  for (int j = 0; j < PI; ++j) {
    const int row = (wid * PI + j) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz(row);
    int q = q0 + row;
    q = q < qmax ? q : qmax;
    poff32[j] = (unsigned)(((long long)q * p.in.cstride + p.in.coff + c * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const int row = (wid * WI + j) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz(row);
    woff32[j] = (unsigned)((((size_t)(n0 + row) * Ktot) + c * 8) * 2);
  }
  // Tile-packed weights (round 5, fp_pack_conv3x3_tiles_f16; BN = 128 only): the 8 KiB a k-step stages are one contiguous run that
  // already is the LDS image, so a piece is 1 KiB of consecutive addresses (8 whole 128-byte lines) instead of 16 half lines of 16 weight rows
  const bool wpk = (BN == 128) && p.Wpk != nullptr;
  if (wpk) {
#pragma unroll
    for (int j = 0; j < WI; ++j) woff32[j] = (unsigned)(((wid * WI + j) * 64 + lane) * 16);
  }
  const int wpk_base = wpk ? bn * (ncc * 9) * W_BYTES : 0;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(wpk ? p.Wpk : p.Wt), 0, 0x7FFFFFFF, 0x00020000);

  auto stage_patch = [&](int cc, int j) {          // piece j (0..3) of this wave of the patch of chunk cc
    unsigned char* dst = patch + (cc & 1) * SW_PATCH_BYTES + (wid * PI + j) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, (int)poff32[j],
                                             cc * (BK * 2), 0, 0);
  };
  auto stage_w = [&](int cc, int tap, int slot) {  // weight columns [tap*Cin + cc*32, +32) of the tile's BN rows
    const int wsoff = wpk ? wpk_base + (cc * 9 + tap) * W_BYTES : (tap * Cin + cc * BK) * 2;
    unsigned char* dst = wring + slot * W_BYTES + wid * (WI * 1024);
#pragma unroll
    for (int j = 0; j < WI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16,
                                               (int)woff32[j], wsoff, 0, 0);
  };

  float16_ acc[2][TM];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addressing: the lane's pixel rows (position in the patch at tap (0,0)) and its weight rows
  const int frow = lane & 31, fhalf = lane >> 5;
  int arow[TM], w_off[2][2];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    int m = m0 + wm * (32 * TM) + t * 32 + frow;
    m = m < p.M ? m : p.M - 1;
    arow[t] = sw_q(p.in, m) - q0;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rw = wn * 64 + t * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) w_off[t][kk] = rw * ROWB + (((2 * kk + fhalf) ^ swz(rw)) << 4);
  }

  // ---- prologue: patch(0) + W of k-steps 0..2 in flight; patch(0) and W(0) landed and visible
#pragma unroll
  for (int j = 0; j < PI; ++j) stage_patch(0, j);
  stage_w(0, 0, 0);
  stage_w(0, 1, 1);
  stage_w(0, 2, 2);
  sw_wait_vm<2 * WI>();
  __builtin_amdgcn_s_barrier();
  SW_CLK(t_cold);
  if (grp) __builtin_amdgcn_s_barrier();           // group 1 sits out interval 0

  half8 fa[2][TM], fw[2][2];
  static_assert(TM == 4, "the four fragment addresses of the next tap are computed behind the four MFMA groups of a k-step");
  int anext[TM];                                   // fragment addresses of the next k-step's tap, inside a patch buffer
#pragma unroll
  for (int t = 0; t < TM; ++t) anext[t] = (arow[t] << 6) + ((fhalf ^ swz(arow[t])) << 4);     // tap (0,0) of the first k-step
  // one k-step = (chunk cc, tap T).  LAST: cc is the last chunk (no next patch; the weight prefetch runs dry).
  // vmcnt bookkeeping (loads retire in order): this wave's pieces of W(s+1) must have landed when it leaves the memory
  // cluster; younger and allowed in flight are W(s+2), W(s+3) and the patch pieces issued in this and the previous step.
  auto kstep = [&](int cc, auto tap_c, auto last_c) {
    constexpr int T = decltype(tap_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    constexpr int ky = T / 3, kx = T - 3 * ky;
    const int s = cc * 9 + T;
    const unsigned char* pb = patch + (cc & 1) * SW_PATCH_BYTES;
    const unsigned char* wb = wring + (s & (SW_NSTW - 1)) * W_BYTES;
    const int shift = ky * Wp + kx;
    // ---- memory cluster.  Round 6: the tap's four fragment addresses (8 vector instructions each) are NOT computed here any more but
    // behind the MFMAs of the previous k-step (anext, below): the stand-alone probe of this loop (scripts/conv_loop_probe,
    // profiles/r06_i_conv_loop_probe.log) put the memory cluster at ~610 clk against the partner group's 512 clk of MFMAs, and without
    // the 32 address instructions at ~535 (matrix-pipe occupancy 0.84 -> 0.96); in the MFMA shadow they are free (<= 5 per MFMA hide)
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int a0 = anext[t];
      fa[0][t] = *reinterpret_cast<const half8*>(pb + a0);
      fa[1][t] = *reinterpret_cast<const half8*>(pb + (a0 ^ 32));
    }
    (void)shift;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int t = 0; t < 2; ++t) fw[kk][t] = *reinterpret_cast<const half8*>(wb + w_off[t][kk]);
    if constexpr (!LAST) {
      if constexpr (T < PI) stage_patch(cc + 1, T);
      // k-step s+3 = (cc, T+3) or (cc+1, T-6)
      if constexpr (T + 3 < 9) stage_w(cc, T + 3, (s + 3) & (SW_NSTW - 1));
      else stage_w(cc + 1, T - 6, (s + 3) & (SW_NSTW - 1));
      // patch pieces are issued at taps 0..PI-1: in flight from this step and the previous one
      constexpr int NP = (T < PI ? 1 : 0) + ((T >= 1 && T - 1 < PI) ? 1 : 0);
      sw_wait_vm<2 * WI + NP>();
    } else {
      if constexpr (T + 3 < 9) stage_w(cc, T + 3, (s + 3) & (SW_NSTW - 1));
      // at T == 0 the previous step (tap 8 of the chunk before) issued no patch piece, and the last chunk issues none
      constexpr int NYOUNGER = T <= 5 ? 2 : (T == 6 ? 1 : 0);
      sw_wait_vm<NYOUNGER * WI>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads retired before the buffers can be refilled
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- compute cluster
    __builtin_amdgcn_s_setprio(1);
    {
      // the NEXT k-step's tap: (T + 1) % 9 -- the patch buffer (chunk parity) is added where the address is used
      constexpr int TN = (T + 1) % 9, kyn = TN / 3, kxn = TN - 3 * kyn;
      const int shift_n = kyn * Wp + kxn;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk][i], fa[kk][j], acc[i][j], 0, 0, 0);
          // one row tile's address behind each group of four MFMAs (TM == 4 row tiles = the four (kk, i) groups)
          const int t = kk * 2 + i;                // a constant after unrolling
          {
            int ar = arow[t];
            asm volatile("" : "+v"(ar));           // keep the per-tap recomputation: 36 hoisted tap addresses would spill
            const int pr = ar + shift_n;
            anext[t] = (pr << 6) + ((fhalf ^ swz(pr)) << 4);
          }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto chunk = [&](int cc, auto last_c) {
    kstep(cc, std::integral_constant<int, 0>{}, last_c);
    kstep(cc, std::integral_constant<int, 1>{}, last_c);
    kstep(cc, std::integral_constant<int, 2>{}, last_c);
    kstep(cc, std::integral_constant<int, 3>{}, last_c);
    kstep(cc, std::integral_constant<int, 4>{}, last_c);
    kstep(cc, std::integral_constant<int, 5>{}, last_c);
    kstep(cc, std::integral_constant<int, 6>{}, last_c);
    kstep(cc, std::integral_constant<int, 7>{}, last_c);
    kstep(cc, std::integral_constant<int, 8>{}, last_c);
  };
  for (int cc = 0; cc + 1 < ncc; ++cc) chunk(cc, std::false_type{});
  chunk(ncc - 1, std::true_type{});
  if (!grp) __builtin_amdgcn_s_barrier();          // group 0 waits out group 1's last compute cluster
  __syncthreads();
  SW_CLK(t_loop);
  ig_epilogue<BM, BN, TM, THREADS, 0>(p, acc, smem, m0, n0, wm, wn, tid, lane, bias_lds);
#ifdef FP_PROFILE_BUILD
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores drained: the workgroup's resources are free from here
  SW_CLK(t_end);
  if (tid == 0) {
    atomicAdd(&sw_dbg[0], t_cold - t_start); atomicAdd(&sw_dbg[1], t_loop - t_cold); atomicAdd(&sw_dbg[2], t_end - t_loop);
    atomicAdd(&sw_dbg[3], 1ull); atomicMin(&sw_dbg[6], t_start); atomicMax(&sw_dbg[7], t_end);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// k_conv_sw_ls -- the same shifted-window convolution as a LOCK-STEP kernel with TWO workgroups per CU (round 3).
// Why: k_conv_sw holds one 512-thread workgroup per CU, so nothing runs on the CU during a tile's cold start (first patch
// + weights: ~4 us) and epilogue (~8 us) -- 16 % of a 256->256 tile, 29 % of a 128-channel tile; making the
// kernel persistent serialised the same waves and was slower.  Here a workgroup is 4 waves (one per SIMD) on a 256 x 128
// tile -- each wave still owns 128 x 64 outputs = 16 MFMAs per k-step, the intensity of the ping-pong kernel -- and its
// LDS is exactly half of the CU's: two patch buffers of 448 rows x 64 B (double buffered across channel chunks) + a ring
// of 3 weight stages of 128 x 64 B = 80 KiB.  Two workgroups are then resident per CU, unsynchronised with each other:
// one's cold start / epilogue / fragment reads run under the other's MFMAs (the matrix pipe arbitrates by age, so two
// workgroups that start in phase drift apart).  One workgroup barrier per k-step: at the top of k-step s every wave has
// waited for its own pieces of W(s) (and of the chunk's patch), so past the barrier W(s) is visible to all and everyone
// is done reading the buffers of k-step s-1, which is where W(s+2) and the next patch piece are sent.
// The per-channel epilogue vectors do not fit beside the 80 KiB during the main loop; they are fetched into the (then
// free) staging area at the start of the epilogue, under the other workgroup's main loop.
constexpr int LS_PROWS = 448, LS_NSTW = 3;

template <int BM, int BN, int TM>
__global__ __launch_bounds__(256, 2) void k_conv_sw_ls(IgemmParams p) {
  constexpr int BK = SW_BK, NW = 4, THREADS = 256;
  constexpr int NWN = BN / 64;
  static_assert((BM / (32 * TM)) * NWN == NW, "4 waves");
  constexpr int ROWB = BK * 2;
  constexpr int W_BYTES = BN * ROWB;
  constexpr int PATCH_BYTES = LS_PROWS * ROWB;
  constexpr int WI = BN / 16 / NW;                  // weight LDS-DMA instructions per wave and k-step
  constexpr int PI = LS_PROWS / 16 / NW;            // patch instructions per wave and chunk: one at each of taps 0..PI-1
  static_assert(WI >= 1 && PI >= 1 && PI <= 8 && LS_PROWS == PI * 16 * NW, "tile shape");
  constexpr int STAGES_BYTES = 2 * PATCH_BYTES + LS_NSTW * W_BYTES;
  static_assert(STAGES_BYTES >= BM * BN * 2 + BM * 16 + IG_BIAS_LDS, "the epilogue tile, its tables and the vectors reuse the staging area");
  auto swz = [](int row) { return (row >> 2) & 3; };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const patch = smem;
  unsigned char* const wring = smem + 2 * PATCH_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / NWN, wn = wid - wm * NWN;

  const int tiles_n = p.N / BN;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (qd8 + 1) : r8 * (qd8 + 1) + (xcd - r8) * qd8) + loc;
  const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int Cin = p.Cin, Ktot = 9 * Cin, Wp = p.in.Wp;
  const int ncc = Cin / BK;

  const int q0 = sw_q(p.in, m0);
  const int qmax = (p.M / p.in.HoWo) * p.in.Hp * Wp - 1;
  unsigned poff32[PI], woff32[WI];
#pragma unroll
  for (int j = 0; j < PI; ++j) {
    const int row = (wid * PI + j) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz(row);
    int q = q0 + row;
    q = q < qmax ? q : qmax;
    poff32[j] = (unsigned)(((long long)q * p.in.cstride + p.in.coff + c * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const int row = (wid * WI + j) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz(row);
    woff32[j] = (unsigned)((((size_t)(n0 + row) * Ktot) + c * 8) * 2);
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Wt), 0, 0x7FFFFFFF, 0x00020000);

  auto stage_patch = [&](int cc, int j) {
    unsigned char* dst = patch + (cc & 1) * PATCH_BYTES + (wid * PI + j) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, (int)poff32[j],
                                             cc * (BK * 2), 0, 0);
  };
  auto stage_w = [&](int cc, int tap, int slot) {
    const int wsoff = (tap * Cin + cc * BK) * 2;
    unsigned char* dst = wring + slot * W_BYTES + wid * (WI * 1024);
#pragma unroll
    for (int j = 0; j < WI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16,
                                               (int)woff32[j], wsoff, 0, 0);
  };

  float16_ acc[2][TM];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  int arow[TM], w_off[2][2];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    int m = m0 + wm * (32 * TM) + t * 32 + frow;
    m = m < p.M ? m : p.M - 1;
    arow[t] = sw_q(p.in, m) - q0;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rw = wn * 64 + t * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) w_off[t][kk] = rw * ROWB + (((2 * kk + fhalf) ^ swz(rw)) << 4);
  }

  // ---- prologue: patch(0), W(0), W(1) requested
#pragma unroll
  for (int j = 0; j < PI; ++j) stage_patch(0, j);
  stage_w(0, 0, 0);
  stage_w(0, 1, 1);

  // k-step s = (chunk cc, tap T); weight slot s % 3 = T % 3.  vmcnt at the top (loads retire in order): this wave's pieces
  // of W(s) must have landed; younger and allowed in flight is what k-step s-1 sent: W(s+1) and its patch piece.
  auto kstep = [&](int cc, auto tap_c, auto last_c, auto first_c) {
    constexpr int T = decltype(tap_c)::value;
    constexpr bool LAST = decltype(last_c)::value;     // last chunk: no next patch, the weight prefetch runs dry
    constexpr bool FIRST = decltype(first_c)::value;   // k-step 0 of the tile: the prologue stands in for k-step -1
    constexpr int ky = T / 3, kx = T - 3 * ky;
    {
      // previous k-step: tap T-1 of this chunk, or tap 8 of the chunk before (never the last chunk, no patch piece at tap 8)
      constexpr bool prev_w = FIRST || T == 0 || !LAST || (T - 1) + 2 < 9;      // did k-step s-1 send W(s+1)?
      constexpr bool prev_p = !FIRST && T >= 1 && !LAST && (T - 1) < PI;        // ... and a patch piece?
      sw_wait_vm<(prev_w ? WI : 0) + (prev_p ? 1 : 0)>();
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LAST) {
      // the patch piece BEFORE the weights: waiting for W(s+2) two k-steps later then also covers this piece
      if constexpr (T < PI) stage_patch(cc + 1, T);
      if constexpr (T + 2 < 9) stage_w(cc, T + 2, (T + 2) % LS_NSTW);
      else stage_w(cc + 1, T - 7, (T + 2) % LS_NSTW);
    } else {
      if constexpr (T + 2 < 9) stage_w(cc, T + 2, (T + 2) % LS_NSTW);
    }
    const unsigned char* pb = patch + (cc & 1) * PATCH_BYTES;
    const unsigned char* wb = wring + (T % LS_NSTW) * W_BYTES;
    const int shift = ky * Wp + kx;
    half8 fa[2][TM], fw[2][2];
    int a0[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      int ar = arow[t];
      asm volatile("" : "+v"(ar));
      const int pr = ar + shift;
      a0[t] = (pr << 6) + ((fhalf ^ swz(pr)) << 4);
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) fa[0][t] = *reinterpret_cast<const half8*>(pb + a0[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t) fw[0][t] = *reinterpret_cast<const half8*>(wb + w_off[t][0]);
#pragma unroll
    for (int t = 0; t < TM; ++t) fa[1][t] = *reinterpret_cast<const half8*>(pb + (a0[t] ^ 32));
#pragma unroll
    for (int t = 0; t < 2; ++t) fw[1][t] = *reinterpret_cast<const half8*>(wb + w_off[t][1]);
    __builtin_amdgcn_sched_barrier(0);   // all twelve fragment reads are requested before the first MFMA
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk][i], fa[kk][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto chunk = [&](int cc, auto last_c, auto first_c) {
    kstep(cc, std::integral_constant<int, 0>{}, last_c, first_c);
    kstep(cc, std::integral_constant<int, 1>{}, last_c, std::false_type{});
    kstep(cc, std::integral_constant<int, 2>{}, last_c, std::false_type{});
    kstep(cc, std::integral_constant<int, 3>{}, last_c, std::false_type{});
    kstep(cc, std::integral_constant<int, 4>{}, last_c, std::false_type{});
    kstep(cc, std::integral_constant<int, 5>{}, last_c, std::false_type{});
    kstep(cc, std::integral_constant<int, 6>{}, last_c, std::false_type{});
    kstep(cc, std::integral_constant<int, 7>{}, last_c, std::false_type{});
    kstep(cc, std::integral_constant<int, 8>{}, last_c, std::false_type{});
  };
  chunk(0, std::false_type{}, std::true_type{});                 // ncc >= 2 (Cin % 64 == 0)
  for (int cc = 1; cc + 1 < ncc; ++cc) chunk(cc, std::false_type{}, std::false_type{});
  chunk(ncc - 1, std::true_type{}, std::false_type{});
  __syncthreads();                                               // every fragment read retired: the staging area is free
  float* bias_lds = reinterpret_cast<float*>(smem + BM * BN * 2 + BM * 16);
  ig_bias_to_lds(p, n0, bias_lds, wid, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // visible after the first barrier of the epilogue
  ig_epilogue<BM, BN, TM, THREADS, 0>(p, acc, smem, m0, n0, wm, wn, tid, lane, bias_lds);
}

template <int BM, int BN, int TM>
int sw_ls_launch(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDS = 2 * LS_PROWS * SW_BK * 2 + LS_NSTW * BN * SW_BK * 2;
  static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
  const long long tiles = (long long)fp_cdiv(p.M, BM) * (p.N / BN);
  FP_REQUIRE(tiles < (1ll << 31), "fp_igemm_f16_fwd: too many tiles");
  FP_SET_MAX_LDS((k_conv_sw_ls<BM, BN, TM>), LDS);
  hipLaunchKernelGGL((k_conv_sw_ls<BM, BN, TM>), dim3((unsigned)tiles), dim3(256), LDS, stream, p);
  FP_CHECK_LAUNCH("fp_igemm_f16_fwd(conv_sw_ls)");
  return FP_OK;
}

template <int BM, int BN, int TM>
int sw_launch(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDS = ig_lds_main<BM, BN>(2 * sw_prows(BM) * SW_BK * 2 + SW_NSTW * BN * SW_BK * 2) + IG_BIAS_LDS;
  static_assert(LDS <= 160 * 1024, "does not fit the 160 KiB LDS");
  const long long tiles = (long long)fp_cdiv(p.M, BM) * (p.N / BN);
  FP_REQUIRE(tiles < (1ll << 31), "fp_igemm_f16_fwd: too many tiles");
  FP_SET_MAX_LDS((k_conv_sw<BM, BN, TM>), LDS);
  hipLaunchKernelGGL((k_conv_sw<BM, BN, TM>), dim3((unsigned)tiles), dim3(512), LDS, stream, p);
  FP_CHECK_LAUNCH("fp_igemm_f16_fwd(conv_sw)");
  return FP_OK;
}

}  // namespace

// The patch of a tile is the run of padded pixels from tap (0,0) of its first output pixel to tap (2,2) of its last:
// BM - 1 + 2 per image-row crossing + 2 Wp + 2 per image crossing + 2 Wp + 3.  It has to fit the patch buffer.
static int sw_span(const IgemmGeom& g, int BM) {
  const int row_cross = (BM - 1) / g.Wo + 1, img_cross = (BM - 1) / g.HoWo + 1;
  return (BM - 1) + 2 * row_cross + (2 * g.Wp + 2) * img_cross + 2 * g.Wp + 3;
}

// which schedule: 1 = lock-step 256 x 128 tiles, two workgroups per CU (k_conv_sw_ls); 0 = ping-pong, one per CU
static int sw_variant(const IgemmParams& p) {
  int v = SW_DEFAULT_VARIANT;
#ifdef FP_PROFILE_BUILD
  static int forced = -2;
  if (forced == -2) {
    const char* e = getenv("FP_CONV_SW");          // profiling build only: ls | pp
    forced = e ? (!strcmp(e, "ls") ? 1 : (!strcmp(e, "pp") ? 0 : -1)) : -1;
  }
  if (forced >= 0) v = forced;
#endif
  // 2 / 3 (round 6 A/B builds): the two-per-CU lock-step kernel only where it measured ahead of the ping-pong kernel at sub-batch size
  // (DESIGN_HISTORY 3.2: 128 -> 128 +3-5 %, 256 -> 256 +2-3 %, 512 -> 512 -12 %): layers of up to 128 / 256 input channels
  if (v == 2) v = p.Cin <= 128 ? 1 : 0;
  if (v == 3) v = p.Cin <= 256 ? 1 : 0;
  if (v == 1 && sw_span(p.in, 256) > LS_PROWS) v = 0;
  return v;
}

// Round 4: 512 x 128 tiles for EVERY layer, also where N is a multiple of 256 (-DSW_FORCE_512=0 restores 256 x 256 there).  Per k-step a
// 256 x 256 tile stages 16 KB of weights + ~3.5 KB of patch for its 128 MFMAs, a 512 x 128 tile 8 KB + ~5 KB: a third fewer LDS-DMA
// pieces per MFMA, and the pieces' issue cost next to MFMAs is what the main loop is bound by; the activation rows are
// then fetched once per column tile (from L2 the second time).  Same k order per accumulator: the same bits.
#ifndef SW_FORCE_512
#define SW_FORCE_512 1
#endif
// (launches of fewer than two 512-row tiles keep the 256 x 256 shape where it exists, so that the smallest batches -- two hypotheses on
// the 20 x 20 layers -- stay on this kernel and on its summation order, as before the change)
static bool sw_use_256(const IgemmParams& p) { return (p.N % 256) == 0 && (!SW_FORCE_512 || p.M < 2 * 512); }
static int sw_tile_rows(const IgemmParams& p) { return sw_variant(p) == 1 ? 256 : (sw_use_256(p) ? 256 : 512); }

bool fp_conv3x3_sw_applicable(const IgemmParams& p) {
  const IgemmGeom& g = p.in;
  if (p.taps != 9 || g.stride != 1 || g.off != 0 || g.bsplit != 0) return false;
  if (g.HoWo % g.Wo != 0 || g.Wp != g.Wo + 2 || g.Hp != g.HoWo / g.Wo + 2) return false;
  const int BM = sw_tile_rows(p);
  if (p.M % g.HoWo != 0 || p.M < 2 * BM) return false;
  if (p.Cin % 64 != 0 || p.N % 128 != 0) return false;
  const long long bytes = (long long)(p.M / g.HoWo) * g.Hp * g.Wp * g.cstride * 2;
  if (bytes >= (1ll << 31)) return false;           // 32-bit byte offsets in the LDS-DMA source addresses
  return sw_span(g, BM) <= (sw_variant(p) == 1 ? LS_PROWS : sw_prows(BM));
}

int fp_conv3x3_sw_tile_rows(const IgemmParams& p) { return sw_tile_rows(p); }

int fp_conv3x3_sw_launch(const IgemmParams& p, hipStream_t stream) {
  if (sw_variant(p) == 1) return sw_ls_launch<256, 128, 4>(p, stream);
  if (sw_use_256(p)) return sw_launch<256, 256, 4>(p, stream);
  return sw_launch<512, 128, 4>(p, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// fp_pack_conv3x3_tiles_f16: one thread per 16-byte chunk of the packed matrix (include/fp_amd.h has the layout)
__global__ __launch_bounds__(256) void k_pack_conv3x3_tiles(const _Float16* __restrict__ w, _Float16* __restrict__ out, int N, int Cin) {
  const int nk = 9 * (Cin / 32);
  const long long total = (long long)N * nk * 4;          // chunks: N rows x nk k-steps x 4 chunks of 8 halves
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= total) return;
  const int pc = (int)(q & 3);
  const int r = (int)((q >> 2) & 127);
  const long long blk = q >> 9;                            // bn * nk + s
  const int s = (int)(blk % nk), bn = (int)(blk / nk);
  const int cc = s / 9, tap = s - 9 * cc;
  const int lc = pc ^ ((r >> 2) & 3);
  const half8 v = *reinterpret_cast<const half8*>(w + (size_t)(bn * 128 + r) * (9 * Cin) + tap * Cin + cc * 32 + lc * 8);
  *reinterpret_cast<half8*>(out + q * 8) = v;
}

extern "C" int fp_pack_conv3x3_tiles_f16(const void* w, void* w_tiles, int N, int Cin, void* stream) {
  FP_REQUIRE(w && w_tiles && w != w_tiles, "fp_pack_conv3x3_tiles_f16: NULL tensor or in-place repack");
  FP_REQUIRE(N > 0 && N % 128 == 0 && Cin > 0 && Cin % 32 == 0, "fp_pack_conv3x3_tiles_f16: N=%d must be a multiple of 128, Cin=%d of 32", N, Cin);
  FP_REQUIRE((((size_t)w | (size_t)w_tiles) & 15) == 0, "fp_pack_conv3x3_tiles_f16: tensors must be 16-byte aligned");
  const long long total = (long long)N * 9 * (Cin / 32) * 4;
  hipLaunchKernelGGL(k_pack_conv3x3_tiles, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)w,
                     (_Float16*)w_tiles, N, Cin);
  FP_CHECK_LAUNCH("fp_pack_conv3x3_tiles_f16");
  return FP_OK;
}
