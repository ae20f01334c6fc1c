// k_igemm_pp -- the "ping-pong" variant of the implicit-GEMM kernel (igemm.hip has the GEMM view, the operand layout and
// the lock-step variants; this file only differs in how the main loop is scheduled).
//
// Why: in the lock-step kernels all 8 waves of a workgroup pass through the same phases at the same time
// (barrier -> LDS-DMA issue -> ds_read_b128 -> MFMA), so the two waves that share a SIMD both issue memory
// instructions while the matrix pipe idles, then both want the pipe (measured: MFMA busy 49-59 %).
// Here the 8 waves form two groups of 4 (one wave per SIMD each; waves w and w+4 share a SIMD).  A k-step is split
// into a MEMORY cluster (fragment reads of this k-step into registers + LDS-DMA issue for the k-step three ahead)
// and a COMPUTE cluster (the MFMAs, on registers only); workgroup barriers separate the clusters and group 1 runs one
// cluster behind group 0, so at any time one wave per SIMD feeds the matrix pipe while its neighbour does memory work.
//
//   barrier interval:   0      1      2      3      4     ...
//   group 0:          mem 0  mma 0  mem 1  mma 1  mem 2
//   group 1:           --    mem 0  mma 0  mem 1  mma 1
//
// Staging: BK = 32 (64-byte LDS rows), 4 stages.  k-step h is read in intervals 2h (group 0) and 2h+1 (group 1); its
// buffer is refilled with k-step h+4 by the DMAs issued in mem h+1, i.e. after barrier 2h+1 (group 0) / 2h+2 (group 1),
// both later than every read of it (each wave drains lgkmcnt before the barrier that ends its memory cluster).  A wave
// leaves mem h only when its own pieces of k-step h+1 have landed (counted vmcnt: the two younger k-steps stay in
// flight), and two barriers separate that from the first read of k-step h+1 by either group.  So every DMA has two
// full k-step periods to land.
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include "igemm_common.h"
#include "igemm_epilogue.h"

// DBG != 0 builds timing-only variants that isolate one resource (results are wrong by construction; FP_IGEMM_DBG):
//   1 every tile stages the A rows of tile 0 (operand stream served by L2), 2 no LDS-DMA in the main loop,
//   3 operands addressed as if the activations were channel-blocked and the weights [k-step][N][32], 9 no epilogue.  Measured at the bench shapes (256->256 / 512->512 conv, TFLOP/s): normal 948 / 992, (1) 1010 / 991,
//   (2) 1272 / 1395, (9) 1185 / 1103 -- see DESIGN.md 3.2.
template <int BM, int BN, int TM, int DBG>
__global__ __launch_bounds__(512, 1) void k_igemm_pp(IgemmParams p) {
  constexpr int BK = 32, NST = 4, NW = 8, THREADS = 512;
  constexpr int NWN = BN / 64;
  static_assert((BM / (32 * TM)) * NWN == NW, "ping-pong needs exactly 8 waves");
  constexpr int ROWB = BK * 2;                     // 64-byte LDS rows
  constexpr int CPK = BK / 8;                      // 4 16-byte chunks per row
  constexpr int RPI = 1024 / ROWB;                 // 16 rows per LDS-DMA instruction
  constexpr int KK = BK / 16;                      // 2 MFMA k-substeps per stage
  constexpr int A_BYTES = BM * ROWB;
  constexpr int STAGE_BYTES = A_BYTES + BN * ROWB;
  constexpr int AI = BM / RPI / NW, WI = BN / RPI / NW;
  static_assert(AI >= 1 && WI >= 1, "tile too small");
  constexpr int LPS = AI + WI;                     // LDS-DMA instructions per wave and stage
  constexpr int LDS_MAIN = ig_lds_main<BM, BN>(NST * STAGE_BYTES);
  auto swz = [](int row) { return (row >> 2) & 3; };   // 4 rows share a 256-byte bank row (igemm.hip, BK = 32 case)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;                        // 0: leads, 1: one cluster behind
  const int wm = wid / NWN, wn = wid - wm * NWN;
  float* bias_lds = reinterpret_cast<float*>(smem + LDS_MAIN);

  // XCD-aware tile order (as igemm.hip)
  const int tiles_n = p.N / BN;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int q = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
  const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int Ktot = p.taps * p.Cin;
  ig_bias_to_lds(p, n0, bias_lds, wid, lane);

  unsigned aoff32[AI], woff32[WI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int row = wid * (AI * RPI) + j * RPI + lane / CPK;
    const int c = (lane % CPK) ^ swz(row);
    int m = (DBG == 1 ? 0 : m0) + row;
    m = m < p.M ? m : p.M - 1;
    aoff32[j] = (unsigned)((ig_row_off(p.in, m) + c * 8) * 2);
    if (DBG == 3) aoff32[j] = (unsigned)(m * 128 + (lane % CPK) * 16);   // timing only: channel-blocked image, pixels 128 B apart
  }
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const int row = wid * (WI * RPI) + j * RPI + lane / CPK;
    const int c = (lane % CPK) ^ swz(row);
    woff32[j] = (unsigned)((((size_t)(n0 + row) * Ktot) + c * 8) * 2);
    if (DBG == 3) woff32[j] = (unsigned)((n0 + row) * 64 + (lane % CPK) * 16);   // timing only: [k-step][N][32] weight image
  }
  const int nk = p.taps * (p.Cin / BK);
  int st_ci0 = 0, st_kx = 0, st_ky = 0, st_k = 0;
  const int inWp = p.in.Wp, inCs = p.in.cstride, Cin = p.Cin;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Wt), 0, 0x7FFFFFFF, 0x00020000);

  auto stage = [&](int buf) {
    int asoff = (((st_ky * inWp + st_kx) * inCs) + st_ci0) * 2;
    int wsoff = st_k * (BK * 2);
    if (DBG == 3) {   // the k-step selects a 64-channel plane (and a half of it) / a contiguous N x 32 weight slab
      asoff = ((st_k >> 1) % (Cin / 64)) * (p.M * 128) + (st_k & 1) * 64;
      wsoff = st_k * (p.N * 64);
    }
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
  auto wait_two_stages_in_flight = [&]() {
    static_assert(2 * LPS == 6 || 2 * LPS == 8 || 2 * LPS == 10, "add the counted wait");
    if (2 * LPS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (2 * LPS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  };

  float16_ acc[2][TM];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

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

  // ---- prologue: k-steps 0..2 in flight, k-step 0 landed and visible
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) stage(s);
  if (nk >= NST - 1) wait_two_stages_in_flight();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();           // group 1 sits out interval 0

  int buf = 0, nbuf = NST - 1;
  half8 fa[KK][TM], fw[KK][2];
  for (int ks = 0; ks < nk; ++ks) {
    // ---- memory cluster
    const unsigned char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int t = 0; t < TM; ++t) fa[kk][t] = *reinterpret_cast<const half8*>(sb + a_off[t][kk]);
#pragma unroll
      for (int t = 0; t < 2; ++t) fw[kk][t] = *reinterpret_cast<const half8*>(sb + w_off[t][kk]);
    }
    if (ks + NST - 1 < nk && DBG != 2) {
      stage(nbuf);
      wait_two_stages_in_flight();                 // own pieces of k-step ks+1 have landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads retired before the buffer can be refilled
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- compute cluster
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk][i], fa[kk][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    buf = (buf + 1 == NST) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == NST) ? 0 : nbuf + 1;
  }
  if (!grp) __builtin_amdgcn_s_barrier();          // group 0 waits out group 1's last compute cluster
  __syncthreads();
  if (DBG == 9) {                                  // timing without the epilogue (every accumulator stays live)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    if (sum == 12345.f) p.Y[tid] = (_Float16)1.f;
    return;
  }
  ig_epilogue<BM, BN, TM, THREADS, DBG>(p, acc, smem, m0, n0, wm, wn, tid, lane, bias_lds);
}

template <int BM, int BN, int TM, int DBG>
static int ig_pp_launch(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDS = ig_lds_main<BM, BN>(4 * (BM + BN) * 32 * 2) + IG_BIAS_LDS;
  static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  const long long tiles = (long long)fp_cdiv(p.M, BM) * (p.N / BN);
  FP_REQUIRE(tiles < (1ll << 31), "fp_igemm_f16_fwd: too many tiles");
  FP_SET_MAX_LDS((k_igemm_pp<BM, BN, TM, DBG>), LDS);
  hipLaunchKernelGGL((k_igemm_pp<BM, BN, TM, DBG>), dim3((unsigned)tiles), dim3(512), LDS, stream, p);
  FP_CHECK_LAUNCH("fp_igemm_f16_fwd");
  return FP_OK;
}


// variant: 0 = 256x256 (N % 256 == 0), otherwise 256x128.  The resource-isolation builds (DBG != 0: results wrong by
// construction) exist only in a profiling build (make PROFILE=1), selected there by FP_IGEMM_DBG.
int fp_igemm_pp_launch(const IgemmParams& p, int variant, hipStream_t stream) {
#ifdef FP_PROFILE_BUILD
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("FP_IGEMM_DBG"); dbg = e ? atoi(e) : 0; }
  if (variant == 0 && dbg) {
    switch (dbg) {
      case 1: return ig_pp_launch<256, 256, 4, 1>(p, stream);
      case 2: return ig_pp_launch<256, 256, 4, 2>(p, stream);
      case 3: return ig_pp_launch<256, 256, 4, 3>(p, stream);
      default: return ig_pp_launch<256, 256, 4, 9>(p, stream);
    }
  }
#endif
  if (variant == 0) return ig_pp_launch<256, 256, 4, 0>(p, stream);
  return ig_pp_launch<256, 128, 2, 0>(p, stream);
}
