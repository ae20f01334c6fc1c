// k_igemm_pps -- persistent form of the ping-pong implicit-GEMM kernel (igemm_pp.hip: schedule of the main loop;
// igemm.hip: GEMM view and operand layout).  One workgroup per CU walks over its tiles (tile = blockIdx.x + i*gridDim.x,
// which keeps a workgroup's tiles on its own XCD), and the seams between tiles are closed:
//
//   * measured on the one-tile-per-workgroup kernel (256 -> 256 conv, 122 k cycles per tile): 8.6 k cycles pass between
//     kernel entry and the first MFMA -- the first operand stage of a tile comes from cold pages (address translation +
//     HBM under load) and with one workgroup per CU nothing else runs meanwhile;
//   * here the first three operand stages (and the bias) of tile i+1 are requested right after the main loop of tile i,
//     BEFORE its epilogue, so they land while the accumulators of tile i are written out;
//   * that needs the staging buffers to stay free during the epilogue: the accumulators go out through a 16 KiB slab
//     (32 rows x 256 channels) in eight steps instead of one 128 KiB tile image laid over the stages.
//
// LDS: 4 stages x 32 KiB | slab 16 KiB | row-offset tables 2 x 2 KiB | bias 2 x 1 KiB  = 150 KiB.
#include <hip/hip_fp16.h>
#include "igemm_common.h"
#include "igemm_epilogue.h"

namespace {

constexpr int PS_BM = 256, PS_BN = 256, PS_TM = 4, PS_BK = 32, PS_NST = 4, PS_THREADS = 512;
constexpr int PS_ROWB = PS_BK * 2;
constexpr int PS_A_BYTES = PS_BM * PS_ROWB;
constexpr int PS_STAGE_BYTES = (PS_BM + PS_BN) * PS_ROWB;            // 32 KiB
constexpr int PS_SLAB_OFF = PS_NST * PS_STAGE_BYTES;                 // 128 KiB
constexpr int PS_SLAB_BYTES = 32 * PS_BN * 2;                        // 16 KiB
constexpr int PS_ROWY_OFF = PS_SLAB_OFF + PS_SLAB_BYTES;
constexpr int PS_ROWR_OFF = PS_ROWY_OFF + PS_BM * 8;
constexpr int PS_BIAS_OFF = PS_ROWR_OFF + PS_BM * 8;
constexpr int PS_LDS = PS_BIAS_OFF + 2 * IG_BIAS_LDS;
static_assert(PS_LDS <= 160 * 1024, "LDS budget");

// Accumulators of one tile (the bias is already in them: they start from it) -> HBM through the slab.  Step s covers tile rows [32 s, 32 s + 32): the four waves that
// own them (wm == s / 4, pixel tile j == s % 4) write 32 x 256 halves, everybody stores 2 x 16 bytes per lane.
template <bool FULL>
__device__ __forceinline__ void ps_epilogue(const IgemmParams& p, float16_ (&acc)[2][PS_TM], unsigned char* smem, int m0, int n0,
                                            int wm, int wn, int tid, int lane) {
  constexpr int CPR = PS_BN / 8;                   // 32 chunks of 16 bytes per row
  unsigned char* slab = smem + PS_SLAB_OFF;
  long long* rowY = reinterpret_cast<long long*>(smem + PS_ROWY_OFF);
  long long* rowR = reinterpret_cast<long long*>(smem + PS_ROWR_OFF);
  if (tid < PS_BM) {
    const int m = m0 + tid;
    const bool in = m < p.M;
    rowY[tid] = in ? ig_row_off(p.out, m) + n0 : -1;
    if (p.R) rowR[tid] = in ? ig_row_off(p.res, m) + n0 : 0;
  }
  __syncthreads();
  // lane -> (row within a step, chunk): two per step
  const int ml0 = tid / CPR, ch = tid % CPR;       // rows ml0 and ml0 + 16
  half8 rv[8][2];
  if (p.R) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = 32 * s + ml0 + 16 * h;
        if (FULL || rowY[r] >= 0) rv[s][h] = *reinterpret_cast<const half8*>(p.R + rowR[r] + ch * 8);
      }
  }
  const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (wm == (s >> 2)) {
      const int j = s & 3;
      const int ml = lane & 31;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
          half4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (_Float16)acc[i][j][g * 4 + e];
          const int chunk = (nl >> 3) ^ (ml & 15);
          *reinterpret_cast<half4*>(slab + ml * (2 * PS_BN) + (chunk << 4) + ((nl & 4) << 1)) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ml = ml0 + 16 * h;
      const long long yo = rowY[32 * s + ml];
      if (!FULL && yo < 0) continue;
      half8 v = *reinterpret_cast<const half8*>(slab + ml * (2 * PS_BN) + ((ch ^ (ml & 15)) << 4));
      if (p.R) v = v + rv[s][h];
      if (p.relu) v = __builtin_elementwise_max(v, zero);
      *reinterpret_cast<half8*>(p.Y + yo + ch * 8) = v;
    }
    if (s < 7) __syncthreads();                    // the slab is rewritten by the next step
  }
}

__global__ __launch_bounds__(PS_THREADS, 1) void k_igemm_pps(IgemmParams p, int tiles) {
  constexpr int BM = PS_BM, BN = PS_BN, TM = PS_TM, BK = PS_BK, NST = PS_NST, NW = 8;
  constexpr int NWN = BN / 64;
  constexpr int ROWB = PS_ROWB, CPK = BK / 8, RPI = 1024 / ROWB, KK = BK / 16;
  constexpr int A_BYTES = PS_A_BYTES, STAGE_BYTES = PS_STAGE_BYTES;
  constexpr int AI = BM / RPI / NW, WI = BN / RPI / NW;
  constexpr int LPS = AI + WI;
  static_assert(2 * LPS == 8, "counted wait below assumes 4 LDS-DMA per wave and stage");
  auto swz = [](int row) { return (row >> 2) & 3; };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;
  const int wm = wid / NWN, wn = wid - wm * NWN;
  float* bias_lds = reinterpret_cast<float*>(smem + PS_BIAS_OFF);

  const int tiles_n = p.N / BN;
  const int q = tiles >> 3, r8 = tiles & 7;
  auto tile_origin = [&](int vb, int& m0, int& n0) {   // XCD-aware order over ALL tiles (igemm.hip), vb & 7 == blockIdx.x & 7
    const int xcd = vb & 7, loc = vb >> 3;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
    const int bm = tile / tiles_n;
    m0 = bm * BM;
    n0 = (tile - bm * tiles_n) * BN;
  };
  const int Ktot = p.taps * p.Cin;
  const int nk = p.taps * (p.Cin / BK);
  const int inWp = p.in.Wp, inCs = p.in.cstride, Cin = p.Cin;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Wt), 0, 0x7FFFFFFF, 0x00020000);

  unsigned aoff32[AI], woff32[WI];
  int st_ci0 = 0, st_kx = 0, st_ky = 0, st_k = 0;
  auto tile_setup = [&](int m0, int n0) {          // operand addressing of a tile; restarts the running k state
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const int row = wid * (AI * RPI) + j * RPI + lane / CPK;
      const int c = (lane % CPK) ^ swz(row);
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
    st_ci0 = 0; st_kx = 0; st_ky = 0; st_k = 0;
  };
  auto stage = [&](int buf) {
    const int asoff = (((st_ky * inWp + st_kx) * inCs) + st_ci0) * 2;
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
  auto request_tile = [&](int m0, int n0, int parity) {   // bias + the first NST-1 operand stages of a tile
    tile_setup(m0, n0);
    ig_bias_to_lds(p, n0, bias_lds + parity * (IG_BIAS_LDS / 4), wid, lane);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nk) stage(s);
  };

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

  int vb = blockIdx.x;
  int m0, n0;
  tile_origin(vb, m0, n0);
  request_tile(m0, n0, 0);
  for (int it = 0;; ++it) {
    // everything this wave has in flight (the tile's first stages, the previous tile's stores) is drained here: loads
    // and stores are not ordered with each other, so a counted wait is not enough at the seam
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the accumulators start from the bias (lane layout of D: channel = 64 wn + 32 i + 8 g + 4 (lane >> 5) + e), which
    // the request of this tile put into LDS; the epilogue then only converts and stores
    float16_ acc[2][TM];
    {
      typedef float float4_ __attribute__((ext_vector_type(4)));
      const float* bl = bias_lds + (it & 1) * (IG_BIAS_LDS / 4) + wn * 64 + 4 * (lane >> 5);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4_ b = *reinterpret_cast<const float4_*>(bl + i * 32 + 8 * g);
#pragma unroll
          for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][g * 4 + e] = b[e];
        }
    }
    if (grp) __builtin_amdgcn_s_barrier();         // group 1 sits out interval 0

    int buf = 0, nbuf = NST - 1;
    half8 fa[KK][TM], fw[KK][2];
    for (int ks = 0; ks < nk; ++ks) {
      const unsigned char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[kk][t] = *reinterpret_cast<const half8*>(sb + a_off[t][kk]);
#pragma unroll
        for (int t = 0; t < 2; ++t) fw[kk][t] = *reinterpret_cast<const half8*>(sb + w_off[t][kk]);
      }
      if (ks + NST - 1 < nk) {
        stage(nbuf);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // own pieces of k-step ks+1 have landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
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
    if (!grp) __builtin_amdgcn_s_barrier();        // group 0 waits out group 1's last compute cluster
    __syncthreads();                               // every fragment read of this tile is done: the stages are free

    const int vbn = vb + gridDim.x;
    const bool has_next = vbn < tiles;
    int m0n = 0, n0n = 0;
    if (has_next) {
      tile_origin(vbn, m0n, n0n);
      request_tile(m0n, n0n, (it + 1) & 1);        // lands during the epilogue below
    }
    if (m0 + BM <= p.M) ps_epilogue<true>(p, acc, smem, m0, n0, wm, wn, tid, lane);
    else ps_epilogue<false>(p, acc, smem, m0, n0, wm, wn, tid, lane);
    if (!has_next) break;
    vb = vbn; m0 = m0n; n0 = n0n;
  }
}

}  // namespace

int fp_igemm_pps_launch(const IgemmParams& p, hipStream_t stream) {
  const long long tiles = (long long)fp_cdiv(p.M, PS_BM) * (p.N / PS_BN);
  FP_REQUIRE(tiles < (1ll << 31), "fp_igemm_f16_fwd: too many tiles");
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cus &= ~7;                                     // a multiple of the 8 XCDs keeps a workgroup's tiles on one XCD
    if (cus < 8) cus = 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_igemm_pps), hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
  }
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  hipLaunchKernelGGL(k_igemm_pps, dim3(grid), dim3(PS_THREADS), PS_LDS, stream, p, (int)tiles);
  FP_CHECK_LAUNCH("fp_igemm_f16_fwd");
  return FP_OK;
}
