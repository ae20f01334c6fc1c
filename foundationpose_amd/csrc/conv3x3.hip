// Shifted-window 3x3 convolution (stride 1, pad 1) for gfx950 -- the fast path behind fp_igemm_f16_fwd for the twelve
// ResnetBasicBlock convs of each encoder (network_modules.py:73-111) where input and output share one padded grid.
//
// Why a second kernel: profiling the generic implicit GEMM (profiles/README.md) showed it pinned at ~0.8 PFLOP/s for
// every tile shape and prefetch depth, with the LDS-DMA operand stream alone taking longer than the MFMAs alone: each
// workgroup re-fetches its activation rows once per tap, 9x, through the CU's vector memory path.  Here the GEMM rows
// are the pixels of the PADDED grid in memory order (border pixels are computed and discarded: +10 % work at 40x40,
// +21 % at 20x20), so the rows a 256-pixel tile needs for tap (ky,kx) are the tile's own rows shifted by
// ky*Wp + kx: ONE contiguous patch of 256 + 2*Wp + 2 pixels x 64 channels is staged per 64-channel chunk and all nine
// taps read it from LDS at a row offset.  Operand traffic through the memory path drops from (256+BN)*128 B to
// BN*128 B + 1/9 patch per k-step (2.3x less at BN=128, 1.7x at BN=256), and every DMA is lane-linear.
//
// Structure: 256 pixels x BN channels per workgroup, 8 waves; k order = channel chunk (outer) x tap (inner);
// patch double-buffered across chunks, weight tiles in a ring with a prefetch distance of NSTW-1 k-steps (counted
// vmcnt + raw s_barrier); fragments, swizzles, MFMA orientation and the transposing epilogue as in igemm.hip.
#include <stdlib.h>
#include "igemm_common.h"

#define CS_BM 256
#define CS_PATCH_ROWS 384                       // 48 LDS-DMA instructions of 8 rows: >= 256 + 2*63 + 2
#define CS_PATCH_BYTES (CS_PATCH_ROWS * 128)

template <int BN, int TM, int NSTW>
__global__ __launch_bounds__(512, 1) void k_conv3x3s1(IgemmParams p, int B) {
  constexpr int NWN = BN / 64;
  constexpr int NWM = CS_BM / (32 * TM);
  static_assert(NWN * NWM == 8, "8 waves");
  constexpr int W_BYTES = BN * IG_BK * 2;
  constexpr int WI = BN / 8 / 8;                 // weight-tile DMA instructions per wave and k-step
  constexpr int PI = CS_PATCH_ROWS / 8 / 8;      // patch DMA instructions per wave and channel chunk (6)
  constexpr int CPR = BN / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* patch = smem;                               // 2 x CS_PATCH_BYTES
  unsigned char* wring = smem + 2 * CS_PATCH_BYTES;          // NSTW x W_BYTES
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / NWN, wn = wid - wm * NWN;

  const int Hp = p.in.Hp, Wp = p.in.Wp;
  const long long Qtot = (long long)B * Hp * Wp;
  const int tiles_n = p.N / BN;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qq = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + loc;
  const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
  const long long q0 = (long long)bm * CS_BM;     // first padded pixel of the tile
  const int n0 = bn * BN;
  const int Ktot = 9 * p.Cin;
  const int ncc = p.Cin / IG_BK;
  const int nk = 9 * ncc;

  // ---- DMA sources.  Patch row r <-> padded pixel q0 - Wp - 1 + r (clamped into the buffer: clamped rows only feed
  //      outputs that are discarded).  Wave w issues patch instructions w, w+8, ... (8 rows each).
  const _Float16* psrc[PI];
#pragma unroll
  for (int j = 0; j < PI; ++j) {
    const int row = (wid + 8 * j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    long long gq = q0 - Wp - 1 + row;
    gq = gq < 0 ? 0 : (gq >= Qtot ? Qtot - 1 : gq);
    psrc[j] = p.A + gq * p.in.cstride + p.in.coff + c * 8;
  }
  const _Float16* wsrc[WI];
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const int row = wid * (WI * 8) + j * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    wsrc[j] = p.Wt + (size_t)(n0 + row) * Ktot + c * 8;
  }
  auto stage_patch = [&](int cc, int buf) {
    unsigned char* dst = patch + buf * CS_PATCH_BYTES;
#pragma unroll
    for (int j = 0; j < PI; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(psrc[j] + cc * IG_BK),
                                       (__attribute__((address_space(3))) void*)(dst + (wid + 8 * j) * 1024), 16, 0, 0);
  };
  auto stage_w = [&](int s, int slot) {   // k-step s = cc*9 + tap  ->  weight columns [tap*Cin + cc*64, +64)
    const int cc = s / 9, tap = s - cc * 9;
    const int woff = tap * p.Cin + cc * IG_BK;
    unsigned char* dst = wring + slot * W_BYTES + wid * (WI * 1024);
#pragma unroll
    for (int j = 0; j < WI; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + woff),
                                       (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
  };

  float16_ acc[2][TM];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  int a_row[TM], w_rowb[2], w_sw[2];
#pragma unroll
  for (int t = 0; t < TM; ++t) a_row[t] = wm * (32 * TM) + t * 32 + frow;     // + tap shift -> patch row
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rw = wn * 64 + t * 32 + frow;
    w_rowb[t] = rw * 128; w_sw[t] = (rw >> 1) & 7;
  }

  // ---- prologue: patch(0), W(0) .. W(NSTW-2)
  stage_patch(0, 0);
#pragma unroll
  for (int s = 0; s < NSTW - 1; ++s)
    if (s < nk) stage_w(s, s);
  int slot = 0, nslot = NSTW - 1;
  int cc = 0, tap = 0;
  for (int s = 0; s < nk; ++s) {
    // W(s) must have landed (and with it everything issued before it, i.e. this chunk's patch).  Younger loads that
    // may stay in flight: W(s+1..s+NSTW-2), plus patch(cc+1) when it was issued one step ago (tap == 1).
    if (s + NSTW - 2 < nk && NSTW > 2) {
      if (tap == 1 && cc + 1 < ncc) {
        if (WI * (NSTW - 2) + PI == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (WI * (NSTW - 2) + PI == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (WI * (NSTW - 2) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (WI * (NSTW - 2) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (tap == 0 && cc + 1 < ncc) stage_patch(cc + 1, (cc + 1) & 1);   // the other patch buffer was last read 9 steps ago
    if (s + NSTW - 1 < nk) stage_w(s + NSTW - 1, nslot);
    const unsigned char* sa = patch + (cc & 1) * CS_PATCH_BYTES;
    const unsigned char* sw = wring + slot * W_BYTES;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int shift = ky * Wp + kx;
    int a_rowb[TM], a_sw[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int pr = a_row[t] + shift;
      a_rowb[t] = pr * 128; a_sw[t] = (pr >> 1) & 7;
    }
    half8 fa[2][TM], fw[2][2];
    auto load_frags = [&](int kk, int sl) {
      const int c = 2 * kk + fhalf;
#pragma unroll
      for (int t = 0; t < TM; ++t) fa[sl][t] = *reinterpret_cast<const half8*>(sa + a_rowb[t] + ((c ^ a_sw[t]) << 4));
#pragma unroll
      for (int t = 0; t < 2; ++t) fw[sl][t] = *reinterpret_cast<const half8*>(sw + w_rowb[t] + ((c ^ w_sw[t]) << 4));
    };
    load_frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) load_frags(kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk & 1][i], fa[kk & 1][j], acc[i][j], 0, 0, 0);
    }
    slot = (slot + 1 == NSTW) ? 0 : slot + 1;
    nslot = (nslot + 1 == NSTW) ? 0 : nslot + 1;
    if (++tap == 9) { tap = 0; ++cc; }
  }
  __syncthreads();

  // ---- epilogue (as igemm.hip), rows = padded pixels; only interior pixels are stored
  unsigned char* E = smem;   // 256 rows x (2*BN) B
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = p.bias[n0 + nl + e];
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int ml = wm * (32 * TM) + j * 32 + (lane & 31);
        half4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (_Float16)(acc[i][j][g * 4 + e] + bv[e]);
        const int chunk = (nl >> 3) ^ (ml & 15);
        *reinterpret_cast<half4*>(E + ml * (2 * BN) + (chunk << 4) + ((nl & 4) << 1)) = v;
      }
    }
  }
  __syncthreads();
  const int Ho = Hp - 2, Wo = Wp - 2;
#pragma unroll
  for (int it = 0; it < (CS_BM * CPR) / 512; ++it) {
    const int qd = tid + it * 512;
    const int ml = qd / CPR, ch = qd % CPR;
    const long long q = q0 + ml;
    if (q >= Qtot) continue;
    const int b = (int)(q / (Hp * Wp));
    const int rem = (int)(q - (long long)b * (Hp * Wp));
    const int y = rem / Wp, x = rem - y * Wp;
    if (y < 1 || y > Ho || x < 1 || x > Wo) continue;       // border pixel of the padded grid: not an output
    const int m = (b * Ho + (y - 1)) * Wo + (x - 1);
    half8 v = *reinterpret_cast<const half8*>(E + ml * (2 * BN) + ((ch ^ (ml & 15)) << 4));
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

template <int BN, int TM, int NSTW>
static int cs_launch(const IgemmParams& p, int B, hipStream_t stream) {
  constexpr int STAGES = 2 * CS_PATCH_BYTES + NSTW * BN * IG_BK * 2;
  constexpr int ETILE = CS_BM * BN * 2;
  constexpr int LDS = STAGES > ETILE ? STAGES : ETILE;
  static_assert(LDS <= 160 * 1024, "does not fit the 160 KiB LDS");
  const long long Qtot = (long long)B * p.in.Hp * p.in.Wp;
  const long long tiles = ((Qtot + CS_BM - 1) / CS_BM) * (p.N / BN);
  FP_REQUIRE(tiles < (1ll << 31), "fp_igemm_f16_fwd: too many tiles");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3s1<BN, TM, NSTW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_conv3x3s1<BN, TM, NSTW>), dim3((unsigned)tiles), dim3(512), LDS, stream, p, B);
  FP_CHECK_LAUNCH("fp_igemm_f16_fwd(conv3x3)");
  return FP_OK;
}

int fp_conv3x3s1_launch(const IgemmParams& p, int B, hipStream_t stream) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("FP_CONV3X3_BN"); forced = e ? atoi(e) : 0; }
  const int bn = forced ? forced : ((p.N % 256) == 0 ? 256 : 128);
  if (bn == 256 && (p.N % 256) == 0) return cs_launch<256, 4, 2>(p, B, stream);
  return cs_launch<128, 2, 3>(p, B, stream);
}
