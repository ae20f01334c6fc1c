// fp_linear_layernorm_fwd -- a 512-wide Linear of nn.TransformerEncoderLayer (refine_network.py:56-70: self_attn.out_proj,
// linear2) fused with the residual add and the post-norm LayerNorm that consume it:
//     branch = f16(x16 @ W^T + b)                    (nn.Linear under autocast: fp32 accumulate + bias, one rounding)
//     z      = resid + f32(branch)                   (fp32 residual stream; resid = x32, or f32(tok16) + pe[row % S])
//     y      = LN(z) * gamma + beta  -> y32 and / or y16
// = fp_igemm_f16_fwd (taps = 1) followed by fp_layernorm_res_fwd, without the (M, 512) branch tensor ever reaching HBM and
// without the second launch.  A workgroup owns 128 complete rows (tile 128 x 512, so the LayerNorm statistics of a row stay
// inside the workgroup): 8 waves, wave w computes channels [64 w, 64 w + 64) of all 128 rows as 4 x 2
// v_mfma_f32_32x32x16_f16 tiles (128 accumulator registers); main loop = the lock-step schedule of k_igemm_f16 (igemm.hip)
// at BK = 32 with 3 LDS stages (operands HBM -> LDS by LDS-DMA, XOR-swizzled 64-byte rows, counted vmcnt + one barrier per
// k-step); the epilogue parks f16(acc + bias) in the swizzled LDS tile of igemm_epilogue.h (128 rows x 1 KiB) and then runs
// the row code of k_layernorm_res512 (rowops_ln.h, the same source) on those rows, 16 rows per wave, four at a time.
// Per element the same instruction sequence as the two-kernel path it replaces.
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "igemm_common.h"
#include "rowops_ln.h"

namespace {

// Tile: BM rows x 512 channels, 8 waves (wave w: channels 64 w .. 64 w + 63 of all rows as (BM / 32) x 2 MFMA tiles), NST LDS stages
// of BK = 32.  Product: <128, 3> (one workgroup per CU).  The profiling build also has <64, 2>: half the rows, 74 KiB of LDS, so
// that two workgroups share a CU and one's HBM-bound LayerNorm tail runs under the other's main loop (FP_LL_TILE=64).
constexpr int LL_BN = 512, LL_NW = 8, LL_THREADS = LL_NW * 64, LL_BK = 32;
constexpr int LL_ROWB = LL_BK * 2;                       // bytes per LDS row (one input row / one output channel, BK halves)
constexpr int LL_KK = LL_BK / 16;                        // MFMA k-substeps per stage
constexpr int LL_W_BYTES = LL_BN * LL_ROWB;              // 32 KiB
constexpr int LL_WI = 4;                                 // W-tile LDS-DMA instructions per wave and stage (A tile: one)
template <int BM, int NST>
struct LlTile {
  static constexpr int TM = BM / 32;
  static constexpr int A_BYTES = BM * LL_ROWB;
  static constexpr int STAGE = A_BYTES + LL_W_BYTES;
  static constexpr int E_BYTES = BM * LL_BN * 2;         // the epilogue tile, laid over the (finished) staging buffers
  static constexpr int MAIN = NST * STAGE > E_BYTES ? NST * STAGE : E_BYTES;
  static constexpr int LDS = MAIN + LL_BN * 4;           // + the bias vector
  static constexpr int ROWS_PER_WAVE = BM / LL_NW;
  static constexpr int WG_PER_CU = LDS <= 80 * 1024 ? 2 : 1;
  static constexpr int R = WG_PER_CU == 2 ? 2 : 4;       // rows a wave normalises together (interleaved reduction chains); two
                                                         // where the workgroup has 128 registers per lane
  static_assert(BM % 32 == 0 && (BM / 16) <= LL_NW && LL_NW % (BM / 16) == 0, "A tile: 16 rows per LDS-DMA instruction, one per wave");
  static_assert(ROWS_PER_WAVE % R == 0, "rows per wave must be a multiple of the LayerNorm group");
  static_assert(NST == 2 || NST == 3, "counted waits are written for 2 or 3 stages");
  static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
};

struct LinearLnParams {
  const _Float16* X;      // (M, K)
  const _Float16* Wt;     // (512, K)
  const float* bias;      // (512) or null
  const float* x32;       // residual stream (M, 512) f32, or null
  const _Float16* tok16;  // ... or tokens (M, 512) f16 + pe
  const float* pe;
  int S;
  const float* gamma;
  const float* beta;
  float eps;
  float* y32;             // either may be null
  _Float16* y16;
  int M, K;
  float* part;            // MEAN forms: chunk sums [M / 16][512] (16 consecutive rows each); S = rows per group
  const _Float16* W2;     // FFN variant (profiling build): second Linear (512, 512) and its bias; Wt / bias are the first (+ ReLU)
  const float* bias2;
};

__device__ __forceinline__ int ll_swz(int row) { return (row >> 2) & 3; }   // chunk swizzle of a 64-byte row (4 rows per bank row)

// the LDS-DMA lives in a plain function: written inline in a kernel TEMPLATE, the address_space(3) cast + builtin made hipcc 7.2
// drop the kernel's host stub without a diagnostic
__device__ __forceinline__ void ll_dma16(const __amdgpu_buffer_rsrc_t& rs, void* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// MEAN (fp_ffn_layernorm_mean_fwd; profiling build: fp_linear_layernorm_mean_fwd): instead of writing the normalised rows, add them up
// per group of p.S rows (the token mean of refine_network.py:90-91 fused with norm2).  The order of that sum must not depend on WHERE
// in the batch a hypothesis sits (sub-batches, shards and single batches have to agree bit for bit): a wave
// therefore owns 16 CONSECUTIVE rows of the tile -- a chunk that lies inside one hypothesis because p.S and the tile height are
// multiples of 16 -- adds them in row order and writes the chunk sum; k_ln_mean_finish adds a hypothesis's S / 16 chunk sums in
// chunk order.
// FFN (profiling build only, fp_ffn_layernorm_mean_fwd): TWO Linears back to back on the tile, linear1 + ReLU -> the 128 x 512
// intermediate parked in the epilogue tile -> linear2 reading its A fragments from that tile and its weights straight from L2 into
// registers (a wave owns 64 output channels, so nobody shares its weight rows: no staging, no barrier in the second loop).
template <int BM, int NST, bool MEAN = false, bool FFN = false>
__global__ __launch_bounds__(LL_THREADS, (LlTile<BM, NST>::WG_PER_CU == 2 ? 4 : 1)) void k_linear_ln512(LinearLnParams p) {
  using T = LlTile<BM, NST>;
  constexpr int LL_BM = BM, LL_NST = NST, LL_TM = T::TM, LL_A_BYTES = T::A_BYTES, LL_STAGE = T::STAGE, LL_MAIN = T::MAIN;
  constexpr int LL_ROWS_PER_WAVE = T::ROWS_PER_WAVE, LL_R = T::R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // = channel group of 64
  float* bias_lds = reinterpret_cast<float*>(smem + LL_MAIN);
  const int m0 = blockIdx.x * LL_BM;

  // bias -> LDS (512 floats: waves 0 and 1 fetch 1 KiB each with one LDS-DMA; the oldest vector-memory operation of the wave,
  // so every later counted wait covers it; visible to the workgroup after the first barrier of the main loop)
  if (wid < 2) {
    float* dst = bias_lds + wid * 256;
    if (p.bias) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, LL_BN * 4, 0x00020000);
      ll_dma16(rs, dst, lane * 16, wid * 1024);
    } else {
      *reinterpret_cast<float4_*>(dst + lane * 4) = float4_{0.f, 0.f, 0.f, 0.f};
    }
  }

  // per-thread staging sources (byte offsets from the tensor bases): wave w loads rows [16 w, +16) of the A tile and rows
  // [64 w, +64) of W; an LDS-DMA instruction writes 1 KiB lane-linear = 16 rows of 64 B, so lane l carries row l / 4 and the
  // LOGICAL chunk that belongs in physical chunk l % 4 of that row.  A tile shorter than 128 rows: waves w and w + BM / 16 carry
  // the same 16 rows to the same place, so that every wave issues the same number of loads per stage (one counted wait for all)
  constexpr int A_GROUPS = LL_BM / 16;
  const int awid = wid % A_GROUPS;
  unsigned aoff32, woff32[LL_WI];
  {
    const int row = awid * 16 + lane / 4;
    const int c = (lane % 4) ^ ll_swz(row);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    aoff32 = (unsigned)(((size_t)m * p.K + c * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < LL_WI; ++j) {
    const int row = wid * 64 + j * 16 + lane / 4;
    const int c = (lane % 4) ^ ll_swz(row);
    woff32[j] = (unsigned)(((size_t)row * p.K + c * 8) * 2);
  }
  const int nk = p.K / LL_BK;
  int st_k = 0;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.X), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Wt), 0, 0x7FFFFFFF, 0x00020000);
  auto stage = [&](int buf) {
    const int soff = st_k * (LL_BK * 2);
    unsigned char* sa = smem + buf * LL_STAGE + awid * 1024;
    unsigned char* sw = smem + buf * LL_STAGE + LL_A_BYTES + wid * (LL_WI * 1024);
    ll_dma16(rsA, sa, (int)aoff32, soff);
#pragma unroll
    for (int j = 0; j < LL_WI; ++j) ll_dma16(rsW, sw + j * 1024, (int)woff32[j], soff);
    ++st_k;
  };

  float16_ acc[2][LL_TM];   // [channel tile i][row tile j]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < LL_TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read addressing: lane reads row (lane & 31) of a 32-row tile, logical chunk 2 kk + (lane >> 5)
  const int frow = lane & 31, fhalf = lane >> 5;
  int a_off[LL_TM][LL_KK], w_off[2][LL_KK];
#pragma unroll
  for (int t = 0; t < LL_TM; ++t) {
    const int ra = t * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < LL_KK; ++kk) a_off[t][kk] = ra * LL_ROWB + (((2 * kk + fhalf) ^ ll_swz(ra)) << 4);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rw = wid * 64 + t * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < LL_KK; ++kk) w_off[t][kk] = LL_A_BYTES + rw * LL_ROWB + (((2 * kk + fhalf) ^ ll_swz(rw)) << 4);
  }

#pragma unroll
  for (int s = 0; s < LL_NST - 1; ++s)
    if (s < nk) stage(s);
  int buf = 0, nbuf = LL_NST - 1;
  for (int ks = 0; ks < nk; ++ks) {
    // stage ks must have landed; with three stages the one issued after it (1 + LL_WI loads per wave) may stay in flight
    if (LL_NST > 2 && ks + LL_NST - 2 < nk) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // everyone's part of stage ks is visible; everyone is done reading stage ks-1
    if (ks + LL_NST - 1 < nk) stage(nbuf);
    const unsigned char* sb = smem + buf * LL_STAGE;
    half8 fa[2][LL_TM], fw[2][2];
    auto load_frags = [&](int kk, int slot) {
#pragma unroll
      for (int t = 0; t < LL_TM; ++t) fa[slot][t] = *reinterpret_cast<const half8*>(sb + a_off[t][kk]);
#pragma unroll
      for (int t = 0; t < 2; ++t) fw[slot][t] = *reinterpret_cast<const half8*>(sb + w_off[t][kk]);
    };
    load_frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < LL_KK; ++kk) {
      if (kk < LL_KK - 1) load_frags(kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above the MFMAs
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < LL_TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk & 1][i], fa[kk & 1][j], acc[i][j], 0, 0, 0);
    }
    buf = (buf + 1 == LL_NST) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == LL_NST) ? 0 : nbuf + 1;
  }
  __syncthreads();   // all fragment reads done before the staging buffers become the epilogue tile

  // residual rows (fp32 stream, or fp16 tokens + positional table) of LL_R rows of this wave, rows wid + 8 (t0 + u); a row
  // past the end of the matrix reads the last row instead: normalised, never stored
  half8 tk[LL_R];
  float rs[LL_R][8];
  // tile row of the wave's (t0 + u)-th row: interleaved (wave w: rows w, w + 8, ...) or, for the MEAN forms, one contiguous chunk
  auto tile_row = [&](int k) { return MEAN ? wid * LL_ROWS_PER_WAVE + k : wid + LL_NW * k; };
  auto request_resid = [&](int t0) {
#pragma unroll
    for (int u = 0; u < LL_R; ++u) {
      const int m = m0 + tile_row(t0 + u);
      const int mc = m < p.M ? m : p.M - 1;
      if (p.x32) {
        load8f(p.x32 + (size_t)mc * 512 + lane * 8, rs[u]);
      } else {
        tk[u] = *reinterpret_cast<const half8*>(p.tok16 + (size_t)mc * 512 + lane * 8);
        load8f(p.pe + (size_t)((unsigned)mc % (unsigned)p.S) * 512 + lane * 8, rs[u]);
      }
    }
  };
  // requested before the accumulators are parked (in flight under the transposition) where the registers allow it: the two-per-CU
  // tile has 128 registers per lane, and its other workgroup covers the latency anyway
  constexpr bool EARLY_RESID = T::WG_PER_CU == 1 && !FFN;   // FFN: the second loop needs the registers
  if constexpr (EARLY_RESID) request_resid(0);

  // ---- epilogue 1: f16(acc + bias) -> E[row][channel], rows of 1 KiB, the low 4 bits of the 16-byte chunk index XORed with
  // (row & 15) (igemm_epilogue.h).  D[i = channel][j = row]: a lane holds row (lane & 31) of a row tile and channels
  // 8 g + 4 (lane >> 5) + {0..3} of a channel tile, g = register >> 2
  unsigned char* E = smem;
  if constexpr (FFN) {
    // ---- linear1 done: H = relu(f16(acc + b1)) -> E, then linear2 out of E
    {
    #pragma unroll
      for (int i = 0; i < 2; ++i) {
    #pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wid * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
          const float4_ bv = *reinterpret_cast<const float4_*>(bias_lds + nl);
    #pragma unroll
          for (int j = 0; j < LL_TM; ++j) {
            const int ml = j * 32 + (lane & 31);
            half4 v;
    #pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (_Float16)(acc[i][j][g * 4 + e] + bv[e]);
            v = __builtin_elementwise_max(v, half4{0, 0, 0, 0});                   // ReLU of linear1
            const int chunk = (nl >> 3) ^ (ml & 15);
            *reinterpret_cast<half4*>(E + ml * (2 * LL_BN) + (chunk << 4) + ((nl & 4) << 1)) = v;
          }
        }
      }
    }
    __syncthreads();                 // H complete; nobody reads b1 any more
    if (wid < 2) {                   // b2 -> the bias area (visible after the barrier that ends the second loop)
      float* dst = bias_lds + wid * 256;
      if (p.bias2) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias2), 0, LL_BN * 4, 0x00020000);
        ll_dma16(rs, dst, lane * 16, wid * 1024);
      } else {
        *reinterpret_cast<float4_*>(dst + lane * 4) = float4_{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < LL_TM; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // W2 fragment of k16-step q for channel tile i: 8 consecutive k of row (output channel) 64 wid + 32 i + (lane & 31),
    // starting at 16 q + 8 (lane >> 5): one 16-byte load per lane, RING steps ahead
    constexpr int NQ = LL_BN / 16, RING = 4;
    const _Float16* w2p = p.W2 + (size_t)(wid * 64 + frow) * LL_BN + fhalf * 8;
    half8 wr[RING][2];
    auto wload = [&](int q, int slot) {
#pragma unroll
      for (int i = 0; i < 2; ++i) wr[slot][i] = *reinterpret_cast<const half8*>(w2p + (size_t)i * 32 * LL_BN + q * 16);
    };
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) wload(q, q);
    // A fragment of k16-step q, row tile j: row 32 j + (lane & 31) of E, logical chunk 2 q + (lane >> 5)
    const unsigned char* erow = E + frow * (2 * LL_BN);
    const int esw = frow & 15;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q + RING - 1 < NQ) wload(q + RING - 1, (q + RING - 1) % RING);
      half8 fa2[LL_TM];
#pragma unroll
      for (int j = 0; j < LL_TM; ++j)
        fa2[j] = *reinterpret_cast<const half8*>(erow + j * 32 * (2 * LL_BN) + (((2 * q + fhalf) ^ esw) << 4));
      __builtin_amdgcn_sched_barrier(0);   // keep the weight loads RING - 1 steps ahead (hipcc otherwise sinks them to their use)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < LL_TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[q % RING][i], fa2[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // b2 has landed (waves 0, 1)
    __syncthreads();                 // every wave is done reading H before linear2's output takes its place
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = wid * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
      const float4_ bv = *reinterpret_cast<const float4_*>(bias_lds + nl);
#pragma unroll
      for (int j = 0; j < LL_TM; ++j) {
        const int ml = j * 32 + (lane & 31);
        half4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (_Float16)(acc[i][j][g * 4 + e] + bv[e]);
        const int chunk = (nl >> 3) ^ (ml & 15);
        *reinterpret_cast<half4*>(E + ml * (2 * LL_BN) + (chunk << 4) + ((nl & 4) << 1)) = v;
      }
    }
  }
  __syncthreads();
  if constexpr (!EARLY_RESID) request_resid(0);

  // ---- epilogue 2: k_layernorm_res512's row code on the tile's rows (wave w: rows w, w + 8, ...), LL_R rows at a time.
  // The residual rows of the NEXT group are requested before the current group is normalised (and those of the first group
  // before the accumulators are parked, see above), so a wave waits for HBM once, not four times; row numbers are wave-uniform:
  // the positional-table row costs a scalar 32-bit modulo, where resid_row's 64-bit one would be a division loop per row.
  float gm[8], bt[8];
  if constexpr (!MEAN) {
    load8f(p.gamma + lane * 8, gm);
    load8f(p.beta + lane * 8, bt);
  }
  float csum[8];           // MEAN: the sum of this wave's chunk of LL_ROWS_PER_WAVE consecutive rows, in row order
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
#pragma unroll 1
  for (int t0 = 0; t0 < LL_ROWS_PER_WAVE; t0 += LL_R) {
    float f[LL_R][8];
#pragma unroll
    for (int u = 0; u < LL_R; ++u) {
      const int r = tile_row(t0 + u);
      const half8 b = *reinterpret_cast<const half8*>(E + r * (2 * LL_BN) + ((lane ^ (r & 15)) << 4));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        f[u][e] = p.x32 ? rs[u][e] : (float)tk[u][e] + rs[u][e];     // resid_row (rowops_ln.h)
        f[u][e] += (float)b[e];
      }
    }
    if (t0 + LL_R < LL_ROWS_PER_WAVE) request_resid(t0 + LL_R);
    ln_rows<LL_R>(p.eps, f);
    if constexpr (MEAN) {
#pragma unroll
      for (int u = 0; u < LL_R; ++u) {
        const int m = m0 + tile_row(t0 + u);
        if (m >= p.M) continue;                            // wave-uniform
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += f[u][e];
      }
      continue;
    }
#pragma unroll
    for (int u = 0; u < LL_R; ++u) {
      const int m = m0 + tile_row(t0 + u);
      if (m >= p.M) continue;                              // wave-uniform
      half8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) { f[u][e] = fmaf(f[u][e], gm[e], bt[e]); h[e] = (_Float16)f[u][e]; }
      if (p.y32) store8f(p.y32 + (size_t)m * 512 + lane * 8, f[u]);
      if (p.y16) *reinterpret_cast<half8*>(p.y16 + (size_t)m * 512 + lane * 8) = h;
    }
  }
  if constexpr (MEAN) {
    static_assert(!MEAN || LL_ROWS_PER_WAVE == 16, "a wave's chunk is 16 rows: groups are multiples of 16 rows");
    // chunk q = (m0 + 16 wid) / 16 of the whole matrix: its sum goes to part[q][512]; chunks past the end do not exist
    const int mrow = m0 + wid * LL_ROWS_PER_WAVE;
    if (mrow < p.M) store8f(p.part + (size_t)(mrow / LL_ROWS_PER_WAVE) * 512 + lane * 8, csum);
  }
}

// out[g][c] = mean over the rows of group g of LN(...) * gamma + beta from the chunk sums (16 rows each), chunks in increasing order
__global__ __launch_bounds__(512) void k_ln_mean_finish(const float* __restrict__ part, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out, int S) {
  const int g = blockIdx.x, c = threadIdx.x;
  const int nch = S / 16;
  const float* src = part + (size_t)g * nch * 512 + c;
  float a = 0.f;
  for (int k = 0; k < nch; ++k) a += src[(size_t)k * 512];
  a *= 1.0f / (float)S;
  out[(size_t)g * 512 + c] = fmaf(a, gamma[c], beta[c]);   // mean(LN(x) * gamma + beta) = mean(LN(x)) * gamma + beta
}

template <int BM, int NST>
int ll_launch(const LinearLnParams& p, hipStream_t stream) {
  constexpr int LDS = LlTile<BM, NST>::LDS;
  FP_SET_MAX_LDS((k_linear_ln512<BM, NST>), LDS);
  hipLaunchKernelGGL((k_linear_ln512<BM, NST>), dim3(fp_cdiv(p.M, BM)), dim3(LL_THREADS), LDS, stream, p);
  FP_CHECK_LAUNCH("fp_linear_layernorm_fwd");
  return FP_OK;
}

}  // namespace

#ifdef FP_PROFILE_BUILD
// Profiling build only (not in include/fp_amd.h, bound ad hoc by scripts/bench_linear_ln_mean.py): linear2 + residual + norm2 +
// token mean of the refiner's encoder layer in one launch + a finish kernel; = fp_igemm_f16_fwd + fp_colmean_f16_fwd with another
// (fixed) summation order of the token mean, so it has to pass the parity gates before it can replace them.
// out (groups, 512) f32; workspace: groups * rows_per_group / 16 * 512 floats.
extern "C" int fp_linear_layernorm_mean_fwd(const void* x16, const void* w16, const float* bias, const float* x32, const float* gamma,
                                            const float* beta, float eps, float* out, float* workspace, size_t workspace_bytes,
                                            int groups, int rows_per_group, int K, int D, void* stream) {
  FP_REQUIRE(groups >= 0, "fp_linear_layernorm_mean_fwd: groups < 0");
  if (groups == 0) return FP_OK;
  FP_REQUIRE(x16 && w16 && x32 && gamma && beta && out && workspace, "fp_linear_layernorm_mean_fwd: NULL tensor");
  FP_REQUIRE(D == 512 && K > 0 && K % LL_BK == 0, "fp_linear_layernorm_mean_fwd: D must be 512, K a multiple of %d", LL_BK);
  FP_REQUIRE(rows_per_group >= 16 && rows_per_group % 16 == 0, "fp_linear_layernorm_mean_fwd: rows_per_group must be a multiple of 16");
  const long long M = (long long)groups * rows_per_group;
  FP_REQUIRE(M * K < (1ll << 30), "fp_linear_layernorm_mean_fwd: operands exceed 2 GiB");
  const int tiles = fp_cdiv((int)M, 128);
  FP_REQUIRE(workspace_bytes >= (size_t)(M / 16) * 512 * sizeof(float), "fp_linear_layernorm_mean_fwd: workspace too small");
  LinearLnParams p;
  p.X = (const _Float16*)x16; p.Wt = (const _Float16*)w16; p.bias = bias; p.x32 = x32; p.tok16 = nullptr; p.pe = nullptr;
  p.S = rows_per_group; p.gamma = gamma; p.beta = beta; p.eps = eps; p.y32 = nullptr; p.y16 = nullptr; p.M = (int)M; p.K = K;
  p.part = workspace; p.W2 = nullptr; p.bias2 = nullptr;
  constexpr int LDS = LlTile<128, 3>::LDS;
  FP_SET_MAX_LDS((k_linear_ln512<128, 3, true>), LDS);
  hipLaunchKernelGGL((k_linear_ln512<128, 3, true>), dim3(tiles), dim3(LL_THREADS), LDS, (hipStream_t)stream, p);
  hipLaunchKernelGGL(k_ln_mean_finish, dim3(groups), dim3(512), 0, (hipStream_t)stream, (const float*)workspace, gamma, beta, out,
                     rows_per_group);
  FP_CHECK_LAUNCH("fp_linear_layernorm_mean_fwd");
  return FP_OK;
}
#endif

// The whole feed-forward half of the refiner's encoder layer in one launch + the finish kernel -- linear1 + ReLU + linear2 +
// residual + norm2 + token mean (refine_network.py:56-70, :90-91); = two fp_igemm_f16_fwd + fp_colmean_f16_fwd with the (M, 512)
// intermediates staying in LDS.  Both Linears are 512 -> 512.  include/fp_amd.h.
extern "C" int fp_ffn_layernorm_mean_fwd(const void* y16, const void* w1, const float* b1, const void* w2, const float* b2,
                                         const float* x32, const float* gamma, const float* beta, float eps, float* out,
                                         float* workspace, size_t workspace_bytes, int groups, int rows_per_group, void* stream) {
  FP_REQUIRE(groups >= 0, "fp_ffn_layernorm_mean_fwd: groups < 0");
  if (groups == 0) return FP_OK;
  FP_REQUIRE(y16 && w1 && w2 && x32 && gamma && beta && out && workspace, "fp_ffn_layernorm_mean_fwd: NULL tensor");
  FP_REQUIRE(rows_per_group >= 16 && rows_per_group % 16 == 0,
             "fp_ffn_layernorm_mean_fwd: rows_per_group=%d must be a multiple of 16 (a wave sums 16 consecutive rows, which have to belong to one group)", rows_per_group);
  const long long M = (long long)groups * rows_per_group;
  FP_REQUIRE(M * 512 < (1ll << 30), "fp_ffn_layernorm_mean_fwd: operands exceed 2 GiB");
  FP_REQUIRE((((size_t)y16 | (size_t)w1 | (size_t)w2 | (size_t)b1 | (size_t)b2 | (size_t)x32 | (size_t)gamma | (size_t)beta) & 15) == 0,
             "fp_ffn_layernorm_mean_fwd: tensors must be 16-byte aligned");
  const int tiles = fp_cdiv((int)M, 128);
  FP_REQUIRE(workspace_bytes >= (size_t)(M / 16) * 512 * sizeof(float), "fp_ffn_layernorm_mean_fwd: workspace too small");
  LinearLnParams p;
  p.X = (const _Float16*)y16; p.Wt = (const _Float16*)w1; p.bias = b1; p.x32 = x32; p.tok16 = nullptr; p.pe = nullptr;
  p.S = rows_per_group; p.gamma = gamma; p.beta = beta; p.eps = eps; p.y32 = nullptr; p.y16 = nullptr; p.M = (int)M; p.K = 512;
  p.part = workspace; p.W2 = (const _Float16*)w2; p.bias2 = b2;
  constexpr int LDS = LlTile<128, 3>::LDS;
  FP_SET_MAX_LDS((k_linear_ln512<128, 3, true, true>), LDS);
  hipLaunchKernelGGL((k_linear_ln512<128, 3, true, true>), dim3(tiles), dim3(LL_THREADS), LDS, (hipStream_t)stream, p);
  hipLaunchKernelGGL(k_ln_mean_finish, dim3(groups), dim3(512), 0, (hipStream_t)stream, (const float*)workspace, gamma, beta, out,
                     rows_per_group);
  FP_CHECK_LAUNCH("fp_ffn_layernorm_mean_fwd");
  return FP_OK;
}

extern "C" int fp_linear_layernorm_fwd(const void* x16, const void* w16, const float* bias, const float* x32, const void* tok16,
                                       const float* pe, int S, const float* gamma, const float* beta, float eps, float* y32,
                                       void* y16, int M, int K, int D, void* stream) {
  FP_REQUIRE(M >= 0, "fp_linear_layernorm_fwd: M < 0");
  if (M == 0) return FP_OK;
  FP_REQUIRE(x16 && w16 && gamma && beta && (y32 || y16), "fp_linear_layernorm_fwd: NULL tensor");
  FP_REQUIRE((x32 != nullptr) != (tok16 != nullptr), "fp_linear_layernorm_fwd: give the residual as x32 OR as tok16 (+ pe)");
  FP_REQUIRE(x32 || (pe && S > 0), "fp_linear_layernorm_fwd: tok16 needs the positional table and its period");
  FP_REQUIRE(D == 512, "fp_linear_layernorm_fwd: D=%d unsupported (d_model of both networks is 512)", D);
  FP_REQUIRE(K > 0 && K % LL_BK == 0, "fp_linear_layernorm_fwd: K=%d must be a multiple of %d", K, LL_BK);
  FP_REQUIRE((long long)M * K < (1ll << 30) && (long long)D * K < (1ll << 30), "fp_linear_layernorm_fwd: operands exceed 2 GiB");
  FP_REQUIRE((((size_t)x16 | (size_t)w16 | (size_t)bias | (size_t)x32 | (size_t)tok16 | (size_t)pe | (size_t)gamma | (size_t)beta |
               (size_t)y32 | (size_t)y16) & 15) == 0, "fp_linear_layernorm_fwd: tensors must be 16-byte aligned");
  LinearLnParams p;
  p.X = (const _Float16*)x16; p.Wt = (const _Float16*)w16; p.bias = bias; p.x32 = x32; p.tok16 = (const _Float16*)tok16; p.pe = pe;
  p.S = S; p.gamma = gamma; p.beta = beta; p.eps = eps; p.y32 = y32; p.y16 = (_Float16*)y16; p.M = M; p.K = K; p.part = nullptr; p.W2 = nullptr; p.bias2 = nullptr;
#ifdef FP_PROFILE_BUILD
  // profiling build only: FP_LL_TILE=64 selects the 64-row tile with two workgroups per CU (bit-identical by construction: the
  // same k order and the same row code; scripts/bench_linear_ln.py)
  static int tile64 = -1;
  if (tile64 < 0) { const char* e = getenv("FP_LL_TILE"); tile64 = (e && !strcmp(e, "64")) ? 1 : 0; }
  if (tile64) return ll_launch<64, 2>(p, (hipStream_t)stream);
#endif
  return ll_launch<128, 3>(p, (hipStream_t)stream);
}
