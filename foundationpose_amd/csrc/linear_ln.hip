// fp_linear_layernorm_fwd -- a 512 -> 512 Linear of nn.TransformerEncoderLayer (refine_network.py:56-70: self_attn.out_proj,
// linear2) fused with the residual add and the post-norm LayerNorm that consume it:
//     branch = f16(x16 @ W^T + b)                    (nn.Linear under autocast: fp32 accumulate + bias, one rounding)
//     z      = resid + f32(branch)                   (fp32 residual stream; resid = x32, or f32(tok16) + pe[row % S])
//     y      = LN(z) * gamma + beta  -> y32 and / or y16
// = fp_igemm_f16_fwd (taps = 1) followed by fp_layernorm_res_fwd, without the (M, 512) branch tensor ever reaching HBM and
// without the second launch; fp_ffn_layernorm_mean_fwd: the whole feed-forward half of the layer + the token mean the same way.
// A workgroup owns 128 complete rows (tile 128 x 512, so the LayerNorm statistics of a row stay inside the workgroup): 8 waves,
// wave w computes channels [64 w, 64 w + 64) of all 128 rows as 4 x 2 v_mfma_f32_32x32x16_f16 tiles (128 accumulator registers).
//
// Operand paths (round 4; the first version staged BOTH operands through a 3-stage LDS ring, lock step, and its second loop read
// the weights with one 16-byte global load per lane -- a row per lane):
//   * the weight rows of a wave are private to it, so they never go through LDS: they are read from L2 into registers, three
//     k-steps ahead, from a FRAGMENT-PACKED copy of the matrix (fp_pack_linear512_f16): the 64 lanes x 16 B of one MFMA operand are
//     one contiguous KiB.  Measured on the unpacked matrix (profiles/r04_ffn_second_loop_variants.log): a wave load whose 64 lanes
//     hit 32 different rows is served at about one lane per clock by the vector cache, 1 KiB per ~64 clk and CU, and a loop of
//     8 waves x 2 such loads per 16 MFMAs ran at half the MFMA rate; the same loop on contiguous KiBs runs at the rate it has with
//     the loads removed;
//   * the A tile (128 rows x K = 512: 128 KiB) fits LDS whole, as 16 k-step blocks of 8 KiB (rows of 64 B, XOR-swizzled): every
//     block has its own place, so there is no ring, no "buffer free" barrier, and the LDS-DMA requests run six k-steps ahead of
//     their use; per k-step ONE barrier makes the next block visible, and it sits between the two MFMA groups of the step with
//     the fragment reads of the following group already issued (the lock-step loop exposed a full LDS round trip per k-step).
// The epilogue parks f16(acc + bias) in the swizzled LDS tile of igemm_epilogue.h (128 rows x 1 KiB, laid over the A blocks) and
// then runs the row code of k_layernorm_res512 (rowops_ln.h, the same source) on those rows, 16 rows per wave, four at a time.
// Per element the same instruction sequence and the same summation order as the two-kernel path it replaces.
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "igemm_common.h"
#include "rowops_ln.h"

#ifdef FP_PROFILE_BUILD
// profiling build only: 100 MHz wall-clock time per phase of k_rows512 (thread 0 of every workgroup), summed over the workgroups of
// a launch (scripts/dbg_linear_ln.py): [0] first loop incl. cold start, [1] park + second loop (FFN), [2] park of the output,
// [3] LayerNorm rows (HBM-bound tail), [4] workgroups, [6] / [7] first start / last end
__device__ unsigned long long ll_dbg[8];
extern "C" int fp_dbg_linear_ln(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, ~0ull, 0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(ll_dbg), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ll_dbg), 8 * sizeof(unsigned long long));
}
#define LL_CLK(t) const unsigned long long t = wall_clock64()
#else
#define LL_CLK(t)
#endif

namespace {

constexpr int LL_BM = 128, LL_BN = 512, LL_K = 512, LL_NW = 8, LL_THREADS = LL_NW * 64, LL_BK = 32;
constexpr int LL_NK = LL_K / LL_BK;                      // 16 k-steps
constexpr int LL_KK = LL_BK / 16;                        // MFMA k-substeps per k-step
constexpr int LL_TM = LL_BM / 32;                        // row tiles of a wave
constexpr int LL_ROWB = LL_BK * 2;                       // bytes per row of an A block
constexpr int LL_A_BLOCK = LL_BM * LL_ROWB;              // 8 KiB: one k-step of the A tile
constexpr int LL_E_BYTES = LL_BM * LL_BN * 2;            // the epilogue tile = the 16 A blocks it is laid over
constexpr int LL_LDS = LL_E_BYTES + LL_BN * 4;           // + the bias vector
constexpr int LL_ROWS_PER_WAVE = LL_BM / LL_NW, LL_R = 4;   // rows a wave normalises together (interleaved reduction chains)
constexpr int LL_LA = 6, LL_LW = 2;                      // k-steps between the request of an A block / a weight fragment and its use
static_assert(LL_NK * LL_A_BLOCK == LL_E_BYTES && LL_LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
static_assert(LL_LA > LL_LW && LL_NK > LL_LA, "request order: A block j + LA - LW goes out just before weight fragment j");

struct LinearLnParams {
  const _Float16* X;      // (M, 512)
  const _Float16* Wp;     // fragment-packed (512, 512), see k_pack_w512
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
  int M;
  int ldx;                // row stride of X in fp16 values (512 = dense; 1024: one head's half of a two-head attention output)
  float* part;            // MEAN form: chunk sums [M / 16][512] (16 consecutive rows each); S = rows per group
  const _Float16* W2p;    // FFN form: second Linear, fragment-packed, and its bias; Wp / bias are the first (+ ReLU)
  const float* bias2;
  const _Float16* W1p;    // TAIL form: Wp / bias = out_proj, W1p / bias1 = linear1 (+ ReLU), W2p / bias2 = linear2; gamma / beta = norm1's
  const float* bias1;     //   (norm2's are applied by k_ln_mean_finish); y32 = scratch for norm1's fp32 output (the second residual)
};

// Fragment-packed weight matrix: for channel group w (64 output channels = one wave), k16-step q, channel tile i (32 channels), the
// 64 lanes' MFMA operands stored back to back,
//     packed[((w * (K / 16) + q) * 2 + i) * 64 + lane][0..7] = W[64 w + 32 i + (lane & 31)][16 q + 8 (lane >> 5) + 0..7]
// so that one wave load reads one contiguous KiB.  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void k_pack_w512(const _Float16* __restrict__ W, _Float16* __restrict__ out, int K) {
  const int o = blockIdx.x * 256 + threadIdx.x;          // chunk index of the packed matrix
  const int lane = o & 63, i = (o >> 6) & 1, rest = o >> 7, nq = K / 16;
  const int q = rest % nq, w = rest / nq;
  if (w >= LL_BN / 64) return;
  const half8 v = *reinterpret_cast<const half8*>(W + (size_t)(64 * w + 32 * i + (lane & 31)) * K + 16 * q + 8 * (lane >> 5));
  *reinterpret_cast<half8*>(out + (size_t)o * 8) = v;
}

__device__ __forceinline__ int ll_swz(int row) { return (row >> 2) & 3; }   // chunk swizzle of a 64-byte row (4 rows per bank row)

// the LDS-DMA lives in a plain function: written inline in a kernel TEMPLATE, the address_space(3) cast + builtin made hipcc 7.2
// drop the kernel's host stub without a diagnostic
__device__ __forceinline__ void ll_dma16(const __amdgpu_buffer_rsrc_t& rs, void* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// s_waitcnt vmcnt(n) with n known after unrolling (the switch folds to one instruction)
__device__ __forceinline__ void ll_wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Vector-memory operations a wave has issued AFTER weight fragment `k1` of the first loop at the point where it waits for it
// (between the two MFMA groups of k-step k1 - 1, or at the end of the prologue for k1 = 0).  Request order: the prologue sends A
// blocks 0 .. LA - LW - 1, then the pairs [A block j + LA - LW, fragment j] for j < LW; k-step t sends [A block t + LA (if it
// exists), fragment t + LW (if it exists)].  An A block is one LDS-DMA instruction per wave, a fragment four loads.
__device__ __forceinline__ constexpr int ll_after_w(int k1) {
  int n = 0;
  for (int j = k1 + 1; j < LL_NK && j <= k1 - 1 + LL_LW; ++j) n += 4;                       // fragments k1 + 1 .. k1 - 1 + LW  (sent up to k-step k1 - 1)
  for (int j = k1 + 1; j <= k1 - 1 + LL_LW; ++j) n += (j + LL_LA - LL_LW < LL_NK) ? 1 : 0;  // the A block that precedes each of them
  return n;
}

// MEAN (fp_ffn_layernorm_mean_fwd): instead of writing the normalised rows, add them up per group of p.S rows (the token mean of
// refine_network.py:90-91 fused with norm2).  The order of that sum must not depend on WHERE in the batch a hypothesis sits
// (sub-batches, shards and single batches have to agree bit for bit): a wave therefore owns 16 CONSECUTIVE rows
// of the tile -- a chunk that lies inside one hypothesis because p.S and the tile height are multiples of 16 -- adds them in row
// order and writes the chunk sum; k_ln_mean_finish adds a hypothesis's S / 16 chunk sums in chunk order.
// FFN: TWO Linears back to back on the tile, linear1 + ReLU -> the 128 x 512 intermediate parked in the epilogue tile -> linear2
// reading its A fragments from that tile (no barrier in the second loop).
// TAIL (round 6, fp_encoder_tail_mean_fwd): everything of the encoder layer behind the attention context in ONE launch -- out_proj
// (first loop, A tile by LDS-DMA) -> park -> norm1's row code (fp32 result to the scratch p.y32, its fp16 rounding IN PLACE into the
// epilogue tile) -> linear1 out of that tile -> the FFN form's park / linear2 / park -> norm2's row code on y32 + token-mean chunk
// sums.  Per element the instruction sequences of k_rows512<false, false> followed by k_rows512<true, true>: the same bits; what is
// gone is the (M, 512) fp16 tensor between them (written + read), one launch and one cold start per layer.
template <bool MEAN, bool FFN, bool TAIL = false>
__global__ __launch_bounds__(LL_THREADS, 1) void k_rows512(LinearLnParams p) {
  static_assert(!TAIL || (MEAN && FFN), "the tail form ends in the FFN + token-mean form");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // = channel group of 64
  float* bias_lds = reinterpret_cast<float*>(smem + LL_E_BYTES);
  const int m0 = blockIdx.x * LL_BM;
  LL_CLK(t_start);

  // bias -> LDS (512 floats: waves 0 and 1 fetch 1 KiB each with one LDS-DMA; the oldest vector-memory operation of the wave,
  // so every later counted wait covers it; visible to the workgroup after the first barrier)
  if (wid < 2) {
    float* dst = bias_lds + wid * 256;
    if (p.bias) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, LL_BN * 4, 0x00020000);
      ll_dma16(rs, dst, lane * 16, wid * 1024);
    } else {
      *reinterpret_cast<float4_*>(dst + lane * 4) = float4_{0.f, 0.f, 0.f, 0.f};
    }
  }

  // A blocks: wave w carries rows [16 w, +16) of every block; an LDS-DMA instruction writes 1 KiB lane-linear = 16 rows of 64 B,
  // so lane l carries row l / 4 and the LOGICAL chunk that belongs in physical chunk l % 4 of that row
  unsigned aoff32;
  {
    const int row = wid * 16 + lane / 4;
    const int c = (lane % 4) ^ ll_swz(row);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    aoff32 = (unsigned)(((size_t)m * p.ldx + c * 8) * 2);
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.X), 0, 0x7FFFFFFF, 0x00020000);
  auto request_a = [&](int ks) { ll_dma16(rsA, smem + ks * LL_A_BLOCK + wid * 1024, (int)aoff32, ks * (LL_BK * 2)); };
  // weight fragments of k-step ks: [kk][channel tile i], each one contiguous KiB of the packed matrix
  half8 wr[LL_LW + 1][LL_KK][2];
  auto request_w = [&](const _Float16* wp, int ks, int slot) {
    const _Float16* src = wp + ((size_t)wid * (LL_K / 16) + ks * LL_KK) * 1024 + lane * 8;
#pragma unroll
    for (int kk = 0; kk < LL_KK; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) wr[slot][kk][i] = *reinterpret_cast<const half8*>(src + (kk * 2 + i) * 512);
  };

  float16_ acc[2][LL_TM];   // [channel tile i][row tile j]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < LL_TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // A fragment addressing inside a block: lane reads row (lane & 31) of a 32-row tile, logical chunk 2 kk + (lane >> 5)
  const int frow = lane & 31, fhalf = lane >> 5;
  // (ll_swz ignores the row tile: 32 t does not reach bits 2-3 of the row; the 128 KiB of blocks need two base registers per kk,
  // everything else is an immediate offset of the read: ks % 8 blocks and the row tile)
  const unsigned char* a_ptr[2][LL_KK];
#pragma unroll
  for (int kk = 0; kk < LL_KK; ++kk) {
    a_ptr[0][kk] = smem + frow * LL_ROWB + (((2 * kk + fhalf) ^ ll_swz(frow)) << 4);
    a_ptr[1][kk] = a_ptr[0][kk] + 8 * LL_A_BLOCK;
  }
  half8 fa[2][LL_TM];
  auto read_a = [&](int ks, int kk, int slot) {
#pragma unroll
    for (int t = 0; t < LL_TM; ++t)
      fa[slot][t] = *reinterpret_cast<const half8*>(a_ptr[ks >> 3][kk] + (ks & 7) * LL_A_BLOCK + t * 32 * LL_ROWB);
  };
  auto mfma_group = [&](int wslot, int kk, int aslot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < LL_TM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[wslot][kk][i], fa[aslot][j], acc[i][j], 0, 0, 0);
  };

  // ---- first loop: x16 @ Wp^T.  Requests in the order of their use (vmcnt counts in order): see ll_after_w
#pragma unroll
  for (int j = 0; j < LL_LA - LL_LW; ++j) request_a(j);
#pragma unroll
  for (int j = 0; j < LL_LW; ++j) { request_a(j + LL_LA - LL_LW); request_w(p.Wp, j, j); }
  ll_wait_vm(ll_after_w(0));        // A block 0 (and the bias) have landed
  __builtin_amdgcn_s_barrier();     // ... everyone's part of them
  read_a(0, 0, 0);
#pragma unroll
  for (int ks = 0; ks < LL_NK; ++ks) {
    if (ks + LL_LA < LL_NK) request_a(ks + LL_LA);
    if (ks + LL_LW < LL_NK) request_w(p.Wp, ks + LL_LW, (ks + LL_LW) % (LL_LW + 1));
    read_a(ks, 1, 1);
    __builtin_amdgcn_sched_barrier(0);   // requests and fragment reads stay above the MFMAs that cover them
    mfma_group(ks % (LL_LW + 1), 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < LL_NK) {
      ll_wait_vm(ll_after_w(ks + 1));    // this wave's part of A block ks + 1 has landed (and weight fragment ks + 1)
      __builtin_amdgcn_s_barrier();      // ... everyone's
      read_a(ks + 1, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(ks % (LL_LW + 1), 1, 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();   // all fragment reads done before the A blocks become the epilogue tile
  LL_CLK(t_loop1);
  // residual rows (fp32 stream, or fp16 tokens + positional table) of LL_R rows of this wave, rows wid + 8 (t0 + u); a row
  // past the end of the matrix reads the last row instead: normalised, never stored
  half8 tk[LL_R];
  float rs[LL_R][8];
  // tile row of the wave's (t0 + u)-th row: interleaved (wave w: rows w, w + 8, ...) or, for the MEAN forms, one contiguous chunk
  auto tile_row = [&](int k) { return MEAN ? wid * LL_ROWS_PER_WAVE + k : wid + LL_NW * k; };
  auto request_resid = [&](int t0, const float* x32) {
#pragma unroll
    for (int u = 0; u < LL_R; ++u) {
      const int m = m0 + tile_row(t0 + u);
      const int mc = m < p.M ? m : p.M - 1;
      if (x32) {
        load8f(x32 + (size_t)mc * 512 + lane * 8, rs[u]);
      } else {
        tk[u] = *reinterpret_cast<const half8*>(p.tok16 + (size_t)mc * 512 + lane * 8);
        load8f(p.pe + (size_t)((unsigned)mc % (unsigned)p.S) * 512 + lane * 8, rs[u]);
      }
    }
  };
  // requested before the accumulators are parked (in flight under the transposition) where the registers allow it
  constexpr bool EARLY_RESID = !FFN;   // FFN: the second loop needs the registers
  const float* const resid32 = TAIL ? p.y32 : p.x32;       // the residual of the LAST row code (TAIL: norm1's output, written below)
  if constexpr (EARLY_RESID) request_resid(0, resid32);

  // ---- epilogue 1: f16(acc + bias) -> E[row][channel], rows of 1 KiB, the low 4 bits of the 16-byte chunk index XORed with
  // (row & 15) (igemm_epilogue.h).  D[i = channel][j = row]: a lane holds row (lane & 31) of a row tile and channels
  // 8 g + 4 (lane >> 5) + {0..3} of a channel tile, g = register >> 2
  unsigned char* E = smem;
  LL_CLK(t_pre);
  // A fragments out of the epilogue tile E (k16-step q, row tile j: row 32 j + (lane & 31), logical chunk 2 q + (lane >> 5)) and a whole
  // 512-deep product from it: weight fragments as in the first loop; nothing is shared between waves: no barrier inside
  // (the row offset goes through an empty asm per product: with two products in one kernel -- the TAIL form -- hipcc otherwise keeps the
  // 32 swizzled fragment addresses of the first alive for the second and spills ~110 registers around the MFMA loops)
  unsigned erow_off = (unsigned)(frow * (2 * LL_BN));
  int esw = frow & 15, efh = fhalf;
  auto read_e = [&](int q, int slot) {
    const unsigned char* erow = E + erow_off;
#pragma unroll
    for (int j = 0; j < LL_TM; ++j)
      fa[slot][j] = *reinterpret_cast<const half8*>(erow + j * 32 * (2 * LL_BN) + (((2 * q + efh) ^ esw) << 4));
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < LL_TM; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  auto product_from_e = [&](const _Float16* wp) {
    asm volatile("" : "+v"(erow_off), "+v"(esw), "+v"(efh));
#pragma unroll
    for (int j = 0; j < LL_LW; ++j) request_w(wp, j, j);
    read_e(0, 0);
#pragma unroll
    for (int ks = 0; ks < LL_NK; ++ks) {
      if (ks + LL_LW < LL_NK) request_w(wp, ks + LL_LW, (ks + LL_LW) % (LL_LW + 1));
      read_e(2 * ks + 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(ks % (LL_LW + 1), 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < LL_NK) read_e(2 * ks + 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(ks % (LL_LW + 1), 1, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // f16(acc + bias) [ReLU] -> E[row][channel]
  auto park = [&](auto relu_c) {
    constexpr bool RELU = decltype(relu_c)::value;
    int lane_p = lane;                       // opaque per call: the store addresses of one park are not kept (spilled) for the next
    asm volatile("" : "+v"(lane_p));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = wid * 64 + i * 32 + 8 * g + 4 * (lane_p >> 5);
        const float4_ bv = *reinterpret_cast<const float4_*>(bias_lds + nl);
#pragma unroll
        for (int j = 0; j < LL_TM; ++j) {
          const int ml = j * 32 + (lane_p & 31);
          half4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (_Float16)(acc[i][j][g * 4 + e] + bv[e]);
          if constexpr (RELU) v = __builtin_elementwise_max(v, half4{0, 0, 0, 0});
          const int chunk = (nl >> 3) ^ (ml & 15);
          *reinterpret_cast<half4*>(E + ml * (2 * LL_BN) + (chunk << 4) + ((nl & 4) << 1)) = v;
        }
      }
    }
  };
  auto load_bias = [&](const float* b) {   // (512) -> the bias area; waves 0 and 1, visible after the next s_waitcnt vmcnt(0) + barrier
    if (wid < 2) {
      float* dst = bias_lds + wid * 256;
      if (b) {
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, LL_BN * 4, 0x00020000);
        ll_dma16(rs_, dst, lane * 16, wid * 1024);
      } else {
        *reinterpret_cast<float4_*>(dst + lane * 4) = float4_{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  if constexpr (TAIL) {
    // ---- out_proj done: branch = f16(acc + bo) -> E, then k_layernorm_res512's row code (norm1) on the tile's rows: the fp32 result
    // goes to the scratch, its fp16 rounding takes the branch's place in E -- the A operand of linear1
    request_resid(0, nullptr);           // tok16 + pe rows, in flight under the transposition
    park(std::false_type{});
    __syncthreads();
    {
      float gm1[8], bt1[8];
      load8f(p.gamma + lane * 8, gm1);
      load8f(p.beta + lane * 8, bt1);
#pragma unroll 1
      for (int t0 = 0; t0 < LL_ROWS_PER_WAVE; t0 += LL_R) {
        float f[LL_R][8];
#pragma unroll
        for (int u = 0; u < LL_R; ++u) {
          const int r = tile_row(t0 + u);
          const half8 b = *reinterpret_cast<const half8*>(E + r * (2 * LL_BN) + ((lane ^ (r & 15)) << 4));
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            f[u][e] = (float)tk[u][e] + rs[u][e];             // resid_row (rowops_ln.h)
            f[u][e] += (float)b[e];
          }
        }
        if (t0 + LL_R < LL_ROWS_PER_WAVE) request_resid(t0 + LL_R, nullptr);
        ln_rows<LL_R>(p.eps, f);
#pragma unroll
        for (int u = 0; u < LL_R; ++u) {
          const int r = tile_row(t0 + u);
          const int m = m0 + r;
          half8 h;
#pragma unroll
          for (int e = 0; e < 8; ++e) { f[u][e] = fmaf(f[u][e], gm1[e], bt1[e]); h[e] = (_Float16)f[u][e]; }
          if (m < p.M) store8f(p.y32 + (size_t)m * 512 + lane * 8, f[u]);        // wave-uniform
          *reinterpret_cast<half8*>(E + r * (2 * LL_BN) + ((lane ^ (r & 15)) << 4)) = h;
        }
      }
    }
    __syncthreads();                 // norm1's fp16 output is complete in E (every wave reads all rows); nobody reads bo any more
    load_bias(p.bias1);
    zero_acc();
    product_from_e(p.W1p);           // linear1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // b1 has landed (waves 0, 1); this wave's y32 stores are out
    __syncthreads();                 // every wave is done reading norm1's output before H takes its place
  }
  if constexpr (FFN) {
    // ---- linear1 done: H = relu(f16(acc + b1)) -> E, then linear2 out of E
    park(std::true_type{});
    __syncthreads();                 // H complete; nobody reads b1 any more
    load_bias(p.bias2);              // b2 -> the bias area (visible after the barrier that ends the second loop)
    zero_acc();
    product_from_e(p.W2p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // b2 has landed (waves 0, 1)
    __syncthreads();                 // every wave is done reading H before linear2's output takes its place
  }
  LL_CLK(t_loop2);
  park(std::false_type{});
  __syncthreads();
  LL_CLK(t_park);
  if constexpr (!EARLY_RESID) request_resid(0, resid32);

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
        f[u][e] = resid32 ? rs[u][e] : (float)tk[u][e] + rs[u][e];     // resid_row (rowops_ln.h)
        f[u][e] += (float)b[e];
      }
    }
    if (t0 + LL_R < LL_ROWS_PER_WAVE) request_resid(t0 + LL_R, resid32);
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
#ifdef FP_PROFILE_BUILD
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LL_CLK(t_end);
  if (tid == 0) {
    atomicAdd(&ll_dbg[0], t_loop1 - t_start); atomicAdd(&ll_dbg[1], t_loop2 - t_pre); atomicAdd(&ll_dbg[2], t_park - t_loop2);
    atomicAdd(&ll_dbg[3], t_end - t_park); atomicAdd(&ll_dbg[4], 1ull); atomicMin(&ll_dbg[6], t_start); atomicMax(&ll_dbg[7], t_end);
  }
#endif
}

// ---- fp_linear512_f16_fwd: y16 = f16(x16 @ W^T + b) for K = 512 and N = 512 n (the in_proj of nn.MultiheadAttention: N = 1536) on the
// same operand paths: the workgroup's 128 x 512 A tile is fetched ONCE and stays in LDS while the N / 512 column blocks are
// computed one after the other (a wave: channels [64 w, +64) of the block), weights from the fragment-packed copy straight into
// registers.  The first block runs the first loop of k_rows512 (A blocks arriving, one barrier per k-step), the others find the
// tile complete: no barrier.  Epilogue without the 128 KiB tile (the A tile keeps LDS): each 32 x 32 accumulator tile goes
// through a wave-private 2 KiB buffer and leaves as 64-byte row pieces (4 consecutive lanes per row).
struct Linear512Params {
  const _Float16* X;      // (M, 512)
  const _Float16* Wp;     // N / 512 fragment-packed (512, 512) blocks back to back
  const float* bias;      // (N) or null
  _Float16* Y;            // (M, N)
  int M, N, relu;
};
constexpr int L5_XBUF = 32 * 64;                                   // a wave's transposition buffer: 32 rows x 32 channels fp16
constexpr int L5_BIAS_OFF = LL_E_BYTES + LL_NW * L5_XBUF;          // bias vector (N floats) behind the eight buffers
constexpr int L5_MAX_N = 3072;                                     // round 6: the in_proj of BOTH refiner heads in one launch (2 x 1536)
constexpr int l5_lds(int N) { return L5_BIAS_OFF + N * 4; }
static_assert(l5_lds(L5_MAX_N) <= 160 * 1024, "tile does not fit the 160 KiB LDS");

__global__ __launch_bounds__(LL_THREADS, 1) void k_linear512(Linear512Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* bias_lds = reinterpret_cast<float*>(smem + L5_BIAS_OFF);
  unsigned char* xbuf = smem + LL_E_BYTES + wid * L5_XBUF;
  const int m0 = blockIdx.x * LL_BM;
  const int nblk = p.N / LL_BN;

  // bias -> LDS: N / 256 pieces of 1 KiB, one LDS-DMA each (piece q by wave q % 8; the oldest vector-memory operations of the wave, so
  // every later counted wait covers them)
  for (int q = wid; q < p.N / 256; q += LL_NW) {
    float* dst = bias_lds + q * 256;
    if (p.bias) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.N * 4, 0x00020000);
      ll_dma16(rs, dst, lane * 16, q * 1024);
    } else {
      *reinterpret_cast<float4_*>(dst + lane * 4) = float4_{0.f, 0.f, 0.f, 0.f};
    }
  }
  unsigned aoff32;
  {
    const int row = wid * 16 + lane / 4;
    const int c = (lane % 4) ^ ll_swz(row);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    aoff32 = (unsigned)(((size_t)m * LL_K + c * 8) * 2);
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.X), 0, 0x7FFFFFFF, 0x00020000);
  auto request_a = [&](int ks) { ll_dma16(rsA, smem + ks * LL_A_BLOCK + wid * 1024, (int)aoff32, ks * (LL_BK * 2)); };
  half8 wr[LL_LW + 1][LL_KK][2];
  auto request_w = [&](const _Float16* wp, int ks, int slot) {
    const _Float16* src = wp + ((size_t)wid * (LL_K / 16) + ks * LL_KK) * 1024 + lane * 8;
#pragma unroll
    for (int kk = 0; kk < LL_KK; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) wr[slot][kk][i] = *reinterpret_cast<const half8*>(src + (kk * 2 + i) * 512);
  };
  float16_ acc[2][LL_TM];
  const int frow = lane & 31, fhalf = lane >> 5;
  // (ll_swz ignores the row tile: 32 t does not reach bits 2-3 of the row; the 128 KiB of blocks need two base registers per kk,
  // everything else is an immediate offset of the read: ks % 8 blocks and the row tile)
  const unsigned char* a_ptr[2][LL_KK];
#pragma unroll
  for (int kk = 0; kk < LL_KK; ++kk) {
    a_ptr[0][kk] = smem + frow * LL_ROWB + (((2 * kk + fhalf) ^ ll_swz(frow)) << 4);
    a_ptr[1][kk] = a_ptr[0][kk] + 8 * LL_A_BLOCK;
  }
  half8 fa[2][LL_TM];
  auto read_a = [&](int ks, int kk, int slot) {
#pragma unroll
    for (int t = 0; t < LL_TM; ++t)
      fa[slot][t] = *reinterpret_cast<const half8*>(a_ptr[ks >> 3][kk] + (ks & 7) * LL_A_BLOCK + t * 32 * LL_ROWB);
  };
  auto mfma_group = [&](int wslot, int kk, int aslot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < LL_TM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[wslot][kk][i], fa[aslot][j], acc[i][j], 0, 0, 0);
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < LL_TM; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  // f16(acc + bias) of column block nb -> Y, tile by tile through the wave's buffer.  D[i = channel][j = row]: a lane holds row
  // (lane & 31) of a row tile and channels 8 g + 4 (lane >> 5) + {0..3} of a channel tile, g = register >> 2
  auto store_block = [&](int nb) {
    const half4 zero4 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int nl = nb * LL_BN + wid * 64 + i * 32;                 // first channel of the tile
#pragma unroll
      for (int j = 0; j < LL_TM; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4_ bv = *reinterpret_cast<const float4_*>(bias_lds + nl + 8 * g + 4 * fhalf);
          half4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (_Float16)(acc[i][j][g * 4 + e] + bv[e]);
          if (p.relu) v = __builtin_elementwise_max(v, zero4);
          *reinterpret_cast<half4*>(xbuf + frow * 64 + ((g ^ ll_swz(frow)) << 4) + 8 * fhalf) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the buffer is wave-private, no barrier needed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = h * 16 + (lane >> 2), c = lane & 3;
          const half8 o = *reinterpret_cast<const half8*>(xbuf + r * 64 + ((c ^ ll_swz(r)) << 4));
          const int m = m0 + j * 32 + r;
          if (m < p.M) *reinterpret_cast<half8*>(p.Y + (size_t)m * p.N + nl + c * 8) = o;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);            // the reads are done before the next tile overwrites the buffer
        __builtin_amdgcn_wave_barrier();
      }
    }
  };

  // ---- column block 0: the first loop of k_rows512
  zero_acc();
#pragma unroll
  for (int j = 0; j < LL_LA - LL_LW; ++j) request_a(j);
#pragma unroll
  for (int j = 0; j < LL_LW; ++j) { request_a(j + LL_LA - LL_LW); request_w(p.Wp, j, j); }
  ll_wait_vm(ll_after_w(0));
  __builtin_amdgcn_s_barrier();
  read_a(0, 0, 0);
#pragma unroll
  for (int ks = 0; ks < LL_NK; ++ks) {
    if (ks + LL_LA < LL_NK) request_a(ks + LL_LA);
    if (ks + LL_LW < LL_NK) request_w(p.Wp, ks + LL_LW, (ks + LL_LW) % (LL_LW + 1));
    read_a(ks, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(ks % (LL_LW + 1), 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < LL_NK) {
      ll_wait_vm(ll_after_w(ks + 1));
      __builtin_amdgcn_s_barrier();
      read_a(ks + 1, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(ks % (LL_LW + 1), 1, 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- the other column blocks: the A tile is complete and visible (the last barrier above came after every block had landed)
  if (nblk > 1) {
#pragma unroll
    for (int j = 0; j < LL_LW; ++j) request_w(p.Wp + (size_t)LL_BN * LL_K, j, j);   // the next block's first weight fragments travel under the stores
  }
  store_block(0);
#pragma unroll 1
  for (int nb = 1; nb < nblk; ++nb) {
    const _Float16* wp = p.Wp + (size_t)nb * LL_BN * LL_K;
    zero_acc();
    read_a(0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < LL_NK; ++ks) {
      if (ks + LL_LW < LL_NK) request_w(wp, ks + LL_LW, (ks + LL_LW) % (LL_LW + 1));
      read_a(ks, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(ks % (LL_LW + 1), 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < LL_NK) read_a(ks + 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(ks % (LL_LW + 1), 1, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (nb + 1 < nblk) {
#pragma unroll
      for (int j = 0; j < LL_LW; ++j) request_w(wp + (size_t)LL_BN * LL_K, j, j);
    }
    store_block(nb);
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

}  // namespace

// Fragment-packed copy of a (512, 512) fp16 weight matrix (row-major [out][in], as nn.Linear holds it) for the two entry points
// below: see k_pack_w512.  Done once per weight matrix (engine.py packs when a plan is built).  include/fp_amd.h.
extern "C" int fp_pack_linear512_f16(const void* w16, void* packed, void* stream) {
  FP_REQUIRE(w16 && packed, "fp_pack_linear512_f16: NULL tensor");
  FP_REQUIRE(w16 != packed, "fp_pack_linear512_f16: cannot pack in place");
  FP_REQUIRE((((size_t)w16 | (size_t)packed) & 15) == 0, "fp_pack_linear512_f16: tensors must be 16-byte aligned");
  hipLaunchKernelGGL(k_pack_w512, dim3(LL_BN * LL_K / 8 / 256), dim3(256), 0, (hipStream_t)stream, (const _Float16*)w16, (_Float16*)packed,
                     LL_K);
  FP_CHECK_LAUNCH("fp_pack_linear512_f16");
  return FP_OK;
}

// y16 (M, N) = f16(x16 (M, 512) @ W^T + bias) [+ ReLU] for N = 512 n <= 2048: nn.Linear under autocast (fp32 accumulation + bias,
// one rounding), the in_proj of nn.MultiheadAttention (N = 1536; refine_network.py:56-70, score_network.py:52-53).  The bits of
// fp_igemm_f16_fwd (taps = 1): the same k order per accumulator.  w_packed: N / 512 blocks of 512 output channels, each
// fp_pack_linear512_f16'ed, back to back.  include/fp_amd.h.
extern "C" int fp_linear512_f16_fwd(const void* x16, const void* w_packed, const float* bias, void* y16, int M, int N, int relu,
                                    void* stream) {
  FP_REQUIRE(M >= 0, "fp_linear512_f16_fwd: M < 0");
  if (M == 0) return FP_OK;
  FP_REQUIRE(x16 && w_packed && y16, "fp_linear512_f16_fwd: NULL tensor");
  FP_REQUIRE(N >= LL_BN && N % LL_BN == 0 && N <= L5_MAX_N, "fp_linear512_f16_fwd: N=%d must be a multiple of 512, at most %d", N, L5_MAX_N);
  FP_REQUIRE((long long)M * N < (1ll << 31), "fp_linear512_f16_fwd: output exceeds 2^31 elements");
  FP_REQUIRE((((size_t)x16 | (size_t)w_packed | (size_t)bias | (size_t)y16) & 15) == 0, "fp_linear512_f16_fwd: tensors must be 16-byte aligned");
  Linear512Params p;
  p.X = (const _Float16*)x16; p.Wp = (const _Float16*)w_packed; p.bias = bias; p.Y = (_Float16*)y16; p.M = M; p.N = N; p.relu = relu ? 1 : 0;
  const int lds = l5_lds(N);
  FP_SET_MAX_LDS(k_linear512, l5_lds(L5_MAX_N));
  hipLaunchKernelGGL(k_linear512, dim3(fp_cdiv(M, LL_BM)), dim3(LL_THREADS), lds, (hipStream_t)stream, p);
  FP_CHECK_LAUNCH("fp_linear512_f16_fwd");
  return FP_OK;
}

// The whole feed-forward half of the refiner's encoder layer in one launch + the finish kernel -- linear1 + ReLU + linear2 +
// residual + norm2 + token mean (refine_network.py:56-70, :90-91); = two fp_igemm_f16_fwd + fp_colmean_f16_fwd with the (M, 512)
// intermediates staying in LDS.  Both Linears are 512 -> 512, weights fragment-packed (fp_pack_linear512_f16).  include/fp_amd.h.
extern "C" int fp_ffn_layernorm_mean_fwd(const void* y16, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2,
                                         const float* x32, const float* gamma, const float* beta, float eps, float* out,
                                         float* workspace, size_t workspace_bytes, int groups, int rows_per_group, void* stream) {
  FP_REQUIRE(groups >= 0, "fp_ffn_layernorm_mean_fwd: groups < 0");
  if (groups == 0) return FP_OK;
  FP_REQUIRE(y16 && w1_packed && w2_packed && x32 && gamma && beta && out && workspace, "fp_ffn_layernorm_mean_fwd: NULL tensor");
  FP_REQUIRE(rows_per_group >= 16 && rows_per_group % 16 == 0,
             "fp_ffn_layernorm_mean_fwd: rows_per_group=%d must be a multiple of 16 (a wave sums 16 consecutive rows, which have to belong to one group)", rows_per_group);
  const long long M = (long long)groups * rows_per_group;
  FP_REQUIRE(M * 512 < (1ll << 30), "fp_ffn_layernorm_mean_fwd: operands exceed 2 GiB");
  FP_REQUIRE((((size_t)y16 | (size_t)w1_packed | (size_t)w2_packed | (size_t)b1 | (size_t)b2 | (size_t)x32 | (size_t)gamma | (size_t)beta) & 15) == 0,
             "fp_ffn_layernorm_mean_fwd: tensors must be 16-byte aligned");
  const int tiles = fp_cdiv((int)M, LL_BM);
  FP_REQUIRE(workspace_bytes >= (size_t)(M / 16) * 512 * sizeof(float), "fp_ffn_layernorm_mean_fwd: workspace too small");
  LinearLnParams p;
  p.X = (const _Float16*)y16; p.Wp = (const _Float16*)w1_packed; p.bias = b1; p.x32 = x32; p.tok16 = nullptr; p.pe = nullptr;
  p.S = rows_per_group; p.gamma = gamma; p.beta = beta; p.eps = eps; p.y32 = nullptr; p.y16 = nullptr; p.M = (int)M; p.ldx = LL_K;
  p.part = workspace; p.W2p = (const _Float16*)w2_packed; p.bias2 = b2; p.W1p = nullptr; p.bias1 = nullptr;
  FP_SET_MAX_LDS((k_rows512<true, true>), LL_LDS);
  hipLaunchKernelGGL((k_rows512<true, true>), dim3(tiles), dim3(LL_THREADS), LL_LDS, (hipStream_t)stream, p);
  hipLaunchKernelGGL(k_ln_mean_finish, dim3(groups), dim3(512), 0, (hipStream_t)stream, (const float*)workspace, gamma, beta, out,
                     rows_per_group);
  FP_CHECK_LAUNCH("fp_ffn_layernorm_mean_fwd");
  return FP_OK;
}

// Everything of nn.TransformerEncoderLayer behind the attention context + the token mean, in ONE launch (+ the finish kernel):
// fp_linear_layernorm_fwd (out_proj + x + sa + norm1) followed by fp_ffn_layernorm_mean_fwd, with norm1's fp16 output staying in LDS
// (k_rows512<true, true, true>).  The same bits as those two calls.  include/fp_amd.h.
extern "C" size_t fp_encoder_tail_workspace_bytes(int groups, int rows_per_group) {
  if (groups <= 0 || rows_per_group <= 0) return 0;
  const size_t M = (size_t)groups * rows_per_group;
  return M * 512 * sizeof(float) + (M / 16 + 1) * 512 * sizeof(float);
}

extern "C" int fp_encoder_tail_mean_fwd(const void* ctx16, int ldx, const void* wo_packed, const float* bo, const void* tok16, const float* pe,
                                        const float* gamma1, const float* beta1, const void* w1_packed, const float* b1,
                                        const void* w2_packed, const float* b2, const float* gamma2, const float* beta2, float eps,
                                        float* out, void* workspace, size_t workspace_bytes, int groups, int rows_per_group, void* stream) {
  FP_REQUIRE(groups >= 0, "fp_encoder_tail_mean_fwd: groups < 0");
  if (groups == 0) return FP_OK;
  FP_REQUIRE(ctx16 && wo_packed && tok16 && pe && gamma1 && beta1 && w1_packed && w2_packed && gamma2 && beta2 && out && workspace,
             "fp_encoder_tail_mean_fwd: NULL tensor");
  FP_REQUIRE(rows_per_group >= 16 && rows_per_group % 16 == 0,
             "fp_encoder_tail_mean_fwd: rows_per_group=%d must be a multiple of 16 (a wave sums 16 consecutive rows, which have to belong to one group)", rows_per_group);
  if (ldx == 0) ldx = LL_K;
  FP_REQUIRE(ldx >= LL_K && ldx % 8 == 0, "fp_encoder_tail_mean_fwd: ldx=%d must be a multiple of 8 and at least 512", ldx);
  const long long M = (long long)groups * rows_per_group;
  FP_REQUIRE(M * ldx < (1ll << 30), "fp_encoder_tail_mean_fwd: operands exceed 2 GiB");
  FP_REQUIRE((((size_t)ctx16 | (size_t)wo_packed | (size_t)bo | (size_t)tok16 | (size_t)pe | (size_t)gamma1 | (size_t)beta1 | (size_t)w1_packed |
               (size_t)b1 | (size_t)w2_packed | (size_t)b2 | (size_t)gamma2 | (size_t)beta2 | (size_t)workspace) & 15) == 0,
             "fp_encoder_tail_mean_fwd: tensors must be 16-byte aligned");
  const size_t need = fp_encoder_tail_workspace_bytes(groups, rows_per_group);
  if (workspace_bytes < need) {
    fp_set_error("fp_encoder_tail_mean_fwd: workspace too small (%zu < %zu bytes, see fp_encoder_tail_workspace_bytes)", workspace_bytes, need);
    return FP_ERR_WORKSPACE;
  }
  LinearLnParams p;
  p.X = (const _Float16*)ctx16; p.ldx = ldx; p.Wp = (const _Float16*)wo_packed; p.bias = bo; p.x32 = nullptr; p.tok16 = (const _Float16*)tok16;
  p.pe = pe; p.S = rows_per_group; p.gamma = gamma1; p.beta = beta1; p.eps = eps; p.y16 = nullptr; p.M = (int)M;
  p.y32 = (float*)workspace;                                   // norm1's fp32 output: the residual of norm2
  p.part = (float*)workspace + (size_t)M * 512;                // chunk sums
  p.W1p = (const _Float16*)w1_packed; p.bias1 = b1; p.W2p = (const _Float16*)w2_packed; p.bias2 = b2;
  FP_SET_MAX_LDS((k_rows512<true, true, true>), LL_LDS);
  hipLaunchKernelGGL((k_rows512<true, true, true>), dim3(fp_cdiv((int)M, LL_BM)), dim3(LL_THREADS), LL_LDS, (hipStream_t)stream, p);
  hipLaunchKernelGGL(k_ln_mean_finish, dim3(groups), dim3(512), 0, (hipStream_t)stream, (const float*)p.part, gamma2, beta2, out, rows_per_group);
  FP_CHECK_LAUNCH("fp_encoder_tail_mean_fwd");
  return FP_OK;
}

extern "C" int fp_linear_layernorm_fwd(const void* x16, const void* w16_packed, const float* bias, const float* x32, const void* tok16,
                                       const float* pe, int S, const float* gamma, const float* beta, float eps, float* y32,
                                       void* y16, int M, int K, int D, int ldx, void* stream) {
  FP_REQUIRE(M >= 0, "fp_linear_layernorm_fwd: M < 0");
  if (M == 0) return FP_OK;
  FP_REQUIRE(x16 && w16_packed && gamma && beta && (y32 || y16), "fp_linear_layernorm_fwd: NULL tensor");
  FP_REQUIRE((x32 != nullptr) != (tok16 != nullptr), "fp_linear_layernorm_fwd: give the residual as x32 OR as tok16 (+ pe)");
  FP_REQUIRE(x32 || (pe && S > 0), "fp_linear_layernorm_fwd: tok16 needs the positional table and its period");
  FP_REQUIRE(D == 512 && K == 512, "fp_linear_layernorm_fwd: K=%d D=%d unsupported (d_model of both networks is 512; the A tile of 128 rows x K lives in LDS whole)", K, D);
  if (ldx == 0) ldx = K;
  FP_REQUIRE(ldx >= K && ldx % 8 == 0, "fp_linear_layernorm_fwd: ldx=%d must be a multiple of 8 and at least K", ldx);
  FP_REQUIRE((long long)M * ldx < (1ll << 30), "fp_linear_layernorm_fwd: operands exceed 2 GiB");
  FP_REQUIRE((((size_t)x16 | (size_t)w16_packed | (size_t)bias | (size_t)x32 | (size_t)tok16 | (size_t)pe | (size_t)gamma | (size_t)beta |
               (size_t)y32 | (size_t)y16) & 15) == 0, "fp_linear_layernorm_fwd: tensors must be 16-byte aligned");
  LinearLnParams p;
  p.X = (const _Float16*)x16; p.Wp = (const _Float16*)w16_packed; p.bias = bias; p.x32 = x32; p.tok16 = (const _Float16*)tok16; p.pe = pe;
  p.S = S; p.gamma = gamma; p.beta = beta; p.eps = eps; p.y32 = y32; p.y16 = (_Float16*)y16; p.M = M; p.ldx = ldx; p.part = nullptr; p.W2p = nullptr; p.bias2 = nullptr; p.W1p = nullptr; p.bias1 = nullptr;
  FP_SET_MAX_LDS((k_rows512<false, false>), LL_LDS);
  hipLaunchKernelGGL((k_rows512<false, false>), dim3(fp_cdiv(M, LL_BM)), dim3(LL_THREADS), LL_LDS, (hipStream_t)stream, p);
  FP_CHECK_LAUNCH("fp_linear_layernorm_fwd");
  return FP_OK;
}
