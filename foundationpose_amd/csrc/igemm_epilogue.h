// Shared epilogue of the implicit-GEMM kernels (igemm.hip, igemm_pp.hip): accumulators (+bias) -> half -> swizzled LDS
// tile E[m][n] -> 16-byte coalesced row stores with the residual add and ReLU applied on the way out.
// Call after a workgroup barrier that retires every read of the staging buffers (the tile reuses them).
// LDS layout of a kernel that uses it: [0, ig_lds_main) shared by the staging buffers of the main loop and, afterwards,
// the E tile + its two row-offset tables; then IG_BIAS_LDS bytes of per-channel vectors (ig_bias_to_lds).
#pragma once
#include "igemm_common.h"

#ifdef FP_PROFILE_BUILD
// profiling build only: 100 MHz wall-clock time per epilogue phase, summed over the workgroups of a launch (one copy per translation
// unit; conv_sw.hip reports its own through fp_dbg_conv_sw): [0] row tables + barrier, [1] residual requests, [2] accumulators ->
// E tile + barrier, [3] E tile -> stores issued (waits for the residual rows), [4] calls
static __device__ unsigned long long ig_epi_dbg[8];
#define IG_CLK(t) const unsigned long long t = wall_clock64()
#else
#define IG_CLK(t)
#endif

#define IG_VEC_FLOATS 256                       // per-channel epilogue vectors in LDS: bias | BatchNorm scale | BatchNorm shift
#define IG_BIAS_LDS (3 * IG_VEC_FLOATS * 4)

// FULL: every row of the tile exists (m0 + BM <= M) -- no per-row predicate, so the LDS reads and the stores of the
// 16 iterations are issued as batches instead of one LDS round trip after the other.
template <int BM, int BN>
constexpr int ig_lds_main(int stage_bytes) {
  return stage_bytes > BM * BN * 2 + BM * 16 ? stage_bytes : BM * BN * 2 + BM * 16;   // E tile + the two row-offset tables
}

template <int BM, int BN, int TM, int THREADS, int DBG, bool FULL>
__device__ __forceinline__ void ig_epilogue_body(const IgemmParams& p, float16_ (&acc)[2][TM], unsigned char* smem, int m0, int n0,
                                                 int wm, int wn, int tid, int lane, const float* bias_lds) {
  constexpr int CPR = BN / 8;                      // 16-byte chunks per row of the epilogue tile
  constexpr int NIT = (BM * CPR) / THREADS;
  // ---- epilogue: accumulators (+bias) -> half -> swizzled LDS tile E[m][n] -> 16-B coalesced row stores
  // Row addressing first: ONE thread per tile row maps the row to its element offsets in the output (and residual)
  // buffer and parks them in LDS behind the E tile; the store loop then reads them back as broadcasts.  (Computing the
  // map per (row, 16-byte chunk), as the first version did, cost 16-32 emulated 64-bit address computations per thread
  // and made the epilogue 20-45 % of the kernel.)
  unsigned char* E = smem;   // BM rows x (2*BN) B, low 4 bits of the chunk index XORed with (m & 15)
  // bias: staged into LDS at kernel entry (ig_bias_to_lds) -- a global load issued here would be exposed in full, and
  // under the operand stream's load that is several thousand cycles
  typedef float float4_ __attribute__((ext_vector_type(4)));
  const bool round_acc = p.round_acc != 0, has_bn = p.bn_scale != nullptr;
  long long* rowY = reinterpret_cast<long long*>(smem + BM * 2 * BN);
  long long* rowR = rowY + BM;
  IG_CLK(te0);
  for (int r = tid; r < BM; r += THREADS) {
    const int m = m0 + r;
    const bool in = m < p.M;
    rowY[r] = in ? ig_row_off(p.out, m) + n0 : -1;
    if (p.R) rowR[r] = in ? ig_row_off(p.res, m) + n0 : 0;
  }
  __syncthreads();
  IG_CLK(te1);
  // The residual rows are requested before the transposition: their HBM latency overlaps it
  half8 rv[NIT];
  if (p.R) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int qd = tid + it * THREADS;
      const int ml = qd / CPR, ch = qd % CPR;
      if (FULL || rowY[ml] >= 0) rv[it] = *reinterpret_cast<const half8*>(p.R + rowR[ml] + ch * 8);
    }
  }
  IG_CLK(te2);
  // D[i = channel][j = pixel]: lane holds pixel (lane & 31), channels 8g + 4*(lane>>5) + {0..3}, g = reg >> 2
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);   // first of 4 consecutive channels (tile-local)
      // per-channel vectors of these 4 channels (LDS broadcast reads; kept out of registers until here)
      const float4_ bv = *reinterpret_cast<const float4_*>(bias_lds + nl);
      float4_ sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (has_bn) {
        sc = *reinterpret_cast<const float4_*>(bias_lds + IG_VEC_FLOATS + nl);
        sh = *reinterpret_cast<const float4_*>(bias_lds + 2 * IG_VEC_FLOATS + nl);
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int ml = wm * (32 * TM) + j * 32 + (lane & 31);
        half4 v;
        if (round_acc) {
          // the reference's autocast op sequence for conv (+ BatchNorm): every op rounds its result to fp16.  Packed
          // fp16 math where it is exact: v_cvt_pk_f16_f32 for the conv output, v_pk_add_f16 for the bias (an IEEE half
          // add of two halves equals their fp32 sum rounded to half: when the fp32 sum is inexact the smaller addend is
          // below 1/8 ulp of the larger), fp32 FMA for BatchNorm (fp32 statistics)
          typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
          const half2_ b01 = {(_Float16)bv[0], (_Float16)bv[1]}, b23 = {(_Float16)bv[2], (_Float16)bv[3]};
          half2_ t01 = {(_Float16)acc[i][j][g * 4 + 0], (_Float16)acc[i][j][g * 4 + 1]};
          half2_ t23 = {(_Float16)acc[i][j][g * 4 + 2], (_Float16)acc[i][j][g * 4 + 3]};
          t01 = t01 + b01;
          t23 = t23 + b23;
          if (has_bn) {
            v[0] = (_Float16)fmaf((float)t01[0], sc[0], sh[0]);
            v[1] = (_Float16)fmaf((float)t01[1], sc[1], sh[1]);
            v[2] = (_Float16)fmaf((float)t23[0], sc[2], sh[2]);
            v[3] = (_Float16)fmaf((float)t23[1], sc[3], sh[3]);
          } else {
            v[0] = t01[0]; v[1] = t01[1]; v[2] = t23[0]; v[3] = t23[1];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (_Float16)(acc[i][j][g * 4 + e] + bv[e]);
        }
        const int chunk = (nl >> 3) ^ (ml & 15);
        *reinterpret_cast<half4*>(E + ml * (2 * BN) + (chunk << 4) + ((nl & 4) << 1)) = v;
      }
    }
  }
  __syncthreads();
  IG_CLK(te3);
  const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int qd = tid + it * THREADS;
    const int ml = qd / CPR, ch = qd % CPR;
    const long long yo = rowY[ml];
    if (!FULL && yo < 0) continue;
    half8 v = *reinterpret_cast<const half8*>(E + ml * (2 * BN) + ((ch ^ (ml & 15)) << 4));
    if (p.R) v = v + rv[it];                       // IEEE half add == the fp32 add of two halves rounded once
    if (p.relu) v = __builtin_elementwise_max(v, zero);
    *reinterpret_cast<half8*>(p.Y + yo + ch * 8) = v;
    if (p.Ype) {                                   // tokens + positional table, rounded for the in_proj GEMM
      const int m = m0 + ml;
      const float* per = p.pe + (size_t)(m % p.pe_period) * p.N + n0 + ch * 8;
      const float4_ e0 = *reinterpret_cast<const float4_*>(per), e1 = *reinterpret_cast<const float4_*>(per + 4);
      half8 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) { w[e] = (_Float16)((float)v[e] + e0[e]); w[4 + e] = (_Float16)((float)v[4 + e] + e1[e]); }
      *reinterpret_cast<half8*>(p.Ype + (size_t)m * p.N + n0 + ch * 8) = w;
    }
  }
#ifdef FP_PROFILE_BUILD
  IG_CLK(te4);
  if (tid == 0) {
    atomicAdd(&ig_epi_dbg[0], te1 - te0); atomicAdd(&ig_epi_dbg[1], te2 - te1); atomicAdd(&ig_epi_dbg[2], te3 - te2);
    atomicAdd(&ig_epi_dbg[3], te4 - te3); atomicAdd(&ig_epi_dbg[4], 1ull);
  }
#endif
}

template <int BM, int BN, int TM, int THREADS, int DBG = 0>
__device__ __forceinline__ void ig_epilogue(const IgemmParams& p, float16_ (&acc)[2][TM], unsigned char* smem, int m0, int n0,
                                            int wm, int wn, int tid, int lane, const float* bias_lds) {
  if (m0 + BM <= p.M) ig_epilogue_body<BM, BN, TM, THREADS, DBG, true>(p, acc, smem, m0, n0, wm, wn, tid, lane, bias_lds);
  else ig_epilogue_body<BM, BN, TM, THREADS, DBG, false>(p, acc, smem, m0, n0, wm, wn, tid, lane, bias_lds);
}
// Kernel entry, BEFORE the first operand stage is requested: wave 0 sends the tile's bias straight to LDS with one
// LDS-DMA (1 KiB = 256 floats; reads past the end of the bias vector return 0 through the buffer descriptor's bound).
// Being the oldest vector-memory operation of the wave it is covered by every later counted vmcnt wait, and it is
// visible to the workgroup after the first barrier of the main loop.  The LDS area is always IG_BIAS_LDS bytes.
// Three vectors of IG_VEC_FLOATS floats: bias, BatchNorm scale, BatchNorm shift (waves 0, 1, 2 fetch one each).
__device__ __forceinline__ void ig_bias_to_lds(const IgemmParams& p, int n0, float* bias_lds, int wid, int lane) {
  if (wid > 2) return;
  const float* src = wid == 0 ? p.bias : (wid == 1 ? p.bn_scale : p.bn_shift);
  float* dst = bias_lds + wid * IG_VEC_FLOATS;
  if (src) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, p.N * 4, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, n0 * 4, 0, 0);
  } else if (wid == 0) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<f4*>(dst + lane * 4) = f4{0.f, 0.f, 0.f, 0.f};
  }
}
