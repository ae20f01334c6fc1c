// fp_attention_f16_fwd -- softmax(q k^T / sqrt(d)) v for the self-attention blocks of the two networks
// (nn.MultiheadAttention(512, 4) inside nn.TransformerEncoderLayer, refine_network.py:56-70, and the MHA layers of
// score_network.py:52-53, :84-88), heads of 128, fp16 in / fp32 softmax and accumulation / fp16 out, as one flash-style
// MFMA kernel: the (B*H, S, S) probability tensor (161 M elements at N=252) never exists.
//
// Input is the in_proj output as it stands: rows of [q(H*128) | k(H*128) | v(H*128)]; output rows of H*128 (heads merged),
// the operand of out_proj.  One workgroup = 4 waves = 4 query tiles of 32 rows of one (sequence, head); K and V of that
// head stream through LDS in blocks of 64 keys (double buffered, the next block is in flight in registers while the
// current one is multiplied).
//
// MFMA bookkeeping (v_mfma_f32_32x32x16_f16; A lane l = A[l&31][8(l>>5)+i], B lane l = B[8(l>>5)+i][l&31],
// D lane l reg r = D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]):
//   * scores are computed TRANSPOSED, S^T = K Q^T (A = K rows, B = Q rows): a lane then owns one query (l&31) and 16 of
//     the 32 keys of a tile, so row max / row sum are 15 lane-local ops + one exchange with lane l^32;
//   * O^T = V^T P^T (A = V^T rows = one d per lane, B = P rows = one query per lane) keeps the output lane-local in the
//     query as well, so the online-softmax rescale is a plain per-lane multiply;
//   * the k index of that second product is a PERMUTATION of the keys: k-slot 8(l>>5)+i of k-step s stands for key
//     16s + 4(l>>5) + (i&3) + 8(i>>2), which is exactly the order in which a lane already holds its probabilities
//     (registers 8s..8s+7 of the score tile) -- no cross-lane movement between the two products; V^T is read from LDS
//     with the same permutation (two 8-byte reads of 4 consecutive keys each).
//   * V arrives [key][d] and is transposed while it is written to LDS (16-bit scatter, XOR-swizzled so that the
//     fragment reads are conflict-free and the scatter is 2-way).
#include <hip/hip_fp16.h>
#include "fp_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16_ __attribute__((ext_vector_type(16)));
typedef unsigned uint4_ __attribute__((ext_vector_type(4)));

constexpr int AT_D = 128;            // head size
constexpr int AT_KB = 64;            // keys per LDS block
constexpr int AT_K_BYTES = AT_KB * AT_D * 2;      // 16 KiB, [key][d], 16-byte chunks XORed with key & 15
constexpr int AT_V_BYTES = AT_D * AT_KB * 2;      // 16 KiB, [d][key position], key position = key ^ vswz(d)
constexpr int AT_BUF = AT_K_BYTES + AT_V_BYTES;
constexpr int AT_LDS = 2 * AT_BUF;                // 64 KiB: two workgroups per CU

__device__ __forceinline__ int vswz(int d) { return (4 * ((d >> 3) ^ (2 * (d & 7)))) & 60; }

__global__ __launch_bounds__(256, 2) void k_attention_f16(const _Float16* __restrict__ qkv, _Float16* __restrict__ out,
                                                          int S, int H, float c /* log2(e)/sqrt(d) */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lq = lane & 31, hi = lane >> 5;
  const int nqg = (S + 127) >> 7;                  // query groups of 128 rows per (sequence, head)
  const int qg = blockIdx.x % nqg, bh = blockIdx.x / nqg;
  const int h = bh % H, b = bh / H;
  const int ld = 3 * H * AT_D, ldo = H * AT_D;
  const _Float16* qp = qkv + (size_t)b * S * ld + h * AT_D;
  const _Float16* kp = qp + H * AT_D;
  const _Float16* vp = kp + H * AT_D;
  const int q0 = qg * 128 + wid * 32;              // first query row of this wave
  const bool wave_active = q0 < S;                 // idle waves still help staging and keep the barriers matched

  // ---- Q fragments (B operand): row q0 + lq, d = 16 kk + 8 hi + 0..7
  half8 qf[8];
  {
    int qr = q0 + lq;
    qr = qr < S ? qr : S - 1;
    const _Float16* src = qp + (size_t)qr * ld + 8 * hi;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = *reinterpret_cast<const half8*>(src + 16 * kk);
  }

  // ---- staging: thread t owns chunks c = t + 256 i (i < 4) of a block: key = c / 16, 16-byte chunk dc = c % 16
  uint4_ rk[4], rv[4];
  auto gload = [&](int blk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cidx = tid + 256 * i;
      int key = blk * AT_KB + (cidx >> 4);
      key = key < S ? key : S - 1;                 // rows past the end are masked below; any finite data will do
      const size_t off = (size_t)key * ld + (cidx & 15) * 8;
      rk[i] = *reinterpret_cast<const uint4_*>(kp + off);
      rv[i] = *reinterpret_cast<const uint4_*>(vp + off);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* kb = smem + buf * AT_BUF;
    unsigned char* vb = kb + AT_K_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cidx = tid + 256 * i;
      const int key = cidx >> 4, dc = cidx & 15;
      *reinterpret_cast<uint4_*>(kb + key * 256 + ((dc ^ (key & 15)) << 4)) = rk[i];
      const half8 v = __builtin_bit_cast(half8, rv[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = 8 * dc + e;
        *reinterpret_cast<_Float16*>(vb + d * 128 + ((key ^ vswz(d)) << 1)) = v[e];
      }
    }
  };

  float16_ o[4];                                   // O^T: d tile dt, lane = query
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = -1e30f, l = 0.f;                       // running row max (raw scores) and row sum, per query = per lane pair

  const int nblk = (S + AT_KB - 1) / AT_KB;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int blk = 0; blk < nblk; ++blk) {
    const int cur = blk & 1;
    if (blk + 1 < nblk) gload(blk + 1);            // in flight while this block is multiplied
    const unsigned char* kb = smem + cur * AT_BUF;
    const unsigned char* vb = kb + AT_K_BYTES;
    if (wave_active) {
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const int key0 = blk * AT_KB + 32 * sb;
        if (key0 >= S) break;                      // wave-uniform
        // ---- S^T tile = K (32 keys) x Q^T
        float16_ s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const int krow = 32 * sb + lq;
        const unsigned char* krp = kb + krow * 256;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const half8 kf = *reinterpret_cast<const half8*>(krp + (((2 * kk + hi) ^ (krow & 15)) << 4));
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], s, 0, 0, 0);
        }
        if (key0 + 32 > S) {                       // keys past the end of the sequence
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= S) s[r] = -1e30f;
        }
        // ---- online softmax (base 2, scores scaled by c inside the exponent)
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);
        if (__any(mn > m)) {                       // rescale what has been accumulated under the old maximum
          const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
          l *= alpha;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
          m = mn;
        }
        const float mc = m * c;
        float rs = 0.f;
        half8 pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c, -mc));
          rs += p;
          pf[r >> 3][r & 7] = (_Float16)p;
        }
        rs += __shfl_xor(rs, 32);
        l += rs;
        // ---- O^T += V^T P^T over the 32 keys (two k-steps of 16 permuted keys)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int d = 32 * dt + lq;
          const unsigned char* vrp = vb + d * 128;
          const int sw = vswz(d);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int kp0 = 32 * sb + 16 * ks + 4 * hi;
            const half4 lo = *reinterpret_cast<const half4*>(vrp + ((kp0 ^ sw) << 1));
            const half4 hi4 = *reinterpret_cast<const half4*>(vrp + (((kp0 + 8) ^ sw) << 1));
            const half8 vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[ks], o[dt], 0, 0, 0);
          }
        }
      }
    }
    if (blk + 1 < nblk) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- normalise, transpose through a wave-private LDS tile [32 queries][128 d] (16-byte chunks XORed with q & 15),
  // store whole 256-byte rows
  unsigned char* tile = smem + wid * (32 * 256);   // 4 x 8 KiB inside buffer 0 + 1 (everyone is past the last barrier)
  if (wave_active) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (_Float16)(o[dt][4 * g + e] * inv);
        const int chunk = (4 * dt + g) ^ (lq & 15);
        *reinterpret_cast<half4*>(tile + lq * 256 + (chunk << 4) + 8 * hi) = v;
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the tile is wave-private, no barrier needed
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = lane + 64 * it;
      const int r = idx >> 4, ch = idx & 15;
      const int q = q0 + r;
      if (q < S) {
        const uint4_ v = *reinterpret_cast<const uint4_*>(tile + r * 256 + ((ch ^ (r & 15)) << 4));
        *reinterpret_cast<uint4_*>(out + ((size_t)b * S + q) * ldo + h * AT_D + ch * 8) = v;
      }
    }
  }
}

}  // namespace

extern "C" int fp_attention_f16_fwd(const void* qkv, void* out, int B, int S, int H, int head_dim, void* stream) {
  FP_REQUIRE(B >= 0 && S >= 0, "fp_attention_f16_fwd: negative size");
  if (B == 0 || S == 0) return FP_OK;
  FP_REQUIRE(qkv && out, "fp_attention_f16_fwd: NULL tensor");
  FP_REQUIRE(head_dim == AT_D, "fp_attention_f16_fwd: head_dim=%d (only 128 is built)", head_dim);
  FP_REQUIRE(H > 0 && ((((size_t)qkv | (size_t)out) & 15) == 0), "fp_attention_f16_fwd: bad head count / unaligned tensors");
  const long long wgs = (long long)B * H * ((S + 127) / 128);
  FP_REQUIRE(wgs < (1ll << 31), "fp_attention_f16_fwd: too many workgroups");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_f16), hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS);
    attr_set = true;
  }
  const float c = 1.4426950408889634f / sqrtf((float)head_dim);
  hipLaunchKernelGGL(k_attention_f16, dim3((unsigned)wgs), dim3(256), AT_LDS, (hipStream_t)stream,
                     (const _Float16*)qkv, (_Float16*)out, S, H, c);
  FP_CHECK_LAUNCH("fp_attention_f16_fwd");
  return FP_OK;
}
