// Shared by igemm.hip and conv3x3.hip: operand addressing and the kernel parameter block of fp_igemm_f16_fwd.
#pragma once
#include <hip/hip_fp16.h>
#include "fp_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16_ __attribute__((ext_vector_type(16)));

#define IG_BK 64

struct IgemmGeom {      // row m -> element offset of pixel (b, y*stride + pad_off, x*stride + pad_off) in a padded NHWC buffer
  int HoWo, Wo;         // output pixels per image / per row (1,1 for a plain GEMM)
  int Hp, Wp;           // padded height / width of the buffer
  int stride;           // spatial stride applied to (oy, ox)
  int off;              // border offset added to the pixel position (0 for the conv input: tap (0,0) = top-left pad)
  int cstride;          // channels per pixel in the buffer
  int coff;             // first channel
  int bsplit;           // image b -> (b % bsplit), channel group (b / bsplit) * cgroup  (0 = off)
  int cgroup;
  unsigned mulP, shrP;  // n / HoWo, n / Wo, n / bsplit as umulhi(n, mul) >> shr (host-computed, exact for n < 2^31)
  unsigned mulW, shrW;
  unsigned mulB, shrB;
};

// division by a runtime constant without the ~40-instruction emulated divide (the row -> address maps run 20 times per
// thread and tile): mul == 0 encodes divisor 1
static inline void ig_fastdiv_init(int d, unsigned* mul, unsigned* shr) {
  if (d <= 1) { *mul = 0; *shr = 0; return; }
  int lg = 0;
  while ((1u << lg) < (unsigned)d) ++lg;              // ceil(log2 d)
  const int p = 31 + lg;
  *mul = (unsigned)(((1ull << p) + (unsigned)d - 1) / (unsigned)d);
  *shr = (unsigned)(p - 32);
}
__device__ __forceinline__ int ig_fastdiv(int n, unsigned mul, unsigned shr) {
  return mul ? (int)(__umulhi((unsigned)n, mul) >> shr) : n;
}

struct IgemmParams {
  const _Float16* A;
  const _Float16* Wt;   // [N][taps*Cin]
  const _Float16* Wpk;  // the same weights tile-packed (fp_pack_conv3x3_tiles_f16) or null: read by k_conv_sw<512,128> only
  const float* bias;    // [N] or null
  const float* bn_scale;   // [N] or null: eval-mode BatchNorm as y = x * scale + shift, applied to the fp16-rounded conv + bias
  const float* bn_shift;
  const _Float16* R;    // residual or null
  _Float16* Y;
  const float* pe;      // optional second output (the positional-embedding add of network_modules.py:133-137 fused into the
  _Float16* Ype;        // last conv): Ype[m][n] = f16(f32(y[m][n]) + pe[m % pe_period][n]), a plain (M, N) matrix
  int pe_period;
  int M, N, Cin, taps;
  int relu;
  int round_acc;        // FP_IGEMM_ROUND_ACC: the accumulator is rounded to fp16 BEFORE the bias is added (nn.Conv2d under
                        // autocast: ATen adds the bias to the fp16 convolution output); 0: one rounding of acc + bias (nn.Linear)
  IgemmGeom in, out, res;
  float* slab;          // split-K only (fp_igemm_f16_splitk_fwd): fp32 partial accumulators in fragment order,
  int nsplit;           //   [split][tile][wave][(i, g, j)][lane] x float4; 0 = not split
};

__device__ __forceinline__ long long ig_row_off(const IgemmGeom& g, int m) {
  const int b = ig_fastdiv(m, g.mulP, g.shrP);
  const int r = m - b * g.HoWo;
  const int oy = ig_fastdiv(r, g.mulW, g.shrW);
  const int ox = r - oy * g.Wo;
  int bb = b, cg = 0;
  if (g.bsplit > 0) { cg = ig_fastdiv(b, g.mulB, g.shrB); bb = b - cg * g.bsplit; }
  return (((long long)bb * g.Hp + (oy * g.stride + g.off)) * g.Wp + (ox * g.stride + g.off)) * g.cstride + g.coff +
         (long long)cg * g.cgroup;
}

