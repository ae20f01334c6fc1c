// Patch-embed convolution of both encoders (refine_network.py:38, score_network.py:37: ConvBNReLU(6 -> 64, 7x7,
// stride 2, pad 3)) for gfx950, NHWC output.  HBM-bound by construction (0.48 GFLOP against 1.95 MB per pair of
// images), so the kernel is organised around streaming:
//   * persistent workgroups of four waves, two per CU (61 KB of LDS each): while one waits for its input patch the other
//     multiplies.  A wave owns a tile of 32 output pixels and ALL 64 output channels: the 64 x 294 weights are loaded
//     ONCE, straight from the PyTorch (64, 6*7*7) layout, into MFMA A-fragments that live in registers for the whole
//     launch (168 VGPRs), so every B-fragment read from LDS feeds two MFMAs (the first versions split the channels
//     over two waves and read every fragment twice: LDS reads, not MFMA or HBM, set their pace)
//     (k re-ordered as (c, ky, kx[8]) = 42 groups of 8 -> 21 k-steps of v_mfma_f32_32x32x16_f16; kx = 7 is a zero weight);
//   * work unit = (image, band of 8 output rows): its 6 x 21 x (W+16) input patch goes HBM -> LDS by
//     global_load_lds_dwordx4 (zero padding comes from a 16-byte zero block, so the DMA stays lane-linear);
//   * B-fragment of a lane = 8 consecutive input pixels of one (c, ky) row starting at 2*ox - 3: five conflict-free
//     dword reads + four v_alignbit (the run starts on an odd element), requested one k-step ahead; no im2col, no index table;
//   * D[channel][pixel] accumulators go through the reference's autocast op sequence -- conv output rounded to fp16,
//     + bias (fp16), eval BatchNorm (fp32 statistics) rounded to fp16, ReLU -- are transposed through a wave-private
//     swizzled LDS tile and leave as 16-byte stores: 128 contiguous bytes per pixel (all 64 channels).
#include <hip/hip_fp16.h>
#include "fp_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16_ __attribute__((ext_vector_type(16)));
typedef unsigned int uint4_ __attribute__((ext_vector_type(4)));

#define C1_CIN 6
#define C1_ROWS 8                    // output rows per band
#define C1_PR (2 * C1_ROWS + 5)      // input rows per band
#define C1_KS 21                     // k-steps of 16 = 42 (c, ky) groups of 8 kx
#define C1_WAVES 4
#define C1_THREADS (64 * C1_WAVES)
#define C1_MAXW 256
#define C1_VEC_BYTES (3 * 64 * 4)    // bias | BatchNorm scale | shift as fp32, in LDS
#define C1_ETILE 4096                // wave-private 32 px x 64 ch transpose tile

__device__ __attribute__((aligned(16))) const unsigned int c1_zero16[4] = {0u, 0u, 0u, 0u};

struct Conv1Params {
  const _Float16* X;     // (B, 6, Hin, Win)
  const _Float16* W;     // (64, 294)
  const float* bias;     // (64) conv bias (fp16-representable values) or null
  const float* scale;    // (64) eval BatchNorm as x * scale + shift, or null
  const float* shift;    // (64)
  _Float16* Y;           // (B, Hout + 2*pad, Wout + 2*pad, 64)
  int B, Hin, Win, Hout, Wout, pad;
  int bands_per_image, total_bands;
  int PW;                // patch row length in halves = Win + 16
};

__device__ __forceinline__ void c1_stage(const Conv1Params& p, int band, unsigned char* patch, int tid) {
  const int b = band / p.bands_per_image;
  const int oy0 = (band - b * p.bands_per_image) * C1_ROWS;
  const int iy0 = 2 * oy0 - 3;
  const int cpr = p.PW >> 3;                       // 16-byte chunks per patch row
  const int nchunks = C1_CIN * C1_PR * cpr;
  const _Float16* Xb = p.X + (size_t)b * C1_CIN * p.Hin * p.Win;
  const int lane = tid & 63, wid = tid >> 6;
  // wave-instruction k of this wave covers chunks [64*(C1_WAVES*k + wid), +64): LDS destination is lane-linear
  for (int base = wid * 64; base < nchunks; base += C1_THREADS) {
    const int ch = base + lane;
    const void* src = c1_zero16;
    if (ch < nchunks) {
      const int q = ch % cpr, t = ch / cpr;
      const int r = t % C1_PR, c = t / C1_PR;
      const int iy = iy0 + r, ix = (q - 1) * 8;   // patch element x = image x + 8
      if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) src = Xb + ((size_t)c * p.Hin + iy) * p.Win + ix;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(patch + (size_t)base * 16), 16, 0, 0);
  }
}

// two workgroups of four waves per CU (one wave of each per SIMD): while one waits for its patch the other multiplies
__global__ __launch_bounds__(C1_THREADS, 2) void k_conv7x7s2_nhwc(Conv1Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int px = lane & 31, kh = lane >> 5;
  const int patch_bytes = ((C1_CIN * C1_PR * p.PW * 2 + 64 * 16 + 1023) / 1024) * 1024;  // DMA may overrun by < 64 chunks
  unsigned char* patch = smem;
  float* vec = reinterpret_cast<float*>(smem + patch_bytes);          // [bias | scale | shift][64]
  unsigned char* etile = smem + patch_bytes + C1_VEC_BYTES + wid * C1_ETILE;

  if (tid < 64) {
    vec[tid] = p.bias ? p.bias[tid] : 0.f;
    vec[64 + tid] = p.scale ? p.scale[tid] : 1.f;
    vec[128 + tid] = p.scale ? p.shift[tid] : 0.f;
  }
  const bool has_bn = p.scale != nullptr;

  // ---- weights -> register-resident A fragments, both channel halves: wf[h][ks] = W[32 h + px][group 2*ks + kh][0..7]
  half8 wf[2][C1_KS];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ks = 0; ks < C1_KS; ++ks) {
      const int g = 2 * ks + kh;          // (c, ky) group
      const _Float16* w = p.W + (size_t)(h * 32 + px) * 294 + g * 7;   // c*49 + ky*7 == g*7
#pragma unroll
      for (int e = 0; e < 7; ++e) wf[h][ks][e] = w[e];
      wf[h][ks][7] = (_Float16)0.f;
    }

  const int band_px = C1_ROWS * p.Wout;
  const int tiles = (band_px + 31) >> 5;
  const int row_bytes = p.PW * 2;
  const int Hp = p.Hout + 2 * p.pad, Wp = p.Wout + 2 * p.pad;

  for (int band = blockIdx.x; band < p.total_bands; band += gridDim.x) {
    c1_stage(p, band, patch, tid);
    __builtin_amdgcn_s_waitcnt(0);     // this band's patch has landed (own DMA) ...
    __syncthreads();                   // ... and everyone's
    const int b = band / p.bands_per_image;
    const int oy0 = (band - b * p.bands_per_image) * C1_ROWS;
    for (int tile = wid; tile < tiles; tile += C1_WAVES) {
      int t = tile * 32 + px;
      t = t < band_px ? t : band_px - 1;
      const int oyl = t / p.Wout, ox = t - oyl * p.Wout;
      // lane base: row 2*oyl of channel 0, dword (ox + 2) of the row  [element 2*ox + 5 = image x 2*ox - 3]
      const int lb = (2 * oyl) * row_bytes + (ox + 2) * 4;         // byte offset inside the patch (= inside smem)
      float16_ acc0, acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
      // B fragment of a lane = 8 consecutive input pixels starting at an odd element: five dwords, shifted by 16 bits.
      // The dwords of k-step ks+1 are requested before the MFMAs of k-step ks.
      unsigned int d[2][5];
      // group g = 2*ks + kh -> patch row (g / 7) * C1_PR + g % 7: the kh = 1 half of the wave reads the row of group
      // 2*ks + 1, which is 1 row further, or 15 rows where (c, ky = 6) is followed by (c + 1, ky = 0)
      const int lb1 = lb + kh * row_bytes;
      const int lb15 = lb + kh * 15 * row_bytes;
      auto bread = [&](int ks, int slot) {
        const int g0 = 2 * ks;
        const int off0 = (g0 / 7) * C1_PR + (g0 % 7);
        int a = ((g0 % 7) == 6 ? lb15 : lb1) + off0 * row_bytes;
        asm volatile("" : "+v"(a));               // one address at a time: 21 hoisted addresses would not fit the register file
        const unsigned int* src = reinterpret_cast<const unsigned int*>(smem + a);
#pragma unroll
        for (int i = 0; i < 5; ++i) d[slot][i] = src[i];
      };
      bread(0, 0);
#pragma unroll
      for (int ks = 0; ks < C1_KS; ++ks) {
        if (ks + 1 < C1_KS) bread(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);         // keep the requests in front of this k-step's MFMAs
        const unsigned int* dd = d[ks & 1];
        uint4_ fv;
        fv[0] = __builtin_amdgcn_alignbit(dd[1], dd[0], 16);
        fv[1] = __builtin_amdgcn_alignbit(dd[2], dd[1], 16);
        fv[2] = __builtin_amdgcn_alignbit(dd[3], dd[2], 16);
        fv[3] = __builtin_amdgcn_alignbit(dd[4], dd[3], 16);
        const half8 fb = __builtin_bit_cast(half8, fv);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][ks], fb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][ks], fb, acc1, 0, 0, 0);
      }
      // ---- epilogue: fp16(conv) + bias -> fp16 -> BN -> fp16 -> ReLU (the autocast op sequence of
      //      nn.Conv2d / nn.BatchNorm2d / nn.ReLU), transpose through the wave-private tile (32 px x 128 B), 16-byte stores
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float16_& acc = h ? acc1 : acc0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          // packed fp16 math where it is exact (see igemm_epilogue.h): conv -> fp16, + bias as an IEEE half add, BatchNorm
          // as an fp32 FMA rounded to fp16, ReLU on the packed halves
          typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
          typedef float float4_ __attribute__((ext_vector_type(4)));
          const int chn = h * 32 + 8 * g4 + 4 * kh;        // first of this lane's 4 consecutive channels
          const float4_ bi = *reinterpret_cast<const float4_*>(vec + chn);
          half2_ t01 = {(_Float16)acc[g4 * 4 + 0], (_Float16)acc[g4 * 4 + 1]};
          half2_ t23 = {(_Float16)acc[g4 * 4 + 2], (_Float16)acc[g4 * 4 + 3]};
          t01 = t01 + half2_{(_Float16)bi[0], (_Float16)bi[1]};
          t23 = t23 + half2_{(_Float16)bi[2], (_Float16)bi[3]};
          half4 v;
          if (has_bn) {
            const float4_ sc = *reinterpret_cast<const float4_*>(vec + 64 + chn);
            const float4_ sh = *reinterpret_cast<const float4_*>(vec + 128 + chn);
            v[0] = (_Float16)fmaf((float)t01[0], sc[0], sh[0]);
            v[1] = (_Float16)fmaf((float)t01[1], sc[1], sh[1]);
            v[2] = (_Float16)fmaf((float)t23[0], sc[2], sh[2]);
            v[3] = (_Float16)fmaf((float)t23[1], sc[3], sh[3]);
          } else {
            v[0] = t01[0]; v[1] = t01[1]; v[2] = t23[0]; v[3] = t23[1];
          }
          v = __builtin_elementwise_max(v, half4{0, 0, 0, 0});
          const int chunk = (chn >> 3) ^ (px & 7);         // 8 chunks of 16 B per pixel row
          *reinterpret_cast<half4*>(etile + px * 128 + (chunk << 4) + ((chn & 4) << 1)) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int qd = lane + 64 * it;
        const int pl = qd >> 3, chunk = qd & 7;
        const int tt = tile * 32 + pl;
        const half8 v = *reinterpret_cast<const half8*>(etile + pl * 128 + ((chunk ^ (pl & 7)) << 4));
        if (tt < band_px) {
          const int oyy = oy0 + tt / p.Wout, oxx = tt % p.Wout;
          if (oyy < p.Hout)
            *reinterpret_cast<half8*>(p.Y + (((size_t)b * Hp + oyy + p.pad) * Wp + oxx + p.pad) * 64 + chunk * 8) = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                   // everyone is done with the patch before the next band overwrites it
  }
}

extern "C" int fp_conv7x7s2_bn_relu_fwd(const void* x, const void* w, const float* bias, const float* bn_scale,
                                        const float* bn_shift, void* y, int B, int Hin, int Win, int pad, void* stream) {
  FP_REQUIRE(B >= 0, "fp_conv7x7s2_bn_relu_fwd: B < 0");
  if (B == 0) return FP_OK;
  FP_REQUIRE(x && w && y, "fp_conv7x7s2_bn_relu_fwd: NULL tensor");
  FP_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fp_conv7x7s2_bn_relu_fwd: bn_scale and bn_shift go together");
  FP_REQUIRE(Hin > 0 && Win > 0 && Hin % 2 == 0 && Win % 8 == 0 && Win <= C1_MAXW,
             "fp_conv7x7s2_bn_relu_fwd: input %dx%d unsupported (even height, width a multiple of 8 and <= %d)", Hin, Win, C1_MAXW);
  FP_REQUIRE(pad == 0 || pad == 1, "fp_conv7x7s2_bn_relu_fwd: pad must be 0 or 1");
  FP_REQUIRE((((size_t)x | (size_t)y) & 15) == 0, "fp_conv7x7s2_bn_relu_fwd: tensors must be 16-byte aligned");
  Conv1Params p;
  p.X = (const _Float16*)x; p.W = (const _Float16*)w; p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.Y = (_Float16*)y;
  p.B = B; p.Hin = Hin; p.Win = Win; p.Hout = Hin / 2; p.Wout = Win / 2; p.pad = pad;
  p.bands_per_image = fp_cdiv(p.Hout, C1_ROWS);
  FP_REQUIRE((long long)B * p.bands_per_image < (1ll << 31), "fp_conv7x7s2_bn_relu_fwd: batch too large");
  p.total_bands = B * p.bands_per_image;
  p.PW = Win + 16;
  const int patch_bytes = ((C1_CIN * C1_PR * p.PW * 2 + 64 * 16 + 1023) / 1024) * 1024;
  const size_t lds = (size_t)patch_bytes + C1_VEC_BYTES + C1_WAVES * C1_ETILE;
  if (lds > 160 * 1024) {
    fp_set_error("fp_conv7x7s2_bn_relu_fwd: input width %d needs %zu bytes of LDS", Win, lds);
    return FP_ERR_UNSUPPORTED;
  }
  int dev = 0;
  static int n_cu[64] = {0};
  (void)hipGetDevice(&dev);
  if (n_cu[dev & 63] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      fp_set_error("fp_conv7x7s2_bn_relu_fwd: cannot query the device");
      return FP_ERR_LAUNCH;
    }
    n_cu[dev & 63] = prop.multiProcessorCount;
  }
  FP_SET_MAX_LDS(k_conv7x7s2_nhwc, 160 * 1024);
  const int grid = p.total_bands < 2 * n_cu[dev & 63] ? p.total_bands : 2 * n_cu[dev & 63];
  hipLaunchKernelGGL(k_conv7x7s2_nhwc, dim3(grid), dim3(C1_THREADS), lds, (hipStream_t)stream, p);
  FP_CHECK_LAUNCH("fp_conv7x7s2_bn_relu_fwd");
  return FP_OK;
}
