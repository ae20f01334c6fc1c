// Patch-embed convolution of both encoders (refine_network.py:38, score_network.py:37: ConvBNReLU(6 -> 64, 7x7,
// stride 2, pad 3)) for gfx950, NHWC output.  By bytes it should be HBM-bound (0.48 GFLOP against 1.95 MB per pair of
// images); measured (shader-clock phase timers of the profiling build, scripts/dbg_conv1.py, 504 images) it is bound by
// latency chains inside a wave: 13 % issuing the patch DMA, 9 % waiting for it, 72 % in the k-loops and epilogues at
// ~40 % MFMA occupancy (the LDS round trip of a B fragment is longer than the two MFMAs it feeds, and 168 of 255
// registers hold weights, so the prefetch is one k-step deep).  0.22 ms for 504 images = 2.6 TB/s.
//   * persistent workgroups of four waves, two per CU (61 KB of LDS each): while one waits for its input patch the other
//     multiplies.  A wave owns a tile of 32 output pixels and ALL 64 output channels: the 64 x 294 weights are staged
//     ONCE per workgroup (coalesced, through LDS) from the PyTorch (64, 6*7*7) layout into MFMA A-fragments that live in
//     registers for the whole launch (168 VGPRs), so every B-fragment read from LDS feeds two MFMAs
//     (k re-ordered as (c, ky, [0, kx 0..6]) = 42 groups of 8 -> 21 k-steps of v_mfma_f32_32x32x16_f16; slot 0 is a zero weight);
//   * work unit = (image, band of 8 output rows): its 6 x 21 x (W+16) input patch goes HBM -> LDS by
//     global_load_lds_dwordx4 (zero padding comes from a 16-byte zero block, so the DMA stays lane-linear);
//   * B-fragment of a lane = the 8 consecutive input pixels 2*ox - 4 .. 2*ox + 3 of one (c, ky) row (the first one meets
//     the zero weight): four conflict-free dword reads used as they are, requested one k-step ahead; no im2col, no
//     index table, no per-k-step VALU work beyond one address add;
//   * D[channel][pixel] accumulators are rounded to fp16 (the conv output of the autocast sequence), transposed through a
//     wave-private swizzled LDS tile, and finished on the way out -- + bias (fp16), eval BatchNorm (fp32 FMA) -> fp16,
//     ReLU -- where a lane always holds the same 8 channels, whose bias / scale / shift stay in registers; 16-byte
//     stores, 128 contiguous bytes per pixel (all 64 channels).
#include <hip/hip_fp16.h>
#include "fp_common.h"
#include "igemm_common.h"   // ig_fastdiv

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
#define C1_ETILE 4096                // wave-private 32 px x 64 ch transpose tile

__device__ __attribute__((aligned(16))) const unsigned int c1_zero16[4] = {0u, 0u, 0u, 0u};

#ifdef FP_PROFILE_BUILD
// profiling build only: shader-clock time per phase, summed over the waves of a launch (scripts/dbg_conv1.py)
__device__ unsigned long long c1_dbg[8];
extern "C" int fp_dbg_conv1(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(c1_dbg), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c1_dbg), 8 * sizeof(unsigned long long));
}
#define C1_CLK(t) const unsigned long long t = __builtin_readcyclecounter()
#define C1_ADD(i, a, b) c1_t[i] += (b) - (a)
#else
#define C1_CLK(t)
#define C1_ADD(i, a, b)
#endif

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
  unsigned mulC, shrC;   // n / (PW / 8)   (ig_fastdiv)
  unsigned mulW, shrW;   // n / Wout
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
      const int t = ig_fastdiv(ch, p.mulC, p.shrC), q = ch - t * cpr;
      const int c = (t * 3121) >> 16, r = t - c * C1_PR;       // t / 21 for t < 6 * 21
      const int iy = iy0 + r, ix = (q - 1) * 8;   // patch element x = image x + 8
      if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) src = Xb + ((c * p.Hin + iy) * p.Win + ix);
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
  unsigned char* etile = smem + patch_bytes + wid * C1_ETILE;

  const bool has_bn = p.scale != nullptr;

  // ---- weights -> register-resident A fragments, both channel halves: wf[h][ks] = W[32 h + px][group 2*ks + kh][0..7].
  // The (64, 294) matrix comes in once per workgroup with coalesced 16-byte loads and is redistributed through LDS
  // (the patch area, before the first band): gathering the fragments straight from global memory is 294 two-byte loads
  // per lane at a lane stride of 588 B, ~9 k cache-line requests per wave, which cost a third of the whole launch.
  {
    constexpr int WCH = 64 * 294 * 2 / 16;            // 2352 16-byte chunks
    for (int c = tid; c < WCH; c += C1_THREADS)
      *reinterpret_cast<uint4_*>(smem + c * 16) = *reinterpret_cast<const uint4_*>(reinterpret_cast<const unsigned char*>(p.W) + c * 16);
  }
  __syncthreads();
  half8 wf[2][C1_KS];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ks = 0; ks < C1_KS; ++ks) {
      const int g = 2 * ks + kh;          // (c, ky) group
      const _Float16* w = reinterpret_cast<const _Float16*>(smem) + (h * 32 + px) * 294 + g * 7;   // c*49 + ky*7 == g*7
#pragma unroll
      for (int e = 0; e < 7; ++e) wf[h][ks][e + 1] = w[e];
      wf[h][ks][0] = (_Float16)0.f;
    }
  __syncthreads();                     // fragments are in registers before the staging area is reused
  // output side: this lane's 8 channels (16-byte chunk lane & 7 of a pixel's 64)
  const int ochunk = lane & 7;
  half8 obias;
  float osc[8], osh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    obias[e] = (_Float16)(p.bias ? p.bias[ochunk * 8 + e] : 0.f);
    osc[e] = p.scale ? p.scale[ochunk * 8 + e] : 1.f;
    osh[e] = p.scale ? p.shift[ochunk * 8 + e] : 0.f;
  }
  const int band_px = C1_ROWS * p.Wout;
  const int tiles = (band_px + 31) >> 5;
  const int row_bytes = p.PW * 2;
  const int Hp = p.Hout + 2 * p.pad, Wp = p.Wout + 2 * p.pad;

#ifdef FP_PROFILE_BUILD
  unsigned long long c1_t[6] = {0, 0, 0, 0, 0, 0};
#endif
  C1_CLK(tk_all);
  for (int band = blockIdx.x; band < p.total_bands; band += gridDim.x) {
    C1_CLK(ta);
    c1_stage(p, band, patch, tid);
    C1_CLK(tb);
    __builtin_amdgcn_s_waitcnt(0);     // this band's patch has landed (own DMA) ...
    __syncthreads();                   // ... and everyone's
    C1_CLK(tc);
    C1_ADD(0, ta, tb); C1_ADD(1, tb, tc);
    const int b = band / p.bands_per_image;
    const int oy0 = (band - b * p.bands_per_image) * C1_ROWS;
    for (int tile = wid; tile < tiles; tile += C1_WAVES) {
      int t = tile * 32 + px;
      t = t < band_px ? t : band_px - 1;
      const int oyl = ig_fastdiv(t, p.mulW, p.shrW), ox = t - oyl * p.Wout;
      // lane base: row 2*oyl of channel 0, dword (ox + 2) of the row  [element 2*ox + 4 = image x 2*ox - 4]
      const int lb = (2 * oyl) * row_bytes + (ox + 2) * 4;         // byte offset inside the patch (= inside smem)
      float16_ acc0, acc1;
      C1_CLK(tk0);
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
      // B fragment of a lane = the 8 consecutive input pixels 2*ox - 4 .. 2*ox + 3 of one (c, ky) row: four dwords, used
      // as they are (the zero weight of the k-group sits in FRONT of the seven taps, which makes the run start on an even
      // element).  The dwords of k-step ks+1 are requested before the MFMAs of k-step ks.
      unsigned int d[2][4];
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
        for (int i = 0; i < 4; ++i) d[slot][i] = src[i];
      };
      bread(0, 0);
#pragma unroll
      for (int ks = 0; ks < C1_KS; ++ks) {
        if (ks + 1 < C1_KS) bread(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);         // keep the requests in front of this k-step's MFMAs
        const unsigned int* dd = d[ks & 1];
        const uint4_ fv = {dd[0], dd[1], dd[2], dd[3]};
        const half8 fb = __builtin_bit_cast(half8, fv);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][ks], fb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][ks], fb, acc1, 0, 0, 0);
      }
#ifdef FP_PROFILE_BUILD
      asm volatile("" : "+v"(acc0), "+v"(acc1));
#endif
      C1_CLK(tk1);
      C1_ADD(2, tk0, tk1);
      // ---- epilogue.  The conv output is rounded to fp16 (first rounding point of the autocast sequence) and transposed
      // through the wave-private tile (32 px x 128 B); the rest of the sequence -- + bias (fp16), eval BatchNorm (fp32 FMA)
      // -> fp16, ReLU -- runs on the way out, where a lane always holds the SAME 8 channels (chunk = lane & 7): their
      // bias / scale / shift live in registers for the whole launch (fetching the vectors of the accumulator layout from
      // LDS per tile was a chain of ~24 dependent LDS round trips: half of the kernel's time)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float16_& acc = h ? acc1 : acc0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int chn = h * 32 + 8 * g4 + 4 * kh;        // first of this lane's 4 consecutive channels
          const half4 v = {(_Float16)acc[g4 * 4 + 0], (_Float16)acc[g4 * 4 + 1], (_Float16)acc[g4 * 4 + 2], (_Float16)acc[g4 * 4 + 3]};
          const int chunk = (chn >> 3) ^ (px & 7);         // 8 chunks of 16 B per pixel row
          *reinterpret_cast<half4*>(etile + px * 128 + (chunk << 4) + ((chn & 4) << 1)) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      half8 ov[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int pl = (lane >> 3) + 8 * it;
        ov[it] = *reinterpret_cast<const half8*>(etile + pl * 128 + ((ochunk ^ (pl & 7)) << 4));
      }
      // first pixel of the tile (wave-uniform), then 8 pixels further per iteration
      const int t0 = tile * 32 + (lane >> 3);
      const int oyt0 = ig_fastdiv(t0, p.mulW, p.shrW);
      int oyy = oy0 + oyt0, oxx = t0 - oyt0 * p.Wout;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        half8 v = ov[it] + obias;                          // IEEE half add == fp32 add of two halves rounded once
        if (has_bn) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (_Float16)fmaf((float)v[e], osc[e], osh[e]);
        }
        v = __builtin_elementwise_max(v, half8{0, 0, 0, 0, 0, 0, 0, 0});
        if (tile * 32 + (lane >> 3) + 8 * it < band_px && oyy < p.Hout)
          *reinterpret_cast<half8*>(p.Y + ((size_t)((b * Hp + oyy + p.pad) * Wp + oxx + p.pad)) * 64 + ochunk * 8) = v;
        oxx += 8;
        while (oxx >= p.Wout) { oxx -= p.Wout; ++oyy; }
      }
      __builtin_amdgcn_wave_barrier();
      C1_CLK(tk2);
      C1_ADD(3, tk1, tk2);
    }
    C1_CLK(td);
    __syncthreads();                   // everyone is done with the patch before the next band overwrites it
    C1_CLK(te);
    C1_ADD(4, td, te);
  }
#ifdef FP_PROFILE_BUILD
  if (lane == 0) {
    for (int i = 0; i < 5; ++i) atomicAdd(&c1_dbg[i], c1_t[i]);
    atomicAdd(&c1_dbg[5], __builtin_readcyclecounter() - tk_all);
    atomicAdd(&c1_dbg[6], 1ull);
  }
#endif
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
  ig_fastdiv_init(p.PW >> 3, &p.mulC, &p.shrC);
  ig_fastdiv_init(p.Wout, &p.mulW, &p.shrW);
  const int patch_bytes = ((C1_CIN * C1_PR * p.PW * 2 + 64 * 16 + 1023) / 1024) * 1024;
  size_t lds = (size_t)patch_bytes + C1_WAVES * C1_ETILE;
  if (lds < 64 * 294 * 2) lds = 64 * 294 * 2;     // the weight staging area of the prologue
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
