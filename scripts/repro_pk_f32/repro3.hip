// Characterisation of the packed-fp32 fault reproduced by repro2.hip (round 4): WHICH operand-select modifiers are affected, and
// WHAT do the wrong lanes hold?
//   hipcc --offload-arch=gfx950 -O2 scripts/repro_pk_f32/repro3.hip -o /tmp/repro3 && /tmp/repro3 [launches]
// victim forms (all write v[6:7] from v[4:5] = (a0, a1), v[2:3] = (b0, b1); v[6:7] preset to sentinels):
//   0  v_pk_mul_f32 v[6:7], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]      (a0 b1, a1 b0)   both halves of src1 crossed
//   1  v_pk_mul_f32 v[6:7], v[4:5], v[2:3] op_sel:[0,1]                      (a0 b1, a1 b1)   low result reads src1.hi
//   2  v_pk_mul_f32 v[6:7], v[4:5], v[2:3] op_sel_hi:[1,0]                   (a0 b0, a1 b0)   high result reads src1.lo
//   3  v_pk_add_f32 v[6:7], v[4:5], 0.5 op_sel_hi:[1,0]                      (a0 + .5, a1 + .5)   the pixel-centre add of k_raster
//   4  v_pk_fma_f32 v[6:7], v[4:5], v[2:3], v[2:3] op_sel_hi:[0,1,1]         (a0 b0 + b0, a0 b1 + b1)   scalar broadcast of src0.lo
//   5  v_pk_mul_f32 v[6:7], v[4:5], v[2:3]                                   (a0 b0, a1 b1)   no modifier (control)
// A wrong pair is classified: `stale` = the sentinels (the write did not happen), `default` = what the instruction returns with the
// modifiers at their defaults (op_sel = 0, op_sel_hi = 1), `other`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float float16_ __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// MFMA loops with the accumulators in AGPRs.  mode 0: four independent MFMAs back to back (repro2's hog_agpr); 1: one s_nop 0 after
// each; 2: ONE accumulator chain (dependent MFMAs); 3: 16x16x32 shape back to back
__global__ __launch_bounds__(256) void hog(float* out, int iters, int mode) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  float16_ c0 = {}, c1 = {}, c2 = {}, c3 = {};
  asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
  for (int it = 0; it < iters; ++it) {
    if (mode == 0)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                   "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                   : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
    else if (mode == 1)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n s_nop 0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n s_nop 0\n"
                   "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n s_nop 0\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n s_nop 0\n"
                   : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
    else if (mode == 2)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n"
                   : "+a"(c0) : "v"(a), "v"(b));
    else
      asm volatile("v_mfma_f32_16x16x32_f16 a[0:3], %0, %1, a[0:3]\n v_mfma_f32_16x16x32_f16 a[4:7], %0, %1, a[4:7]\n"
                   "v_mfma_f32_16x16x32_f16 a[8:11], %0, %1, a[8:11]\n v_mfma_f32_16x16x32_f16 a[12:15], %0, %1, a[12:15]\n"
                   : : "v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define SENT0 12345.0f
#define SENT1 54321.0f
// cnt[lane * 4 + {0: wrong, 1: stale, 2: default-modifier result, 3: other}]
__device__ unsigned g_nrec;
__device__ float g_rec[64][8];       // lane, x, y, a0, a1, b0, b1, iteration of the first wrong pairs

__global__ __launch_bounds__(256) void victim(unsigned* cnt, const float* __restrict__ data, int iters, int form) {
  const int tid = threadIdx.x;
  unsigned w = 0, st = 0, df = 0, ot = 0;
  for (int it = 0; it < iters; ++it) {
    const int p = (tid + 256 * it + 97 * blockIdx.x) & 65535;
    const float a0 = data[p], a1 = data[p + 65536], b0 = data[p + 131072], b1 = data[p + 196608];
    float x, y, xr, yr, xd, yd;                  // result, expected, result with default modifiers
#define RUN(INSN)                                                                                                      \
    asm volatile("v_mov_b32 v4, %2\n v_mov_b32 v5, %3\n v_mov_b32 v2, %4\n v_mov_b32 v3, %5\n"                           \
                 "v_mov_b32 v6, 0x4640e400\n v_mov_b32 v7, 0x47543100\n" INSN "\n v_mov_b32 %0, v6\n v_mov_b32 %1, v7\n"  \
                 : "=&v"(x), "=&v"(y) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v2", "v3", "v4", "v5", "v6", "v7")
    if (form == 0) { RUN("v_pk_mul_f32 v[6:7], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]"); xr = a0 * b1; yr = a1 * b0; xd = a0 * b0; yd = a1 * b1; }
    else if (form == 1) { RUN("v_pk_mul_f32 v[6:7], v[4:5], v[2:3] op_sel:[0,1]"); xr = a0 * b1; yr = a1 * b1; xd = a0 * b0; yd = a1 * b1; }
    else if (form == 2) { RUN("v_pk_mul_f32 v[6:7], v[4:5], v[2:3] op_sel_hi:[1,0]"); xr = a0 * b0; yr = a1 * b0; xd = a0 * b0; yd = a1 * b1; }
    else if (form == 3) { RUN("v_pk_add_f32 v[6:7], v[4:5], 0.5 op_sel_hi:[1,0]"); xr = a0 + 0.5f; yr = a1 + 0.5f; xd = a0 + 0.5f; yd = a1 + 0.0f; }
    else if (form == 4) { RUN("v_pk_fma_f32 v[6:7], v[4:5], v[2:3], v[2:3] op_sel_hi:[0,1,1]"); xr = __builtin_fmaf(a0, b0, b0); yr = __builtin_fmaf(a0, b1, b1); xd = xr; yd = __builtin_fmaf(a1, b1, b1); }
    else { RUN("v_pk_mul_f32 v[6:7], v[4:5], v[2:3]"); xr = a0 * b0; yr = a1 * b1; xd = xr; yd = yr; }
    asm volatile("" : "+v"(x), "+v"(y));
    if (x != xr || y != yr) {
      ++w;
      if (form <= 1) {
        const unsigned r = atomicAdd(&g_nrec, 1u);
        if (r < 64) { g_rec[r][0] = (float)(tid & 63); g_rec[r][1] = x; g_rec[r][2] = y; g_rec[r][3] = a0; g_rec[r][4] = a1; g_rec[r][5] = b0; g_rec[r][6] = b1; g_rec[r][7] = (float)it; }
      }
      if (x == SENT0 && y == SENT1) ++st;
      else if (x == xd && y == yd) ++df;
      else ++ot;
    }
  }
  if (w) { atomicAdd(&cnt[(tid & 63) * 4], w); atomicAdd(&cnt[(tid & 63) * 4 + 1], st); atomicAdd(&cnt[(tid & 63) * 4 + 2], df); atomicAdd(&cnt[(tid & 63) * 4 + 3], ot); }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 40;
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  unsigned* cnt; float* sink; float* data;
  CHECK(hipMalloc(&cnt, 256 * sizeof(unsigned)));
  CHECK(hipMalloc(&sink, 4096 * 256 * sizeof(float)));
  std::vector<float> hdata(4 * 65536);
  unsigned st = 12345u;
  for (auto& v : hdata) { st = st * 1664525u + 1013904223u; v = (float)(st >> 8) * (1.0f / 16777216.0f) + 0.05f; }
  CHECK(hipMalloc(&data, hdata.size() * sizeof(float)));
  CHECK(hipMemcpy(data, hdata.data(), hdata.size() * sizeof(float), hipMemcpyHostToDevice));
  const char* forms[6] = {"mul op_sel:[0,1] op_sel_hi:[1,0]", "mul op_sel:[0,1]", "mul op_sel_hi:[1,0]", "add v, 0.5 op_sel_hi:[1,0]",
                          "fma op_sel_hi:[0,1,1]", "mul (no modifier)"};
  const char* hogs[5] = {"alone", "4 MFMA 32x32x16 back to back", "... s_nop 0 between", "one dependent chain", "4 MFMA 16x16x32"};
  for (int h = 0; h < 5; ++h)
    for (int form = 0; form < 6; ++form) {
      if (h >= 2 && form != 0 && form != 3) continue;
      CHECK(hipMemsetAsync(cnt, 0, 256 * sizeof(unsigned), s0));
      CHECK(hipStreamSynchronize(s0));
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(victim, dim3(380), dim3(256), 0, s0, cnt, data, 400, form);
        if (h) hipLaunchKernelGGL(hog, dim3(2048), dim3(256), 0, s1, sink, 600, h - 1);
        CHECK(hipStreamSynchronize(s0));
        CHECK(hipStreamSynchronize(s1));
      }
      std::vector<unsigned> c(256);
      CHECK(hipMemcpy(c.data(), cnt, 256 * sizeof(unsigned), hipMemcpyDeviceToHost));
      unsigned long long q[4] = {0, 0, 0, 0}, k[4] = {0, 0, 0, 0};
      for (int l = 0; l < 64; ++l) { q[l >> 4] += c[l * 4]; for (int j = 0; j < 4; ++j) k[j] += c[l * 4 + j]; }
      printf("%-34s | %-30s: wrong %8llu = stale %llu + default-modifier result %llu + other %llu; lanes 0-15 / 16-31 / 32-47 / 48-63: %llu / %llu / %llu / %llu\n",
             forms[form], hogs[h], k[0], k[1], k[2], k[3], q[0], q[1], q[2], q[3]);
    }
  float rec[64][8];
  unsigned nrec = 0;
  CHECK(hipMemcpyFromSymbol(&nrec, HIP_SYMBOL(g_nrec), sizeof(nrec)));
  CHECK(hipMemcpyFromSymbol(rec, HIP_SYMBOL(g_rec), sizeof(rec)));
  printf("first wrong pairs of the op_sel forms (x should be a0 * b1):\n");
  for (unsigned r = 0; r < (nrec < 12 ? nrec : 12); ++r) {
    const float* q = rec[r];
    printf("  lane %2.0f it %3.0f: x = %.9g y = %.9g | a0 %.9g a1 %.9g b0 %.9g b1 %.9g | a0*b1 %.9g a0*b0 %.9g a1*b1 %.9g a1*b0 %.9g x/a0 %.9g x/a1 %.9g\n", q[0], q[7], q[1], q[2],
           q[3], q[4], q[5], q[6], q[3] * q[6], q[3] * q[5], q[4] * q[6], q[4] * q[5], q[1] / q[3], q[1] / q[4]);
  }
  return 0;
}
