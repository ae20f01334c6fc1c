// Minimal reproducer attempt for the finding of DESIGN.md 3.5: a packed-fp32 VALU instruction (v_pk_add_f32) in a plain
// VALU kernel returns wrong values in lanes 48..63 of a wave while an MFMA-heavy kernel of ANOTHER stream shares the CU.
//   hipcc --offload-arch=gfx950 -O2 repro.hip -o repro && ./repro
// Kernel `victim` runs, per lane and iteration, the instruction sequence the compiler had produced in k_raster's shading
// loop (raster.hip at -O3 with SLP vectorisation):
//     v_mad_u64_u32 v[a:a+1], sdst, s_neg_w, v_row, v[p:p+1]      ; i = p - w * row   (64-bit multiply-add, low half used)
//     v_cvt_f32_i32 v[a+1], v_j                                   ; y = float(j)
//     v_cvt_f32_i32 v[a],   v[a]                                  ; x = float(i)
//     v_pk_add_f32  v[a:a+1], v[a:a+1], 0.5                       ; (x + 0.5, y + 0.5)
// and checks the pair against the same values computed with scalar-fp32 VALU instructions.  Kernel `hog` is a register-
// resident MFMA loop.  Modes: victim alone; victim first, then the hog on a second stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float float16_ __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void hog(float* out, int iters) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  float16_ c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// bad[lane] += number of iterations in which the packed result of that lane differed from the scalar one
__global__ __launch_bounds__(256) void victim(unsigned* bad, int iters, int w, int packed) {
  const int tid = threadIdx.x;
  unsigned nbad = 0;
  const int negw = -w;
  for (int it = 0; it < iters; ++it) {
    const int p = tid + 256 * (it & 15) + blockIdx.x;          // pixel index, as in the shading loop
    const int row = p / w;                                    // emulated division, like the original
    const int j = row + 16 * (blockIdx.x & 7);
    float x, y;
    if (packed) {
      asm volatile(
          "v_mov_b32 v2, %2\n"
          "v_mov_b32 v3, 0\n"
          "v_mad_u64_u32 v[2:3], s[10:11], %3, %4, v[2:3]\n"
          "v_cvt_f32_i32_e32 v3, %5\n"
          "v_cvt_f32_i32_e32 v2, v2\n"
          "v_pk_add_f32 v[2:3], v[2:3], 0.5 op_sel_hi:[1,0]\n"
          "v_mov_b32 %0, v2\n"
          "v_mov_b32 %1, v3\n"
          : "=&v"(x), "=&v"(y)
          : "v"(p), "s"(negw), "v"(row), "v"(j)
          : "v2", "v3", "s10", "s11");
    } else {
      x = (float)(p - w * row) + 0.5f;
      y = (float)j + 0.5f;
    }
    const float xr = (float)(p - w * row) + 0.5f, yr = (float)j + 0.5f;
    asm volatile("" : "+v"(x), "+v"(y));
    nbad += (x != xr || y != yr) ? 1u : 0u;
  }
  if (nbad) atomicAdd(&bad[tid & 63], nbad);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 60;
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  unsigned* bad; float* sink;
  CHECK(hipMalloc(&bad, 64 * sizeof(unsigned)));
  CHECK(hipMalloc(&sink, 4096 * 256 * sizeof(float)));
  for (int mode = 0; mode < 4; ++mode) {     // 0: scalar alone, 1: packed alone, 2: scalar + hog, 3: packed + hog
    const int packed = mode & 1, with_hog = mode >> 1;
    CHECK(hipMemsetAsync(bad, 0, 64 * sizeof(unsigned), s0));
    CHECK(hipStreamSynchronize(s0));
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(victim, dim3(38 * 10), dim3(256), 0, s0, bad, 400, 160, packed);   // the grid of k_raster at 38 hypotheses
      if (with_hog) hipLaunchKernelGGL(hog, dim3(2048), dim3(256), 0, s1, sink, 600);
      CHECK(hipStreamSynchronize(s0));
      CHECK(hipStreamSynchronize(s1));
    }
    std::vector<unsigned> h(64);
    CHECK(hipMemcpy(h.data(), bad, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
    unsigned long long tot = 0, q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
    printf("%-7s %-9s: %llu wrong (x, y) pairs; by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %llu / %llu / %llu / %llu\n",
           packed ? "packed" : "scalar", with_hog ? "+ MFMA" : "alone", tot, q[0], q[1], q[2], q[3]);
  }
  return 0;
}
