// Stand-alone reproducer for the instruction scripts/repro_pk_f32/make_variants.py + run_variants.py isolated in k_raster
// (round 4): with every other packed instruction of the SLP build replaced by its scalar halves, ONE is enough for
// wrong lanes 48..63 next to an MFMA kernel of another stream:
//     v_pk_mul_f32 v[30:31], v[32:33], v[30:31] op_sel:[0,1] op_sel_hi:[1,0]
// -- an IN-PLACE packed multiply whose halves are CROSSED: the low result (written to v30) reads v31, the high result (written to
// v31) reads v30.  Architecturally every source is read before any result is written.
//   hipcc --offload-arch=gfx950 -O2 scripts/repro_pk_f32/repro2.hip -o /tmp/repro2 && /tmp/repro2 [launches]
// `victim` runs that instruction (and three controls: the same crossing into OTHER destination registers, the in-place form with
// straight halves, plain scalar multiplies) on per-lane data under a lane-dependent branch, and compares with v_mul_f32 results;
// `hog` is a register-resident MFMA loop on a second stream.
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

// the same loop with the accumulators in AGPRs (what rocBLAS / Tensile kernels and the library's implicit-GEMM kernels do; the
// attention kernel, which never triggered the fault, keeps its accumulators in VGPRs -- like `hog` above)
__global__ __launch_bounds__(256) void hog_agpr(float* out, int iters) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  float16_ c0 = {}, c1 = {}, c2 = {}, c3 = {};
  asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                 "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                 : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// form 0: scalar multiplies; 1: in place, crossed halves (the isolated instruction); 2: crossed halves, other destination;
// 3: in place, straight halves.  bad[lane] += iterations whose pair differs from the scalar result
__global__ __launch_bounds__(256) void victim(unsigned* bad, const float* __restrict__ data, int iters, int form) {
  const int tid = threadIdx.x;
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const int p = (tid + 256 * it + 97 * blockIdx.x) & 65535;
    const float a0 = data[p], a1 = data[p + 65536], b0 = data[p + 131072], b1 = data[p + 196608];
    if (a0 < 0.3f) continue;                      // divergent EXEC, as in the resolve loop (only covered pixels are shaded)
    float x, y;                                   // expected: crossed forms (a0 * b1, a1 * b0); straight form (a0 * b0, a1 * b1)
    if (form == 1) {
      asm volatile(
          "v_mov_b32 v4, %2\n v_mov_b32 v5, %3\n v_mov_b32 v2, %4\n v_mov_b32 v3, %5\n"
          "v_pk_mul_f32 v[2:3], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]\n"
          "v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n"
          : "=&v"(x), "=&v"(y) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v2", "v3", "v4", "v5");
    } else if (form == 2) {
      asm volatile(
          "v_mov_b32 v4, %2\n v_mov_b32 v5, %3\n v_mov_b32 v2, %4\n v_mov_b32 v3, %5\n"
          "v_pk_mul_f32 v[6:7], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]\n"
          "v_mov_b32 %0, v6\n v_mov_b32 %1, v7\n"
          : "=&v"(x), "=&v"(y) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v2", "v3", "v4", "v5", "v6", "v7");
    } else if (form == 3) {
      asm volatile(
          "v_mov_b32 v4, %2\n v_mov_b32 v5, %3\n v_mov_b32 v2, %4\n v_mov_b32 v3, %5\n"
          "v_pk_mul_f32 v[2:3], v[4:5], v[2:3]\n"
          "v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n"
          : "=&v"(x), "=&v"(y) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v2", "v3", "v4", "v5");
    } else if (form >= 4 && form <= 7) {
      // write-after-read: the instructions BEHIND the packed multiply overwrite its source registers (what follows site 9 in
      // k_raster: v_mov_b32 v32, v36 -- v32 is a source of the packed multiply in front of it).  form 4: immediately; 5, 6, 7: after
      // s_nop 0 / 1 / 3
#define WAR_BODY(NOP)                                                                                                   \
      asm volatile(                                                                                                     \
          "v_mov_b32 v4, %2\n v_mov_b32 v5, %3\n v_mov_b32 v2, %4\n v_mov_b32 v3, %5\n v_mov_b32 v8, 0x7fc00000\n"      \
          "v_pk_mul_f32 v[6:7], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]\n" NOP                                     \
          "v_mov_b32 v4, v8\n v_mov_b32 v5, v8\n v_mov_b32 v2, v8\n v_mov_b32 v3, v8\n"                                \
          "v_mov_b32 %0, v6\n v_mov_b32 %1, v7\n"                                                                      \
          : "=&v"(x), "=&v"(y) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v2", "v3", "v4", "v5", "v6", "v7", "v8")
      if (form == 4) { WAR_BODY(""); }
      else if (form == 5) { WAR_BODY("s_nop 0\n"); }
      else if (form == 6) { WAR_BODY("s_nop 1\n"); }
      else { WAR_BODY("s_nop 3\n"); }
    } else {
      x = a0 * b1; y = a1 * b0;
    }
    const float xr = form == 3 ? a0 * b0 : a0 * b1, yr = form == 3 ? a1 * b1 : a1 * b0;
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
  unsigned* bad; float* sink; float* data;
  CHECK(hipMalloc(&bad, 64 * sizeof(unsigned)));
  CHECK(hipMalloc(&sink, 4096 * 256 * sizeof(float)));
  std::vector<float> hdata(4 * 65536);
  unsigned st = 12345u;
  for (auto& v : hdata) { st = st * 1664525u + 1013904223u; v = (float)(st >> 8) * (1.0f / 16777216.0f) + 0.05f; }
  CHECK(hipMalloc(&data, hdata.size() * sizeof(float)));
  CHECK(hipMemcpy(data, hdata.data(), hdata.size() * sizeof(float), hipMemcpyHostToDevice));
  const char* names[8] = {"scalar v_mul_f32 x 2", "in place, crossed halves", "crossed halves, other dst", "in place, straight halves",
                          "sources overwritten behind", "... after s_nop 0", "... after s_nop 1", "... after s_nop 3"};
  const char* hogs[3] = {"alone", "+ MFMA (VGPR acc)", "+ MFMA (AGPR acc)"};
  for (int with_hog = 0; with_hog < 3; ++with_hog)
    for (int form = 0; form < 8; ++form) {
      CHECK(hipMemsetAsync(bad, 0, 64 * sizeof(unsigned), s0));
      CHECK(hipStreamSynchronize(s0));
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(victim, dim3(38 * 10), dim3(256), 0, s0, bad, data, 400, form);   // the grid of k_raster at 38 hypotheses
        if (with_hog == 1) hipLaunchKernelGGL(hog, dim3(2048), dim3(256), 0, s1, sink, 600);
        if (with_hog == 2) hipLaunchKernelGGL(hog_agpr, dim3(2048), dim3(256), 0, s1, sink, 600);
        CHECK(hipStreamSynchronize(s0));
        CHECK(hipStreamSynchronize(s1));
      }
      std::vector<unsigned> h(64);
      CHECK(hipMemcpy(h.data(), bad, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
      unsigned long long tot = 0, q[4] = {0, 0, 0, 0};
      for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
      printf("%-26s %-18s: %8llu wrong pairs; by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %llu / %llu / %llu / %llu\n", names[form],
             hogs[with_hog], tot, q[0], q[1], q[2], q[3]);
    }
  return 0;
}
