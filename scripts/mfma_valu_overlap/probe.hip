// Does VALU work overlap with MFMA work on a CDNA4 SIMD?  (round 4, after the attention experiments: DESIGN.md 3.35)
//   hipcc --offload-arch=gfx950 -O3 -pthread -o /tmp/mvprobe scripts/mfma_valu_overlap/probe.hip && /tmp/mvprobe
// One workgroup per CU, `WAVES` waves per SIMD; every wave runs ITER iterations of one of three bodies:
//   M : 16 independent v_mfma_f32_32x32x16_f16 (4 accumulator chains)                      -> 16 x 32 = 512 clk of matrix pipe
//   V : NV independent v_fma_f32 (8 chains)                                                 -> NV x 4 clk of VALU
//   MV: the same 16 MFMAs and NV FMAs in ONE instruction stream, one MFMA every NV / 16 VALU instructions
// and mode "split" gives the MFMA body to even waves and the VALU body to odd waves of a SIMD.  Every configuration runs ~0.25 s
// while a host thread samples the shader clock (amdgpu hwmon freq1_input of the device's PCI address): printed are ns per iteration,
// the mean clock, and their product = shader clocks per iteration -- the chip clocks down under MFMA load, so only clocks compare.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <glob.h>
#include <limits.h>
#include <string>
#include <thread>
#include <unistd.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16_ __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int NV>   // MODE 0: M, 1: V, 2: MV interleaved, 3: split by wave parity
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
  float16_ acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (lane + i);
  const float m = 1.0001f, c = 0.0001f;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wid & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wid & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV / 16; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], m, c);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 3], 0, 0, 0);
      }
      if (do_v) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], m, c);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static std::string g_freq_file;
static void find_freq_file() {
  char bus[64] = {0};
  CK(hipDeviceGetPCIBusId(bus, sizeof(bus), 0));
  for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
  glob_t g;
  if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", 0, nullptr, &g) == 0) {
    for (size_t i = 0; i < g.gl_pathc; ++i) {
      char real[PATH_MAX];
      if (realpath(g.gl_pathv[i], real) && strstr(real, bus)) g_freq_file = g.gl_pathv[i];
    }
    if (g_freq_file.empty() && g.gl_pathc) g_freq_file = g.gl_pathv[0];
    globfree(&g);
  }
}
static double read_mhz() {
  FILE* f = fopen(g_freq_file.c_str(), "r");
  if (!f) return 0;
  double v = 0;
  if (fscanf(f, "%lf", &v) != 1) v = 0;
  fclose(f);
  return v / 1e6;
}

struct Res { float ns; double mhz; };
template <int MODE, int NV>
Res run(int threads, float* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, 2000);
  CK(hipDeviceSynchronize());
  const int iters = 400000;                       // ~0.15-0.4 s
  std::atomic<bool> stop(false);
  double sum = 0; int n = 0;
  std::thread th([&] { usleep(20000); while (!stop.load()) { const double m = read_mhz(); if (m > 0) { sum += m; ++n; } usleep(2000); } });
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  stop.store(true);
  th.join();
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return Res{ms * 1e6f / iters, n ? sum / n : 0.0};
}
static void show(const char* name, Res r) { printf("   %-34s %7.0f ns  %5.0f MHz  %7.0f clk\n", name, r.ns, r.mhz, r.ns * r.mhz * 1e-3); }

int main() {
  float* out;
  CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  find_freq_file();
  printf("per iteration and wave: 16 MFMAs 32x32x16 (512 clk of matrix pipe), NV fp32 FMAs; clock from %s\n", g_freq_file.c_str());
  for (int waves = 1; waves <= 2; ++waves) {
    const int th = waves * 256;
    printf("%d wave(s) per SIMD\n", waves);
    show("MFMA only", run<0, 128>(th, out));
    show("VALU only, NV=128", run<1, 128>(th, out));
    show("interleaved in one stream, NV=128", run<2, 128>(th, out));
    show("split by wave parity, NV=128", run<3, 128>(th, out));
    show("VALU only, NV=256", run<1, 256>(th, out));
    show("interleaved in one stream, NV=256", run<2, 256>(th, out));
    show("split by wave parity, NV=256", run<3, 256>(th, out));
  }
  return 0;
}
