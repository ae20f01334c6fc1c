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

// MODE 0: M, 1: V, 2: MV interleaved, 3: split by wave parity (wid & 1), 4: split by SIMD partner ((wid >> 2) & 1), 5: as 4 with the
// MFMA waves at s_setprio 1.  Round 5 correction: a workgroup's waves go to the SIMDs round-robin, so waves w and w + 4 share a SIMD and
// MODE 3 at 8 waves puts two MFMA waves on the even SIMDs and two VALU waves on the odd ones -- it never co-locates the two kinds.  MODE 4
// does: waves 0-3 (one per SIMD) multiply, waves 4-7 (their SIMD partners) run the vector body.
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
  float16_ acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (lane + i);
  float m = 1.0001f, c = 0.0001f;
  asm volatile("" : "+v"(m), "+v"(c));
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wid & 1) == 0) || (MODE >= 4 && ((wid >> 2) & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wid & 1) == 1) || (MODE >= 4 && ((wid >> 2) & 1) == 1);
  if (MODE == 5 && do_m) __builtin_amdgcn_s_setprio(1);
  // Round 5: both bodies are `asm volatile` statements, so the instruction stream IS the source order.  (Round 4's bodies were builtins:
  // hipcc packed the fp32 FMAs into half-rate v_pk_fma_f32, folded the chains and moved the MFMAs of an iteration together, so its
  // "interleaved" mode measured something else.)  Independent registers on both sides: no hazard between an MFMA and a filler; an
  // accumulate chain needs no wait state between its MFMAs.
#define P_MFMA(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b))
#define P_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c))
  asm volatile("s_nop 7" ::: "memory");          // a / b / acc were just written by VALU code
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        P_MFMA(j & 3);
#pragma unroll
        for (int q = (NV * j) / 16; q < (NV * (j + 1)) / 16; ++q) P_FMA(q & 7);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int j = 0; j < 16; ++j) P_MFMA(j & 3);
      }
      if (do_v) {
#pragma unroll
        for (int q = 0; q < NV; ++q) P_FMA(q & 7);
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // an MFMA's D is readable 12 states after issue: nothing pads an asm MFMA
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
    // fillers per MFMA gap: 1, 2, 4, 5, 6, 8, 16 (the guide's table: <= 5 single-issue instructions hide in a 32-clk gap)
    show("VALU only, NV=16", run<1, 16>(th, out));
    show("interleaved, NV=16  (1 per MFMA)", run<2, 16>(th, out));
    show("VALU only, NV=32", run<1, 32>(th, out));
    show("interleaved, NV=32  (2 per MFMA)", run<2, 32>(th, out));
    show("VALU only, NV=64", run<1, 64>(th, out));
    show("interleaved, NV=64  (4 per MFMA)", run<2, 64>(th, out));
    show("VALU only, NV=80", run<1, 80>(th, out));
    show("interleaved, NV=80  (5 per MFMA)", run<2, 80>(th, out));
    show("VALU only, NV=96", run<1, 96>(th, out));
    show("interleaved, NV=96  (6 per MFMA)", run<2, 96>(th, out));
    show("VALU only, NV=128", run<1, 128>(th, out));
    show("interleaved, NV=128 (8 per MFMA)", run<2, 128>(th, out));
    show("VALU only, NV=256", run<1, 256>(th, out));
    show("interleaved, NV=256 (16 per MFMA)", run<2, 256>(th, out));
    show("split by wave parity, NV=128", run<3, 128>(th, out));
    show("split by wave parity, NV=256", run<3, 256>(th, out));
    if (waves == 2) {
      // the two kinds of wave on the SAME SIMD: waves 0-3 multiply, their partners 4-7 run the vector body
      show("SIMD partners M | V, NV=64", run<4, 64>(th, out));
      show("SIMD partners M | V, NV=128", run<4, 128>(th, out));
      show("SIMD partners M | V, NV=192", run<4, 192>(th, out));
      show("SIMD partners M | V, NV=256", run<4, 256>(th, out));
      show("SIMD partners M | V, NV=512", run<4, 512>(th, out));
      show("SIMD partners, M at prio 1, NV=128", run<5, 128>(th, out));
      show("SIMD partners, M at prio 1, NV=256", run<5, 256>(th, out));
      show("SIMD partners, M at prio 1, NV=512", run<5, 512>(th, out));
    }
  }
  return 0;
}
