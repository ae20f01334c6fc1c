// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on (MI355X: 8 XCDs x 32 CUs)?  For a few mask patterns:
// launch 4096 workgroups that spin a little and record (XCC_ID, HW_ID) -> number of workgroups per XCD and distinct CU ids.
//   hipcc --offload-arch=gfx950 -O2 scripts/cu_mask/probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <set>
#include <vector>

__global__ void k_probe(uint32_t* out, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int NB = 4096;
  uint32_t* d;
  CK(hipMalloc(&d, NB * 8));
  std::vector<uint32_t> h(NB * 2);
  struct Pat { const char* name; uint32_t w[8]; };
  Pat pats[] = {
    {"all 256", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
    {"bits 0..127", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
    {"bits 128..255", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}},
    {"bits i%8 < 4", {0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu}},
    {"bits i%8 >= 4", {0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u}},
    {"bits 0..31", {~0u, 0, 0, 0, 0, 0, 0, 0}},
    {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
  };
  for (auto& p : pats) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, p.w);
    if (e != hipSuccess) { printf("%-16s create failed: %s\n", p.name, hipGetErrorString(e)); continue; }
    CK(hipMemsetAsync(d, 0xFF, NB * 8, s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_probe, dim3(NB), dim3(256), 0, s, d, 20000);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost));
    int per_xcc[16] = {0};
    std::set<uint64_t> cus;
    for (int b = 0; b < NB; ++b) {
      const uint32_t xcc = h[2 * b] & 0xF, hw = h[2 * b + 1];
      per_xcc[xcc]++;
      // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] (gfx9 layout)
      cus.insert(((uint64_t)xcc << 32) | (hw & 0xFF00u));
    }
    printf("%-16s %.3f ms  distinct (xcc, se, sh, cu): %3zu   workgroups per XCC:", p.name, ms, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
    printf("\n");
    CK(hipStreamDestroy(s));
  }
  return 0;
}
