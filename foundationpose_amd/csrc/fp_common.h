// Shared helpers for libfp_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/fp_amd.h"

void fp_set_error(const char* fmt, ...);

#define FP_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      fp_set_error(__VA_ARGS__);         \
      return FP_ERR_INVALID_ARG;         \
    }                                    \
  } while (0)

#define FP_CHECK_LAUNCH(name)                                             \
  do {                                                                    \
    hipError_t e_ = hipGetLastError();                                    \
    if (e_ != hipSuccess) {                                               \
      fp_set_error("%s launch failed: %s", name, hipGetErrorString(e_));  \
      return FP_ERR_LAUNCH;                                               \
    }                                                                     \
  } while (0)

struct fp_mesh {
  const float* pos;
  const float* nrm;
  const int32_t* faces;
  const float* uv;
  const int32_t* uv_idx;
  const float* tex;
  const float* vcol;
  int V, T, Ht, Wt;
};

struct fp_k9 { float v[9]; };
struct fp_k9d { double v[9]; };

static inline int fp_cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device).
// Use as FP_SET_MAX_LDS(kernel_symbol, bytes) right before the launch.
#define FP_SET_MAX_LDS(kernel, bytes)                                                                              \
  do {                                                                                                             \
    static unsigned long long done_ = 0ull;                                                                        \
    int dev_ = 0;                                                                                                  \
    (void)hipGetDevice(&dev_);                                                                                     \
    if (!((done_ >> (dev_ & 63)) & 1ull)) {                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (bytes));                                                                          \
      done_ |= 1ull << (dev_ & 63);                                                                                \
    }                                                                                                              \
  } while (0)
