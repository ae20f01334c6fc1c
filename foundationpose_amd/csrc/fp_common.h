// Shared helpers for libfp_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
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
// Use as FP_SET_MAX_LDS(kernel_symbol, bytes) right before the launch.  Thread-safe (one host thread per device is the
// multi-GPU use): the per-device "done" bit is published with an atomic OR only AFTER the attribute call has returned, so a
// racing thread at worst repeats the (idempotent) call; devices >= 64 are not cached.  A failing call is reported.
#define FP_SET_MAX_LDS(kernel, bytes)                                                                                 \
  do {                                                                                                                \
    static std::atomic<unsigned long long> done_{0ull};                                                               \
    int dev_ = 0;                                                                                                     \
    (void)hipGetDevice(&dev_);                                                                                        \
    if (dev_ < 0 || dev_ >= 64 || !((done_.load(std::memory_order_acquire) >> dev_) & 1ull)) {                        \
      const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel),                               \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));                 \
      if (e_ != hipSuccess) {                                                                                         \
        fp_set_error("hipFuncSetAttribute(%s, %d bytes of LDS) failed on device %d: %s", #kernel, (int)(bytes), dev_, \
                     hipGetErrorString(e_));                                                                          \
        return FP_ERR_LAUNCH;                                                                                         \
      }                                                                                                               \
      if (dev_ >= 0 && dev_ < 64) done_.fetch_or(1ull << dev_, std::memory_order_release);                            \
    }                                                                                                                 \
  } while (0)
