// Shared host-side helpers for the C-ABI entry points of libdevo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/devo_hip.h"

namespace devo {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DEVO_ERR_LAUNCH;
  }
  return DEVO_OK;
}

inline int blocks_for(long long n, int threads, int cap = 1 << 20) {
  long long b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// "Have I raised this kernel's dynamic-LDS limit on the CURRENT device yet?"  hipFuncSetAttribute is per device: a process that uses a second
// GPU must set it there too (a launch of > 64 KB of LDS fails otherwise).  One bit per device ordinal (mod 64), set atomically; the caller
// does its hipFuncSetAttribute calls when this returns true.
struct PerDeviceOnce {
  std::atomic<unsigned long long> mask{0ull};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return true; }
    const unsigned long long bit = 1ull << (d & 63);
    return (mask.fetch_or(bit, std::memory_order_relaxed) & bit) == 0ull;
  }
};

#define DEVO_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      devo::set_error(__VA_ARGS__);        \
      return DEVO_ERR_ARG;                 \
    }                                      \
  } while (0)

}  // namespace devo
