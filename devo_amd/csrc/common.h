// Shared host-side helpers for the C-ABI entry points of libdevo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

#define DEVO_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      devo::set_error(__VA_ARGS__);        \
      return DEVO_ERR_ARG;                 \
    }                                      \
  } while (0)

}  // namespace devo
