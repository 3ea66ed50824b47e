// The locality plan's ordering step as a device function: corr.hip wraps it into corr_order_kernel, ba.hip runs it as the
// workgroups 1 .. G of k_prepare_and_order (next to the BA's index preparation in workgroup 0: latency-bound kernels that do
// not depend on each other).
#pragma once
#include "corr_tile.h"
#include <stdlib.h>

namespace devo {

constexpr int ORDER_THREADS = 1024;
constexpr int ORDER_MAXBINS = CORR_ORDER_MAXBINS;
constexpr int ORDER_MAX_WG = 64;

// Workgroups of the ordering step: about 1024 edges of their own each.
inline int corr_order_workgroups(long long BE, long long nbins) {
  static const int per_wg = [] { const char* e = getenv("DEVO_ORDER_EDGES_PER_WG"); const int v = e ? atoi(e) : 0; return v >= 64 ? v : 1024; }();   // (tuning switch)
  long long g = BE / per_wg;
  if (g > ORDER_MAX_WG) g = ORDER_MAX_WG;
  if (g > nbins) g = nbins;
  return g < 1 ? 1 : (int)g;
}

// Counting sort of the edges by plan bin, the heavy list first, by G workgroups that exchange NOTHING: workgroup g owns the bins
// [g nbins / G, (g + 1) nbins / G).  It reads ALL the bins (G x 4 BE bytes out of the L2), counts the edges of lower bins — heavy
// ones are bin -1 — which is the slot its own range starts at, and counting-sorts its own edges through LDS counters.  What this
// buys: an LDS atomic retires one lane per cycle, and the sort needs 2 BE of them — 18 µs of the former single-workgroup kernel at
// cfg2's 21 600 edges, 250 µs at the stress configuration's 262 144; here every compute unit does 2 BE / G.
// CACHE > 0: the ceil(BE / 1024) <= CACHE bins of a thread are loaded at once into registers (one round trip to memory instead of
// one per loop iteration and pass); CACHE == 0: any BE, bins re-read by both passes.
template <int CACHE>
__device__ __forceinline__ void corr_order_body(const int* __restrict__ bins, int BE, int nbins, int* __restrict__ order, int g, int G,
                                                bool starts = false) {     // starts: a GROUP plan — every bin's first slot goes into the plan's tail
  __shared__ int s_cnt[ORDER_MAXBINS + 1];
  __shared__ int s_base[2];                                 // [0] = edges below this workgroup's range, [1] = heavy cursor (g == 0)
  constexpr bool CACHED = CACHE > 0;
  const int lane = threadIdx.x & 63;
  const int b0 = (int)((long long)g * nbins / G), b1 = (int)((long long)(g + 1) * nbins / G), nown = b1 - b0;
  int breg[CACHED ? CACHE : 1];
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < (CACHED ? CACHE : 1); i++) {
      const int be = threadIdx.x + ORDER_THREADS * i;
      breg[i] = be < BE ? bins[be] : 0x7fffffff;
    }
  }
  auto bin_at = [&](int i) -> int {
    if (CACHED) return breg[i];
    const int be = threadIdx.x + ORDER_THREADS * i;
    return be < BE ? bins[be] : 0x7fffffff;
  };
  const int iters = CACHED ? CACHE : (BE + ORDER_THREADS - 1) / ORDER_THREADS;   // block-uniform (ballots below)
  for (int i = threadIdx.x; i < nown; i += ORDER_THREADS) s_cnt[i] = 0;
  if (threadIdx.x < 2) s_base[threadIdx.x] = 0;
  __syncthreads();
  int below = 0;
  // CACHE == 0 (edge lists beyond 32 K per thread-column: the stress configuration's 262 144 edges): the bins are read in batches of UB
  // loads in flight per thread — one at a time, each of a thread's 256 reads waited for its round trip: 183 us for the two passes
  constexpr int UB = 16;
  auto count_one = [&](int bin) {
    below += (bin < b0) ? 1 : 0;
    if (bin >= b0 && bin < b1) atomicAdd(&s_cnt[bin - b0], 1);
  };
  if constexpr (CACHED) {
#pragma unroll
    for (int i = 0; i < iters; i++) count_one(bin_at(i));
  } else {
    for (int i0 = 0; i0 < iters; i0 += UB) {
      int v[UB];
#pragma unroll
      for (int j = 0; j < UB; j++) v[j] = (i0 + j < iters) ? bin_at(i0 + j) : 0x7fffffff;
#pragma unroll
      for (int j = 0; j < UB; j++) count_one(v[j]);
    }
  }
  below = wave_inclusive_sum(below);
  if (lane == 63 && below != 0) atomicAdd(&s_base[0], below);
  __syncthreads();
  if (threadIdx.x < 64) {                                   // exclusive scan of the own bins by one wave, starting at the base slot
    int carry = s_base[0];
    for (int base = 0; base < nown; base += 64) {
      const int i = base + threadIdx.x;
      const int v = (i < nown) ? s_cnt[i] : 0;
      const int x = wave_inclusive_sum(v);
      if (i < nown) {
        s_cnt[i] = carry + x - v;
        if (starts) order[2 * BE + 2 + b0 + i] = carry + x - v;
      }
      // the last bin of all is the DEAD class of the pyramid plan (corr_plan_bin): its size goes behind the scratch half
      if (g == G - 1 && i == nown - 1) { order[2 * BE + 1] = v; if (starts) order[2 * BE + 2 + nbins] = BE; }
      carry += __builtin_amdgcn_readlane(x, 63);
    }
  }
  __syncthreads();
  auto place_one = [&](int i, int bin) {
    const int be = threadIdx.x + ORDER_THREADS * i;
    if (g == 0) {                                           // (block-uniform) the heavy list, in front of bin 0
      const unsigned long long hv = __ballot(bin < 0);
      int hbase = 0;
      if (hv != 0ull && lane == 0) hbase = atomicAdd(&s_base[1], __popcll(hv));
      hbase = __shfl(hbase, 0);
      if (bin < 0) order[hbase + __popcll(hv & ((1ull << lane) - 1ull))] = be;
    }
    if (bin >= b0 && bin < b1) order[atomicAdd(&s_cnt[bin - b0], 1)] = be;
  };
  if constexpr (CACHED) {
#pragma unroll
    for (int i = 0; i < iters; i++) place_one(i, bin_at(i));
  } else {
    for (int i0 = 0; i0 < iters; i0 += UB) {                // (iters is block-uniform: the ballots see whole waves; slots past BE carry 0x7fffffff)
      int v[UB];
#pragma unroll
      for (int j = 0; j < UB; j++) v[j] = (i0 + j < iters) ? bin_at(i0 + j) : 0x7fffffff;
#pragma unroll
      for (int j = 0; j < UB; j++) if (i0 + j < iters) place_one(i0 + j, v[j]);
    }
  }
  if (g == 0 && threadIdx.x == 0) order[BE] = s_base[0];   // b0 == 0: the edges below bin 0 are the heavy ones
}

}  // namespace devo
