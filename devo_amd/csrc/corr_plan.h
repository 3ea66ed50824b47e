// The locality plan's ordering step as a device function: corr.hip wraps it into corr_order_kernel, ba.hip runs it as the
// second workgroup of k_prepare_and_order (next to the BA's index preparation: both are single-workgroup, latency-bound
// kernels that do not depend on each other).
#pragma once
#include "corr_tile.h"

namespace devo {

constexpr int ORDER_THREADS = 1024;
constexpr int ORDER_MAXBINS = CORR_ORDER_MAXBINS;

// One workgroup: LDS counting sort of the bins; the heavy list first.  CACHE > 0: the ceil(BE / 1024) <= CACHE bins of a
// thread are loaded at once into registers (one round trip to memory instead of one per loop iteration and pass);
// CACHE == 0: any BE, bins re-read by both passes.
// `stage` (LDS, `stage_cap` ints, may be null / 0): the ordered list is assembled there and written out with coalesced stores —
// 20 000 scattered 4-byte stores from ONE compute unit take longer than the whole sort.
template <int CACHE>
__device__ __forceinline__ void corr_order_body(const int* __restrict__ bins, int BE, int nbins, int* __restrict__ order,
                                                int* stage = nullptr, int stage_cap = 0) {
  __shared__ int s_cnt[ORDER_MAXBINS];
  __shared__ int s_heavy[2];                                // [0] = count (pass 1), [1] = cursor (pass 2)
  constexpr bool CACHED = CACHE > 0;
  const int lane = threadIdx.x & 63;
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
  for (int i = threadIdx.x; i < nbins; i += ORDER_THREADS) s_cnt[i] = 0;
  if (threadIdx.x < 2) s_heavy[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < iters; i++) {
    const int bin = bin_at(i);
    const unsigned long long hv = __ballot(bin < 0);
    if (bin >= 0 && bin < nbins) atomicAdd(&s_cnt[bin], 1);
    if (hv != 0ull && lane == 0) atomicAdd(&s_heavy[0], __popcll(hv));
  }
  __syncthreads();
  const int n_heavy = s_heavy[0];
  if (threadIdx.x < 64) {                                   // exclusive scan of the bins by one wave, starting after the heavy list
    int carry = n_heavy;
    for (int base = 0; base < nbins; base += 64) {
      const int i = base + threadIdx.x;
      const int v = (i < nbins) ? s_cnt[i] : 0;
      const int x = wave_inclusive_sum(v);
      if (i < nbins) s_cnt[i] = carry + x - v;
      carry += __builtin_amdgcn_readlane(x, 63);
    }
  }
  __syncthreads();
  const bool staged = stage != nullptr && BE <= stage_cap;
  int* dst = staged ? stage : order;
#pragma unroll
  for (int i = 0; i < iters; i++) {
    const int be = threadIdx.x + ORDER_THREADS * i;
    const int bin = bin_at(i);
    const unsigned long long hv = __ballot(bin < 0);
    int hbase = 0;
    if (hv != 0ull && lane == 0) hbase = atomicAdd(&s_heavy[1], __popcll(hv));
    hbase = __shfl(hbase, 0);
    if (bin < 0) dst[hbase + __popcll(hv & ((1ull << lane) - 1ull))] = be;
    else if (bin < nbins) dst[atomicAdd(&s_cnt[bin], 1)] = be;
  }
  if (staged) {
    __syncthreads();
    for (int i = threadIdx.x; i < BE; i += ORDER_THREADS) order[i] = stage[i];
  }
  if (threadIdx.x == 0) order[BE] = n_heavy;
}

}  // namespace devo
