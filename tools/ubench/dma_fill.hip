// LDS-DMA fill-rate microbenchmark (gfx950): every CU runs ONE 512-thread workgroup whose 8 waves keep `inflight` KB of
// global_load_lds_dwordx4 requests outstanding, region-shaped (runs of `run` positions x 16 B, row stride `rowstride` B,
// 4 planes `plane` B apart), walking over a working set of `ws` bytes.  Prints bytes / cycle / CU and TB/s.
//   hipcc --offload-arch=gfx950 -O3 -o dma_fill dma_fill.hip && ./dma_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_off) : "memory");
}
template <int MODE>   // 0: LDS-DMA, 1: global_load_dwordx4 -> registers -> ds_write_b128
__global__ __launch_bounds__(1024) void k(const char* __restrict__ src, size_t ws, int iters, int batch, int run, int rowstride,
                                         size_t plane, unsigned long long* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[128 * 1024];   // (16 waves x 4 planes x 1 KB x 2 slots)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  unsigned seed = blockIdx.x * 2654435761u + wave * 40503u;
  const size_t lane_off = (size_t)(lane / run) * rowstride + (size_t)(lane % run) * 16;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    for (int b = 0; b < batch; b++) {                       // `batch` x 4 planes x 1 KB per wave in flight
      seed = seed * 1664525u + 1013904223u;
      const size_t base = ((size_t)(seed >> 8) * 4096) % (ws - 4 * plane - 64 * rowstride);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const char* g = src + base + q * plane + lane_off;
        const unsigned lo = lds0 + (unsigned)((wave * 4 + q) * 1024 + (b & 1) * 65536);
        if (MODE == 0) dma16(g, lo);
        else { const uint4 v = *reinterpret_cast<const uint4*>(g); *reinterpret_cast<uint4*>(lds + (lo - lds0) + lane * 16) = v; }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (lds[seed & 1023] == 77);
}
int main() {
  const size_t total = 512ull << 20;
  char* src; hipMalloc(&src, total); hipMemset(src, 1, total);
  unsigned long long* out; hipMalloc(&out, 256 * 8);
  const int iters = 200;
  // second table: how the rate depends on the number of waves that issue (one workgroup per CU)
  for (int nwaves : {8, 1, 2, 4, 16})
  for (int mode = 0; mode < (nwaves == 8 ? 2 : 1); mode++)
    for (size_t ws : {(size_t)2 << 20, (size_t)24 << 20, (size_t)160 << 20, (size_t)500 << 20})
      for (int batch : {1, 2, 4, 8}) {
        if (nwaves != 8 && (batch == 1 || batch == 8 || ws == ((size_t)160 << 20))) continue;
        const int run = 20, rowstride = 2560; const size_t plane = 307200;
        for (int rep = 0; rep < 2; rep++) {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * nwaves), 0, 0, src, ws, iters, batch, run, rowstride, plane, out);
          else hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * nwaves), 0, 0, src, ws, iters, batch, run, rowstride, plane, out);
          hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(256); hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
        double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= 256;
        const double bytes = (double)iters * batch * 4 * 1024 * nwaves;
        printf("%s %2d waves, ws %4zu MB, %2d KB in flight per wave: %.1f B/cycle/CU (%.0f cycles per batch)\n", mode ? "reg+ds_write" : "lds-dma     ", nwaves,
               ws >> 20, batch * 4, bytes / cyc, cyc / iters);
      }
  return 0;
}
