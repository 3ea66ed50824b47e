// What a workgroup-wide hand-over costs on gfx950, for the single-workgroup Cholesky of k_ba_solve_chain (ba.hip):
//   1. s_barrier with W waves (an otherwise empty loop)
//   2. a flag in LDS: wave 0 writes a counter, the other waves spin on ds_read until they see it (no barrier)
//   3. the cost of one dependent VALU instruction, of v_readlane -> VALU, and of an LDS write -> read round trip for ONE wave when the
//      other waves are idle / are hammering the LDS
//   hipcc --offload-arch=gfx950 -O3 -o wave_sync wave_sync.hip && ./wave_sync
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_barrier(unsigned long long* out, int iters) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

__global__ void k_flag(unsigned long long* out, int iters) {
  __shared__ volatile int flag, ack[16];
  const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (threadIdx.x == 0) flag = 0;
  if (threadIdx.x < 16) ack[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 1; i <= iters; i++) {
    if (wv == 0) {
      // wait until every consumer acknowledged step i - 1, then publish step i
      for (int w = 1; w < nw; w++) while (ack[w] < i - 1) __builtin_amdgcn_s_sleep(1);
      if ((threadIdx.x & 63) == 0) flag = i;
    } else {
      while (flag < i) __builtin_amdgcn_s_sleep(1);
      if ((threadIdx.x & 63) == 0) ack[wv] = i;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

__global__ void k_chain(unsigned long long* out, float* sink, int iters, int noisy) {
  __shared__ float buf[4096];
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (wv != 0) {                                   // the other waves: idle, or streaming LDS reads
    float acc = 0;
    if (noisy) for (int i = 0; i < iters * 8; i++) acc += buf[(threadIdx.x * 4 + i * 64) & 4095];
    sink[threadIdx.x] = acc;
    return;
  }
  float x = sink[ln] + 1.0f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters / 16; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) x = __builtin_fmaf(x, 1.0001f, 0.5f);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[1] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters / 16; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) x = x * 1.0001f + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 7));
  }
  t1 = __builtin_readcyclecounter();
  out[2] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters / 16; i++)
#pragma unroll
  for (int u = 0; u < 16; u++) {
    buf[ln] = x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    x += buf[(ln + 1) & 63];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  t1 = __builtin_readcyclecounter();
  out[3] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters / 16; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) x = __frsqrt_rn(x + 2.0f);
  }
  t1 = __builtin_readcyclecounter();
  out[4] = t1 - t0;
  float y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters / 16; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) { y0 = __builtin_fmaf(y0, 1.0001f, 0.5f); y1 = __builtin_fmaf(y1, 1.0001f, 0.5f); y2 = __builtin_fmaf(y2, 1.0001f, 0.5f); y3 = __builtin_fmaf(y3, 1.0001f, 0.5f); }
  }
  t1 = __builtin_readcyclecounter();
  out[5] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) asm volatile("s_nop 0");
  t1 = __builtin_readcyclecounter();
  out[6] = t1 - t0;
  sink[ln] = x + y0 + y1 + y2 + y3;
}

int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 64); hipMalloc(&sink, 4096 * 4); hipMemset(sink, 0, 4096 * 4);
  unsigned long long h[8];
  const int it = 2000;
  for (int threads : {64, 256, 512, 1024}) {
    hipLaunchKernelGGL(k_barrier, dim3(1), dim3(threads), 0, 0, d, it);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("s_barrier, %2d waves: %.1f cycles per barrier\n", threads / 64, (double)h[0] / it);
  }
  for (int threads : {128, 256, 512, 1024}) {
    hipLaunchKernelGGL(k_flag, dim3(1), dim3(threads), 0, 0, d, it);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("LDS flag + acknowledgements, %2d waves: %.1f cycles per hand-over\n", threads / 64, (double)h[0] / it);
  }
  for (int noisy : {0, 1}) {
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(1024), 0, 0, d, sink, it, noisy);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("one wave, the other 15 %s: dependent v_fma %.1f, v_readlane + v_fma %.1f, LDS write -> read %.1f, v_rsq + v_add %.1f, independent v_fma (4 chains) %.1f, an empty loop iteration %.1f cycles\n",
           noisy ? "streaming LDS reads" : "idle", (double)h[1] / it, (double)h[2] / it, (double)h[3] / it, (double)h[4] / it, (double)h[5] / it, (double)h[6] / it);
  }
  return 0;
}
