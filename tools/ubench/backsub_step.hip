// The blocked back-substitution of k_ba_solve_chain (ba.hip) alone in one wave: L^T x = z, 6 x 6 blocks, z in registers (lane r = row r),
// the block's inverse and the lane's six entries of L^T fetched a step ahead.  Cycles per block step for the loop as the kernel has it
// (runtime N) and fully unrolled (compile-time N): what the loop structure, the scalar lane indices and the branches cost.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o backsub_step backsub_step.hip && ./backsub_step
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ float lane_value(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

template <int NFIX>
__global__ void k_backsub(unsigned long long* out, float* sink, int Nrt, int reps) {
  constexpr int LD = 100;
  __shared__ float A[85 * LD];
  __shared__ float Li[14 * 36];
  __shared__ float xs[96];
  const int tid = threadIdx.x;
  for (int i = tid; i < 85 * LD; i += 64) A[i] = 0.001f * ((i * 7) % 13);
  for (int i = tid; i < 14 * 36; i += 64) Li[i] = ((i % 36) % 7 == 0) ? 0.5f : 0.002f * (i % 5);
  __syncthreads();
  const int N = NFIX ? NFIX : Nrt, n6 = 6 * N;
  const unsigned long long t0 = __builtin_readcyclecounter();
  float acc = 0.0f;
  for (int rep = 0; rep < reps; rep++) {
    const int lr0 = min(tid, n6 - 1), lr1 = min(tid + 64, n6 - 1), lc = min(tid, 5);
    float z0 = (tid < n6) ? A[tid * LD + n6] : 0.0f, z1 = (tid + 64 < n6) ? A[(tid + 64) * LD + n6] : 0.0f;
    struct StepOps { float li[6], a0[6]; };
    auto fetch = [&](int jb, StepOps& o) {
      const float* Lb = Li + jb * 36 + lc;
      const float* rowp = A + lr0 * LD + 6 * jb;
#pragma unroll
      for (int k = 0; k < 6; k++) { o.li[k] = Lb[k * 6]; o.a0[k] = rowp[k]; }
    };
    auto step = [&](int jb, const StepOps& o) {
      const int j0 = 6 * jb;
      float zb[6], xb[6];
#pragma unroll
      for (int k = 0; k < 6; k++) { const int r = j0 + k; zb[k] = lane_value((r >= 64) ? z1 : z0, r & 63); }
      float xc = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; k++) xc += o.li[k] * zb[k];
      if (tid < 6) xs[j0 + tid] = xc;
#pragma unroll
      for (int k = 0; k < 6; k++) xb[k] = lane_value(xc, k);
      float v0 = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; k++) v0 += o.a0[k] * xb[k];
      if (tid < j0) z0 -= v0;
      if (j0 > 64) {
        const float* rowp = A + lr1 * LD + j0;
        float v1 = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; k++) v1 += rowp[k] * xb[k];
        if (tid + 64 < j0) z1 -= v1;
      }
    };
    if (NFIX) {
      StepOps o[2];
      fetch(N - 1, o[0]);
#pragma unroll
      for (int jb = NFIX - 1; jb >= 0; jb--) {
        if (jb > 0) fetch(jb - 1, o[(NFIX - jb) & 1]);
        step(jb, o[(NFIX - 1 - jb) & 1]);
      }
    } else {
      StepOps oa, ob;
      int jb = N - 1;
      fetch(jb, oa);
      while (true) {
        fetch(max(jb - 1, 0), ob);
        step(jb, oa);
        if (--jb < 0) break;
        fetch(max(jb - 1, 0), oa);
        step(jb, ob);
        if (--jb < 0) break;
      }
    }
    wave_lds_sync();
    acc += z0 + z1 + xs[tid];
    A[tid * LD + n6] = acc * 1e-6f;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[0] = t1 - t0;
  sink[tid] = acc;
}

int main() {
  unsigned long long* d; float* sink;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 1024);
  unsigned long long h[8];
  const int reps = 200;
  hipLaunchKernelGGL(k_backsub<0>, dim3(1), dim3(64), 0, 0, d, sink, 14, reps);
  (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("runtime N = 14 (the kernel's loop): %.0f cycles per block step\n", (double)h[0] / reps / 14);
  hipLaunchKernelGGL(k_backsub<14>, dim3(1), dim3(64), 0, 0, d, sink, 14, reps);
  (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("compile-time N = 14 (fully unrolled): %.0f cycles per block step\n", (double)h[0] / reps / 14);
  return 0;
}
