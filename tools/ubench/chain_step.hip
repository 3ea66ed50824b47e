// The serial part of one block step of k_ba_solve_chain (ba.hip), alone in one wave (the other 15 waves idle or streaming LDS reads):
// six rows solved against L_bb, the next diagonal block formed, gathered, factored, stored.  Cycles per step for each variant of the
// gather (v_readlane / LDS) and for the parts on their own.
//   hipcc --offload-arch=gfx950 -O3 -o chain_step chain_step.hip && ./chain_step
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool chol6(float L[6][6], float inv[6]) {
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    float d = L[c][c];
#pragma unroll
    for (int k = 0; k < c; k++) d -= L[c][k] * L[c][k];
    if (!(d > 0.0f)) ok = false;
    inv[c] = __frsqrt_rn(d);
    L[c][c] = d * inv[c];
#pragma unroll
    for (int a = c + 1; a < 6; a++) {
      float v = L[a][c];
#pragma unroll
      for (int k = 0; k < c; k++) v -= L[a][k] * L[c][k];
      L[a][c] = v * inv[c];
    }
  }
  return ok;
}

template <int VARIANT>
__global__ void k_step(unsigned long long* out, float* sink, int iters, int noisy) {
  constexpr int LD = 100, N = 14;
  __shared__ float A[85 * LD];
  __shared__ float Ld[N * 36];
  __shared__ __attribute__((aligned(16))) float s_blk[64];
  __shared__ int s_fail;
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 85 * LD; i += blockDim.x) A[i] = ((i / LD) == (i % LD)) ? 4.0f : 0.01f * ((i * 7) % 13);
  for (int i = threadIdx.x; i < N * 36; i += blockDim.x) Ld[i] = ((i % 36) % 7 == 0) ? 0.5f : 0.02f * (i % 5);
  __syncthreads();
  if (wv != 0) {
    float acc = 0;
    if (noisy) for (int i = 0; i < iters * 40; i++) acc += A[(threadIdx.x * 4 + i * 64) % (85 * LD)];
    sink[threadIdx.x] = acc;
    return;
  }
  auto lane_value = [](float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); };
  const int la = ln < 36 ? ln / 6 : 5, lc = ln < 36 ? ln % 6 : 5;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    const int jb = it % (N - 1), j0 = 6 * jb, j1 = j0 + 6;
    const float* pr = A + (j1 + la) * LD + j0;
    const float* pc = A + (j1 + lc) * LD + j0;
    float d;
    if (VARIANT != 3) {
      const float dv = pr[6 + lc];
      float Lb[6][6], inv[6];
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = 0; c < a; c++) Lb[a][c] = Ld[jb * 36 + a * 6 + c];
        inv[a] = Ld[jb * 36 + a * 7];
      }
      f2 v[6], x[6];
#pragma unroll
      for (int q = 0; q < 6; q++) v[q] = f2{pr[q], pc[q]};
#pragma unroll
      for (int c = 0; c < 6; c++) {
        f2 t = v[c];
#pragma unroll
        for (int k = 0; k < c; k++) t -= x[k] * Lb[c][k];
        x[c] = t * inv[c];
      }
      float acc = 0.0f;
#pragma unroll
      for (int q = 0; q < 6; q++) acc += x[q].x * x[q].y;
      d = dv - acc;
    } else d = pr[6 + lc];                                       // variant 3: no row solves
    float L[6][6];
    if (VARIANT == 0 || VARIANT == 3) {
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c <= a; c++) L[a][c] = lane_value(d, a * 6 + c);
    } else if (VARIANT == 1) {                                  // through LDS, every lane reads the block
      s_blk[ln] = d;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c <= a; c++) L[a][c] = s_blk[a * 6 + c];
    } else {                                                    // variant 2: no gather at all (every lane factors garbage of its own)
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c <= a; c++) L[a][c] = d + (a == c ? 4.0f : 0.01f * (a + c));
    }
    float inv[6];
    const bool ok = chol6(L, inv);
    if (ln == 0) {
      if (!ok) s_fail = 1;
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c <= a; c++) Ld[(jb + 1) * 36 + a * 6 + c] = (a == c) ? inv[a] : L[a][c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[ln] = Ld[ln] + s_fail;
}

int main() {
  unsigned long long* d; float* sink;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 4096 * 4);
  unsigned long long h[8];
  const int it = 1300;
  const char* names[4] = {"v_readlane gather", "LDS gather", "no gather", "no row solves (readlane gather)"};
  for (int noisy : {0, 1})
    for (int v = 0; v < 4; v++) {
      // (one wave launches alone when the others would only idle: the barrier then counts one wave)
      const int threads = noisy ? 1024 : 64;
      if (v == 0) hipLaunchKernelGGL(k_step<0>, dim3(1), dim3(threads), 0, 0, d, sink, it, noisy);
      if (v == 1) hipLaunchKernelGGL(k_step<1>, dim3(1), dim3(threads), 0, 0, d, sink, it, noisy);
      if (v == 2) hipLaunchKernelGGL(k_step<2>, dim3(1), dim3(threads), 0, 0, d, sink, it, noisy);
      if (v == 3) hipLaunchKernelGGL(k_step<3>, dim3(1), dim3(threads), 0, 0, d, sink, it, noisy);
      (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      printf("%-34s %s: %.0f cycles per step\n", names[v], noisy ? "(15 waves streaming LDS reads)" : "(alone)", (double)h[0] / it);
    }
  return 0;
}
