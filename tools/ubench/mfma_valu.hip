// Micro-benchmark (debug tool, not part of the library): do v_mfma_f32_16x16x32_f16 streams and ordinary vector instructions of
// DIFFERENT waves on one SIMD overlap?  One workgroup per CU of 4 * WPS waves (wave w sits on SIMD w % 4): the first `nm` waves of every SIMD
// run a chain-free MFMA stream (6 accumulators), the other waves an FMA stream of independent registers; each kind alone, then together.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o tools/ubench/mfma_valu && tools/ubench/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k(float* out, int iters, int n_mfma_waves, int n_valu_waves, long long* cyc) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = wave >> 2;                       // this wave's index among the waves of its SIMD
  const bool do_mfma = slot < n_mfma_waves, do_valu = !do_mfma && slot < n_mfma_waves + n_valu_waves;
  h8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (_Float16)(lane * 0.001f + j); b[j] = (_Float16)(j * 0.5f - lane * 0.002f); }
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
  float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = lane + 4, v5 = lane + 5, v6 = lane + 6, v7 = lane + 7;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (do_mfma) {
    for (int it = 0; it < iters; it++) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c5, 0, 0, 0);
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 3; u++) {
        v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f); v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
        v4 = __builtin_fmaf(v4, 1.0001f, 0.5f); v5 = __builtin_fmaf(v5, 1.0001f, 0.5f); v6 = __builtin_fmaf(v6, 1.0001f, 0.5f); v7 = __builtin_fmaf(v7, 1.0001f, 0.5f);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 2000;
  printf("per iteration: 6 MFMAs (16x16x32 f16) per MFMA wave, 24 FMAs per vector wave; cycles per iteration of the slowest wave of each kind\n");
  for (int wps = 1; wps <= 4; wps++)
    for (int nm = 0; nm <= wps; nm++) {
      const int nv = wps - nm;
      hipMemset(cyc, 0, 64 * 8);
      hipLaunchKernelGGL(k, dim3(256), dim3(256 * wps), 0, 0, out, iters, nm, nv, cyc);
      hipDeviceSynchronize();
      long long h[16]; hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
      long long tm = 0, tv = 0;
      for (int w = 0; w < 4 * wps; w++) { if ((w >> 2) < nm) tm = h[w] > tm ? h[w] : tm; else tv = h[w] > tv ? h[w] : tv; }
      printf("waves per SIMD %d: %d MFMA + %d vector:  MFMA waves %7.1f cycles/iter (%5.1f per MFMA)   vector waves %7.1f cycles/iter (%4.1f per FMA)\n", wps, nm, nv,
             (double)tm / iters, (double)tm / iters / 6, (double)tv / iters, (double)tv / iters / 24);
    }
  return 0;
}
