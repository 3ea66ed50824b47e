// Primitives the tile lookup kernel (corr_tile_mfma.h) rests on, checked on the GPU:
//   1. buffer_load_dwordx4 ... offen lds: lanes whose offset is out of range write ZEROS to LDS (no access); m0 base + 16 * lane
//   2. fp32 -> fp16 (hi, lo) split with v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16: hi + lo == x to 2^-22
//   3. v_mfma_f32_16x16x32_f16 operand / result layout: A lane l = row l % 16, k = 8 (l / 16) .. + 7; B lane l = column l % 16,
//      same k; D lane l = column l % 16, rows 4 (l / 16) .. + 3
//   hipcc --offload-arch=gfx950 -O3 -o tile_prims tile_prims.hip && ./tile_prims
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_dma(const float* src, float* out, unsigned nbytes) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[8192];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) reinterpret_cast<float*>(lds)[i] = -7.0f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  const unsigned voff = (lane % 3 == 1) ? 0x80000000u : (lane % 3 == 2 ? nbytes - 8 : lane * 16u);   // out of range / straddling the end / fine
  const unsigned soff = 256;
  const unsigned ldsoff = (unsigned)(uintptr_t)lds + 2048;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(ldsoff), "s"(soff) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 2048; i += 64) out[i] = reinterpret_cast<float*>(lds)[i];
}

__global__ void k_split(const float* src, float* out) {
  const float x0 = src[2 * threadIdx.x], x1 = src[2 * threadIdx.x + 1];
  unsigned hi, lo = 0;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
  asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x0));
  asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
  _Float16 h[2], l[2];
  __builtin_memcpy(h, &hi, 4); __builtin_memcpy(l, &lo, 4);
  out[4 * threadIdx.x + 0] = (float)h[0]; out[4 * threadIdx.x + 1] = (float)l[0];
  out[4 * threadIdx.x + 2] = (float)h[1]; out[4 * threadIdx.x + 3] = (float)l[1];
}

__global__ void k_mfma(const _Float16* A /*16x32 row-major*/, const _Float16* B /*32x16 (k, n) row-major*/, float* D /*16x16*/) {
  const int lane = threadIdx.x, r = lane & 15, kg = lane >> 4;
  h8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = A[r * 32 + 8 * kg + i]; b[i] = B[(8 * kg + i) * 16 + r]; }
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; i++) D[(4 * kg + i) * 16 + r] = acc[i];
}

template <int K32> __global__ void k_rate(float* out, int iters) {
  h8 a, b; for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)0.5f; }
  f4 acc[8]; for (int j = 0; j < 8; j++) acc[j] = f4{0, 0, 0, 0};
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (K32) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[j], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int j = 0; j < 8; j++) s += acc[j][0];
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = (float)(t1 - t0) / (float)(iters * 8); out[blockIdx.x * 2 + 1] = s; }
}

int main() {
  int bad = 0;
  { // 1
    const unsigned n = 4096; float *src, *out; hipMalloc(&src, n * 4 + 64); hipMalloc(&out, 2048 * 4);
    std::vector<float> h(n + 16); for (unsigned i = 0; i < n + 16; i++) h[i] = 1.0f + i; hipMemcpy(src, h.data(), (n + 16) * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, 0, src, out, n * 4);
    std::vector<float> o(2048); hipMemcpy(o.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
    int zero_ok = 1, data_ok = 1, untouched_ok = 1, straddle = 0;
    for (int lane = 0; lane < 64; lane++) for (int j = 0; j < 4; j++) {
      const float v = o[512 + lane * 4 + j];
      if (lane % 3 == 1) zero_ok &= (v == 0.0f);
      else if (lane % 3 == 0) data_ok &= (v == 1.0f + (256 / 4) + lane * 4 + j);
      else if (lane == 2) printf("   straddling lane word %d: %g\n", j, v), straddle++;
    }
    for (int i = 0; i < 512; i++) untouched_ok &= (o[i] == -7.0f);
    for (int i = 512 + 256; i < 2048; i++) untouched_ok &= (o[i] == -7.0f);
    printf("1. buffer_load_dwordx4 lds: out-of-range lanes write zeros: %s; in-range data at m0 + 16 lane (soffset added): %s; rest of LDS untouched: %s\n",
           zero_ok ? "yes" : "NO", data_ok ? "yes" : "NO", untouched_ok ? "yes" : "NO");
    bad += !zero_ok + !data_ok + !untouched_ok;
  }
  { // 2
    float *src, *out; hipMalloc(&src, 128 * 4); hipMalloc(&out, 256 * 4);
    std::vector<float> h(128); for (int i = 0; i < 128; i++) h[i] = (i % 2 ? -1.0f : 1.0f) * expf(-8.0f + 0.11f * i) * 1.2345678f; hipMemcpy(src, h.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, src, out);
    std::vector<float> o(256); hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
    double worst = 0; for (int i = 0; i < 128; i++) { const double e = fabs((double)o[2 * i] + (double)o[2 * i + 1] - (double)h[i]) / fabs((double)h[i]); if (e > worst) worst = e; }
    printf("2. fp16 hi + lo split: worst relative error %.3g (2^-22 = %.3g) over |x| in [%.2g, %.2g]\n", worst, pow(2.0, -22), fabs(h[0]), fabs(h[127]));
    bad += worst > 4.8e-7;
  }
  { // 3
    _Float16 *A, *B; float* D; hipMalloc(&A, 512 * 2); hipMalloc(&B, 512 * 2); hipMalloc(&D, 256 * 4);
    std::vector<_Float16> ha(512), hb(512); for (int i = 0; i < 512; i++) { ha[i] = (_Float16)((i * 37 % 17) - 8); hb[i] = (_Float16)((i * 11 % 13) - 6); }
    hipMemcpy(A, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, A, B, D);
    std::vector<float> d(256); hipMemcpy(d.data(), D, 1024, hipMemcpyDeviceToHost);
    int ok = 1; for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) { float s = 0; for (int k = 0; k < 32; k++) s += (float)ha[m * 32 + k] * (float)hb[k * 16 + n]; ok &= (s == d[m * 16 + n]); }
    printf("3. v_mfma_f32_16x16x32_f16 layout as assumed: %s\n", ok ? "yes" : "NO");
    bad += !ok;
  }
  { // 4: issue rate of the two shapes, one wave per SIMD (256 CUs x 4 waves)
    float* out; hipMalloc(&out, 2048 * 8);
    hipLaunchKernelGGL(k_rate<1>, dim3(1024), dim3(64), 0, 0, out, 2000); hipDeviceSynchronize();
    float a[2]; hipMemcpy(a, out, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_rate<0>, dim3(1024), dim3(64), 0, 0, out, 2000); hipDeviceSynchronize();
    float b[2]; hipMemcpy(b, out, 8, hipMemcpyDeviceToHost);
    printf("4. cycles per MFMA, one wave per SIMD, 8 independent accumulators: 16x16x32_f16 %.1f, 16x16x16_f16 %.1f\n", a[0], b[0]);
  }
  return bad;
}
