// Micro-benchmark (debug tool, not part of the library): issue rate of v_mfma_f32_4x4x1_16b_f32 streams on gfx950 with
// the fillers the lookup kernel carries (out-of-range buffer loads, scalar address math), for 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_4x4.hip -o /tmp/mfma_4x4 && /tmp/mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define STEP3(U, B) \
  a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, (B), a0, 4, (U), 0); \
  a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x1, (B), a1, 4, (U), 0); \
  a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x2, (B), a2, 4, (U), 0)
#define STEP6(U, B) STEP3(U, B); \
  c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, (B), c0, 4, (U), 0); \
  c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x1, (B), c1, 4, (U), 0); \
  c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x2, (B), c2, 4, (U), 0)

template <int MODE>
__global__ __launch_bounds__(64) void k(const float* src, float* out, int iters, unsigned nbytes, unsigned voff, long long* cyc) {
  f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, c0 = a0, c1 = a0, c2 = a0;
  const int lane = threadIdx.x;
  float x0 = src[lane], x1 = src[lane + 64], x2 = src[lane + 128];
  f4 b0 = {x0, x1, x2, x0}, b1 = b0, b2 = b0, b3 = b0;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  unsigned off = voff + lane * 16;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 1 || MODE == 4) {       // 4 x b128 + 3 x b32 buffer loads (out of range when voff >= nbytes), used next iteration
      u4 l0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      u4 l1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024, 0, 0);
      u4 l2 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 2048, 0, 0);
      u4 l3 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 3072, 0, 0);
      unsigned m0 = __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4096, 0, 0);
      unsigned m1 = __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4352, 0, 0);
      unsigned m2 = __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4608, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      STEP3(0, b0[0]); STEP3(1, b0[1]); STEP3(2, b0[2]); STEP3(3, b0[3]); STEP3(4, b1[0]); STEP3(5, b1[1]); STEP3(6, b1[2]); STEP3(7, b1[3]);
      STEP3(8, b2[0]); STEP3(9, b2[1]); STEP3(10, b2[2]); STEP3(11, b2[3]); STEP3(12, b3[0]); STEP3(13, b3[1]); STEP3(14, b3[2]); STEP3(15, b3[3]);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_memcpy(&b0, &l0, 16); __builtin_memcpy(&b1, &l1, 16); __builtin_memcpy(&b2, &l2, 16); __builtin_memcpy(&b3, &l3, 16);
      x0 += __builtin_bit_cast(float, m0); x1 += __builtin_bit_cast(float, m1); x2 += __builtin_bit_cast(float, m2);
      if (MODE == 4) off += 5120;
    } else if (MODE == 5 || MODE == 6) { // only the 4 wide loads / only the 3 narrow ones
      u4 l0 = {0, 0, 0, 0}, l1 = l0, l2 = l0, l3 = l0; unsigned m0 = 0, m1 = 0, m2 = 0;
      if (MODE == 5) {
        l0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); l1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024, 0, 0);
        l2 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 2048, 0, 0); l3 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 3072, 0, 0);
      } else {
        m0 = __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4096, 0, 0); m1 = __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4352, 0, 0);
        m2 = __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4608, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      STEP3(0, b0[0]); STEP3(1, b0[1]); STEP3(2, b0[2]); STEP3(3, b0[3]); STEP3(4, b1[0]); STEP3(5, b1[1]); STEP3(6, b1[2]); STEP3(7, b1[3]);
      STEP3(8, b2[0]); STEP3(9, b2[1]); STEP3(10, b2[2]); STEP3(11, b2[3]); STEP3(12, b3[0]); STEP3(13, b3[1]); STEP3(14, b3[2]); STEP3(15, b3[3]);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 5) { __builtin_memcpy(&b0, &l0, 16); __builtin_memcpy(&b1, &l1, 16); __builtin_memcpy(&b2, &l2, 16); __builtin_memcpy(&b3, &l3, 16); }
      else { x0 += __builtin_bit_cast(float, m0); x1 += __builtin_bit_cast(float, m1); x2 += __builtin_bit_cast(float, m2); }
    } else if (MODE == 7 || MODE == 8 || MODE == 9) {   // 144 plain / DPP vector FMAs (9 chains) [+ the 4 wide loads]
      u4 l0 = {0, 0, 0, 0}, l1 = l0, l2 = l0, l3 = l0;
      if (MODE == 8) {
        l0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); l1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024, 0, 0);
        l2 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 2048, 0, 0); l3 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 3072, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const float bv = (u < 4 ? b0 : u < 8 ? b1 : u < 12 ? b2 : b3)[u & 3];
        if (MODE == 9) {
          asm volatile("s_nop 1\n"
                       "v_fmac_f32_dpp %0, %9, %10 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                       "v_fmac_f32_dpp %1, %9, %10 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                       "v_fmac_f32_dpp %2, %9, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                       "v_fmac_f32_dpp %3, %9, %10 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                       "v_fmac_f32_dpp %4, %9, %10 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                       "v_fmac_f32_dpp %5, %9, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                       "v_fmac_f32_dpp %6, %9, %10 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                       "v_fmac_f32_dpp %7, %9, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                       "v_fmac_f32_dpp %8, %9, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                       : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(a2[0])
                       : "v"(x0), "v"(bv));
        } else {
          a0[0] = __builtin_fmaf(x0, bv, a0[0]); a0[1] = __builtin_fmaf(x1, bv, a0[1]); a0[2] = __builtin_fmaf(x2, bv, a0[2]);
          a0[3] = __builtin_fmaf(x0, bv, a0[3]); a1[0] = __builtin_fmaf(x1, bv, a1[0]); a1[1] = __builtin_fmaf(x2, bv, a1[1]);
          a1[2] = __builtin_fmaf(x0, bv, a1[2]); a1[3] = __builtin_fmaf(x1, bv, a1[3]); a2[0] = __builtin_fmaf(x2, bv, a2[0]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 8) { __builtin_memcpy(&b0, &l0, 16); __builtin_memcpy(&b1, &l1, 16); __builtin_memcpy(&b2, &l2, 16); __builtin_memcpy(&b3, &l3, 16); }
    } else if (MODE == 2) {             // six accumulator chains
      STEP6(0, b0[0]); STEP6(1, b0[1]); STEP6(2, b0[2]); STEP6(3, b0[3]); STEP6(4, b1[0]); STEP6(5, b1[1]); STEP6(6, b1[2]); STEP6(7, b1[3]);
    } else if (MODE == 3) {             // one chain
      for (int u = 0; u < 48; u++) a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, b0[0], a0, 4, 0, 0);
    } else {                            // three chains, nothing else
      STEP3(0, b0[0]); STEP3(1, b0[1]); STEP3(2, b0[2]); STEP3(3, b0[3]); STEP3(4, b1[0]); STEP3(5, b1[1]); STEP3(6, b1[2]); STEP3(7, b1[3]);
      STEP3(8, b2[0]); STEP3(9, b2[1]); STEP3(10, b2[2]); STEP3(11, b2[3]); STEP3(12, b3[0]); STEP3(13, b3[1]); STEP3(14, b3[2]); STEP3(15, b3[3]);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  f4 s = a0 + a1 + a2 + c0 + c1 + c2;
  out[blockIdx.x * 64 + lane] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, const float* src, float* out, long long* cyc, unsigned nbytes, unsigned voff) {
  const int iters = 400;
  for (int w : {1, 2, 4, 5, 8}) {
    const int grid = 256 * 4 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, src, out, iters, nbytes, voff, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, src, out, iters, nbytes, voff, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= grid;
    const double nm = (double)iters * 48;
    printf("%-34s waves/SIMD %d: %7.1f us, wave cycles per MFMA %6.2f, SIMD cycles per MFMA (wall, 2.4 GHz) %6.2f\n", name, w, ms * 1e3,
           mean / nm, ms * 1e-3 * 2.4e9 / (nm * w));
  }
}

int main() {
  float *src, *out; long long* cyc;
  const unsigned nbytes = 64u << 20;
  hipMalloc(&src, nbytes); hipMemset(src, 0, nbytes); hipMalloc(&out, 256 * 4 * 8 * 64 * 4); hipMalloc(&cyc, 256 * 4 * 8 * 8);
  run<0>("3 chains", src, out, cyc, nbytes, 0);
  run<3>("1 chain", src, out, cyc, nbytes, 0);
  run<2>("6 chains", src, out, cyc, nbytes, 0);
  run<1>("3 chains + 7 out-of-range loads", src, out, cyc, nbytes, 0x80000000u);
  run<1>("3 chains + 7 loads, same lines", src, out, cyc, nbytes, 0);
  run<4>("3 chains + 7 loads, streaming", src, out, cyc, nbytes, 0);
  run<5>("3 chains + 4 x b128 loads", src, out, cyc, nbytes, 0);
  run<6>("3 chains + 3 x b32 loads", src, out, cyc, nbytes, 0);
  printf("(below: 144 vector FMAs per iteration; the per-MFMA columns are per 3 FMAs)\n");
  run<7>("144 v_fma", src, out, cyc, nbytes, 0);
  run<8>("144 v_fma + 4 x b128 loads", src, out, cyc, nbytes, 0);
  run<9>("144 v_fmac_dpp", src, out, cyc, nbytes, 0);
  return 0;
}
