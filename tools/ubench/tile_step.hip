// One trailing-update step of k_ba_solve_chain (ba.hip) as the tile waves run it: the factor of the current block out of LDS, the operand
// rows solved against it, the 16 x 16 tile on the matrix cores, the panel written transposed, one barrier.  Cycles per step for W tile
// waves (the kernel has 12; 15 / 10 / 6 / 3 / 1 tiles live as the factorisation proceeds) and with parts left out.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tile_step tile_step.hip && ./tile_step
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// PARTS bit 0: factor load, 1: substitution, 2: tile read / write, 3: matrix cores, 4: partner exchange, 5: transposed panel store,
// bit 7: as bit 6, with the broadcast folded into the substitution's instructions (v_fmac_f32_dpp / v_mul_f32_dpp, inline asm)
// bit 6: the factor as TWO ds_read_b32 (lane l holds compact entry l % 16) + DPP row_newbcast operands instead of 7 broadcast reads
template <int PARTS>
__global__ void k_tiles(unsigned long long* out, float* sink, int iters, int active) {
  constexpr int LD = 100, N = 14, rows = 85;
  __shared__ float A[rows * LD];
  __shared__ float Ld[N * 36];
  __shared__ float s_dump[64];
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, mm = ln & 15, kq = ln >> 4;
  for (int i = threadIdx.x; i < rows * LD; i += blockDim.x) A[i] = ((i / LD) == (i % LD)) ? 4.0f : 0.001f * ((i * 7) % 13);
  for (int i = threadIdx.x; i < N * 36; i += blockDim.x) Ld[i] = ((i % 36) % 7 == 0) ? 0.5f : 0.002f * (i % 5);
  __syncthreads();
  int I = 0;
  while ((I + 1) * (I + 2) / 2 <= wv) I++;
  const int J = wv - I * (I + 1) / 2;
  const int rrel0 = 16 * I + 4 * kq, crel = 16 * J + mm;
  int thr[4];
  for (int i = 0; i < 4; i++) { const int r = rrel0 + i; thr[i] = (crel <= r && r >= 6) ? min((rows - r + 5) / 6 - 1, (rows - 1 - crel + 5) / 6 - 1) : -1; }
  const int rel_a = 16 * I + mm, rel_mine = kq < 2 ? rel_a : crel, keep = J == 0, thr_t = J == 0 ? (rows - rel_a + 5) / 6 - 1 : -1;
  const int idx0 = (6 + rrel0) * LD + 6 + crel;
  auto from_partner = [&](float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(ln < 32 ? r[1] : r[0]);
  };
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    const int jb = it % 3;                                      // (the first steps: every tile exists)
    if (wv < active) {
      const int j0 = 6 * jb, j1 = j0 + 6;
      float Lb[6][6], inv[6];
      if (PARTS & 64) {
        const float r0 = Ld[jb * 36 + (ln & 15)], r1 = Ld[jb * 36 + 16 + (ln & 7)];
#define BC(v, k) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + (k), 0xf, 0xf, true))
        Lb[1][0] = BC(r0, 0);
        Lb[2][0] = BC(r0, 1);
        Lb[2][1] = BC(r0, 2);
        Lb[3][0] = BC(r0, 3);
        Lb[3][1] = BC(r0, 4);
        Lb[3][2] = BC(r0, 5);
        Lb[4][0] = BC(r0, 6);
        Lb[4][1] = BC(r0, 7);
        Lb[4][2] = BC(r0, 8);
        Lb[4][3] = BC(r0, 9);
        Lb[5][0] = BC(r0, 10);
        Lb[5][1] = BC(r0, 11);
        Lb[5][2] = BC(r0, 12);
        Lb[5][3] = BC(r0, 13);
        Lb[5][4] = BC(r0, 14);
        inv[0] = BC(r0, 15);
        inv[1] = BC(r1, 0);
        inv[2] = BC(r1, 1);
        inv[3] = BC(r1, 2);
        inv[4] = BC(r1, 3);
        inv[5] = BC(r1, 4);
#undef BC
      } else
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = 0; c < a; c++) Lb[a][c] = (PARTS & 1) ? Ld[jb * 36 + a * 6 + c] : 0.01f * (a + c);
        inv[a] = (PARTS & 1) ? Ld[jb * 36 + a * 7] : 0.5f;
      }
      const float* pr = A + __mul24(min(j1 + rel_mine, rows - 1), LD) + j0;
      float v[6], x[6];
#pragma unroll
      for (int q = 0; q < 6; q++) v[q] = pr[q];
      f4 c;
      float* dst[4];
      float* p0 = A + idx0 + jb * (6 * LD + 6);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        dst[i] = (jb < thr[i]) ? p0 + i * LD : s_dump + ln;
        c[i] = (PARTS & 4) ? *dst[i] : 0.0f;
      }
      if (PARTS & 128) {
        const float r0 = Ld[jb * 36 + (ln & 15)], r1 = Ld[jb * 36 + 16 + (ln & 7)];
        { float t = v[0];
          asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(x[0]) : "v"(r0), "v"(t)); }
        { float t = v[1];
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[0]));
          asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(x[1]) : "v"(r1), "v"(t)); }
        { float t = v[2];
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[0]));
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[1]));
          asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "=v"(x[2]) : "v"(r1), "v"(t)); }
        { float t = v[3];
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[0]));
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[1]));
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[2]));
          asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "=v"(x[3]) : "v"(r1), "v"(t)); }
        { float t = v[4];
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[0]));
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[1]));
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[2]));
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[3]));
          asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x[4]) : "v"(r1), "v"(t)); }
        { float t = v[5];
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:10 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[0]));
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[1]));
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[2]));
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:13 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[3]));
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:14 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(r0), "v"(x[4]));
          asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "=v"(x[5]) : "v"(r1), "v"(t)); }
      } else
      if (PARTS & 2) {
#pragma unroll
        for (int cc = 0; cc < 6; cc++) {
          float t = v[cc];
#pragma unroll
          for (int k = 0; k < cc; k++) t -= x[k] * Lb[cc][k];
          x[cc] = t * inv[cc];
        }
      } else {
#pragma unroll
        for (int cc = 0; cc < 6; cc++) x[cc] = v[cc] * inv[cc];
      }
      const bool odd = kq & 1, hi = kq & 2;
      const float u = odd ? x[1] : x[0], w = odd ? x[3] : x[2];
      const float mine = hi ? w : u, theirs = hi ? u : w;
      const float s2 = odd ? x[5] : x[4];
      const float got1 = (PARTS & 16) ? from_partner(theirs) : theirs, got2 = (PARTS & 16) ? from_partner(s2) : s2;
      const float a1 = hi ? got1 : mine, b1 = hi ? mine : got1;
      const float a2 = hi ? 0.0f : s2, b2 = hi ? 0.0f : got2;
      if (PARTS & 8) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, -b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, -b2, c, 0, 0, 0);
      } else { c[0] += a1 * b1; c[1] += a2 * b2; }
      if (PARTS & 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) *dst[i] = c[i];
      } else if (c[0] + c[1] + c[2] + c[3] == 12345.0f) *dst[0] = c[0];
      if (keep && (PARTS & 32)) {
        const bool ok = jb < thr_t;
        float* t1 = ok ? A + __mul24(j0 + kq, LD) + j1 + rel_a : s_dump + ln;
        float* t2 = (ok && !hi) ? A + __mul24(j0 + 4 + kq, LD) + j1 + rel_a : s_dump + ln;
        *t1 = a1;
        *t2 = a2;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = A[threadIdx.x] + Ld[ln];
}

template <int PARTS>
static void run(const char* what, unsigned long long* d, float* sink) {
  unsigned long long h[8];
  const int it = 1500;
  printf("%-44s", what);
  for (int active : {1, 3, 6, 12}) {
    hipLaunchKernelGGL(k_tiles<PARTS>, dim3(1), dim3(768), 0, 0, d, sink, it, active);
    (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("  %2d tiles: %5.0f", active, (double)h[0] / it);
  }
  printf("   cycles per step\n");
}

int main() {
  unsigned long long* d; float* sink;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 4096 * 4);
  run<63>("everything", d, sink);
  run<62>("without the factor load", d, sink);
  run<61>("without the substitution", d, sink);
  run<59>("without the tile read / write", d, sink);
  run<55>("without the matrix cores", d, sink);
  run<47>("without the partner exchange", d, sink);
  run<31>("without the transposed panel", d, sink);
  run<0>("nothing but the operand rows", d, sink);
  run<63 + 64>("everything, factor through DPP row_newbcast", d, sink);
  run<62 + 128>("factor folded into v_fmac_f32_dpp / v_mul_f32_dpp", d, sink);
  return 0;
}
