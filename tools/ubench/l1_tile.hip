// Feasibility of "level 1 of the lookup from frame tiles in LDS", fp16 storage and fp32 storage (split into fp16 hi + lo planes)
// — "level 1 of the lookup from frame tiles in LDS" (DESIGN.md §7): how long does a workgroup take that stages 1/16 of a
// level-1 frame (17 x 19 positions x 128 fp16 channels = 83 KB, with its 9-position margin) ONCE and then serves the ~90 edges whose
// box origin lies in the tile's core from LDS?  Per edge: 6 M-tiles of 16 box positions x 4 K steps on v_mfma_f32_16x16x32_f16 (A =
// positions out of LDS, B = the patch from a [edge][16 px][128 ch] array), the 96 x 16 raw sums through a per-wave LDS scratch, a
// blend-shaped epilogue (441 outputs, 4 taps each).  Synthetic addresses and values: this measures time, not results.
//   hipcc --offload-arch=gfx950 -O3 -o l1_tile l1_tile.hip && ./l1_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int RW = 19, RH = 17, NPOS = RW * RH, PITCH = 272;         // region, bytes per staged position (256 + 16: bank spread)
constexpr int WAVES = 8, PST = 100, SCR = 16 * PST;                     // per-wave scratch: [16 pixels][96 positions (+4: bank spread)] floats

template <bool STAGE, bool MFMA, bool EPI, bool PREFETCH>
__global__ __launch_bounds__(WAVES * 64) void k_tile(const uint4* __restrict__ fmap, const uint4* __restrict__ patches, _Float16* __restrict__ out,
                                                     int edges_per_wg, int frame_positions) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* region = lds;
  float* scratch = reinterpret_cast<float*>(lds + NPOS * PITCH) + (threadIdx.x >> 6) * SCR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kg = lane >> 4;
  // ---- stage the tile: NPOS positions x 16 chunks of 16 bytes, rows of the frame are 40 positions wide
  if (STAGE) {
    const int frame = blockIdx.x / 16, tile = blockIdx.x % 16, ty = (tile / 4) * 8, tx = (tile % 4) * 10;
    for (int c = tid; c < NPOS * 16; c += WAVES * 64) {
      const int p = c >> 4, q = c & 15, y = ty + p / RW, x = tx + p % RW;
      const int src = (frame * frame_positions + (y % 30) * 40 + (x % 40)) * 16 + q;
      *reinterpret_cast<uint4*>(region + p * PITCH + q * 16) = fmap[src];
    }
  }
  __syncthreads();
  float sink = 0.0f;
  // (edge-independent epilogue indices: output o = lane + 64 j -> scratch offset of its tap (0, 0))
  int s0j[7];
#pragma unroll
  for (int j = 0; j < 7; j++) {
    const int o = min(lane + 64 * j, 440), p = o / 49, tap = o - 49 * p, a = tap / 7, c = tap - 7 * a;
    s0j[j] = p * PST + a * 9 + c;
  }
  h8 bnext[4];
  {
    const int eg0 = blockIdx.x * edges_per_wg + wave;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) bnext[ks] = __builtin_bit_cast(h8, patches[((size_t)eg0 * 16 + m) * 16 + 4 * ks + kg]);
  }
  for (int e = wave; e < edges_per_wg; e += WAVES) {
    const int eg = blockIdx.x * edges_per_wg + e;
    const int ox = (eg * 7) % 10, oy = (eg * 3) % 8;                   // box origin inside the core
    h8 b[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) b[ks] = bnext[ks];
    if (PREFETCH) {                                                    // the next edge's patch while this one is multiplied
      const int en = min(e + WAVES, edges_per_wg - 1), egn = blockIdx.x * edges_per_wg + en;
#pragma unroll
      for (int ks = 0; ks < 4; ks++) bnext[ks] = __builtin_bit_cast(h8, patches[((size_t)egn * 16 + m) * 16 + 4 * ks + kg]);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ks++) b[ks] = __builtin_bit_cast(h8, patches[((size_t)eg * 16 + m) * 16 + 4 * ks + kg]);
    }
    f4 acc[6];
#pragma unroll
    for (int t = 0; t < 6; t++) {
      const int i = min(16 * t + m, 80), iy = i / 9, ix = i - 9 * iy;
      const unsigned char* ap = region + ((oy + iy) * RW + ox + ix) * PITCH + kg * 16;
      acc[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const h8 a = *reinterpret_cast<const h8*>(ap + ks * 64);
        if (MFMA) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[ks], acc[t], 0, 0, 0);
        else acc[t][0] += (float)a[0] * (float)b[ks][0];
      }
    }
    // raw sums -> scratch [pixel m][position 16 t + 4 kg + i]: one 16-byte store per tile
#pragma unroll
    for (int t = 0; t < 6; t++) *reinterpret_cast<f4*>(scratch + m * PST + 16 * t + 4 * kg) = acc[t];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (EPI) {
      const float fx = 0.25f + 0.001f * (float)(eg & 63), fy = 0.5f;
#pragma unroll
      for (int j = 0; j < 7; j++) {
        const int o = lane + 64 * j;
        if (o < 441) {
          const int s0 = s0j[j];                                       // tap (a, c) of pixel p inside the 9 x 9 box
          const float v00 = scratch[s0], v01 = scratch[s0 + 1], v10 = scratch[s0 + 9], v11 = scratch[s0 + 10];
          const float v = (1 - fx) * (1 - fy) * v00 + fx * (1 - fy) * v01 + (1 - fx) * fy * v10 + fx * fy * v11;
          out[(size_t)eg * 441 + o] = (_Float16)v;
        }
      }
    } else sink += scratch[lane];
    __builtin_amdgcn_wave_barrier();
  }
  if (sink == 12345.0f) out[0] = (_Float16)sink;
}

// ---- fp32 storage: the tile is split into fp16 hi + lo planes WHILE it is staged (x = hi + lo, 22 significant bits); products on the
// same MFMA shape as a_hi b_hi + a_hi b_lo + a_lo b_hi (fp32 accumulation).  Tile = 13 x 17 positions (core of 4 x 8 origins), 4 waves.
constexpr int RW2 = 17, RH2 = 13, NPOS2 = RW2 * RH2, PITCH2 = 528, WAVES2 = 4;
template <bool EPI>
__global__ __launch_bounds__(WAVES2 * 64) void k_tile_f32(const float4* __restrict__ fmap, const float4* __restrict__ patches, float* __restrict__ out,
                                                          int edges_per_wg, int frame_positions) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* region = lds;
  float* scratch = reinterpret_cast<float*>(lds + NPOS2 * PITCH2) + (threadIdx.x >> 6) * SCR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kg = lane >> 4;
  auto split = [](float4 a, float4 b, h8& hi, h8& lo) {               // 8 floats -> 8 (hi, lo) pairs
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; i++) { const _Float16 h = (_Float16)v[i]; hi[i] = h; lo[i] = (_Float16)(v[i] - (float)h); }
  };
  {
    const int tile = blockIdx.x % 40, frame = blockIdx.x / 40, ty = (tile / 5) * 4, tx = (tile % 5) * 8;
    for (int c = tid; c < NPOS2 * 16; c += WAVES2 * 64) {             // chunk = 8 channels of one position: 32 bytes of fp32
      const int p = c >> 4, q = c & 15, y = ty + p / RW2, x = tx + p % RW2;
      const size_t src = ((size_t)(frame * frame_positions + (y % 30) * 40 + (x % 40)) * 16 + q) * 2;
      h8 hi, lo;
      split(fmap[src], fmap[src + 1], hi, lo);
      *reinterpret_cast<h8*>(region + p * PITCH2 + q * 16) = hi;
      *reinterpret_cast<h8*>(region + p * PITCH2 + 256 + q * 16) = lo;
    }
  }
  __syncthreads();
  int s0j[7];
#pragma unroll
  for (int j = 0; j < 7; j++) {
    const int o = min(lane + 64 * j, 440), p = o / 49, tap = o - 49 * p, a = tap / 7, c = tap - 7 * a;
    s0j[j] = p * PST + a * 9 + c;
  }
  float sink = 0.0f;
  for (int e = wave; e < edges_per_wg; e += WAVES2) {
    const int eg = blockIdx.x * edges_per_wg + e;
    const int ox = (eg * 7) % 8, oy = (eg * 3) % 4;
    h8 bh[4], bl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const size_t src = (((size_t)eg * 16 + m) * 16 + 4 * ks + kg) * 2;
      split(patches[src], patches[src + 1], bh[ks], bl[ks]);
    }
    f4 acc[6];
#pragma unroll
    for (int t = 0; t < 6; t++) {
      const int i = min(16 * t + m, 80), iy = i / 9, ix = i - 9 * iy;
      const unsigned char* ap = region + ((oy + iy) * RW2 + ox + ix) * PITCH2 + kg * 16;
      acc[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const h8 ah = *reinterpret_cast<const h8*>(ap + ks * 64), al = *reinterpret_cast<const h8*>(ap + 256 + ks * 64);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ks], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ks], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ks], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 6; t++) *reinterpret_cast<f4*>(scratch + m * PST + 16 * t + 4 * kg) = acc[t];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (EPI) {
      const float fx = 0.25f + 0.001f * (float)(eg & 63), fy = 0.5f;
#pragma unroll
      for (int j = 0; j < 7; j++) {
        const int o = lane + 64 * j;
        if (o < 441) {
          const int s0 = s0j[j];
          const float v00 = scratch[s0], v01 = scratch[s0 + 1], v10 = scratch[s0 + 9], v11 = scratch[s0 + 10];
          out[(size_t)eg * 441 + o] = (1 - fx) * (1 - fy) * v00 + fx * (1 - fy) * v01 + (1 - fx) * fy * v10 + fx * fy * v11;
        }
      }
    } else sink += scratch[lane];
    __builtin_amdgcn_wave_barrier();
  }
  if (sink == 12345.0f) out[0] = sink;
}

template <bool E> float run_f32(const float4* fmap, const float4* patches, float* out, int wgs, int epw) {
  const size_t lds = NPOS2 * PITCH2 + WAVES2 * SCR * 4;
  hipFuncSetAttribute((const void*)k_tile_f32<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_tile_f32<E>), dim3(wgs), dim3(WAVES2 * 64), lds, 0, fmap, patches, out, epw, 1200);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k_tile_f32<E>), dim3(wgs), dim3(WAVES2 * 64), lds, 0, fmap, patches, out, epw, 1200);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 20 * 1e3f;
}

template <bool S, bool M, bool E, bool P> float run(const uint4* fmap, const uint4* patches, _Float16* out, int wgs, int epw) {
  const size_t lds = NPOS * PITCH + WAVES * SCR * 4;
  hipFuncSetAttribute((const void*)k_tile<S, M, E, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_tile<S, M, E, P>), dim3(wgs), dim3(WAVES * 64), lds, 0, fmap, patches, out, epw, 1200);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k_tile<S, M, E, P>), dim3(wgs), dim3(WAVES * 64), lds, 0, fmap, patches, out, epw, 1200);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 20 * 1e3f;
}

int main() {
  const int frames = 15, wgs = frames * 16, epw = 90, E = wgs * epw;
  uint4 *fmap, *patches; _Float16* out;
  hipMalloc(&fmap, (size_t)frames * 1200 * 256); hipMalloc(&patches, (size_t)E * 16 * 256); hipMalloc(&out, (size_t)E * 441 * 2);
  hipMemset(fmap, 0, (size_t)frames * 1200 * 256); hipMemset(patches, 0, (size_t)E * 16 * 256);
  printf("level-1 tile prototype: %d workgroups x %d edges (E = %d), LDS %zu B per workgroup\n", wgs, epw, E, (size_t)NPOS * PITCH + WAVES * SCR * 4);
  printf("  everything               : %7.1f us per launch\n", run<true, true, true, true>(fmap, patches, out, wgs, epw));
  printf("  patches not prefetched   : %7.1f us\n", run<true, true, true, false>(fmap, patches, out, wgs, epw));
  printf("  without the epilogue     : %7.1f us\n", run<true, true, false, true>(fmap, patches, out, wgs, epw));
  printf("  without the MFMAs        : %7.1f us\n", run<true, false, true, true>(fmap, patches, out, wgs, epw));
  printf("  without staging the tile : %7.1f us\n", run<false, true, true, true>(fmap, patches, out, wgs, epw));
  printf("  (per-edge kernel, level 1 alone, fp16: 42.4 us per launch at E = 21 600 — bench.py --per-level-launches)\n");
  {
    const int wgs2 = frames * 40, epw2 = 36;                           // 21 600 edges again
    float4 *f32map, *p32; float* o32;
    hipMalloc(&f32map, (size_t)frames * 1200 * 512); hipMalloc(&p32, (size_t)E * 16 * 512); hipMalloc(&o32, (size_t)E * 441 * 4);
    hipMemset(f32map, 0, (size_t)frames * 1200 * 512); hipMemset(p32, 0, (size_t)E * 16 * 512);
    printf("fp32 storage, hi + lo planes split at staging time: %d workgroups x %d edges, LDS %zu B per workgroup\n", wgs2, epw2,
           (size_t)NPOS2 * PITCH2 + WAVES2 * SCR * 4);
    printf("  everything               : %7.1f us per launch\n", run_f32<true>(f32map, p32, o32, wgs2, epw2));
    printf("  without the epilogue     : %7.1f us\n", run_f32<false>(f32map, p32, o32, wgs2, epw2));
    printf("  (per-edge kernel, level 1 alone, fp32: 90.4 us per launch)\n");
  }
  return 0;
}
