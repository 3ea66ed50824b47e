// Register-load fill-rate microbenchmark (gfx950): what a CU takes in per cycle from its XCD's L2 / its own L1, by ACCESS SHAPE.
// Every CU runs `wpc` waves (one workgroup); each wave keeps `depth` buffer_load_dwordx4 (1 KB per wave-instruction) in flight
// without ever draining (s_waitcnt vmcnt(depth - 1) per step), walking pseudo-random item bases over a working set of `ws` bytes.
// Shapes (what one 1 KB wave-instruction touches):
//   0 contiguous : lane l -> base + 16 l                                   (8 lines of 128 B, every byte used)
//   1 stride64   : lane l -> base + 64 l + 16 q, q = instruction % 4       (the fp32 per-edge kernel: 32+ lines, a quarter of each)
//   2 planes4x16 : lane l -> base + plane (l / 16) + rowwalk(l % 16) * 16  (dense-MFMA A operand from 8-half channel planes, 10-wide box)
//   3 stride256  : lane l -> base + 256 l + 16 q, q = instruction % 16     (channels-last fp16, one pixel per lane: 64 lines)
//   4 box64      : lane l -> base + row(l / 10) * rowstride + (l % 10) * 64 + 16 q   (the shipped kernel exactly: 10-wide box rows of 64-B cells)
//   6 dma_fill's : tools/ubench/dma_fill.hip's requests: 4 KB-aligned bases, runs of 20 x 16 B 2560 B apart, 4 planes 307200 B apart
//   7 cell64     : lane l -> base + 64 (l % 16) + 16 (l / 16): 1 KB contiguous in the dense-MFMA A-operand lane order (a quad = 4 cells)
//   8 cell64 box : the same with the 16 positions walking a 10-wide box of 64-byte cells (rows `rowstride` apart)
//   9 row16 far / 10 row16 near / 11 box near / 12 box far+48: variants of shape 2 — the tile's 16 positions in ONE row or walking a 10-wide box, the four
//     16-byte planes 307 200 B apart (the channel-blocked pyramid) or one row pitch apart, rows starting 48 bytes into a line
//   5 L1 window  : shape 0 inside a per-CU window of `l1win` bytes (16 KB: served by the CU's own L1 after the first touch)
//   hipcc --offload-arch=gfx950 -O3 -o l2_fill l2_fill.hip && ./l2_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned v4i __attribute__((ext_vector_type(4)));
template <int DEPTH, int SHAPE>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ src, unsigned ws, unsigned wsmask, int iters, unsigned rowstride, unsigned plane,
                                         unsigned l1win, unsigned long long* out, int* sink) {
  constexpr int QN = SHAPE == 1 || SHAPE == 4 || SHAPE == 6 ? 4 : SHAPE == 3 ? 16 : 1;
  constexpr bool WALK = SHAPE == 1 || SHAPE == 3 || SHAPE == 4;     // the QN instructions of an item walk over its 16-byte pieces      // instructions that walk over one item
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, ws, 0x00020000);
  const unsigned seed = __builtin_amdgcn_readfirstlane(blockIdx.x * 2654435761u + wave * 40503u + 12345u);
  unsigned lane_off;
  if (SHAPE == 0 || SHAPE == 5) lane_off = lane * 16;
  else if (SHAPE == 1) lane_off = lane * 64;
  else if (SHAPE == 2) { const int i = lane & 15; lane_off = (lane >> 4) * plane + (i / 10) * rowstride + (i % 10) * 16; }
  else if (SHAPE == 3) lane_off = lane * 256;
  else if (SHAPE == 7) lane_off = (lane & 15) * 64 + (lane >> 4) * 16;
  else if (SHAPE == 8) { const int i = lane & 15; lane_off = (i / 10) * rowstride + (i % 10) * 64 + (lane >> 4) * 16; }
  else if (SHAPE == 9)  { lane_off = (lane >> 4) * plane + (lane & 15) * 16; }                                   // 16 positions of ONE row, planes far apart
  else if (SHAPE == 10) { lane_off = (lane >> 4) * 2560u + (lane & 15) * 16; }                                  // 16 positions of one row, planes one row pitch apart
  else if (SHAPE == 11) { const int i = lane & 15; lane_off = (lane >> 4) * 2560u + (i / 10) * (16u * 2560u) + (i % 10) * 16; }   // 10-wide box, planes a row pitch apart
  else if (SHAPE == 12) { const int i = lane & 15; lane_off = (lane >> 4) * plane + (i / 10) * rowstride + (i % 10) * 16 + 48; }    // planes4x16, rows starting 48 B into a line
  else if (SHAPE == 6) { lane_off = (lane / 20) * 2560 + (lane % 20) * 16; }
  else lane_off = (lane / 10) * rowstride + (lane % 10) * 64;
  const unsigned cu_win = (blockIdx.x * l1win) & wsmask;
  v4i acc = {0, 0, 0, 0};
  v4i ring[DEPTH];
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int d = 0; d < DEPTH; d++) ring[d] = (v4i){0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      const unsigned n = (unsigned)(it * DEPTH + d);
      unsigned h = (seed + n / QN) * 2654435761u;
      h ^= h >> 15;
      const unsigned base = SHAPE == 5 ? cu_win + (h & (l1win - 1) & ~1023u) : SHAPE == 6 ? (h & wsmask & ~4095u) + (n % QN) * plane : (h & wsmask & ~127u);
      acc += ring[d];     // consume the oldest load (forces the wait on it only)
      ring[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + (WALK ? (n % QN) * 16 : 0u), base, 0);
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; d++) acc += ring[d];
  __syncthreads();                                  // the LAST wave's end: the arbiter favours old waves, wave 0 alone finishes early
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc.x == 0x7fffffff) sink[0] = acc.y + acc.z + acc.w;
}
template <int SHAPE>
void run(const char* name, const char* src, unsigned long long* out, int* sink) {
  const int iters = 600;
  for (unsigned ws : {2u << 20, 16u << 20, 256u << 20}) {
    if (SHAPE == 5 && ws != (2u << 20)) continue;
    for (int wpc : {4, 8, 16}) {
      double rate[2], tbs[2];
      for (int dsel = 0; dsel < 2; dsel++) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
          if (rep == 1) (void)hipEventRecord(e0, 0);
          if (dsel == 0) hipLaunchKernelGGL((k<4, SHAPE>), dim3(256), dim3(64 * wpc), 0, 0, src, 511u << 20, ws - 1, iters, 10240u, 307200u, 16384u, out, sink);
          else hipLaunchKernelGGL((k<16, SHAPE>), dim3(256), dim3(64 * wpc), 0, 0, src, 511u << 20, ws - 1, iters, 10240u, 307200u, 16384u, out, sink);
          if (rep == 1) (void)hipEventRecord(e1, 0);
          (void)hipDeviceSynchronize();
        }
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        tbs[dsel] = (double)iters * (dsel ? 16 : 4) * 1024 * wpc * 256 / (ms * 1e-3) / 1e12;
        std::vector<unsigned long long> h(256); (void)hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
        double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= 256;
        rate[dsel] = (double)iters * (dsel ? 16 : 4) * 1024 * wpc / cyc;
      }
      printf("%s ws %3u MB, %2d waves/CU: %5.1f B/cycle/CU (%5.2f TB/s by events) at 4 KB in flight per wave, %5.1f (%5.2f) at 16 KB\n", name, ws >> 20, wpc, rate[0], tbs[0], rate[1], tbs[1]);
    }
  }
}
int main() {
  const size_t total = 512ull << 20;
  char* src; (void)hipMalloc(&src, total); (void)hipMemset(src, 1, total);
  unsigned long long* out; (void)hipMalloc(&out, 256 * 8);
  int* sink; (void)hipMalloc(&sink, 4);
  run<0>("contiguous", src, out, sink);
  run<1>("stride64  ", src, out, sink);
  run<2>("planes4x16", src, out, sink);
  run<3>("stride256 ", src, out, sink);
  run<4>("box64     ", src, out, sink);
  run<5>("L1 window ", src, out, sink);
  run<6>("dma_fill's", src, out, sink);
  run<7>("cell64    ", src, out, sink);
  run<8>("cell64 box", src, out, sink);
  run<9>("row16 far ", src, out, sink);
  run<10>("row16 near", src, out, sink);
  run<11>("box near  ", src, out, sink);
  run<12>("box far+48", src, out, sink);
  return 0;
}
