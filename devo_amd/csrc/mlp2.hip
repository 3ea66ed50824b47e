// Linear - ReLU - Linear as ONE kernel (fp16 storage, fp32 accumulation): the two-layer chains of the Update operator — the corr MLP's head
// (enet.py:59-61: Linear(882, 384), ReLU, Linear(384, 384)) and the neighbour mixers c1 / c2 (enet.py:46-50: Linear, ReLU, Linear on
// `mask * net[:, ix]`, added to net) — which ran as two launches of k_linear_f16 (14 - 19 us each at 21 600 rows, each a pass of its
// 16.6 MB input and output through HBM, plus a gather kernel in front of c1 / c2).
//
//   y[r] = (residual[r] +)  W2 relu(W1 x[src(r)] + b1) + b2,      src(r) = r  or  gather[r]  (< 0: a zero row — the neighbour mask)
//
// One workgroup = 128 rows x all 384 columns, a wave = 16 rows x 24 column tiles (96 accumulator registers per lane).  The 384-wide
// intermediate never leaves the wave: layer 1's result goes, ReLU'd and rounded to fp16 like the two-launch form's, through one KB of LDS per
// wave from the result layout into the A-operand layout and stays in 48 registers; layer 2 multiplies it from there.  Both weights arrive pre-packed as B-operand images (devo_upd_mlp2_pack_weight: one KB per (K step of 32,
// column tile)) through ONE ring of three 24 KB stages that runs on from layer 1 into layer 2 (LDS-DMA two steps ahead, one barrier per
// step; a piece is read by all eight waves); layer 1's rows arrive the same way, 64 contiguous bytes of a row per quad of lanes, in a
// ring of three 1 KB slabs per wave (every vector-memory request of the loops is an LDS-DMA: the waits are counted by hand).  The result
// leaves through the idle weight ring as whole 16-byte pieces, with the bias and the optional residual.
// 152 KB of LDS: one workgroup (two waves per SIMD) per CU; 21 600 rows = 169 workgroups.
#include "common.h"
#include <hip/hip_fp16.h>

namespace devo {

typedef _Float16 mp_h8 __attribute__((ext_vector_type(8)));
typedef float mp_f4 __attribute__((ext_vector_type(4)));
typedef unsigned mp_u4 __attribute__((ext_vector_type(4)));

constexpr int MP_N = 384;                          // width of both layers' outputs
constexpr int MP_T = MP_N / 16;                    // column tiles
constexpr int MP_ROWS = 128;                       // rows per workgroup: 8 waves x 16
constexpr int MP_NK = MP_N / 32;                   // K steps of layer 2
constexpr int MP_STAGE = MP_T * 1024;              // one K step (32) of a weight image
constexpr int MP_NST = 3;                          // ring of row slabs per wave (requested two steps ahead)
constexpr int MP_NSTW = 5;                         // ring of weight stages (requested four steps ahead: two whole stages in flight behind the one awaited)
constexpr int MP_SLAB = 1024;                      // a wave's 16 input rows x 64 bytes of one K step
constexpr int MP_LDS = MP_NSTW * MP_STAGE + 8 * MP_NST * MP_SLAB + 8 * 1024 + 2 * MP_N * 4;   // weight ring | row slabs | one KB per wave for the layout change | both biases: 158 720 B

// W fp16, element (n, k) at W[n * s_n + k * s_k], n < 384 -> [ceil(K / 32)][24 tiles][64 lanes][16 B]: lane (n, kg) of tile t and step s holds
// k = 32 s + 8 kg .. + 7 of column 16 t + n (zeros past K)
__global__ __launch_bounds__(256) void k_mlp2_pack(const __half* __restrict__ W, int64_t s_n, int64_t s_k, int K, mp_u4* __restrict__ out) {
  const int nk = (K + 31) / 32;
  const int total = nk * MP_T * 64;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, t = (i >> 6) % MP_T, s = (i >> 6) / MP_T;
    const int n = 16 * t + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    mp_h8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = k0 + e < K ? (_Float16)__half2float(W[(int64_t)n * s_n + (int64_t)(k0 + e) * s_k]) : (_Float16)0.f;
    out[i] = __builtin_bit_cast(mp_u4, v);
  }
}

__device__ __forceinline__ void mp_dma16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned lds_addr) {
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(lds_addr), "s"(soff) : "memory");
}

__device__ __forceinline__ void mp_wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool GATHER>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_mlp2_f16(
    const __half* __restrict__ x, int64_t ldx, int x_rows, const int64_t* __restrict__ gather, const mp_u4* __restrict__ w1, const __half* __restrict__ b1,
    int K1, const mp_u4* __restrict__ w2, const __half* __restrict__ b2, const __half* residual, __half* y, int64_t ldy, int M) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char mp_lds[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, mi = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.x * MP_ROWS + 16 * wv;                    // this wave's 16 rows (all 384 columns of them)
  const int nk1 = (K1 + 31) / 32, nst = nk1 + MP_NK;
  constexpr unsigned OFF_NONE = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(x), 0, (unsigned)(((int64_t)(x_rows - 1) * ldx + K1) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<mp_u4*>(w1), 0, (unsigned)(nk1 * MP_STAGE), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<mp_u4*>(w2), 0, (unsigned)(MP_NK * MP_STAGE), 0x00020000);
  const unsigned lds0 = (unsigned)(uintptr_t)mp_lds;
  // layer 1's rows: lane L requests piece L & 3 (16 bytes) of row L >> 2 — a quad of lanes = the 64 contiguous bytes of a row's K step —
  // into slot L of the wave's slab; lane (mi, kg) then reads slot 4 mi + kg
  unsigned char* const slab = mp_lds + MP_NSTW * MP_STAGE + wv * (MP_NST * MP_SLAB);
  unsigned char* const bounce = mp_lds + MP_NSTW * MP_STAGE + 8 * MP_NST * MP_SLAB + wv * 1024;     // 1 KB per wave: result layout -> A-operand layout
  float* const biasl = reinterpret_cast<float*>(mp_lds + MP_NSTW * MP_STAGE + 8 * MP_NST * MP_SLAB + 8 * 1024);      // b1 | b2 as floats
  for (int c = tid; c < 2 * MP_N; c += 512) {
    const __half* bp = c < MP_N ? b1 : b2;
    biasl[c] = bp ? __half2float(bp[c < MP_N ? c : c - MP_N]) : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // (the last compiler-counted loads in front of the loops)
  unsigned arow = OFF_NONE;
  {
    const int r = row0 + (lane >> 2);
    long long src = r < M ? (GATHER ? (long long)gather[r] : (long long)r) : -1;
    if (src >= 0 && src < x_rows) arow = (unsigned)((src * ldx) * 2 + 16 * (lane & 3));
  }
  auto stage = [&](int g) {                                           // this wave's 3 of stage g's 24 one-KB pieces (layer 1's image, then layer 2's)
    const int buf = g % MP_NSTW;
#pragma unroll
    for (int q = 0; q < MP_T / 8; q++) {
      const int piece = wv + 8 * q;
      const unsigned voff = g < nst ? (unsigned)lane * 16u : OFF_NONE;
      const bool first = g < nk1;
      mp_dma16(voff, first ? rs1 : rs2, (unsigned)(((first ? g : g - nk1) * MP_T + piece) * 1024), lds0 + (unsigned)(buf * MP_STAGE + piece * 1024));
    }
  };
  auto load_a = [&](int s) {                                          // K step s of this wave's rows -> slab s % 3 (behind the last step: zeros, no access)
    mp_dma16(s < nk1 ? arow : OFF_NONE, rsx, (unsigned)s * 64u, (unsigned)(uintptr_t)slab + (unsigned)((s % MP_NST) * MP_SLAB));
  };
  mp_f4 acc[MP_T];
#pragma unroll
  for (int t = 0; t < MP_T; t++) acc[t] = mp_f4{0.f, 0.f, 0.f, 0.f};
  // Request order: rows(0) W(0) rows(1) W(1) W(2) W(3), then per step g: rows(g + 2) W(g + 4).  Every vector-memory request of the loops is an
  // LDS-DMA, 1 per row slab and 3 per weight stage and wave, so the waits are counted by hand: behind step g everything up to rows(g + 1) must
  // be there — W(g + 1) and W(g + 2) are older — which leaves W(g + 3), rows(g + 2), W(g + 4) = 7 requests in flight.
  load_a(0);
  stage(0);
  load_a(1);
  stage(1);
  stage(2);
  stage(3);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                   // rows(0) and W(0) are there
  __syncthreads();
  // a step's 24 products: the weight fragments of six tiles are read while the previous six multiply (left to itself the compiler keeps two
  // fragments in flight and waits for the LDS before every product: 2.6 k cycles per step where the matrix pipe and the LDS need 0.8 k each)
  auto products = [&](int g, mp_h8 a) {
    const mp_u4* sb = reinterpret_cast<const mp_u4*>(mp_lds + (g % MP_NSTW) * MP_STAGE) + lane;
    constexpr int GB = 6;
    mp_u4 b[2][GB];
#pragma unroll
    for (int u = 0; u < GB; u++) b[0][u] = sb[u * 64];
#pragma unroll
    for (int k = 0; k < MP_T / GB; k++) {
      if (k + 1 < MP_T / GB) {
#pragma unroll
        for (int u = 0; u < GB; u++) b[(k + 1) & 1][u] = sb[((k + 1) * GB + u) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < GB; u++) acc[k * GB + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(mp_h8, b[k & 1][u]), acc[k * GB + u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- layer 1: K steps 0 .. nk1 - 1
  for (int g = 0; g < nk1; g++) {
    mp_h8 a = __builtin_bit_cast(mp_h8, *reinterpret_cast<const mp_u4*>(slab + (g % MP_NST) * MP_SLAB + (4 * mi + kg) * 16));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    load_a(g + 2);
    stage(g + 4);
    if (32 * g + 32 > K1) {                                           // the last step of a K that is not a multiple of 32: what lies behind the row is not part of it
#pragma unroll
      for (int e = 0; e < 8; e++) a[e] = 32 * g + 8 * kg + e < K1 ? a[e] : (_Float16)0.f;
    }
    products(g, a);
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __syncthreads();
  }
  // ---- the intermediate relu(acc + b1), fp16 (rounded like the two-launch form's), from the result layout (lane = column n, rows 4 kg + j) into
  //      the A-operand layout (lane = row mi, 8 consecutive k) through the wave's own KB of LDS, one K step (two column tiles) at a time; it stays
  //      in 48 registers
  mp_u4 a2[MP_NK];
#pragma unroll
  for (int s2 = 0; s2 < MP_NK; s2++) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int T = 2 * s2 + h;
      const float bv = biasl[16 * T + mi];
      unsigned char* cell = bounce + ((2 * h + (mi >> 3)) * 16) * 16 + (mi & 7) * 2;
#pragma unroll
      for (int j = 0; j < 4; j++) *reinterpret_cast<_Float16*>(cell + (4 * kg + j) * 16) = (_Float16)fmaxf(acc[T][j] + bv, 0.f);
    }
    mp_wave_lds_fence();                                              // (a wave's LDS operations execute in order; this pins the compiler's)
    a2[s2] = *reinterpret_cast<const mp_u4*>(bounce + lane * 16);
    mp_wave_lds_fence();
  }
#pragma unroll
  for (int t = 0; t < MP_T; t++) acc[t] = mp_f4{0.f, 0.f, 0.f, 0.f};
  // ---- layer 2: K steps nk1 .. nk1 + 11, the rows from the registers
#pragma unroll
  for (int s2 = 0; s2 < MP_NK; s2++) {
    const int g = nk1 + s2;
    load_a(g + 2);                                                    // (no rows left: the out-of-range offset — the request count per step stays 4)
    stage(g + 4);
    products(g, __builtin_bit_cast(mp_h8, a2[s2]));
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- result: acc + b2, half of the columns at a time through the wave's share of the (now idle) weight ring — [16 rows][192 columns] fp16,
  //      row pitch 400 bytes — and out as whole 16-byte pieces with the residual
  unsigned char* const outb = mp_lds + wv * (16 * 400);
#pragma unroll
  for (int hc = 0; hc < 2; hc++) {
#pragma unroll
    for (int t = 0; t < 12; t++) {
      const int T = 12 * hc + t;
      const float bv = biasl[MP_N + 16 * T + mi];
#pragma unroll
      for (int j = 0; j < 4; j++) *reinterpret_cast<_Float16*>(outb + (4 * kg + j) * 400 + (16 * t + mi) * 2) = (_Float16)(acc[T][j] + bv);
    }
    mp_wave_lds_fence();
#pragma unroll
    for (int it = 0; it < 6; it++) {
      const int c = it * 64 + lane, r = c / 24, c8 = c - r * 24;     // row of the wave, group of 8 columns of the half
      const int row = row0 + r;
      mp_h8 v = __builtin_bit_cast(mp_h8, *reinterpret_cast<const mp_u4*>(outb + r * 400 + c8 * 16));
      if (row < M) {
        const int64_t o = (int64_t)row * ldy + 192 * hc + 8 * c8;
        if (residual) {
          const mp_h8 rr = __builtin_bit_cast(mp_h8, *reinterpret_cast<const mp_u4*>(residual + o));
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = (_Float16)((float)v[e] + (float)rr[e]);
        }
        *reinterpret_cast<mp_u4*>(y + o) = __builtin_bit_cast(mp_u4, v);
      }
    }
    mp_wave_lds_fence();
  }
}

}  // namespace devo

using namespace devo;

extern "C" {

size_t devo_upd_mlp2_weight_bytes(int K) { return K > 0 ? (size_t)((K + 31) / 32) * MP_STAGE : 0; }

int devo_upd_mlp2_pack_weight(const void* W, int64_t s_n, int64_t s_k, int K, void* wimage, devo_stream_t stream) {
  DEVO_REQUIRE(K > 0 && W && wimage && (reinterpret_cast<uintptr_t>(wimage) & 15) == 0, "devo_upd_mlp2_pack_weight: bad arguments (K = %d)", K);
  const int total = (K + 31) / 32 * MP_T * 64;
  hipLaunchKernelGGL(k_mlp2_pack, dim3((unsigned)blocks_for(total, 256, 1024)), dim3(256), 0, (hipStream_t)stream, (const __half*)W, s_n, s_k, K, (mp_u4*)wimage);
  return check_launch("devo_upd_mlp2_pack_weight");
}

int devo_upd_mlp2_f16(const void* x, int64_t ldx, int x_rows, const int64_t* gather, const void* w1image, const void* b1, int K1, const void* w2image,
                      const void* b2, const void* residual, void* y, int64_t ldy, int M, devo_stream_t stream) {
  DEVO_REQUIRE(M >= 0 && K1 > 0 && x_rows > 0, "devo_upd_mlp2_f16: bad sizes (%d rows, K = %d)", M, K1);
  if (M == 0) return DEVO_OK;
  DEVO_REQUIRE(x && w1image && w2image && y && ldx >= K1 && ldx % 2 == 0 && ldy >= MP_N && ldy % 8 == 0,
               "devo_upd_mlp2_f16: null tensor, input rows shorter than K or an odd number of elements apart, or output rows that are not a multiple of 8 elements apart");
  DEVO_REQUIRE(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(w1image) | reinterpret_cast<uintptr_t>(w2image)) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(x) & 3) == 0, "devo_upd_mlp2_f16: y / residual / the weight images must be 16-byte aligned, x 4-byte");
  DEVO_REQUIRE(((int64_t)(x_rows - 1) * ldx + K1) * 2 < (1LL << 31), "devo_upd_mlp2_f16: input beyond 2 GB");
  static PerDeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute((const void*)k_mlp2_f16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MP_LDS);
    (void)hipFuncSetAttribute((const void*)k_mlp2_f16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MP_LDS);
    (void)hipGetLastError();
  }
  const dim3 grid((unsigned)((M + MP_ROWS - 1) / MP_ROWS)), block(512);
  if (gather)
    hipLaunchKernelGGL(k_mlp2_f16<true>, grid, block, MP_LDS, (hipStream_t)stream, (const __half*)x, ldx, x_rows, gather, (const mp_u4*)w1image, (const __half*)b1, K1,
                       (const mp_u4*)w2image, (const __half*)b2, (const __half*)residual, (__half*)y, ldy, M);
  else
    hipLaunchKernelGGL(k_mlp2_f16<false>, grid, block, MP_LDS, (hipStream_t)stream, (const __half*)x, ldx, x_rows, gather, (const mp_u4*)w1image, (const __half*)b1, K1,
                       (const mp_u4*)w2image, (const __half*)b2, (const __half*)residual, (__half*)y, ldy, M);
  return check_launch("devo_upd_mlp2_f16");
}

}  // extern "C"
