// Dense layers of the Update operator in fp32 STORAGE on the fp16 matrix cores (devo/enet.py:41-78, blocks.py:15-48: the Linear layers
// the reference hands to cuBLAS).
//
// The training step (BASELINE configurations 3 / 4) spends 38 % of its GPU time in fp32 library GEMMs of ONE shape family — 18 000 edge
// rows x 384 x 384 — at 57-75 us each (hipBLASLt on the fp32 matrix pipe: gfx950 has no TF32 / xf32 path).  Here the same product runs
// on v_mfma_f32_16x16x32_f16 with every fp32 value split into fp16 hi + lo (x = hi + lo to 2^-22 relative; the altcorr lookup's
// arithmetic, corr_mm.h):  x y = hi lo' + lo hi' + hi hi'  with fp32 accumulation — 3 dense MFMAs per 16 x 16 x 32 block where the fp32 pipe
// needs 8 of its own, at 16 x their rate.
//
// fp16 has 5 exponent bits: a gradient row of magnitude 1e-7 would lose its lo part (and most of its hi part) to the subnormal range.
// So both operands are scaled by powers of two (exact) before the split and the product is scaled back in the epilogue:
//   * every weight column n by s_n = 2^(8 - exponent(max_k |W[n][k]|)), once per weight version (devo_upd_split_weight);
//   * every activation row by a running scale: set from the first K step's values (max -> [2^8, 2^9)), and raised — with the row's
//     accumulators rescaled by the same power of two — whenever a later value of the row would leave the fp16 range (> 2^14 scaled).
//     Values smaller than 2^-11 of the scale's reference lose lo bits, an error of < 2^-33 of that reference: far below the fp32
//     accumulation's own rounding.
// The weight is split ONCE per version into the B-operand image (one 1 KB piece per (K step, column tile, hi | lo), lane-linear); the
// activations are split in registers on their way from memory.
//
// One workgroup = 128 rows x 96 columns (4 waves x 32 rows, 6 column tiles): per K step of 32 the 12 KB of weight pieces arrive by
// LDS-DMA into a ring of three stages (two steps ahead, one barrier per step); each wave DMAs its own 32 rows x 128 bytes of activations
// into its LDS slab (eight lanes per row: full lines; one step ahead), reads them back in the A-operand layout (slots XOR-swizzled by
// row: conflict-free), scales + splits them and issues 36 MFMAs: every weight fragment read from LDS feeds two row tiles.  Workgroups
// of one row block run on one XCD (their column blocks re-read the rows from that XCD's L2).  The result tile goes through LDS so that
// rows leave as whole 16-byte pieces, scaled back, with bias, ReLU (from a given column on) and an optional residual applied.
// 53 760 B of LDS and 106 VGPRs: three workgroups per CU.  k_linear_f16 at the end of the file is the same shape for fp16 storage.
#include "common.h"
#include <hip/hip_fp16.h>
#include <algorithm>
#include <map>
#include <vector>

namespace devo {

typedef _Float16 ln_h8 __attribute__((ext_vector_type(8)));
typedef float ln_f4 __attribute__((ext_vector_type(4)));
typedef unsigned ln_u4 __attribute__((ext_vector_type(4)));

constexpr int LN_NT = 6;                          // column tiles of 16 per workgroup (96 columns)
constexpr int LN_MT = 2;                          // row tiles of 16 per wave
constexpr int LN_BN = LN_NT * 16;
constexpr int LN_BM = 4 * LN_MT * 16;             // rows per workgroup (4 waves x 32)
constexpr int LN_STAGE = LN_NT * 2 * 1024;        // bytes of weight pieces per K step (hi | lo per tile)
constexpr int LN_NSTAGE = 3;
constexpr int LN_EPI_LD = LN_BN + 4;              // row pitch (floats) of the result tile in LDS
constexpr int LN_TILE_BYTES = 16 * LN_EPI_LD * 4;          // one row tile of a wave's result at a time
constexpr int LN_RING = (LN_NSTAGE * LN_STAGE > 4 * LN_TILE_BYTES) ? LN_NSTAGE * LN_STAGE : 4 * LN_TILE_BYTES;
constexpr int LN_SLAB = LN_MT * 16 * 128;         // one wave's activations of one K step: 32 rows x 128 bytes
constexpr int LN_AUX = LN_RING + 4 * LN_SLAB;     // behind the ring and the four slabs:
constexpr int LN_LDS = LN_AUX + 4 * LN_MT * 16 * 4;   // one float per row and wave (rescale factors, then the inverse row scales): 53 760 B = 42 granules of 1 280 B, three workgroups per CU
constexpr int LN_EXP_TARGET = 8;                  // a scale puts its reference magnitude into [2^8, 2^9)
constexpr float LN_RAISE = 16384.f;               // ... and is raised when a scaled value exceeds 2^14 (fp16's largest: 65504)

// 8 fp32 values -> fp16 hi (8) and lo (8): hi = rn(x), lo = rn(x - hi)
__device__ __forceinline__ void ln_split8(const float (&x)[8], ln_h8& hi, ln_h8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[j]) : "v"(x[2 * j]), "v"(x[2 * j + 1]));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l[j]) : "v"(h[j]), "v"(x[2 * j]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l[j]) : "v"(h[j]), "v"(x[2 * j + 1]));
  }
  const ln_u4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
  hi = __builtin_bit_cast(ln_h8, hv);
  lo = __builtin_bit_cast(ln_h8, lv);
}

// biased exponent of the power of two that brings magnitude m into [2^TARGET, 2^(TARGET+1))
__device__ __forceinline__ int ln_scale_exp(float m) {
  int e = (int)((__float_as_uint(m) >> 23) & 255u);
  e = e < 16 ? 16 : e;                             // zero / tiny reference: a finite scale (2^(8 + 111))
  return 127 + LN_EXP_TARGET + 127 - e;            // in [16, 246]
}
__device__ __forceinline__ float ln_pow2(int biased) { return __uint_as_float((unsigned)(biased < 0 ? 0 : (biased > 254 ? 254 : biased)) << 23); }

// W (element (n, k) at W[n * s_n + k * s_k]: the forward's weight [N, K], or its transpose view for dX = dY W) -> the B-operand image
// [N / 96][K / 32][6 tiles][hi | lo][64 lanes][16 B]: lane (n, kg) of tile t holds k = 32 s + 8 kg .. + 7 of column 96 nb + 16 t + n,
// scaled by the column's power of two; then N floats: the inverse column scales.
// pass 1: one wave per column n: its scale exponent (as the inverse scale's slot holds it for pass 2) — the largest |W[n][.]|
__global__ __launch_bounds__(256) void k_weight_scales(const float* __restrict__ W, int64_t s_n, int64_t s_k, int N, int n_real, int K, float* __restrict__ inv) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // N here: the column count rounded up to whole blocks of 96
  if (n >= N) return;
  float m = 0.f;
  if (n < n_real)
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(W[(int64_t)n * s_n + (int64_t)k * s_k]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) inv[n] = ln_pow2(254 - ln_scale_exp(m));
}

__global__ __launch_bounds__(256) void k_split_weight(const float* __restrict__ W, int64_t s_n, int64_t s_k, int N, int n_real, int K, ln_u4* __restrict__ out) {
  const int nk = (K + 31) / 32;
  const long long total = (long long)(N / LN_BN) * nk * LN_NT * 64;
  const float* inv = reinterpret_cast<const float*>(out) + (size_t)N * nk * 32;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    const long long r = i >> 6;
    const int t = (int)(r % LN_NT);
    const long long r2 = r / LN_NT;
    const int s = (int)(r2 % nk), nb = (int)(r2 / nk);
    const int n = nb * LN_BN + 16 * t + (lane & 15), kg = lane >> 4, k0 = 32 * s + 8 * kg;
    const float sc = ln_pow2(254 - (int)(__float_as_uint(inv[n]) >> 23));      // the reciprocal of a power of two
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (k0 + j < K && n < n_real) ? W[(int64_t)n * s_n + (int64_t)(k0 + j) * s_k] * sc : 0.f;      // (K rounded up to whole steps, N to whole blocks: zeros)
    ln_h8 hi, lo;
    ln_split8(v, hi, lo);
    ln_u4* dst = out + ((r2 * LN_NT + t) * 2) * 64 + lane;
    dst[0] = __builtin_bit_cast(ln_u4, hi);
    dst[64] = __builtin_bit_cast(ln_u4, lo);
  }
}

// LDS-DMA: 16 bytes per lane from (descriptor, per-lane offset, scalar offset) to lds_addr + 16 * lane; hipcc's waitcnt insertion does
// not know it (the loop counts its own vmcnt before the barrier)
__device__ __forceinline__ void ln_dma16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned lds_addr) {
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(lds_addr), "s"(soff) : "memory");
}

// y[M, N] = act(x[M, K] B + bias), B = the split weight image.  grid = 8 * ceil(row blocks / 8) * (N / 96), one row block's column
// blocks on one XCD.
template <bool TRACE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_linear_split(
    const float* __restrict__ x, int64_t ldx, const ln_u4* __restrict__ wsplit, const float* __restrict__ bias, const float* residual,
    const float* __restrict__ gate, float* y, int64_t ldy, int M, int N, int K, int relu_from, int dbg, unsigned long long* wgtrace) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char ln_lds[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, mi = lane & 15, kg = lane >> 4;
  const int NB = (N + LN_BN - 1) / LN_BN, Np = NB * LN_BN, nk = (K + 31) / 32;
  // DEVO_LN_DBG: 1 no activation loads, 2 no stores, 4 no weight DMA, 8 no MFMAs, 16 cycle stamps of workgroup 0 into y[0][..]
  unsigned long long tst[32];
  int nst = 0;
  auto stamp = [&]() { if constexpr (TRACE) { if (nst < 32) tst[nst] = __builtin_readcyclecounter(); nst++; } };
  stamp();
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = (slot / NB) * 8 + xcd, nb = slot - (slot / NB) * NB;
  if (rb * LN_BM >= M) return;
  constexpr unsigned OFF_NONE = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)(((int64_t)(M - 1) * ldx + K) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<ln_u4*>(wsplit), 0, (unsigned)((int64_t)Np * nk * 128), 0x00020000);
  const int row_w = rb * LN_BM + LN_MT * 16 * wv;                     // this wave's first row
  // Activations: each wave brings its own 32 rows x 32 floats of a K step into its LDS slab by lane-linear DMA — eight lanes per row, a
  // quad of lanes inside one 128-byte line (16 addresser cycles per KB; 16 bytes per lane at a row stride cost 64) — and reads them back
  // in the A-operand layout.  Slot p of row r holds the row's piece p ^ (r & 7): the 8 rows a ds_read_b128 serves per cycle then sit
  // in different banks.
  unsigned avoff[4];                                                  // DMA q: row 8 q + lane / 8, piece (lane % 8) ^ (row % 8)
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = row_w + 8 * q + (lane >> 3);
    avoff[q] = (row < M && !(dbg & 1)) ? (unsigned)(((int64_t)row * ldx) * 4 + 16 * ((lane & 7) ^ ((lane >> 3) & 7))) : OFF_NONE;
  }
  unsigned aslot[LN_MT][2];                                           // byte offsets of this lane's two pieces (channels 8 kg .. + 7) of row 16 mt + mi
#pragma unroll
  for (int mt = 0; mt < LN_MT; mt++)
#pragma unroll
    for (int h = 0; h < 2; h++) aslot[mt][h] = (unsigned)((16 * mt + mi) * 128 + 16 * ((2 * kg + h) ^ (mi & 7)));
  const unsigned lds0 = (unsigned)(uintptr_t)ln_lds;
  unsigned char* slab = ln_lds + LN_RING + wv * LN_SLAB;
  float* rowf = reinterpret_cast<float*>(ln_lds + LN_AUX) + wv * (LN_MT * 16);
  float* colf = reinterpret_cast<float*>(ln_lds + LN_RING);          // [inverse scale | bias][96]: in the first slab once the K loop is over
  float col_inv = 0.f, col_bias = 0.f;                                // (fetched now, stored then)
  if (tid < LN_BN) {
    col_inv = reinterpret_cast<const float*>(wsplit)[(size_t)Np * nk * 32 + nb * LN_BN + tid];
    col_bias = (bias && nb * LN_BN + tid < N) ? bias[nb * LN_BN + tid] : 0.f;
  }
  // requests past the last K step keep the pipeline's shape (the compiler's and the loop's own vmcnt bookkeeping see ONE path) but
  // carry the out-of-range offset: no memory access, zeros back
  auto stage = [&](int s) {                                          // this wave's 3 of the stage's 12 one-KB pieces
    const int buf = s % LN_NSTAGE;
    const unsigned voff = (s < nk && !(dbg & 4)) ? (unsigned)lane * 16u : OFF_NONE;
#pragma unroll
    for (int q = 0; q < LN_NT * 2 / 4; q++) {
      const int piece = wv + 4 * q;
      ln_dma16(voff, rsw, (unsigned)(((nb * nk + s) * (LN_NT * 2) + piece) * 1024), lds0 + (unsigned)(buf * LN_STAGE + piece * 1024));
    }
  };
  auto load_a = [&](int s) {                                         // K step s of this wave's rows -> its slab
#pragma unroll
    for (int q = 0; q < 4; q++)
      ln_dma16(s < nk ? avoff[q] : OFF_NONE, rsx, (unsigned)s * 128u, lds0 + (unsigned)(LN_RING + wv * LN_SLAB + q * 1024));
  };
  ln_f4 acc[LN_MT][LN_NT];
#pragma unroll
  for (int mt = 0; mt < LN_MT; mt++)
#pragma unroll
    for (int t = 0; t < LN_NT; t++) acc[mt][t] = ln_f4{0.f, 0.f, 0.f, 0.f};
  int esc[LN_MT];                                                     // biased exponent of each row's scale
  float sc[LN_MT];
#pragma unroll
  for (int mt = 0; mt < LN_MT; mt++) { esc[mt] = 0; sc[mt] = 0.f; }

  load_a(0);
  stage(0);
  stage(1);
  asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                    // (the second stage's three requests may stay in flight)
  __syncthreads();
  stamp();

  auto step = [&](int s) {
    // this step's activations out of the slab, then the slab is free for the next step's (requested one step ahead; the weight pieces two)
    ln_u4 cur[LN_MT][2];
#pragma unroll
    for (int mt = 0; mt < LN_MT; mt++)
#pragma unroll
      for (int h = 0; h < 2; h++) cur[mt][h] = *reinterpret_cast<const ln_u4*>(slab + aslot[mt][h]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    load_a(s + 1);
    stage(s + 2);
    stamp();
    // ---- scale: this lane's 2 x 8 values against the row scales
    float xv[LN_MT][8];
    float mx[LN_MT];
    bool raise = false;
#pragma unroll
    for (int mt = 0; mt < LN_MT; mt++) {
      __builtin_memcpy(&xv[mt][0], &cur[mt][0], 16);
      __builtin_memcpy(&xv[mt][4], &cur[mt][1], 16);
      if (32 * s + 32 > K) {                                           // the last step of a K that is not a multiple of 32: what lies behind the row is not part of it
#pragma unroll
        for (int j = 0; j < 8; j++) xv[mt][j] = 32 * s + 8 * kg + j < K ? xv[mt][j] : 0.f;
      }
      float m = fmaxf(fmaxf(fabsf(xv[mt][0]), fabsf(xv[mt][1])), fabsf(xv[mt][2]));
      m = fmaxf(fmaxf(m, fabsf(xv[mt][3])), fabsf(xv[mt][4]));
      m = fmaxf(fmaxf(m, fabsf(xv[mt][5])), fmaxf(fabsf(xv[mt][6]), fabsf(xv[mt][7])));
      mx[mt] = m;
      raise = raise || !(m * sc[mt] <= LN_RAISE);                     // (also: NaN)
    }
    if (s == 0 || __builtin_amdgcn_ballot_w64(raise) != 0ull) {      // rare after the first step: new scales, accumulators follow
#pragma unroll
      for (int mt = 0; mt < LN_MT; mt++) {
        float m = mx[mt];
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const bool need = s == 0 || !(m * sc[mt] <= LN_RAISE);
        const int e_new = need ? ln_scale_exp(m) : esc[mt];
        if (kg == 0) rowf[16 * mt + mi] = s == 0 ? 1.f : ln_pow2(127 + e_new - esc[mt]);
        esc[mt] = e_new;
        sc[mt] = ln_pow2(e_new);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (s != 0) {
#pragma unroll
        for (int mt = 0; mt < LN_MT; mt++) {
          const ln_f4 f = *reinterpret_cast<const ln_f4*>(rowf + 16 * mt + 4 * kg);
#pragma unroll
          for (int t = 0; t < LN_NT; t++) acc[mt][t] *= f;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    ln_h8 ah[LN_MT], al[LN_MT];
#pragma unroll
    for (int mt = 0; mt < LN_MT; mt++) {
#pragma unroll
      for (int j = 0; j < 8; j++) xv[mt][j] *= sc[mt];
      ln_split8(xv[mt], ah[mt], al[mt]);
    }
    const ln_u4* sb = reinterpret_cast<const ln_u4*>(ln_lds + (s % LN_NSTAGE) * LN_STAGE) + lane;
    stamp();
    if (!(dbg & 8))
#pragma unroll
    for (int tg = 0; tg < LN_NT; tg += 3) {
      ln_h8 bh[3], bl[3];
#pragma unroll
      for (int u = 0; u < 3; u++) {
        bh[u] = __builtin_bit_cast(ln_h8, sb[((tg + u) * 2 + 0) * 64]);
        bl[u] = __builtin_bit_cast(ln_h8, sb[((tg + u) * 2 + 1) * 64]);
      }
#pragma unroll
      for (int u = 0; u < 3; u++)                                      // small terms first; the same accumulator again 6 MFMAs later
#pragma unroll
        for (int mt = 0; mt < LN_MT; mt++) acc[mt][tg + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl[u], acc[mt][tg + u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 3; u++)
#pragma unroll
        for (int mt = 0; mt < LN_MT; mt++) acc[mt][tg + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh[u], acc[mt][tg + u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 3; u++)
#pragma unroll
        for (int mt = 0; mt < LN_MT; mt++) acc[mt][tg + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh[u], acc[mt][tg + u], 0, 0, 0);
    }
    // step s + 1's activations (requested in this step) and weight pieces (one step ago) have landed; this step's 3 weight requests may stay in flight
    stamp();
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    stamp();
    __syncthreads();                                                   // every wave is done with this stage's buffer
    stamp();
  };
  for (int s = 0; s < nk; s++) step(s);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the trailing (empty) requests, before the ring becomes the result tile
  if (tid < LN_BN) { colf[tid] = col_inv; colf[LN_BN + tid] = col_bias; }
  __syncthreads();
  // ---- result: lane (n, rg) holds rows 4 rg + r of column 16 t + n -> this wave's [16][96 + 4] tile in LDS (one row tile at a time:
  //      three workgroups per CU) -> whole rows out
  float* tile = reinterpret_cast<float*>(ln_lds) + wv * (16 * LN_EPI_LD);
  const int col0 = nb * LN_BN;
  constexpr int PPR = LN_BN / 4;                                       // 16-byte pieces per row
  const bool vec_out = (ldy & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(gate)) & 15) == 0;
#pragma unroll
  for (int mt = 0; mt < LN_MT; mt++) {
    if (kg == 0) rowf[16 * mt + mi] = ln_pow2(254 - esc[mt]);
#pragma unroll
    for (int t = 0; t < LN_NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) tile[(4 * kg + r) * LN_EPI_LD + 16 * t + mi] = acc[mt][t][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 16 * PPR / 64; it++) {
      const int idx = it * 64 + lane, r = idx / PPR, c4 = idx - r * PPR;
      ln_f4 v = *reinterpret_cast<const ln_f4*>(tile + r * LN_EPI_LD + 4 * c4);
      v = v * rowf[16 * mt + r] * *reinterpret_cast<const ln_f4*>(colf + 4 * c4) + *reinterpret_cast<const ln_f4*>(colf + LN_BN + 4 * c4);
      if (col0 + 4 * c4 >= relu_from) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const int row = row_w + 16 * mt + r, col = col0 + 4 * c4;
      if (row < M && col < N && !(dbg & 2)) {
        float* dst = y + (int64_t)row * ldy + col;
        const float* res = residual ? residual + (int64_t)row * ldy + col : nullptr;      // (may be y itself: read, then written, by this lane)
        const float* gt = gate ? gate + (int64_t)row * ldy + col : nullptr;      // a ReLU's output: this gradient passes where it did not clip
        if (vec_out && col + 4 <= N) {
          if (gt) {
            const ln_f4 q = *reinterpret_cast<const ln_f4*>(gt);
            v.x = q.x > 0.f ? v.x : 0.f; v.y = q.y > 0.f ? v.y : 0.f; v.z = q.z > 0.f ? v.z : 0.f; v.w = q.w > 0.f ? v.w : 0.f;
          }
          if (res) v += *reinterpret_cast<const ln_f4*>(res);
          *reinterpret_cast<ln_f4*>(dst) = v;
        } else {                                                       // rows that are not 16-byte aligned (the corr MLP's 882 columns), the last columns of such a row
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (col + e < N) dst[e] = ((!gt || gt[e] > 0.f) ? v[e] : 0.f) + (res ? res[e] : 0.f);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                   // the tile is read before the next row tile overwrites it
  }
  stamp();
  if (TRACE && wgtrace && tid == 0) {                                  // DEVO_LN_DBG = 48: every workgroup's start / end / hardware id
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    wgtrace[4 * blockIdx.x + 0] = tst[0]; wgtrace[4 * blockIdx.x + 1] = __builtin_readcyclecounter();
    wgtrace[4 * blockIdx.x + 2] = hw; wgtrace[4 * blockIdx.x + 3] = xcc;
  }
  if (TRACE && !wgtrace && blockIdx.x == 0 && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 32; i++) y[i] = i < nst ? (float)(long long)(tst[i] - tst[0]) : -1.f;
  }
}


// ---------------------------------------------------------------------------------------------------------------- fp16 storage
// The same workgroup shape for the operator's inference precision (devo.py:71-77 runs the update operator under autocast: fp16 rows and
// weights, fp32 accumulation): no split, no scales — one MFMA per 16 x 16 x 32 block, K steps of 64 (the row slabs, the weight ring and
// the request counts are the fp32 kernel's byte for byte).
// W fp16 (element (n, k) at W[n * s_n + k * s_k]) -> [N / 96][ceil(K / 64)][6 tiles][2 halves of the step][64 lanes][16 B]: lane (n, kg)
// of tile t, half j holds k = 64 s + 16 kg + 8 j .. + 7 of column 96 nb + 16 t + n (zeros past K).
__global__ __launch_bounds__(256) void k_pack_weight_f16(const __half* __restrict__ W, int64_t s_n, int64_t s_k, int N, int K, ln_u4* __restrict__ out) {
  const int nk = (K + 63) / 64;
  const long long total = (long long)(N / LN_BN) * nk * LN_NT * 2 * 64;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int j = (int)(r & 1); r >>= 1;
    const int t = (int)(r % LN_NT); r /= LN_NT;
    const int s = (int)(r % nk), nb = (int)(r / nk);
    const int n = nb * LN_BN + 16 * t + (lane & 15), k0 = 64 * s + 16 * (lane >> 4) + 8 * j;
    ln_h8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = k0 + e < K ? (_Float16)__half2float(W[(int64_t)n * s_n + (int64_t)(k0 + e) * s_k]) : (_Float16)0.f;
    out[i] = __builtin_bit_cast(ln_u4, v);
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_linear_f16(
    const __half* __restrict__ x, int64_t ldx, const ln_u4* __restrict__ wimg, const __half* __restrict__ bias, const __half* residual,
    __half* y, int64_t ldy, int M, int N, int K, int relu_from) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char ln_lds[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, mi = lane & 15, kg = lane >> 4;
  const int NB = N / LN_BN, nk = (K + 63) / 64;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = (slot / NB) * 8 + xcd, nb = slot - (slot / NB) * NB;
  if (rb * LN_BM >= M) return;
  constexpr unsigned OFF_NONE = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(x), 0, (unsigned)(((int64_t)(M - 1) * ldx + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<ln_u4*>(wimg), 0, (unsigned)((int64_t)N * nk * 128), 0x00020000);
  const int row_w = rb * LN_BM + LN_MT * 16 * wv;
  unsigned avoff[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = row_w + 8 * q + (lane >> 3);
    avoff[q] = row < M ? (unsigned)(((int64_t)row * ldx) * 2 + 16 * ((lane & 7) ^ ((lane >> 3) & 7))) : OFF_NONE;
  }
  unsigned aslot[LN_MT][2];
#pragma unroll
  for (int mt = 0; mt < LN_MT; mt++)
#pragma unroll
    for (int h = 0; h < 2; h++) aslot[mt][h] = (unsigned)((16 * mt + mi) * 128 + 16 * ((2 * kg + h) ^ (mi & 7)));
  const unsigned lds0 = (unsigned)(uintptr_t)ln_lds;
  unsigned char* slab = ln_lds + LN_RING + wv * LN_SLAB;
  float* colf = reinterpret_cast<float*>(ln_lds + LN_RING);          // bias [96], in the first slab once the K loop is over
  float col_bias = 0.f;
  if (tid < LN_BN && bias) col_bias = __half2float(bias[nb * LN_BN + tid]);
  auto stage = [&](int s) {
    const int buf = s % LN_NSTAGE;
    const unsigned voff = s < nk ? (unsigned)lane * 16u : OFF_NONE;
#pragma unroll
    for (int q = 0; q < LN_NT * 2 / 4; q++) {
      const int piece = wv + 4 * q;
      ln_dma16(voff, rsw, (unsigned)(((nb * nk + s) * (LN_NT * 2) + piece) * 1024), lds0 + (unsigned)(buf * LN_STAGE + piece * 1024));
    }
  };
  auto load_a = [&](int s) {
#pragma unroll
    for (int q = 0; q < 4; q++)
      ln_dma16(s < nk ? avoff[q] : OFF_NONE, rsx, (unsigned)s * 128u, lds0 + (unsigned)(LN_RING + wv * LN_SLAB + q * 1024));
  };
  ln_f4 acc[LN_MT][LN_NT];
#pragma unroll
  for (int mt = 0; mt < LN_MT; mt++)
#pragma unroll
    for (int t = 0; t < LN_NT; t++) acc[mt][t] = ln_f4{0.f, 0.f, 0.f, 0.f};
  load_a(0);
  stage(0);
  stage(1);
  asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < nk; s++) {
    ln_h8 a[LN_MT][2];
#pragma unroll
    for (int mt = 0; mt < LN_MT; mt++)
#pragma unroll
      for (int h = 0; h < 2; h++) a[mt][h] = __builtin_bit_cast(ln_h8, *reinterpret_cast<const ln_u4*>(slab + aslot[mt][h]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    load_a(s + 1);
    stage(s + 2);
    if (64 * s + 64 > K) {                                             // the last step of a K that is not a multiple of 64: what lies behind the row is not part of it
#pragma unroll
      for (int mt = 0; mt < LN_MT; mt++)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
          for (int e = 0; e < 8; e++) a[mt][h][e] = 64 * s + 16 * kg + 8 * h + e < K ? a[mt][h][e] : (_Float16)0.f;
    }
    const ln_u4* sb = reinterpret_cast<const ln_u4*>(ln_lds + (s % LN_NSTAGE) * LN_STAGE) + lane;
#pragma unroll
    for (int tg = 0; tg < LN_NT; tg += 3) {
      ln_h8 b0[3], b1[3];
#pragma unroll
      for (int u = 0; u < 3; u++) {
        b0[u] = __builtin_bit_cast(ln_h8, sb[((tg + u) * 2 + 0) * 64]);
        b1[u] = __builtin_bit_cast(ln_h8, sb[((tg + u) * 2 + 1) * 64]);
      }
#pragma unroll
      for (int u = 0; u < 3; u++)
#pragma unroll
        for (int mt = 0; mt < LN_MT; mt++) acc[mt][tg + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][0], b0[u], acc[mt][tg + u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 3; u++)
#pragma unroll
        for (int mt = 0; mt < LN_MT; mt++) acc[mt][tg + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][1], b1[u], acc[mt][tg + u], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid < LN_BN) colf[tid] = col_bias;
  __syncthreads();
  float* tile = reinterpret_cast<float*>(ln_lds) + wv * (16 * LN_EPI_LD);
  const int col0 = nb * LN_BN;
  constexpr int PPR = LN_BN / 4;
#pragma unroll
  for (int mt = 0; mt < LN_MT; mt++) {
#pragma unroll
    for (int t = 0; t < LN_NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) tile[(4 * kg + r) * LN_EPI_LD + 16 * t + mi] = acc[mt][t][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 16 * PPR / 64; it++) {
      const int idx = it * 64 + lane, r = idx / PPR, c4 = idx - r * PPR;
      ln_f4 v = *reinterpret_cast<const ln_f4*>(tile + r * LN_EPI_LD + 4 * c4) + *reinterpret_cast<const ln_f4*>(colf + 4 * c4);
      if (col0 + 4 * c4 >= relu_from) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const int row = row_w + 16 * mt + r;
      if (row < M) {
        __half* dst = y + (int64_t)row * ldy + col0 + 4 * c4;
        if (residual) {
          const uint2 rr = *reinterpret_cast<const uint2*>(residual + (int64_t)row * ldy + col0 + 4 * c4);
          const float2 r0 = __half22float2(*reinterpret_cast<const __half2*>(&rr.x)), r1 = __half22float2(*reinterpret_cast<const __half2*>(&rr.y));
          v.x += r0.x; v.y += r0.y; v.z += r1.x; v.w += r1.y;
        }
        const __half2 o0 = __floats2half2_rn(v.x, v.y), o1 = __floats2half2_rn(v.z, v.w);
        uint2 o;
        o.x = *reinterpret_cast<const unsigned*>(&o0); o.y = *reinterpret_cast<const unsigned*>(&o1);
        *reinterpret_cast<uint2*>(dst) = o;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace devo

using namespace devo;

extern "C" {

size_t devo_upd_split_weight_bytes(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  const size_t Np = (size_t)(N + LN_BN - 1) / LN_BN * LN_BN;
  return Np * ((K + 31) / 32 * 32) * 4 + Np * 4;
}

int devo_upd_split_weight(const float* W, int64_t s_n, int64_t s_k, int N, int K, void* wsplit, devo_stream_t stream) {
  DEVO_REQUIRE(N > 0 && K > 0, "devo_upd_split_weight: bad sizes (%d x %d)", N, K);
  DEVO_REQUIRE(W && wsplit && (reinterpret_cast<uintptr_t>(wsplit) & 15) == 0, "devo_upd_split_weight: null / unaligned tensor");
  const int nk = (K + 31) / 32, Np = (N + LN_BN - 1) / LN_BN * LN_BN;
  const long long total = (long long)(Np / LN_BN) * nk * LN_NT * 64;
  hipLaunchKernelGGL(k_weight_scales, dim3((unsigned)((Np + 3) / 4)), dim3(256), 0, (hipStream_t)stream, W, s_n, s_k, Np, N, K, reinterpret_cast<float*>(wsplit) + (size_t)Np * nk * 32);
  hipLaunchKernelGGL(k_split_weight, dim3((unsigned)blocks_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, W, s_n, s_k, Np, N, K, (ln_u4*)wsplit);
  return check_launch("devo_upd_split_weight");
}

int devo_upd_linear_split(const float* x, int64_t ldx, const void* wsplit, const float* bias, const float* residual, const float* gate, float* y,
                          int64_t ldy, int M, int N, int K, int relu_from, devo_stream_t stream) {
  DEVO_REQUIRE(M >= 0 && N > 0 && K > 0, "devo_upd_linear_split: bad sizes (%d x %d x %d)", M, N, K);
  if (M == 0) return DEVO_OK;
  DEVO_REQUIRE(x && wsplit && y && ldx >= K && ldy >= N, "devo_upd_linear_split: null tensor or rows shorter than the matrix");
  DEVO_REQUIRE(((reinterpret_cast<uintptr_t>(wsplit)) & 15) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias) |
                 reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(gate)) & 3) == 0, "devo_upd_linear_split: the weight image must be 16-byte aligned, everything else 4-byte");
  const int nk = (K + 31) / 32, NB = (N + LN_BN - 1) / LN_BN;
  DEVO_REQUIRE(((int64_t)(M - 1) * ldx + K) * 4 < (1LL << 31) && (int64_t)NB * LN_BN * nk * 128 < (1LL << 31), "devo_upd_linear_split: operand beyond 2 GB");
  static_assert(LN_LDS <= 64 * 1024, "the workgroup's LDS fits the default dynamic limit");
  const int RB = (M + LN_BM - 1) / LN_BM;
  static const int dbg = getenv("DEVO_LN_DBG") ? atoi(getenv("DEVO_LN_DBG")) : 0;
  const unsigned nwg = (unsigned)(((RB + 7) / 8) * 8 * NB);
  unsigned long long* wgtrace = nullptr;
  if ((dbg & 48) == 48) { (void)hipMalloc(&wgtrace, (size_t)nwg * 32); (void)hipMemset(wgtrace, 0, (size_t)nwg * 32); }
  hipLaunchKernelGGL((dbg & 16) ? k_linear_split<true> : k_linear_split<false>, dim3(nwg), dim3(256), LN_LDS, (hipStream_t)stream, x, ldx,
                     (const ln_u4*)wsplit, bias, residual, gate, y, ldy, M, N, K, relu_from < 0 ? 0 : relu_from, dbg, wgtrace);
  if (wgtrace) {                                                      // debug: residency of the launch's workgroups over time, per CU
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nwg * 4);
    (void)hipMemcpy(h.data(), wgtrace, (size_t)nwg * 32, hipMemcpyDeviceToHost);
    (void)hipFree(wgtrace);
    unsigned long long t0 = ~0ull, t1 = 0;
    double sum = 0; int cnt = 0;
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;
    for (unsigned i = 0; i < nwg; i++) {
      if (!h[4 * i + 1]) continue;
      t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 1]); sum += (double)(h[4 * i + 1] - h[4 * i]); cnt++;
      per_cu[((h[4 * i + 3] & 15) << 16) | (h[4 * i + 2] & 0xff00)].push_back({h[4 * i], h[4 * i + 1]});      // XCC | SE, SH, CU
    }
    int maxc = 0; double avgc = 0;
    for (auto& kv : per_cu) {
      int best = 0;
      for (auto& a : kv.second) { int c = 0; for (auto& b : kv.second) c += (b.first <= a.first && a.first < b.second); best = std::max(best, c); }
      maxc = std::max(maxc, best); avgc += best;
    }
    fprintf(stderr, "[linear trace] %d workgroups on %zu CUs, span %llu cycles, mean workgroup %0.f cycles; most workgroups resident on one CU at a time: max %d, mean over CUs %.2f\n",
            cnt, per_cu.size(), t1 - t0, sum / std::max(cnt, 1), maxc, avgc / std::max<size_t>(per_cu.size(), 1));
    unsigned long long last_start = 0;
    for (unsigned i = 0; i < nwg; i++) if (h[4 * i + 1]) last_start = std::max(last_start, h[4 * i] - t0);
    fprintf(stderr, "[linear trace] the last workgroup starts %llu cycles after the first\n", last_start);
  }
  return check_launch("devo_upd_linear_split");
}

size_t devo_upd_pack_weight_f16_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || N % LN_BN != 0) return 0;
  return (size_t)N * ((K + 63) / 64 * 64) * 2;
}

int devo_upd_pack_weight_f16(const void* W, int64_t s_n, int64_t s_k, int N, int K, void* wimage, devo_stream_t stream) {
  DEVO_REQUIRE(N > 0 && K > 0 && N % LN_BN == 0, "devo_upd_pack_weight_f16: N must be a multiple of 96 (got %d x %d)", N, K);
  DEVO_REQUIRE(W && wimage && (reinterpret_cast<uintptr_t>(wimage) & 15) == 0, "devo_upd_pack_weight_f16: null / unaligned tensor");
  const long long total = (long long)(N / LN_BN) * ((K + 63) / 64) * LN_NT * 2 * 64;
  hipLaunchKernelGGL(k_pack_weight_f16, dim3((unsigned)blocks_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const __half*)W, s_n, s_k, N, K, (ln_u4*)wimage);
  return check_launch("devo_upd_pack_weight_f16");
}

int devo_upd_linear_f16(const void* x, int64_t ldx, const void* wimage, const void* bias, const void* residual, void* y, int64_t ldy, int M, int N,
                        int K, int relu_from, devo_stream_t stream) {
  DEVO_REQUIRE(M >= 0 && N > 0 && K > 0 && N % LN_BN == 0, "devo_upd_linear_f16: N must be a multiple of 96 (got %d x %d)", N, K);
  if (M == 0) return DEVO_OK;
  DEVO_REQUIRE(x && wimage && y && ldx >= K && ldy >= N && ldy % 4 == 0 && ldx % 2 == 0, "devo_upd_linear_f16: null tensor, rows shorter than the matrix, input rows that are not a multiple of 2 or output rows that are not a multiple of 4 elements apart");
  DEVO_REQUIRE(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 7) == 0 && (reinterpret_cast<uintptr_t>(wimage) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(x) & 3) == 0 && (reinterpret_cast<uintptr_t>(bias) & 1) == 0, "devo_upd_linear_f16: y / residual must be 8-byte, the weight image 16-byte, x 4-byte aligned");
  const int nk = (K + 63) / 64;
  DEVO_REQUIRE(((int64_t)(M - 1) * ldx + K) * 2 < (1LL << 31) && (int64_t)N * nk * 128 < (1LL << 31), "devo_upd_linear_f16: operand beyond 2 GB");
  const int RB = (M + LN_BM - 1) / LN_BM, NB = N / LN_BN;
  hipLaunchKernelGGL(k_linear_f16, dim3((unsigned)(((RB + 7) / 8) * 8 * NB)), dim3(256), LN_LDS, (hipStream_t)stream, (const __half*)x, ldx, (const ln_u4*)wimage,
                     (const __half*)bias, (const __half*)residual, (__half*)y, ldy, M, N, K, relu_from < 0 ? 0 : relu_from);
  return check_launch("devo_upd_linear_f16");
}

}  // extern "C"
