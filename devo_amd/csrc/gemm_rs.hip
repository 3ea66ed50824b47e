// Dense layers of the Update operator, fp16 storage, second structure (devo/enet.py:41-78, blocks.py:15-48):
// the ROWS of a workgroup live in LDS once, the WEIGHTS go from the L2 straight into the registers of the wave that multiplies them.
//
// linear.hip's shape (128 rows x 96 columns per workgroup, weights and rows through LDS-DMA rings) is bound by what a CU takes in through
// the LDS-DMA path: 8 - 14 B/cycle/CU (profiles/r05_mlp2.txt) where the matrix pipe wants 48.  Vector loads into registers run at 56
// B/cycle/CU for 16 contiguous bytes per lane (profiles/r04_l2_fill.txt).  So here:
//   * one workgroup = 16 MT rows x 384 columns, 8 waves; wave w owns columns 48 w .. 48 w + 47 (3 tiles of 16) of ALL the rows:
//     8 x 3 accumulator tiles;
//   * the rows (16 MT x K halves, <= 98 KB) are loaded ONCE — 16-byte pieces through registers, written to LDS at a row pitch of
//     2 K + 16 bytes (the 8 rows a ds_read_b128 serves per cycle then sit in different bank groups) — and read back as A operands by
//     every wave: 8 KB per wave and K step of 32, 64 KB per CU and step against 768 matrix-pipe cycles per SIMD;
//   * a wave's share of the weight image (36 KB at K = 384: [wave][K step][tile][lane][16 B], one contiguous stream per wave) arrives by
//     buffer_load_dwordx4 three steps ahead of the products — no LDS, no barrier in the K loop, every wait counted by the compiler;
//   * N = 768 (the concatenated gate | res[0] and f | g layers) = two passes over the same rows in LDS;
//   * the chain form (Linear - ReLU - Linear, mlp2.hip's job) writes relu(h) back into the row tile and runs the second layer from there.
// Traffic into a CU per 128 rows: 96 KB of rows + 288 KB of weights (linear.hip: 4 x (96 + 72) KB through LDS-DMA).
#include "common.h"
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace devo {

typedef _Float16 rs_h8 __attribute__((ext_vector_type(8)));
typedef float rs_f4 __attribute__((ext_vector_type(4)));
typedef unsigned rs_u4 __attribute__((ext_vector_type(4)));
typedef _Float16 rs_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rs_f4 rs_cvt4(rs_h4 h) { return __builtin_convertvector(h, rs_f4); }
__device__ __forceinline__ rs_h4 rs_pack4(rs_f4 v) { return __builtin_convertvector(v, rs_h4); }

constexpr int RS_NW = 8;                           // waves per workgroup
constexpr int RS_NT = 3;                           // column tiles of 16 per wave
constexpr int RS_BN = RS_NW * RS_NT * 16;          // 384 columns per pass
constexpr int RS_RB = 4;                           // weight ring in registers: K steps (RS_RB - 1 ahead of the products)
constexpr int RS_EPI_LD = RS_NT * 16 + 4;          // row pitch (floats) of a wave's result tile in LDS

// W fp16 (element (n, k) at W[n * s_n + k * s_k]; N a multiple of 384) -> [N / 384][8 waves][ceil(K / 32)][3 tiles][64 lanes][16 B]:
// lane (n, kg) of tile t holds k = 32 s + 8 kg .. + 7 of column 384 nb + 48 w + 16 t + n (zeros past K)
__global__ __launch_bounds__(256) void k_rs_pack_f16(const __half* __restrict__ W, int64_t s_n, int64_t s_k, int N, int K, rs_u4* __restrict__ out) {
  const int nk = (K + 31) / 32;
  const long long total = (long long)(N / RS_BN) * RS_NW * nk * RS_NT * 64;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int t = (int)(r % RS_NT); r /= RS_NT;
    const int s = (int)(r % nk); r /= nk;
    const int w = (int)(r % RS_NW), nb = (int)(r / RS_NW);
    const int n = nb * RS_BN + 48 * w + 16 * t + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    rs_h8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = k0 + e < K ? (_Float16)__half2float(W[(int64_t)n * s_n + (int64_t)(k0 + e) * s_k]) : (_Float16)0.f;
    out[i] = __builtin_bit_cast(rs_u4, v);
  }
}

// y[M, N] = act(x[M, K] W^T + bias) [+ residual], N = 384 NB.  grid = ceil(M / (16 MT)).  NK = K steps of 32 (K <= 32 NK).
template <int MT, int NK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(MT <= 3 ? 4 : 1, MT <= 3 ? 4 : 8))) void k_rs_linear_f16(const __half* __restrict__ x, int64_t ldx, const rs_u4* __restrict__ wimg,
                                                        const __half* __restrict__ bias, const __half* residual, __half* y, int64_t ldy, int M,
                                                        int NB, int K, int relu_from, unsigned long long* trace) {
  unsigned long long tst[6] = {0, 0, 0, 0, 0, 0};                      // DEVO_RS_TRACE: cycle stamps of every workgroup's wave 0
  tst[0] = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(1024))) unsigned char rs_lds[];
  constexpr int KP = 32 * NK, PITCH = 2 * KP + 16, PPR = KP / 8, ROWS = 16 * MT;
  constexpr int NPIECE = ROWS * PPR, AP = (NPIECE + 511) / 512;
  constexpr unsigned OFF_NONE = 0x80000000u;
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, mi = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.x * ROWS;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(x), 0, (unsigned)(((int64_t)(M - 1) * ldx + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<rs_u4*>(wimg), 0, (unsigned)((int64_t)NB * RS_NW * NK * RS_NT * 1024), 0x00020000);
  // ---- the rows: memory -> registers -> LDS (piece p = 16 bytes: row p / PPR, halves 8 (p % PPR) .. + 7)
  rs_u4 ap[AP];
#pragma unroll
  for (int i = 0; i < AP; i++) {
    const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
    const bool ok = p < NPIECE && row0 + row < M && 8 * c < K;
    ap[i] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? (unsigned)(((int64_t)(row0 + row) * ldx) * 2 + 16 * c) : OFF_NONE, 0, 0);
  }
  // ---- this wave's weights: K step s, tile t = 1 KB at ((nb * 8 + w) * NK + s) * 3 + t
  rs_u4 b[RS_RB][RS_NT];
  const unsigned bvoff = (unsigned)lane * 16u;
  auto load_b = [&](int nb, int s, int slot) {
#pragma unroll
    for (int t = 0; t < RS_NT; t++)
      b[slot][t] = __builtin_amdgcn_raw_buffer_load_b128(rsw, bvoff, (unsigned)((((nb * RS_NW + wv) * NK + s) * RS_NT + t) * 1024), 0);
  };
#pragma unroll
  for (int s = 0; s < RS_RB - 1 && s < NK; s++) load_b(0, s, s);
#pragma unroll
  for (int i = 0; i < AP; i++) {
    const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
    if (8 * c + 8 > K && 8 * c < K) {                                  // a K that is not a multiple of 8: what lies behind the row is not part of it
      rs_h8 v = __builtin_bit_cast(rs_h8, ap[i]);
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = 8 * c + e < K ? v[e] : (_Float16)0.f;
      ap[i] = __builtin_bit_cast(rs_u4, v);
    }
    if (p < NPIECE) *reinterpret_cast<rs_u4*>(rs_lds + row * PITCH + 16 * c) = ap[i];
  }
  tst[1] = __builtin_readcyclecounter();
  __syncthreads();
  tst[2] = __builtin_readcyclecounter();
  const unsigned char* arow = rs_lds + mi * PITCH + 16 * kg;          // this lane's piece of row 16 mt + mi, K step s: + 16 mt PITCH + 64 s
  float* tile = reinterpret_cast<float*>(rs_lds) + wv * (16 * RS_EPI_LD);
  for (int nb = 0; nb < NB; nb++) {
    rs_f4 acc[MT][RS_NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int t = 0; t < RS_NT; t++) acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
    rs_u4 af[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) af[0][mt] = *reinterpret_cast<const rs_u4*>(arow + 16 * mt * PITCH);
#pragma unroll
    for (int s = 0; s < NK; s++) {
      if (s + 1 < NK) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) af[(s + 1) & 1][mt] = *reinterpret_cast<const rs_u4*>(arow + 16 * mt * PITCH + 64 * (s + 1));
      }
      if (s + RS_RB - 1 < NK) load_b(nb, s + RS_RB - 1, (s + RS_RB - 1) % RS_RB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int t = 0; t < RS_NT; t++)
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rs_h8, b[s % RS_RB][t]), __builtin_bit_cast(rs_h8, af[s & 1][mt]), acc[mt][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    tst[3] = __builtin_readcyclecounter();
    // the next pass's first weights travel while this one's results leave
    if (nb + 1 < NB) {
#pragma unroll
      for (int s = 0; s < RS_RB - 1 && s < NK; s++) load_b(nb + 1, s, s);
    }
    // ---- result (the weights are the A operand of the products): lane (i, g) holds row 16 mt + i, columns 16 t + 4 g .. + 3.  The last pass
    //      reuses the row tile's LDS (everybody is done with it); earlier passes write behind it
    const bool last = nb + 1 == NB;
    if (last) __syncthreads();
    float* tl = last ? tile : reinterpret_cast<float*>(rs_lds + ROWS * PITCH) + wv * (16 * RS_EPI_LD);
    const int col_w = nb * RS_BN + 48 * wv;
    rs_f4 bs[RS_NT];
#pragma unroll
    for (int t = 0; t < RS_NT; t++) bs[t] = bias ? rs_cvt4(*reinterpret_cast<const rs_h4*>(reinterpret_cast<const _Float16*>(bias) + col_w + 16 * t + 4 * kg)) : rs_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
      for (int t = 0; t < RS_NT; t++) {
        rs_f4 v = acc[mt][t] + bs[t];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = col_w + 16 * t + 4 * kg + r >= relu_from ? fmaxf(v[r], 0.f) : v[r];
        *reinterpret_cast<rs_f4*>(tl + mi * RS_EPI_LD + 16 * t + 4 * kg) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int it = 0; it < 2; it++) {                                 // 16 rows x 6 pieces of 8 columns
        const int idx = it * 64 + lane, r = idx / 6, c = idx - r * 6;
        const int row = row0 + 16 * mt + r;
        if (idx < 96 && row < M) {
          rs_f4 v0 = *reinterpret_cast<const rs_f4*>(tl + r * RS_EPI_LD + 8 * c), v1 = *reinterpret_cast<const rs_f4*>(tl + r * RS_EPI_LD + 8 * c + 4);
          __half* dst = y + (int64_t)row * ldy + col_w + 8 * c;
          if (residual) {                                              // (may be y itself: read, then written, by this lane)
            const rs_h8 q = __builtin_bit_cast(rs_h8, *reinterpret_cast<const rs_u4*>(residual + (int64_t)row * ldy + col_w + 8 * c));
            v0.x += (float)q[0]; v0.y += (float)q[1]; v0.z += (float)q[2]; v0.w += (float)q[3];
            v1.x += (float)q[4]; v1.y += (float)q[5]; v1.z += (float)q[6]; v1.w += (float)q[7];
          }
          rs_h8 o;
          o[0] = (_Float16)v0.x; o[1] = (_Float16)v0.y; o[2] = (_Float16)v0.z; o[3] = (_Float16)v0.w;
          o[4] = (_Float16)v1.x; o[5] = (_Float16)v1.y; o[6] = (_Float16)v1.z; o[7] = (_Float16)v1.w;
          *reinterpret_cast<rs_u4*>(dst) = __builtin_bit_cast(rs_u4, o);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                                 // the tile is read before the next row tile overwrites it
    }
  }
  if (trace && tid == 0) {
    tst[4] = __builtin_readcyclecounter();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    tst[5] = hw;
#pragma unroll
    for (int i = 0; i < 6; i++) trace[6 * blockIdx.x + i] = tst[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------- row-resident chains
// The same products with the rows STAYING in LDS between layers: what follows the frame-pair aggregation in the update operator is
// row-local (enet.py:52-57, 96-99; blocks.py:29-48):  LN -> GatedResidual -> LN -> GatedResidual -> heads — six 384 x 384 products, two
// LayerNorms, two gates and the two 2-wide heads per edge row.  One workgroup keeps 96 rows (X) and one intermediate (R) in LDS, its
// waves own 48 columns of every product; the res[2] outputs wait in their owners' elements of R while the gate product runs, the
// LayerNorm statistics of a row are summed across the eight waves through LDS.  Rounding points are those of the layer-by-layer path
// (every layer output and LayerNorm output rounded to fp16).
template <int NK>
__device__ __forceinline__ void rs_prefetch(const __amdgpu_buffer_rsrc_t rsw, unsigned wbase, unsigned bvoff, rs_u4 (&b)[RS_RB][RS_NT]) {
#pragma unroll
  for (int s = 0; s < RS_RB - 1 && s < NK; s++)
#pragma unroll
    for (int t = 0; t < RS_NT; t++) b[s][t] = __builtin_amdgcn_raw_buffer_load_b128(rsw, bvoff, wbase + (unsigned)((s * RS_NT + t) * 1024), 0);
}

// acc += rows (LDS tile, this lane's pieces at arow + 16 mt PITCH + 64 s) x this wave's weight stream (steps 0 .. RS_RB - 2 already in b).
// NEXT: the loop's last steps request the first RS_RB - 1 steps of the stream the NEXT loop multiplies (no bubble between products).
template <int MT, int NK, int PITCH, bool NEXT = false>
__device__ __forceinline__ void rs_kloop(const unsigned char* arow, const __amdgpu_buffer_rsrc_t rsw, unsigned wbase, unsigned bvoff, rs_u4 (&b)[RS_RB][RS_NT],
                                         rs_f4 (&acc)[MT][RS_NT], const __amdgpu_buffer_rsrc_t nrsw = __amdgpu_buffer_rsrc_t(), unsigned nwbase = 0) {
  static_assert(NK % RS_RB == 0, "ring slots line up across loops");
  rs_u4 af[2][MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) af[0][mt] = *reinterpret_cast<const rs_u4*>(arow + 16 * mt * PITCH);
#pragma unroll
  for (int s = 0; s < NK; s++) {
    if (s + 1 < NK) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++) af[(s + 1) & 1][mt] = *reinterpret_cast<const rs_u4*>(arow + 16 * mt * PITCH + 64 * (s + 1));
    }
    if (s + RS_RB - 1 < NK) {
#pragma unroll
      for (int t = 0; t < RS_NT; t++)
        b[(s + RS_RB - 1) % RS_RB][t] = __builtin_amdgcn_raw_buffer_load_b128(rsw, bvoff, wbase + (unsigned)(((s + RS_RB - 1) * RS_NT + t) * 1024), 0);
    } else if (NEXT) {
#pragma unroll
      for (int t = 0; t < RS_NT; t++)
        b[(s + RS_RB - 1) % RS_RB][t] = __builtin_amdgcn_raw_buffer_load_b128(nrsw, bvoff, nwbase + (unsigned)(((s + RS_RB - 1 - NK) * RS_NT + t) * 1024), 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int t = 0; t < RS_NT; t++)
        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rs_h8, b[s % RS_RB][t]), __builtin_bit_cast(rs_h8, af[s & 1][mt]), acc[mt][t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ float rs_row16_sum(float v) {               // over the 16 lanes of a DPP row, result in all of them
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
  return v;
}

struct RsGru {
  const __half* x; const __half* hy; const int* grp;                   // rows: x[e] + hy[grp[e]]
  const __half* ln0_g; const __half* ln0_b;
  const rs_u4* w_gr[2]; const __half* b_gr[2];                         // [gate | res[0]] images (N = 768) and biases of the two GatedResiduals
  const rs_u4* w_r2[2]; const __half* b_r2[2];                         // res[2]
  const __half* ln2_g; const __half* ln2_b;
  const __half* Wd; const __half* bd; const __half* Ww; const __half* bw;
  __half* net_out; __half* delta; __half* weight;
  float* net_out32;                                                    // OUT32 instantiation: the new state in fp32 (what autocast returns)
  int E; float eps0, eps2;
  unsigned long long* trace;                                           // debug (DEVO_RS_TRACE): 24 cycle stamps of every workgroup's wave 0
};

constexpr int RG_MT = 6, RG_NK = 12, RG_PITCH = 2 * 32 * RG_NK + 16, RG_ROWS = 16 * RG_MT;
constexpr int RG_VEC = 768 + 384 + 768 + 384 + 384 + 384;             // halves: b_gr[0] | b_r2[0] | b_gr[1] | b_r2[1] | ln2 gamma | ln2 beta
constexpr int RG_LDS = 2 * RG_ROWS * RG_PITCH + RG_VEC * 2 + RG_ROWS * 8 * 2 * 4 + RG_ROWS * 2 * 4;      // 163 584 B
static_assert(RG_LDS <= 160 * 1024, "the chain's LDS");

template <bool OUT32>
__global__ __launch_bounds__(512) void k_rs_gru_f16(RsGru a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char rs_lds[];
  unsigned long long tst[24];
  int nst = 0;
  auto stamp = [&]() { if (a.trace) { if (nst < 24) tst[nst] = __builtin_readcyclecounter(); nst++; } };
  stamp();
  constexpr int MT = RG_MT, NK = RG_NK, PITCH = RG_PITCH, ROWS = RG_ROWS, D = 384;
  unsigned char* X = rs_lds;
  unsigned char* R = rs_lds + ROWS * PITCH;
  _Float16* vec = reinterpret_cast<_Float16*>(rs_lds + 2 * ROWS * PITCH);
  float* part = reinterpret_cast<float*>(rs_lds + 2 * ROWS * PITCH + RG_VEC * 2);      // [row][wave][sum | sum of squares]
  float* stat = part + ROWS * 16;                                                        // [row][mean | rstd]
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, mi = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.x * ROWS, E = a.E;
  const unsigned bvoff = (unsigned)lane * 16u;
  __amdgpu_buffer_rsrc_t rs_gr[2], rs_r2[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    rs_gr[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<rs_u4*>(a.w_gr[i]), 0, 2u * D * D * 2u, 0x00020000);
    rs_r2[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<rs_u4*>(a.w_r2[i]), 0, (unsigned)(D * D * 2), 0x00020000);
  }
  auto wbase = [&](int nb) { return (unsigned)(((nb * RS_NW + wv) * NK) * RS_NT * 1024); };
  rs_u4 b[RS_RB][RS_NT];
  rs_prefetch<NK>(rs_gr[0], wbase(1), bvoff, b);
  // ---- the layers' vectors -> LDS
#pragma unroll
  for (int k = 0; k < (RG_VEC + 511) / 512; k++) {
    const int i = tid + 512 * k;
    if (i < RG_VEC) {
      const __half* src = i < 768 ? a.b_gr[0] + i : i < 1152 ? a.b_r2[0] + (i - 768) : i < 1920 ? a.b_gr[1] + (i - 1152) : i < 2304 ? a.b_r2[1] + (i - 1920)
                          : i < 2688 ? a.ln2_g + (i - 2304) : a.ln2_b + (i - 2688);
      vec[i] = (_Float16)__half2float(*src);
    }
  }
  // ---- S0: X = LN0(x + hy[grp]) — a quarter wave per row, three rounds of 32 rows
  {
    const int l16 = tid & 15;
    rs_u4 xv[3][3], hv[3][3];
#pragma unroll
    for (int p = 0; p < 3; p++) {
      const int row = row0 + 32 * p + (tid >> 4), rr = row < E ? row : E - 1;
      const __half* xr = a.x + (int64_t)rr * D;
      const __half* hr = a.hy + (int64_t)a.grp[rr] * D;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        xv[p][k] = *reinterpret_cast<const rs_u4*>(xr + (l16 + 16 * k) * 8);
        hv[p][k] = *reinterpret_cast<const rs_u4*>(hr + (l16 + 16 * k) * 8);
      }
    }
    rs_u4 gm[3], bt[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      gm[k] = *reinterpret_cast<const rs_u4*>(a.ln0_g + (l16 + 16 * k) * 8);
      bt[k] = *reinterpret_cast<const rs_u4*>(a.ln0_b + (l16 + 16 * k) * 8);
    }
#pragma unroll
    for (int p = 0; p < 3; p++) {
      const int rloc = 32 * p + (tid >> 4);
      const bool live = row0 + rloc < E;
      float t[3][8];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const rs_h8 xh = __builtin_bit_cast(rs_h8, xv[p][k]), hh = __builtin_bit_cast(rs_h8, hv[p][k]);
#pragma unroll
        for (int i = 0; i < 8; i++) { t[k][i] = (float)xh[i] + (float)hh[i]; s += t[k][i]; }
      }
      const float mean = rs_row16_sum(s) / (float)D;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) { const float d = t[k][i] - mean; q += d * d; }
      const float rstd = live ? rsqrtf(rs_row16_sum(q) / (float)D + a.eps0) : 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const rs_h8 g8 = __builtin_bit_cast(rs_h8, gm[k]), b8 = __builtin_bit_cast(rs_h8, bt[k]);
        rs_h8 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = (_Float16)((t[k][i] - mean) * rstd * (float)g8[i] + (float)b8[i]);
        *reinterpret_cast<rs_u4*>(X + rloc * PITCH + (l16 + 16 * k) * 16) = __builtin_bit_cast(rs_u4, o);
      }
    }
  }
  stamp();
  __syncthreads();
  stamp();
  const unsigned char* arowX = X + mi * PITCH + 16 * kg;
  const unsigned char* arowR = R + mi * PITCH + 16 * kg;
  // the weights are the A operand of the products: this lane holds row 16 mt + mi, columns colw + 16 t .. + 3 of a result
  const int colw = 48 * wv + 4 * kg;
  unsigned char* eX = X + mi * PITCH + colw * 2;                       // + 16 mt PITCH + 32 t: 8 bytes
  unsigned char* eR = R + mi * PITCH + colw * 2;
  auto vec4 = [&](int off, int t) { return rs_cvt4(*reinterpret_cast<const rs_h4*>(vec + off + colw + 16 * t)); };
#pragma unroll
  for (int rep = 0; rep < 2; rep++) {
    const int o_gr = rep == 0 ? 0 : 1152, o_r2 = rep == 0 ? 768 : 1920;
    rs_f4 acc[MT][RS_NT];
    // ---- R = relu(X Wr0 + br0)
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int t = 0; t < RS_NT; t++) acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
    rs_kloop<MT, NK, PITCH, true>(arowX, rs_gr[rep], wbase(1), bvoff, b, acc, rs_r2[rep], wbase(0));
    stamp();
#pragma unroll
    for (int t = 0; t < RS_NT; t++) {
      const rs_f4 bs = vec4(o_gr + D, t);
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        rs_f4 v = acc[mt][t] + bs;
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
        *reinterpret_cast<rs_h4*>(eR + 16 * mt * PITCH + 32 * t) = rs_pack4(v);
      }
    }
    stamp();
    __syncthreads();
    stamp();
    // ---- res = R Wr2 + br2 -> this lane's own elements of R (once every wave has read R)
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int t = 0; t < RS_NT; t++) acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
    rs_kloop<MT, NK, PITCH, true>(arowR, rs_r2[rep], wbase(0), bvoff, b, acc, rs_gr[rep], wbase(0));
    stamp();
    __syncthreads();
    stamp();
#pragma unroll
    for (int t = 0; t < RS_NT; t++) {
      const rs_f4 bs = vec4(o_r2, t);
#pragma unroll
      for (int mt = 0; mt < MT; mt++) *reinterpret_cast<rs_h4*>(eR + 16 * mt * PITCH + 32 * t) = rs_pack4(acc[mt][t] + bs);
    }
    // ---- gate = X Wg + bg;  t = x + sigmoid(gate) res
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int t = 0; t < RS_NT; t++) acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
    stamp();
    if (rep == 0) rs_kloop<MT, NK, PITCH, true>(arowX, rs_gr[0], wbase(0), bvoff, b, acc, rs_gr[1], wbase(1));
    else rs_kloop<MT, NK, PITCH>(arowX, rs_gr[1], wbase(0), bvoff, b, acc);
    stamp();
#pragma unroll
    for (int t = 0; t < RS_NT; t++) {
      const rs_f4 bs = vec4(o_gr, t);
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        const int off = 16 * mt * PITCH + 32 * t;
        const rs_f4 gv = rs_cvt4(rs_pack4(acc[mt][t] + bs));           // (rounded where the layer-by-layer path stores it)
        const rs_f4 res = rs_cvt4(*reinterpret_cast<const rs_h4*>(eR + off)), xv = rs_cvt4(*reinterpret_cast<const rs_h4*>(eX + off));
#pragma unroll
        for (int r = 0; r < 4; r++) acc[mt][t][r] = xv[r] + res[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-gv[r]));
      }
    }
    stamp();
    if (rep == 0) {
      // ---- X = LN2(t): a row's sums over this wave's 48 columns (12 in this lane, then over the four lanes of the row), then over the
      //      eight waves through LDS
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int t = 0; t < RS_NT; t++)
#pragma unroll
          for (int r = 0; r < 4; r++) { s += acc[mt][t][r]; q += acc[mt][t][r] * acc[mt][t][r]; }
        s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        if (kg == 0) { part[((16 * mt + mi) * 8 + wv) * 2] = s; part[((16 * mt + mi) * 8 + wv) * 2 + 1] = q; }
      }
      __syncthreads();
      if (tid < ROWS) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 8; w++) { s += part[(tid * 8 + w) * 2]; q += part[(tid * 8 + w) * 2 + 1]; }
        const float mean = s / (float)D, var = fmaxf(q / (float)D - mean * mean, 0.f);
        stat[2 * tid] = mean; stat[2 * tid + 1] = rsqrtf(var + a.eps2);
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < RS_NT; t++) {
        const rs_f4 g = vec4(2304, t), bb = vec4(2688, t);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          const float mean = stat[2 * (16 * mt + mi)], rstd = stat[2 * (16 * mt + mi) + 1];
          *reinterpret_cast<rs_h4*>(eX + 16 * mt * PITCH + 32 * t) = rs_pack4((acc[mt][t] - mean) * rstd * g + bb);
        }
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int t = 0; t < RS_NT; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) *reinterpret_cast<rs_h4*>(eX + 16 * mt * PITCH + 32 * t) = rs_pack4(acc[mt][t]);
      __syncthreads();
    }
  }
  stamp();
  // ---- net out (whole rows) and the two heads on relu(net): a quarter wave per row
  {
    const int l16 = tid & 15;
    rs_u4 wd0[3], wd1[3], ww0[3], ww1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      wd0[k] = *reinterpret_cast<const rs_u4*>(a.Wd + (l16 + 16 * k) * 8); wd1[k] = *reinterpret_cast<const rs_u4*>(a.Wd + D + (l16 + 16 * k) * 8);
      ww0[k] = *reinterpret_cast<const rs_u4*>(a.Ww + (l16 + 16 * k) * 8); ww1[k] = *reinterpret_cast<const rs_u4*>(a.Ww + D + (l16 + 16 * k) * 8);
    }
    const float bd0 = __half2float(a.bd[0]), bd1 = __half2float(a.bd[1]), bw0 = __half2float(a.bw[0]), bw1 = __half2float(a.bw[1]);
#pragma unroll
    for (int p = 0; p < 3; p++) {
      const int rloc = 32 * p + (tid >> 4), row = row0 + rloc;
      const bool live = row < E;
      float d0 = 0.f, d1 = 0.f, w0 = 0.f, w1 = 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const rs_u4 v = *reinterpret_cast<const rs_u4*>(X + rloc * PITCH + (l16 + 16 * k) * 16);
        if constexpr (OUT32) {
          if (live) {
            const rs_h8 h8 = __builtin_bit_cast(rs_h8, v);
            float* op_ = a.net_out32 + (int64_t)row * D + (l16 + 16 * k) * 8;
            *reinterpret_cast<rs_f4*>(op_) = rs_f4{(float)h8[0], (float)h8[1], (float)h8[2], (float)h8[3]};
            *reinterpret_cast<rs_f4*>(op_ + 4) = rs_f4{(float)h8[4], (float)h8[5], (float)h8[6], (float)h8[7]};
          }
        } else {
          if (live) *reinterpret_cast<rs_u4*>(a.net_out + (int64_t)row * D + (l16 + 16 * k) * 8) = v;
        }
        // relu on fp16 pairs, products of fp16 pairs summed in fp32 (v_dot2_f32_f16: exact products, like the fp32 multiply-adds they replace)
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const unsigned vi = v[i], a0 = wd0[k][i], a1 = wd1[k][i], c0 = ww0[k][i], c1 = ww1[k][i];      // (scalars: a bit cast of a vector ELEMENT reads the vector's first one)
          const h2 u = __builtin_elementwise_max(__builtin_bit_cast(h2, vi), h2{(_Float16)0.f, (_Float16)0.f});
          d0 = __builtin_amdgcn_fdot2(u, __builtin_bit_cast(h2, a0), d0, false);
          d1 = __builtin_amdgcn_fdot2(u, __builtin_bit_cast(h2, a1), d1, false);
          w0 = __builtin_amdgcn_fdot2(u, __builtin_bit_cast(h2, c0), w0, false);
          w1 = __builtin_amdgcn_fdot2(u, __builtin_bit_cast(h2, c1), w1, false);
        }
      }
      d0 = rs_row16_sum(d0); d1 = rs_row16_sum(d1); w0 = rs_row16_sum(w0); w1 = rs_row16_sum(w1);
      if (live && l16 == 0) {
        a.delta[(int64_t)row * 2] = __float2half(d0 + bd0); a.delta[(int64_t)row * 2 + 1] = __float2half(d1 + bd1);
        a.weight[(int64_t)row * 2] = __float2half(1.0f / (1.0f + __expf(-(w0 + bw0))));
        a.weight[(int64_t)row * 2 + 1] = __float2half(1.0f / (1.0f + __expf(-(w1 + bw1))));
      }
    }
  }
  stamp();
  if (a.trace && tid == 0) {
#pragma unroll
    for (int i = 0; i < 24; i++) a.trace[24 * blockIdx.x + i] = i < nst ? tst[i] : 0ull;
  }
}

// Linear - ReLU - Linear [+ residual] on gathered rows (enet.py:46-50, 86-91: net + c(mask * net[:, idx])), one launch: the gathered rows in
// X, relu(h) in R, the residual rows (the workgroup's own, not gathered) into X while the second product runs, results through X so that
// whole rows leave.  gather: i64 [M], negative = a zero row; null = the rows themselves.
// MLP = false: no layers, the rows are x + hy[grp] (the expand-add behind a SoftAgg, enet.py:93: written back to y, which may be x).
// FG: behind either, the 768-wide f | g layer of the NEXT SoftAgg (blocks.py:36-40) on the rows still in X: fg [M, 768], two more products.
template <bool MLP, bool FG>
__global__ __launch_bounds__(512) void k_rs_mlp2_f16(const __half* x, int64_t ldx, int x_rows, const int64_t* __restrict__ gather,
                                                      const rs_u4* __restrict__ w1, const __half* __restrict__ b1, const rs_u4* __restrict__ w2,
                                                      const __half* __restrict__ b2, const __half* residual, __half* y, int M,
                                                      const __half* __restrict__ hy, const int* __restrict__ grp, const rs_u4* __restrict__ wfg,
                                                      const __half* __restrict__ bfg, __half* __restrict__ fg) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char rs_lds[];
  constexpr int MT = RG_MT, NK = RG_NK, PITCH = RG_PITCH, ROWS = RG_ROWS, D = 384, PPR = D / 8, AP = ROWS * PPR / 512;
  static_assert(ROWS * PPR % 512 == 0, "pieces per thread");
  unsigned char* X = rs_lds;
  unsigned char* R = rs_lds + ROWS * PITCH;
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, mi = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.x * ROWS;
  const unsigned bvoff = (unsigned)lane * 16u;
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<rs_u4*>(w1), 0, (unsigned)(D * D * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<rs_u4*>(w2), 0, (unsigned)(D * D * 2), 0x00020000);
  const unsigned wbase = (unsigned)((wv * NK) * RS_NT * 1024);
  const __amdgpu_buffer_rsrc_t rsfg = __builtin_amdgcn_make_buffer_rsrc(const_cast<rs_u4*>(wfg), 0, FG ? (unsigned)(2 * D * D * 2) : 0u, 0x00020000);
  auto wbfg = [&](int nb) { return (unsigned)(((nb * RS_NW + wv) * NK) * RS_NT * 1024); };
  // ---- the (gathered) rows -> X
  rs_u4 ap[AP];
#pragma unroll
  for (int i = 0; i < AP; i++) {
    const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
    int64_t src = row0 + row < M ? ((MLP && gather) ? gather[row0 + row] : (int64_t)(row0 + row)) : -1;
    if (src >= x_rows) src = -1;
    ap[i] = src >= 0 ? *reinterpret_cast<const rs_u4*>(x + src * ldx + 8 * c) : rs_u4{0u, 0u, 0u, 0u};
  }
  rs_u4 b[RS_RB][RS_NT];
  if (MLP) rs_prefetch<NK>(rs1, wbase, bvoff, b);
  else if (FG) rs_prefetch<NK>(rsfg, wbfg(0), bvoff, b);
  if (!MLP) {                                                          // x + hy[grp], rounded as the expand-add kernel stores it, and back to y
    rs_u4 hp[AP];
#pragma unroll
    for (int i = 0; i < AP; i++) {
      const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
      hp[i] = row0 + row < M ? *reinterpret_cast<const rs_u4*>(hy + (int64_t)grp[row0 + row] * D + 8 * c) : rs_u4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < AP; i++) {
      const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
      const rs_h8 a8 = __builtin_bit_cast(rs_h8, ap[i]), h8 = __builtin_bit_cast(rs_h8, hp[i]);
      rs_h8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = (_Float16)((float)a8[e] + (float)h8[e]);
      ap[i] = __builtin_bit_cast(rs_u4, o);
      if (row0 + row < M) *reinterpret_cast<rs_u4*>(y + (int64_t)(row0 + row) * D + 8 * c) = ap[i];
    }
  }
#pragma unroll
  for (int i = 0; i < AP; i++) {
    const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
    *reinterpret_cast<rs_u4*>(X + row * PITCH + 16 * c) = ap[i];
  }
  __syncthreads();
  const unsigned char* arowX = X + mi * PITCH + 16 * kg;
  const unsigned char* arowR = R + mi * PITCH + 16 * kg;
  const int colw = 48 * wv + 4 * kg;
  unsigned char* eX = X + mi * PITCH + colw * 2;
  unsigned char* eR = R + mi * PITCH + colw * 2;
  rs_f4 acc[MT][RS_NT];
  if (MLP) {
  // ---- R = relu(X W1 + b1)
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int t = 0; t < RS_NT; t++) acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
  rs_kloop<MT, NK, PITCH, true>(arowX, rs1, wbase, bvoff, b, acc, rs2, wbase);
#pragma unroll
  for (int t = 0; t < RS_NT; t++) {
    const rs_f4 bs = rs_cvt4(*reinterpret_cast<const rs_h4*>(reinterpret_cast<const _Float16*>(b1) + colw + 16 * t));
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      rs_f4 v = acc[mt][t] + bs;
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
      *reinterpret_cast<rs_h4*>(eR + 16 * mt * PITCH + 32 * t) = rs_pack4(v);
    }
  }
  __syncthreads();                                                     // R complete, everybody done with X
  if (residual) {                                                      // the residual rows (the workgroup's own) travel while the second product runs
#pragma unroll
    for (int i = 0; i < AP; i++) {
      const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
      ap[i] = row0 + row < M ? *reinterpret_cast<const rs_u4*>(residual + (int64_t)(row0 + row) * D + 8 * c) : rs_u4{0u, 0u, 0u, 0u};
    }
  }
  // ---- y = R W2 + b2 [+ residual]
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int t = 0; t < RS_NT; t++) acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
  if (FG) rs_kloop<MT, NK, PITCH, true>(arowR, rs2, wbase, bvoff, b, acc, rsfg, wbfg(0));
  else rs_kloop<MT, NK, PITCH>(arowR, rs2, wbase, bvoff, b, acc);
  if (residual) {
#pragma unroll
    for (int i = 0; i < AP; i++) {
      const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
      *reinterpret_cast<rs_u4*>(X + row * PITCH + 16 * c) = ap[i];
    }
    __syncthreads();                                                   // the residual rows are in X
  }
#pragma unroll
  for (int t = 0; t < RS_NT; t++) {
    const rs_f4 bs = rs_cvt4(*reinterpret_cast<const rs_h4*>(reinterpret_cast<const _Float16*>(b2) + colw + 16 * t));
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      rs_f4 v = acc[mt][t] + bs;
      if (residual) v += rs_cvt4(*reinterpret_cast<const rs_h4*>(eX + 16 * mt * PITCH + 32 * t));
      *reinterpret_cast<rs_h4*>(eX + 16 * mt * PITCH + 32 * t) = rs_pack4(v);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < AP; i++) {
    const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
    if (row0 + row < M) *reinterpret_cast<rs_u4*>(y + (int64_t)(row0 + row) * D + 8 * c) = *reinterpret_cast<const rs_u4*>(X + row * PITCH + 16 * c);
  }
  }   // MLP
  if (FG) {
    // ---- fg = X [Wf | Wg]^T + [bf | bg]: two products on the rows in X, each through R so that whole row halves leave
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int t = 0; t < RS_NT; t++) acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
      if (nb == 0) rs_kloop<MT, NK, PITCH, true>(arowX, rsfg, wbfg(0), bvoff, b, acc, rsfg, wbfg(1));
      else rs_kloop<MT, NK, PITCH>(arowX, rsfg, wbfg(1), bvoff, b, acc);
      if (nb == 1) __syncthreads();                                    // the first half has left R
#pragma unroll
      for (int t = 0; t < RS_NT; t++) {
        const rs_f4 bs = rs_cvt4(*reinterpret_cast<const rs_h4*>(reinterpret_cast<const _Float16*>(bfg) + nb * D + colw + 16 * t));
#pragma unroll
        for (int mt = 0; mt < MT; mt++) *reinterpret_cast<rs_h4*>(eR + 16 * mt * PITCH + 32 * t) = rs_pack4(acc[mt][t] + bs);
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < AP; i++) {
        const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
        if (row0 + row < M) *reinterpret_cast<rs_u4*>(fg + (int64_t)(row0 + row) * (2 * D) + nb * D + 8 * c) = *reinterpret_cast<const rs_u4*>(R + row * PITCH + 16 * c);
      }
    }
  }
}

// The correlation branch and the first LayerNorm of the update operator, one launch (enet.py:59-66, 82-83):
//   c = l5(relu(LN3(l2(relu(l0(corr))))));  x = LN(net + inp + c).
// l0 has 882 inputs: its rows arrive in three K chunks of 384 (X, R, X again), the accumulators stay; then h, LN3's output and c take
// turns in R / X / R; the last LayerNorm runs row-wise (a quarter wave per row) with net and inp read as whole rows.
struct RsCorr {
  const __half* corr; int64_t ldc;                                     // [E, 882], rows 4-byte aligned
  const rs_u4* w0; const __half* b0; const rs_u4* w2; const __half* b2; const __half* ln3_g; const __half* ln3_b;
  const rs_u4* w5; const __half* b5; const __half* net; const __half* inp; const __half* ln_g; const __half* ln_b; __half* out;
  int E, K0; float eps3, eps;
  const float* net32;                                                  // NET32 instantiation: the recurrent state as the caller holds it (autocast keeps it in fp32)
};
constexpr int RC_VEC = 5 * 384;                                        // halves: b0 | b2 | ln3 gamma | ln3 beta | b5
constexpr int RC_LDS = 2 * RG_ROWS * RG_PITCH + RC_VEC * 2 + RG_ROWS * 8 * 2 * 4 + RG_ROWS * 2 * 4;

template <bool NET32>
__global__ __launch_bounds__(512) void k_rs_corr_f16(RsCorr a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char rs_lds[];
  constexpr int MT = RG_MT, NK = RG_NK, PITCH = RG_PITCH, ROWS = RG_ROWS, D = 384, PPR = D / 8, AP = ROWS * PPR / 512;
  constexpr int NK0 = 28, AP2 = ROWS * 16 / 512;                       // l0: 28 K steps (882 inputs); the third chunk: 128 inputs = 16 pieces per row
  constexpr unsigned OFF_NONE = 0x80000000u;
  unsigned char* X = rs_lds;
  unsigned char* R = rs_lds + ROWS * PITCH;
  _Float16* vec = reinterpret_cast<_Float16*>(rs_lds + 2 * ROWS * PITCH);
  float* part = reinterpret_cast<float*>(rs_lds + 2 * ROWS * PITCH + RC_VEC * 2);
  float* stat = part + ROWS * 16;
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, mi = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.x * ROWS, E = a.E, K0 = a.K0;
  const unsigned bvoff = (unsigned)lane * 16u;
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(a.corr), 0, (unsigned)(((int64_t)(E - 1) * a.ldc + K0) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<rs_u4*>(a.w0), 0, (unsigned)(D * NK0 * 32 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<rs_u4*>(a.w2), 0, (unsigned)(D * D * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs5 = __builtin_amdgcn_make_buffer_rsrc(const_cast<rs_u4*>(a.w5), 0, (unsigned)(D * D * 2), 0x00020000);
  const unsigned wb0 = (unsigned)((wv * NK0) * RS_NT * 1024), wb = (unsigned)((wv * NK) * RS_NT * 1024);
  // ---- chunks 0 and 1 of the rows -> X, R
  rs_u4 ap[2][AP];
#pragma unroll
  for (int ch = 0; ch < 2; ch++)
#pragma unroll
    for (int i = 0; i < AP; i++) {
      const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
      ap[ch][i] = __builtin_amdgcn_raw_buffer_load_b128(rsc, row0 + row < E ? (unsigned)(((int64_t)(row0 + row) * a.ldc + 384 * ch + 8 * c) * 2) : OFF_NONE, 0, 0);
    }
  rs_u4 b[RS_RB][RS_NT];
  rs_prefetch<NK>(rs0, wb0, bvoff, b);
#pragma unroll
  for (int k = 0; k < (RC_VEC + 511) / 512; k++) {
    const int i = tid + 512 * k;
    if (i < RC_VEC) {
      const __half* src = i < 384 ? a.b0 + i : i < 768 ? a.b2 + (i - 384) : i < 1152 ? a.ln3_g + (i - 768) : i < 1536 ? a.ln3_b + (i - 1152) : a.b5 + (i - 1536);
      vec[i] = (_Float16)__half2float(*src);
    }
  }
#pragma unroll
  for (int ch = 0; ch < 2; ch++)
#pragma unroll
    for (int i = 0; i < AP; i++) {
      const int p = tid + 512 * i, row = p / PPR, c = p - row * PPR;
      *reinterpret_cast<rs_u4*>((ch ? R : X) + row * PITCH + 16 * c) = ap[ch][i];
    }
  // the third chunk (inputs 768 .. 895; what lies behind input K0 is not part of the row) travels while the first two multiply
  rs_u4 a2[AP2];
#pragma unroll
  for (int i = 0; i < AP2; i++) {
    const int p = tid + 512 * i, row = p >> 4, c = p & 15, k = 768 + 8 * c;
    a2[i] = __builtin_amdgcn_raw_buffer_load_b128(rsc, (row0 + row < E && k < K0) ? (unsigned)(((int64_t)(row0 + row) * a.ldc + k) * 2) : OFF_NONE, 0, 0);
  }
  __syncthreads();
  const unsigned char* arowX = X + mi * PITCH + 16 * kg;
  const unsigned char* arowR = R + mi * PITCH + 16 * kg;
  const int colw = 48 * wv + 4 * kg;
  unsigned char* eX = X + mi * PITCH + colw * 2;
  unsigned char* eR = R + mi * PITCH + colw * 2;
  auto vec4 = [&](int off, int t) { return rs_cvt4(*reinterpret_cast<const rs_h4*>(vec + off + colw + 16 * t)); };
  rs_f4 acc[MT][RS_NT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int t = 0; t < RS_NT; t++) acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
  // ---- h = relu(corr W0 + b0): 12 + 12 + 4 K steps
  rs_kloop<MT, NK, PITCH, true>(arowX, rs0, wb0, bvoff, b, acc, rs0, wb0 + 12u * RS_NT * 1024u);
  rs_kloop<MT, NK, PITCH, true>(arowR, rs0, wb0 + 12u * RS_NT * 1024u, bvoff, b, acc, rs0, wb0 + 24u * RS_NT * 1024u);
  __syncthreads();                                                     // everybody is done with X (and with R)
#pragma unroll
  for (int i = 0; i < AP2; i++) {
    const int p = tid + 512 * i, row = p >> 4, c = p & 15, k = 768 + 8 * c;
    if (k < K0 && k + 8 > K0) {
      rs_h8 v = __builtin_bit_cast(rs_h8, a2[i]);
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = k + e < K0 ? v[e] : (_Float16)0.f;
      a2[i] = __builtin_bit_cast(rs_u4, v);
    }
    *reinterpret_cast<rs_u4*>(X + row * PITCH + 16 * c) = a2[i];
  }
  __syncthreads();
  rs_kloop<MT, 4, PITCH, true>(arowX, rs0, wb0 + 24u * RS_NT * 1024u, bvoff, b, acc, rs2, wb);
#pragma unroll
  for (int t = 0; t < RS_NT; t++) {
    const rs_f4 bs = vec4(0, t);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      rs_f4 v = acc[mt][t] + bs;
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
      *reinterpret_cast<rs_h4*>(eR + 16 * mt * PITCH + 32 * t) = rs_pack4(v);
      acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  // ---- X = relu(LN3(h W2 + b2))
  rs_kloop<MT, NK, PITCH, true>(arowR, rs2, wb, bvoff, b, acc, rs5, wb);
#pragma unroll
  for (int t = 0; t < RS_NT; t++) {
    const rs_f4 bs = vec4(384, t);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt][t] = rs_cvt4(rs_pack4(acc[mt][t] + bs));      // (rounded where the layer-by-layer path stores it)
  }
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int t = 0; t < RS_NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) { s += acc[mt][t][r]; q += acc[mt][t][r] * acc[mt][t][r]; }
    s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
    s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
    if (kg == 0) { part[((16 * mt + mi) * 8 + wv) * 2] = s; part[((16 * mt + mi) * 8 + wv) * 2 + 1] = q; }
  }
  __syncthreads();
  if (tid < ROWS) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; w++) { s += part[(tid * 8 + w) * 2]; q += part[(tid * 8 + w) * 2 + 1]; }
    const float mean = s / (float)D, var = fmaxf(q / (float)D - mean * mean, 0.f);
    stat[2 * tid] = mean; stat[2 * tid + 1] = rsqrtf(var + a.eps3);
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < RS_NT; t++) {
    const rs_f4 g = vec4(768, t), bb = vec4(1152, t);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const float mean = stat[2 * (16 * mt + mi)], rstd = stat[2 * (16 * mt + mi) + 1];
      rs_f4 v = (acc[mt][t] - mean) * rstd * g + bb;
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
      *reinterpret_cast<rs_h4*>(eX + 16 * mt * PITCH + 32 * t) = rs_pack4(v);
      acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  // ---- R = c = X W5 + b5
  rs_kloop<MT, NK, PITCH>(arowX, rs5, wb, bvoff, b, acc);
#pragma unroll
  for (int t = 0; t < RS_NT; t++) {
    const rs_f4 bs = vec4(1536, t);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) *reinterpret_cast<rs_h4*>(eR + 16 * mt * PITCH + 32 * t) = rs_pack4(acc[mt][t] + bs);
  }
  __syncthreads();
  // ---- out = LN(net + inp + c): a quarter wave per row
  {
    const int l16 = tid & 15;
    rs_u4 nv[3][3], nw[NET32 ? 3 : 1][3], iv[3][3];
#pragma unroll
    for (int p = 0; p < 3; p++) {
      const int row = row0 + 32 * p + (tid >> 4), rr = row < E ? row : E - 1;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if constexpr (NET32) {
          const float* np_ = a.net32 + (int64_t)rr * D + (l16 + 16 * k) * 8;
          nv[p][k] = *reinterpret_cast<const rs_u4*>(np_);
          nw[p][k] = *reinterpret_cast<const rs_u4*>(np_ + 4);
        } else {
          nv[p][k] = *reinterpret_cast<const rs_u4*>(a.net + (int64_t)rr * D + (l16 + 16 * k) * 8);
        }
        iv[p][k] = *reinterpret_cast<const rs_u4*>(a.inp + (int64_t)rr * D + (l16 + 16 * k) * 8);
      }
    }
    rs_u4 gm[3], bt[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      gm[k] = *reinterpret_cast<const rs_u4*>(a.ln_g + (l16 + 16 * k) * 8);
      bt[k] = *reinterpret_cast<const rs_u4*>(a.ln_b + (l16 + 16 * k) * 8);
    }
#pragma unroll
    for (int p = 0; p < 3; p++) {
      const int rloc = 32 * p + (tid >> 4), row = row0 + rloc;
      float t[3][8];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const rs_h8 ih = __builtin_bit_cast(rs_h8, iv[p][k]),
                    ch = __builtin_bit_cast(rs_h8, *reinterpret_cast<const rs_u4*>(R + rloc * PITCH + (l16 + 16 * k) * 16));
        float xf[8];
        if constexpr (NET32) {
          const rs_f4 lo4 = __builtin_bit_cast(rs_f4, nv[p][k]), hi4 = __builtin_bit_cast(rs_f4, nw[p][k]);
#pragma unroll
          for (int i = 0; i < 4; i++) { xf[i] = lo4[i]; xf[4 + i] = hi4[i]; }
        } else {
          const rs_h8 xh = __builtin_bit_cast(rs_h8, nv[p][k]);
#pragma unroll
          for (int i = 0; i < 8; i++) xf[i] = (float)xh[i];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) { t[k][i] = xf[i] + (float)ih[i] + (float)ch[i]; s += t[k][i]; }
      }
      const float mean = rs_row16_sum(s) / (float)D;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) { const float d = t[k][i] - mean; q += d * d; }
      const float rstd = rsqrtf(rs_row16_sum(q) / (float)D + a.eps);
      if (row < E) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const rs_h8 g8 = __builtin_bit_cast(rs_h8, gm[k]), b8 = __builtin_bit_cast(rs_h8, bt[k]);
          rs_h8 o;
#pragma unroll
          for (int i = 0; i < 8; i++) o[i] = (_Float16)((t[k][i] - mean) * rstd * (float)g8[i] + (float)b8[i]);
          *reinterpret_cast<rs_u4*>(a.out + (int64_t)row * D + (l16 + 16 * k) * 8) = __builtin_bit_cast(rs_u4, o);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- fp32 storage
// The same structure for fp32 rows on the fp16 matrix cores with exact hi + lo splits (linear.hip's arithmetic: x y = hi lo' + lo hi' +
// hi hi', fp32 accumulation, power-of-two scales per row and per weight column): the rows are scaled by their OWN largest magnitude (the
// whole row is there when it is loaded: no running scale), split ONCE into two fp16 tiles in LDS (hi | lo, 75 KB each), every wave
// streams its hi and lo weight fragments into registers one K step ahead; three products per block.  The result goes through LDS (over
// the row tiles) so that whole rows leave, with the ReLU adjoint mask and the residual read as whole rows too.  One workgroup = 96 rows x
// 384 columns; a 768-wide layer = two column blocks (blockIdx.y) that split the rows again.
__device__ __forceinline__ int rs_scale_exp(float m) {                 // biased exponent of the power of two that brings m into [2^8, 2^9)
  int e = (int)((__float_as_uint(m) >> 23) & 255u);
  e = e < 16 ? 16 : e;
  return 127 + 8 + 127 - e;
}
__device__ __forceinline__ float rs_pow2(int biased) { return __uint_as_float((unsigned)(biased < 0 ? 0 : (biased > 254 ? 254 : biased)) << 23); }
__device__ __forceinline__ void rs_split8(const float (&x)[8], rs_u4& hi, rs_u4& lo) {      // hi = rn(x), lo = rn(x - hi)
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[j]) : "v"(x[2 * j]), "v"(x[2 * j + 1]));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l[j]) : "v"(h[j]), "v"(x[2 * j]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l[j]) : "v"(h[j]), "v"(x[2 * j + 1]));
  }
  hi = rs_u4{h[0], h[1], h[2], h[3]};
  lo = rs_u4{l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ float rs_row16_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false)));
  return v;
}

// W fp32 (element (n, k) at W[n * s_n + k * s_k]; N a multiple of 384) -> [N / 384][8 waves][ceil(K / 32)][3 tiles][hi | lo][64 lanes][16 B], every
// column scaled by its power of two; then N floats: the inverse column scales.  Pass 1: the scales.
__global__ __launch_bounds__(256) void k_rs_wscale(const float* __restrict__ W, int64_t s_n, int64_t s_k, int N, int K, float* __restrict__ inv) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float m = 0.f;
  for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(W[(int64_t)n * s_n + (int64_t)k * s_k]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) inv[n] = rs_pow2(254 - rs_scale_exp(m));
}
__global__ __launch_bounds__(256) void k_rs_wsplit(const float* __restrict__ W, int64_t s_n, int64_t s_k, int N, int K, rs_u4* __restrict__ out) {
  const int nk = (K + 31) / 32;
  const long long total = (long long)(N / RS_BN) * RS_NW * nk * RS_NT * 64;
  const float* inv = reinterpret_cast<const float*>(out + (size_t)total * 2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int t = (int)(r % RS_NT); r /= RS_NT;
    const int s = (int)(r % nk); r /= nk;
    const int w = (int)(r % RS_NW), nb = (int)(r / RS_NW);
    const int n = nb * RS_BN + 48 * w + 16 * t + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    const float sc = rs_pow2(254 - (int)(__float_as_uint(inv[n]) >> 23));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = k0 + j < K ? W[(int64_t)n * s_n + (int64_t)(k0 + j) * s_k] * sc : 0.f;
    rs_u4 hi, lo;
    rs_split8(v, hi, lo);
    rs_u4* dst = out + ((i >> 6) * 2) * 64 + lane;
    dst[0] = hi;
    dst[64] = lo;
  }
}

constexpr int RF_YLD = 388;                                            // row pitch (floats) of the result tile
constexpr int RF_LDS = 2 * RG_ROWS * RG_PITCH + RG_ROWS * 4 + 2 * 384 * 4;
static_assert(RG_ROWS * RF_YLD * 4 <= 2 * RG_ROWS * RG_PITCH, "the result tile lies over the row tiles");

__global__ __launch_bounds__(512) void k_rs_linear_split(const float* __restrict__ x, int64_t ldx, const rs_u4* __restrict__ wimg, const float* __restrict__ bias,
                                                          const float* residual, const float* __restrict__ gate, float* y, int64_t ldy, int M, int N, int K,
                                                          int relu_from) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char rs_lds[];
  constexpr int MT = RG_MT, NK = RG_NK, PITCH = RG_PITCH, ROWS = RG_ROWS, D = 384;
  unsigned char* XH = rs_lds;
  unsigned char* XL = rs_lds + ROWS * PITCH;
  float* rinv = reinterpret_cast<float*>(rs_lds + 2 * ROWS * PITCH);   // [row]: the inverse row scales
  float* cvec = rinv + ROWS;                                           // [inverse column scale | bias][384]
  float* Y = reinterpret_cast<float*>(rs_lds);
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, mi = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.x * ROWS, nb = blockIdx.y, NB = gridDim.y;
  const unsigned bvoff = (unsigned)lane * 16u;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<rs_u4*>(wimg), 0, (unsigned)((int64_t)NB * RS_NW * NK * RS_NT * 2048), 0x00020000);
  const unsigned wbase = (unsigned)(((nb * RS_NW + wv) * NK) * RS_NT * 2048);
  // this wave's weights: K step s, tile t = hi (1 KB) | lo (1 KB) at wbase + (s * 3 + t) * 2 KB; two steps in registers
  rs_u4 bh[2][RS_NT], bl[2][RS_NT];
  auto load_b = [&](int s) {
#pragma unroll
    for (int t = 0; t < RS_NT; t++) {
      bh[s & 1][t] = __builtin_amdgcn_raw_buffer_load_b128(rsw, bvoff, wbase + (unsigned)((s * RS_NT + t) * 2048), 0);
      bl[s & 1][t] = __builtin_amdgcn_raw_buffer_load_b128(rsw, bvoff, wbase + (unsigned)((s * RS_NT + t) * 2048 + 1024), 0);
    }
  };
  if (tid < D) {
    const float* inv = reinterpret_cast<const float*>(wimg + (size_t)NB * RS_NW * NK * RS_NT * 128);
    cvec[tid] = inv[nb * D + tid];
    cvec[D + tid] = bias ? bias[nb * D + tid] : 0.f;
  }
  // ---- the rows: a quarter wave per row (16 lanes x 3 pieces of 8 floats), scaled by the row's own power of two, split, -> XH | XL
  {
    const int l16 = tid & 15;
    rs_f4 xv[3][3][2];
#pragma unroll
    for (int p = 0; p < 3; p++) {
      const int row = row0 + 32 * p + (tid >> 4);
      const float* xr = x + (int64_t)(row < M ? row : M - 1) * ldx;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int c = (l16 + 16 * k) * 8;
#pragma unroll
        for (int h = 0; h < 2; h++) xv[p][k][h] = (c + 4 * h < K) ? *reinterpret_cast<const rs_f4*>(xr + c + 4 * h) : rs_f4{0.f, 0.f, 0.f, 0.f};
      }
    }
    load_b(0);
#pragma unroll
    for (int p = 0; p < 3; p++) {
      const int rloc = 32 * p + (tid >> 4);
      const bool live = row0 + rloc < M;
      float m = 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
          for (int e = 0; e < 4; e++) m = fmaxf(m, fabsf(xv[p][k][h][e]));
      m = rs_row16_max(m);
      const int ex = rs_scale_exp(m);
      const float sc = live ? rs_pow2(ex) : 0.f;
      if (l16 == 0) rinv[rloc] = rs_pow2(254 - ex);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; e++) { v[e] = xv[p][k][0][e] * sc; v[4 + e] = xv[p][k][1][e] * sc; }
        rs_u4 hi, lo;
        rs_split8(v, hi, lo);
        *reinterpret_cast<rs_u4*>(XH + rloc * PITCH + (l16 + 16 * k) * 16) = hi;
        *reinterpret_cast<rs_u4*>(XL + rloc * PITCH + (l16 + 16 * k) * 16) = lo;
      }
    }
  }
  __syncthreads();
  const unsigned char* arH = XH + mi * PITCH + 16 * kg;
  const unsigned char* arL = XL + mi * PITCH + 16 * kg;
  rs_f4 acc[MT][RS_NT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int t = 0; t < RS_NT; t++) acc[mt][t] = rs_f4{0.f, 0.f, 0.f, 0.f};
  // ---- half steps: three row tiles at a time (their hi and lo fragments one half step ahead), the weights one K step ahead
  constexpr int HM = MT / 2;
  rs_u4 ah[2][HM], al[2][HM];
#pragma unroll
  for (int u = 0; u < HM; u++) {
    ah[0][u] = *reinterpret_cast<const rs_u4*>(arH + 16 * u * PITCH);
    al[0][u] = *reinterpret_cast<const rs_u4*>(arL + 16 * u * PITCH);
  }
#pragma unroll
  for (int h = 0; h < 2 * NK; h++) {
    const int s = h >> 1, hf = h & 1;
    if (h + 1 < 2 * NK) {
      const int s1 = (h + 1) >> 1, hf1 = (h + 1) & 1;
#pragma unroll
      for (int u = 0; u < HM; u++) {
        ah[(h + 1) & 1][u] = *reinterpret_cast<const rs_u4*>(arH + 16 * (HM * hf1 + u) * PITCH + 64 * s1);
        al[(h + 1) & 1][u] = *reinterpret_cast<const rs_u4*>(arL + 16 * (HM * hf1 + u) * PITCH + 64 * s1);
      }
    }
    if (hf == 0 && s + 1 < NK) load_b(s + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < HM; u++)                                       // small terms first; the same accumulator again 9 products later
#pragma unroll
      for (int t = 0; t < RS_NT; t++)
        acc[HM * hf + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rs_h8, bl[s & 1][t]), __builtin_bit_cast(rs_h8, ah[h & 1][u]), acc[HM * hf + u][t], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < HM; u++)
#pragma unroll
      for (int t = 0; t < RS_NT; t++)
        acc[HM * hf + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rs_h8, bh[s & 1][t]), __builtin_bit_cast(rs_h8, al[h & 1][u]), acc[HM * hf + u][t], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < HM; u++)
#pragma unroll
      for (int t = 0; t < RS_NT; t++)
        acc[HM * hf + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rs_h8, bh[s & 1][t]), __builtin_bit_cast(rs_h8, ah[h & 1][u]), acc[HM * hf + u][t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                                                     // everybody is done with the row tiles: the result tile takes their place
  const int colw = 48 * wv + 4 * kg;
#pragma unroll
  for (int t = 0; t < RS_NT; t++) {
    const rs_f4 ci = *reinterpret_cast<const rs_f4*>(cvec + colw + 16 * t), bs = *reinterpret_cast<const rs_f4*>(cvec + D + colw + 16 * t);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      rs_f4 v = acc[mt][t] * rinv[16 * mt + mi] * ci + bs;
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = nb * D + colw + 16 * t + r >= relu_from ? fmaxf(v[r], 0.f) : v[r];
      *reinterpret_cast<rs_f4*>(Y + (16 * mt + mi) * RF_YLD + colw + 16 * t) = v;
    }
  }
  __syncthreads();
  constexpr int YP = ROWS * (D / 4) / 512;                             // 16-byte pieces per thread: whole rows leave
#pragma unroll
  for (int i = 0; i < YP; i++) {
    const int p = tid + 512 * i, row = p / (D / 4), c = p - row * (D / 4);
    if (row0 + row < M) {
      rs_f4 v = *reinterpret_cast<const rs_f4*>(Y + row * RF_YLD + 4 * c);
      const int64_t o = (int64_t)(row0 + row) * ldy + nb * D + 4 * c;
      if (gate) {                                                      // a ReLU's output: this gradient passes where it did not clip
        const rs_f4 q = *reinterpret_cast<const rs_f4*>(gate + o);
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = q[r] > 0.f ? v[r] : 0.f;
      }
      if (residual) v += *reinterpret_cast<const rs_f4*>(residual + o);     // (may be y itself: read, then written, by this thread)
      *reinterpret_cast<rs_f4*>(y + o) = v;
    }
  }
}

template <int MT, int NK>
static int rs_launch(const void* x, int64_t ldx, const void* wimg, const void* bias, const void* residual, void* y, int64_t ldy, int M, int NB, int K,
                     int relu_from, hipStream_t stream, unsigned long long* trace = nullptr) {
  constexpr int ROWS = 16 * MT, PITCH = 2 * 32 * NK + 16;
  // the row tile; the last pass's result tiles lie over it (26.6 KB: more than a 32-row tile), earlier passes' behind it
  const int lds = std::max(ROWS * PITCH, RS_NW * 16 * RS_EPI_LD * 4) + (NB > 1 ? RS_NW * 16 * RS_EPI_LD * 4 : 0);
  static PerDeviceOnce attr_done;
  if (attr_done.first()) {
    DEVO_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_linear_f16<MT, NK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                 "devo_upd_rs_linear_f16: cannot raise the dynamic LDS limit");
  }
  hipLaunchKernelGGL((k_rs_linear_f16<MT, NK>), dim3((unsigned)((M + ROWS - 1) / ROWS)), dim3(512), lds, stream, (const __half*)x, ldx, (const rs_u4*)wimg,
                     (const __half*)bias, (const __half*)residual, (__half*)y, ldy, M, NB, K, relu_from, trace);
  return check_launch("devo_upd_rs_linear_f16");
}

template <bool MLP, bool FG>
static int rs_chain_launch(const void* x, int64_t ldx, int x_rows, const int64_t* gather, const void* w1img, const void* b1, const void* w2img, const void* b2,
                           const void* residual, void* y, int M, const void* hy, const int* grp, const void* wfg, const void* bfg, void* fg, hipStream_t stream) {
  static PerDeviceOnce attr_done;
  if (attr_done.first()) {
    DEVO_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_mlp2_f16<MLP, FG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                 "devo_upd_rs_mlp2_f16: cannot raise the dynamic LDS limit");
  }
  hipLaunchKernelGGL((k_rs_mlp2_f16<MLP, FG>), dim3((unsigned)((M + RG_ROWS - 1) / RG_ROWS)), dim3(512), 2 * RG_ROWS * RG_PITCH, stream, (const __half*)x, ldx, x_rows, gather,
                     (const rs_u4*)w1img, (const __half*)b1, (const rs_u4*)w2img, (const __half*)b2, (const __half*)residual, (__half*)y, M, (const __half*)hy, grp,
                     (const rs_u4*)wfg, (const __half*)bfg, (__half*)fg);
  return check_launch("devo_upd_rs_mlp2_f16");
}

}  // namespace devo

using namespace devo;

extern "C" {

// bytes of the weight image of devo_upd_rs_pack_weight_f16 for a [N, K] fp16 weight (N a multiple of 384)
size_t devo_upd_rs_weight_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || N % RS_BN) return 0;
  return (size_t)N * ((K + 31) / 32 * 32) * 2;
}

int devo_upd_rs_pack_weight_f16(const void* W, int64_t s_n, int64_t s_k, int N, int K, void* img, void* stream) {
  DEVO_REQUIRE(W && img && N > 0 && K > 0 && N % RS_BN == 0, "devo_upd_rs_pack_weight_f16: N (%d) must be a multiple of 384", N);
  const long long total = (long long)N * ((K + 31) / 32) * 4;
  hipLaunchKernelGGL(k_rs_pack_f16, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const __half*)W, s_n, s_k, N, K,
                     (rs_u4*)img);
  return check_launch("devo_upd_rs_pack_weight_f16");
}

// does devo_upd_rs_linear_f16 take this shape?
int devo_upd_rs_supported(int N, int K) { return N > 0 && N % RS_BN == 0 && K > 352 && K <= 384; }

// y[M, N] = act(x W^T + bias) [+ residual] (fp16 storage, fp32 accumulation; the ReLU from column relu_from on; residual may be y):
// the Update operator's Linear layers (enet.py:41-78).  x rows 4-byte aligned, y / residual rows 16-byte aligned.
int devo_upd_rs_linear_f16(const void* x, int64_t ldx, const void* wimg, const void* bias, const void* residual, void* y, int64_t ldy, int M, int N, int K,
                           int relu_from, void* stream) {
  DEVO_REQUIRE(x && wimg && y && M > 0, "devo_upd_rs_linear_f16: null argument");
  DEVO_REQUIRE(devo_upd_rs_supported(N, K), "devo_upd_rs_linear_f16: N (%d) must be a multiple of 384 and K (%d) in (352, 384]", N, K);
  DEVO_REQUIRE(!(ldy & 7) && !(reinterpret_cast<uintptr_t>(y) & 15) && !(reinterpret_cast<uintptr_t>(residual) & 15) && !(ldx & 1) && !(reinterpret_cast<uintptr_t>(x) & 3),
               "devo_upd_rs_linear_f16: alignment (y / residual rows 16 bytes, x rows 4 bytes)");
  DEVO_REQUIRE(((int64_t)(M - 1) * ldx + K) * 2 < (1ll << 31), "devo_upd_rs_linear_f16: x beyond 2 GB");
  static const int mt = [] { const char* e = getenv("DEVO_RS_MT"); return e ? atoi(e) : 6; }();
  const int NB = N / RS_BN;
  if (relu_from < 0) relu_from = 0;
  static const bool tr = getenv("DEVO_RS_TRACE") != nullptr;          // debug: stamps of one launch (synchronous), summary on stderr
  if (tr) {
    const int rows = mt == 6 ? 96 : mt == 4 ? 64 : mt == 3 ? 48 : mt == 2 ? 32 : 128, nwg = (M + rows - 1) / rows;
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, (size_t)nwg * 48) != hipSuccess) return DEVO_ERR_LAUNCH;
    int rc = mt == 6 ? rs_launch<6, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream, d)
           : mt == 4 ? rs_launch<4, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream, d)
           : mt == 3 ? rs_launch<3, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream, d)
           : mt == 2 ? rs_launch<2, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream, d)
                     : rs_launch<8, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream, d);
    (void)hipStreamSynchronize((hipStream_t)stream);
    std::vector<unsigned long long> h((size_t)nwg * 6);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    double ph[4] = {0, 0, 0, 0};
    unsigned long long lo = ~0ull, hi = 0;
    for (int g = 0; g < nwg; g++) {
      for (int i = 0; i < 4; i++) ph[i] += (double)(h[6 * g + i + 1] - h[6 * g + i]);
      lo = std::min(lo, h[6 * g]); hi = std::max(hi, h[6 * g + 4]);
    }
    fprintf(stderr, "rs trace: %d workgroups x %d rows, NB %d: mean cycles rows->LDS %.0f | barrier %.0f | K loops (all but the last pass's epilogue) %.0f | last epilogue %.0f | first start to last end %llu\n",
            nwg, rows, NB, ph[0] / nwg, ph[1] / nwg, ph[2] / nwg, ph[3] / nwg, hi - lo);
    return rc;
  }
  // few rows (the SoftAggs' h layers: 1 440 / 210 rows at cfg2, 2 112 at the steady-state graph): 32-row workgroups — the products of a 96-row tile
  // on three CUs are time added to the latency of the weight stream, which is what such a launch consists of (9.4 -> ~6 us).  Swept in round 6
  // (tools/exp_r06y.sh): 32 rows win up to 12 288 rows (2 112: 9.1 -> 5.3 us, 8 192: 9.9 -> 7.4, 12 288: 10.6 -> 10.0), 96 rows at 21 600 (12.8 / 15.0)
  static const int small_m = [] { const char* e = getenv("DEVO_RS_SMALL_M"); return e ? atoi(e) : 12288; }();   // (tuning switch)
  if (M <= small_m && !getenv("DEVO_RS_MT")) return rs_launch<2, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream);
  if (mt == 6) return rs_launch<6, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream);
  if (mt == 2) return rs_launch<2, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream);
  if (mt == 3) return rs_launch<3, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream);
  if (mt == 4) return rs_launch<4, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream);
  return rs_launch<8, 12>(x, ldx, wimg, bias, residual, y, ldy, M, NB, K, relu_from, (hipStream_t)stream);
}


// What follows the frame-pair aggregation in the update operator, one launch, fp16 storage (enet.py:52-57, 96-99; blocks.py:29-48):
//   net = LN0(x + hy[group_of]);  net = LN2(net + sigmoid(gate1(net)) res1(net));  net_out = net + sigmoid(gate3(net)) res3(net);
//   delta = d(relu(net_out)), weight = sigmoid(w(relu(net_out))).   x, net_out [E, 384] contiguous, hy [groups, 384]; the [gate | res[0]]
//   weights concatenated to [768, 384] and res[2] [384, 384] as devo_upd_rs_pack_weight_f16 images; every vector fp16, 16-byte aligned.
static int rs_gru_impl(const void* x, const void* hy, const int* group_of, const void* ln0_w, const void* ln0_b, float eps0, const void* wgr1_img,
                       const void* bgr1, const void* wr2_1_img, const void* br2_1, const void* ln2_w, const void* ln2_b, float eps2, const void* wgr3_img,
                       const void* bgr3, const void* wr2_3_img, const void* br2_3, const void* Wd, const void* bd, const void* Ww, const void* bw,
                       void* net_out, void* delta, void* weight, int E, void* stream, bool out32) {
  DEVO_REQUIRE(x && hy && group_of && ln0_w && ln0_b && wgr1_img && bgr1 && wr2_1_img && br2_1 && ln2_w && ln2_b && wgr3_img && bgr3 && wr2_3_img && br2_3 && Wd &&
                   bd && Ww && bw && net_out && delta && weight && E > 0,
               "devo_upd_rs_gru_f16: null argument");
  DEVO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hy) | reinterpret_cast<uintptr_t>(ln0_w) | reinterpret_cast<uintptr_t>(ln0_b) |
                 reinterpret_cast<uintptr_t>(Wd) | reinterpret_cast<uintptr_t>(Ww) | reinterpret_cast<uintptr_t>(net_out) | reinterpret_cast<uintptr_t>(wgr1_img) |
                 reinterpret_cast<uintptr_t>(wr2_1_img) | reinterpret_cast<uintptr_t>(wgr3_img) | reinterpret_cast<uintptr_t>(wr2_3_img)) & 15) == 0,
               "devo_upd_rs_gru_f16: 16-byte alignment");
  static PerDeviceOnce attr_done;
  if (attr_done.first()) {
    DEVO_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_gru_f16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_gru_f16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                 "devo_upd_rs_gru_f16: cannot raise the dynamic LDS limit");
  }
  RsGru a;
  a.x = (const __half*)x; a.hy = (const __half*)hy; a.grp = group_of; a.ln0_g = (const __half*)ln0_w; a.ln0_b = (const __half*)ln0_b;
  a.w_gr[0] = (const rs_u4*)wgr1_img; a.b_gr[0] = (const __half*)bgr1; a.w_r2[0] = (const rs_u4*)wr2_1_img; a.b_r2[0] = (const __half*)br2_1;
  a.w_gr[1] = (const rs_u4*)wgr3_img; a.b_gr[1] = (const __half*)bgr3; a.w_r2[1] = (const rs_u4*)wr2_3_img; a.b_r2[1] = (const __half*)br2_3;
  a.ln2_g = (const __half*)ln2_w; a.ln2_b = (const __half*)ln2_b; a.Wd = (const __half*)Wd; a.bd = (const __half*)bd; a.Ww = (const __half*)Ww; a.bw = (const __half*)bw;
  a.net_out = out32 ? nullptr : (__half*)net_out; a.net_out32 = out32 ? (float*)net_out : nullptr;
  a.delta = (__half*)delta; a.weight = (__half*)weight; a.E = E; a.eps0 = eps0; a.eps2 = eps2; a.trace = nullptr;
  const int nwg = (E + RG_ROWS - 1) / RG_ROWS;
  static const bool tr = getenv("DEVO_RS_TRACE") != nullptr;          // debug: stamps of this launch (synchronous), mean cycles per phase on stderr
  if (tr && hipMalloc(&a.trace, (size_t)nwg * 24 * 8) != hipSuccess) a.trace = nullptr;
  if (out32) hipLaunchKernelGGL(k_rs_gru_f16<true>, dim3((unsigned)nwg), dim3(512), RG_LDS, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_rs_gru_f16<false>, dim3((unsigned)nwg), dim3(512), RG_LDS, (hipStream_t)stream, a);
  if (a.trace) {
    (void)hipStreamSynchronize((hipStream_t)stream);
    std::vector<unsigned long long> h((size_t)nwg * 24);
    (void)hipMemcpy(h.data(), a.trace, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(a.trace);
    fprintf(stderr, "rs gru trace (%d workgroups), mean cycles per phase:", nwg);
    for (int i = 0; i + 1 < 24; i++) {
      double d = 0; int n = 0;
      for (int g = 0; g < nwg; g++) if (h[24 * g + i + 1]) { d += (double)(h[24 * g + i + 1] - h[24 * g + i]); n++; }
      if (n) fprintf(stderr, " %.0f", d / n);
    }
    double tot = 0;
    for (int g = 0; g < nwg; g++) { int l = 23; while (l > 0 && !h[24 * g + l]) l--; tot += (double)(h[24 * g + l] - h[24 * g]); }
    fprintf(stderr, " | total %.0f\n", tot / nwg);
  }
  return check_launch("devo_upd_rs_gru_f16");
}
int devo_upd_rs_gru_f16(const void* x, const void* hy, const int* group_of, const void* ln0_w, const void* ln0_b, float eps0, const void* wgr1_img,
                        const void* bgr1, const void* wr2_1_img, const void* br2_1, const void* ln2_w, const void* ln2_b, float eps2, const void* wgr3_img,
                        const void* bgr3, const void* wr2_3_img, const void* br2_3, const void* Wd, const void* bd, const void* Ww, const void* bw,
                        void* net_out, void* delta, void* weight, int E, void* stream) {
  return rs_gru_impl(x, hy, group_of, ln0_w, ln0_b, eps0, wgr1_img, bgr1, wr2_1_img, br2_1, ln2_w, ln2_b, eps2, wgr3_img, bgr3, wr2_3_img, br2_3, Wd, bd, Ww, bw,
                     net_out, delta, weight, E, stream, false);
}
// ... with net_out as fp32 [E, 384] (the values the fp16 form stores, widened: what devo.py:311's autocast call returns — no conversion pass behind the launch)
int devo_upd_rs_gru_f16_out32(const void* x, const void* hy, const int* group_of, const void* ln0_w, const void* ln0_b, float eps0, const void* wgr1_img,
                              const void* bgr1, const void* wr2_1_img, const void* br2_1, const void* ln2_w, const void* ln2_b, float eps2, const void* wgr3_img,
                              const void* bgr3, const void* wr2_3_img, const void* br2_3, const void* Wd, const void* bd, const void* Ww, const void* bw,
                              float* net_out, void* delta, void* weight, int E, void* stream) {
  return rs_gru_impl(x, hy, group_of, ln0_w, ln0_b, eps0, wgr1_img, bgr1, wr2_1_img, br2_1, ln2_w, ln2_b, eps2, wgr3_img, bgr3, wr2_3_img, br2_3, Wd, bd, Ww, bw,
                     net_out, delta, weight, E, stream, true);
}


// l2(relu(l1(x[gather]))) [+ residual] as one launch, both layers 384 -> 384 (enet.py:46-50, 86-91); x rows 16-byte aligned (ldx a multiple
// of 8), residual / y [M, 384] contiguous; gather i64 [M] (negative or >= x_rows: a zero row) or null.  Images: devo_upd_rs_pack_weight_f16.
// wfg / bfg / fg (all or none): the [768, 384] f | g layer of the SoftAgg that follows, on the result rows: fg [M, 768].
int devo_upd_rs_mlp2_fg_f16(const void* x, int64_t ldx, int x_rows, const int64_t* gather, const void* w1img, const void* b1, const void* w2img, const void* b2,
                            const void* residual, void* y, int M, const void* wfg, const void* bfg, void* fg, void* stream) {
  DEVO_REQUIRE(x && w1img && b1 && w2img && b2 && y && M > 0 && x_rows > 0, "devo_upd_rs_mlp2_f16: null argument");
  DEVO_REQUIRE(ldx >= 384 && ldx % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
                                               reinterpret_cast<uintptr_t>(w1img) | reinterpret_cast<uintptr_t>(w2img) | reinterpret_cast<uintptr_t>(wfg) |
                                               reinterpret_cast<uintptr_t>(fg)) & 15) == 0 &&
                   ((reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(bfg)) & 7) == 0,
               "devo_upd_rs_mlp2_f16: alignment");
  DEVO_REQUIRE((wfg != nullptr) == (bfg != nullptr) && (wfg != nullptr) == (fg != nullptr), "devo_upd_rs_mlp2_fg_f16: the f | g layer needs its image, its bias and its output");
  DEVO_REQUIRE(y != x || !gather, "devo_upd_rs_mlp2_f16: gathered rows cannot be updated in place");
  if (wfg) return rs_chain_launch<true, true>(x, ldx, x_rows, gather, w1img, b1, w2img, b2, residual, y, M, nullptr, nullptr, wfg, bfg, fg, (hipStream_t)stream);
  return rs_chain_launch<true, false>(x, ldx, x_rows, gather, w1img, b1, w2img, b2, residual, y, M, nullptr, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
}
int devo_upd_rs_mlp2_f16(const void* x, int64_t ldx, int x_rows, const int64_t* gather, const void* w1img, const void* b1, const void* w2img, const void* b2,
                         const void* residual, void* y, int M, void* stream) {
  return devo_upd_rs_mlp2_fg_f16(x, ldx, x_rows, gather, w1img, b1, w2img, b2, residual, y, M, nullptr, nullptr, nullptr, stream);
}
// x += hy[group_of] (the expand-add behind a SoftAgg, enet.py:93; in place) and fg = x [Wf | Wg]^T + b of the SoftAgg that follows, one launch:
// x [M, 384] contiguous, hy [groups, 384], fg [M, 768]; 16-byte alignment; the image of devo_upd_rs_pack_weight_f16 for the [768, 384] weight.
int devo_upd_rs_expand_fg_f16(void* x, const void* hy, const int* group_of, const void* wfg, const void* bfg, void* fg, int M, void* stream) {
  DEVO_REQUIRE(x && hy && group_of && wfg && bfg && fg && M > 0, "devo_upd_rs_expand_fg_f16: null argument");
  DEVO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hy) | reinterpret_cast<uintptr_t>(wfg) | reinterpret_cast<uintptr_t>(fg)) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(bfg) & 7) == 0, "devo_upd_rs_expand_fg_f16: alignment");
  return rs_chain_launch<false, true>(x, 384, M, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, x, M, hy, group_of, wfg, bfg, fg, (hipStream_t)stream);
}

// The correlation branch and the first LayerNorm of the update operator as one launch, fp16 storage (enet.py:59-66, 82-83):
//   c = l5(relu(LN3(l2(relu(l0(corr))))));  out = LN(net + inp + c).   corr [E, K0] with 768 < K0 <= 896 (DEVO: 882), rows 4-byte aligned; net / inp /
//   out [E, 384] contiguous; weight images of devo_upd_rs_pack_weight_f16 ([384, K0], [384, 384], [384, 384]); vectors fp16, 16-byte aligned.
static int rs_corr_impl(const void* corr, int64_t ldc, int K0, const void* w0img, const void* b0, const void* w2img, const void* b2, const void* ln3_w,
                        const void* ln3_b, float eps3, const void* w5img, const void* b5, const void* net, const void* inp, const void* ln_w, const void* ln_b,
                        float eps, void* out, int E, void* stream, bool net32) {
  DEVO_REQUIRE(corr && w0img && b0 && w2img && b2 && ln3_w && ln3_b && w5img && b5 && net && inp && ln_w && ln_b && out && E > 0, "devo_upd_rs_corr_f16: null argument");
  DEVO_REQUIRE(K0 > 768 && K0 <= 896 && ldc >= K0 && ldc % 2 == 0 && (reinterpret_cast<uintptr_t>(corr) & 3) == 0, "devo_upd_rs_corr_f16: 768 < K0 (%d) <= 896, rows 4-byte aligned", K0);
  DEVO_REQUIRE(((reinterpret_cast<uintptr_t>(net) | reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(ln_w) |
                 reinterpret_cast<uintptr_t>(ln_b) | reinterpret_cast<uintptr_t>(w0img) | reinterpret_cast<uintptr_t>(w2img) | reinterpret_cast<uintptr_t>(w5img)) & 15) == 0,
               "devo_upd_rs_corr_f16: 16-byte alignment");
  DEVO_REQUIRE(((int64_t)(E - 1) * ldc + K0) * 2 < (1ll << 31), "devo_upd_rs_corr_f16: corr beyond 2 GB");
  static PerDeviceOnce attr_done;
  if (attr_done.first()) {
    DEVO_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_corr_f16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_corr_f16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                 "devo_upd_rs_corr_f16: cannot raise the dynamic LDS limit");
  }
  RsCorr a;
  a.corr = (const __half*)corr; a.ldc = ldc; a.w0 = (const rs_u4*)w0img; a.b0 = (const __half*)b0; a.w2 = (const rs_u4*)w2img; a.b2 = (const __half*)b2;
  a.ln3_g = (const __half*)ln3_w; a.ln3_b = (const __half*)ln3_b; a.w5 = (const rs_u4*)w5img; a.b5 = (const __half*)b5;
  a.net = net32 ? nullptr : (const __half*)net; a.net32 = net32 ? (const float*)net : nullptr;
  a.inp = (const __half*)inp; a.ln_g = (const __half*)ln_w; a.ln_b = (const __half*)ln_b; a.out = (__half*)out; a.E = E; a.K0 = K0; a.eps3 = eps3; a.eps = eps;
  if (net32) hipLaunchKernelGGL(k_rs_corr_f16<true>, dim3((unsigned)((E + RG_ROWS - 1) / RG_ROWS)), dim3(512), RC_LDS, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_rs_corr_f16<false>, dim3((unsigned)((E + RG_ROWS - 1) / RG_ROWS)), dim3(512), RC_LDS, (hipStream_t)stream, a);
  return check_launch("devo_upd_rs_corr_f16");
}
int devo_upd_rs_corr_f16(const void* corr, int64_t ldc, int K0, const void* w0img, const void* b0, const void* w2img, const void* b2, const void* ln3_w,
                         const void* ln3_b, float eps3, const void* w5img, const void* b5, const void* net, const void* inp, const void* ln_w, const void* ln_b,
                         float eps, void* out, int E, void* stream) {
  return rs_corr_impl(corr, ldc, K0, w0img, b0, w2img, b2, ln3_w, ln3_b, eps3, w5img, b5, net, inp, ln_w, ln_b, eps, out, E, stream, false);
}
// ... with `net` as fp32 [E, 384] (devo.py:311 under autocast keeps the recurrent state in fp32: it enters the sum net + inp + c unrounded, no conversion pass in front)
int devo_upd_rs_corr_f16_net32(const void* corr, int64_t ldc, int K0, const void* w0img, const void* b0, const void* w2img, const void* b2, const void* ln3_w,
                               const void* ln3_b, float eps3, const void* w5img, const void* b5, const float* net, const void* inp, const void* ln_w,
                               const void* ln_b, float eps, void* out, int E, void* stream) {
  return rs_corr_impl(corr, ldc, K0, w0img, b0, w2img, b2, ln3_w, ln3_b, eps3, w5img, b5, net, inp, ln_w, ln_b, eps, out, E, stream, true);
}


// fp32 storage on the fp16 matrix cores with exact hi + lo splits, the row-resident structure: weight image of a [N, K] fp32 weight (element
// (n, k) at W[n * s_n + k * s_k]: the forward's weight or its transpose view for dX = dY W), N a multiple of 384.
size_t devo_upd_rs_split_weight_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || N % RS_BN) return 0;
  return (size_t)N * ((K + 31) / 32 * 32) * 4 + (size_t)N * 4;
}
int devo_upd_rs_split_weight(const float* W, int64_t s_n, int64_t s_k, int N, int K, void* img, void* stream) {
  DEVO_REQUIRE(W && img && N > 0 && K > 0 && N % RS_BN == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0, "devo_upd_rs_split_weight: N (%d) must be a multiple of 384", N);
  float* inv = reinterpret_cast<float*>(static_cast<unsigned char*>(img) + (size_t)N * ((K + 31) / 32 * 32) * 4);
  hipLaunchKernelGGL(k_rs_wscale, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, W, s_n, s_k, N, K, inv);
  const long long total = (long long)N * ((K + 31) / 32) * 4;
  hipLaunchKernelGGL(k_rs_wsplit, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, W, s_n, s_k, N, K, (rs_u4*)img);
  return check_launch("devo_upd_rs_split_weight");
}
int devo_upd_rs_split_supported(int N, int K) { return N > 0 && N % RS_BN == 0 && K > 352 && K <= 384 && K % 4 == 0; }
// y[M, N] = act(x W^T + bias) [gated] [+ residual], fp32 in and out (devo_upd_linear_split's contract; x, y, residual, gate rows 16-byte aligned)
int devo_upd_rs_linear_split(const float* x, int64_t ldx, const void* wimg, const float* bias, const float* residual, const float* gate, float* y, int64_t ldy,
                             int M, int N, int K, int relu_from, void* stream) {
  DEVO_REQUIRE(x && wimg && y && M > 0, "devo_upd_rs_linear_split: null argument");
  DEVO_REQUIRE(devo_upd_rs_split_supported(N, K), "devo_upd_rs_linear_split: N (%d) must be a multiple of 384 and K (%d) a multiple of 4 in (352, 384]", N, K);
  DEVO_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
                                                  reinterpret_cast<uintptr_t>(gate) | reinterpret_cast<uintptr_t>(wimg)) & 15) == 0,
               "devo_upd_rs_linear_split: 16-byte alignment of the rows");
  static PerDeviceOnce attr_done;
  if (attr_done.first()) {
    DEVO_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_linear_split), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                 "devo_upd_rs_linear_split: cannot raise the dynamic LDS limit");
  }
  hipLaunchKernelGGL(k_rs_linear_split, dim3((unsigned)((M + RG_ROWS - 1) / RG_ROWS), (unsigned)(N / RS_BN)), dim3(512), RF_LDS, (hipStream_t)stream, x, ldx,
                     (const rs_u4*)wimg, bias, residual, gate, y, ldy, M, N, K, relu_from < 0 ? 0 : relu_from);
  return check_launch("devo_upd_rs_linear_split");
}

}  // extern "C"
