// Weight gradients of the Update operator's Linear layers, dW = dY^T X (and the bias gradient, the column sums of dY), in fp32 STORAGE on the
// fp16 matrix cores — the third product of a Linear layer's training step next to csrc/linear.hip's y and dX (devo/enet.py:41-78,
// blocks.py:15-48 through torch.autograd).
//
// dW[i][j] = sum over the 18 000 edge rows r of dY[r][i] X[r][j]: a 384 x 384 result from two tall operands.  The library runs it on a
// handful of tiles (148 us; 75 us as a batched product over 16 row chunks + a sum + the bias gradient's own reduction).  Here the rows are
// the K dimension of v_mfma_f32_16x16x32_f16: both operands are needed with 8 consecutive ROWS of one column per lane, the transpose of how
// they lie in memory.  Per step of 32 rows a workgroup brings a [32][128] tile of dY and one of X into LDS by lane-linear DMA (32 lanes per
// row: full lines), and every lane reads its 8 values of a column with four ds_read2_b32 (16 bytes of padding behind every pair of rows put
// the four row groups of a tile into different banks).  Values are scaled by per-COLUMN running powers of two (as linear.hip scales rows: a
// gradient column of 1e-9 keeps its 22 bits), split exactly into fp16 hi + lo, and multiplied as hi lo' + lo hi' + hi hi' with fp32 accumulation.
// Workgroup = a 128 x 128 block of dW (4 waves x 64 x 64) over a slice of the rows; the slices' partial blocks go to a workspace and a
// second kernel adds them up (no atomics: the result is reproducible).
#include "common.h"
#include <hip/hip_fp16.h>
#include <algorithm>

namespace devo {

typedef _Float16 dw_h8 __attribute__((ext_vector_type(8)));
typedef float dw_f4 __attribute__((ext_vector_type(4)));
typedef unsigned dw_u4 __attribute__((ext_vector_type(4)));

constexpr int DW_BT = 128;                        // block of dW per workgroup: 128 columns of dY x 128 columns of X
constexpr int DW_ROWS = 32;                       // rows per step (the MFMA's K)
constexpr int DW_INSTR = 1024 + 16;               // LDS bytes of one DMA instruction: two rows of 512 B, then padding
constexpr int DW_TILE = 16 * DW_INSTR;            // one operand tile of a stage
constexpr int DW_STAGE = 2 * DW_TILE;
constexpr int DW_LDS = 2 * DW_STAGE;              // two stages: 66 560 B, two workgroups per CU
constexpr int DW_EXP_TARGET = 8;
constexpr float DW_RAISE = 16384.f;

__device__ __forceinline__ void dw_split8(const float (&x)[8], dw_h8& hi, dw_h8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[j]) : "v"(x[2 * j]), "v"(x[2 * j + 1]));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l[j]) : "v"(h[j]), "v"(x[2 * j]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l[j]) : "v"(h[j]), "v"(x[2 * j + 1]));
  }
  const dw_u4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
  hi = __builtin_bit_cast(dw_h8, hv);
  lo = __builtin_bit_cast(dw_h8, lv);
}
__device__ __forceinline__ int dw_scale_exp(float m) {
  int e = (int)((__float_as_uint(m) >> 23) & 255u);
  e = e < 16 ? 16 : e;
  return 127 + DW_EXP_TARGET + 127 - e;
}
__device__ __forceinline__ float dw_pow2(int biased) { return __uint_as_float((unsigned)(biased < 0 ? 0 : (biased > 254 ? 254 : biased)) << 23); }
__device__ __forceinline__ void dw_dma16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned lds_addr) {
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(lds_addr) : "memory");
}

// grid = (blocks of dW) x splits.  part [splits][No][Ni], gpart [splits][No] (the bias gradient's partial sums, written by the workgroups of
// the first block column).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_dw_split(
    const float* __restrict__ G, int64_t ldg, const float* __restrict__ X, int64_t ldx, int R, int No, int Ni, int steps_per_split,
    float* __restrict__ part, float* __restrict__ gpart) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char dw_lds[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, kg = lane >> 4;
  const int wi = wv >> 1, wj = wv & 1;
  constexpr unsigned OFF_NONE = 0x80000000u;
  const int nbj = (Ni + DW_BT - 1) / DW_BT, Nip = nbj * DW_BT, blk = blockIdx.x, bi = blk / nbj, bj = blk - bi * nbj, sp = blockIdx.y;
  const int i0 = bi * DW_BT, j0 = bj * DW_BT;
  const int T = (R + DW_ROWS - 1) / DW_ROWS;
  const int s_begin = sp * steps_per_split, s_end = min(T, s_begin + steps_per_split);
  const unsigned lds0 = (unsigned)(uintptr_t)dw_lds;
  // DMA: instruction t of a tile = rows 2 t, 2 t + 1 (32 lanes of 16 bytes each); this wave issues t = wv, wv + 4, wv + 8, wv + 12 of both operands
  unsigned gvoff[4], xvoff[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = 2 * (wv + 4 * q) + (lane >> 5);
    // (a piece that starts behind the last column fetches nothing; one that straddles it brings values of the next row, zeroed at the operand read)
    gvoff[q] = i0 + (lane & 31) * 4 < No ? (unsigned)(((int64_t)row * ldg + i0) * 4 + (lane & 31) * 16) : OFF_NONE;
    xvoff[q] = j0 + (lane & 31) * 4 < Ni ? (unsigned)(((int64_t)row * ldx + j0) * 4 + (lane & 31) * 16) : OFF_NONE;
  }
  // ONE descriptor per operand for the whole matrix (round 6: building two per step was 780 of a step's 4 800 cycles, profiles/r06_dw_split.txt):
  // rows past the end of the matrix are out of range (zeros, no access); operands stay below 2 GB (checked by the host)
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G), 0, (unsigned)(((int64_t)(R - 1) * ldg + No) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (unsigned)(((int64_t)(R - 1) * ldx + Ni) * 4), 0x00020000);
  const unsigned gstep = (unsigned)((int64_t)DW_ROWS * ldg * 4), xstep = (unsigned)((int64_t)DW_ROWS * ldx * 4);
  auto request = [&](int s, int buf) {                                 // rows 32 s .. of both operands -> stage buf
    const bool any = s < s_end && s * DW_ROWS < R;
    const unsigned go = (unsigned)s * gstep, xo = (unsigned)s * xstep;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      dw_dma16(any && gvoff[q] != OFF_NONE ? gvoff[q] + go : OFF_NONE, rg, lds0 + (unsigned)(buf * DW_STAGE + (wv + 4 * q) * DW_INSTR));
      dw_dma16(any && xvoff[q] != OFF_NONE ? xvoff[q] + xo : OFF_NONE, rx, lds0 + (unsigned)(buf * DW_STAGE + DW_TILE + (wv + 4 * q) * DW_INSTR));
    }
  };
  // operand reads: lane (column li of tile t, row group kg) -> rows 8 kg + q, q = 0 .. 7: four ds_read2_b32 (rows 2 u, 2 u + 1)
  const unsigned a_base = (unsigned)(4 * kg * DW_INSTR + (64 * wi + li) * 4);
  const unsigned b_base = (unsigned)(DW_TILE + 4 * kg * DW_INSTR + (64 * wj + li) * 4);
  auto read8 = [&](int buf, unsigned base, int t, float (&v)[8]) {
    const unsigned char* p = dw_lds + buf * DW_STAGE + base + t * 64;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      v[2 * u] = *reinterpret_cast<const float*>(p + u * DW_INSTR);
      v[2 * u + 1] = *reinterpret_cast<const float*>(p + u * DW_INSTR + 512);
    }
  };

  dw_f4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = dw_f4{0.f, 0.f, 0.f, 0.f};
  float scA[4], scB[4], gsum[4];
  int eA[4], eB[4];
#pragma unroll
  for (int t = 0; t < 4; t++) { scA[t] = 0.f; scB[t] = 0.f; eA[t] = 0; eB[t] = 0; gsum[t] = 0.f; }
  const bool do_bias = bj == 0 && wj == 0;                            // wave-uniform
  const bool edge_a = i0 + DW_BT > No, edge_b = j0 + DW_BT > Ni;

#ifdef DW_TRACE                                                        // debug build: cycle stamps of workgroup (0, 0), wave 0 -> the first floats of its partial block
  unsigned long long tst[40]; int nst = 0;
#define DW_STAMP() do { if (nst < 40) tst[nst] = __builtin_readcyclecounter(); nst++; } while (0)
#else
#define DW_STAMP() do {} while (0)
#endif
  DW_STAMP();
  request(s_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  DW_STAMP();
  for (int s = s_begin; s < s_end; s++) {
    const int buf = (s - s_begin) & 1;
    request(s + 1, buf ^ 1);
    DW_STAMP();
    float xa[4][8], xb[4][8];
    bool raise = false;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      read8(buf, a_base, t, xa[t]);
      read8(buf, b_base, t, xb[t]);
      if (edge_a && i0 + 64 * wi + 16 * t + li >= No) {                  // (wave-uniform guards: only the last block of a matrix whose width is not a multiple of 128)
#pragma unroll
        for (int j = 0; j < 8; j++) xa[t][j] = 0.f;
      }
      if (edge_b && j0 + 64 * wj + 16 * t + li >= Ni) {
#pragma unroll
        for (int j = 0; j < 8; j++) xb[t][j] = 0.f;
      }
    }
    float ma[4], mb[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      float m = fmaxf(fmaxf(fabsf(xa[t][0]), fabsf(xa[t][1])), fabsf(xa[t][2]));
      m = fmaxf(fmaxf(m, fabsf(xa[t][3])), fabsf(xa[t][4]));
      m = fmaxf(fmaxf(m, fabsf(xa[t][5])), fmaxf(fabsf(xa[t][6]), fabsf(xa[t][7])));
      ma[t] = m;
      float n = fmaxf(fmaxf(fabsf(xb[t][0]), fabsf(xb[t][1])), fabsf(xb[t][2]));
      n = fmaxf(fmaxf(n, fabsf(xb[t][3])), fabsf(xb[t][4]));
      n = fmaxf(fmaxf(n, fabsf(xb[t][5])), fmaxf(fabsf(xb[t][6]), fabsf(xb[t][7])));
      mb[t] = n;
      raise = raise || !(m * scA[t] <= DW_RAISE) || !(n * scB[t] <= DW_RAISE);
      if (do_bias) gsum[t] += ((xa[t][0] + xa[t][1]) + (xa[t][2] + xa[t][3])) + ((xa[t][4] + xa[t][5]) + (xa[t][6] + xa[t][7]));
    }
    DW_STAMP();
    const bool first = s == s_begin;
    if (first || __builtin_amdgcn_ballot_w64(raise) != 0ull) {        // rare after the first step: new column scales, the accumulators follow
      float fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        float m = ma[t];
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const bool need = first || !(m * scA[t] <= DW_RAISE);
        const int en = need ? dw_scale_exp(m) : eA[t];
        fa[t] = first ? 1.f : dw_pow2(127 + en - eA[t]);
        eA[t] = en; scA[t] = dw_pow2(en);
        float n = mb[t];
        n = fmaxf(n, __shfl_xor(n, 16));
        n = fmaxf(n, __shfl_xor(n, 32));
        const bool needb = first || !(n * scB[t] <= DW_RAISE);
        const int enb = needb ? dw_scale_exp(n) : eB[t];
        fb[t] = first ? 1.f : dw_pow2(127 + enb - eB[t]);
        eB[t] = enb; scB[t] = dw_pow2(enb);
      }
      if (!first) {
#pragma unroll
        for (int a = 0; a < 4; a++) {
          dw_f4 fr;                                                     // D rows 4 kg + r of tile a = columns 4 kg + r of the dY tile
#pragma unroll
          for (int r = 0; r < 4; r++) fr[r] = __shfl(fa[a], 4 * kg + r);
#pragma unroll
          for (int b = 0; b < 4; b++) acc[a][b] = acc[a][b] * fr * fb[b];
        }
      }
    }
    DW_STAMP();
    dw_h8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int j = 0; j < 8; j++) { xa[t][j] *= scA[t]; xb[t][j] *= scB[t]; }
      dw_split8(xa[t], ah[t], al[t]);
      dw_split8(xb[t], bh[t], bl[t]);
    }
    DW_STAMP();
#pragma unroll
    for (int a = 0; a < 4; a++)                                        // small terms first; the same accumulator again 16 MFMAs later
#pragma unroll
      for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
    DW_STAMP();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the next step's tiles have landed ...
    __syncthreads();                                                   // ... and every wave is done with this step's
    DW_STAMP();
  }
  // ---- partial block: D[4 kg + r][li] of tile (a, b) = dW[i0 + 64 wi + 16 a + 4 kg + r][j0 + 64 wj + 16 b + li], scaled back
  const int Nop = (No + DW_BT - 1) / DW_BT * DW_BT;
  float* pb = part + ((size_t)sp * Nop + i0 + 64 * wi) * Nip + j0 + 64 * wj;
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const float ia = dw_pow2(254 - eA[a]);
    dw_f4 ir;
#pragma unroll
    for (int r = 0; r < 4; r++) ir[r] = __shfl(ia, 4 * kg + r);
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const float ib = dw_pow2(254 - eB[b]);
#pragma unroll
      for (int r = 0; r < 4; r++) pb[(size_t)(16 * a + 4 * kg + r) * Nip + 16 * b + li] = acc[a][b][r] * ir[r] * ib;
    }
  }
#ifdef DW_TRACE
  DW_STAMP();
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < 40; i++) part[i] = i < nst ? (float)(long long)(tst[i] - tst[0]) : -1.f;
  }
#endif
  if (do_bias) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
      float g = gsum[t];
      g += __shfl_xor(g, 16);
      g += __shfl_xor(g, 32);
      if (kg == 0) gpart[(size_t)sp * Nop + i0 + 64 * wi + 16 * t + li] = g;
    }
  }
}

// dW[i][j] = sum over the splits of part[s][i][j] (partials padded to whole blocks: [Nop][Nip]); db[i] likewise
__global__ __launch_bounds__(256) void k_dw_reduce(const float* __restrict__ part, const float* __restrict__ gpart, int splits, int No, int Ni, int Nop, int Nip,
                                                   float* __restrict__ dW, int64_t ld_dw, float* __restrict__ db) {
  const int q4 = Nip / 4, n4 = No * q4;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    const int row = i / q4, col = 4 * (i - row * q4);
    if (col >= Ni) return;
    dw_f4 t = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; s++) t += *reinterpret_cast<const dw_f4*>(part + ((size_t)s * Nop + row) * Nip + col);
    float* dst = dW + (size_t)row * ld_dw + col;
    if (col + 4 <= Ni && (ld_dw & 3) == 0 && (reinterpret_cast<uintptr_t>(dW) & 15) == 0) *reinterpret_cast<dw_f4*>(dst) = t;
    else
      for (int e = 0; e < 4; e++) if (col + e < Ni) dst[e] = t[e];
  } else if (db && i - n4 < No) {
    float t = 0.f;
    for (int s = 0; s < splits; s++) t += gpart[(size_t)s * Nop + (i - n4)];
    db[i - n4] = t;
  }
}

static int dw_splits(int R, int No, int Ni) {
  static const int env = getenv("DEVO_DW_SPLITS") ? atoi(getenv("DEVO_DW_SPLITS")) : 0;
  const int T = (R + DW_ROWS - 1) / DW_ROWS, blocks = ((No + DW_BT - 1) / DW_BT) * ((Ni + DW_BT - 1) / DW_BT);
  int s = env > 0 ? env : (256 + blocks - 1) / blocks;               // about one workgroup per CU ...
  s = std::min(s, std::max(1, T / 8));                                // ... of at least 8 steps
  return std::max(1, s);
}

}  // namespace devo

using namespace devo;

extern "C" {

size_t devo_upd_dw_workspace_bytes(int R, int No, int Ni) {
  if (R <= 0 || No <= 0 || Ni <= 0) return 0;
  const size_t Nop = (size_t)(No + DW_BT - 1) / DW_BT * DW_BT, Nip = (size_t)(Ni + DW_BT - 1) / DW_BT * DW_BT;
  return (size_t)dw_splits(R, No, Ni) * (Nop * Nip + Nop) * 4;
}

int devo_upd_dw_split(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, int R, int No, int Ni, void* workspace, float* dW,
                      int64_t ld_dw, float* db, devo_stream_t stream) {
  DEVO_REQUIRE(R > 0 && No > 0 && Ni > 0, "devo_upd_dw_split: bad sizes (%d rows, %d x %d)", R, No, Ni);
  DEVO_REQUIRE(dY && X && workspace && dW && ld_dy >= No && ld_x >= Ni && ld_dw >= Ni, "devo_upd_dw_split: null tensor or rows shorter than the matrix");
  DEVO_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dW)) & 3) == 0,
               "devo_upd_dw_split: the workspace must be 16-byte aligned");
  DEVO_REQUIRE((int64_t)R * ld_dy * 4 < (1LL << 31) && (int64_t)R * ld_x * 4 < (1LL << 31), "devo_upd_dw_split: operand beyond 2 GB");
  const int S = dw_splits(R, No, Ni), T = (R + DW_ROWS - 1) / DW_ROWS, per = (T + S - 1) / S;
  const int nbi = (No + DW_BT - 1) / DW_BT, nbj = (Ni + DW_BT - 1) / DW_BT, Nop = nbi * DW_BT, Nip = nbj * DW_BT;
  float* part = static_cast<float*>(workspace);
  float* gpart = part + (size_t)S * Nop * Nip;
  // (per call, like ba.hip: the attribute belongs to the current device's copy of the kernel)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dw_split), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS) != hipSuccess) {
    (void)hipGetLastError();
    set_error("devo_upd_dw_split: cannot reserve %d bytes of LDS", DW_LDS);
    return DEVO_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(k_dw_split, dim3((unsigned)(nbi * nbj), (unsigned)S), dim3(256), DW_LDS, (hipStream_t)stream, dY, ld_dy, X, ld_x, R, No, Ni,
                     per, part, gpart);
  const int n = No * (Nip / 4) + No;
  hipLaunchKernelGGL(k_dw_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, gpart, S, No, Ni, Nop, Nip, dW, ld_dw, db);
  return check_launch("devo_upd_dw_split");
}

}  // extern "C"
