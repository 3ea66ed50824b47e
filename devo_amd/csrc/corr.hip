// altcorr for gfx950: patch <-> frame local correlation lookup (forward + backward) and patchify.
// Replaces devo/altcorr/correlation_kernel.cu (reference module cuda_corr, correlation.cpp:57-63).
//
// Design: the reference launches one thread per (edge, pixel, tap) and walks the 128 channels
// with a stride of H*W elements, then blends/permutes with ~14 ATen kernels.  Here ONE WAVE owns one edge; the 9 patch
// pixels' (2R+2)^2 windows overlap almost entirely, so only their union bounding box (~10x10 px) is fetched, with
// 16-byte loads from a channel-blocked (or channels-last) pyramid, and the bilinear blend + axis swap + output
// permutation are fused into the epilogue (no raw D x D tensor, no temporaries, no stack copy).  Two fast kernels:
//   corr_fwd_mfma_kernel (corr_mfma.h, fp32 / C = 128): position-centric, lanes = box pixels, products on the matrix cores;
//   corr_fwd_cl_kernel (below, fp32 / fp16, any C % 8 == 0): tap-centric, box staged through LDS, DPP-broadcast FMAs.
#include "common.h"
#include "corr_tile.h"
#include "corr_plan.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <type_traits>

namespace devo {

constexpr int PP = 9;          // patch pixels (P = 3)
constexpr int MAXD = 12;       // 2*R+2 for R <= 5
constexpr int KC = CORR_KC;     // channels staged per LDS chunk
constexpr int NT = 128;        // threads per workgroup (2 waves): one chunk of 128 box positions

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f32<double>(double v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ double from_f32<double>(float v) { return (double)v; }

// accumulation type of the generic kernel: fp64 tensors accumulate (and blend) in fp64, like the reference's scalar_t arithmetic
// (AT_DISPATCH_FLOATING_TYPES_AND_HALF, correlation_kernel.cu:203), so that gradcheck-style fp64 lookups are fp64-accurate
template <typename T> struct AccOf { typedef float type; };
template <> struct AccOf<double> { typedef double type; };

// output stores bypass the caches' allocation (written once, read by a later kernel)
// Output stores: plain write-back stores.  Rounds 1-2 streamed them (nontemporal: "keep the L2 for feature rows"); measured in round 3
// (profiles/r03_store_policy.txt): the streaming form writes every partially covered 32-byte sector on its own — 95.9 MB per cfg2 launch for
// 76.2 MB of output — while write-back stores merge an edge's consecutive 216-byte rounds in the L2: 75.2 MB, and the lookup is 3 % faster.
// DEVO_CORR_NT_STORES (A/B builds, tools/build_variant.sh) brings the streaming form back.
#ifdef DEVO_CORR_NT_STORES
__device__ __forceinline__ void store_streamed(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void store_streamed(double* p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void store_streamed(__half* p, __half v) {
  __builtin_nontemporal_store(__half_as_ushort(v), reinterpret_cast<unsigned short*>(p));
}
#else
__device__ __forceinline__ void store_streamed(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_streamed(double* p, double v) { *p = v; }
__device__ __forceinline__ void store_streamed(__half* p, __half v) { *p = v; }
#endif

__device__ __forceinline__ int floor_to_int(float v) { return corr_floor_to_int(v); }

struct EdgeGeom {
  int ox[PP], oy[PP];     // window origin (tap 0,0) of each patch pixel in frame coordinates
  float dx[PP], dy[PP];   // sub-pixel fractions
};

// Bilinear epilogue of correlation_kernel.cu:221-232 on the raw window held in LDS.
// Evaluated without FMA contraction in the reference's order: ((1-dx)(1-dy))*r00 + (dx(1-dy))*r01 + ...
__device__ __forceinline__ float blend4(float dx, float dy, float r00, float r01, float r10, float r11) {
#pragma clang fp contract(off)
  float o = ((1.0f - dx) * (1.0f - dy)) * r00;
  o = o + (dx * (1.0f - dy)) * r01;
  o = o + ((1.0f - dx) * dy) * r10;
  o = o + (dx * dy) * r11;
  return o;
}

template <typename T, typename A = float>
__device__ __forceinline__ void corr_epilogue(const A* sraw, const float* sdx, const float* sdy, T* outp,
                                              int D, int64_t lstride) {
  const int Dm = D - 1;
  const int total = Dm * Dm * PP;
  for (int l = threadIdx.x; l < total; l += blockDim.x) {
    int p = l % PP;              // i0*3 + j0
    int a = (l / PP) % Dm;       // y offset  (logical dim 3)
    int c = l / (PP * Dm);       // x offset  (logical dim 2: permute(0,1,3,2,4,5), correlation_kernel.cu:232)
    const A* r = sraw + p * D * D + a * D + c;
    if constexpr (sizeof(A) == 8) {                       // fp64: same expression order in double
      const double dx = sdx[p], dy = sdy[p];
      double o = ((1.0 - dx) * (1.0 - dy)) * r[0];
      o = o + (dx * (1.0 - dy)) * r[1];
      o = o + ((1.0 - dx) * dy) * r[D];
      o = o + (dx * dy) * r[D + 1];
      outp[(int64_t)l * lstride] = (T)o;
    } else {
      float o = blend4(sdx[p], sdy[p], r[0], r[1], r[D], r[D + 1]);
      outp[(int64_t)l * lstride] = from_f32<T>(o);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Fast path: fmap2 channels-last (channel stride 1), C % KC == 0.
// ONE WAVE PER EDGE, no workgroup barriers, TAP-centric: lane t owns window tap (a, c) = (t / D, t % D) of every
// one of the 9 patch pixels (D = 2r+2 = 8 -> exactly one wave of taps; r = 5 -> 144 taps, NG = 3 per lane) and
// keeps 9 accumulators.  Per KC-channel chunk the wave stages
//   * the union bounding box of the 9 windows (<= TILEPOS positions x KC channels, coalesced 16-byte loads issued
//     one chunk ahead) and
//   * the patch features of the chunk, transposed to [pixel][KC],
// into its private LDS tile.  For pixel p a lane reads ITS tap's position of the box (ds_read_b128 = 4 channels)
// and issues 4 FMAs whose patch operand f1[k][p] is broadcast INSIDE the instruction from lane p of the same
// 16-lane row (DPP row_newbcast:p): no scalar-memory latency in the loop, no broadcast LDS reads, no wasted
// multiply-adds (exactly the 9 * D^2 * C products the lookup needs).
// Boxes that do not fit the tile (patch pixels spread apart) stage the 9 windows one after the other.
// -------------------------------------------------------------------------------------------------
constexpr int WPB = 1;                      // waves (edges) per workgroup (the heavy-first schedule assumes 1)
constexpr int F1ROW = KC + 4;               // row stride of the transposed patch chunk

__device__ __forceinline__ void wave_lds_fence() {
  // a wave's LDS instructions execute in order; this only pins the compiler's ordering
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// 12 FMAs: three pixels (P0, P0+1, P0+2) x four channels.  acc_i += f1[k+u][P0+i] * v_i[u]; w[u] holds
// f1[k+u][lane & 15], read through DPP from lane P0+i of the row.  The three accumulators are interleaved so that
// consecutive instructions are independent.
#define DEVO_FMA_LINE(ACC, W, V, P) "v_fmac_f32_dpp " ACC ", " W ", " V " row_newbcast:" #P " row_mask:0xf bank_mask:0xf\n"
#define DEVO_FMA_TRIPLE(P0, P1, P2)                                                                    \
  asm("s_nop 1\n" /* VALU-written VGPR -> DPP read needs 2 wait states; hipcc does not pad inside asm */ \
      DEVO_FMA_LINE("%0", "%3", "%7", P0) DEVO_FMA_LINE("%1", "%3", "%11", P1) DEVO_FMA_LINE("%2", "%3", "%15", P2) \
      DEVO_FMA_LINE("%0", "%4", "%8", P0) DEVO_FMA_LINE("%1", "%4", "%12", P1) DEVO_FMA_LINE("%2", "%4", "%16", P2) \
      DEVO_FMA_LINE("%0", "%5", "%9", P0) DEVO_FMA_LINE("%1", "%5", "%13", P1) DEVO_FMA_LINE("%2", "%5", "%17", P2) \
      DEVO_FMA_LINE("%0", "%6", "%10", P0) DEVO_FMA_LINE("%1", "%6", "%14", P1) DEVO_FMA_LINE("%2", "%6", "%18", P2) \
      : "+v"(a0), "+v"(a1), "+v"(a2)                                                                    \
      : "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w), "v"(v0.x), "v"(v0.y), "v"(v0.z), "v"(v0.w), "v"(v1.x), "v"(v1.y),    \
        "v"(v1.z), "v"(v1.w), "v"(v2.x), "v"(v2.y), "v"(v2.z), "v"(v2.w))

__device__ __forceinline__ void fma_px012(float& a0, float& a1, float& a2, float4 w, float4 v0, float4 v1, float4 v2) { DEVO_FMA_TRIPLE(0, 1, 2); }
__device__ __forceinline__ void fma_px345(float& a0, float& a1, float& a2, float4 w, float4 v0, float4 v1, float4 v2) { DEVO_FMA_TRIPLE(3, 4, 5); }
__device__ __forceinline__ void fma_px678(float& a0, float& a1, float& a2, float4 w, float4 v0, float4 v1, float4 v2) { DEVO_FMA_TRIPLE(6, 7, 8); }
// 4 FMAs of one pixel (split boxes: the pixels of a stage are selected with wave-uniform branches)
#define DEVO_FMA_ONE(P)                                                                                       \
  asm("s_nop 1\n" DEVO_FMA_LINE("%0", "%1", "%5", P) DEVO_FMA_LINE("%0", "%2", "%6", P)                          \
          DEVO_FMA_LINE("%0", "%3", "%7", P) DEVO_FMA_LINE("%0", "%4", "%8", P)                                  \
      : "+v"(a)                                                                                               \
      : "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w))
template <int P> __device__ __forceinline__ void fma_one(float& a, float4 w, float4 v);
template <> __device__ __forceinline__ void fma_one<0>(float& a, float4 w, float4 v) { DEVO_FMA_ONE(0); }
template <> __device__ __forceinline__ void fma_one<1>(float& a, float4 w, float4 v) { DEVO_FMA_ONE(1); }
template <> __device__ __forceinline__ void fma_one<2>(float& a, float4 w, float4 v) { DEVO_FMA_ONE(2); }
template <> __device__ __forceinline__ void fma_one<3>(float& a, float4 w, float4 v) { DEVO_FMA_ONE(3); }
template <> __device__ __forceinline__ void fma_one<4>(float& a, float4 w, float4 v) { DEVO_FMA_ONE(4); }
template <> __device__ __forceinline__ void fma_one<5>(float& a, float4 w, float4 v) { DEVO_FMA_ONE(5); }
template <> __device__ __forceinline__ void fma_one<6>(float& a, float4 w, float4 v) { DEVO_FMA_ONE(6); }
template <> __device__ __forceinline__ void fma_one<7>(float& a, float4 w, float4 v) { DEVO_FMA_ONE(7); }
template <> __device__ __forceinline__ void fma_one<8>(float& a, float4 w, float4 v) { DEVO_FMA_ONE(8); }


// One pyramid level as the staged kernel sees it.
struct CorrLevel {
  const void* fmap2;
  int H2, W2;
  int64_t s_b, s_n, s_h, s_w;     // element strides of batch, frame, row, column
  int64_t chunk_stride;           // elements between consecutive KC-channel chunks of a pixel (KC, or the block stride)
  int cb_shift;                   // log2 of the channels stored contiguously per pixel (channel block); 30 for channels-last
  int64_t block_stride;           // elements between consecutive channel blocks (unused for channels-last)
  unsigned frame_bytes;           // extent of one frame (all blocks) in bytes: the buffer-load bound of the matrix-core kernel
  bool staged_ok, mfma_ok;        // which of the two fast kernels can read this level
  bool mm_ok;                     // ... and the dense-product kernel (corr_mm.h)
  int64_t out_offset;             // element offset of this level inside an edge's output record
  float coord_div;                // coordinates are divided by this (pyramid level scale)
  bool split;                     // fp32 level in the split-blocked format of devo_corr_pyramid_split (only the dense-product kernel reads it)
  const int* exps;                // ... its scale exponent per (batch, frame)
};

// nlev == 1: workgroup g -> edge slot g of level 0.  nlev == 2 (fused pyramid lookup): the two levels alternate in
// groups of 8 workgroups (one per XCD), so that every CU runs fine-level waves (which wait on HBM/L2) next to
// coarse-level waves (LDS/VALU-bound) and each level keeps its XCD-aware edge order.
template <typename T, int NG, int RMAX>
__global__ __launch_bounds__(WPB * 64) void corr_fwd_cl_kernel(
    const T* __restrict__ fmap1, CorrLevel lv0, CorrLevel lv1, int nlev, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, T* __restrict__ out, int BE, int E, int Np, int n2,
    int C, int64_t out_estride, int64_t out_lstride, int R, const int* __restrict__ order,
    unsigned long long* __restrict__ trace) {
  const int lvl = (nlev == 2) ? ((blockIdx.x >> 3) & 1) : 0;                      // wave-uniform
  const int gid = (nlev == 2) ? (((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) : blockIdx.x;
  const int nitems = (nlev == 2) ? (gridDim.x >> 1) : gridDim.x;                    // workgroups of this level
  const CorrLevel& LV = lvl ? lv1 : lv0;
  const T* __restrict__ fmap2 = static_cast<const T*>(LV.fmap2);
  const int H2 = LV.H2, W2 = LV.W2;
  const int64_t s_b = LV.s_b, s_n = LV.s_n, s_h = LV.s_h, s_w = LV.s_w, chunk_stride = LV.chunk_stride, out_offset = LV.out_offset;
  const float coord_div = LV.coord_div;
  constexpr int TILEPOS = tile_positions(NG);          // box positions staged per chunk
  constexpr int TILE_SLOTS = tile_slots(NG);           // 16-byte slots of the box tile
  constexpr int F2_FLOATS = TILE_SLOTS * 4;            // box tile
  constexpr int F1_FLOATS = PP * F1ROW;                // transposed patch chunk
  constexpr int DMAX = 2 * RMAX + 2;
  constexpr int RW_FLOATS = (PP * (DMAX * DMAX + 1) + 3) / 4 * 4;   // the 9 raw windows [p][a][c], row stride D*D+1
  constexpr int ZERO_OFF = F2_FLOATS + F1_FLOATS;      // KC zeros: what the taps of pixels outside the current stage read
  constexpr int WAVE_FLOATS = F2_FLOATS + F1_FLOATS + KC;   // the raw windows reuse the box tile once all chunks are done
  static_assert(RW_FLOATS <= F2_FLOATS, "raw windows must fit in the box tile");
  static_assert(NG * 64 >= DMAX * DMAX && TILEPOS >= NG * 64, "tap groups must cover the window");
  __shared__ __attribute__((aligned(16))) float s_tile[WPB * WAVE_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Schedule (with a plan): the first `nh` workgroups take the plan's HEAVY edges (multi-pass boxes) so that the
  // longest work items start first on every XCD; the others take the (frame, row-band)-sorted edges XCD-aware:
  // workgroup g runs on XCD g % 8 (observed dispatch order) and every XCD owns one contiguous slice of the sorted list,
  // so that its private L2 sees each feature row about once.  (Bijective for any grid size.)
  const int g = gid * WPB + wave;                            // WPB == 1: one wave per workgroup
  const int nh = order ? min(max(order[BE], 0), BE) : 0;
  int slot;
  if (g < nh) {
    slot = g;
  } else {
    const int nwg = nitems, xcd = gid & 7;
    auto heavy_on = [&](int x) -> int { return nh > x ? (nh - x + 7) >> 3 : 0; };          // heavy workgroups on XCD x
    auto total_on = [&](int x) -> int { return nwg > x ? (nwg - x + 7) >> 3 : 0; };        // all workgroups on XCD x
    int start = nh;
    for (int x = 0; x < xcd; x++) start += total_on(x) - heavy_on(x);
    slot = start + (gid >> 3) - heavy_on(xcd);
  }
  if (slot >= BE) return;                                   // wave-uniform; no barriers in this kernel
  const unsigned long long t_start = trace ? __builtin_readcyclecounter() : 0ULL;
  const int be = order ? order[slot] : slot;
  float* tile = s_tile + wave * WAVE_FLOATS;
  float* f1t = tile + F2_FLOATS;
  float* rawwin = tile;
  if (lane < KC) tile[ZERO_OFF + lane] = 0.0f;
  const int D = 2 * R + 2, ntap = D * D;
  const int b = be / E, e = be - b * E;

  // ---- geometry: lane p (< 9) owns patch pixel p
  float px = 0.0f, py = 0.0f;
  if (lane < PP) {
    px = coords[((int64_t)be * 2 + 0) * PP + lane] / coord_div;
    py = coords[((int64_t)be * 2 + 1) * PP + lane] / coord_div;
  }
  unsigned long long t_geo = 0, t_first = 0, t_loop = 0;
  if (trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_geo = __builtin_readcyclecounter(); }
  const int my_ox = floor_to_int(px) - R, my_oy = floor_to_int(py) - R;
  const float my_dx = px - floorf(px), my_dy = py - floorf(py);
  int ox[PP], oy[PP];
#pragma unroll
  for (int p = 0; p < PP; p++) { ox[p] = __builtin_amdgcn_readlane(my_ox, p); oy[p] = __builtin_amdgcn_readlane(my_oy, p); }
  int xmin = ox[0], xmax = ox[0], ymin = oy[0], ymax = oy[0];
#pragma unroll
  for (int p = 1; p < PP; p++) {
    xmin = min(xmin, ox[p]); xmax = max(xmax, ox[p]);
    ymin = min(ymin, oy[p]); ymax = max(ymax, oy[p]);
  }
  const long long npos_ll = (long long)(xmax - xmin + D) * (ymax - ymin + D);
  auto pitch_of = [](int w) -> int { return tile_pitch(w); };
  auto fits = [&](int w, int h) -> bool { return tile_fits(w, h, NG); };
  const bool whole = (npos_ll <= TILEPOS) && fits(xmax - xmin + D, ymax - ymin + D);   // the union box fits the tile (the usual case)

  const int64_t pi = ii[e];
  const int64_t fj = jj[e];
  const T* __restrict__ f1 = fmap1 + ((int64_t)b * Np + pi) * C * PP;           // [C][9]
  const T* __restrict__ f2 = fmap2 + (int64_t)b * s_b + fj * s_n;
  T* outp = out + (int64_t)be * out_estride + out_offset;

  constexpr int VEC = 16 / sizeof(T);            // elements per 16-byte load
  constexpr int PARTS = KC / VEC;                // 16-byte loads per position
  constexpr int ITERS = (TILEPOS * PARTS + 63) / 64;   // 16-byte loads per lane to fill the tile
  constexpr int F1N = (KC * PP + 63) / 64;       // patch-chunk elements per lane (144 -> 3)
  // patch chunk: element el = lane + 64 j of the [KC][9] block goes to f1t[el % 9][el / 9]
  int f1dst[F1N];
#pragma unroll
  for (int j = 0; j < F1N; j++) { const int el = lane + 64 * j; f1dst[j] = (el < KC * PP) ? (el % PP) * F1ROW + el / PP : -1; }
  const float* wrow = f1t + min(lane & 15, PP - 1) * F1ROW;        // this lane's patch pixel (lanes 9..15 of a row unused)

  // this lane's taps
  int ta[NG], tc[NG];
#pragma unroll
  for (int g = 0; g < NG; g++) { const int t = min(lane + 64 * g, ntap - 1); ta[g] = t / D; tc[g] = t - ta[g] * D; }

  float acc[NG][PP];
#pragma unroll
  for (int g = 0; g < NG; g++)
#pragma unroll
    for (int p = 0; p < PP; p++) acc[g][p] = 0.0f;

  // One "stage" = one box staged chunk by chunk.  Modes: the union box of all 9 pixels (1 stage, the usual case);
  // if that does not fit the tile, the three pixel rows {0,1,2} {3,4,5} {6,7,8} one after the other (3 stages);
  // if even a row's box is too large, the 9 windows one by one (9 stages).
  int txmin[3], txmax[3], tymin[3], tymax[3];
  bool rows_fit = true;
#pragma unroll
  for (int t = 0; t < 3; t++) {
    txmin[t] = min(ox[3 * t], min(ox[3 * t + 1], ox[3 * t + 2])); txmax[t] = max(ox[3 * t], max(ox[3 * t + 1], ox[3 * t + 2]));
    tymin[t] = min(oy[3 * t], min(oy[3 * t + 1], oy[3 * t + 2])); tymax[t] = max(oy[3 * t], max(oy[3 * t + 1], oy[3 * t + 2]));
    rows_fit = rows_fit && ((long long)(txmax[t] - txmin[t] + D) * (tymax[t] - tymin[t] + D) <= TILEPOS) &&
               fits(txmax[t] - txmin[t] + D, tymax[t] - tymin[t] + D);
  }
  const int mode = whole ? 0 : (rows_fit ? 1 : 2);
  const int nstage = (mode == 0) ? 1 : (mode == 1 ? 3 : PP);
  for (int sp = 0; sp < nstage; sp++) {
    int bx0, by0, bw, npos;
    if (mode == 0) { bx0 = xmin; by0 = ymin; bw = xmax - xmin + D; npos = (int)npos_ll; }
    else if (mode == 1) {
      bx0 = sp == 0 ? txmin[0] : (sp == 1 ? txmin[1] : txmin[2]);
      by0 = sp == 0 ? tymin[0] : (sp == 1 ? tymin[1] : tymin[2]);
      const int bx1 = sp == 0 ? txmax[0] : (sp == 1 ? txmax[1] : txmax[2]);
      const int by1 = sp == 0 ? tymax[0] : (sp == 1 ? tymax[1] : tymax[2]);
      bw = bx1 - bx0 + D; npos = bw * (by1 - by0 + D);
    } else { bx0 = __builtin_amdgcn_readlane(my_ox, sp); by0 = __builtin_amdgcn_readlane(my_oy, sp); bw = D; npos = ntap; }
    const int PT = pitch_of(bw);
    // LDS offset (in floats) of this lane's tap for every pixel (only the pixels of this stage are used)
    int rowoff[NG][PP];
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
      for (int p = 0; p < PP; p++)
      {
          const bool in_stage = (mode == 0) || (mode == 1 ? (p / 3 == sp) : (p == sp));     // wave-uniform
          rowoff[g][p] = in_stage ? ((oy[p] - by0 + ta[g]) * PT + SPP * (ox[p] - bx0 + tc[g])) * 4 : ZERO_OFF;
        }

    unsigned sbyte[ITERS];                          // byte offset of each staged 16-byte piece inside the target frame
    int sdst[ITERS];                                // its LDS destination (floats); -1 = none
    bool sok[ITERS];
    {
      // Each 8-lane group (the unit a ds_write_b128 is serviced in) takes ONE 16-byte piece index of 8 consecutive
      // positions: their LDS slots are 3 apart (odd) = 8 different bank quads, so the staging store is conflict-free
      // (piece-major lanes would put pieces 0/1 of positions x and x+8/3 on the same banks: 2-way in every group).
      constexpr int STEP = 64 / PARTS;
      const int part = (lane >> 3) % PARTS;
      const int pos_l = ((lane >> 3) / PARTS) * 8 + (lane & 7);
      const float inv_bw = 1.0f / (float)bw;
      const int sh32 = (int)s_h, sw32 = (int)s_w;   // the launcher guarantees that a frame spans < 2^31 bytes
#pragma unroll
      for (int it = 0; it < ITERS; it++) {
        const int pos = pos_l + it * STEP;
        const int pyy = (int)(((float)pos + 0.5f) * inv_bw);     // exact: pos < 256 <= 2^8, error margin 0.5 / bw
        const int pxx = pos - pyy * bw;
        const int gy = by0 + pyy, gx = bx0 + pxx;
        sdst[it] = (pos < npos) ? (pyy * PT + SPP * pxx) * 4 + part * VEC : -1;
        sok[it] = (pos < npos) && gy >= 0 && gy < H2 && gx >= 0 && gx < W2;
        sbyte[it] = (unsigned)(gy * sh32 + gx * sw32 + part * VEC) * (unsigned)sizeof(T);
      }
    }
    // ---- global -> registers one whole chunk ahead of its use (the loads fly under the FMAs).  (Two chunks ahead
    //      costs 28 more VGPRs = one wave per SIMD less, and measured slower.)
    uint4 raw[ITERS];
    T raw1[F1N];
    auto fetch = [&](int64_t kch, int kc1) {
      const char* fb = reinterpret_cast<const char*>(f2 + kch);            // wave-uniform base + 32-bit lane offset
#pragma unroll
      for (int it = 0; it < ITERS; it++) {
        if (sok[it]) raw[it] = *reinterpret_cast<const uint4*>(fb + sbyte[it]);
      }
#pragma unroll
      for (int j = 0; j < F1N; j++) if (f1dst[j] >= 0) raw1[j] = f1[(int64_t)kc1 * PP + lane + 64 * j];
    };
#pragma unroll
    for (int it = 0; it < ITERS; it++) raw[it] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < F1N; j++) raw1[j] = from_f32<T>(0.0f);
    int64_t koff = 0;
    fetch(koff, 0);
    if (trace && sp == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_first = __builtin_readcyclecounter(); }

    // registers -> LDS (as fp32), then the next chunk's loads are issued so that they fly under the FMAs
    auto stage = [&](int kc) {
      wave_lds_fence();                            // previous chunk's reads are done before the tile is overwritten
#pragma unroll
      for (int it = 0; it < ITERS; it++) {
        if (sdst[it] >= 0) {
          const T* rv = reinterpret_cast<const T*>(&raw[it]);
          float* dst = tile + sdst[it];
#pragma unroll
          for (int u = 0; u < VEC; u += 4)
            *reinterpret_cast<float4*>(dst + u) =
                make_float4(to_f32<T>(rv[u]), to_f32<T>(rv[u + 1]), to_f32<T>(rv[u + 2]), to_f32<T>(rv[u + 3]));
        }
      }
#pragma unroll
      for (int j = 0; j < F1N; j++) if (f1dst[j] >= 0) f1t[f1dst[j]] = to_f32<T>(raw1[j]);
      wave_lds_fence();
      if (kc + KC < C) { koff += chunk_stride; fetch(koff, kc + KC); }
    };
    // ---- 9 accumulators per tap, 4 channels per step.  Branch-free: the pixels that are not part of this stage (split
    //      boxes only) read the zero slot, so their FMAs add 0.  The tap reads of the next pixel row are in flight under
    //      the FMAs of the current one.
    for (int kc = 0; kc < C; kc += KC) {
      stage(kc);
#pragma unroll
      for (int k = 0; k < KC; k += 4) {
        const float4 w = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
        for (int g = 0; g < NG; g++) {
          const float4 v0 = *reinterpret_cast<const float4*>(tile + rowoff[g][0] + k);
          const float4 v1 = *reinterpret_cast<const float4*>(tile + rowoff[g][1] + k);
          const float4 v2 = *reinterpret_cast<const float4*>(tile + rowoff[g][2] + k);
          const float4 v3 = *reinterpret_cast<const float4*>(tile + rowoff[g][3] + k);
          const float4 v4 = *reinterpret_cast<const float4*>(tile + rowoff[g][4] + k);
          const float4 v5 = *reinterpret_cast<const float4*>(tile + rowoff[g][5] + k);
          fma_px012(acc[g][0], acc[g][1], acc[g][2], w, v0, v1, v2);
          __builtin_amdgcn_sched_barrier(0);     // at most two rows of tap reads (24 VGPRs) in flight
          const float4 v6 = *reinterpret_cast<const float4*>(tile + rowoff[g][6] + k);
          const float4 v7 = *reinterpret_cast<const float4*>(tile + rowoff[g][7] + k);
          const float4 v8 = *reinterpret_cast<const float4*>(tile + rowoff[g][8] + k);
          fma_px345(acc[g][3], acc[g][4], acc[g][5], w, v3, v4, v5);
          __builtin_amdgcn_sched_barrier(0);
          fma_px678(acc[g][6], acc[g][7], acc[g][8], w, v6, v7, v8);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // ---- raw windows [p][a][c] (row stride D*D+1: conflict-free epilogue reads); they overwrite the dead box tile
  wave_lds_fence();
  if (trace) t_loop = __builtin_readcyclecounter();
#pragma unroll
  for (int g = 0; g < NG; g++) {
    if (lane + 64 * g < ntap) {
#pragma unroll
      for (int p = 0; p < PP; p++) rawwin[p * (D * D + 1) + ta[g] * D + tc[g]] = acc[g][p];
    }
  }
  wave_lds_fence();
  // ---- fused bilinear blend + axis swap + output permutation (correlation_kernel.cu:221-232)
  const int Dm = D - 1;
  const int total = Dm * Dm * PP;
  {
    // output element l = (cx * Dm + a) * 9 + p: cx = x offset (logical dim 2: permute(0,1,3,2,4,5)), a = y offset, p = i0*3+j0.
    // l advances by 64 = 7 * 9 + 1 per pass: the indices are carried instead of re-divided.
    int q = lane / PP, p = lane - q * PP;
    int cx = q / Dm, a = q - cx * Dm;
    T* op = outp + (int64_t)lane * out_lstride;
    const int64_t ostep = 64 * out_lstride;
    for (int l0 = 0; l0 < total; l0 += 64) {        // wave-uniform trip count: the shuffles below need all lanes
      const float dxp = __shfl(my_dx, p), dyp = __shfl(my_dy, p);
      if (l0 + lane < total) {
        const float* r = rawwin + p * (D * D + 1) + a * D + cx;
        store_streamed(op, from_f32<T>(blend4(dxp, dyp, r[0], r[1], r[D], r[D + 1])));      // keep the L2 for feature rows
      }
      op += ostep;
      p += 1; a += 7;
      if (p >= PP) { p -= PP; a += 1; }
      while (a >= Dm) { a -= Dm; cx += 1; }
    }
  }
  if (trace && lane == 0) {                          // debug: per-wave (start, end, box size, hw id)
    unsigned long long* t = trace + ((size_t)lvl * BE + slot) * 8;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[0] = t_start; t[1] = __builtin_readcyclecounter(); t[2] = (unsigned long long)npos_ll; t[3] = blockIdx.x;
    t[4] = t_geo; t[5] = t_first; t[6] = t_loop;
  }
}

#include "corr_mfma.h"
#include "corr_mm.h"

// -------------------------------------------------------------------------------------------------
// Locality plan: order[] = heavy edge slots, then the rest sorted by (batch, target frame, 16-row band of the
// patch centre).  Binning on all CUs, then one workgroup's LDS counting sort (bins = frames x bands).  The order only affects which edges run
// together (L2 reuse of the feature rows); results are independent of it.
// -------------------------------------------------------------------------------------------------
constexpr int BIN_THREADS = 256;

// Step 1 (all CUs): bin of every edge slot -> bins[be]; -1 marks a HEAVY edge (the box of its 9 windows exceeds
// tile of the staged kernel: it is staged in several passes and runs 2-4x longer) — heavy edges go to the front of the plan.
__global__ __launch_bounds__(BIN_THREADS) void corr_bin_kernel(const float* __restrict__ coords,
                                                               const int64_t* __restrict__ jj, int BE, int E, int n2, int H2,
                                                               float coord_div, int nb, int D, int ng, CorrPlanMode pm,
                                                               int* __restrict__ bins) {
  const int be = blockIdx.x * BIN_THREADS + threadIdx.x;
  if (be >= BE) return;
  const int b = be / E, e = be - b * E;
  const float* c = coords + (int64_t)be * 2 * PP;
  // union box of the 9 windows, computed exactly like the lookup kernel does (floor_to_int of the scaled coordinate)
  int xs[PP], ys[PP];
#pragma unroll
  for (int p = 0; p < PP; p++) { xs[p] = floor_to_int(c[p] / coord_div); ys[p] = floor_to_int(c[PP + p] / coord_div); }
  const int bin = corr_plan_bin(xs, ys, c[4] / coord_div, c[PP + 4] / coord_div, b, (int)jj[e], n2, H2, nb, D, ng, pm.W2, pm.l1,
                                pm.heavy_cells, pm.dead_bin);
  bins[be] = bin;
}

// Step 2: counting sort of the bins by gridDim.x independent workgroups, the heavy list first (corr_plan.h).
template <int CACHE>
__global__ __launch_bounds__(ORDER_THREADS) void corr_order_kernel(const int* __restrict__ bins, int BE, int nbins,
                                                                   int* __restrict__ order, int starts) {
  corr_order_body<CACHE>(bins, BE, nbins, order, (int)blockIdx.x, (int)gridDim.x, starts != 0);
}

// -------------------------------------------------------------------------------------------------
// Generic path: arbitrary fmap2 strides (e.g. the reference's NCHW pyramid), any C.
// One workgroup per edge, one tap per lane, channel loop with the tensor's own strides.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void corr_fwd_generic_kernel(
    const T* __restrict__ fmap1, const T* __restrict__ fmap2, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, T* __restrict__ out, int E, int Np, int n2,
    int C, int H2, int W2, int64_t s_b, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int64_t out_estride,
    int64_t out_lstride, int64_t out_offset, int R, float coord_div) {
  typedef typename AccOf<T>::type A;
  __shared__ A s_raw[PP * MAXD * MAXD];
  __shared__ float s_dx[PP], s_dy[PP];
  __shared__ int s_ox[PP], s_oy[PP];
  const int D = 2 * R + 2;
  const int be = blockIdx.x;
  const int b = be / E, e = be % E;
  const int tid = threadIdx.x;
  if (tid < PP) {
    float x = coords[((int64_t)be * 2 + 0) * PP + tid] / coord_div;
    float y = coords[((int64_t)be * 2 + 1) * PP + tid] / coord_div;
    s_ox[tid] = floor_to_int(x) - R;
    s_oy[tid] = floor_to_int(y) - R;
    s_dx[tid] = x - floorf(x);
    s_dy[tid] = y - floorf(y);
  }
  __syncthreads();
  const T* __restrict__ f1 = fmap1 + ((int64_t)b * Np + ii[e]) * C * PP;
  const T* __restrict__ f2 = fmap2 + (int64_t)b * s_b + jj[e] * s_n;
  for (int o = tid; o < PP * D * D; o += NT) {
    int p = o / (D * D), a = (o / D) % D, c = o % D;
    int gy = s_oy[p] + a, gx = s_ox[p] + c;
    A s = 0;
    if (gy >= 0 && gy < H2 && gx >= 0 && gx < W2) {
      const T* src = f2 + (int64_t)gy * s_h + (int64_t)gx * s_w;
      if constexpr (sizeof(A) == 8) { for (int k = 0; k < C; k++) s = fma((double)f1[k * PP + p], (double)src[(int64_t)k * s_c], s); }
      else { for (int k = 0; k < C; k++) s = fmaf(to_f32<T>(f1[k * PP + p]), to_f32<T>(src[(int64_t)k * s_c]), s); }
    }
    s_raw[o] = s;
  }
  __syncthreads();
  corr_epilogue<T, A>(s_raw, s_dx, s_dy, out + (int64_t)be * out_estride + out_offset, D, out_lstride);
}

// -------------------------------------------------------------------------------------------------
// Backward (fp32).  One workgroup per edge, ONE CHANNEL PER LANE: for every box position the lane reads
// its channel of fmap2 (coalesced when channels-last), updates 9 register accumulators of d_fmap1 and
// emits ONE atomic per (position, channel) into d_fmap2 — consecutive lanes hit consecutive addresses.
// The reference issues 2*C scalar atomics per (pixel, tap) thread (correlation_kernel.cu:182-188).
// -------------------------------------------------------------------------------------------------
// SEG = true (segment-reduced backward, channels-last fmap2): this kernel computes d_fmap1 only and hands the window gradients G
// [9][D][D], the edge's geometry (BwdMeta) and its membership in the target frame's edge list to corr_bwd_tile_kernel, which owns
// d_fmap2 tile by tile — no global atomics on d_fmap2.
struct BwdMeta { int frame, patch, x0, y0, x1, y1, ox[PP], oy[PP]; };      // 24 ints; box clipped to the frame
template <bool SEG, int RMAX>
__global__ __launch_bounds__(NT) void corr_bwd_kernel(
    const float* __restrict__ fmap1, const float* __restrict__ fmap2, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, const float* __restrict__ grad,
    float* __restrict__ d1, float* __restrict__ d2, int E, int Np, int n2, int C, int H2, int W2, int64_t s_b,
    int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int R, float* __restrict__ gs, BwdMeta* __restrict__ meta,
    int* __restrict__ lists, int* __restrict__ cursors, int cap, unsigned long long* __restrict__ trace) {
  // One workgroup per edge.  Every dependent memory round trip of this kernel costs ~4 k cycles under load (all workgroups are
  // resident at once), so the phases are ordered for few of them: (1) coordinates + indices, (2) gradient block + patch features
  // together, then LDS only until the feature rows, which are requested 8 at a time, one step ahead of their use.
  unsigned long long t_prev = trace ? __builtin_readcyclecounter() : 0ull;
  auto stamp = [&](int ph) {                                  // debug (DEVO_CORR_BWD_TRACE): cycles of phase ph, thread 0 of every workgroup
    if (trace && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_readcyclecounter(); trace[(size_t)blockIdx.x * 8 + ph] = t - t_prev; t_prev = __builtin_readcyclecounter(); }
  };
  constexpr int DMX = 2 * RMAX + 2;
  __shared__ float s_g[PP * DMX * DMX];     // gradient of the raw D x D windows (correlation_kernel.cu:259-269)
  __shared__ float s_grad[(DMX - 1) * (DMX - 1) * PP];
  __shared__ float s_dx[PP], s_dy[PP];
  __shared__ int s_ox[PP], s_oy[PP];
  const int D = 2 * R + 2, Dm = D - 1;
  const int be = blockIdx.x;
  const int b = be / E, e = be % E;
  const int tid = threadIdx.x;
  const int64_t pi = ii[e], fj = jj[e];
  if (tid < PP) {
    float x = coords[((int64_t)be * 2 + 0) * PP + tid];
    float y = coords[((int64_t)be * 2 + 1) * PP + tid];
    s_ox[tid] = floor_to_int(x) - R;
    s_oy[tid] = floor_to_int(y) - R;
    s_dx[tid] = x - floorf(x);
    s_dy[tid] = y - floorf(y);
  }
  const float* __restrict__ f1 = fmap1 + ((int64_t)b * Np + pi) * C * PP;
  const float* __restrict__ f2 = fmap2 + (int64_t)b * s_b + fj * s_n;
  float* g1 = d1 + ((int64_t)b * Np + pi) * C * PP;
  float* g2 = d2 + (int64_t)b * s_b + fj * s_n;
  // the edge's gradient block (contiguous, logical [c][a][i0][j0]) with unconditional coalesced loads + this thread's patch
  // features of the first channel round: one round trip
  constexpr int GIT = ((DMX - 1) * (DMX - 1) * PP + NT - 1) / NT;
  float gl[GIT], w0[PP];
  {
    const float* g = grad + (int64_t)be * Dm * Dm * PP;
    const int ng = Dm * Dm * PP;
#pragma unroll
    for (int i = 0; i < GIT; i++) gl[i] = g[min(tid + i * NT, ng - 1)];
#pragma unroll
    for (int p = 0; p < PP; p++) w0[p] = f1[min(tid, C - 1) * PP + p];
#pragma unroll
    for (int i = 0; i < GIT; i++) if (tid + i * NT < ng) s_grad[tid + i * NT] = gl[i];
  }
  __syncthreads();
  stamp(0);
  {
    const float inv_dd = __builtin_amdgcn_rcpf((float)(D * D)), inv_d = __builtin_amdgcn_rcpf((float)D);
    for (int o = tid; o < PP * D * D; o += NT) {
      const int p = (int)(((float)o + 0.5f) * inv_dd), r_ = o - p * D * D;
      const int a = (int)(((float)r_ + 0.5f) * inv_d), c = r_ - a * D;
      const float dx = s_dx[p], dy = s_dy[p];
      float s = 0.0f;
      auto G = [&](int aa, int cc) -> float {
        return (aa >= 0 && aa < Dm && cc >= 0 && cc < Dm) ? s_grad[(cc * Dm + aa) * PP + p] : 0.0f;
      };
      s += (1.0f - dx) * (1.0f - dy) * G(a, c);
      s += dx * (1.0f - dy) * G(a, c - 1);
      s += (1.0f - dx) * dy * G(a - 1, c);
      s += dx * dy * G(a - 1, c - 1);
      const int gy = s_oy[p] + a, gx = s_ox[p] + c;
      if (!(gy >= 0 && gy < H2 && gx >= 0 && gx < W2)) s = 0.0f;   // out-of-bounds taps contribute nothing (:182)
      s_g[o] = s;
    }
  }
  __syncthreads();
  stamp(1);

  int xmin = s_ox[0], xmax = s_ox[0], ymin = s_oy[0], ymax = s_oy[0];
#pragma unroll
  for (int p = 1; p < PP; p++) {
    xmin = min(xmin, s_ox[p]); xmax = max(xmax, s_ox[p]);
    ymin = min(ymin, s_oy[p]); ymax = max(ymax, s_oy[p]);
  }
  // clip the box to the frame: positions outside carry zero gradient
  const int x0 = max(xmin, 0), x1 = min(xmax + D, W2), y0 = max(ymin, 0), y1 = min(ymax + D, H2);

  // Per box position the 9 window gradients that land on it depend on the position only: lane l of every wave builds them for
  // position base + l in registers, and the wave walks the 64 positions reading them with v_readlane (no LDS traffic — broadcast
  // ds_read_b128 rows cost the full 1 KB of LDS bandwidth each —, no barrier); the feature rows are requested 8 at a time, one
  // step ahead of their use.
  const int bw = max(x1 - x0, 0), npos = bw * max(y1 - y0, 0);
  const float inv_bw = __builtin_amdgcn_rcpf((float)max(bw, 1));
  const int lane = tid & 63;
  constexpr int RU = 8;                                         // feature rows in flight per thread (x 2: the step ahead)
  for (int k0 = 0; k0 < C; k0 += NT) {
    const int k = k0 + tid;
    const bool live = k < C;
    float w[PP], acc[PP];
#pragma unroll
    for (int p = 0; p < PP; p++) { w[p] = k0 == 0 ? w0[p] : (live ? f1[k * PP + p] : 0.0f); acc[p] = 0.0f; }
    const int64_t kc = (int64_t)min(k, C - 1) * s_c;
    for (int base = 0; base < npos; base += 64) {
      const int q = min(base + lane, npos - 1);
      const int ry = (int)(((float)q + 0.5f) * inv_bw), rx = q - ry * bw;
      const int gy = y0 + ry, gx = x0 + rx;
      float tg[PP];
      bool any = false;
#pragma unroll
      for (int p = 0; p < PP; p++) {
        const int a = gy - s_oy[p], c = gx - s_ox[p];
        tg[p] = (a >= 0 && a < D && c >= 0 && c < D) ? s_g[p * D * D + a * D + c] : 0.0f;
        any |= (tg[p] != 0.0f);
      }
      const int toff = (int)((int64_t)gy * s_h + (int64_t)gx * s_w);          // (in-frame offsets fit 31 bits: checked by the launcher)
      const unsigned long long anym = __ballot(any);                           // rows with any gradient at all
      const int cnt = min(64, npos - base);
      auto request = [&](float (&v)[RU], int r) {
#pragma unroll
        for (int u = 0; u < RU; u++) v[u] = f2[(int64_t)__builtin_amdgcn_readlane(toff, min(r + u, cnt - 1)) + kc];
      };
      float vn[RU];
      request(vn, 0);
      for (int r = 0; r < cnt; r += RU) {
        float v[RU];
#pragma unroll
        for (int u = 0; u < RU; u++) v[u] = vn[u];
        if (r + RU < cnt) request(vn, r + RU);                  // wave-uniform
#pragma unroll
        for (int u = 0; u < RU; u++) {
          if (r + u < cnt && ((anym >> (r + u)) & 1ull)) {      // wave-uniform
            float t = 0.0f;
#pragma unroll
            for (int p = 0; p < PP; p++) {
              const float g_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tg[p]), r + u));
              acc[p] = fmaf(g_, v[u], acc[p]); t = fmaf(g_, w[p], t);
            }
            if (!SEG && live) atomicAdd(g2 + (int64_t)__builtin_amdgcn_readlane(toff, r + u) + kc, t);
          }
        }
      }
    }
    stamp(2);
    if (live) {
#pragma unroll
      for (int p = 0; p < PP; p++) atomicAdd(g1 + k * PP + p, acc[p]);
    }
  }
  if (SEG) {
    // hand-over to corr_bwd_tile_kernel, last (nothing in this kernel waits for these stores / the list slot)
    for (int o = tid; o < PP * D * D; o += NT) gs[(int64_t)be * (PP * D * D) + o] = s_g[o];
    if (tid == 0) {
      BwdMeta m;
      m.frame = b * n2 + (int)fj; m.patch = b * Np + (int)pi; m.x0 = x0; m.y0 = y0; m.x1 = x1; m.y1 = y1;
#pragma unroll
      for (int p = 0; p < PP; p++) { m.ox[p] = s_ox[p]; m.oy[p] = s_oy[p]; }
      meta[be] = m;
      if (x1 > x0 && y1 > y0) lists[(int64_t)m.frame * cap + atomicAdd(&cursors[m.frame], 1)] = be;      // (order: whoever comes first)
    }
  }
  stamp(3);
}

// d_fmap2 of the segment-reduced backward: ONE workgroup owns a tile (frame, band of BH rows, slab of 16 channels) of the
// channels-last gradient, accumulates every edge of the frame whose box touches the band into the tile in LDS (ds_add_f32) and
// stores the tile once — plain stores, every tile is written (no memset, no global atomics).  Wave w takes the band's edges
// w, w + 4, ..: lane = (position of 4, channel of 16); the edge's window gradients G sit in a wave-private LDS area.
constexpr int BWD_CS = 16;                    // channels per slab (64 bytes of a channels-last pixel)
constexpr int BWD_THREADS = 256;
__global__ __launch_bounds__(BWD_THREADS) void corr_bwd_tile_kernel(
    const float* __restrict__ fmap1, const float* __restrict__ gs, const BwdMeta* __restrict__ meta, const int* __restrict__ lists,
    const int* __restrict__ cursors, float* __restrict__ d2, int n2, int C, int H2, int W2, int64_t s_b, int64_t s_n, int64_t s_h,
    int64_t s_w, int D, int BH, int cap) {
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  const int DD = D * D;
  float* const tile = s_dyn;                                  // [BH][W2][16]
  float* const s_gw = tile + (size_t)BH * W2 * BWD_CS;        // [4 waves][64 rows of 12 floats]
  constexpr int SCAN = 128;                                   // edges examined per round (threads 0..127)
  __shared__ int s_em[SCAN][16];                              // per listed edge: be, patch, x0, x1, y0, y1, 9 x (ox - x0 | (oy - y0) << 16)
  __shared__ int s_cnt[BWD_THREADS / 64 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slab = blockIdx.x, band = blockIdx.y, frame = blockIdx.z;
  const int ylo = band * BH, yhi = min(ylo + BH, H2);
  for (int i = tid; i < BH * W2 * BWD_CS / 4; i += BWD_THREADS) reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int ne = cursors[frame];
  const int ch = lane & 15, pq = lane >> 4;
  float* const gw = s_gw + wave * (64 * 12);
  for (int base = 0; base < ne; base += SCAN) {
    // the frame's edges base .. base + 127 whose box touches this band -> s_em (compacted), geometry included: the per-edge
    // work below starts without a dependent global load
    int e = -1;
    BwdMeta m_;
    if (tid < SCAN && base + tid < ne) {
      e = lists[(int64_t)frame * cap + base + tid];
      m_ = meta[e];
      if (!(m_.y0 < yhi && m_.y1 > ylo)) e = -1;
    }
    const unsigned long long m = __ballot(e >= 0);
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();                                          // (also: the tile is zeroed / the previous chunk is done)
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BWD_THREADS / 64; w++) { const int c = s_cnt[w]; off += w < wave ? c : 0; total += c; }
    if (e >= 0) {
      int* d = s_em[off + __popcll(m & ((1ull << lane) - 1ull))];
      d[0] = e; d[1] = m_.patch; d[2] = m_.x0; d[3] = m_.x1; d[4] = m_.y0; d[5] = m_.y1;
#pragma unroll
      for (int p = 0; p < PP; p++) d[6 + p] = ((m_.ox[p] - m_.x0) & 0xffff) | ((m_.oy[p] - m_.y0) << 16);      // (origins may lie outside the clipped box: signed halves)
    }
    __syncthreads();
    for (int k = wave; k < total; k += BWD_THREADS / 64) {    // wave-uniform
      const int* d = s_em[k];
      const int be = d[0], pidx = d[1], x0 = d[2], x1 = d[3], ya = max(d[4], ylo), yb = min(d[5], yhi), by0 = d[4];
      float w1[PP];
#pragma unroll
      for (int p = 0; p < PP; p++) w1[p] = fmap1[((int64_t)pidx * C + slab * BWD_CS + ch) * PP + p];
      const int bw = x1 - x0, npos = bw * (yb - ya);
      const float inv_bw = __builtin_amdgcn_rcpf((float)bw);
      const float* ge = gs + (int64_t)be * (PP * DD);
      for (int pb = 0; pb < npos; pb += 64) {
        // lane = position: its 9 window gradients (unconditional gathers: one round trip) -> the wave's rows [9 values, tile index]
        const int q = min(pb + lane, npos - 1);
        const int ry = (int)(((float)q + 0.5f) * inv_bw), rx = q - ry * bw;
        const int gy = ya + ry, gx = x0 + rx;
        float gv[PP];
#pragma unroll
        for (int p = 0; p < PP; p++) {
          const int o = d[6 + p];
          const int a = gy - by0 - (o >> 16), c = rx - (int)(short)(o & 0xffff);
          const bool in = a >= 0 && a < D && c >= 0 && c < D;
          const float g = ge[p * DD + (in ? a * D + c : 0)];
          gv[p] = in ? g : 0.0f;
        }
#pragma unroll
        for (int p = 0; p < PP; p++) gw[lane * 12 + p] = gv[p];
        gw[lane * 12 + 9] = __int_as_float(((gy - ylo) * W2 + gx) * BWD_CS);
        wave_lds_fence();
        const int cnt = min(64, npos - pb);
        for (int r = pq; r < cnt; r += 4) {                       // lane = (position r of 4, channel)
          const float4 ta = *reinterpret_cast<const float4*>(&gw[r * 12]), tb = *reinterpret_cast<const float4*>(&gw[r * 12 + 4]),
                       tc = *reinterpret_cast<const float4*>(&gw[r * 12 + 8]);
          float t = ta.x * w1[0];
          t = fmaf(ta.y, w1[1], t); t = fmaf(ta.z, w1[2], t); t = fmaf(ta.w, w1[3], t); t = fmaf(tb.x, w1[4], t);
          t = fmaf(tb.y, w1[5], t); t = fmaf(tb.z, w1[6], t); t = fmaf(tb.w, w1[7], t); t = fmaf(tc.x, w1[8], t);
          __hip_atomic_fetch_add(&tile[__float_as_int(tc.y) + ch], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        wave_lds_fence();                                       // the next chunk / edge overwrites the wave's rows
      }
    }
    __syncthreads();                                          // s_em is rebuilt by the next chunk
  }
  __syncthreads();
  // the tile -> d_fmap2 (channels-last: 64 contiguous bytes per pixel and slab)
  float* const base2 = d2 + (int64_t)(frame / n2) * s_b + (int64_t)(frame % n2) * s_n + slab * BWD_CS;
  for (int i = tid; i < (yhi - ylo) * W2 * (BWD_CS / 4); i += BWD_THREADS) {
    const int pix = i >> 2, qd = i & 3;
    const int ry = pix / W2, x = pix - ry * W2;
    *reinterpret_cast<float4*>(base2 + (int64_t)(ylo + ry) * s_h + (int64_t)x * s_w + qd * 4) = reinterpret_cast<const float4*>(tile)[i];
  }
}

// -------------------------------------------------------------------------------------------------
// patchify (correlation_kernel.cu:16-80): integer-offset gather of (2R+2)^2 windows, one lane per output.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ void patchify_fwd_kernel(const T* __restrict__ net, const float* __restrict__ coords, T* __restrict__ out,
                                    int M, int C, int H, int W, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int R,
                                    int64_t total) {
  const int D = 2 * R + 2;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < total; n += (int64_t)blockDim.x * gridDim.x) {
    int c = n % D; int64_t r = n / D;
    int a = r % D; r /= D;
    int k = r % C; r /= C;
    int m = r % M; int b = r / M;
    float x = coords[((int64_t)b * M + m) * 2], y = coords[((int64_t)b * M + m) * 2 + 1];
    int i = floor_to_int(y) + a - R, j = floor_to_int(x) + c - R;
    T v = from_f32<T>(0.0f);
    if (i >= 0 && i < H && j >= 0 && j < W) v = net[b * sb + k * sc + i * sh + j * sw];
    out[n] = v;
  }
}

}  // namespace devo
#include "corr_bwd_mfma.h"
namespace devo {

template <typename T>
__global__ void patchify_bwd_kernel(const float* __restrict__ coords, const T* __restrict__ grad, T* __restrict__ dnet,
                                    int M, int C, int H, int W, int R, int64_t total, int64_t sb, int64_t sc, int64_t sh, int64_t sw) {
  const int D = 2 * R + 2;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < total; n += (int64_t)blockDim.x * gridDim.x) {
    int c = n % D; int64_t r = n / D;
    int a = r % D; r /= D;
    int k = r % C; r /= C;
    int m = r % M; int b = r / M;
    float x = coords[((int64_t)b * M + m) * 2], y = coords[((int64_t)b * M + m) * 2 + 1];
    int i = floor_to_int(y) + a - R, j = floor_to_int(x) + c - R;
    if (i >= 0 && i < H && j >= 0 && j < W) atomicAdd(dnet + b * sb + k * sc + i * sh + j * sw, grad[n]);
  }
}

// -------------------------------------------------------------------------------------------------
// Pyramid build for the lookup (devo/devo.py:526-527, devo/utils.py:70-79): from NCHW feature frames to the
// channel-blocked storage the staged lookup kernel wants, both levels in one pass over the input:
//   l0[f][c/8][y][x][c%8]        = fmap[f][c][y][x]                                   (avg_pool2d(.,1,1) = copy)
//   l1[f][c/8][y/4][x/4][c%8]    = mean of the 4x4 window (F.avg_pool2d(fmap, 4, 4): floor(H/4) x floor(W/4))
// One workgroup = 8 channels x 4 rows x 128 columns, transposed through LDS: reads are coalesced along x of one
// channel row, writes are whole 32-byte pixel blocks of consecutive pixels.
// -------------------------------------------------------------------------------------------------
constexpr int PYR_WC = 128;
template <typename T>
__global__ __launch_bounds__(256) void pyramid_blocked_kernel(const T* __restrict__ fmap, T* __restrict__ l0, T* __restrict__ l1,
                                                              int C, int H, int W, int64_t f_stride, int64_t l0_fstride,
                                                              int64_t l1_fstride) {
  __shared__ float tile[8][4][PYR_WC + 1];
  const int nrb = (H + 3) / 4, ncb = C / 8;
  const int f = blockIdx.x / (ncb * nrb), rem = blockIdx.x - f * ncb * nrb;
  const int cb = rem / nrb, rb = rem - cb * nrb;
  const int x0 = blockIdx.y * PYR_WC, y0 = rb * 4;
  const int wc = min(PYR_WC, W - x0);
  const T* src = fmap + (int64_t)f * f_stride + (int64_t)cb * 8 * H * W;
  for (int i = threadIdx.x; i < 8 * 4 * PYR_WC; i += 256) {
    const int x = i % PYR_WC, r = (i / PYR_WC) % 4, ch = i / (4 * PYR_WC);
    const int y = y0 + r;
    tile[ch][r][x] = (x < wc && y < H) ? to_f32<T>(src[((int64_t)ch * H + y) * W + x0 + x]) : 0.0f;
  }
  __syncthreads();
  T* d0 = l0 + (int64_t)f * l0_fstride + (int64_t)cb * H * W * 8;
  for (int i = threadIdx.x; i < 4 * PYR_WC * 8; i += 256) {              // consecutive threads -> consecutive output elements
    const int ch = i % 8, x = (i / 8) % PYR_WC, r = i / (8 * PYR_WC);
    const int y = y0 + r;
    if (x < wc && y < H) d0[((int64_t)y * W + x0 + x) * 8 + ch] = from_f32<T>(tile[ch][r][x]);
  }
  const int H4 = H / 4, W4 = W / 4;
  if (l1 && rb < H4) {
    T* d1 = l1 + (int64_t)f * l1_fstride + ((int64_t)cb * H4 + rb) * W4 * 8;
    for (int i = threadIdx.x; i < (PYR_WC / 4) * 8; i += 256) {
      const int ch = i % 8, xp = i / 8;
      const int gx = x0 / 4 + xp;
      if (gx < W4) {
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int u = 0; u < 4; u++) sum += tile[ch][r][4 * xp + u];
        d1[(int64_t)gx * 8 + ch] = from_f32<T>(sum * 0.0625f);
      }
    }
  }
}

}  // namespace devo

using namespace devo;

// Describes one level for the fast kernels; false = neither of them can read it (generic kernel, or an error for
// channel-blocked storage, which only the fast kernels understand).
static bool corr_mfma_enabled() {                // DEVO_CORR_MFMA=0: fp32 lookups take the staged (tap-centric) kernel instead
  static const char* env = getenv("DEVO_CORR_MFMA");
  static const bool on = !(env && env[0] == '0');
  return on;
}

template <typename T>
static bool staged_level(const void* fmap2, int C, int H2, int W2, const int64_t* f2s, int cblock, int64_t out_offset,
                         float coord_div, CorrLevel* lv, int* err, const int* exps = nullptr) {
  // channel-blocked storage [.., C/cb, H, W, cb]: f2s[2] is the stride between channel blocks, the cb channels of a
  // pixel are contiguous.  The staged kernel wants cb == its channel chunk (8), the matrix-core kernel any multiple of 4.
  // cblock == DEVO_CBLOCK_SPLIT8 (fp32 only): the split-blocked format of devo_corr_pyramid_split — the strides of an 8-channel
  // blocked level, every 32-byte pixel block holding fp16 (hi0..7 | lo0..7) — which only the dense-product kernel reads.
  const bool split = cblock == DEVO_CBLOCK_SPLIT8;
  *err = DEVO_OK;
  if (split && (sizeof(T) != 4 || exps == nullptr)) {
    set_error("devo_corr_forward: the split-blocked format is an fp32 format and needs its scale exponents");
    *err = DEVO_ERR_ARG;
    return false;
  }
  if (split) cblock = 8;
  const bool blocked = cblock > 1;
  const int64_t v = 16 / sizeof(T);
  const int cb = blocked ? cblock : C;
  const long long plane_bytes = ((long long)(H2 - 1) * f2s[3] + (long long)(W2 - 1) * f2s[4] + (blocked ? cb : C)) * (long long)sizeof(T);
  const long long frame_bytes = blocked ? (long long)(C / (cb > 0 ? cb : 1) - 1) * f2s[2] * (long long)sizeof(T) + plane_bytes : plane_bytes;
  const bool aligned = sizeof(T) <= 4 && (blocked || f2s[2] == 1) && (f2s[3] % v == 0) && (f2s[4] % v == 0) && (f2s[0] % v == 0) &&
                       (f2s[1] % v == 0) && (!blocked || f2s[2] % v == 0) && ((reinterpret_cast<uintptr_t>(fmap2) & 15) == 0) &&
                       f2s[3] >= 0 && f2s[4] >= 0 && (!blocked || (f2s[2] >= 0 && C % cb == 0));
  lv->staged_ok = !split && aligned && (C % KC == 0) && (!blocked || cblock == KC) && plane_bytes < (1LL << 31);   // 32-bit in-plane offsets
  // the matrix-core kernel: C = 128, 16-byte pieces inside a channel block, piece offsets linear in the step
  const bool cb_ok = !blocked || (sizeof(T) == 4 ? (cb == 4 || cb == 8 || cb == 16) : (cb == 8 || cb == 16 || cb == 32));
  const bool c_ok = sizeof(T) == 4 ? (C == 64 || C == 128) : (C == 128 || C == 256);            // 4 or 8 steps per pass
  lv->mfma_ok = !split && aligned && sizeof(T) <= 4 && corr_mfma_enabled() && c_ok && cb_ok &&
                frame_bytes < (1LL << 31);                                                             // 32-bit in-frame offsets
  // the dense-product kernel (corr_mm.h): 16-byte pieces of 8 channels inside a channel block, 32 channels per K step; fp16 levels as
  // they are, fp32 levels in the split-blocked format only (raw fp32 levels take the 4x4 matrix-core kernel: exact fp32 products)
  lv->mm_ok = aligned && frame_bytes < (1LL << 31) && C % 32 == 0 &&
              (sizeof(T) == 2 ? (lv->mfma_ok && C <= 256) : (sizeof(T) == 4 && split && corr_mfma_enabled() && C <= 128));
  lv->split = split; lv->exps = exps;
  if (!lv->staged_ok && !lv->mfma_ok && !lv->mm_ok) {
    if (blocked) {
      set_error("devo_corr_forward: channel-blocked fmap2 needs fp32 / fp16, 16-byte aligned strides and cblock == %d (got %d)", KC, cblock);
      *err = DEVO_ERR_UNSUPPORTED;
    }
    return false;
  }
  lv->fmap2 = fmap2; lv->H2 = H2; lv->W2 = W2;
  lv->s_b = f2s[0]; lv->s_n = f2s[1]; lv->s_h = f2s[3]; lv->s_w = f2s[4];
  lv->chunk_stride = blocked ? f2s[2] : KC;
  lv->cb_shift = 30; if (blocked) { int sh = 0; while ((1 << sh) < cb) sh++; lv->cb_shift = sh; }
  lv->block_stride = blocked ? f2s[2] : 0;
  lv->frame_bytes = (unsigned)(frame_bytes < (1LL << 31) ? frame_bytes : 0);
  lv->out_offset = out_offset; lv->coord_div = coord_div;
  return true;
}

// nlev = 1 or 2 levels in ONE launch of the staged kernel
// What the last forward lookup of this thread launched (devo_corr_forward_last_path) and, once per process and kind, a line on stderr when a
// call takes one of the slow kernels at a size where it matters (DEVO_LOG_FALLBACK=0 silences it): a layout / dtype / stride the fast kernels
// do not read used to cost 2 - 24 x without a trace.
enum { CORR_PATH_MM = 0, CORR_PATH_MFMA4 = 1, CORR_PATH_STAGED = 2, CORR_PATH_GENERIC = 3, CORR_PATH_MM_GROUPS = 4 };
static thread_local int g_corr_fwd_path = -1;
static void corr_note_path(int path, long long BE, const char* why) {
  g_corr_fwd_path = path;
  if (path != CORR_PATH_GENERIC && path != CORR_PATH_STAGED) return;
  static bool noted[8] = {false, false, false, false, false, false, false, false};
  static const bool quiet = [] { const char* e = getenv("DEVO_LOG_FALLBACK"); return e && e[0] == '0'; }();
  if (quiet || noted[path] || BE < 2048) return;
  noted[path] = true;
  fprintf(stderr, "[devo_hip] cuda_corr.forward: %s kernel for %lld edges (%s); cuda_corr.last_forward_path() names the kernel of every call\n",
          path == CORR_PATH_GENERIC ? "the generic (slowest, ~24x)" : "the staged tap-centric (~2x)", BE, why);
}

template <typename T>
static int launch_staged(const void* fmap1, const CorrLevel& lv0, const CorrLevel& lv1, int nlev, const float* coords,
                         const int64_t* ii, const int64_t* jj, void* out, long long BE, int E, int Np, int n2, int C,
                         int64_t oes, int64_t ols, int R, const int* order, hipStream_t st) {
  if (lv0.split || (nlev == 2 && lv1.split)) {
    set_error("devo_corr_forward: a split-blocked level is only readable by the dense-product kernel (needs fmap1_t, out_lstride > 0, both levels split-blocked)");
    return DEVO_ERR_UNSUPPORTED;
  }
  const unsigned per_level = (nlev == 2) ? (unsigned)((BE + 7) / 8 * 8) : (unsigned)BE;      // whole groups of 8 alternate
  dim3 grid(per_level * nlev), block(WPB * 64);
  static const bool force4 = getenv("DEVO_CORR_NP4") != nullptr;      // debug switch: run the r > 3 instantiation
  unsigned long long* trace = nullptr;                                // debug switch: per-wave cycle stamps to stderr
  const bool do_trace = getenv("DEVO_CORR_TRACE") != nullptr;
  const size_t nrec = (size_t)BE * nlev;
  if (do_trace) { (void)hipMalloc(&trace, nrec * 64); (void)hipMemset(trace, 0, nrec * 64); }
  const bool mfma = lv0.mfma_ok && (nlev == 1 || lv1.mfma_ok);
  corr_note_path(mfma ? CORR_PATH_MFMA4 : CORR_PATH_STAGED, BE, "channel count / alignment outside the matrix-core kernels, or DEVO_CORR_MFMA=0");
  if (mfma) {                                                         // matrix-core kernel (corr_mfma.h)
    static const char* split_env = getenv("DEVO_CORR_SPLIT_LEVELS");  // debug: fused lookups as two sets of workgroups
    const bool both = nlev == 2 && !(split_env && split_env[0] == '1') && !do_trace;
    typedef typename std::conditional<std::is_same<T, double>::value, float, T>::type MT;   // (never fp64: mfma_ok is false)
    typedef void (*mfma_fn_t)(const MT*, CorrLevel, CorrLevel, int, const float*, const int64_t*, const int64_t*, MT*, int, int,
                              int, int, int, int64_t, int64_t, int, const int*, unsigned long long*, int);
    // steps of 16 (fp32) / 32 (fp16) channels per pass, 4 or 8 of them (16 would not fit the patch into the registers):
    // C = 64 / 128 (fp32), 128 / 256 (fp16)
    constexpr int SC = sizeof(MT) == 2 ? 32 : 16;
    const int ngr = C / SC;
#define DEVO_MFMA_PICK(NGR) (both ? (R <= 3 ? corr_fwd_mfma_kernel<MT, 3, NGR, 2> : corr_fwd_mfma_kernel<MT, 5, NGR, 2>) \
                                  : (R <= 3 ? corr_fwd_mfma_kernel<MT, 3, NGR, 1> : corr_fwd_mfma_kernel<MT, 5, NGR, 1>))
    const mfma_fn_t fn = ngr == 4 ? DEVO_MFMA_PICK(4) : DEVO_MFMA_PICK(8);
#undef DEVO_MFMA_PICK
    // DEVO_MFMA_EPW edges (waves) per workgroup; grids are whole groups of 8 workgroups (one per XCD)
    const unsigned wg_level = (unsigned)(((BE + DEVO_MFMA_EPW - 1) / DEVO_MFMA_EPW + 7) / 8 * 8);
    const dim3 mgrid(DEVO_MFMA_EPW == 1 ? (both ? (unsigned)BE : per_level * nlev) : (both || nlev == 1 ? wg_level : wg_level * 2)), mblock(64 * DEVO_MFMA_EPW);
    // Waves per CU: the kernel's registers allow 16.  Every resident wave streams its edge's boxes through the XCD's 4 MB L2, and
    // plan neighbours share them: when the boxes of all resident waves together overrun the L2 (large radius, dense patch graph:
    // BASELINE's stress configuration) the shared lines are evicted before the neighbour asks for them and the kernel becomes
    // HBM-bound on re-reads.  An unused dynamic LDS allocation caps the resident waves (DEVO_MFMA_WAVES_PER_CU overrides).
    size_t occ_pad = 0;
    {
      static const char* occ_env = getenv("DEVO_MFMA_WAVES_PER_CU");
      int occ = occ_env ? atoi(occ_env) : 0;
      if (occ > 0 && occ < 16) {
        const size_t per_wg = (size_t)(160 * 1024) / (size_t)occ;
        const size_t stat = 10 * 1024;                                   // (static LDS of one wave's workgroup, rounded up)
        occ_pad = per_wg > stat + 512 ? ((per_wg - stat) / 512) * 512 : 0;
        if (occ_pad > 64 * 1024 - stat) {
          if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)occ_pad) != hipSuccess) { (void)hipGetLastError(); occ_pad = 64 * 1024 - stat; }
        }
      }
    }
    hipLaunchKernelGGL(fn, mgrid, mblock, occ_pad, st, (const MT*)fmap1, lv0, lv1, nlev, coords, ii, jj, (MT*)out, (int)BE, E, Np, n2,
                       C, oes, ols, R, order, trace, 0);
  } else
  if (!(lv0.staged_ok && (nlev == 1 || lv1.staged_ok))) {
    set_error("devo_corr_forward: this channel-blocked layout is only readable by the matrix-core kernel (fp32: C = 64 / 128, fp16: C = 128 / 256)");
    if (trace) (void)hipFree(trace);
    return DEVO_ERR_UNSUPPORTED;
  } else if (R <= 3 && !force4)   // (the <3,5> instantiation has room for every supported radius)
    hipLaunchKernelGGL((corr_fwd_cl_kernel<T, 1, 3>), grid, block, 0, st, (const T*)fmap1, lv0, lv1, nlev, coords, ii, jj,
                       (T*)out, (int)BE, E, Np, n2, C, oes, ols, R, order, trace);
  else
    hipLaunchKernelGGL((corr_fwd_cl_kernel<T, 3, 5>), grid, block, 0, st, (const T*)fmap1, lv0, lv1, nlev, coords, ii, jj,
                       (T*)out, (int)BE, E, Np, n2, C, oes, ols, R, order, trace);
  if (do_trace) {
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nrec * 8);
    (void)hipMemcpy(h.data(), trace, nrec * 64, hipMemcpyDeviceToHost);
    for (int l = 0; l < nlev; l++) {
      double ph[4] = {0, 0, 0, 0}, sum = 0, mx = 0;
      long cnt = 0;
      for (long long i = 0; i < BE; i++) {
        const unsigned long long* t = &h[((size_t)l * BE + i) * 8];
        if (!t[1]) continue;
        ph[0] += (double)(t[4] - t[0]); ph[1] += (double)(t[5] - t[4]); ph[2] += (double)(t[6] - t[5]); ph[3] += (double)(t[1] - t[6]);
        const double d = (double)(t[1] - t[0]);
        sum += d; if (d > mx) mx = d;
        cnt++;
      }
      if (!cnt) cnt = 1;
      fprintf(stderr, "[corr trace] level slot %d: wave mean %.0f max %.0f cycles; phase means: geometry %.0f, first chunk %.0f, channel loop %.0f, epilogue %.0f\n",
              l, sum / cnt, mx, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt);
    }
    (void)hipFree(trace);
  }
  return check_launch("devo_corr_forward");
}

// Dense-product lookup (corr_mm.h): both levels of an edge in one wave, the patch operand from its transposed copy fmap1_t.
static bool corr_mm_enabled() {                  // DEVO_CORR_MM=0: the 4x4 matrix-core kernel (corr_mfma.h: exact fp32 products) instead
  static const char* env = getenv("DEVO_CORR_MM");
  static const bool on = !(env && env[0] == '0');
  return on;
}
template <typename T>
static bool mm_eligible(const CorrLevel& l0, const CorrLevel& l1, const void* fmap1_t, long long BE, int Np, int C) {
  return corr_mm_enabled() && fmap1_t != nullptr && l0.mm_ok && l1.mm_ok && l0.out_offset >= 0 && l1.out_offset >= 0 && (long long)Np * C * PP * (long long)sizeof(T) < (1LL << 40) &&
         BE < (1LL << 30) && (reinterpret_cast<uintptr_t>(fmap1_t) & 15) == 0;
}
template <typename T>
static int launch_mm(const void* fmap1_t, const CorrLevel& lv0, const CorrLevel& lv1, int nlev, const float* coords, const int64_t* ii,
                     const int64_t* jj, void* out, long long BE, int E, int Np, int n2, int C, int64_t oes, int64_t ols, int R,
                     const int* order, hipStream_t st, int order_kind = 0) {
  typedef typename std::conditional<std::is_same<T, double>::value, float, T>::type MT;
  typedef void (*mm_fn_t)(const MT*, CorrLevel, CorrLevel, int, const float*, const int64_t*, const int64_t*, MT*, int, int, int, int, int,
                          int64_t, int64_t, int, const int*, int, unsigned long long*, const int*, MmGroupArgs);
  // fp32: the patch operand's scale exponents sit behind its records (devo_corr_patch_operand_bytes)
  const int* exp1 = sizeof(MT) == 4 ? reinterpret_cast<const int*>(static_cast<const char*>(fmap1_t) + (size_t)(BE / E) * Np * C * PP * 4) : nullptr;
  const int nks = C / 32;
  mm_fn_t fn = nullptr;
#define DEVO_MM_PICK_L(NKS, NLV) (R == 3 ? corr_fwd_mm_kernel<MT, 3, NKS, NLV, 3> : R == 5 ? corr_fwd_mm_kernel<MT, 5, NKS, NLV, 5> : \
                                  R < 3 ? corr_fwd_mm_kernel<MT, 3, NKS, NLV, 0> : corr_fwd_mm_kernel<MT, 5, NKS, NLV, 0>)
#define DEVO_MM_PICK(NKS) (nlev == 2 ? DEVO_MM_PICK_L(NKS, 2) : DEVO_MM_PICK_L(NKS, 1))
  if constexpr (sizeof(MT) == 2) fn = nks == 1 ? DEVO_MM_PICK(1) : nks == 2 ? DEVO_MM_PICK(2) : nks == 4 ? DEVO_MM_PICK(4) : nks == 8 ? DEVO_MM_PICK(8) : nullptr;
  else fn = nks == 1 ? DEVO_MM_PICK(1) : nks == 2 ? DEVO_MM_PICK(2) : nks == 4 ? DEVO_MM_PICK(4) : nullptr;
#undef DEVO_MM_PICK
#undef DEVO_MM_PICK_L
  if (!fn) { set_error("devo_corr_forward_pyramid2: C = %d not supported by the dense-product kernel", C); return DEVO_ERR_UNSUPPORTED; }
  corr_note_path(CORR_PATH_MM, BE, "");
  // GROUP form (corr_mm.h, NW > 1): a group plan, radius 3, C = 128, level 1 = a quarter-resolution level in 8-channel blocks whose
  // region fits the LDS next to the waves' result areas
  static const bool group_off = []() { const char* e = getenv("DEVO_CORR_GROUP"); return e && e[0] == '0'; }();
  if (order_kind == DEVO_PLAN_GROUPS && order != nullptr && nlev == 2 && R == 3 && nks == 4 && lv1.cb_shift == 3 && !group_off && lv0.H2 / 4 == lv1.H2 &&
      lv0.W2 / 4 == lv1.W2 && lv0.coord_div == 1.0f && lv1.coord_div == 4.0f && BE / E == 1) {
    const long long nbins = corr_grp_nbins(1, n2, lv0.H2, lv0.W2, 4);
    if (nbins > 0) {
      corr_note_path(CORR_PATH_MM_GROUPS, BE, "");
      constexpr int NWG = sizeof(MT) == 2 ? 16 : 8;                  // waves per workgroup: what the registers allow per CU (one workgroup per CU: the region)
      mm_fn_t gfn = corr_fwd_mm_kernel<MT, 3, 4, 2, 3, NWG>;
      // per wave: result area + geometry records (the kernel's WAVE_LDS)
      constexpr int cap = sizeof(MT) == 4 ? 128 : DEVO_MM_CAP;       // (the kernel's CAP)
      constexpr int rw_floats = (PP * (8 * 8 + 1) + 3) / 4 * 4 > PP * (cap + 4) ? (PP * (8 * 8 + 1) + 3) / 4 * 4 : PP * (cap + 4);
      constexpr int wave_lds = rw_floats * 4 + 2 * 16 * 4 * 4 + ((2 * PP * 2 * 4 + 15) / 16) * 16;
      const int lds = mm_region_bytes<MT>(128) + NWG * wave_lds;
      static PerDeviceOnce attr_set;
      if (attr_set.first()) {
        if (hipFuncSetAttribute((const void*)gfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { (void)hipGetLastError(); set_error("devo_corr_forward_pyramid2: %d bytes of LDS refused", lds); return DEVO_ERR_LAUNCH; }
      }
      const MmGroupArgs ga{order + 2 * BE + 2, (int)nbins, corr_grp_count(lv1.H2), corr_grp_count(lv1.W2)};
      const unsigned items = (unsigned)nbins * MM_ITEMS_PER_BIN;
      const unsigned gwg = (items + 7) / 8 * 8;
      unsigned long long* gtrace = nullptr;                          // debug switch: per-wave 100 MHz stamps -> a schedule summary on stderr
      if (getenv("DEVO_CORR_TRACE") != nullptr) { (void)hipMalloc(&gtrace, (size_t)gwg * NWG * 64); (void)hipMemset(gtrace, 0, (size_t)gwg * NWG * 64); }
      hipLaunchKernelGGL(gfn, dim3(gwg), dim3(64 * NWG), lds, st, (const MT*)fmap1_t, lv0, lv1, nlev, coords, ii, jj, (MT*)out, (int)BE, E, Np, n2, C,
                         oes, ols, R, order, 0, gtrace, exp1, ga);
      if (gtrace) {
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)gwg * NWG * 8);
        (void)hipMemcpy(h.data(), gtrace, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t i = 0; i < (size_t)gwg * NWG; i++) if (h[i * 8 + 3]) { t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 3]); }
        // per compute unit (XCC id, SE id, CU id of HW_ID): busy time = union of its workgroups' spans; items; edges
        struct CuAcc { double busy = 0, first = 1e30, last = 0; int wgs = 0, live = 0, edges = 0; };
        std::vector<CuAcc> cu(16 * 1024);
        double live_span = 0, empty_span = 0, stage = 0, item = 0, tail = 0; long nlive = 0, nempty = 0;
        for (unsigned g = 0; g < gwg; g++) {
          unsigned long long a = ~0ull, b = 0, ready = 0, done_min = ~0ull, done_max = 0; int edges = 0; unsigned long long hw = 0;
          for (int w = 0; w < NWG; w++) {
            const unsigned long long* t = &h[((size_t)g * NWG + w) * 8];
            if (!t[3]) continue;
            a = std::min(a, t[0]); b = std::max(b, t[3]); ready = std::max(ready, t[1]); hw = t[4]; edges = (int)t[5];
            if (t[2]) { done_min = std::min(done_min, t[2]); done_max = std::max(done_max, t[2]); }
          }
          if (!b) continue;
          const unsigned key = (unsigned)(((hw >> 32) & 0xf) << 10 | ((hw >> 13) & 0x7) << 7 | ((hw >> 8) & 0xf) << 3 | 0) & 16383u;   // xcc | se | cu
          CuAcc& c = cu[key];
          c.busy += (double)(b - a); c.first = std::min(c.first, (double)a); c.last = std::max(c.last, (double)b); c.wgs++; c.edges += edges;
          if (edges > 0) { c.live++; nlive++; live_span += (double)(b - a); if (ready) stage += (double)(ready - a); if (done_max) { item += (double)(done_max - a); tail += (double)(done_max - done_min); } }
          else { nempty++; empty_span += (double)(b - a); }
        }
        int ncu = 0; double bsum = 0, bmax = 0, emax = 0, esum = 0, lmax = 0;
        for (auto& c : cu) if (c.wgs) { ncu++; bsum += c.busy; bmax = std::max(bmax, c.busy); esum += c.edges; emax = std::max(emax, (double)c.edges); lmax = std::max(lmax, (double)c.live); }
        fprintf(stderr, "[corr group trace] %u workgroups (%ld with an item, %ld without), kernel span %.1f us; compute units seen %d: busy mean %.1f max %.1f us, item edges mean %.1f max %.0f, "
                "items max %.0f; per item: span %.1f us (region ready after %.1f, last wave leaves the item after %.1f, first-to-last wave %.1f); workgroup without item: %.2f us\n",
                gwg, nlive, nempty, (double)(t1 - t0) * 0.01, ncu, ncu ? bsum / ncu * 0.01 : 0.0, bmax * 0.01, ncu ? esum / ncu : 0.0, emax, lmax,
                nlive ? live_span / nlive * 0.01 : 0.0, nlive ? stage / nlive * 0.01 : 0.0, nlive ? item / nlive * 0.01 : 0.0, nlive ? tail / nlive * 0.01 : 0.0,
                nempty ? empty_span / nempty * 0.01 : 0.0);
        (void)hipFree(gtrace);
      }
      return check_launch("devo_corr_forward_pyramid2 (dense-product kernel, group form)");
    }
  }
  unsigned long long* trace = nullptr;                                // debug switch: per-wave cycle stamps to stderr
  const bool do_trace = getenv("DEVO_CORR_TRACE") != nullptr;
  if (do_trace) { (void)hipMalloc(&trace, (size_t)BE * 64); (void)hipMemset(trace, 0, (size_t)BE * 64); }
  const unsigned nwg = DEVO_MM_EPW == 1 ? (unsigned)BE : (unsigned)(((BE + DEVO_MM_EPW - 1) / DEVO_MM_EPW + 7) / 8 * 8);   // whole groups of 8 (one per XCD)
  // DEVO_CORR_LDS_PAD=<bytes> (tuning switch, round 6's bounded attempt on the stress size): unused dynamic LDS per one-wave workgroup, i.e. a
  // cap on the edges resident per compute unit (160 KB / (6.5 KB + pad)) and with it on the footprint an XCD streams through its L2 at a time
  static const int lds_pad = [] { const char* e = getenv("DEVO_CORR_LDS_PAD"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 60 * 1024 ? v : 0; }();
  hipLaunchKernelGGL(fn, dim3(nwg), dim3(64 * DEVO_MM_EPW), (size_t)lds_pad, st, (const MT*)fmap1_t, lv0, lv1, nlev, coords, ii, jj, (MT*)out, (int)BE, E, Np, n2, C,
                     oes, ols, R, order, 0, trace, exp1, MmGroupArgs{nullptr, 0, 0, 0});
  if (do_trace) {
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)BE * 8);
    (void)hipMemcpy(h.data(), trace, (size_t)BE * 64, hipMemcpyDeviceToHost);
    double ph[4] = {0, 0, 0, 0}, sum = 0, mx = 0, tiles = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    long cnt = 0;
    for (long long i = 0; i < BE; i++) {
      const unsigned long long* t = &h[(size_t)i * 8];
      if (!t[4]) continue;
      for (int q = 0; q < 4; q++) ph[q] += (double)(t[q + 1] - t[q]);
      const double d = (double)(t[4] - t[0]);
      sum += d; if (d > mx) mx = d;
      tiles += (double)(t[5] % 1000 + t[5] / 1000);
      if (t[0] < tmin) tmin = t[0];
      if (t[4] > tmax) tmax = t[4];
      cnt++;
    }
    if (!cnt) cnt = 1;
    fprintf(stderr, "[corr mm trace] %ld waves, kernel span %.0f cycles; wave mean %.0f max %.0f cycles; phase means: plan slot + indices + coordinates %.0f, "
            "geometry + patch operand + first tiles %.0f, tile loop %.0f, epilogue %.0f; %.1f tiles per edge\n",
            cnt, (double)(tmax - tmin), sum / cnt, mx, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, tiles / cnt);
    (void)hipFree(trace);
  }
  return check_launch("devo_corr_forward_pyramid2 (dense-product kernel)");
}

template <typename T>
static int launch_corr_fwd(const void* fmap1, const void* fmap2, const float* coords, const int64_t* ii,
                           const int64_t* jj, void* out, int B, int E, int Np, int n2, int C, int H2, int W2,
                           const int64_t* f2s, int cblock, int64_t oes, int64_t ols, int64_t ooff, int R, const int* order,
                           float coord_div, const void* fmap1_t, const int* fmap2_exps, hipStream_t st) {
  const long long BE = (long long)B * E;
  CorrLevel lv;
  int err;
  if (staged_level<T>(fmap2, C, H2, W2, f2s, cblock, ooff, coord_div, &lv, &err, fmap2_exps)) {
    if constexpr (!std::is_same<T, double>::value)
      if (ols > 0 && mm_eligible<T>(lv, lv, fmap1_t, BE, Np, C))
        return launch_mm<T>(fmap1_t, lv, lv, 1, coords, ii, jj, out, BE, E, Np, n2, C, oes, ols, R, order, st);
    return launch_staged<T>(fmap1, lv, lv, 1, coords, ii, jj, out, BE, E, Np, n2, C, oes, ols, R, order, st);
  }
  if (err) return err;
  if (cblock == DEVO_CBLOCK_SPLIT8) { set_error("devo_corr_forward: split-blocked fmap2 not readable (fp32, C %% 32 == 0, C <= 128, aligned strides)"); return DEVO_ERR_UNSUPPORTED; }
  corr_note_path(CORR_PATH_GENERIC, BE, "fp64, a raw NCHW level or strides the blocked kernels do not read");
  dim3 grid((unsigned)BE), block(NT);
  hipLaunchKernelGGL(corr_fwd_generic_kernel<T>, grid, block, 0, st, (const T*)fmap1, (const T*)fmap2, coords, ii,
                     jj, (T*)out, E, Np, n2, C, H2, W2, f2s[0], f2s[1], f2s[2], f2s[3], f2s[4], oes, ols, ooff, R, coord_div);
  return check_launch("devo_corr_forward");
}

extern "C" {

size_t devo_corr_patch_operand_bytes(int n_patches, int C, int dtype) {
  if (n_patches < 0 || C <= 0 || C % 8 != 0) return 0;
  if (dtype == DEVO_F16) return (size_t)n_patches * C * PP * 2;
  if (dtype == DEVO_F32) return (size_t)n_patches * C * PP * 4 + (size_t)n_patches * 4;      // split records | one scale exponent per patch
  return 0;
}

int devo_corr_patch_transpose(const void* fmap1, void* fmap1_t, int n_patches, int C, int dtype, devo_stream_t stream) {
  return devo_corr_patch_transpose_range(fmap1, fmap1_t, n_patches, 0, n_patches, C, dtype, stream);
}

// patches [first, first + count) of an operand of n_patches: DEVO rewrites ONE frame's patches (gmap_[slot] = ..., devo.py:524) per frame
int devo_corr_patch_transpose_range(const void* fmap1, void* fmap1_t, int n_patches, int first, int count, int C, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(n_patches >= 0 && C > 0 && C % 8 == 0, "devo_corr_patch_transpose: bad sizes (C must be a multiple of 8)");
  DEVO_REQUIRE(first >= 0 && count >= 0 && first + count <= n_patches, "devo_corr_patch_transpose_range: patches [%d, %d) of %d", first, first + count, n_patches);
  DEVO_REQUIRE(dtype == DEVO_F32 || dtype == DEVO_F16, "devo_corr_patch_transpose: fp32 / fp16 only");
  if (count == 0) return DEVO_OK;
  DEVO_REQUIRE(fmap1 && fmap1_t, "devo_corr_patch_transpose: null tensor");
  const size_t lds = (size_t)C * PP * (dtype == DEVO_F32 ? 4 : 2);
  DEVO_REQUIRE(lds <= 48 * 1024, "devo_corr_patch_transpose: C = %d too large", C);
  const size_t per = (size_t)C * PP;                                  // elements per patch, source and operand alike
  if (dtype == DEVO_F32)
    hipLaunchKernelGGL(corr_patch_transpose_kernel<float>, dim3((unsigned)count), dim3(256), lds, (hipStream_t)stream, (const float*)fmap1 + per * first,
                       (float*)fmap1_t + per * first, reinterpret_cast<int*>(static_cast<char*>(fmap1_t) + (size_t)n_patches * per * 4) + first, count, C);
  else
    hipLaunchKernelGGL(corr_patch_transpose_kernel<__half>, dim3((unsigned)count), dim3(256), lds, (hipStream_t)stream, (const __half*)fmap1 + per * first,
                       (__half*)fmap1_t + per * first, (int*)nullptr, count, C);
  return check_launch("devo_corr_patch_transpose");
}

int devo_corr_forward(const void* fmap1, const void* fmap2, const float* coords, const int64_t* ii,
                      const int64_t* jj, void* out, int B, int E, int Np, int n2, int C, int P, int H2, int W2,
                      const int64_t* f2s, int cblock, int64_t out_estride, int64_t out_lstride, int64_t out_offset,
                      int radius, int dtype, const int* order, float coord_div, const void* fmap1_t, const int* fmap2_exps, devo_stream_t stream) {
  DEVO_REQUIRE(P == 3, "devo_corr_forward: patch size P must be 3 (got %d)", P);
  DEVO_REQUIRE(coord_div > 0.0f, "devo_corr_forward: coord_div must be positive");
  DEVO_REQUIRE(radius >= 0 && 2 * radius + 2 <= MAXD, "devo_corr_forward: radius %d unsupported (max 5)", radius);
  DEVO_REQUIRE(B >= 0 && E >= 0 && C > 0 && H2 > 0 && W2 > 0, "devo_corr_forward: bad sizes");
  DEVO_REQUIRE(f2s != nullptr, "devo_corr_forward: fmap2 strides missing");
  if ((long long)B * E == 0) return DEVO_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case DEVO_F32: return launch_corr_fwd<float>(fmap1, fmap2, coords, ii, jj, out, B, E, Np, n2, C, H2, W2, f2s, cblock, out_estride, out_lstride, out_offset, radius, order, coord_div, fmap1_t, fmap2_exps, st);
    case DEVO_F16: return launch_corr_fwd<__half>(fmap1, fmap2, coords, ii, jj, out, B, E, Np, n2, C, H2, W2, f2s, cblock, out_estride, out_lstride, out_offset, radius, order, coord_div, fmap1_t, fmap2_exps, st);
    case DEVO_F64: return launch_corr_fwd<double>(fmap1, fmap2, coords, ii, jj, out, B, E, Np, n2, C, H2, W2, f2s, cblock, out_estride, out_lstride, out_offset, radius, order, coord_div, fmap1_t, fmap2_exps, st);
  }
  set_error("devo_corr_forward: unknown dtype %d", dtype);
  return DEVO_ERR_UNSUPPORTED;
}

int devo_corr_forward_pyramid2(const void* fmap1, const void* fmap2_l0, const void* fmap2_l1, const float* coords,
                               const int64_t* ii, const int64_t* jj, void* out, int B, int E, int Np, int n2, int C, int P,
                               const int* hw /* host: H0, W0, H1, W1 */, const int64_t* f2s /* host: 5 + 5 */,
                               const int* cblock /* host, 2 */, int64_t out_estride, int64_t out_lstride,
                               const int64_t* out_offset /* host, 2 */, int radius, int dtype, const int* order,
                               const float* coord_div /* host, 2 */, const void* fmap1_t, const int* fmap2_exps_l0, const int* fmap2_exps_l1,
                               int order_kind, devo_stream_t stream) {
  DEVO_REQUIRE(P == 3, "devo_corr_forward_pyramid2: patch size P must be 3 (got %d)", P);
  DEVO_REQUIRE(radius >= 0 && 2 * radius + 2 <= MAXD, "devo_corr_forward_pyramid2: radius %d unsupported (max 5)", radius);
  DEVO_REQUIRE(hw && f2s && cblock && out_offset && coord_div, "devo_corr_forward_pyramid2: missing level description");
  DEVO_REQUIRE(B >= 0 && E >= 0 && C > 0 && hw[0] > 0 && hw[1] > 0 && hw[2] > 0 && hw[3] > 0, "devo_corr_forward_pyramid2: bad sizes");
  DEVO_REQUIRE(coord_div[0] > 0.0f && coord_div[1] > 0.0f, "devo_corr_forward_pyramid2: coord_div must be positive");
  const long long BE = (long long)B * E;
  if (BE == 0) return DEVO_OK;
  CorrLevel l0, l1;
  int e0 = DEVO_OK, e1 = DEVO_OK;
  bool ok = false;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DEVO_F32) {
    ok = staged_level<float>(fmap2_l0, C, hw[0], hw[1], f2s, cblock[0], out_offset[0], coord_div[0], &l0, &e0, fmap2_exps_l0) &&
         staged_level<float>(fmap2_l1, C, hw[2], hw[3], f2s + 5, cblock[1], out_offset[1], coord_div[1], &l1, &e1, fmap2_exps_l1);
    if (e0 == DEVO_ERR_ARG || e1 == DEVO_ERR_ARG) return DEVO_ERR_ARG;
    if (ok && out_lstride > 0 && mm_eligible<float>(l0, l1, fmap1_t, BE, Np, C))
      return launch_mm<float>(fmap1_t, l0, l1, 2, coords, ii, jj, out, BE, E, Np, n2, C, out_estride, out_lstride, radius, order, st, order_kind);
    if (ok) return launch_staged<float>(fmap1, l0, l1, 2, coords, ii, jj, out, BE, E, Np, n2, C, out_estride, out_lstride, radius, order, st);
  } else if (dtype == DEVO_F16) {
    ok = staged_level<__half>(fmap2_l0, C, hw[0], hw[1], f2s, cblock[0], out_offset[0], coord_div[0], &l0, &e0) &&
         staged_level<__half>(fmap2_l1, C, hw[2], hw[3], f2s + 5, cblock[1], out_offset[1], coord_div[1], &l1, &e1);
    if (ok && out_lstride > 0 && mm_eligible<__half>(l0, l1, fmap1_t, BE, Np, C))
      return launch_mm<__half>(fmap1_t, l0, l1, 2, coords, ii, jj, out, BE, E, Np, n2, C, out_estride, out_lstride, radius, order, st, order_kind);
    if (ok) return launch_staged<__half>(fmap1, l0, l1, 2, coords, ii, jj, out, BE, E, Np, n2, C, out_estride, out_lstride, radius, order, st);
  }
  // not both levels readable by the staged kernel: the caller issues one devo_corr_forward per level instead
  set_error("devo_corr_forward_pyramid2: levels not eligible for the fused launch (layout / dtype)");
  return DEVO_ERR_UNSUPPORTED;
}

int devo_corr_pyramid_split(const void* fmap2, const int64_t* f2s, int cblock, int F, int C, int H, int W, void* dst, int64_t dst_fstride,
                            int* exps, devo_stream_t stream) {
  return devo_corr_pyramid_split_frames(fmap2, f2s, cblock, F, C, H, W, dst, dst_fstride, exps, exps ? exps + F : nullptr, stream);
}

// the same with the F ints of scratch given separately: frames [k, k + F) of a ring keep their exponents at exps_ring + k while their
// neighbours' stay untouched (fmap1_[:, slot] = ..., devo.py:526: one slot per frame)
int devo_corr_pyramid_split_frames(const void* fmap2, const int64_t* f2s, int cblock, int F, int C, int H, int W, void* dst, int64_t dst_fstride,
                                   int* exps, int* scratch, devo_stream_t stream) {
  DEVO_REQUIRE(F >= 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0, "devo_corr_pyramid_split: bad sizes (C must be a multiple of 8)");
  DEVO_REQUIRE(f2s != nullptr && cblock >= 0 && (cblock <= 1 || C % cblock == 0), "devo_corr_pyramid_split: bad layout description");
  if (F == 0) return DEVO_OK;
  DEVO_REQUIRE(fmap2 && dst && exps && scratch, "devo_corr_pyramid_split: null tensor");
  DEVO_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && dst_fstride % 4 == 0 && dst_fstride >= (int64_t)C * H * W, "devo_corr_pyramid_split: dst must be 16-byte aligned frames of C*H*W elements");
  hipStream_t st = (hipStream_t)stream;
  unsigned* maxbits = reinterpret_cast<unsigned*>(scratch);
  if (hipMemsetAsync(maxbits, 0, (size_t)F * 4, st) != hipSuccess) { (void)hipGetLastError(); set_error("devo_corr_pyramid_split: memset failed"); return DEVO_ERR_LAUNCH; }
  const SplitSrc S{f2s[0], f2s[1], f2s[2], f2s[3], cblock};
  const long long per = (long long)(C / 8) * H * W;
  const dim3 grid((unsigned)std::min<long long>((per + 255) / 256, 2048), (unsigned)F);
  // the max pass: ~8 records per thread, at most 256 workgroups per frame and ~2 048 in all (each ends in one atomic on the frames' shared line)
  const long long want = std::max<long long>(1, std::min<long long>((per + 2047) / 2048, std::max<long long>(8, 2048 / F)));
  const dim3 mgrid((unsigned)std::min<long long>(want, 256), (unsigned)F);
  hipLaunchKernelGGL(corr_split_max_kernel, mgrid, dim3(256), 0, st, (const float*)fmap2, S, C, H, W, maxbits);
  hipLaunchKernelGGL(corr_split_kernel, grid, dim3(256), 0, st, (const float*)fmap2, S, C, H, W, (const unsigned*)maxbits, (float*)dst, dst_fstride, exps);
  return check_launch("devo_corr_pyramid_split");
}

int devo_pyramid_build(const void* fmap, void* l0, void* l1, int F, int C, int H, int W, int64_t fmap_fstride,
                       int64_t l0_fstride, int64_t l1_fstride, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(F >= 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0, "devo_pyramid_build: bad sizes (C must be a multiple of 8)");
  DEVO_REQUIRE(fmap && l0, "devo_pyramid_build: null tensor");
  if (F == 0) return DEVO_OK;
  const dim3 grid((unsigned)((long long)F * (C / 8) * ((H + 3) / 4)), (unsigned)((W + PYR_WC - 1) / PYR_WC)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DEVO_F32)
    hipLaunchKernelGGL(pyramid_blocked_kernel<float>, grid, block, 0, st, (const float*)fmap, (float*)l0, (float*)l1, C, H, W, fmap_fstride, l0_fstride, l1_fstride);
  else if (dtype == DEVO_F16)
    hipLaunchKernelGGL(pyramid_blocked_kernel<__half>, grid, block, 0, st, (const __half*)fmap, (__half*)l0, (__half*)l1, C, H, W, fmap_fstride, l0_fstride, l1_fstride);
  else { set_error("devo_pyramid_build: fp32 / fp16 only"); return DEVO_ERR_UNSUPPORTED; }
  return check_launch("devo_pyramid_build");
}

int devo_corr_order(const float* coords, const int64_t* jj, int* order, int B, int E, int n2, int P, int H2,
                    float coord_scale, int radius, int W2, int l1, devo_stream_t stream) {
  DEVO_REQUIRE(P == 3, "devo_corr_order: patch size P must be 3 (got %d)", P);
  DEVO_REQUIRE(B >= 0 && E >= 0 && n2 > 0 && H2 > 0 && coord_scale > 0.0f, "devo_corr_order: bad sizes");
  const long long BE = (long long)B * E;
  if (BE == 0) return DEVO_OK;
  const CorrPlanGeom pg = corr_plan_geom(B, n2, H2);
  const int nb = corr_plan_pack(pg);
  DEVO_REQUIRE(pg.nb > 0 && BE < (1LL << 30), "devo_corr_order: too many frames (%d x %d)", B, n2);
  DEVO_REQUIRE(radius >= 0 && 2 * radius + 2 <= MAXD, "devo_corr_order: radius %d unsupported (max 5)", radius);
  DEVO_REQUIRE(l1 == 0 || (l1 >= 2 && W2 > 0), "devo_corr_order: a group plan needs the level's width and an integer level ratio >= 2");
  int* bins = order + BE + 1;                                 // scratch half of the plan buffer
  const long long nbins = l1 >= 2 ? corr_grp_nbins(B, n2, H2, W2, l1) : corr_plan_nbins(B, n2, pg);
  if (l1 >= 2 && (nbins == 0 || radius != 3)) {
    set_error("devo_corr_order: no group plan for this geometry (radius 3 only, at most %d groups: %d frames of %d x %d)", CORR_ORDER_MAXBINS, B * n2, H2, W2);
    return DEVO_ERR_UNSUPPORTED;
  }
  const CorrPlanMode pm{W2, l1, 16 * corr_region_tmax(radius), (int)nbins - 1};
  if (coords != nullptr)                                      // NULL: devo_transform has already written the bins
    hipLaunchKernelGGL(corr_bin_kernel, dim3((unsigned)((BE + BIN_THREADS - 1) / BIN_THREADS)), dim3(BIN_THREADS), 0,
                       (hipStream_t)stream, coords, jj, (int)BE, E, n2, H2, coord_scale, nb, 2 * radius + 2,
                       radius <= 3 ? 1 : 3, pm, bins);
  typedef void (*order_fn_t)(const int*, int, int, int*, int);
  const long long per_thread = (BE + ORDER_THREADS - 1) / ORDER_THREADS;
  order_fn_t order_fn = per_thread <= 8 ? corr_order_kernel<8> : per_thread <= 16 ? corr_order_kernel<16> :
                        per_thread <= 24 ? corr_order_kernel<24> : per_thread <= 32 ? corr_order_kernel<32> :
                        per_thread <= 48 ? corr_order_kernel<48> : per_thread <= 64 ? corr_order_kernel<64> : corr_order_kernel<0>;   // (DEVO's steady state: 45 312 edges = 45 per thread)
  hipLaunchKernelGGL(order_fn, dim3((unsigned)corr_order_workgroups(BE, nbins)), dim3(ORDER_THREADS), 0, (hipStream_t)stream, bins, (int)BE,
                     (int)nbins, order, l1 >= 2 ? 1 : 0);
  return check_launch("devo_corr_order");
}

static thread_local int g_corr_bwd_path = -1;                      // what the last devo_corr_backward of this thread launched
int devo_corr_backward_last_path(void) { return g_corr_bwd_path; }
int devo_corr_forward_last_path(void) { return g_corr_fwd_path; }

size_t devo_corr_backward_workspace_bytes(int B, int E, int Np, int n2, int C, int radius, int channels_last) {
  if (B <= 0 || E <= 0 || Np <= 0 || n2 <= 0 || C <= 0 || radius < 0) return 0;
  if (!channels_last) return 0;                                    // every other layout takes the one-kernel atomic path, which needs none
  const size_t BE = (size_t)B * E, D = 2 * (size_t)radius + 2, frames = (size_t)B * n2;
  const size_t gs = BE * PP * D * D * 4, product = ((gs + 15) & ~(size_t)15) + frames * BE * PP * 16 + (size_t)B * Np * C * PP * 4 + frames * 4;
  const size_t seg = gs + BE * sizeof(BwdMeta) + frames * BE * 4 + frames * 4;
  static const bool want_seg = getenv("DEVO_CORR_BWD_SEG") != nullptr, want_atomic = getenv("DEVO_CORR_BWD_ATOMIC") != nullptr;
  if (want_seg) return seg <= ((size_t)512 << 20) ? seg : 0;       // (the opt-in segment-reduced path is sized on its own)
  if (want_atomic || C % 128 != 0) return 0;
  return product <= ((size_t)1024 << 20) ? product : 0;            // (larger problems take the atomic path)
}

int devo_corr_backward(const void* fmap1, const void* fmap2, const float* coords, const int64_t* ii,
                       const int64_t* jj, const float* grad, void* fmap1_grad, void* fmap2_grad, int B, int E, int Np,
                       int n2, int C, int P, int H2, int W2, const int64_t* f2s, int64_t f2_numel_span, int radius,
                       int dtype, void* ws, size_t ws_bytes, devo_stream_t stream) {
  DEVO_REQUIRE(P == 3, "devo_corr_backward: patch size P must be 3 (got %d)", P);
  DEVO_REQUIRE(radius >= 0 && 2 * radius + 2 <= MAXD, "devo_corr_backward: radius %d unsupported (max 5)", radius);
  if (dtype != DEVO_F32) { set_error("devo_corr_backward: fp32 only (the reference's grad accessor is float)"); return DEVO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  const long long BE = (long long)B * E;
  DEVO_REQUIRE((long long)(H2 - 1) * f2s[3] + (long long)(W2 - 1) * f2s[4] < (1LL << 31) && f2s[3] >= 0 && f2s[4] >= 0,
               "devo_corr_backward: a frame of fmap2 must span less than 2^31 elements (row stride %lld)", (long long)f2s[3]);
  // Segment-reduced path (opt-in, DEVO_CORR_BWD_SEG=1): channels-last fmap2 (the C channels of a pixel contiguous, pixels and rows
  // C / W C apart), 16-channel slabs, the scratch (window gradients + geometry + per-frame edge lists) within 512 MB.  Measured
  // slower than the one-kernel atomic path at DEVO's sizes (profiles/README.md, r02y), which therefore stays the default.
  static const bool force_atomic = getenv("DEVO_CORR_BWD_SEG") == nullptr;
  const int D = 2 * radius + 2;
  int BH = 8;
  while (BH > 1 && (size_t)BH * W2 * BWD_CS * 4 + 4 * 64 * 12 * 4 > 54 * 1024) BH >>= 1;      // (+ 8.3 KB static: 64 KB per workgroup)
  const long long frames = (long long)B * n2;
  const size_t gs_bytes = (size_t)BE * PP * D * D * 4, meta_bytes = (size_t)BE * sizeof(BwdMeta), list_bytes = (size_t)frames * (size_t)BE * 4;
  const size_t tile_lds = (size_t)BH * W2 * BWD_CS * 4 + 4 * 64 * 12 * 4;
  const bool seg = !force_atomic && BE > 0 && f2s[2] == 1 && f2s[4] == C && f2s[3] == (int64_t)W2 * C && f2s[1] >= (int64_t)H2 * W2 * C &&
                   C % BWD_CS == 0 && tile_lds <= 54 * 1024 && gs_bytes + meta_bytes + list_bytes <= (512ull << 20) && BE < (1LL << 31) &&
                   frames <= 65535 && (H2 + BH - 1) / BH <= 65535 && (reinterpret_cast<uintptr_t>(fmap2_grad) & 15) == 0 &&
                   ws != nullptr && ws_bytes >= gs_bytes + meta_bytes + list_bytes + (size_t)frames * 4 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0;
  // Product form (default where it applies; DEVO_CORR_BWD_ATOMIC=1: the one-kernel atomic path): channels-last fmap2 with C % 128 == 0 —
  // d_fmap1 per edge and d_fmap2 per frame tile as matrix-core products, no atomics on d_fmap2 (corr_bwd_mfma.h).
  static const bool no_product = getenv("DEVO_CORR_BWD_ATOMIC") != nullptr || getenv("DEVO_CORR_BWD_SEG") != nullptr;
  const size_t f1t_bytes = (size_t)B * Np * C * PP * 4, pair_bytes = (size_t)frames * (size_t)BE * PP * 16;      // (a frame's window list can hold every edge)
  const size_t product_ws = ((gs_bytes + 15) & ~(size_t)15) + pair_bytes + f1t_bytes + (size_t)frames * 4;
  const bool product = !no_product && BE > 0 && ws != nullptr && ws_bytes >= product_ws && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 && f2s[2] == 1 && f2s[4] == C && f2s[3] == (int64_t)W2 * C && f2s[1] >= (int64_t)H2 * W2 * C &&
                       C % 128 == 0 && BE * PP * D * D < (1LL << 31) && frames <= 65535 && (H2 + 1) / 2 <= 65535 &&
                       W2 <= 32767 && H2 <= 32767 && (long long)H2 * W2 < (1LL << 22) &&      // (the edge kernel packs ry << 16 | rx and splits q with an fp32 reciprocal)
                       gs_bytes + pair_bytes + f1t_bytes <= (1024ull << 20) &&
                       (reinterpret_cast<uintptr_t>(fmap2_grad) & 15) == 0 && (reinterpret_cast<uintptr_t>(fmap2) & 15) == 0 &&
                       f2s[0] % 4 == 0 && f2s[1] % 4 == 0;
  if ((!product && hipMemsetAsync(fmap1_grad, 0, sizeof(float) * (size_t)B * Np * C * PP, st) != hipSuccess) ||
      (!seg && !product && hipMemsetAsync(fmap2_grad, 0, sizeof(float) * (size_t)f2_numel_span, st) != hipSuccess)) {
    set_error("devo_corr_backward: memset failed");
    return DEVO_ERR_LAUNCH;
  }
  g_corr_bwd_path = product ? DEVO_CORR_BWD_PRODUCT : seg ? DEVO_CORR_BWD_SEGMENTS : DEVO_CORR_BWD_ATOMIC;
  if (BE == 0) return DEVO_OK;
  unsigned long long* btrace = nullptr;                               // debug switch: phase cycles of the per-edge kernel to stderr
  static const bool do_btrace = getenv("DEVO_CORR_BWD_TRACE") != nullptr;
  if (do_btrace) { (void)hipMalloc(&btrace, (size_t)BE * 64); (void)hipMemset(btrace, 0, (size_t)BE * 64); }
  auto dump_trace = [&]() {
    if (!do_btrace) return;
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> hv((size_t)BE * 8);
    (void)hipMemcpy(hv.data(), btrace, (size_t)BE * 64, hipMemcpyDeviceToHost);
    double h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long i = 0; i < BE; i++) for (int q = 0; q < 8; q++) h[q] += (double)hv[(size_t)i * 8 + q];
    fprintf(stderr, "[corr bwd trace] per-edge kernel, mean cycles per workgroup: coordinates + gradient block + patch %.0f | window gradients %.0f | rows %.0f | patch atomics + hand-over %.0f\n",
            (double)h[0] / BE, (double)h[1] / BE, (double)h[2] / BE, (double)h[3] / BE);
    (void)hipFree(btrace);
  };
  if (product) {
    char* scratch = reinterpret_cast<char*>(ws);
    const size_t pair_off = (gs_bytes + 15) & ~(size_t)15, f1t_off = pair_off + pair_bytes, cur_off = f1t_off + f1t_bytes;
    float* gs = reinterpret_cast<float*>(scratch);
    BwdPair* pairs = reinterpret_cast<BwdPair*>(scratch + pair_off);
    float* f1t = reinterpret_cast<float*>(scratch + f1t_off);
    int* cursors = reinterpret_cast<int*>(scratch + cur_off);
    hipLaunchKernelGGL(corr_bwd_patch_t_kernel, dim3((unsigned)(B * Np)), dim3(256), (size_t)C * PP * 4, st, (const float*)fmap1, f1t, (float*)fmap1_grad, cursors,
                       (int)frames, B * Np, C);
    hipLaunchKernelGGL((radius <= 3 ? corr_bwd_edge_kernel<3> : corr_bwd_edge_kernel<5>), dim3((unsigned)BE), dim3(64 * BWE_KS), 0, st,
                       (const float*)fmap2, coords, ii, jj, grad, (float*)fmap1_grad, BE, E, Np, n2, C, H2, W2, f2s[0], f2s[1], radius, gs, pairs, cursors,
                       (int)BE);
    const int tiles_x = (W2 + 7) / 8;
    // 8 x 8 tiles (4 waves) where that gives every SIMD a few waves, 8 x 2 tiles (1 wave) for small levels (DEVO's level 1)
    const bool small_level = (long long)tiles_x * (C / 128) * ((H2 + 7) / 8) * frames < 2048;
    if (small_level)
      hipLaunchKernelGGL(corr_bwd_frame_kernel<1>, dim3((unsigned)(tiles_x * (C / 128)), (unsigned)((H2 + 1) / 2), (unsigned)frames), dim3(64), 0, st,
                         (const float*)f1t, (const float*)gs, (const BwdPair*)pairs, (const int*)cursors, (float*)fmap2_grad, n2, C, H2, W2, f2s[0], f2s[1], D,
                         (int)BE, tiles_x);
    else
      hipLaunchKernelGGL(corr_bwd_frame_kernel<4>, dim3((unsigned)(tiles_x * (C / 128)), (unsigned)((H2 + 7) / 8), (unsigned)frames), dim3(256), 0, st,
                         (const float*)f1t, (const float*)gs, (const BwdPair*)pairs, (const int*)cursors, (float*)fmap2_grad, n2, C, H2, W2, f2s[0], f2s[1], D,
                         (int)BE, tiles_x);
    const int rc = check_launch("devo_corr_backward");
    if (do_btrace) (void)hipFree(btrace);
    return rc;
  }
  if (seg) {
    char* scratch = reinterpret_cast<char*>(ws);
    const size_t cur_off = gs_bytes + meta_bytes + list_bytes;
    float* gs = reinterpret_cast<float*>(scratch);
    BwdMeta* meta = reinterpret_cast<BwdMeta*>(scratch + gs_bytes);
    int* lists = reinterpret_cast<int*>(scratch + gs_bytes + meta_bytes);
    int* cursors = reinterpret_cast<int*>(scratch + cur_off);
    (void)hipMemsetAsync(cursors, 0, (size_t)frames * 4, st);
    hipLaunchKernelGGL((radius <= 3 ? corr_bwd_kernel<true, 3> : corr_bwd_kernel<true, 5>), dim3((unsigned)BE), dim3(NT), 0, st, (const float*)fmap1, (const float*)fmap2, coords, ii, jj,
                       grad, (float*)fmap1_grad, (float*)fmap2_grad, E, Np, n2, C, H2, W2, f2s[0], f2s[1], f2s[2], f2s[3], f2s[4], radius,
                       gs, meta, lists, cursors, (int)BE, btrace);
    hipLaunchKernelGGL(corr_bwd_tile_kernel, dim3((unsigned)(C / BWD_CS), (unsigned)((H2 + BH - 1) / BH), (unsigned)frames), dim3(BWD_THREADS),
                       tile_lds, st, (const float*)fmap1, gs, meta, lists, cursors, (float*)fmap2_grad, n2, C, H2, W2, f2s[0], f2s[1], f2s[3],
                       f2s[4], D, BH, (int)BE);
    dump_trace();
    return check_launch("devo_corr_backward");
  }
  hipLaunchKernelGGL((radius <= 3 ? corr_bwd_kernel<false, 3> : corr_bwd_kernel<false, 5>), dim3((unsigned)BE), dim3(NT), 0, st, (const float*)fmap1,
                     (const float*)fmap2, coords, ii, jj, grad, (float*)fmap1_grad, (float*)fmap2_grad, E, Np, n2, C, H2,
                     W2, f2s[0], f2s[1], f2s[2], f2s[3], f2s[4], radius, (float*)nullptr, (BwdMeta*)nullptr, (int*)nullptr, (int*)nullptr, 0, btrace);
  dump_trace();
  return check_launch("devo_corr_backward");
}

int devo_patchify_forward(const void* net, const float* coords, void* out, int B, int M, int C, int H, int W,
                          const int64_t* ns, int radius, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(radius >= 0 && ns != nullptr, "devo_patchify_forward: bad arguments");
  const int D = 2 * radius + 2;
  int64_t total = (int64_t)B * M * C * D * D;
  if (total == 0) return DEVO_OK;
  hipStream_t st = (hipStream_t)stream;
  int blocks = blocks_for(total, 256, 8192);
#define PF(T) hipLaunchKernelGGL(patchify_fwd_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)net, coords, (T*)out, M, C, H, W, ns[0], ns[1], ns[2], ns[3], radius, total)
  if (dtype == DEVO_F32) PF(float); else if (dtype == DEVO_F16) PF(__half); else if (dtype == DEVO_F64) PF(double);
  else { set_error("devo_patchify_forward: unknown dtype %d", dtype); return DEVO_ERR_UNSUPPORTED; }
#undef PF
  return check_launch("devo_patchify_forward");
}

int devo_patchify_backward(const float* coords, const void* grad, void* net_grad, int B, int M, int C, int H, int W,
                           const int64_t* gs, int radius, int dtype, devo_stream_t stream) {
  const int D = 2 * radius + 2;
  int64_t total = (int64_t)B * M * C * D * D;
  hipStream_t st = (hipStream_t)stream;
  size_t esz = dtype == DEVO_F64 ? 8 : 4;
  if (dtype != DEVO_F32 && dtype != DEVO_F64) { set_error("devo_patchify_backward: F32/F64 only"); return DEVO_ERR_UNSUPPORTED; }
  // gs: element strides of net_grad (NULL: contiguous [B, C, H, W]); a dense permutation (channels-last: what the encoders' convolutions
  // hand over and take back) — the B C H W elements behind net_grad are zeroed
  const int64_t cs[4] = {(int64_t)C * H * W, (int64_t)H * W, W, 1};
  if (!gs) gs = cs;
  int64_t span = 1;
  const int dims[4] = {B, C, H, W};
  for (int d = 0; d < 4; d++) { if (gs[d] < 0) { set_error("devo_patchify_backward: negative stride"); return DEVO_ERR_ARG; } span += (int64_t)(dims[d] - 1) * gs[d]; }
  if (B * (int64_t)C * H * W > 0 && span != (int64_t)B * C * H * W) { set_error("devo_patchify_backward: net_grad must be a dense (permuted) tensor"); return DEVO_ERR_ARG; }
  if (hipMemsetAsync(net_grad, 0, esz * (size_t)B * C * H * W, st) != hipSuccess) { set_error("devo_patchify_backward: memset failed"); return DEVO_ERR_LAUNCH; }
  if (total == 0) return DEVO_OK;
  int blocks = blocks_for(total, 256, 8192);
  if (dtype == DEVO_F32) hipLaunchKernelGGL(patchify_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, coords, (const float*)grad, (float*)net_grad, M, C, H, W, radius, total, gs[0], gs[1], gs[2], gs[3]);
  else hipLaunchKernelGGL(patchify_bwd_kernel<double>, dim3(blocks), dim3(256), 0, st, coords, (const double*)grad, (double*)net_grad, M, C, H, W, radius, total, gs[0], gs[1], gs[2], gs[3]);
  return check_launch("devo_patchify_backward");
}

}  // extern "C"
