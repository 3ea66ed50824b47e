// altcorr lookup as ONE dense matrix product per edge and level (included by corr.hip inside namespace devo).
//
// The reference's lookup (correlation_kernel.cu:82-136) is, per edge, a contraction over the C channels of the 9 patch pixels with
// every position of the union bounding box of their windows:  raw[p][x] = sum_k fmap1[p][k] * fmap2[x][k].  corr_mfma.h runs it as
// 4 x 4 outer products (v_mfma_f32_4x4x1/4x4x4: 1 / 4 channels per instruction, one position per lane); here it is a 16 x 16 x 32
// product on the DENSE matrix pipe (v_mfma_f32_16x16x32_f16, 4x / 16x the multiply rate of the 4x4 forms):
//     M = 16 box positions (a TILE), N = 16 columns of which 9 are the patch pixels, K = 32 channels.
//   * A operand = the pyramid, straight from memory into registers: lane (i, kg) = (lane % 16, lane / 16) loads the 16 bytes that hold
//     the 8 channels 32 s + 8 kg .. + 7 (fp16) of position i of the tile — ONE buffer_load_dwordx4 per lane, tile and K step, and with
//     channel blocks of 32 halves (64-byte cells) the 64 lanes of an instruction read whole cache lines: every line of an edge's box
//     passes the L1's tag lookup once per edge instead of once per 16-byte piece (the per-position loads of corr_mfma.h touch each line
//     4 times: tools/ubench/l2_fill.hip measures 1 tag per cycle = 32 B/cycle/CU for that shape against >= 56 for whole lines).
//   * B operand = the patch, transposed once per version of fmap1 to [patch][pixel][channel] (devo_corr_patch_transpose; fp32: split
//     into fp16 hi | lo there, once): lane (n, kg) holds channels 32 s + 8 kg .. + 7 of pixel n (columns 9..15: zeros through the
//     buffer range check), C / 32 x 4 registers per edge.
//   * D = 16 positions x 16 columns: lane (n, rg) holds positions 4 rg .. 4 rg + 3 of column n = one ds_write_b128 into the level's
//     result area in LDS ([pixel][position], the layout corr_mfma.h's fused blend epilogue already reads).
// A box of 107 positions is 7 tiles = 28 loads + 28 MFMAs per level (corr_mfma.h, fp16: 33 loads, 200 MFMAs, 9 x 2 LDS stores per lane).
// fp32 storage: every value is split exactly into fp16 hi + lo (22 significant bits; x = hi + lo + eps, |eps| <= 2^-22 |x|) and the
// product is hi * hi' + lo * hi' + hi * lo' with fp32 accumulation: 3 dense MFMAs per K step instead of 32 x 3 fp32 4x4x1 MFMAs.
// Tiles are fetched RT - 1 tiles ahead of their products into a ring of register sets that runs on from level 0 into level 1.
// Boxes larger than the result area (patch pixels spread far apart: the plan's HEAVY class) walk the 9 windows one after the other,
// one column of the product each.
#pragma once

typedef _Float16 mm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 mm_h2 __attribute__((ext_vector_type(2)));
typedef float mm_f4 __attribute__((ext_vector_type(4)));

#ifndef DEVO_MM_WAVES
#define DEVO_MM_WAVES 4        // waves per SIMD, fp16 storage (128 registers)
#endif
#ifndef DEVO_MM_WAVES32
#define DEVO_MM_WAVES32 2      // fp32 storage (the hi / lo patch, twice the bytes in flight, the held outputs: up to 200 registers; the kernel's time does not
                               // depend on 2, 3 or 4 waves per SIMD: the texture addresser is the busy unit).  fp16, radius 4 - 5: one wave less
#endif
// Edges (= waves) per workgroup: EPW CONSECUTIVE plan slots — image neighbours — run on one CU at the same time, so that the lines their
// boxes share can meet in the CU's 32 KB L1 instead of being fetched from the L2 once per edge.  DEVO_MM_SYNC = 1 additionally keeps the
// waves of a workgroup on the same tile slot (one s_barrier per slot: ~60 cycles) — an L1 line lives about a thousand cycles.
#ifndef DEVO_MM_EPW
#define DEVO_MM_EPW 1
#endif
#ifndef DEVO_MM_RT16
#define DEVO_MM_RT16 4          // fp16 storage: tiles in the register ring (3 in flight ahead of the products)
#endif
#ifndef DEVO_MM_ALIGN4
#define DEVO_MM_ALIGN4 1        // fp16 storage: boxes start at a multiple of 4 positions and are a multiple of 4 wide: every quad of lanes reads inside ONE 128-byte line
#endif
#ifndef DEVO_MM_SYNC
#define DEVO_MM_SYNC 0
#endif

// fp32 operands are stored pre-split ("split" formats, written once per version of the tensor by devo_corr_patch_transpose /
// devo_corr_pyramid_split): a value x of a group with scale exponent e (per patch / per frame, chosen so that the group's largest
// magnitude lands in [2^13, 2^14)) is the fp16 pair  hi = rn(x 2^-e),  lo = rn(x 2^-e - hi):  x 2^-e = hi + lo to 2^-22 relative (a lo below
// fp16's normal range keeps 2^-25 absolute, i.e. 2^-39 of the group's largest magnitude).  The power-of-two scaling is exact, no input
// magnitude overflows fp16, and the kernel multiplies the blended sums by 2^(e_patch + e_frame) at the end (folded into the blend weights).
// (v_cvt_pk_f16_f32: two values per instruction; v_fma_mix: fp32 arithmetic on the fp16 hi.)
__device__ __forceinline__ void mm_split2(float a, float b, unsigned& h, unsigned& l) {
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
  // (mixlo keeps the destination's upper half, which mixhi then overwrites: no initialisation needed)
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
}
// 8 fp32 values, scaled by 2^-e -> the 32-byte split record (hi0..7 | lo0..7)
__device__ __forceinline__ void mm_split8_scaled(const float* x, int e, v4u32& hi, v4u32& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; j++) mm_split2(ldexpf(x[2 * j], -e), ldexpf(x[2 * j + 1], -e), h[j], l[j]);
  hi = v4u32{h[0], h[1], h[2], h[3]};
  lo = v4u32{l[0], l[1], l[2], l[3]};
}
// scale exponent of a group whose largest magnitude has the fp32 bit pattern `maxbits` (sign cleared): floor(log2(max)) - 13
// (zero / denormal groups: as if the largest magnitude were 2^-126; inf / nan: the largest finite exponent — the values propagate)
__host__ __device__ __forceinline__ int mm_scale_exp(unsigned maxbits) {
  const int ef = (int)((maxbits >> 23) & 0xffu);
  return (ef == 0 ? -126 : (ef == 255 ? 127 : ef - 127)) - 13;
}

// fmap1 [N][C][9] -> the patch (B) operand of corr_fwd_mm_kernel.  fp16: [N][9][C], 16 contiguous bytes per (pixel, 8 channels).
// fp32: [N][9][C / 8][hi0..7 | lo0..7] split records (32 bytes per pixel and 8 channels) with ONE scale exponent per patch, written to
// exps[n] (the int32 tail of the operand buffer).
template <typename T>
__global__ __launch_bounds__(256) void corr_patch_transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int* __restrict__ exps, int N, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pt_lds[];
  T* s = reinterpret_cast<T*>(pt_lds);
  const int n = blockIdx.x;
  if (n >= N) return;
  const T* in = src + (int64_t)n * C * PP;
  T* o = dst + (int64_t)n * C * PP;
  if constexpr (sizeof(T) == 2) {
    for (int i = threadIdx.x; i < C * PP; i += 256) s[i] = in[i];
    __syncthreads();
    for (int i = threadIdx.x; i < C * PP; i += 256) { const int p = i / C, c = i - p * C; o[i] = s[c * PP + p]; }
  } else {
    __shared__ unsigned s_max;
    if (threadIdx.x == 0) s_max = 0u;
    __syncthreads();
    unsigned mx = 0u;
    for (int i = threadIdx.x; i < C * PP; i += 256) { const T v = in[i]; s[i] = v; mx = max(mx, __float_as_uint((float)v) & 0x7fffffffu); }
    atomicMax(&s_max, mx);
    __syncthreads();
    const int e = mm_scale_exp(s_max);
    if (threadIdx.x == 0) exps[n] = e;
    for (int i = threadIdx.x; i < (C / 8) * PP; i += 256) {
      const int p = i / (C / 8), c8 = i - p * (C / 8);
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = (float)s[(8 * c8 + j) * PP + p];
      v4u32 hi, lo;
      mm_split8_scaled(x, e, hi, lo);
      reinterpret_cast<v4u32*>(o)[2 * i] = hi;
      reinterpret_cast<v4u32*>(o)[2 * i + 1] = lo;
    }
  }
}

// fp32 pyramid in any strided layout -> the split-blocked format [F][C / 8][H][W][hi0..7 | lo0..7] (the byte layout of a channel-blocked
// fp32 level with 8 channels per block: 32 bytes per pixel and block), one scale exponent per frame.  Pass 1: the frames' largest
// magnitudes (bit patterns, atomicMax) into maxbits[F]; pass 2: the records, exps[f] = the frame's exponent.
struct SplitSrc { int64_t s_n, s_c, s_h, s_w; int cb; };     // element strides of frame, channel (block), row, column; cb > 1: cb channels contiguous per block
__device__ __forceinline__ int64_t mm_src_ch(const SplitSrc& S, int c) { return S.cb > 1 ? (int64_t)(c / S.cb) * S.s_c + (c % S.cb) : (int64_t)c * S.s_c; }
__global__ __launch_bounds__(256) void corr_split_max_kernel(const float* __restrict__ src, SplitSrc S, int C, int H, int W, unsigned* __restrict__ maxbits) {
  const int f = blockIdx.y;
  const long long per = (long long)(C / 8) * H * W;
  unsigned mx = 0u;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), blk = (int)(i / ((long long)W * H));
    const float* p = src + (int64_t)f * S.s_n + (int64_t)y * S.s_h + (int64_t)x * S.s_w;
#pragma unroll
    for (int j = 0; j < 8; j++) mx = max(mx, __float_as_uint(p[mm_src_ch(S, 8 * blk + j)]) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  // ONE atomic per workgroup (round 6): the frames' maxima share a 128-byte line and device-scope atomics on one line retire one after the
  // other (~8.5 ns each) — with one per wave of 1 200 workgroups per frame the whole-ring pass of level 0 spent 1.3 ms in 153 600 of them
  __shared__ unsigned s_mx[4];
  if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned m = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
    if (m != 0u) atomicMax(&maxbits[f], m);
  }
}
__global__ __launch_bounds__(256) void corr_split_kernel(const float* __restrict__ src, SplitSrc S, int C, int H, int W, const unsigned* __restrict__ maxbits,
                                                         float* __restrict__ dst, int64_t dst_fstride, int* __restrict__ exps) {
  const int f = blockIdx.y;
  const long long per = (long long)(C / 8) * H * W;
  const int e = mm_scale_exp(maxbits[f]);
  if (blockIdx.x == 0 && threadIdx.x == 0) exps[f] = e;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), blk = (int)(i / ((long long)W * H));
    const float* p = src + (int64_t)f * S.s_n + (int64_t)y * S.s_h + (int64_t)x * S.s_w;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = p[mm_src_ch(S, 8 * blk + j)];
    v4u32 hi, lo;
    mm_split8_scaled(v, e, hi, lo);
    v4u32* o = reinterpret_cast<v4u32*>(dst + (int64_t)f * dst_fstride + i * 8);
    o[0] = hi; o[1] = lo;
  }
}

// Plan -> edge slot of workgroup `gid` of `nitems` (corr_plan_slot's map: heavy edges first, the rest XCD-aware), in closed form:
// sum over x < X of ceil((n - x) / 8) [n > x]  =  X * (n / 8) + min(X, n % 8).
__device__ __forceinline__ int mm_plan_slot(const int* __restrict__ order, int BE, int gid, int nitems) {
  const int nh = order ? min(max(order[BE], 0), BE) : 0;
  if (gid < nh) return gid;
  const int xcd = gid & 7;
  auto upto = [&](int n) -> int { return xcd * (n >> 3) + min(xcd, n & 7); };
  const int heavy_here = nh > xcd ? (nh - xcd + 7) >> 3 : 0;
  return nh + upto(nitems) - upto(nh) + (gid >> 3) - heavy_here;
}

// NW > 1 — the GROUP form (level index 1 from LDS; NL == 2, RFIX == 3, a group plan: corr_tile.h "group plan").  The texture addresser is
// the unit the per-edge form saturates (TA_BUSY 95-99 %, profiles/README.md r04), and at DEVO's patch density every level-1 position
// is wanted by ~90 edges.  A workgroup of NW waves takes ONE ITEM of the plan — up to MM_ITEM_EDGES edges of one group: one target frame,
// patch centres inside one tile of 6 x 6 level-1 cells — and stages the group's REGION of level 1 (15 x 15 positions, all channels: 56 KB
// fp16 / 113 KB fp32 split records) ONCE with LDS-DMA (contiguous kilobytes: 16 addresser cycles each, ~113 per item instead of 20 - 40 per
// EDGE).  Its waves then walk the item's edges exactly as the per-edge form does, level index 0 through the addresser — and level index
// 1's tiles with ds_read_b128 from the region, a unit that was idle.  Edges whose level-1 box leaves the region (the plan sorts
// them into the HEAVY class), heavy and dead edges keep the per-edge paths inside the same launch.
struct MmGroupArgs { const int* starts; int nbins, ngy, ngx; };     // bin starts of the group plan (nbins + 1 entries), bins, groups per frame
#ifndef DEVO_MM_ITEM_EDGES
#define DEVO_MM_ITEM_EDGES (1 << 20)
#endif
#ifndef DEVO_MM_ITEMS_PER_BIN
#define DEVO_MM_ITEMS_PER_BIN 1
#endif
constexpr int MM_ITEM_EDGES = DEVO_MM_ITEM_EDGES;      // an item = at most this many edges of one group (larger groups: several items, each staging the region)
constexpr int MM_ITEMS_PER_BIN = DEVO_MM_ITEMS_PER_BIN;    // workgroups reserved per bin (the last one takes whatever is left)
template <typename T> constexpr int mm_region_bytes(int C) {        // the staged region (radius 3), whole kilobytes (LDS-DMA writes 1 KB per instruction)
  return ((CORR_GRP_T + 2 * 3 + 3) * (CORR_GRP_T + 2 * 3 + 3) * C * (int)sizeof(T) + 1023) / 1024 * 1024;
}

template <typename T, int RMAX, int NKS, int NL, int RFIX, int NW = 1>   // NKS = C / 32 K steps per tile; NL = levels per wave; RFIX > 0: the radius is this constant
__global__ __launch_bounds__(64 * (NW > 1 ? NW : DEVO_MM_EPW)) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? DEVO_MM_WAVES - (RMAX > 3 ? 1 : 0) : DEVO_MM_WAVES32, sizeof(T) == 2 ? DEVO_MM_WAVES - (RMAX > 3 ? 1 : 0) : DEVO_MM_WAVES32))) void corr_fwd_mm_kernel(
    const T* __restrict__ fmap1_t, CorrLevel lv0, CorrLevel lv1, int nlev, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, T* __restrict__ out, int BE, int E, int Np, int n2,
    int C, int64_t out_estride, int64_t out_lstride, int R_arg, const int* __restrict__ order, int heavy_only,
    unsigned long long* __restrict__ trace, const int* __restrict__ exp1, MmGroupArgs grp) {
  constexpr bool LDS1 = NW > 1;
  static_assert(!LDS1 || (NL == 2 && RFIX == 3 && RMAX == 3 && DEVO_MM_EPW == 1), "the group form: fused two-level lookups at radius 3");
  const int R = RFIX > 0 ? RFIX : R_arg;        // (DEVO's radius 3 and the stress configuration's 5 as constants: window sizes, loop bounds and
                                                //  the epilogue's guards fold away)
  constexpr bool HALF = sizeof(T) == 2;
  constexpr unsigned ESZ = sizeof(T);
  constexpr int LPS = HALF ? 1 : 2;                 // 16-byte loads per lane and K step
  constexpr int RT = HALF ? DEVO_MM_RT16 : 2;       // ring of tiles (RT - 1 tiles in flight ahead of the products)
  const int wlvl = (NL == 1 && nlev == 2) ? ((blockIdx.x >> 3) & 1) : 0;                      // wave-uniform
  const int wgid = (NL == 1 && nlev == 2) ? (((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) : blockIdx.x;
  const int nwg = (NL == 1 && nlev == 2) ? (gridDim.x >> 1) : gridDim.x;
  auto second = [&](int l) -> bool { return NL == 2 ? (l != 0) : (wlvl != 0); };   // does index l mean pyramid level 1?
#define LVF(l, F) (second(l) ? lv1.F : lv0.F)
  constexpr int DMAX = 2 * RMAX + 2;
  // Result area (one, reused level after level), one of two layouts (as in corr_mfma.h):
  //   box layout  [p][BOXS]        slot s of pixel p at p * BOXS + s (boxes of <= CAP positions; BOXS = CAP + 4 so that the 8 lanes
  //                                of a ds_write_b128 group, 8 pixels x the same 4 slots, fall into different banks)
  //   raw windows [p][D*D + 1]     tap (a, c) of pixel p: window-by-window tiles (larger boxes)
  constexpr bool ALIGN4 = DEVO_MM_ALIGN4 != 0 && sizeof(T) == 2;     // measured (profiles/r04_lookup_experiments.txt, 9): fp16 64 -> 60 us, fp32 120 -> 124
#ifndef DEVO_MM_CAP
#define DEVO_MM_CAP 160
#endif
  // (a box beyond CAP positions walks the 9 windows one after the other — 36 tiles where a 12 x 12 box has 9: with CAP = 128, 6 % of cfg2's
  //  edges did, 18 % of all tiles)
  constexpr int CAP = RMAX <= 3 ? ((LDS1 && !HALF) ? 128 : DEVO_MM_CAP) : 256;      // (fp32 group form: the region leaves 5.4 KB per wave)
  static_assert(RMAX > 3 || (CAP % 16 == 0 && CAP >= 128 && CAP <= 192), "result area: 8 .. 12 tiles");
  constexpr int BOXS = CAP + 4;
  constexpr int RWIN_FLOATS = (PP * (DMAX * DMAX + 1) + 3) / 4 * 4;
  constexpr int RW_FLOATS = RWIN_FLOATS > PP * BOXS ? RWIN_FLOATS : PP * BOXS;
  constexpr int EPW = DEVO_MM_EPW;
  const int wv = (EPW > 1 || LDS1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;      // this wave's edge inside the workgroup
  // per wave: ONE level's result area (a level is blended before the next one's tiles land) | per (level index, pixel): dx, dy, tap (0, 0)'s
  // index in the result area, row stride | window origins (window-by-window tiles only)
  constexpr int WAVE_LDS = RW_FLOATS * 4 + NL * 16 * 4 * 4 + ((NL * PP * 2 * 4 + 15) / 16) * 16;
  constexpr int REGION_BYTES = LDS1 ? mm_region_bytes<T>(32 * NKS) : 0;
  float* s_rawwin;
  float (*s_geo)[16][4];
  int (*s_org)[PP][2];
  unsigned char* region = nullptr;
  if constexpr (LDS1) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char mm_dyn[];       // region | NW per-wave areas
    region = mm_dyn;
    unsigned char* w0 = mm_dyn + REGION_BYTES + wv * WAVE_LDS;
    s_rawwin = reinterpret_cast<float*>(w0);
    s_geo = reinterpret_cast<float (*)[16][4]>(w0 + RW_FLOATS * 4);
    s_org = reinterpret_cast<int (*)[PP][2]>(w0 + RW_FLOATS * 4 + NL * 16 * 4 * 4);
  } else {
    __shared__ __attribute__((aligned(16))) float s_rawwin_all[EPW][RW_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_geo_all[EPW][NL][16][4];
    __shared__ int s_org_all[EPW][NL][PP][2];
    s_rawwin = s_rawwin_all[wv];
    s_geo = s_geo_all[wv];
    s_org = s_org_all[wv];
  }
  const int lane = threadIdx.x & 63;
  int slot = 0;
  // ---- group form: this workgroup's item (a slot range of one bin of the group plan) and its region of level 1
  constexpr int GRP_RW = CORR_GRP_T + 2 * 3 + 3;          // region side in positions (15)
  constexpr unsigned RPOS = 8u * ESZ;                     // bytes per position and 8-channel block (level 1 is stored in 8-channel blocks)
  constexpr unsigned RROW = GRP_RW * RPOS, RBLK = GRP_RW * GRP_RW * RPOS;
  int it_lo = 0, it_hi = 0, it_b = -1, it_f = -1, ry0 = 0, rx0 = 0, hv_n = 0, dd_n = 0;
  unsigned long long g_st[4] = {0, 0, 0, 0};               // debug (DEVO_CORR_TRACE, group form): 100 MHz stamps of this wave: start, region ready, item done, end
  if (LDS1 && trace) g_st[0] = __builtin_amdgcn_s_memrealtime();
  bool region_pending = false;                            // this wave still owes the workgroup's ONE barrier (after its share of the DMA has landed)
  if constexpr (LDS1) {
    const int per = ((int)gridDim.x + 7) >> 3;            // workgroup g runs on XCD g % 8: every XCD owns a contiguous range of items (= of bins = of frames)
    const int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    const int bin = item / MM_ITEMS_PER_BIN, ki = item - bin * MM_ITEMS_PER_BIN;
    hv_n = min(max(order[BE], 0), BE);
    dd_n = min(max(order[2 * BE + 1], 0), BE - hv_n);
    if (bin < grp.nbins - 1) {                             // (the last bin is the DEAD class: per-edge work like the heavy one, spread over all waves)
      const int s0 = grp.starts[bin], s1 = grp.starts[bin + 1], cnt = max(s1 - s0, 0);
      const int nit = min(MM_ITEMS_PER_BIN, (cnt + MM_ITEM_EDGES - 1) / MM_ITEM_EDGES);
      if (ki < nit) {
        it_lo = s0 + (int)((long long)cnt * ki / nit); it_hi = s0 + (int)((long long)cnt * (ki + 1) / nit);
        const int gpf = grp.ngy * grp.ngx, bf = bin / gpf, g2 = bin - bf * gpf;
        it_b = bf / n2; it_f = bf - it_b * n2;
        ry0 = (g2 / grp.ngx) * CORR_GRP_T - 3 - 1; rx0 = (g2 % grp.ngx) * CORR_GRP_T - 3 - 1;
      }
    }
#ifdef DEVO_MM_DBG_NOSTAGE    // timing experiment (wrong results): nothing is staged
    if (false) {
#else
    if (it_hi > it_lo && it_f >= 0) {
#endif
      // the region [block][row][column][RPOS bytes], a verbatim copy of the level's blocks: chunk q = bytes q KB .. of it, 16 per lane;
      // rows / columns outside the frame and the bytes behind the region carry the out-of-range offset: the DMA writes zeros for them
      const T* base = static_cast<const T*>(lv1.fmap2) + (int64_t)it_b * lv1.s_b + (int64_t)it_f * lv1.s_n;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, lv1.frame_bytes, 0x00020000);
      const unsigned bb = (unsigned)lv1.block_stride * ESZ, shb1 = (unsigned)lv1.s_h * ESZ, swb1 = (unsigned)lv1.s_w * ESZ;
      const unsigned lds0 = (unsigned)(uintptr_t)region;
      for (int q = wv; q < REGION_BYTES / 1024; q += NW) {
        const unsigned Lb = (unsigned)q * 1024u + (unsigned)lane * 16u;
        const unsigned blk = Lb / RBLK, rem = Lb - blk * RBLK, row = rem / RROW, cbyte = rem - row * RROW, col = cbyte / RPOS;
        const int gy = ry0 + (int)row, gx = rx0 + (int)col;
        const bool ok = blk < (unsigned)(C / 8) && (unsigned)gy < (unsigned)lv1.H2 && (unsigned)gx < (unsigned)lv1.W2;
        const unsigned voff = ok ? blk * bb + (unsigned)gy * shb1 + (unsigned)gx * swb1 + (cbyte - col * RPOS) : 0x80000000u;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)q * 1024u));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(dst) : "memory");
      }
      region_pending = true;
    }
  }
  auto region_ready = [&]() {                             // once per wave of a staging workgroup: own DMA landed, then everybody's
    if constexpr (LDS1) {
      if (region_pending) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        region_pending = false;
        if (trace) g_st[1] = __builtin_amdgcn_s_memrealtime();
      }
    }
  };
  if constexpr (!LDS1) {
  if (EPW == 1) slot = mm_plan_slot(order, BE, wgid, nwg);
  else {                                       // workgroup g (on XCD g % 8) takes EPW consecutive slots of its XCD's contiguous share of the plan
    const int per = (nwg + 7) >> 3;
    slot = (((wgid & 7) * per + (wgid >> 3)) * EPW) + wv;
    if ((wgid >> 3) >= per) slot = BE;
  }
  if (heavy_only) {
    slot = (int)blockIdx.x * EPW + wv;
    if (slot >= (order ? min(max(order[BE], 0), BE) : 0)) return;
  }
  if (slot >= BE) return;                     // wave-uniform; no workgroup barriers in this (per-edge) form
  }
  // group form: the item's edges wv, wv + NW, .. first, then this wave's share of the HEAVY and DEAD classes (per-edge paths, no region)
  for (int turn = 0; ; turn++) {
  bool in_item = false;
  if constexpr (LDS1) {
    const int idx = it_lo + wv + turn * NW;
    if (idx < it_hi) { slot = idx; in_item = true; }
    else {
      region_ready();                          // (a wave without an edge of the item still owes the barrier)
      if (trace && g_st[2] == 0) g_st[2] = __builtin_amdgcn_s_memrealtime();
      const int nturn = (it_hi - it_lo - wv + NW - 1) / NW;                     // turns this wave spent on the item (>= 0)
      const int h = ((int)blockIdx.x * NW + wv) + (turn - max(nturn, 0)) * (int)gridDim.x * NW;
      if (h >= hv_n + dd_n) break;
      slot = h < hv_n ? h : BE - dd_n + (h - hv_n);          // the HEAVY class in front of the plan, the DEAD one at its end
    }
  } else if (turn > 0) break;
  unsigned long long t_st[5] = {0, 0, 0, 0, 0};  // debug (DEVO_CORR_TRACE): cycle stamps of this wave's phases
  if (trace) t_st[0] = __builtin_readcyclecounter();
  const int be = order ? order[slot] : slot;
  const int D = 2 * R + 2, ntap = D * D;
  const int b = be / E, e = be - b * E;
  const int64_t pi = ii[e];
  const int64_t fj = jj[e];
  // fp32 (split records): the sums come out scaled by 2^-(e_patch + e_frame); the blend weights carry the inverse (exact powers of two)
  int sexp[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) sexp[l] = 0;
  if constexpr (sizeof(T) == 4) {
    const int e1 = exp1 ? exp1[(int64_t)b * Np + pi] : 0;
#pragma unroll
    for (int l = 0; l < NL; l++) { const int* ex = LVF(l, exps); sexp[l] = e1 + (ex ? ex[(int64_t)b * n2 + fj] : 0); }
  }

  // ---- geometry.  Lane 16 l + p owns patch pixel p at level index l: both levels are worked out side by side (the 18 coordinates
  //      come through the scalar cache and are written into both lane groups)
  const int lp = lane & 15, lsel = (NL == 2 && lane >= 16) ? 1 : 0;
  float cpx = 0.0f, cpy = 0.0f;
  {
    const float* __restrict__ ce = coords + (int64_t)be * (2 * PP);
    float cv[2 * PP];
#pragma unroll
    for (int p = 0; p < 2 * PP; p++) cv[p] = ce[p];
#pragma unroll
    for (int p = 0; p < PP; p++) {
      asm("v_writelane_b32 %0, %1, %2" : "+v"(cpx) : "s"(cv[p]), "n"(p));
      asm("v_writelane_b32 %0, %1, %2" : "+v"(cpy) : "s"(cv[PP + p]), "n"(p));
      if (NL == 2) {
        asm("v_writelane_b32 %0, %1, %2" : "+v"(cpx) : "s"(cv[p]), "n"(16 + p));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(cpy) : "s"(cv[PP + p]), "n"(16 + p));
      }
    }
  }
  if (trace) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_st[1] = __builtin_readcyclecounter(); }
  float qx, qy;                               // coordinates at this lane's level: coords / div (the reference's true division)
  {
    auto pow2 = [](float d) -> bool { return (__float_as_uint(d) & 0x807fffffu) == 0u && d >= 1.0f; };
    bool all_pow2 = true;
#pragma unroll
    for (int l = 0; l < NL; l++) all_pow2 = all_pow2 && pow2(LVF(l, coord_div));
    const float dv = (NL == 2 && lsel) ? LVF(NL - 1, coord_div) : LVF(0, coord_div);
    if (all_pow2) {                           // wave-uniform: x * (1 / 2^k) is the correctly rounded x / 2^k
      const float iv = 1.0f / dv;
      qx = cpx * iv; qy = cpy * iv;
    } else { qx = cpx / dv; qy = cpy / dv; }
  }
  const float flx = floorf(qx), fly = floorf(qy);
  const int ox = floor_to_int(qx) - R, oy = floor_to_int(qy) - R;      // this lane's window origin
  // min / max over the 9 pixels of each 16-lane row with DPP row shifts: the result sits in lane 15 of the row
  auto row_min = [&](int v) -> int {
    v = (lp < PP) ? v : 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x111, 0xf, 0xf, false));     // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x112, 0xf, 0xf, false));     // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x114, 0xf, 0xf, false));     // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x118, 0xf, 0xf, false));     // row_shr:8
    return v;
  };
  const int vxmin = row_min(ox), vxmax = row_min(-ox), vymin = row_min(oy), vymax = row_min(-oy);
  const int WT = (ntap + 15) >> 4;                // tiles per window in window mode (a window's taps padded to whole tiles)
  struct Geo { int xmin, ymin, bw, bh, nslots, ntile; bool box_mode; float inv_bw; };      // wave-uniform
  auto make_geo = [&](int l) -> Geo {
    const int xmin = __builtin_amdgcn_readlane(vxmin, 15 + 16 * l), xmax = -__builtin_amdgcn_readlane(vxmax, 15 + 16 * l);
    const int ymin = __builtin_amdgcn_readlane(vymin, 15 + 16 * l), ymax = -__builtin_amdgcn_readlane(vymax, 15 + 16 * l);
    Geo g;
    g.xmin = xmin; g.ymin = ymin; g.bw = xmax - xmin + D; g.bh = ymax - ymin + D;
    if (ALIGN4 && !(LDS1 && l == 1)) {                     // (the group form's level index 1 comes from LDS: the tight box)
      // The texture addresser retires one quad of lanes per cycle and 128-byte line it touches (tools/ubench/l2_fill.hip).  A quad is four
      // consecutive box positions (16 or 32 bytes each): with the box's left edge at a multiple of 4 positions and its width a multiple
      // of 4, no quad straddles a line or a box row, for ~30 % more positions.  It pays with 16-byte position records (fp16); with fp32's
      // 32-byte records the tight box stays faster.  (Boxes that would no longer fit the result area keep their tight form.)
      const int xa = xmin & ~3, bwa = (xmax + D - xa + 3) & ~3;
      if ((long long)bwa * (ymax - ymin + D) <= (long long)CAP) { g.xmin = xa; g.bw = bwa; }
    }
    const long long npos_ll = (long long)g.bw * (ymax - ymin + D);
    g.box_mode = npos_ll <= (long long)CAP;              // else: the 9 windows one after the other
    g.nslots = g.box_mode ? (int)npos_ll : PP * ntap;
    g.ntile = g.box_mode ? ((int)npos_ll + 15) >> 4 : PP * WT;
    // the whole box outside the frame: every tap is 0 (correlation_kernel.cu:136) — no tiles for this level; its result area is
    // zeros read through a D x D box at the area's start
    if (xmax + D <= 0 || ymax + D <= 0 || xmin >= LVF(l, W2) || ymin >= LVF(l, H2)) { g.nslots = 0; g.ntile = 0; g.box_mode = true; g.bw = D; }
    g.inv_bw = __builtin_amdgcn_rcpf((float)g.bw);
    return g;
  };
  const Geo g0 = make_geo(0);
  const Geo g1 = (NL == 2) ? make_geo(1) : g0;
  bool all_box = g0.box_mode;
  if (NL == 2) all_box = all_box && g1.box_mode;
  {
    // what the epilogue needs per (level index, pixel): blend fractions, where tap (0, 0) sits in the result area, the row stride
    const bool l1 = NL == 2 && lsel;
    const bool boxm = l1 ? g1.box_mode : g0.box_mode, live = (l1 ? g1.nslots : g0.nslots) > 0;
    const int bw_ = l1 ? g1.bw : g0.bw, xm = l1 ? g1.xmin : g0.xmin, ym = l1 ? g1.ymin : g0.ymin;
    const int p_ = min(lp, PP - 1);
    const int fb = !boxm ? p_ * (ntap + 1) : live ? p_ * BOXS + (oy - ym) * bw_ + (ox - xm) : p_ * BOXS;
    if (lp < PP && lane < 16 * NL) {
      *reinterpret_cast<float4*>(&s_geo[lsel][lp][0]) = float4{qx - flx, qy - fly, __int_as_float(fb), __int_as_float(boxm ? bw_ : D)};
      if (!all_box) { s_org[lsel][lp][0] = ox; s_org[lsel][lp][1] = oy; }
    }
  }
  const int nt0 = g0.ntile, ntot = (NL == 2) ? nt0 + g1.ntile : nt0;     // flat tile list: level index 0, then 1

  auto frame_rsrc = [&](int l) -> __amdgpu_buffer_rsrc_t {
    const T* base = static_cast<const T*>(LVF(l, fmap2)) + (int64_t)b * LVF(l, s_b) + fj * LVF(l, s_n);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, LVF(l, frame_bytes), 0x00020000);
  };
  constexpr unsigned OFF_NONE = 0x80000000u;               // > any frame (the launcher guarantees < 2^31 bytes)
  const int mi = lane & 15, kg = lane >> 4;                // A: row (position) / K group.  B, D: column (pixel) / K group, row group

  // Channel c of a position sits at byte  piece(c) = block(c) * block_bytes + (c % cb) * ESZ  behind the position's offset.
  // Lane piece = channels 8 kg .. + 7 of the step (32 channels): fp16 16 bytes; fp32 (split records, 8 channels per block): the
  // block's 16 bytes of hi halves, the lo halves 16 bytes behind them (`second`).
  struct Pieces { unsigned step, lane0, second; };
  auto pieces_of = [&](int l) -> Pieces {
    const int sh = LVF(l, cb_shift);
    const unsigned bb = (unsigned)LVF(l, block_stride) * ESZ;
    auto piece = [&](unsigned c) -> unsigned { const unsigned blk = c >> sh; return blk * bb + (c - (blk << sh)) * ESZ; };
    return Pieces{piece(32), piece(8u * (unsigned)kg), 16u};
  };

  // ---- B operand: the patch, channels 32 s + 8 kg .. + 7 of pixel mi (columns >= 9: zeros); fp32: hi and lo halves of the split record
  mm_h8 bh[NKS], bl[HALF ? 1 : NKS];
  {
    const T* __restrict__ f1 = fmap1_t + ((int64_t)b * Np + pi) * C * PP;           // [9][C]
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(f1), 0, (unsigned)(C * PP) * ESZ, 0x00020000);
    const unsigned boff = mi < PP ? (unsigned)(mi * C + 8 * kg) * ESZ : OFF_NONE;
#pragma unroll
    for (int s = 0; s < NKS; s++) {
      if constexpr (HALF) {
        bh[s] = __builtin_bit_cast(mm_h8, __builtin_amdgcn_raw_buffer_load_b128(rs1, boff, (unsigned)s * 64u, 0));
      } else {
        // fp32 patches arrive as split records (devo_corr_patch_transpose): every 8 channels as 32 bytes of fp16 (hi0..7 | lo0..7)
        bh[s] = __builtin_bit_cast(mm_h8, __builtin_amdgcn_raw_buffer_load_b128(rs1, boff, (unsigned)s * 128u, 0));
        bl[s] = __builtin_bit_cast(mm_h8, __builtin_amdgcn_raw_buffer_load_b128(rs1, boff, (unsigned)s * 128u + 16u, 0));
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);        // patch loads first: the loop's s_waitcnt counts assume they are the oldest
  if (trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_st[2] = __builtin_readcyclecounter(); }

  v4u32 rb[RT][NKS][LPS];
  // the products of one tile (ring slot r) -> 16 positions x 16 columns of sums
  auto multiply = [&](int r) -> mm_f4 {
    mm_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NKS; s++) {
      if constexpr (HALF) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mm_h8, rb[r][s][0]), bh[s], acc, 0, 0, 0);
      } else {
        const mm_h8 ah = __builtin_bit_cast(mm_h8, rb[r][s][0]), al = __builtin_bit_cast(mm_h8, rb[r][s][1]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[s], acc, 0, 0, 0);       // small terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[s], acc, 0, 0, 0);
      }
    }
    return acc;
  };
  // ---- fused bilinear blend + axis swap + output permutation (correlation_kernel.cu:221-232), level by level.
  //      Output element (l, t), t = q * 9 + p with q = cx * Dm + a (cx = x offset: permute(0,1,3,2,4,5), a = y offset), goes to
  //      out[be * estride + t * lstride + offset(l)].
  //      Lane (a, p) = (lane / 9, lane % 9) blends ROW a of pixel p's window: the Dm + 1 taps of rows a, a + 1 read once, Dm outputs in the
  //      reference's blend order.  Level index 0's outputs wait in registers while level index 1's tiles reuse the result area; the
  //      standard stacked record (torch.stack([c0, c1], -1)) then leaves as ONE 4- / 8-byte piece per output pair: the 63 lanes of a
  //      store write 63 consecutive pieces.
  const int Dm = D - 1;
  constexpr int DMM = 2 * RMAX + 1;
  constexpr int NRND = (DMM + 6) / 7;                         // rounds of 7 window rows (radius <= 3: one)
  const bool paired = NL == 2 && out_lstride == 2 && lv0.out_offset == 0 && lv1.out_offset == 1 && (out_estride & 1) == 0 &&
                      (reinterpret_cast<uintptr_t>(out) & 7) == 0;
  const int ar = (lane * 57) >> 9, ep = lane - 9 * ar;                                     // lane / 9, lane % 9 for lane < 64
  float held[NL][NRND][DMM];                                  // blended outputs of (level index, round, x offset) for this lane's (a, p)
  auto blend = [&](int l) {
    wave_lds_fence();
    const float4 ge = *reinterpret_cast<const float4*>(&s_geo[l][ep][0]);
    float w0, w1, w2, w3;
    {
#pragma clang fp contract(off)
      w0 = (1.0f - ge.x) * (1.0f - ge.y); w1 = ge.x * (1.0f - ge.y); w2 = (1.0f - ge.x) * ge.y; w3 = ge.x * ge.y;   // blend4's factors
    }
    if constexpr (!HALF) { w0 = ldexpf(w0, sexp[l]); w1 = ldexpf(w1, sexp[l]); w2 = ldexpf(w2, sexp[l]); w3 = ldexpf(w3, sexp[l]); }
    const int rs = __float_as_int(ge.w);
#pragma unroll
    for (int rd = 0; rd < NRND; rd++) {
      const int a = 7 * rd + ar;
      if (7 * rd < Dm) {
        const float* r0 = s_rawwin + __float_as_int(ge.z) + min(a, Dm - 1) * rs;      // (lanes beyond the last row read row Dm - 1: in range)
        float t0[DMM + 1], u0[DMM + 1];
#pragma unroll
        for (int c = 0; c <= DMM; c++) if (c <= Dm) { t0[c] = r0[c]; u0[c] = r0[rs + c]; }
#pragma unroll
        for (int cx = 0; cx < DMM; cx++) {
          if (cx < Dm) {
            float o;
            {
#pragma clang fp contract(off)
              o = w0 * t0[cx]; o = o + w1 * t0[cx + 1]; o = o + w2 * u0[cx]; o = o + w3 * u0[cx + 1];
            }
            held[l][rd][cx] = o;
          }
        }
      }
    }
    wave_lds_fence();                                         // the taps are in registers before the next level's tiles overwrite the area
  };
  // Tile counts a radius-3 box can have: 0 (the box misses the frame) or 4 .. 8 (64 .. 128 positions): for those the tile loops exist as
  // straight-line code per count (a uniform switch picks one): no dead fetch, no scalar bookkeeping, every wait counted by the compiler.
  constexpr bool EXACT = RFIX == 3 && RMAX == 3;
  auto exact_count = [](int n, int hi) -> bool { return n == 0 || (n >= 4 && n <= hi); };
  constexpr int K0MAX = CAP / 16;                          // tiles of the largest box the result area holds (level index 1: up to 8)
  if (ntot == 0) {
    // no level's box touches its frame (the plan's DEAD class, 11 % of cfg2's edges): every output is 0 (correlation_kernel.cu:136) —
    // no result area, no blend
#pragma unroll
    for (int l = 0; l < NL; l++)
#pragma unroll
      for (int rd = 0; rd < NRND; rd++)
#pragma unroll
        for (int cx = 0; cx < DMM; cx++) held[l][rd][cx] = 0.0f;
  } else
  if (EXACT && all_box && exact_count(g0.ntile, K0MAX) && (NL == 1 || exact_count(g1.ntile, 8))) {
    // Every wave-instruction of a load occupies the CU's texture addresser (16 quads, ~1.4 cycles each: a quad of four positions x 16 bytes
    // straddles a 128-byte line three times in eight) whether its lanes fetch or not — the addresser is ~97 % busy in this kernel
    // (TA_BUSY, profiles/README.md r04) — so nothing is fetched that a box does not have.
    constexpr int XR = 2;                                    // ring of 2 tiles: one in flight ahead of the products
    v4u32 (&xr)[RT][NKS][LPS] = rb;
    __amdgpu_buffer_rsrc_t rsl[NL];
    unsigned lane_piece[NL], step_b[NL], second_b[NL], shb[NL], swb[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
      rsl[l] = frame_rsrc(l);
      const Pieces pc = pieces_of(l);
      lane_piece[l] = pc.lane0; step_b[l] = pc.step; second_b[l] = pc.second;
      shb[l] = (unsigned)LVF(l, s_h) * ESZ; swb[l] = (unsigned)LVF(l, s_w) * ESZ;
    }
    const float fmi = (float)mi + 0.5f;
    auto fetch_lt = [&](int ring, int l, int t) {            // (ring, l, t: constants after inlining)
      const Geo& G = l ? g1 : g0;
      const int sl = t * 16 + mi;
      const int pyy = (int)((fmi + (float)(t * 16)) * G.inv_bw);      // exact: sl < 2^16, error margin 0.5 / bw
      const int gy = G.ymin + pyy, gx = G.xmin + (sl - __mul24(pyy, G.bw));
      const bool ok = sl < G.nslots && (unsigned)gy < (unsigned)LVF(l, H2) && (unsigned)gx < (unsigned)LVF(l, W2);
      const unsigned voff = ok ? (unsigned)gy * shb[l] + (unsigned)gx * swb[l] + lane_piece[l] : OFF_NONE;
#ifdef DEVO_MM_DBG_NOL1      // timing experiment (wrong results): level index 1's tiles are not fetched at all (what a level read from LDS would leave of the addresser's time)
      if (l == 1) {
#pragma unroll
        for (int s = 0; s < NKS; s++) { xr[ring][s][0] = v4u32{voff, voff ^ (unsigned)s, voff, voff}; if constexpr (!HALF) xr[ring][s][1] = v4u32{voff, voff, voff ^ (unsigned)s, voff}; }
        return;
      }
#endif
#pragma unroll
      for (int s = 0; s < NKS; s++) {
        xr[ring][s][0] = __builtin_amdgcn_raw_buffer_load_b128(rsl[l], voff, (unsigned)s * step_b[l], 0);
        if constexpr (!HALF) xr[ring][s][1] = __builtin_amdgcn_raw_buffer_load_b128(rsl[l], voff, (unsigned)s * step_b[l] + second_b[l], 0);
      }
    };
    float* const dst = s_rawwin + mi * BOXS + 4 * kg;        // lane (n, rg): positions 4 rg .. + 3 of column n
    // level `lv` with K tiles, its tile 0 already in flight in ring slot OFS; NEXT: the following level's tile 0 is fetched behind the last one
    auto run_level = [&](auto lv_c, auto k_c, auto ofs_c, auto next_c) {
      constexpr int LV = decltype(lv_c)::value, K = decltype(k_c)::value, OFS = decltype(ofs_c)::value;
      constexpr bool NEXT = decltype(next_c)::value;
#pragma unroll
      for (int i = 0; i < K; i++) {
        if (i + 1 < K) fetch_lt((OFS + i + 1) % XR, LV, i + 1);
        else if (NEXT) fetch_lt((OFS + i + 1) % XR, LV + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        const mm_f4 acc = multiply((OFS + i) % XR);
        if (mi < PP) *reinterpret_cast<mm_f4*>(dst + i * 16) = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // group form: level index 1's tiles from the staged region — position (gy, gx) of block blk at (gy - ry0) RROW + (gx - rx0) RPOS +
    // blk RBLK; lanes beyond the box read its last position (their rows of the product are never looked at)
    auto fetch_lds = [&](int ring, int t) {
      const int sl = min(t * 16 + mi, g1.nslots - 1);
      const int pyy = (int)(((float)sl + 0.5f) * g1.inv_bw);
      const int gy = g1.ymin + pyy - ry0, gx = g1.xmin + (sl - __mul24(pyy, g1.bw)) - rx0;
      const unsigned char* a = region + (unsigned)gy * RROW + (unsigned)gx * RPOS + (unsigned)kg * RBLK;
#pragma unroll
      for (int s = 0; s < NKS; s++) {
        xr[ring][s][0] = *reinterpret_cast<const v4u32*>(a + (unsigned)s * 4u * RBLK);
        if constexpr (!HALF) xr[ring][s][1] = *reinterpret_cast<const v4u32*>(a + (unsigned)s * 4u * RBLK + 16u);
      }
    };
    auto run_level_lds = [&](auto k_c) {
      constexpr int K = decltype(k_c)::value;
      fetch_lds(0, 0);
#pragma unroll
      for (int i = 0; i < K; i++) {
        if (i + 1 < K) fetch_lds((i + 1) % XR, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        const mm_f4 acc = multiply(i % XR);
        if (mi < PP) *reinterpret_cast<mm_f4*>(dst + i * 16) = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // does level index 1 of this edge come from the region?  (wave-uniform; what the group plan promises for the item's edges, checked)
    bool from_lds = false;
    if constexpr (LDS1)
      from_lds = in_item && it_f >= 0 && b == it_b && (int)fj == it_f && lv1.cb_shift == 3 && g1.ntile > 0 && g1.xmin >= rx0 && g1.ymin >= ry0 &&
                 g1.xmin + g1.bw <= rx0 + GRP_RW && g1.ymin + g1.bh <= ry0 + GRP_RW;
#ifdef DEVO_MM_DBG_NOLDS     // timing experiment: the group form's schedule with level index 1 through the addresser
    from_lds = false;
#endif
    using std::integral_constant;
    auto zero_area = [&]() { for (int i = lane; i < PP * BOXS; i += 64) s_rawwin[i] = 0.0f; };
    const int k0 = g0.ntile, k1 = NL == 2 ? g1.ntile : 0;
    // level index 0 (its tile 0 — or, without tiles, level index 1's — goes first)
    if (k0 > 0) fetch_lt(0, 0, 0); else if (NL == 2 && !LDS1) fetch_lt(0, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    constexpr bool N2 = NL == 2 && !LDS1;                    // (group form: no fetch of level index 1's first tile behind level index 0's last)
    switch (k0) {
      case 4: run_level(integral_constant<int, 0>{}, integral_constant<int, 4>{}, integral_constant<int, 0>{}, integral_constant<bool, N2>{}); break;
      case 5: run_level(integral_constant<int, 0>{}, integral_constant<int, 5>{}, integral_constant<int, 0>{}, integral_constant<bool, N2>{}); break;
      case 6: run_level(integral_constant<int, 0>{}, integral_constant<int, 6>{}, integral_constant<int, 0>{}, integral_constant<bool, N2>{}); break;
      case 7: run_level(integral_constant<int, 0>{}, integral_constant<int, 7>{}, integral_constant<int, 0>{}, integral_constant<bool, N2>{}); break;
      case 8: run_level(integral_constant<int, 0>{}, integral_constant<int, 8>{}, integral_constant<int, 0>{}, integral_constant<bool, N2>{}); break;
      case 9: if constexpr (K0MAX >= 9) run_level(integral_constant<int, 0>{}, integral_constant<int, 9>{}, integral_constant<int, 0>{}, integral_constant<bool, N2>{}); break;
      case 10: if constexpr (K0MAX >= 10) run_level(integral_constant<int, 0>{}, integral_constant<int, 10>{}, integral_constant<int, 0>{}, integral_constant<bool, N2>{}); break;
      case 11: if constexpr (K0MAX >= 11) run_level(integral_constant<int, 0>{}, integral_constant<int, 11>{}, integral_constant<int, 0>{}, integral_constant<bool, N2>{}); break;
      case 12: if constexpr (K0MAX >= 12) run_level(integral_constant<int, 0>{}, integral_constant<int, 12>{}, integral_constant<int, 0>{}, integral_constant<bool, N2>{}); break;
      default: zero_area(); break;                           // no tiles: its box of zeros
    }
    if constexpr (LDS1) {
      blend(0);
      if (k1 == 0) zero_area();
      if (from_lds) {
        region_ready();
        switch (k1) {
          case 4: run_level_lds(integral_constant<int, 4>{}); break;
          case 5: run_level_lds(integral_constant<int, 5>{}); break;
          case 6: run_level_lds(integral_constant<int, 6>{}); break;
          case 7: run_level_lds(integral_constant<int, 7>{}); break;
          case 8: run_level_lds(integral_constant<int, 8>{}); break;
          default: break;
        }
      } else if (k1 > 0) {                                   // (not promised by the plan: through the addresser, like the per-edge form)
        fetch_lt(0, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        switch (k1) {
          case 4: run_level(integral_constant<int, 1>{}, integral_constant<int, 4>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
          case 5: run_level(integral_constant<int, 1>{}, integral_constant<int, 5>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
          case 6: run_level(integral_constant<int, 1>{}, integral_constant<int, 6>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
          case 7: run_level(integral_constant<int, 1>{}, integral_constant<int, 7>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
          case 8: run_level(integral_constant<int, 1>{}, integral_constant<int, 8>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
          default: break;
        }
      }
    } else
    if constexpr (NL == 2) {
      blend(0);                                              // (level index 1's tile 0 is in flight)
      if (k1 == 0) zero_area();
      // level index 1 starts in ring slot k0 % 2
      switch (k1 * 2 + (k0 & 1)) {
        case 8: run_level(integral_constant<int, 1>{}, integral_constant<int, 4>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
        case 9: run_level(integral_constant<int, 1>{}, integral_constant<int, 4>{}, integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
        case 10: run_level(integral_constant<int, 1>{}, integral_constant<int, 5>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
        case 11: run_level(integral_constant<int, 1>{}, integral_constant<int, 5>{}, integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
        case 12: run_level(integral_constant<int, 1>{}, integral_constant<int, 6>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
        case 13: run_level(integral_constant<int, 1>{}, integral_constant<int, 6>{}, integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
        case 14: run_level(integral_constant<int, 1>{}, integral_constant<int, 7>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
        case 15: run_level(integral_constant<int, 1>{}, integral_constant<int, 7>{}, integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
        case 16: run_level(integral_constant<int, 1>{}, integral_constant<int, 8>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
        case 17: run_level(integral_constant<int, 1>{}, integral_constant<int, 8>{}, integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
        default: break;                                      // (k1 == 0: the tile fetched ahead found nothing)
      }
    }
    blend(NL - 1);
  } else
  if (all_box) {
    // ---- the usual case: ONE flat list of the tiles both levels really have (level index 0's, then 1's), RT tiles per loop turn.
    //      Every wave-instruction of a load occupies the CU's texture addresser for 16 cycles whether its lanes fetch or not (measured:
    //      with all fetches switched off the kernel kept 3/4 of its time, profiles/README.md r04) — so no slot of a static schedule is
    //      spent on tiles a box does not have; what remains are the RT - 1 fetches that run ahead past the last tile.  The level's
    //      descriptors sit in scalar registers twice, a tile selects its set.
    struct LS { __amdgpu_buffer_rsrc_t rs; unsigned step, second, shb, swb; int H2, W2, xmin, ymin, bw, nslots; float inv_bw; };
    LS ls[NL];
    unsigned lane_piece[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const Geo& G = l ? g1 : g0;
      const Pieces pc = pieces_of(l);
      ls[l] = LS{frame_rsrc(l), pc.step, pc.second, (unsigned)LVF(l, s_h) * ESZ, (unsigned)LVF(l, s_w) * ESZ, LVF(l, H2), LVF(l, W2),
                 G.xmin, G.ymin, G.bw, G.nslots, G.inv_bw};
      lane_piece[l] = pc.lane0;
    }
    const float fmi = (float)mi + 0.5f;
    // level index 0's tile count is rounded up to whole loop turns (its last turn may carry tiles beyond the box: they fetch nothing),
    // so that a turn never straddles the two levels and the blend of level index 0 sits BETWEEN two loops instead of inside one
    const int nt0p = NL == 2 ? (nt0 + RT - 1) / RT * RT : nt0, ntotp = NL == 2 ? nt0p + g1.ntile : nt0;
    auto fetch_tile = [&](int ring, int tt) {
      const bool l1 = NL == 2 && tt >= nt0p;                 // wave-uniform
      const LS& S = l1 ? ls[NL - 1] : ls[0];
      const int t = l1 ? tt - nt0p : tt;
      const int sl = t * 16 + mi;
      const int pyy = (int)((fmi + (float)(t * 16)) * S.inv_bw);      // exact: sl < 2^16, error margin 0.5 / bw
      const int gy = S.ymin + pyy, gx = S.xmin + (sl - __mul24(pyy, S.bw));
#ifdef DEVO_MM_DBG_FRAC      // timing experiment (wrong results): fetch only DEVO_MM_DBG_FRAC / 8 of every box
      const bool ok = tt < ntotp && sl < (S.nslots * DEVO_MM_DBG_FRAC) / 8 && (unsigned)gy < (unsigned)S.H2 && (unsigned)gx < (unsigned)S.W2;
#else
      const bool ok = tt < ntotp && sl < S.nslots && (unsigned)gy < (unsigned)S.H2 && (unsigned)gx < (unsigned)S.W2;
#endif
      const unsigned voff = ok ? (unsigned)gy * S.shb + (unsigned)gx * S.swb + (l1 ? lane_piece[NL - 1] : lane_piece[0]) : OFF_NONE;
#pragma unroll
      for (int s = 0; s < NKS; s++) {
        rb[ring][s][0] = __builtin_amdgcn_raw_buffer_load_b128(S.rs, voff, (unsigned)s * S.step, 0);
        if constexpr (!HALF) rb[ring][s][1] = __builtin_amdgcn_raw_buffer_load_b128(S.rs, voff, (unsigned)s * S.step + S.second, 0);
      }
    };
#pragma unroll
    for (int r = 0; r < RT - 1; r++) { fetch_tile(r, r); __builtin_amdgcn_sched_barrier(0); }
    float* const dst = s_rawwin + mi * BOXS + 4 * kg;        // lane (n, rg): positions 4 rg .. + 3 of column n
    auto turn = [&](int t0, int tbase, int tend) {           // tiles t0 .. t0 + RT - 1 of the level whose first flat tile is tbase
#pragma unroll
      for (int r = 0; r < RT; r++) {
        fetch_tile((r + RT - 1) % RT, t0 + r + RT - 1);
        __builtin_amdgcn_sched_barrier(0);
        const mm_f4 acc = multiply(r);
        if (t0 + r < tend && mi < PP) *reinterpret_cast<mm_f4*>(dst + (t0 + r - tbase) * 16) = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (g0.ntile == 0) { for (int i = lane; i < PP * BOXS; i += 64) s_rawwin[i] = 0.0f; }      // a level without tiles: its box of zeros
    for (int t0 = 0; t0 < nt0p; t0 += RT) turn(t0, 0, nt0);
    if (NL == 2) {
      blend(0);                                              // (level index 1's first tiles are in flight)
      if (g1.ntile == 0) { for (int i = lane; i < PP * BOXS; i += 64) s_rawwin[i] = 0.0f; }
      for (int t0 = nt0p; t0 < ntotp; t0 += RT) turn(t0, nt0p, ntotp);
    }
    blend(NL - 1);
  } else {
    // ---- boxes beyond the result area (the plan's HEAVY class): level by level a dynamic list of tiles, window by window where needed
    wave_lds_fence();                                        // (s_org)
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const Geo& G = l ? g1 : g0;
      const int ntl = G.ntile;
      if (G.nslots == 0)                                     // a level without tiles: its D x D box of zeros
        for (int i = lane; i < PP * BOXS; i += 64) s_rawwin[i] = 0.0f;
      const __amdgpu_buffer_rsrc_t rs = frame_rsrc(l);
      const Pieces pc = pieces_of(l);
      auto fetch = [&](int ring, int t) {
        int gy, gx;
        bool listed;
        if (G.box_mode) {
          const int sl = t * 16 + mi;
          const int pyy = (int)(((float)sl + 0.5f) * G.inv_bw);
          gy = G.ymin + pyy; gx = G.xmin + (sl - pyy * G.bw);
          listed = sl < G.nslots;
        } else {
          const int wp = t / WT, tw = (t - wp * WT) * 16 + mi;      // (wave-uniform window, lane's tap)
          const int ta = tw / D;
          gy = s_org[l][min(wp, PP - 1)][1] + ta; gx = s_org[l][min(wp, PP - 1)][0] + (tw - ta * D);
          listed = tw < ntap;
        }
        const bool ok = listed && t < ntl && gy >= 0 && gy < LVF(l, H2) && gx >= 0 && gx < LVF(l, W2);
        const unsigned voff = ok ? (unsigned)(gy * (int)LVF(l, s_h) + gx * (int)LVF(l, s_w)) * ESZ + pc.lane0 : OFF_NONE;
#pragma unroll
        for (int s = 0; s < NKS; s++) {
          rb[ring][s][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)s * pc.step, 0);
          if constexpr (!HALF) rb[ring][s][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)s * pc.step + pc.second, 0);
        }
      };
#pragma unroll
      for (int r = 0; r < RT - 1; r++) { fetch(r, r); __builtin_amdgcn_sched_barrier(0); }
      for (int t0 = 0; t0 < ntl; t0 += RT) {
#pragma unroll
        for (int r = 0; r < RT; r++) {
          const int t = t0 + r;
          fetch((r + RT - 1) % RT, t + RT - 1);
          __builtin_amdgcn_sched_barrier(0);
          const mm_f4 acc = multiply(r);
          if (t < ntl) {                                              // wave-uniform; no memory loads inside
            if (G.box_mode) {
              if (mi < PP) *reinterpret_cast<mm_f4*>(s_rawwin + mi * BOXS + t * 16 + 4 * kg) = acc;
            } else {
              const int wp = t / WT, tw = (t - wp * WT) * 16 + 4 * kg;
              if (mi == wp) {
#pragma unroll
                for (int j = 0; j < 4; j++) if (tw + j < ntap) s_rawwin[wp * (ntap + 1) + tw + j] = acc[j];
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      blend(l);
    }
  }
  if (trace) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t_st[3] = __builtin_readcyclecounter(); }
  // ---- stores: lane (a, p) owns outputs (cx, a, p) of every level index
  {
    T* const rec = out + (int64_t)be * out_estride;
#pragma unroll
    for (int rd = 0; rd < NRND; rd++) {
      const int a = 7 * rd + ar;
      if (7 * rd < Dm && lane < 63 && a < Dm) {
        const int t0 = a * PP + ep;                          // t of x offset 0; x offset cx adds cx * Dm * 9
#pragma unroll
        for (int cx = 0; cx < DMM; cx++) {
          if (cx < Dm) {
            if (paired) {
              T* op = rec + (t0 + cx * (Dm * PP)) * 2;
              if constexpr (HALF) {
                const mm_h2 v = {(_Float16)held[0][rd][cx], (_Float16)held[NL - 1][rd][cx]};
                *reinterpret_cast<mm_h2*>(op) = v;
              } else {
                typedef float f2v __attribute__((ext_vector_type(2)));
                const f2v v = {held[0][rd][cx], held[NL - 1][rd][cx]};
                *reinterpret_cast<f2v*>(op) = v;
              }
            } else {
#pragma unroll
              for (int l = 0; l < NL; l++)
                store_streamed(rec + (int64_t)(t0 + cx * (Dm * PP)) * out_lstride + LVF(l, out_offset), from_f32<T>(held[l][rd][cx]));
            }
          }
        }
      }
    }
  }
  if (!LDS1 && trace && lane == 0) {                 // per-wave cycle stamps (launch_mm prints the phase means)
    unsigned long long* t = trace + (size_t)slot * 8;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[0] = t_st[0]; t[1] = t_st[1]; t[2] = t_st[2]; t[3] = t_st[3]; t[4] = __builtin_readcyclecounter(); t[5] = (unsigned long long)(nt0 + 1000 * (ntot - nt0));
  }
  if constexpr (LDS1) wave_lds_fence();                    // (the next edge's records overwrite this one's result area / geometry)
  }   // turns
  if (LDS1 && trace && lane == 0) {
    unsigned long long* t = trace + ((size_t)blockIdx.x * NW + wv) * 8;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    t[0] = g_st[0]; t[1] = g_st[1]; t[2] = g_st[2]; t[3] = __builtin_amdgcn_s_memrealtime();
    t[4] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xf) << 32); t[5] = (unsigned long long)max(it_hi - it_lo, 0);
  }
#undef LVF
}
