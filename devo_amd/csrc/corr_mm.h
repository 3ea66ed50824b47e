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
#define DEVO_MM_WAVES32 3      // fp32 storage (the hi / lo patch and twice the bytes in flight: 168 registers)
#endif

// split of 8 fp32 values into fp16 hi and lo halves, x = hi + lo to 2^-22 (|lo| below the fp16 normal range keeps 2^-25 absolute):
// hi = rn(x) (v_cvt_pk_f16_f32, two values per instruction), lo = rn(x - hi) (v_fma_mix: fp32 arithmetic on the fp16 hi): 12 instructions
__device__ __forceinline__ void mm_split8(const v4u32 a, const v4u32 b, mm_h8& hi, mm_h8& lo) {
  unsigned h[4], l[4];
  const unsigned x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[j]) : "v"(x[2 * j]), "v"(x[2 * j + 1]));
    // (mixlo keeps the destination's upper half, which mixhi then overwrites: no initialisation needed)
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l[j]) : "v"(h[j]), "v"(x[2 * j]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l[j]) : "v"(h[j]), "v"(x[2 * j + 1]));
  }
  const v4u32 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
  hi = __builtin_bit_cast(mm_h8, hv);
  lo = __builtin_bit_cast(mm_h8, lv);
}

// 4 fp32 values -> fp16 (hi0..3 | lo0..3), x = hi + lo to 2^-22: the patch operand's stored form
__device__ __forceinline__ mm_h8 mm_split4(mm_f4 x) {
  unsigned h01, h23, l01, l23;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(x[0]), "v"(x[1]));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(x[2]), "v"(x[3]));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h01), "v"(x[0]));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h01), "v"(x[1]));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h23), "v"(x[2]));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h23), "v"(x[3]));
  const v4u32 r = {h01, h23, l01, l23};
  return __builtin_bit_cast(mm_h8, r);
}

// fmap1 [N][C][9] -> [N][9][C], the patch (B) operand of corr_fwd_mm_kernel: 16 contiguous bytes per (pixel, 4 | 8 channels).  fp32: every
// group of 4 channels is stored as fp16 (hi0..3 | lo0..3), x = hi + lo — the form the kernel multiplies (same 16 bytes).
template <typename T>
__global__ __launch_bounds__(256) void corr_patch_transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pt_lds[];
  T* s = reinterpret_cast<T*>(pt_lds);
  const int n = blockIdx.x;
  if (n >= N) return;
  const T* in = src + (int64_t)n * C * PP;
  T* o = dst + (int64_t)n * C * PP;
  for (int i = threadIdx.x; i < C * PP; i += 256) s[i] = in[i];
  __syncthreads();
  if constexpr (sizeof(T) == 2) {
    for (int i = threadIdx.x; i < C * PP; i += 256) { const int p = i / C, c = i - p * C; o[i] = s[c * PP + p]; }
  } else {
    for (int i = threadIdx.x; i < (C / 4) * PP; i += 256) {
      const int p = i / (C / 4), c4 = i - p * (C / 4);
      const mm_f4 x = {(float)s[(4 * c4) * PP + p], (float)s[(4 * c4 + 1) * PP + p], (float)s[(4 * c4 + 2) * PP + p], (float)s[(4 * c4 + 3) * PP + p]};
      reinterpret_cast<mm_h8*>(o)[i] = mm_split4(x);
    }
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

template <typename T, int RMAX, int NKS, int NL, int RFIX>   // NKS = C / 32 K steps per tile; NL = levels per wave; RFIX > 0: the radius is this constant
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? DEVO_MM_WAVES : DEVO_MM_WAVES32, sizeof(T) == 2 ? DEVO_MM_WAVES : DEVO_MM_WAVES32))) void corr_fwd_mm_kernel(
    const T* __restrict__ fmap1_t, CorrLevel lv0, CorrLevel lv1, int nlev, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, T* __restrict__ out, int BE, int E, int Np, int n2,
    int C, int64_t out_estride, int64_t out_lstride, int R_arg, const int* __restrict__ order, int heavy_only,
    unsigned long long* __restrict__ trace) {
  const int R = RFIX > 0 ? RFIX : R_arg;        // (DEVO's radius 3 and the stress configuration's 5 as constants: window sizes, loop bounds and
                                                //  the epilogue's guards fold away)
  constexpr bool HALF = sizeof(T) == 2;
  constexpr unsigned ESZ = sizeof(T);
  constexpr int LPS = HALF ? 1 : 2;                 // 16-byte loads per lane and K step
  constexpr int RT = HALF ? 4 : 2;                  // ring of tiles (RT - 1 tiles in flight ahead of the products)
  const int wlvl = (NL == 1 && nlev == 2) ? ((blockIdx.x >> 3) & 1) : 0;                      // wave-uniform
  const int wgid = (NL == 1 && nlev == 2) ? (((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) : blockIdx.x;
  const int nwg = (NL == 1 && nlev == 2) ? (gridDim.x >> 1) : gridDim.x;
  auto second = [&](int l) -> bool { return NL == 2 ? (l != 0) : (wlvl != 0); };   // does index l mean pyramid level 1?
#define LVF(l, F) (second(l) ? lv1.F : lv0.F)
  constexpr int DMAX = 2 * RMAX + 2;
  // Result area per level index, one of two layouts (as in corr_mfma.h):
  //   box layout  [p][BOXS]        slot s of pixel p at p * BOXS + s (boxes of <= CAP positions; BOXS = CAP + 4 so that the 8 lanes
  //                                of a ds_write_b128 group, 8 pixels x the same 4 slots, fall into different banks)
  //   raw windows [p][D*D + 1]     tap (a, c) of pixel p: window-by-window tiles (larger boxes)
  constexpr int CAP = RMAX <= 3 ? 128 : 256;
  constexpr int BOXS = CAP + 4;
  constexpr int RWIN_FLOATS = (PP * (DMAX * DMAX + 1) + 3) / 4 * 4;
  constexpr int RW_FLOATS = RWIN_FLOATS > PP * BOXS ? RWIN_FLOATS : PP * BOXS;
  __shared__ __attribute__((aligned(16))) float s_rawwin[NL * RW_FLOATS];
  __shared__ __attribute__((aligned(16))) float s_geo[NL][16][4];      // per (level index, pixel): dx, dy, tap (0, 0)'s index in the result area, row stride
  __shared__ int s_org[NL][PP][2];                                     // window origins (window-by-window tiles only)
  const int lane = threadIdx.x & 63;
  int slot = mm_plan_slot(order, BE, wgid, nwg);
  if (heavy_only) {
    slot = (int)blockIdx.x;
    if (slot >= (order ? min(max(order[BE], 0), BE) : 0)) return;
  }
  if (slot >= BE) return;                     // wave-uniform; no workgroup barriers in this kernel
  unsigned long long t_st[5] = {0, 0, 0, 0, 0};  // debug (DEVO_CORR_TRACE): cycle stamps of this wave's phases
  if (trace) t_st[0] = __builtin_readcyclecounter();
  const int be = order ? order[slot] : slot;
  const int D = 2 * R + 2, ntap = D * D;
  const int b = be / E, e = be - b * E;
  const int64_t pi = ii[e];
  const int64_t fj = jj[e];

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
  struct Geo { int xmin, ymin, bw, nslots, ntile; bool box_mode; float inv_bw; };      // wave-uniform
  auto make_geo = [&](int l) -> Geo {
    const int xmin = __builtin_amdgcn_readlane(vxmin, 15 + 16 * l), xmax = -__builtin_amdgcn_readlane(vxmax, 15 + 16 * l);
    const int ymin = __builtin_amdgcn_readlane(vymin, 15 + 16 * l), ymax = -__builtin_amdgcn_readlane(vymax, 15 + 16 * l);
    Geo g;
    g.xmin = xmin; g.ymin = ymin; g.bw = xmax - xmin + D;
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
  // fp16: lane piece = channels 8 kg .. + 7 of the step (32 channels);  fp32: two pieces, channels 4 kg .. + 3 and 16 + 4 kg .. + 3
  // (so that ONE instruction covers 16 consecutive channels = 64 bytes of each of its 16 positions).
  struct Pieces { unsigned step, lane0, second; };
  auto pieces_of = [&](int l) -> Pieces {
    const int sh = LVF(l, cb_shift);
    const unsigned bb = (unsigned)LVF(l, block_stride) * ESZ;
    auto piece = [&](unsigned c) -> unsigned { const unsigned blk = c >> sh; return blk * bb + (c - (blk << sh)) * ESZ; };
    return Pieces{piece(32), piece(HALF ? 8u * (unsigned)kg : 4u * (unsigned)kg), piece(16)};
  };

  // ---- B operand: the patch, channels 32 s + (8 kg .. + 7 | 4 kg .. + 3, 16 + 4 kg .. + 3) of pixel mi (columns >= 9: zeros)
  mm_h8 bh[NKS], bl[HALF ? 1 : NKS];
  {
    const T* __restrict__ f1 = fmap1_t + ((int64_t)b * Np + pi) * C * PP;           // [9][C]
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(f1), 0, (unsigned)(C * PP) * ESZ, 0x00020000);
    const unsigned boff = mi < PP ? (unsigned)(mi * C + (HALF ? 8 : 4) * kg) * ESZ : OFF_NONE;
#pragma unroll
    for (int s = 0; s < NKS; s++) {
      if constexpr (HALF) {
        bh[s] = __builtin_bit_cast(mm_h8, __builtin_amdgcn_raw_buffer_load_b128(rs1, boff, (unsigned)s * 64u, 0));
      } else {
        // fp32 patches arrive split already (devo_corr_patch_transpose): every 4 channels as 16 bytes of fp16 (hi0..3 | lo0..3)
        const v4u32 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs1, boff, (unsigned)s * 128u, 0);
        const v4u32 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs1, boff, (unsigned)s * 128u + 64u, 0);
        const v4u32 hv = {v0.x, v0.y, v1.x, v1.y}, lv = {v0.z, v0.w, v1.z, v1.w};
        bh[s] = __builtin_bit_cast(mm_h8, hv);
        bl[s] = __builtin_bit_cast(mm_h8, lv);
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
        mm_h8 ah, al;
        mm_split8(rb[r][s][0], rb[r][s][1], ah, al);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[s], acc, 0, 0, 0);       // small terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[s], acc, 0, 0, 0);
      }
    }
    return acc;
  };
  constexpr int TPL = CAP / 16;                              // tile slots per level of the static schedule
  if (all_box) {
    // ---- the usual case, as ONE straight line: NL x TPL tile slots whose level and tile number are compile-time constants (a slot
    //      beyond its level's last tile fetches nothing — every lane out of range — and stores zeros behind the box), so the loop has
    //      no branches, no scalar bookkeeping, immediate LDS offsets, and each level's descriptors are the kernel arguments themselves.
    __amdgpu_buffer_rsrc_t rsl[NL];
    unsigned lane_piece[NL], step_b[NL], second_b[NL];
    unsigned shb[NL], swb[NL];                               // byte strides of a row / a column
#pragma unroll
    for (int l = 0; l < NL; l++) {
      rsl[l] = frame_rsrc(l);
      const Pieces pc = pieces_of(l);
      lane_piece[l] = pc.lane0; step_b[l] = pc.step; second_b[l] = pc.second;
      shb[l] = (unsigned)LVF(l, s_h) * ESZ; swb[l] = (unsigned)LVF(l, s_w) * ESZ;
    }
    const float fmi = (float)mi + 0.5f;
    auto fetch_slot = [&](int ring, int i) {                 // (i is a constant after unrolling)
      const int l = i / TPL, t = i - l * TPL;
      const Geo& G = l ? g1 : g0;
      const int sl = t * 16 + mi;
      const int pyy = (int)((fmi + (float)(t * 16)) * G.inv_bw);      // exact: sl < 2^16, error margin 0.5 / bw
      const int gy = G.ymin + pyy, gx = G.xmin + (sl - __mul24(pyy, G.bw));
      const bool ok = sl < G.nslots && (unsigned)gy < (unsigned)LVF(l, H2) && (unsigned)gx < (unsigned)LVF(l, W2);
      const unsigned voff = ok ? (unsigned)gy * shb[l] + (unsigned)gx * swb[l] + lane_piece[l] : OFF_NONE;
#pragma unroll
      for (int s = 0; s < NKS; s++) {
        rb[ring][s][0] = __builtin_amdgcn_raw_buffer_load_b128(rsl[l], voff, (unsigned)s * step_b[l], 0);
        if constexpr (!HALF) rb[ring][s][1] = __builtin_amdgcn_raw_buffer_load_b128(rsl[l], voff, (unsigned)s * step_b[l] + second_b[l], 0);
      }
    };
    constexpr int NSLOT = NL * TPL;
#pragma unroll
    for (int r = 0; r < RT - 1; r++) { fetch_slot(r, r); __builtin_amdgcn_sched_barrier(0); }
    float* const dst = s_rawwin + mi * BOXS + 4 * kg;        // lane (n, rg): positions 4 rg .. + 3 of column n
#pragma unroll
    for (int i = 0; i < NSLOT; i++) {
      if (i + RT - 1 < NSLOT) fetch_slot((i + RT - 1) % RT, i + RT - 1);
      __builtin_amdgcn_sched_barrier(0);
      const mm_f4 acc = multiply(i % RT);
      if (mi < PP) *reinterpret_cast<mm_f4*>(dst + (i / TPL) * RW_FLOATS + (i % TPL) * 16) = acc;
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // ---- boxes beyond the result area (the plan's HEAVY class): a dynamic list of tiles, window by window where needed
    wave_lds_fence();                                        // (s_org)
    auto tile_off = [&](int tt) -> unsigned {                // byte offset of this lane's position in flat tile tt (OFF_NONE: nothing)
      const int l = (NL == 2 && tt >= nt0) ? 1 : 0;
      const Geo& G = l ? g1 : g0;
      const int t = l ? tt - nt0 : tt;
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
        gy = s_org[l][wp][1] + ta; gx = s_org[l][wp][0] + (tw - ta * D);
        listed = tw < ntap;
      }
      const bool ok = listed && tt < ntot && gy >= 0 && gy < LVF(l, H2) && gx >= 0 && gx < LVF(l, W2);
      return ok ? (unsigned)(gy * (int)LVF(l, s_h) + gx * (int)LVF(l, s_w)) * ESZ : OFF_NONE;
    };
    auto fetch = [&](int ring, int tt) {
      const int l = (NL == 2 && tt >= nt0) ? 1 : 0;
      const __amdgpu_buffer_rsrc_t rs = frame_rsrc(l);
      const Pieces pc = pieces_of(l);
      const unsigned off = tile_off(tt);
      const unsigned voff = off == OFF_NONE ? OFF_NONE : off + pc.lane0;
#pragma unroll
      for (int s = 0; s < NKS; s++) {
        rb[ring][s][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)s * pc.step, 0);
        if constexpr (!HALF) rb[ring][s][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)s * pc.step + pc.second, 0);
      }
    };
#pragma unroll
    for (int l = 0; l < NL; l++)                             // a level without tiles: its D x D box of zeros
      if ((l ? g1.nslots : g0.nslots) == 0)
        for (int i = lane; i < PP * BOXS; i += 64) s_rawwin[l * RW_FLOATS + i] = 0.0f;
#pragma unroll
    for (int r = 0; r < RT - 1; r++) { fetch(r, r); __builtin_amdgcn_sched_barrier(0); }
    for (int t0 = 0; t0 < ntot; t0 += RT) {
#pragma unroll
      for (int r = 0; r < RT; r++) {
        const int tt = t0 + r;
        fetch((r + RT - 1) % RT, tt + RT - 1);
        __builtin_amdgcn_sched_barrier(0);
        const mm_f4 acc = multiply(r);
        if (tt < ntot) {                                            // wave-uniform; no memory loads inside
          const int l = (NL == 2 && tt >= nt0) ? 1 : 0;
          const bool boxm = l ? g1.box_mode : g0.box_mode;
          const int t = l ? tt - nt0 : tt;
          float* rawwin = s_rawwin + l * RW_FLOATS;
          if (boxm) {
            if (mi < PP) *reinterpret_cast<mm_f4*>(rawwin + mi * BOXS + t * 16 + 4 * kg) = acc;
          } else {
            const int wp = t / WT, tw = (t - wp * WT) * 16 + 4 * kg;
            if (mi == wp) {
#pragma unroll
              for (int j = 0; j < 4; j++) if (tw + j < ntap) rawwin[wp * (ntap + 1) + tw + j] = acc[j];
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  wave_lds_fence();
  if (trace) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t_st[3] = __builtin_readcyclecounter(); }
  // ---- fused bilinear blend + axis swap + output permutation (correlation_kernel.cu:221-232).
  //      Output element (l, t), t = q * 9 + p with q = cx * Dm + a (cx = x offset: permute(0,1,3,2,4,5), a = y offset), goes to
  //      out[be * estride + t * lstride + offset(l)].
  const int Dm = D - 1;
  const bool paired = NL == 2 && out_lstride == 2 && lv0.out_offset == 0 && lv1.out_offset == 1 && (out_estride & 1) == 0 &&
                      (reinterpret_cast<uintptr_t>(out) & 7) == 0;
  if (paired) {
    // The standard stacked record (torch.stack([c0, c1], -1)): lane (a, p) = (lane / 9, lane % 9) does ROW a of pixel p's window at BOTH
    // levels — 2 x (Dm + 1) taps of rows a, a + 1 per level read once, Dm outputs per level in the reference's blend order — and
    // stores the two levels' values of one output as ONE 4- / 8-byte piece: the 63 lanes of a store write 63 consecutive pieces.
    constexpr int DMM = 2 * RMAX + 1;
    const int ar = (lane * 57) >> 9, p = lane - 9 * ar;                                  // lane / 9 for lane < 64
    const float4 ga = *reinterpret_cast<const float4*>(&s_geo[0][p][0]), gb = *reinterpret_cast<const float4*>(&s_geo[NL - 1][p][0]);
    float w[2][4];
    {
#pragma clang fp contract(off)
      w[0][0] = (1.0f - ga.x) * (1.0f - ga.y); w[0][1] = ga.x * (1.0f - ga.y); w[0][2] = (1.0f - ga.x) * ga.y; w[0][3] = ga.x * ga.y;
      w[1][0] = (1.0f - gb.x) * (1.0f - gb.y); w[1][1] = gb.x * (1.0f - gb.y); w[1][2] = (1.0f - gb.x) * gb.y; w[1][3] = gb.x * gb.y;
    }
    const int rs0 = __float_as_int(ga.w), rs1_ = __float_as_int(gb.w);
    T* const rec = out + (int64_t)be * out_estride;
    for (int a0 = 0; a0 < Dm; a0 += 7) {                     // (radius <= 3: one round)
      const int a = a0 + ar;
      if (lane < 63 && a < Dm) {
        const float* r0 = s_rawwin + __float_as_int(ga.z) + a * rs0;
        const float* r1 = s_rawwin + RW_FLOATS + __float_as_int(gb.z) + a * rs1_;
        float t0[DMM + 1], t1[DMM + 1], u0[DMM + 1], u1[DMM + 1];
#pragma unroll
        for (int c = 0; c <= DMM; c++) if (c <= Dm) { t0[c] = r0[c]; u0[c] = r0[rs0 + c]; t1[c] = r1[c]; u1[c] = r1[rs1_ + c]; }
        T* op = rec + (a * PP + p) * 2;
#pragma unroll
        for (int cx = 0; cx < DMM; cx++) {
          if (cx < Dm) {
            float o0, o1;
            {
#pragma clang fp contract(off)
              o0 = w[0][0] * t0[cx]; o0 = o0 + w[0][1] * t0[cx + 1]; o0 = o0 + w[0][2] * u0[cx]; o0 = o0 + w[0][3] * u0[cx + 1];
              o1 = w[1][0] * t1[cx]; o1 = o1 + w[1][1] * t1[cx + 1]; o1 = o1 + w[1][2] * u1[cx]; o1 = o1 + w[1][3] * u1[cx + 1];
            }
            if constexpr (HALF) {
              const mm_h2 v = {(_Float16)o0, (_Float16)o1};
              *reinterpret_cast<mm_h2*>(op + cx * (Dm * PP * 2)) = v;
            } else {
              typedef float f2v __attribute__((ext_vector_type(2)));
              const f2v v = {o0, o1};
              *reinterpret_cast<f2v*>(op + cx * (Dm * PP * 2)) = v;
            }
          }
        }
      }
    }
  } else {
    // any other output layout: one element per lane and round (corr_mfma.h's epilogue)
    constexpr int NPL = PP * NL, GRPS = 64 / NPL;               // 18 (p, l) pairs x 3 q's, or 9 x 7
    const int nq = Dm * Dm;
    const int grp = lane / NPL, pl = lane - grp * NPL;
    const int p = pl / NL, l = pl - p * NL;
    const bool active = grp < GRPS;
    const float4 gg = *reinterpret_cast<const float4*>(&s_geo[l][p][0]);
    float w00, w01, w10, w11;
    {
#pragma clang fp contract(off)
      w00 = (1.0f - gg.x) * (1.0f - gg.y); w01 = gg.x * (1.0f - gg.y); w10 = (1.0f - gg.x) * gg.y; w11 = gg.x * gg.y;   // blend4's factors
    }
    const int rstride = __float_as_int(gg.w);
    int q = grp;
    int cx = 0, a = q;
    while (a >= Dm) { a -= Dm; cx += 1; }
    const float* rw = s_rawwin + l * RW_FLOATS + __float_as_int(gg.z);
    T* op = out + (int64_t)be * out_estride + (int64_t)(q * PP + p) * out_lstride + LVF(l, out_offset);
    const int64_t ostep = (int64_t)(GRPS * PP) * out_lstride;
    for (int q0 = 0; q0 < nq; q0 += GRPS) {
      if (active && q < nq) {
        const float* r = rw + a * rstride + cx;
        float o;
        {
#pragma clang fp contract(off)
          o = w00 * r[0]; o = o + w01 * r[1]; o = o + w10 * r[rstride]; o = o + w11 * r[rstride + 1];
        }
        store_streamed(op, from_f32<T>(o));
      }
      op += ostep;
      q += GRPS; a += GRPS;
      if (a >= Dm) { a -= Dm; cx += 1; }
      if (a >= Dm) { a -= Dm; cx += 1; }
      while (a >= Dm) { a -= Dm; cx += 1; }
    }
  }
  if (trace && lane == 0) {                          // per-wave cycle stamps (launch_mm prints the phase means)
    unsigned long long* t = trace + (size_t)slot * 8;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[0] = t_st[0]; t[1] = t_st[1]; t[2] = t_st[2]; t[3] = t_st[3]; t[4] = __builtin_readcyclecounter(); t[5] = (unsigned long long)(nt0 + 1000 * (ntot - nt0));
  }
#undef LVF
}
