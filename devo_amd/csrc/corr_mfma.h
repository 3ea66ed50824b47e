// altcorr lookup on the matrix cores, fp32 or fp16 storage, C = 128 (included by corr.hip inside namespace devo).
//
// ONE WAVE PER EDGE, POSITION-centric: lane l owns one pixel of the union bounding box of the 9 patch pixels'
// (2r+2)^2 windows (64 positions per pass; a 10x10 box takes two passes) and reads that pixel's channels STRAIGHT
// from the pyramid into registers — no LDS tile, no staging stores, no tap reads.  The products run on
// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 outer products per instruction, block b = positions 4b..4b+3
// (the B operand is the lane's own feature value of one channel) against 4 patch pixels (the A operand).  The A
// operand of all 16 blocks is taken from ONE block of the A register (cbsz:4 abid:u), so a single register loaded
// as  lane (u, i) <- f1[k0 + u][4g + i]  feeds the 16 channels k0..k0+15 of pixel group g: the whole patch is
// 24 registers, loaded once per edge, and never touches LDS either.  Per channel and pass: 3 MFMAs (pixel groups
// {0-3} {4-7} {8}), 2 passes of the matrix pipe each, while the vector ALU stays free for the addressing and the
// other waves' epilogues.
// After a pass every lane scatters its 9 sums to the taps they are (position - window origin of pixel p, if inside
// the window) of the raw windows [p][a][c] in LDS (2.3 KB per wave and level = the only LDS of the kernel); the fused
// bilinear / permutation epilogue is the one of corr_fwd_cl_kernel.
// Boxes of any size work (ceil(npos / 64) passes); when the patch pixels are spread so far apart that the box
// holds more positions than the 9 windows together, the passes walk the windows one after the other instead.
//
// NL = 2 (the fused pyramid lookup): the wave does BOTH pyramid levels of its edge, one after the other — the plan
// entry, the coordinates and the 24 patch registers are fetched once instead of twice (an edge's start-up, three
// dependent memory round trips, costs about as much as one level's channel loop), the feature fetches run on from the
// last pass of level 0 into the first pass of level 1, and the epilogue writes the two levels' interleaved outputs
// (torch.stack([c0, c1], -1)) as whole lines.
#pragma once

#ifndef DEVO_MFMA_RING
#define DEVO_MFMA_RING 4
#endif
#ifndef DEVO_MFMA_WAVES
#define DEVO_MFMA_WAVES 4
#endif
// Edges (= waves) per workgroup.  The waves of a workgroup never synchronise; putting the waves of EPW CONSECUTIVE plan slots
// into one workgroup only guarantees that spatial neighbours (the plan sorts by frame, 16-row band, 8-px column) run on the
// same CU at the same time, so that their overlapping boxes could meet in the CU's L1.  Measured on cfg2 (profiles/README.md,
// r02): 1 / 2 / 4 / 8 edges per workgroup = 191 / 208 / 211 / 222 us (fp32), 99 / 100 / 104 / 128 us (fp16) — the lost
// heavy-first schedule and the coarser dispatch cost more than the shared lines save, so the default stays 1.
#ifndef DEVO_MFMA_EPW
#define DEVO_MFMA_EPW 1
#endif
typedef float mfma_acc4 __attribute__((ext_vector_type(4)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
typedef _Float16 mfma_h4 __attribute__((ext_vector_type(4)));
// fp16 storage: v_mfma_f32_4x4x4_16b_f16 takes 4 channels per instruction (exact fp16 products, fp32 accumulation): a step is
// 32 channels = the same four 16-byte fetches per lane, 8 MFMAs per pixel group instead of 16, the patch is 12 registers.

// Plan -> edge slot of workgroup `gid` of `nitems` (see corr_fwd_cl_kernel: heavy edges first, the rest XCD-aware)
__device__ __forceinline__ int corr_plan_slot(const int* __restrict__ order, int BE, int gid, int nitems) {
  const int nh = order ? min(max(order[BE], 0), BE) : 0;
  if (gid < nh) return gid;
  const int xcd = gid & 7;
  auto heavy_on = [&](int x) -> int { return nh > x ? (nh - x + 7) >> 3 : 0; };
  auto total_on = [&](int x) -> int { return nitems > x ? (nitems - x + 7) >> 3 : 0; };
  int start = nh;
  for (int x = 0; x < xcd; x++) start += total_on(x) - heavy_on(x);
  return start + (gid >> 3) - heavy_on(xcd);
}

#define DEVO_MFMA_STEP_H(U, BV)                                                       \
  acc0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, (BV), acc0, 4, (U), 0);              \
  acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, (BV), acc1, 4, (U), 0);              \
  acc2 = __builtin_amdgcn_mfma_f32_4x4x4f16(a2, (BV), acc2, 4, (U), 0)
#define DEVO_MFMA_STEP(U, BV)                                                         \
  acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, (BV), acc0, 4, (U), 0);              \
  acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, (BV), acc1, 4, (U), 0);              \
  acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, (BV), acc2, 4, (U), 0)

template <typename T, int RMAX, int NGR, int NL>   // NGR = C / (16 | 32) steps per pass (a multiple of the ring); NL = levels per wave
__global__ __launch_bounds__(64 * DEVO_MFMA_EPW) __attribute__((amdgpu_waves_per_eu(DEVO_MFMA_WAVES, DEVO_MFMA_WAVES))) void corr_fwd_mfma_kernel(
    const T* __restrict__ fmap1, CorrLevel lv0, CorrLevel lv1, int nlev, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, T* __restrict__ out, int BE, int E, int Np, int n2,
    int C, int64_t out_estride, int64_t out_lstride, int R, const int* __restrict__ order,
    unsigned long long* __restrict__ trace, int heavy_only) {
  // NL == 1 with nlev == 2: the levels alternate in groups of 8 workgroups (see corr_fwd_cl_kernel); NL == 2: one
  // workgroup per edge does both.  lev(l) = the level this wave works on as its l-th.
  constexpr int EPW = DEVO_MFMA_EPW;
  const int wv = EPW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;      // this wave's edge inside the workgroup
  const int wlvl = (NL == 1 && nlev == 2) ? ((blockIdx.x >> 3) & 1) : 0;                      // wave-uniform
  const int wgid = (NL == 1 && nlev == 2) ? (((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) : blockIdx.x;
  const int nwg = (NL == 1 && nlev == 2) ? (gridDim.x >> 1) : gridDim.x;
  constexpr bool HALF = sizeof(T) == 2;
  // (a step is 32 channels of fp16 / 16 of fp32: four 16-byte pieces of a position)
  constexpr unsigned ESZ = sizeof(T);
  auto second = [&](int l) -> bool { return NL == 2 ? (l != 0) : (wlvl != 0); };   // does index l mean pyramid level 1?
#define LVF(l, F) (second(l) ? lv1.F : lv0.F)
  constexpr int DMAX = 2 * RMAX + 2;
  // Per level index one LDS area that holds the 9 sums of every box position until the epilogue, in one of two layouts:
  //   box layout  [p][BOXS]         position s of pixel p at p * BOXS + s: a pass stores its 9 sums with 9 immediate-offset
  //                                 ds_writes (boxes of <= BOXS positions: the usual case)
  //   raw windows [p][D*D + 1]      tap (a, c) of pixel p (large boxes, window-by-window passes): conditional scatter
  // The epilogue reads both as  base(p) + a * rowstride + c  (rowstride = box width / D).
  constexpr int BOXS = 128;
  constexpr int RWIN_FLOATS = (PP * (DMAX * DMAX + 1) + 3) / 4 * 4;
  constexpr int RW_FLOATS = RWIN_FLOATS > PP * BOXS ? RWIN_FLOATS : PP * BOXS;
  __shared__ __attribute__((aligned(16))) float s_rawwin_all[EPW][NL * RW_FLOATS];
  float* const s_rawwin = s_rawwin_all[wv];
  const int lane = threadIdx.x & 63;
  // EPW == 1: the plan's heavy-first / XCD-aware slot map.  EPW > 1: workgroup g (on XCD g % 8) takes EPW consecutive slots of
  // its XCD's contiguous share of the plan.
  int slot;
  if (EPW == 1) slot = corr_plan_slot(order, BE, wgid, nwg);
  else { const int per = (nwg + 7) >> 3; slot = (((wgid & 7) * per + (wgid >> 3)) * EPW) + wv; if ((wgid >> 3) >= per) slot = BE; }
  if (heavy_only) {                           // only the plan's HEAVY class (slots 0 .. order[BE] - 1)
    slot = (int)blockIdx.x * EPW + wv;
    if (slot >= (order ? min(max(order[BE], 0), BE) : 0)) return;
  }
  if (slot >= BE) return;                     // wave-uniform; no workgroup barriers in this kernel
  const unsigned long long t_start = trace ? __builtin_readcyclecounter() : 0ULL;
  const int be = order ? order[slot] : slot;
  const int D = 2 * R + 2, ntap = D * D;
  const int b = be / E, e = be - b * E;
  const int64_t pi = ii[e];                 // requested together with the coordinates (one round trip, not two)
  const int64_t fj = jj[e];

  // ---- geometry: lane p (< 9) owns patch pixel p
  // (the 18 coordinates come through the scalar cache: a vector load would queue behind the other waves' feature fetches)
  float cpx = 0.0f, cpy = 0.0f;             // undivided coordinates of this lane's patch pixel
  {
    const float* __restrict__ ce = coords + (int64_t)be * (2 * PP);
    float cv[2 * PP];
#pragma unroll
    for (int p = 0; p < 2 * PP; p++) cv[p] = ce[p];             // 18 scalar loads in flight together (no branch around them)
#pragma unroll
    for (int p = 0; p < PP; p++) {                              // scalar -> lane p (one instruction each, no compare + select)
      asm("v_writelane_b32 %0, %1, %2" : "+v"(cpx) : "s"(cv[p]), "n"(p));
      asm("v_writelane_b32 %0, %1, %2" : "+v"(cpy) : "s"(cv[PP + p]), "n"(p));
    }
  }
  unsigned long long t_geo = 0, t_first = 0, t_loop = 0;
  if (trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_geo = __builtin_readcyclecounter(); }
  // scaled coordinates of the lane's pixel per level index.  coords / div is the reference's true division; for a power of
  // two (1 and 4 in DEVO) x * (1 / div) is the same correctly rounded value, and it spares ~12 vector instructions per
  // division (every vector instruction of this kernel competes with the MFMAs for the SIMD's issue slot).
  float qx[NL], qy[NL];
  {
    auto pow2 = [](float d) -> bool { return (__float_as_uint(d) & 0x807fffffu) == 0u && d >= 1.0f; };
    bool all_pow2 = true;
#pragma unroll
    for (int l = 0; l < NL; l++) all_pow2 = all_pow2 && pow2(LVF(l, coord_div));
    if (all_pow2) {                                                   // wave-uniform
#pragma unroll
      for (int l = 0; l < NL; l++) { const float iv = 1.0f / LVF(l, coord_div); qx[l] = cpx * iv; qy[l] = cpy * iv; }
    } else {
#pragma unroll
      for (int l = 0; l < NL; l++) { const float dv = LVF(l, coord_div); qx[l] = cpx / dv; qy[l] = cpy / dv; }
    }
  }
  auto origin_x = [&](int l) -> int { return floor_to_int((NL == 2 && l) ? qx[NL - 1] : qx[0]) - R; };
  auto origin_y = [&](int l) -> int { return floor_to_int((NL == 2 && l) ? qy[NL - 1] : qy[0]) - R; };
  struct Geo { int xmin, ymin, bw, nslots, npass; bool box_mode, boxlay; float inv_bw; };      // wave-uniform
  // min / max over lanes 0..8 inside the first 16-lane row with 4 DPP row shifts each (hipcc turns min/max chains over
  // readlane results into vector min3/max3 plus moves: 90 vector instructions per level instead of 24)
  auto row_min = [&](int v) -> int {
    v = (lane < PP) ? v : 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x111, 0xf, 0xf, false));     // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x112, 0xf, 0xf, false));     // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x114, 0xf, 0xf, false));     // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x118, 0xf, 0xf, false));     // row_shr:8
    return __builtin_amdgcn_readlane(v, 15);
  };
  auto make_geo = [&](int l) -> Geo {
    const int mox = origin_x(l), moy = origin_y(l);
    const int xmin = row_min(mox), xmax = -row_min(-mox), ymin = row_min(moy), ymax = -row_min(-moy);
    Geo g;
    g.xmin = xmin; g.ymin = ymin; g.bw = xmax - xmin + D;
    const long long npos_ll = (long long)g.bw * (ymax - ymin + D);
    g.box_mode = npos_ll <= (long long)PP * ntap;        // else: the 9 windows one after the other
    g.nslots = g.box_mode ? (int)npos_ll : PP * ntap;
    // the whole box outside the frame (11 % of cfg2's edges at both levels, 18 % at level 0): every tap is 0 (correlation_kernel.cu:136)
    // — no passes at all for this level, the epilogue writes zeros
    if (xmax + D <= 0 || ymax + D <= 0 || xmin >= LVF(l, W2) || ymin >= LVF(l, H2)) { g.nslots = 0; g.box_mode = true; }
    g.npass = (g.nslots + 63) >> 6;
    g.boxlay = g.box_mode && g.nslots <= BOXS;
    g.inv_bw = __builtin_amdgcn_rcpf((float)g.bw);       // 1 ulp is plenty for the row / column split below
    return g;
  };
  const Geo g0 = make_geo(0);
  const Geo g1 = (NL == 2) ? make_geo(1) : g0;
  // window origins of the 9 pixels per level index -> LDS (the scatter reads them back as broadcasts: LDS instructions,
  // not vector ALU — every vector instruction of this kernel competes with the MFMAs for the SIMD's issue slot);
  // sub-pixel fractions -> one register pair: lane p holds level index 0's, lane 16 + p level index 1's
  __shared__ int s_org_all[EPW][NL][PP][2];
  int (*const s_org)[PP][2] = s_org_all[wv];
  float fdx, fdy;
  {
#pragma unroll
    for (int l = 0; l < NL; l++) if (lane < PP) { s_org[l][lane][0] = origin_x(l); s_org[l][lane][1] = origin_y(l); }
    const int src = lane & 15;                                   // lanes 16.. take the fractions of lane - 16 at level index 1
    const bool hi = NL == 2 && lane >= 16;
    const float ax = __shfl(qx[0], src), ay = __shfl(qy[0], src), bx = __shfl(qx[NL - 1], src), by = __shfl(qy[NL - 1], src);
    const float qx_ = hi ? bx : ax, qy_ = hi ? by : ay;
    fdx = qx_ - floorf(qx_); fdy = qy_ - floorf(qy_);
  }
  // where the epilogue finds tap (0, 0) of pixel p (lane p: level index 0, lane 16 + p: level index 1), and the row stride
  int fbase;
  {
    const int p_ = min(lane & 15, PP - 1), l_ = (NL == 2 && lane >= 16) ? 1 : 0;
    const Geo& G = l_ ? g1 : g0;
    fbase = G.boxlay ? p_ * BOXS + (s_org[l_][p_][1] - G.ymin) * G.bw + (s_org[l_][p_][0] - G.xmin) : p_ * (ntap + 1);
  }
  const int np0 = g0.npass, nseg = (NL == 2) ? np0 + g1.npass : np0;     // segments = passes of level 0, then of level 1
  auto seg_level = [&](int s) -> int { return (NL == 2 && s >= np0) ? 1 : 0; };

  // Raw buffer descriptors (base, byte size): a lane whose offset is >= the size gets 0 WITHOUT a memory access — that is
  // how lanes beyond the box, positions outside the image and the fetches that run ahead past the last step are switched off
  // without a branch (a fetch behind a branch makes the compiler wait for ALL outstanding loads at the join).
  auto frame_rsrc = [&](int l) -> __amdgpu_buffer_rsrc_t {
    const T* base = static_cast<const T*>(LVF(l, fmap2)) + (int64_t)b * LVF(l, s_b) + fj * LVF(l, s_n);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, LVF(l, frame_bytes), 0x00020000);
  };
  const T* __restrict__ f1 = fmap1 + ((int64_t)b * Np + pi) * C * PP;           // [C][9]
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(f1), 0, (unsigned)(C * PP) * ESZ, 0x00020000);
  constexpr unsigned OFF_NONE = 0x80000000u;               // > any frame (the launcher guarantees < 2^31 bytes)

  // A operand: lane (u, i) = (lane >> 2, lane & 3) holds f1[k0 + u][4g + i] (pixel 8 repeated in the unused rows of group 2).
  // The whole patch (NGR steps x 3 registers) stays in registers for all passes (and both levels) of the edge.
  const int au = lane >> 2, ai = lane & 3;
  const unsigned aoff0 = (unsigned)(au * PP + ai) * 4u, aoff1 = aoff0 + 16u, aoff2 = (unsigned)(au * PP + 8) * 4u;
  float pa[HALF ? 1 : NGR][3];                 // fp32: lane (u, i) <- f1[16 g + u][4 grp + i]
  mfma_h4 ph[HALF ? NGR / 2 : 1][3];            // fp16: lane (u, i) <- f1[64 h + 4 u .. + 3][4 grp + i] (four channels per MFMA)
  if constexpr (!HALF) {
#pragma unroll
    for (int g = 0; g < NGR; g++) {
      const unsigned ka = (unsigned)g * (16u * PP * 4u);        // in the scalar offset operand: no vector adds
      pa[g][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, aoff0, ka, 0));
      pa[g][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, aoff1, ka, 0));
      pa[g][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, aoff2, ka, 0));
    }
  } else {
    const unsigned hpix[3] = {(unsigned)ai, (unsigned)(4 + ai), 8u};
#pragma unroll
    for (int h = 0; h < NGR / 2; h++)
#pragma unroll
      for (int grp = 0; grp < 3; grp++) {
        unsigned short r4[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
          r4[r] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs1, (unsigned)((4 * au + r) * PP + hpix[grp]) * 2u, (unsigned)h * (64u * PP * 2u), 0);
        __builtin_memcpy(&ph[h][grp], r4, sizeof(r4));
      }
  }
  __builtin_amdgcn_sched_barrier(0);        // patch loads first: the loop's s_waitcnt counts assume they are the oldest

  // position of this lane in segment s: frame pixel (gy, gx), whether it is one of the box's positions, whether it lies
  // inside the image, and its byte offset inside the frame (OFF_NONE = fetch nothing; every segment past the last one)
  struct Pos { int gy, gx; bool listed, inside; unsigned off; };
  auto position = [&](int sg) -> Pos {
    const int l = seg_level(sg);
    const Geo& G = l ? g1 : g0;
    const int ps = l ? sg - np0 : sg;
    Pos q;
    const int s = ps * 64 + lane;
    const int sc = max(min(s, G.nslots - 1), 0);
    if (G.box_mode) {
      const int pyy = (int)(((float)sc + 0.5f) * G.inv_bw);     // exact: sc < 2^16, error margin 0.5 / bw
      q.gy = G.ymin + pyy; q.gx = G.xmin + (sc - pyy * G.bw);
    } else {
      const int wp = sc / ntap, t = sc - wp * ntap;
      const int ta = t / D;
      q.gy = s_org[l][wp][1] + ta; q.gx = s_org[l][wp][0] + (t - ta * D);          // (window origins live in LDS)
    }
    q.listed = s < G.nslots && sg < nseg;
    q.inside = q.gy >= 0 && q.gy < LVF(l, H2) && q.gx >= 0 && q.gx < LVF(l, W2);
    q.off = (q.listed && q.inside) ? (unsigned)(q.gy * (int)LVF(l, s_h) + q.gx * (int)LVF(l, s_w)) * ESZ : OFF_NONE;
    return q;
  };

  // One step = 16 channels of one pass = 4 x 16 bytes of the lane's position.  Steps are fetched RING-1 ahead of their
  // products into a ring of register sets, running on across pass and level boundaries (the fetches of the segment after
  // the last one are out of range = no memory access).  The step loop is fully unrolled: ring slots, patch registers and
  // the MFMAs' abid are all static.
  // 16-byte piece q (4 channels) of step g starts at channel c = 16 g + 4 q: block c / cb, offset c % cb
  // (cb = 1 << cb_shift; channels-last = one block of all channels, cb_shift = 30)
  auto as_f4 = [](v4u32 v) -> float4 { float4 f; __builtin_memcpy(&f, &v, sizeof(f)); return f; };
  constexpr int RING = DEVO_MFMA_RING;
  float4 rb[RING][4];
  // Byte offset of piece q of step g = g * G16 + d[q]: linear in g for the layouts the launcher lets through (channel
  // blocks of 4, 8 or 16 channels, or channels-last), so a step costs 4 scalar adds besides its 4 loads.
  struct Pieces { unsigned g16, d1, d2, d3; };       // byte offsets: one step further / pieces 1..3 of a step
  auto pieces_of = [&](int l) -> Pieces {
    const int sh = LVF(l, cb_shift);
    const unsigned bb = (unsigned)LVF(l, block_stride) * ESZ;
    constexpr unsigned PCH = 16 / ESZ;                            // channels per 16-byte piece (4 | 8)
    auto piece = [&](unsigned c) -> unsigned { const unsigned blk = c >> sh; return blk * bb + (c - (blk << sh)) * ESZ; };
    return Pieces{piece(4 * PCH), piece(PCH), piece(2 * PCH), piece(3 * PCH)};
  };
  auto fetch = [&](int ring, int g, unsigned off, __amdgpu_buffer_rsrc_t rs, const Pieces& pc) {
    // lane offset in the vector operand (the range check looks at it alone), piece offset in the scalar one: no vector add
    const unsigned base = (unsigned)g * pc.g16;
    rb[ring][0] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs, off, base, 0));
    rb[ring][1] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs, off, base + pc.d1, 0));
    rb[ring][2] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs, off, base + pc.d2, 0));
    rb[ring][3] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs, off, base + pc.d3, 0));
  };
  mfma_acc4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  Pos cur = position(0), nxt = position(1);
  {
    const int lf = seg_level(0);                           // (level 0 has no passes when its box misses the frame)
    const __amdgpu_buffer_rsrc_t r0 = frame_rsrc(lf);
    const Pieces pc0 = pieces_of(lf);
#pragma unroll
    for (int g = 0; g < RING - 1; g++) { fetch(g, g % NGR, cur.off, r0, pc0); __builtin_amdgcn_sched_barrier(0); }
  }
  if (trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_first = __builtin_readcyclecounter(); }
  for (int sg = 0; sg < nseg; sg++) {
    const int lc = seg_level(sg), ln = seg_level(min(sg + 1, nseg - 1));         // wave-uniform
    const __amdgpu_buffer_rsrc_t rsc = frame_rsrc(lc), rsn = frame_rsrc(ln);
    const Pieces pcc = pieces_of(lc), pcn = pieces_of(ln);
#pragma unroll
    for (int g = 0; g < NGR; g++) {
      if (g + RING - 1 < NGR) fetch((g + RING - 1) % RING, g + RING - 1, cur.off, rsc, pcc);       // (static condition)
      else fetch((g + RING - 1) % RING, g + RING - 1 - NGR, nxt.off, rsn, pcn);
      __builtin_amdgcn_sched_barrier(0);
      {
        const float4 b0 = rb[g % RING][0], b1 = rb[g % RING][1], b2 = rb[g % RING][2], b3 = rb[g % RING][3];
        if constexpr (!HALF) {
          const float a0 = pa[g][0], a1 = pa[g][1], a2 = pa[g][2];
          DEVO_MFMA_STEP(0, b0.x);  DEVO_MFMA_STEP(1, b0.y);  DEVO_MFMA_STEP(2, b0.z);  DEVO_MFMA_STEP(3, b0.w);
          DEVO_MFMA_STEP(4, b1.x);  DEVO_MFMA_STEP(5, b1.y);  DEVO_MFMA_STEP(6, b1.z);  DEVO_MFMA_STEP(7, b1.w);
          DEVO_MFMA_STEP(8, b2.x);  DEVO_MFMA_STEP(9, b2.y);  DEVO_MFMA_STEP(10, b2.z); DEVO_MFMA_STEP(11, b2.w);
          DEVO_MFMA_STEP(12, b3.x); DEVO_MFMA_STEP(13, b3.y); DEVO_MFMA_STEP(14, b3.z); DEVO_MFMA_STEP(15, b3.w);
        } else {
          // channels 32 g + 4 q .. + 3 = block (8 (g & 1) + q) of patch register pair g / 2; B = the q-th 8 bytes of the step
          const mfma_h4 a0 = ph[g >> 1][0], a1 = ph[g >> 1][1], a2 = ph[g >> 1][2];
          mfma_h4 bq[8];
          __builtin_memcpy(&bq[0], &b0, 16); __builtin_memcpy(&bq[2], &b1, 16); __builtin_memcpy(&bq[4], &b2, 16); __builtin_memcpy(&bq[6], &b3, 16);
          if ((g & 1) == 0) {                    // (abid must be a literal: the unrolled loop folds this branch)
            DEVO_MFMA_STEP_H(0, bq[0]); DEVO_MFMA_STEP_H(1, bq[1]); DEVO_MFMA_STEP_H(2, bq[2]); DEVO_MFMA_STEP_H(3, bq[3]);
            DEVO_MFMA_STEP_H(4, bq[4]); DEVO_MFMA_STEP_H(5, bq[5]); DEVO_MFMA_STEP_H(6, bq[6]); DEVO_MFMA_STEP_H(7, bq[7]);
          } else {
            DEVO_MFMA_STEP_H(8, bq[0]); DEVO_MFMA_STEP_H(9, bq[1]); DEVO_MFMA_STEP_H(10, bq[2]); DEVO_MFMA_STEP_H(11, bq[3]);
            DEVO_MFMA_STEP_H(12, bq[4]); DEVO_MFMA_STEP_H(13, bq[5]); DEVO_MFMA_STEP_H(14, bq[6]); DEVO_MFMA_STEP_H(15, bq[7]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- end of a pass.  Scatter: position (gy, gx) is tap (gy - oy[p], gx - ox[p]) of pixel p if that lies inside
    //      its window.  Out-of-image positions contribute exactly 0 (correlation_kernel.cu:136: within_bounds).
    {
      float* rawwin = s_rawwin + lc * RW_FLOATS;
      const int (*org)[2] = s_org[lc];
      const bool boxlay = lc ? g1.boxlay : g0.boxlay;                       // wave-uniform
      const int ps = lc ? sg - np0 : sg;
      if (cur.listed) {
        const float v[PP] = {acc0[0], acc0[1], acc0[2], acc0[3], acc1[0], acc1[1], acc1[2], acc1[3], acc2[0]};
        if (boxlay) {
          float* dst = rawwin + ps * 64 + lane;
#pragma unroll
          for (int p = 0; p < PP; p++) dst[p * BOXS] = cur.inside ? v[p] : 0.0f;
        } else {
#pragma unroll
          for (int p = 0; p < PP; p++) {
            const int ta = cur.gy - org[p][1], tc = cur.gx - org[p][0];
            if ((unsigned)ta < (unsigned)D && (unsigned)tc < (unsigned)D)
              rawwin[p * (ntap + 1) + ta * D + tc] = cur.inside ? v[p] : 0.0f;
          }
        }
      }
    }
    acc0 = mfma_acc4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0; acc2 = acc0;
    cur = nxt;
    nxt = position(sg + 2);
  }
  wave_lds_fence();
  if (trace) t_loop = __builtin_readcyclecounter();
  // ---- fused bilinear blend + axis swap + output permutation (correlation_kernel.cu:221-232).
  //      Output element (l, t): level index l, t = q * 9 + p with q = cx * Dm + a (cx = x offset: permute(0,1,3,2,4,5),
  //      a = y offset), p = i0*3+j0, goes to  out[be * estride + t * lstride + offset(l)].  A lane keeps ITS (p, l) for the
  //      whole epilogue — blend weights, result-area base and row stride are lane constants, no shuffles in the loop — and
  //      walks q = grp, grp + GRPS, ...; the 9 * NL * GRPS active lanes of a round write consecutive addresses of the standard
  //      stacked layout (lstride NL, offsets 0 / 1).
  {
    const int Dm = D - 1, nq = Dm * Dm;
    constexpr int NPL = PP * NL, GRPS = 64 / NPL;               // 18 (p, l) pairs x 3 q's, or 9 x 7
    const int grp = lane / NPL, pl = lane - grp * NPL;
    const int p = pl / NL, l = pl - p * NL;
    const bool active = grp < GRPS;
    const float dxp = __shfl(fdx, p + 16 * l), dyp = __shfl(fdy, p + 16 * l);
    const int base = __shfl(fbase, p + 16 * l);
    float w00, w01, w10, w11;
    {
#pragma clang fp contract(off)
      w00 = (1.0f - dxp) * (1.0f - dyp); w01 = dxp * (1.0f - dyp); w10 = (1.0f - dxp) * dyp; w11 = dxp * dyp;   // blend4's factors
    }
    const int rstride = (l ? g1.boxlay : g0.boxlay) ? (l ? g1.bw : g0.bw) : D;       // per lane (l is)
    const bool lvl_live = (l ? g1.nslots : g0.nslots) > 0;                          // (a level without passes left its result area untouched)
    int q = grp;
    int cx = 0, a = q;
    while (a >= Dm) { a -= Dm; cx += 1; }           // (grp < 7: only windows smaller than 7 x 7 take a turn)
    const float* rw = s_rawwin + l * RW_FLOATS + base;
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
        store_streamed(op, from_f32<T>(lvl_live ? o : 0.0f));
      }
      op += ostep;
      q += GRPS; a += GRPS;
      if (a >= Dm) { a -= Dm; cx += 1; }             // (two plain selects cover every radius >= 3; the loop is for tiny windows)
      if (a >= Dm) { a -= Dm; cx += 1; }
      while (a >= Dm) { a -= Dm; cx += 1; }
    }
  }
  if (trace && lane == 0) {                          // debug: per-wave cycle stamps (see launch_staged)
    unsigned long long* t = trace + ((size_t)wlvl * BE + slot) * 8;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[0] = t_start; t[1] = __builtin_readcyclecounter(); t[2] = (unsigned long long)g0.nslots; t[3] = blockIdx.x;
    t[4] = t_geo; t[5] = t_first; t[6] = t_loop;
  }
#undef LVF
}
#undef DEVO_MFMA_STEP
#undef DEVO_MFMA_STEP_H
