// altcorr lookup on the matrix cores, fp32 (included by corr.hip inside namespace devo).
//
// ONE WAVE PER EDGE, POSITION-centric: lane l owns one pixel of the union bounding box of the 9 patch pixels'
// (2r+2)^2 windows (64 positions per pass; a 10x10 box takes two passes) and reads that pixel's channels STRAIGHT
// from the pyramid into registers — no LDS tile, no staging stores, no tap reads.  The products run on
// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 outer products per instruction, block b = positions 4b..4b+3
// (the B operand is the lane's own feature value of one channel) against 4 patch pixels (the A operand).  The A
// operand of all 16 blocks is taken from ONE block of the A register (cbsz:4 abid:u), so a single register loaded
// as  lane (u, i) <- f1[k0 + u][4g + i]  feeds the 16 channels k0..k0+15 of pixel group g: the whole patch chunk
// is 3 registers per 16 channels and never touches LDS either.  Per channel and pass: 3 MFMAs (pixel groups
// {0-3} {4-7} {8}), 2 passes of the matrix pipe each, while the vector ALU stays free for the addressing and the
// other waves' epilogues.
// After a pass every lane scatters its 9 sums to the taps they are (position - window origin of pixel p, if inside
// the window) of the raw windows [p][a][c] in LDS (2.3 KB per wave = the only LDS of the kernel); the fused
// bilinear / permutation epilogue is the one of corr_fwd_cl_kernel.
// Boxes of any size work (ceil(npos / 64) passes); when the patch pixels are spread so far apart that the box
// holds more positions than the 9 windows together, the passes walk the windows one after the other instead.
#pragma once

#ifndef DEVO_MFMA_WPB
#define DEVO_MFMA_WPB 1
#endif
#ifndef DEVO_MFMA_RING
#define DEVO_MFMA_RING 4
#endif
#ifndef DEVO_MFMA_WAVES
#define DEVO_MFMA_WAVES 4
#endif
typedef float mfma_acc4 __attribute__((ext_vector_type(4)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

// Plan -> first edge slot of workgroup `gid` of `nitems`, each workgroup taking `per` consecutive slots (see
// corr_fwd_cl_kernel: workgroups with heavy edges first, the rest XCD-aware)
__device__ __forceinline__ int corr_plan_slot(const int* __restrict__ order, int BE, int gid, int nitems, int per) {
  const int nh = order ? (min(max(order[BE], 0), BE) + per - 1) / per : 0;
  if (gid < nh) return gid * per;
  const int xcd = gid & 7;
  auto heavy_on = [&](int x) -> int { return nh > x ? (nh - x + 7) >> 3 : 0; };
  auto total_on = [&](int x) -> int { return nitems > x ? (nitems - x + 7) >> 3 : 0; };
  int start = nh;
  for (int x = 0; x < xcd; x++) start += total_on(x) - heavy_on(x);
  return (start + (gid >> 3) - heavy_on(xcd)) * per;
}

#define DEVO_MFMA_STEP(U, BV)                                                         \
  acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, (BV), acc0, 4, (U), 0);              \
  acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, (BV), acc1, 4, (U), 0);              \
  acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, (BV), acc2, 4, (U), 0)

template <int RMAX, int NGR>       // NGR = C / 16 steps per pass (a multiple of 4)
__global__ __launch_bounds__(64 * DEVO_MFMA_WPB) __attribute__((amdgpu_waves_per_eu(DEVO_MFMA_WAVES, DEVO_MFMA_WAVES))) void corr_fwd_mfma_kernel(
    const float* __restrict__ fmap1, CorrLevel lv0, CorrLevel lv1, int nlev, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, float* __restrict__ out, int BE, int E, int Np, int n2,
    int C, int64_t out_estride, int64_t out_lstride, int R, const int* __restrict__ order,
    unsigned long long* __restrict__ trace) {
  const int lvl = (nlev == 2) ? ((blockIdx.x >> 3) & 1) : 0;                      // wave-uniform
  const int gid = (nlev == 2) ? (((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) : blockIdx.x;
  const int nitems = (nlev == 2) ? (gridDim.x >> 1) : gridDim.x;
  const CorrLevel& LV = lvl ? lv1 : lv0;
  const float* __restrict__ fmap2 = static_cast<const float*>(LV.fmap2);
  const int H2 = LV.H2, W2 = LV.W2;
  constexpr int DMAX = 2 * RMAX + 2;
  constexpr int WPB = DEVO_MFMA_WPB;          // waves = edges per workgroup: consecutive plan slots share a CU's L1
  constexpr int RW_FLOATS = (PP * (DMAX * DMAX + 1) + 3) / 4 * 4;
  __shared__ __attribute__((aligned(16))) float s_rawwin[WPB * RW_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* rawwin = s_rawwin + wave * RW_FLOATS;
  const int slot = corr_plan_slot(order, BE, gid, nitems, WPB) + wave;
  if (slot >= BE) return;                     // wave-uniform; no workgroup barriers in this kernel
  const unsigned long long t_start = trace ? __builtin_readcyclecounter() : 0ULL;
  const int be = order ? order[slot] : slot;
  const int D = 2 * R + 2, ntap = D * D;
  const int b = be / E, e = be - b * E;
  const int64_t pi = ii[e];                 // requested together with the coordinates (one round trip, not two)
  const int64_t fj = jj[e];

  // ---- geometry: lane p (< 9) owns patch pixel p
  // (the 18 coordinates come through the scalar cache: a vector load would queue behind the other waves' feature fetches)
  float px = 0.0f, py = 0.0f;
  {
    const float* __restrict__ ce = coords + (int64_t)be * (2 * PP);
    float cv[2 * PP];
#pragma unroll
    for (int p = 0; p < 2 * PP; p++) cv[p] = ce[p];             // 18 scalar loads in flight together (no branch around them)
#pragma unroll
    for (int p = 0; p < PP; p++) { px = (lane == p) ? cv[p] : px; py = (lane == p) ? cv[PP + p] : py; }
    px = px / LV.coord_div; py = py / LV.coord_div;
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
  const int bw = xmax - xmin + D;
  const long long npos_ll = (long long)bw * (ymax - ymin + D);
  const bool box_mode = npos_ll <= (long long)PP * ntap;        // else: the 9 windows one after the other
  const int nslots = box_mode ? (int)npos_ll : PP * ntap;
  const int npass = (nslots + 63) >> 6;

  const float* __restrict__ f1 = fmap1 + ((int64_t)b * Np + pi) * C * PP;           // [C][9]
  float* outp = out + (int64_t)be * out_estride + LV.out_offset;
  // Raw buffer descriptors (base, byte size): a lane whose offset is >= the size gets 0 WITHOUT a memory access — that is
  // how lanes beyond the box, positions outside the image and the fetches that run ahead past the last step are switched off
  // without a branch (a fetch behind a branch makes the compiler wait for ALL outstanding loads at the join).
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(fmap2 + (int64_t)b * LV.s_b + fj * LV.s_n), 0, LV.frame_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f1), 0, (unsigned)(C * PP) * 4u, 0x00020000);
  constexpr unsigned OFF_NONE = 0x80000000u;               // > any frame (the launcher guarantees < 2^31 bytes)

  // 16-byte piece q (4 channels) of the 16-channel step kg starts at channel c = 16 kg + 4 q: block c / cb, offset c % cb
  // (cb is a power of two, 1 << cb_shift; channels-last = one block of all channels, cb_shift = 30)
  const int cb_shift = LV.cb_shift;
  const unsigned block_bytes = (unsigned)LV.block_stride * (unsigned)sizeof(float);
  auto piece = [&](int c) -> unsigned {
    const unsigned blk = (unsigned)c >> cb_shift;
    return blk * block_bytes + ((unsigned)c - (blk << cb_shift)) * (unsigned)sizeof(float);
  };

  // A operand: lane (u, i) = (lane >> 2, lane & 3) holds f1[k0 + u][4g + i] (pixel 8 repeated in the unused rows of group 2).
  // The whole patch (NGR steps x 3 registers) stays in registers for all passes of the edge.
  const int au = lane >> 2, ai = lane & 3;
  const unsigned aoff0 = (unsigned)(au * PP + ai) * 4u, aoff1 = aoff0 + 16u, aoff2 = (unsigned)(au * PP + 8) * 4u;
  float pa[NGR][3];
#pragma unroll
  for (int g = 0; g < NGR; g++) {
    const unsigned ka = (unsigned)g * (16u * PP * 4u);
    pa[g][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, ka + aoff0, 0, 0));
    pa[g][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, ka + aoff1, 0, 0));
    pa[g][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, ka + aoff2, 0, 0));
  }
  __builtin_amdgcn_sched_barrier(0);        // patch loads first: the loop's s_waitcnt counts assume they are the oldest
  const float inv_bw = 1.0f / (float)bw;
  const int sh32 = (int)LV.s_h, sw32 = (int)LV.s_w;

  // position of this lane in pass ps: frame pixel (gy, gx), whether it is one of the box's positions, whether it lies
  // inside the image, and its byte offset inside the frame (OFF_NONE = fetch nothing)
  struct Pos { int gy, gx; bool listed, inside; unsigned off; };
  auto position = [&](int ps) -> Pos {
    Pos q;
    const int s = ps * 64 + lane;
    const int sc = min(s, nslots - 1);
    if (box_mode) {
      const int pyy = (int)(((float)sc + 0.5f) * inv_bw);     // exact: sc < 2^16, error margin 0.5 / bw
      q.gy = ymin + pyy; q.gx = xmin + (sc - pyy * bw);
    } else {
      const int wp = sc / ntap, t = sc - wp * ntap;
      const int ta = t / D;
      q.gy = __shfl(my_oy, wp) + ta; q.gx = __shfl(my_ox, wp) + (t - ta * D);
    }
    q.listed = s < nslots;
    q.inside = q.gy >= 0 && q.gy < H2 && q.gx >= 0 && q.gx < W2;
    q.off = (q.listed && q.inside) ? (unsigned)(q.gy * sh32 + q.gx * sw32) * (unsigned)sizeof(float) : OFF_NONE;
    return q;
  };

  // One step = 16 channels of one pass = 4 x 16 bytes of the lane's position.  Steps are fetched THREE ahead of their
  // products into a ring of four register sets, running on across pass boundaries (the fetches of the pass after the
  // last one are out of range = no memory access).  The step loop is fully unrolled: ring slots, patch registers and
  // the MFMAs' abid are all static.
  auto as_f4 = [](v4u32 v) -> float4 { float4 f; __builtin_memcpy(&f, &v, sizeof(f)); return f; };
  constexpr int RING = DEVO_MFMA_RING;       // register sets (steps in flight + the one being multiplied); divides NGR
  float4 rb[RING][4];
  auto fetch = [&](int slot, int g, unsigned off) {
#pragma unroll
    for (int q = 0; q < 4; q++) rb[slot][q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs2, off + piece(16 * g + 4 * q), 0, 0));
  };
  mfma_acc4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  Pos cur = position(0), nxt = position(1);
#pragma unroll
  for (int g = 0; g < RING - 1; g++) { fetch(g, g % NGR, cur.off); __builtin_amdgcn_sched_barrier(0); }
  if (trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_first = __builtin_readcyclecounter(); }
  for (int ps = 0; ps < npass; ps++) {
#pragma unroll
    for (int g = 0; g < NGR; g++) {
      fetch((g + RING - 1) % RING, (g + RING - 1) % NGR, (g + RING - 1 < NGR) ? cur.off : nxt.off);
      __builtin_amdgcn_sched_barrier(0);
      {
        const float a0 = pa[g][0], a1 = pa[g][1], a2 = pa[g][2];
        const float4 b0 = rb[g % RING][0], b1 = rb[g % RING][1], b2 = rb[g % RING][2], b3 = rb[g % RING][3];
        DEVO_MFMA_STEP(0, b0.x);  DEVO_MFMA_STEP(1, b0.y);  DEVO_MFMA_STEP(2, b0.z);  DEVO_MFMA_STEP(3, b0.w);
        DEVO_MFMA_STEP(4, b1.x);  DEVO_MFMA_STEP(5, b1.y);  DEVO_MFMA_STEP(6, b1.z);  DEVO_MFMA_STEP(7, b1.w);
        DEVO_MFMA_STEP(8, b2.x);  DEVO_MFMA_STEP(9, b2.y);  DEVO_MFMA_STEP(10, b2.z); DEVO_MFMA_STEP(11, b2.w);
        DEVO_MFMA_STEP(12, b3.x); DEVO_MFMA_STEP(13, b3.y); DEVO_MFMA_STEP(14, b3.z); DEVO_MFMA_STEP(15, b3.w);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- end of a pass.  Scatter: position (gy, gx) is tap (gy - oy[p], gx - ox[p]) of pixel p if that lies inside
    //      its window.  Out-of-image positions contribute exactly 0 (correlation_kernel.cu:136: within_bounds).
    if (cur.listed) {
      const float v[PP] = {acc0[0], acc0[1], acc0[2], acc0[3], acc1[0], acc1[1], acc1[2], acc1[3], acc2[0]};
#pragma unroll
      for (int p = 0; p < PP; p++) {
        const int ta = cur.gy - oy[p], tc = cur.gx - ox[p];
        if ((unsigned)ta < (unsigned)D && (unsigned)tc < (unsigned)D)
          rawwin[p * (ntap + 1) + ta * D + tc] = cur.inside ? v[p] : 0.0f;
      }
    }
    acc0 = mfma_acc4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0; acc2 = acc0;
    cur = nxt;
    nxt = position(ps + 2);
  }
  wave_lds_fence();
  if (trace) t_loop = __builtin_readcyclecounter();
  // ---- fused bilinear blend + axis swap + output permutation (correlation_kernel.cu:221-232), as in corr_fwd_cl_kernel
  const int Dm = D - 1;
  const int total = Dm * Dm * PP;
  {
    int q = lane / PP, p = lane - q * PP;
    int cx = q / Dm, a = q - cx * Dm;
    float* op = outp + (int64_t)lane * out_lstride;
    const int64_t ostep = 64 * out_lstride;
    for (int l0 = 0; l0 < total; l0 += 64) {        // wave-uniform trip count: the shuffles below need all lanes
      const float dxp = __shfl(my_dx, p), dyp = __shfl(my_dy, p);
      if (l0 + lane < total) {
        const float* r = rawwin + p * (ntap + 1) + a * D + cx;
        store_streamed(op, blend4(dxp, dyp, r[0], r[1], r[D], r[D + 1]));
      }
      op += ostep;
      p += 1; a += 7;
      if (p >= PP) { p -= PP; a += 1; }
      while (a >= Dm) { a -= Dm; cx += 1; }
    }
  }
  if (trace && lane == 0) {                          // debug: per-wave cycle stamps (see launch_staged)
    unsigned long long* t = trace + ((size_t)lvl * BE + slot) * 8;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[0] = t_start; t[1] = __builtin_readcyclecounter(); t[2] = (unsigned long long)npos_ll; t[3] = blockIdx.x;
    t[4] = t_geo; t[5] = t_first; t[6] = t_loop;
  }
}
#undef DEVO_MFMA_STEP
