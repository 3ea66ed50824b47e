// altcorr lookup, fp32 / r <= 3 fast path with LDS-direct staging (included by corr.hip inside namespace devo).
//
// ONE WAVE PER EDGE, no workgroup barriers, TAP-centric like corr_fwd_cl_kernel (lane t owns one tap of the
// (2r+2)^2 window of all 9 patch pixels, 9 accumulators, patch operand broadcast inside the FMA by DPP), but the
// box tile never passes through registers:
//   * the union box of the 9 windows is fetched 4 channels (= one 16-byte slot per position) at a time with
//     global_load_lds_dwordx4 straight into a double-buffered LDS tile — no staging VGPRs, no ds_write pass, no
//     exec-masked load/store pairs; out-of-image positions and padding lanes read a zero buffer instead, so every
//     lane is always active;
//   * the tile is lane-linear (slot s = row * PT + column; PT = box width rounded up to an even number that is
//     not a multiple of 16).  The lane -> tap map is chosen per pitch so that every 16-lane group of a
//     ds_read_b128 ({0-3,12-15,20-27}, ...) covers two tap rows whose slots differ by 8 (mod 16): rows (a, a+d)
//     with d = 8 / lowbit(PT mod 16) — conflict-free for every box width without padding the rows to 8 mod 16;
//   * the patch features [C][9] are transposed once per edge into LDS as [pixel][C+4] (one ds_read_b128 per
//     4-channel step gives the lane's DPP source row).
// Boxes that do not fit the tile are split greedily into pixel groups whose boxes fit (each group re-streams
// the channels); the group path selects pixels with wave-uniform branches.
#pragma once

constexpr int DMA_SLOTS = 160;                  // 16-byte slots per tile buffer: 2.5 wave-wide loads
constexpr int DMA_BUF_FLOATS = DMA_SLOTS * 4;
constexpr int DMA_ZERO_BYTES = 16384;           // >= 4*C: the zero source is walked like a feature column
__device__ uint4 g_corr_zero[DMA_ZERO_BYTES / 16];   // never written: stays zero-initialised

__device__ __forceinline__ void dma16(const char* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Plan -> edge slot of this workgroup (one wave per workgroup).  With a plan the first `nh` workgroups take the
// HEAVY edges (multi-pass boxes) so that the longest work items start first; the others take the
// (frame, row-band)-sorted edges XCD-aware: workgroup g runs on XCD g % 8 (observed dispatch order) and every XCD
// owns one contiguous slice of the sorted list, so that its private L2 sees each feature row about once.
// Bijective for any grid size.
__device__ __forceinline__ int corr_edge_slot(const int* __restrict__ order, int BE) {
  const int g = blockIdx.x;
  const int nh = order ? min(max(order[BE], 0), BE) : 0;
  if (g < nh) return g;
  const int nwg = gridDim.x, xcd = blockIdx.x & 7;
  auto heavy_on = [&](int x) -> int { return nh > x ? (nh - x + 7) >> 3 : 0; };          // heavy workgroups on XCD x
  auto total_on = [&](int x) -> int { return nwg > x ? (nwg - x + 7) >> 3 : 0; };        // all workgroups on XCD x
  int start = nh;
  for (int x = 0; x < xcd; x++) start += total_on(x) - heavy_on(x);
  return start + (blockIdx.x >> 3) - heavy_on(xcd);
}

// Row pitch of a box of width w, in slots.  Single-box edges: even and never a multiple of 16.  Split boxes: = 2 (mod 4),
// so that every pixel group of the edge gets the same lane -> tap map (d = 4) and the accumulators of different groups
// belong to the same taps.
__device__ __forceinline__ int dma_pitch(int w, bool split) {
  if (split) return w + ((2 - w) & 3);
  int pt = (w + 1) & ~1;
  if ((pt & 15) == 0) pt += 2;
  return pt;
}

__global__ __launch_bounds__(64) void corr_fwd_dma_kernel(
    const float* __restrict__ fmap1, const float* __restrict__ fmap2, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, float* __restrict__ out, int BE, int E, int Np, int n2,
    int C, int H2, int W2, int64_t s_b, int64_t s_n, int64_t s_h, int64_t s_w, int64_t out_estride,
    int64_t out_lstride, int64_t out_offset, int R, const int* __restrict__ order, unsigned long long* __restrict__ trace,
    float coord_div) {
  extern __shared__ __attribute__((aligned(16))) float dma_smem[];
  float* const buf0 = dma_smem;
  float* const buf1 = dma_smem + DMA_BUF_FLOATS;
  float* const f1t = dma_smem + 2 * DMA_BUF_FLOATS;          // [9][C + 4]
  const int F1S = C + 4;
  const int lane = threadIdx.x;
  const int slot = corr_edge_slot(order, BE);
  if (slot >= BE) return;                                    // wave-uniform; no barriers in this kernel
  const unsigned long long t_start = trace ? __builtin_readcyclecounter() : 0ULL;
  const int be = order ? order[slot] : slot;
  const int D = 2 * R + 2;
  const int b = be / E, e = be - b * E;

  // ---- geometry: lane p (< 9) owns patch pixel p
  float px = 0.0f, py = 0.0f;
  if (lane < PP) {
    px = coords[((int64_t)be * 2 + 0) * PP + lane] / coord_div;
    py = coords[((int64_t)be * 2 + 1) * PP + lane] / coord_div;
  }
  const int64_t pi = ii[e];
  const int64_t fj = jj[e];
  unsigned long long t_geo = 0, t_first = 0, t_loop = 0;
  if (trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_geo = __builtin_readcyclecounter(); }
  // Window origins, clamped to [-D, size]: a window that starts at or beyond these bounds lies entirely outside the
  // image (all of its taps are zero, correlation_kernel.cu:121-123), wherever exactly it is.
  const int my_ox = min(max(floor_to_int(px) - R, -D), W2), my_oy = min(max(floor_to_int(py) - R, -D), H2);
  const float my_dx = px - floorf(px), my_dy = py - floorf(py);
  int ox[PP], oy[PP];
#pragma unroll
  for (int p = 0; p < PP; p++) { ox[p] = __builtin_amdgcn_readlane(my_ox, p); oy[p] = __builtin_amdgcn_readlane(my_oy, p); }

  const float* __restrict__ f1 = fmap1 + ((int64_t)b * Np + pi) * C * PP;           // [C][9]
  const float* __restrict__ f2 = fmap2 + (int64_t)b * s_b + fj * s_n;
  float* outp = out + (int64_t)be * out_estride + out_offset;

  // ---- patch features -> LDS, transposed to [pixel][C + 4]; element el = 9 c + p, six loads in flight
  {
    const int n1 = C * PP;
    int c = lane / PP, p = lane - c * PP;
    for (int el0 = lane; el0 < n1; el0 += 64 * 6) {
      float v[6];
#pragma unroll
      for (int u = 0; u < 6; u++) v[u] = (el0 + 64 * u < n1) ? f1[el0 + 64 * u] : 0.0f;
#pragma unroll
      for (int u = 0; u < 6; u++) {
        if (el0 + 64 * u < n1) f1t[p * F1S + c] = v[u];
        p += 1; c += 7;                                       // 64 = 7 * 9 + 1
        if (p >= PP) { p -= PP; c += 1; }
      }
    }
  }
  const float* wrow = f1t + min(lane & 15, PP - 1) * F1S;     // this lane's patch pixel (lanes 9..15 of a row unused)

  // ---- lane -> (ds_read_b128 lane group g, index j inside the group)
  int g, j;
  {
    const int l32 = lane & 31;
    int gs;
    if (l32 < 4) { gs = 0; j = l32; }
    else if (l32 < 12) { gs = 1; j = l32 - 4; }
    else if (l32 < 16) { gs = 0; j = l32 - 8; }
    else if (l32 < 20) { gs = 1; j = l32 - 8; }
    else if (l32 < 28) { gs = 0; j = l32 - 12; }
    else { gs = 1; j = l32 - 16; }
    g = (lane >> 5) * 2 + gs;
  }

  float acc[PP];
#pragma unroll
  for (int p = 0; p < PP; p++) acc[p] = 0.0f;
  int ta = 0, tc = 0;
  bool tap_ok = false;
  const int nchunk = C >> 2;

  unsigned remaining = 0x1ffu;
  bool first_stage = true;
  bool split;
  {
    int x0 = ox[0], x1 = ox[0], y0 = oy[0], y1 = oy[0];
#pragma unroll
    for (int p = 1; p < PP; p++) { x0 = min(x0, ox[p]); x1 = max(x1, ox[p]); y0 = min(y0, oy[p]); y1 = max(y1, oy[p]); }
    split = (y1 - y0 + D) * dma_pitch(x1 - x0 + D, false) > DMA_SLOTS;
  }
  while (remaining) {                                         // wave-uniform: one pass per pixel group (usually one)
    int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
    unsigned grp = 0;
#pragma unroll
    for (int q = 0; q < PP; q++) {
      if ((remaining >> q) & 1u) {
        if (!grp) { bx0 = bx1 = ox[q]; by0 = by1 = oy[q]; grp = 1u << q; }
        else {
          const int nx0 = min(bx0, ox[q]), nx1 = max(bx1, ox[q]), ny0 = min(by0, oy[q]), ny1 = max(by1, oy[q]);
          if ((ny1 - ny0 + D) * dma_pitch(nx1 - nx0 + D, split) <= DMA_SLOTS) { bx0 = nx0; bx1 = nx1; by0 = ny0; by1 = ny1; grp |= 1u << q; }
        }
      }
    }
    remaining &= ~grp;
    const int bw = bx1 - bx0 + D, bh = by1 - by0 + D, PT = dma_pitch(bw, split), nslots = bh * PT;

    // lane -> tap for this pitch: group g reads tap rows (a, a + d), d = 8 / lowbit(PT mod 16)
    {
      const int low = PT & (-PT) & 15;                        // 2, 4 or 8
      const int ld = (low == 2) ? 2 : (low == 4 ? 1 : 0), d = 1 << ld;
      const int a = ((g >> ld) << (ld + 1)) | (g & (d - 1));
      ta = a + ((j & 8) ? d : 0);
      tc = j & 7;
      tap_ok = (ta < D) && (tc < D);
      if (!tap_ok) { ta = 0; tc = 0; }
    }
    int rowoff[PP];                                            // float offset of this lane's tap inside a tile buffer
#pragma unroll
    for (int p = 0; p < PP; p++) {
      const int o = ((oy[p] - by0 + ta) * PT + (ox[p] - bx0 + tc)) * 4;
      rowoff[p] = ((grp >> p) & 1u) ? o : 0;
    }

    // tile slots of this lane: s = lane + 64 i -> (row, column) of the box -> source address (or the zero buffer)
    const char* gp[3];
    {
      const float inv_pt = 1.0f / (float)PT;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int s = lane + 64 * i;
        const int row = (int)(((float)s + 0.5f) * inv_pt);   // exact: s < 192, PT <= 32
        const int col = s - row * PT;
        const int gy = by0 + row, gx = bx0 + col;
        const bool ok = (row < bh) && (col < bw) && gy >= 0 && gy < H2 && gx >= 0 && gx < W2;
        gp[i] = ok ? reinterpret_cast<const char*>(f2 + (int64_t)gy * s_h + (int64_t)gx * s_w)
                   : reinterpret_cast<const char*>(g_corr_zero);
      }
    }
    const bool ld1 = nslots > 64, ld2 = nslots > 128;         // wave-uniform
    auto issue = [&](float* dst) {
      dma16(gp[0], dst);
      if (ld1) dma16(gp[1], dst + 256);
      if (ld2 && lane < 32) dma16(gp[2], dst + 512);         // the third load covers slots 128..159 only
      gp[0] += 16; gp[1] += 16; gp[2] += 16;
    };
    issue(buf0);
    if (trace && first_stage) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_first = __builtin_readcyclecounter(); }
    first_stage = false;

    if (grp == 0x1ffu) {
      // ---- the usual case: all 9 windows in one box
      for (int c = 0; c < nchunk; c++) {
        const float* cur = (c & 1) ? buf1 : buf0;
        float* nxt = (c & 1) ? buf0 : buf1;
        const float4 w = *reinterpret_cast<const float4*>(wrow + 4 * c);
        const float4 v0 = *reinterpret_cast<const float4*>(cur + rowoff[0]);
        const float4 v1 = *reinterpret_cast<const float4*>(cur + rowoff[1]);
        const float4 v2 = *reinterpret_cast<const float4*>(cur + rowoff[2]);
        const float4 v3 = *reinterpret_cast<const float4*>(cur + rowoff[3]);
        const float4 v4 = *reinterpret_cast<const float4*>(cur + rowoff[4]);
        const float4 v5 = *reinterpret_cast<const float4*>(cur + rowoff[5]);
        const float4 v6 = *reinterpret_cast<const float4*>(cur + rowoff[6]);
        const float4 v7 = *reinterpret_cast<const float4*>(cur + rowoff[7]);
        const float4 v8 = *reinterpret_cast<const float4*>(cur + rowoff[8]);
        if (c + 1 < nchunk) issue(nxt);                       // next chunk lands in the other buffer under the FMAs
        fma_px012(acc[0], acc[1], acc[2], w, v0, v1, v2);
        fma_px345(acc[3], acc[4], acc[5], w, v3, v4, v5);
        fma_px678(acc[6], acc[7], acc[8], w, v6, v7, v8);
      }
    } else {
      // ---- a pixel group of a split box
      for (int c = 0; c < nchunk; c++) {
        const float* cur = (c & 1) ? buf1 : buf0;
        float* nxt = (c & 1) ? buf0 : buf1;
        const float4 w = *reinterpret_cast<const float4*>(wrow + 4 * c);
        float4 v[PP];
#pragma unroll
        for (int p = 0; p < PP; p++) v[p] = *reinterpret_cast<const float4*>(cur + rowoff[p]);
        if (c + 1 < nchunk) issue(nxt);
        if (grp & 0x001u) fma_one<0>(acc[0], w, v[0]);
        if (grp & 0x002u) fma_one<1>(acc[1], w, v[1]);
        if (grp & 0x004u) fma_one<2>(acc[2], w, v[2]);
        if (grp & 0x008u) fma_one<3>(acc[3], w, v[3]);
        if (grp & 0x010u) fma_one<4>(acc[4], w, v[4]);
        if (grp & 0x020u) fma_one<5>(acc[5], w, v[5]);
        if (grp & 0x040u) fma_one<6>(acc[6], w, v[6]);
        if (grp & 0x080u) fma_one<7>(acc[7], w, v[7]);
        if (grp & 0x100u) fma_one<8>(acc[8], w, v[8]);
      }
    }
    wave_lds_fence();
  }

  // ---- raw windows [p][a][c] (row stride D*D+1: conflict-free epilogue reads); they overwrite the dead tile
  float* rawwin = buf0;
  if (trace) t_loop = __builtin_readcyclecounter();
  if (tap_ok) {
#pragma unroll
    for (int p = 0; p < PP; p++) rawwin[p * (D * D + 1) + ta * D + tc] = acc[p];
  }
  wave_lds_fence();
  // ---- fused bilinear blend + axis swap + output permutation (correlation_kernel.cu:221-232)
  // output element l = (cx * Dm + a) * 9 + p   (cx = x offset = logical dim 2, a = y offset: permute(0,1,3,2,4,5))
  const int Dm = D - 1;
  const int total = Dm * Dm * PP;
  {
    int q = lane / PP, p = lane - q * PP;
    int cx = q / Dm, a = q - cx * Dm;
    float* op = outp + (int64_t)lane * out_lstride;
    const int64_t ostep = 64 * out_lstride;
    for (int l0 = 0; l0 < total; l0 += 64) {                  // wave-uniform trip count: the shuffles need all lanes
      const float dxp = __shfl(my_dx, p), dyp = __shfl(my_dy, p);
      if (l0 + lane < total) {
        const float* r = rawwin + p * (D * D + 1) + a * D + cx;
        *op = blend4(dxp, dyp, r[0], r[1], r[D], r[D + 1]);
      }
      op += ostep;
      p += 1; a += 7;                                          // l += 64 = 7 * 9 + 1
      if (p >= PP) { p -= PP; a += 1; }
      while (a >= Dm) { a -= Dm; cx += 1; }
    }
  }
  if (trace && lane == 0) {                                    // debug: per-wave cycle stamps
    unsigned long long* t = trace + (size_t)slot * 8;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[0] = t_start; t[1] = __builtin_readcyclecounter(); t[2] = 0; t[3] = blockIdx.x;
    t[4] = t_geo; t[5] = t_first; t[6] = t_loop;
  }
}
