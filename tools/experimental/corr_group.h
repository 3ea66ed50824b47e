// altcorr lookup for a GROUP of neighbouring edges on the dense matrix-core shape (v_mfma_f32_16x16x32_f16), operands straight
// from global memory into registers (included by corr.hip inside namespace devo).  Semantics: correlation_kernel.cu:82-136 (raw
// window) + :221-232 (blend, permute); devo.py:215-217 (two levels stacked).
//
// Why: the per-edge kernel (corr_mfma.h) pulls every edge's own boxes through its CU — 1.0 GB (fp16) per cfg2 launch — and a CU
// takes in only 11 B/cycle from HBM and 22-28 B/cycle from L2 (tools/ubench/dma_fill.hip): it sits at that limit.  The
// region-staged kernel (corr_dense.h) shares a staged region between edges but cannot issue its LDS fills fast enough from a few
// waves.  Here the sharing happens in REGISTERS instead:
//   * a workgroup (8 waves, two per CU) takes GE = 16 consecutive edges of the locality plan (same frame, neighbouring positions).
//     Their 144 patch pixels are the N dimension of the product: 9 N-tiles of 16 COLUMNS (edge, pixel).  The B operand — all 144
//     columns x 128 channels, transposed once per workgroup — stays in LDS (36 KB, swizzled: one conflict-free ds_read_b128
//     per lane and K step);
//   * the M dimension is the union REGION of the edges' windows (a PASS = a run of edges of one frame whose bounding box stays
//     below a cap), cut into tiles of 16 positions (4 quads of 4 pixels of one row).  A wave takes every 8th tile: the lane's A
//     operand is 4 x 16 bytes loaded straight from the channel-blocked pyramid into REGISTERS (buffer loads; out-of-image = out
//     of range = 0; the next tile is requested before the current one is multiplied), and ONE loaded tile serves every N-tile
//     whose windows touch it — each pyramid byte enters the CU once per group instead of once per edge;
//   * the 16 x 16 results of a (tile, N-tile) product are dot products of 16 positions with 16 columns; the lane (column n,
//     quad g) scatters its 4 values into the column's raw window  taps[level][column][D x D]  in LDS where they fall inside it
//     (most do not: windows are 8 x 8 of a ~26 x 30 region; tile / N-tile pairs that cannot meet are skipped by scalar tests);
//   * no barrier inside the product loop, no LDS staging of the features.  The levels run one after the other (coarse first)
//     over ONE raw-window area; each level's epilogue (blend, axis swap, permutation) writes its slice of the output record.
// fp16 storage, C = 128 (four K = 32 steps), radius <= 3.  Other cases take the per-edge kernel.
// Debug: DEVO_GP_STATS=1 (phase cycles / tile counts to stderr, tools/group_stats.py); ablation builds -DGP_DBG_NOPAIRS / _NOSCATTER /
// _NOLOAD / _NOEPI (tools/build_variant.sh; profiles/r02y_group_ablation.txt).  Measured slower than the per-edge kernel on cfg2
// (132 us against 99 per fp16 launch): DESIGN.md 3.1d says where the time goes.
#pragma once

typedef _Float16 gp_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 gp_h2 __attribute__((ext_vector_type(2)));
typedef float gp_f4 __attribute__((ext_vector_type(4)));
typedef short gp_s2 __attribute__((ext_vector_type(2)));

constexpr int GP_WAVES = 8;
constexpr int GP_THREADS = GP_WAVES * 64;
constexpr int GP_GE = 16;                          // edges per workgroup
constexpr int GP_NCOL = GP_GE * PP;                // 144 columns
constexpr int GP_NT = GP_NCOL / 16;                // 9 N-tiles
constexpr int GP_DMAX = 8;                         // 2 * 3 + 2
constexpr int GP_TAPS = GP_DMAX * GP_DMAX + 1;     // floats per column: D x D taps + 1 (bank spread, dump slot)
constexpr int GP_CAP = 1024;                       // positions of a pass's region
constexpr int GP_C = 128;                          // channels
static_assert(GP_NCOL % 16 == 0, "whole N-tiles");

// min / max over the 16 lanes of a row, result in lane 15 of the row (row_shr 1, 2, 4, 8)
__device__ __forceinline__ int gp_row_min(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x111, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x112, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x114, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x118, 0xf, 0xf, false));
  return v;
}

// workgroup barrier that waits for the wave's LDS operations only (global stores stay in flight)
__device__ __forceinline__ void gp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, int NL>
__global__ __launch_bounds__(GP_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void corr_fwd_group_kernel(
    const T* __restrict__ fmap1, CorrLevel lv0, CorrLevel lv1, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, T* __restrict__ out, int BE, int E, int Np, int n2,
    int C, int64_t out_estride, int64_t out_lstride, int R, const int* __restrict__ order,
    unsigned long long* __restrict__ stats) {
  static_assert(sizeof(T) == 2, "fp16 storage");
  constexpr unsigned ESZ = 2;
  constexpr int GE = GP_GE, NCOL = GP_NCOL, NTN = GP_NT, NKS = GP_C / 32;
  constexpr int PATCH_BYTES = GP_C * PP * (int)ESZ;           // 2304: one edge's patch, raw [C][9]
  constexpr int RAW_BYTES = GE * PATCH_BYTES;                 // 36 864
  constexpr int BT_BYTES = NCOL * GP_C * (int)ESZ;            // 36 864: transposed [column][16 pieces of 8 channels], swizzled
  constexpr int TAPS_BYTES = NCOL * GP_TAPS * 4;              // 37 440: the raw windows of ONE level
  constexpr int ARENA = TAPS_BYTES > RAW_BYTES ? TAPS_BYTES : RAW_BYTES;
  constexpr unsigned OFF_NONE = 0x80000000u;
#define LVF(l, F) ((NL == 2 && (l)) ? lv1.F : lv0.F)

  __shared__ __attribute__((aligned(16))) unsigned char s_bt[BT_BYTES];   // B operand
  __shared__ __attribute__((aligned(16))) unsigned char s_arena[ARENA];   // raw patches, then the raw windows of the level at work
  __shared__ int s_ox[NL][NCOL], s_oy[NL][NCOL];            // window origins (tap 0, 0) per column, frame coordinates
  __shared__ float s_fx[NL][NCOL], s_fy[NL][NCOL];          // sub-pixel fractions per column
  __shared__ __attribute__((aligned(16))) int s_box[NL][GE][4];   // x0, y0, x1, y1 (exclusive) of an edge's union box
  __shared__ __attribute__((aligned(16))) int s_pass[NL][GE][8];  // ranks lo | hi << 8, region x0, y0, quads per row, rows, first tile, heavy, frame
  __shared__ int s_jb[NL][GE][NTN];                         // per pass and N-tile: window bounds relative to the region (4 bytes), -1 = none
  __shared__ int s_npass[NL], s_ntile[NL];
  __shared__ int s_be[GE], s_frame[GE], s_pi[GE], s_perm[GE];
  float* const s_taps = reinterpret_cast<float*>(s_arena);
  float* const s_key = reinterpret_cast<float*>(&s_jb[0][0][0]);            // (sort keys: only before the passes are planned)
  int* const s_kx = &s_jb[0][0][0] + GE;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = 2 * R + 2, Dm = D - 1;
  // group of the plan this workgroup owns: workgroup g runs on XCD g % 8 as that XCD's k-th; the XCDs take runs of 4 consecutive
  // groups in turn (neighbouring groups share an L2; the plan's long items — its first slots — start first on every XCD)
  const int ngroups = (BE + GE - 1) / GE;
  int group;
  {
    const int g = blockIdx.x, x = g & 7, k = g >> 3;
    group = (k >> 2) * 32 + x * 4 + (k & 3);
    if (group >= ngroups) return;
  }
  const int gstart = group * GE, nge = min(GE, BE - gstart);
  unsigned long long t_prev = stats ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long t_begin = t_prev;
  auto stamp = [&](int ph) {                                  // debug (DEVO_GP_STATS): cycles of phase ph, first wave
    if (stats && tid == 0) { const unsigned long long t = __builtin_readcyclecounter(); atomicAdd(&stats[ph], t - t_prev); t_prev = t; }
  };
  unsigned n_tiles = 0, n_pairs = 0;
  unsigned long long t_wait = 0;

  // ---- 1. the group's edges, sorted by (frame, 8-px column, centre row): neighbours become neighbours in the N dimension
  if (tid < GE) {
    int be = 0, fr = 0x7fffffff, pi = 0, kx = 0;
    float cy = 0.0f;
    if (tid < nge) {
      be = order ? order[gstart + tid] : gstart + tid;
      const int b = be / E, e = be - b * E;
      fr = b * n2 + (int)jj[e]; pi = b * Np + (int)ii[e];
      const float cx = coords[(int64_t)be * (2 * PP) + PP / 2];
      cy = coords[(int64_t)be * (2 * PP) + PP + PP / 2];
      if (!(cy == cy)) cy = 0.0f;
      kx = floor_to_int(cx) >> 3;
    }
    s_be[tid] = be; s_frame[tid] = fr; s_pi[tid] = pi; s_key[tid] = cy; s_kx[tid] = kx;
  }
  __syncthreads();
  int my_rank = 0;
  if (tid < GE) {
    const int fr = s_frame[tid], kx = s_kx[tid];
    const float cy = s_key[tid];
#pragma unroll
    for (int k = 0; k < GE; k++) {
      const int fk = s_frame[k], xk = s_kx[k];
      const float ck = s_key[k];
      const bool before = fk != fr ? fk < fr : (xk != kx ? xk < kx : (ck != cy ? ck < cy : k < tid));
      my_rank += before ? 1 : 0;
    }
    s_perm[my_rank] = tid;                                     // (edges beyond nge sort last: frame = INT_MAX)
  }
  __syncthreads();

  // ---- 2. column geometry (thread t < 144: column t = (rank, pixel)); raw patches -> LDS
  if (tid < NCOL) {
    const int r = tid / PP, p = tid - r * PP;
    const int k = s_perm[r];
    const bool live = r < nge;
    float cx = 0.0f, cy = 0.0f;
    if (live) { const float* ce = coords + (int64_t)s_be[k] * (2 * PP); cx = ce[p]; cy = ce[PP + p]; }
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const float dv = LVF(l, coord_div);
      const float qx = cx / dv, qy = cy / dv;                  // the reference's true division (coords / s)
      s_ox[l][tid] = floor_to_int(qx) - R; s_oy[l][tid] = floor_to_int(qy) - R;
      s_fx[l][tid] = qx - floorf(qx); s_fy[l][tid] = qy - floorf(qy);
    }
  }
  {
    constexpr int PIECES = PATCH_BYTES / 16;                   // 144 per edge
    const char* const f1 = reinterpret_cast<const char*>(fmap1);
#pragma unroll
    for (int i = 0; i < (GE * PIECES + GP_THREADS - 1) / GP_THREADS; i++) {
      const int id = tid + GP_THREADS * i;
      const int r = id / PIECES, pc = id - r * PIECES;
      if (id < GE * PIECES) {
        gp_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < nge) v = *reinterpret_cast<const gp_f4*>(f1 + (int64_t)s_pi[s_perm[r]] * PATCH_BYTES + pc * 16);
        *reinterpret_cast<gp_f4*>(s_arena + r * PATCH_BYTES + pc * 16) = v;
      }
    }
  }
  __syncthreads();

  // ---- 3. union box per edge and level; patch transposition raw [C][9] -> Bt [column][piece ^ (column & 15)][8 channels]
  if (tid < GE * NL) {
    const int l = tid / GE, r = tid - l * GE;
    int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -0x7fffffff, y1 = -0x7fffffff;
#pragma unroll
    for (int p = 0; p < PP; p++) {
      const int ox = s_ox[l][r * PP + p], oy = s_oy[l][r * PP + p];
      x0 = min(x0, ox); x1 = max(x1, ox); y0 = min(y0, oy); y1 = max(y1, oy);
    }
    s_box[l][r][0] = x0; s_box[l][r][1] = y0; s_box[l][r][2] = x1 + D; s_box[l][r][3] = y1 + D;
  }
  {
    const unsigned short* const raw = reinterpret_cast<const unsigned short*>(s_arena);
#pragma unroll
    for (int i = 0; i < (NCOL * 16 + GP_THREADS - 1) / GP_THREADS; i++) {
      const int id = tid + GP_THREADS * i;
      if (id < NCOL * 16) {
        const int col = id >> 4, qd = id & 15;
        const int r = col / PP, p = col - r * PP;
        const unsigned short* src = raw + r * (PATCH_BYTES / 2) + (8 * qd) * PP + p;
        unsigned w[4];
#pragma unroll
        for (int c = 0; c < 4; c++) w[c] = (unsigned)src[(2 * c) * PP] | ((unsigned)src[(2 * c + 1) * PP] << 16);
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u4*>(s_bt + col * 256 + ((qd ^ (col & 15)) << 4)) = u4{w[0], w[1], w[2], w[3]};
      }
    }
  }
  __syncthreads();

  // ---- 4. passes per level (wave l plans level l: lane r holds the box of rank r, the scan runs on scalars; lane p ends up with
  //      pass p).  A pass = a run of ranks of one frame whose bounding box is no larger than the boxes it replaces together.
  if (wave < NL) {
    const int l = wave;
    const int rr = min(lane, GE - 1);
    const int ex0 = s_box[l][rr][0], ey0 = s_box[l][rr][1], ex1 = s_box[l][rr][2], ey1 = s_box[l][rr][3];
    const int efr = s_frame[s_perm[rr]];
    int o_lh = 0, o_x0 = 0, o_y0 = 0, o_x1 = 0, o_y1 = 0, o_hv = 0, o_fr = 0;
    int np = 0, cx0 = 0, cy0 = 0, cx1 = 0, cy1 = 0, cfr = 0, clo = 0;
    bool open = false;
    auto emit = [&](int lo, int hi, int x0, int y0, int x1, int y1, int hv, int fr) {
      const bool me = lane == np;
      o_lh = me ? (lo | (hi << 8)) : o_lh; o_x0 = me ? x0 : o_x0; o_y0 = me ? y0 : o_y0; o_x1 = me ? x1 : o_x1; o_y1 = me ? y1 : o_y1;
      o_hv = me ? hv : o_hv; o_fr = me ? fr : o_fr;
      np++;
    };
#pragma unroll 1
    for (int r = 0; r < nge; r++) {
      const int x0 = __builtin_amdgcn_readlane(ex0, r), y0 = __builtin_amdgcn_readlane(ey0, r);
      const int x1 = __builtin_amdgcn_readlane(ex1, r), y1 = __builtin_amdgcn_readlane(ey1, r);
      const int fr = __builtin_amdgcn_readlane(efr, r);
      const long long area = (long long)(x1 - x0) * (y1 - y0);
      const bool heavy = area > GP_CAP;
      if (open && !heavy && fr == cfr) {
        const int ux0 = min(cx0, x0), uy0 = min(cy0, y0), ux1 = max(cx1, x1), uy1 = max(cy1, y1);
        const long long ua = (long long)(ux1 - ux0) * (uy1 - uy0), ca = (long long)(cx1 - cx0) * (cy1 - cy0);
        if (ua <= GP_CAP && ua <= ca + area) { cx0 = ux0; cy0 = uy0; cx1 = ux1; cy1 = uy1; continue; }
      }
      if (open) emit(clo, r, cx0, cy0, cx1, cy1, 0, cfr);
      if (heavy) { emit(r, r + 1, x0, y0, x1, y1, 1, fr); open = false; }
      else { open = true; clo = r; cx0 = x0; cy0 = y0; cx1 = x1; cy1 = y1; cfr = fr; }
    }
    if (open) emit(clo, nge, cx0, cy0, cx1, cy1, 0, cfr);
    const int wq = (o_x1 - o_x0 + 3) >> 2, rows = o_y1 - o_y0;
    const int nt = (lane < np && !o_hv) ? (wq * rows + 3) >> 2 : 0;
    const int incl = wave_inclusive_sum(nt);
    if (lane < np) {
      typedef int i4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<i4*>(&s_pass[l][lane][0]) = i4{o_lh, o_x0, o_y0, wq};
      *reinterpret_cast<i4*>(&s_pass[l][lane][4]) = i4{rows, incl - nt, o_hv, o_fr};
    }
    if (lane == 63) { s_npass[l] = np; s_ntile[l] = incl; }
  }
  __syncthreads();
  // bounds of every N-tile's windows per pass, relative to the region (thread (l, p, j))
  if (tid < NL * GE * NTN) {
    const int l = tid / (GE * NTN), pj = tid - l * (GE * NTN), p = pj / NTN, j = pj - p * NTN;
    int packed = -1;
    if (p < s_npass[l] && !s_pass[l][p][6]) {
      const int lh = s_pass[l][p][0], x0 = s_pass[l][p][1], y0 = s_pass[l][p][2];
      const int clo = (lh & 0xff) * PP, chi = (lh >> 8) * PP;
      int mnx = 255, mny = 255, mxx = 0, mxy = 0;
      bool any = false;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const int col = 16 * j + c;
        const int rx = s_ox[l][col] - x0, ry = s_oy[l][col] - y0;
        if (col >= clo && col < chi) { mnx = min(mnx, rx); mny = min(mny, ry); mxx = max(mxx, rx); mxy = max(mxy, ry); any = true; }
      }
      if (any) packed = mnx | (mny << 8) | (mxx << 16) | (mxy << 24);        // (all within 0..127: a pass's region is <= 1024 positions)
    }
    s_jb[l][p][j] = packed;
  }
  __syncthreads();                                             // the arena is free for the raw windows; passes are visible
  stamp(0);

  const int n16 = lane & 15, kg = lane >> 4;
  const int ep = lane % PP, eg = lane / PP;                    // epilogue: lane (g, p), lanes 0..62
  // lane constants of the B reads (byte offsets inside an N-tile's 4 KB of Bt) and of the scatter
  unsigned boff[NKS];
#pragma unroll
  for (int s = 0; s < NKS; s++) boff[s] = (unsigned)(n16 * 256 + (((4 * s + kg) ^ n16) << 4));
  const unsigned char* const taps_lane = s_arena + n16 * (GP_TAPS * 4);

#pragma unroll 1
  for (int lx = 0; lx < NL; lx++) {
    const int l = NL - 1 - lx;                                 // the coarse level first
    // ---- 5. products of level l
    const int H2 = LVF(l, H2), W2 = LVF(l, W2);
    const int sh = LVF(l, cb_shift);
    const unsigned bb = (unsigned)LVF(l, block_stride) * ESZ;
    auto piece = [&](unsigned c) -> unsigned { const unsigned blk = c >> sh; return blk * bb + (c - (blk << sh)) * ESZ; };
    const unsigned lane_piece = piece(8u * (unsigned)kg), step_piece = piece(32u);
    const unsigned sh_ = (unsigned)LVF(l, s_h) * ESZ, sw_ = (unsigned)LVF(l, s_w) * ESZ;
    const int npass = s_npass[l], ntiles = s_ntile[l];
    const int* const oxl = &s_ox[l][0];
    const int* const oyl = &s_oy[l][0];

    struct Pass { int x0, y0, wq, nquad, t0, t1, clo, chi, jb; float inv_wq; __amdgpu_buffer_rsrc_t rs; };
    auto frame_rsrc = [&](int frame) -> __amdgpu_buffer_rsrc_t {
      const T* fbase = static_cast<const T*>(LVF(l, fmap2)) + (int64_t)(frame / n2) * LVF(l, s_b) + (int64_t)(frame % n2) * LVF(l, s_n);
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(fbase), 0, LVF(l, frame_bytes), 0x00020000);
    };
    auto load_pass = [&](int p) -> Pass {                      // p wave-uniform
      Pass P;
      const int lh = s_pass[l][p][0];
      P.x0 = s_pass[l][p][1]; P.y0 = s_pass[l][p][2]; P.wq = s_pass[l][p][3];
      const int rows = s_pass[l][p][4];
      P.t0 = s_pass[l][p][5];
      P.nquad = P.wq * rows; P.t1 = P.t0 + ((P.nquad + 3) >> 2);
      P.clo = (lh & 0xff) * PP; P.chi = (lh >> 8) * PP;
      P.inv_wq = __builtin_amdgcn_rcpf((float)P.wq);
      P.rs = frame_rsrc(s_pass[l][p][7]);
      P.jb = s_jb[l][p][min(lane, NTN - 1)];                   // lane j < 9: bounds of N-tile j
      return P;
    };
    auto tile_off = [&](const Pass& P, int t) -> unsigned {    // the lane's A position of the pass's tile t: quad 4 t + (n16 >> 2), pixel n16 & 3
      const int q = 4 * t + (n16 >> 2);
      const int row = (int)(((float)q + 0.5f) * P.inv_wq), xq = q - row * P.wq;
      const int x = P.x0 + 4 * xq + (n16 & 3), y = P.y0 + row;
      const bool in = q < P.nquad && x >= 0 && x < W2 && y >= 0 && y < H2;
      return in ? (unsigned)y * sh_ + (unsigned)x * sw_ + lane_piece : OFF_NONE;
    };
    auto fetch = [&](gp_f4 (&a)[NKS], const Pass& P, int t) {
#ifdef GP_DBG_NOLOAD
      const unsigned off = tile_off(P, t) | OFF_NONE;
#else
      const unsigned off = tile_off(P, t);
#endif
#pragma unroll
      for (int s = 0; s < NKS; s++) a[s] = __builtin_bit_cast(gp_f4, __builtin_amdgcn_raw_buffer_load_b128(P.rs, off, (unsigned)s * step_piece, 0));
    };
    // one loaded tile against every N-tile whose windows it can touch; the B reads and origins of the next pair are requested
    // before the current pair is multiplied and scattered
    auto do_tile = [&](const Pass& P, int t, const gp_h8 (&a)[NKS]) {
      const int qc = 4 * t + kg;                               // the lane's C positions: quad 4 t + kg, pixels 0..3 of it
      const int rowc = (int)(((float)qc + 0.5f) * P.inv_wq), xqc = qc - rowc * P.wq;
      const int r0 = __builtin_amdgcn_readlane(rowc, 0), r1 = __builtin_amdgcn_readlane(rowc, 48);     // rows of quads 4 t, 4 t + 3
      const int xa = __builtin_amdgcn_readlane(xqc, 0) * 4;
      const int xlo = r0 == r1 ? xa : 0, xhi = r0 == r1 ? xa + 15 : 4 * P.wq - 1;
      const int jb = P.jb;
      const int mnx = jb & 0xff, mny = (jb >> 8) & 0xff, mxx = (jb >> 16) & 0xff, mxy = (jb >> 24) & 0xff;
      const bool h = lane < NTN && jb != -1 && r1 >= mny && r0 < mxy + D && xhi >= mnx && xlo < mxx + D;
      unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)(__ballot(h) & 0x1ffull));
      n_tiles++;
#ifdef GP_DBG_NOPAIRS
      mask = 0;
#endif
      if (mask == 0) return;                                   // wave-uniform
      auto read_b = [&](gp_h8 (&b)[NKS], int j) {
        const unsigned char* bp = s_bt + j * 4096;
#pragma unroll
        for (int s = 0; s < NKS; s++) b[s] = *reinterpret_cast<const gp_h8*>(bp + boff[s]);
      };
      gp_h8 bc[NKS], bn[NKS];
      int j = __builtin_ctz(mask);
      read_b(bc, j);
      int oxc = oxl[16 * j + n16], oyc = oyl[16 * j + n16];
      for (;;) {
        mask &= mask - 1;
        const bool more = mask != 0;                           // wave-uniform
        const int jn = more ? __builtin_ctz(mask) : j;
        int oxn = oxc, oyn = oyc;
        if (more) { read_b(bn, jn); oxn = oxl[16 * jn + n16]; oyn = oyl[16 * jn + n16]; }
        n_pairs++;
        gp_f4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NKS; s++) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], bc[s], c, 0, 0, 0);
        const int col = 16 * j + n16;
        const bool act = col >= P.clo && col < P.chi;
        const int dxb = 4 * xqc - (oxc - P.x0), dy = act ? rowc - (oyc - P.y0) : -1;
        const bool rowok = (unsigned)dy < (unsigned)D;
        float* const cb = reinterpret_cast<float*>(const_cast<unsigned char*>(taps_lane) + j * (16 * GP_TAPS * 4));
        const int a0 = dy * GP_DMAX + dxb;
#ifdef GP_DBG_NOSCATTER
        if (c[0] == 12345.678f) cb[a0] = c[1] + c[2] + c[3];
#else
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const bool ok = rowok && (unsigned)(dxb + i) < (unsigned)D;
          cb[ok ? a0 + i : GP_DMAX * GP_DMAX] = c[i];
        }
#endif
        if (!more) break;
#pragma unroll
        for (int s = 0; s < NKS; s++) bc[s] = bn[s];
        oxc = oxn; oyc = oyn; j = jn;
      }
    };

    // (a) the level's tiles as ONE list over its passes; this wave takes tiles wave, wave + 8, ..; the next tile is requested
    //     (also across a pass boundary) before the current one is worked on
    if (wave < ntiles) {                                       // wave-uniform
      int Tn = wave, pn = 0;
      while (Tn >= s_pass[l][pn][5] + (((s_pass[l][pn][3] * s_pass[l][pn][4] + 3) >> 2) * (s_pass[l][pn][6] ? 0 : 1))) pn++;
      Pass Pn = load_pass(pn);
      gp_f4 a_nxt[NKS];
      fetch(a_nxt, Pn, Tn - Pn.t0);
#pragma unroll 1
      for (;;) {
        const Pass Pc = Pn;
        const int Tc = Tn;
        gp_h8 a[NKS];
        if (stats) {
          const unsigned long long tw0 = __builtin_readcyclecounter();
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (tid == 0) t_wait += __builtin_readcyclecounter() - tw0;
        }
#pragma unroll
        for (int s = 0; s < NKS; s++) a[s] = __builtin_bit_cast(gp_h8, a_nxt[s]);
        Tn += GP_WAVES;
        const bool more = Tn < ntiles;                          // wave-uniform
        if (more) {
          if (Tn >= Pn.t1) {
            do { pn++; } while (s_pass[l][pn][6] || Tn >= s_pass[l][pn][5] + ((s_pass[l][pn][3] * s_pass[l][pn][4] + 3) >> 2));
            Pn = load_pass(pn);
          }
          fetch(a_nxt, Pn, Tn - Pn.t0);
        }
        do_tile(Pc, Tc - Pc.t0, a);
        if (!more) break;
      }
    }
    // (b) heavy edges (their own box exceeds the cap: patch pixels spread apart): window by window
#pragma unroll 1
    for (int ps = 0; ps < npass; ps++) {
      if (!s_pass[l][ps][6]) continue;                         // wave-uniform
      const int r = s_pass[l][ps][0] & 0xff;
      const __amdgpu_buffer_rsrc_t rs = frame_rsrc(s_pass[l][ps][7]);
#pragma unroll 1
      for (int sub = 0; sub < PP; sub++) {
        const int col = r * PP + sub;
        Pass P;
        P.x0 = oxl[col]; P.y0 = oyl[col]; P.wq = (D + 3) >> 2; P.nquad = P.wq * D; P.t0 = 0; P.t1 = (P.nquad + 3) >> 2;
        P.clo = col; P.chi = col + 1; P.inv_wq = __builtin_amdgcn_rcpf((float)P.wq); P.rs = rs;
        P.jb = lane == (col >> 4) ? 0 : -1;                   // the window is the region: origin (0, 0)
#pragma unroll 1
        for (int t = wave; t < P.t1; t += GP_WAVES) {
          gp_f4 af[NKS];
          fetch(af, P, t);
          gp_h8 a[NKS];
#pragma unroll
          for (int s = 0; s < NKS; s++) a[s] = __builtin_bit_cast(gp_h8, af[s]);
          do_tile(P, t, a);
        }
      }
    }
    stamp(1);
    gp_barrier();
    stamp(2);

    // ---- 6. epilogue of level l: wave w writes the slices of ranks w, w + 8; lane (g, p) = (lane / 9, lane % 9), lanes 0..62:
    //      output t = 63 j + lane = q * 9 + p with q = 7 j + g = cx * Dm + a
    {
      constexpr int NQ = ((GP_DMAX - 1) * (GP_DMAX - 1) + 6) / 7;
      const int nq = Dm * Dm;
      const bool act = lane < 63;
      const int64_t ooff = LVF(l, out_offset);
#ifdef GP_DBG_NOEPI
      if (R == 77)
#endif
#pragma unroll 1
      for (int r = wave; r < nge; r += GP_WAVES) {
        const int be = s_be[s_perm[r]];
        T* op = out + (int64_t)be * out_estride + ooff;
        const int col = r * PP + ep;
        const float dxp = s_fx[l][col], dyp = s_fy[l][col];
        float w00, w01, w10, w11;
        {
#pragma clang fp contract(off)
          w00 = (1.0f - dxp) * (1.0f - dyp); w01 = dxp * (1.0f - dyp); w10 = (1.0f - dxp) * dyp; w11 = dxp * dyp;
        }
        const float* rw = s_taps + col * GP_TAPS;
        float rv[NQ][4];
#pragma unroll
        for (int j = 0; j < NQ; j++) {
          const int q = min(7 * j + eg, nq - 1);
          const int cx = q / Dm, a = q - cx * Dm;
          const float* rr = rw + a * GP_DMAX + cx;
          rv[j][0] = rr[0]; rv[j][1] = rr[1]; rv[j][2] = rr[GP_DMAX]; rv[j][3] = rr[GP_DMAX + 1];
        }
#pragma unroll
        for (int j = 0; j < NQ; j++) {
          const int q = 7 * j + eg;
          float o;
          {
#pragma clang fp contract(off)
            o = w00 * rv[j][0]; o = o + w01 * rv[j][1]; o = o + w10 * rv[j][2]; o = o + w11 * rv[j][3];
          }
          if (act && q < nq) store_streamed(op + (int64_t)(q * PP + ep) * out_lstride, from_f32<T>(o));
        }
      }
    }
    stamp(3);
    if (lx + 1 < NL) gp_barrier();                             // the next level's products overwrite the raw windows
  }
  if (stats && tid == 0) {
    atomicAdd(&stats[4], 1ull); atomicAdd(&stats[5], (unsigned long long)n_tiles); atomicAdd(&stats[6], (unsigned long long)n_pairs);
    atomicAdd(&stats[7], (unsigned long long)(s_npass[0] + (NL == 2 ? s_npass[NL - 1] : 0))); atomicMax(&stats[8], __builtin_readcyclecounter() - t_begin);
    atomicAdd(&stats[9], t_wait);
  }
#undef LVF
}
