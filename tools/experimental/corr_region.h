// altcorr lookup, REGION-SHARED: neighbouring edges of one target frame read their boxes out of ONE staged copy of the frame region
// they cover (included by corr.hip inside namespace devo; semantics: devo/altcorr/correlation_kernel.cu:82-136,221-232).
//
// Why: the per-edge kernel (corr_mfma.h) pulls every edge's own ~107 + ~74 box positions through its CU — 2.0 GB per cfg2 launch —
// and sits at the rate at which a CU can take bytes in.  At DEVO's patch density a level-0 pixel is wanted by ~8
// edges and a level-1 pixel by ~90, so a workgroup that stages the union region of ~20 image-neighbours ONCE moves 4-5x fewer bytes.
//
// Structure.  A workgroup takes a CHUNK of consecutive slots of the locality plan (edges of one frame band, sorted by 8-px column),
// sorts them by box origin and forms greedy ROUNDS: runs of <= NW * EW edges of one frame whose union region (clipped to the frame)
// fits the staging buffer at both pyramid levels.  Per round and level (level 1 first, its results wait in registers):
//   * the region is streamed through LDS in channel SLABS (32 fp16 / 16 fp32 channels = 64 bytes per position) with
//     buffer_load_dwordx4 ... lds (no staging registers; positions outside the region / the frame are out-of-range offsets = zeros),
//     double-buffered: slab s + 1 is in flight while slab s is multiplied; the round's patch slabs travel the same way;
//   * LDS image of a slab: 4 planes [16-byte piece][position] — the A operand of v_mfma_f32_16x16x32_f16 (16 box positions x 32 k)
//     is one ds_read_b128 per lane; B = the patch (9 of 16 columns used) from the patch area; every wave keeps the accumulators of
//     its EW edges (TMAX tiles of 16 box positions each) in registers across the slabs;
//   * fp32 storage: when a slab has landed, every 16-byte piece (4 channels) is rewritten IN PLACE as fp16 (hi0..3 | lo0..3), x = hi + lo
//     (v_cvt_pk_f16_f32 + v_fma_mix: 6 instructions per piece, once per staged value); the patches arrive split already
//     (devo_corr_patch_transpose).  K = [hi | lo] of 16 channels against [hi | hi] and [lo | 0] of the patch: x y = xh yh + xl yh + xh yl
//     with fp32 accumulation (2^-22 relative per factor; domain |feature| <= 65504);
//   * epilogue: the 16 x 16 result tiles go through a per-wave scratch [pixel][box position] (one 16-byte store per tile); lane
//     (pixel p, window row a) reads rows a, a + 1 of its window and blends the 2r + 1 outputs of that row in the reference's operation
//     order; the level-0 epilogue writes both levels' interleaved record (torch.stack([c0, c1], -1)).
// Edges the rounds cannot take (more than TMAX tiles at a level) are computed tap by tap straight from memory by one wave (correct for
// any coordinates, slow: the plan sends such edges to the per-edge kernel instead).  The dead tail of the plan (edges whose boxes lie
// outside the frame at both levels: 11 % at cfg2) is zero-filled.
#pragma once

typedef _Float16 rg_h8 __attribute__((ext_vector_type(8)));
typedef float rg_f4 __attribute__((ext_vector_type(4)));
typedef unsigned rg_u4 __attribute__((ext_vector_type(4)));

template <int RMAX_, int NW_, int EW_, int TMAX_, int RC_, int NBUF_ = 2, int CHMAX_ = 64, bool XPRE_ = true>
struct RgShape {
  // XPRE: the next stage's first slab is requested BEFORE the current stage's epilogue (buffer 1; the scratch must fit buffer 0).
  // Without it the scratch may span both buffers and the request follows the epilogue.
  static constexpr bool XPRE = XPRE_ && NBUF_ == 2;
  static constexpr int RMAX = RMAX_, NW = NW_, EW = EW_, TMAX = TMAX_, RC = RC_;
  // NBUF = 2: one workgroup per CU, slab s + 1 lands while slab s is multiplied.  NBUF = 1: ONE slab buffer and two workgroups per CU —
  // a workgroup alternates between requesting / waiting for a slab and multiplying it, its neighbour on the CU fills the gaps (and
  // hides its prologue, set-up and epilogue as well).
  static constexpr int NBUF = NBUF_;
  static constexpr int DMAX = 2 * RMAX + 2, DM = DMAX - 1, NPASS = (DM + 6) / 7;
  static constexpr int THREADS = 64 * NW;
  static constexpr int SLOTS = NW * EW;                         // edges per round
  static constexpr int CHMAX = CHMAX_;                          // plan slots per chunk (one thread each in the prologue)
  static constexpr int NBCH = (SLOTS * PP * 64 + 1023) / 1024;  // 1 KB pieces of one patch slab: [slot][pixel][64 B]
  static constexpr int BUNIT = NBCH * 1024;
  static constexpr int SP = TMAX * 16 + (RMAX <= 3 ? 4 : 0);    // scratch: floats per pixel row (+4: bank spread)
  static constexpr int SCRW = PP * SP * 4;                      // scratch bytes per wave
  static constexpr int BUFMIN = RC * 4096 + BUNIT;              // one slab buffer: region planes, then the patch slab(s) ...
  static constexpr int BUFSZ = ((!XPRE || BUFMIN > NW * SCRW) ? BUFMIN : NW * SCRW + 1023) / 1024 * 1024;   // ... and room for every wave's scratch
  static constexpr int DPW = (RC + NW - 1) / NW, BPW = (NBCH + NW - 1) / NW;   // DMA pieces per wave, plane and unit
  static_assert(NW * SCRW <= (XPRE ? 1 : NBUF) * BUFSZ, "the epilogue scratch lives in slab buffer 0 (buffer 1 takes the next stage's first slab meanwhile)");
  static constexpr int LDS_BYTES = NBUF * BUFSZ;
  static constexpr int WPS = (NW * (NBUF == 1 ? 2 : 1) + 3) / 4;   // waves per SIMD the launch needs (registers: 512 / WPS)
  static_assert(SP >= DMAX * DMAX, "the tap-by-tap path keeps raw windows in the scratch");
  static_assert(THREADS >= CHMAX && SLOTS <= CHMAX, "one prologue thread per plan slot");
  static_assert(16 * TMAX < 64 * RC && TMAX % 4 == 0, "a single edge must fit the region; tiles go in batches of 4");
};

// LDS-DMA: 16 bytes per lane, source = buffer descriptor + per-lane offset (out of range: zeros, no access) + scalar offset,
// destination = lds_addr + 16 * lane.  Issued through asm: hipcc's waitcnt insertion does not know it (rg_wait_dma()).
__device__ __forceinline__ void rg_dma16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned lds_addr) {
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);     // (wave-uniform by construction; pins them to scalar registers)
  soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(lds_addr), "s"(soff) : "memory");
}
__device__ __forceinline__ void rg_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// a wave's LDS instructions execute in order: this waits for them and pins the compiler's ordering — without the fence's wait for
// vector-memory operations (the next stage's DMA is in flight during an epilogue)
__device__ __forceinline__ void rg_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void rg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 4 fp32 values -> fp16 (hi0..3 | lo0..3), x = hi + lo to 2^-22 (|lo| below the fp16 normal range keeps 2^-25 absolute)
__device__ __forceinline__ rg_h8 rg_split4(rg_f4 x) {
  unsigned h01, h23, l01, l23;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(x[0]), "v"(x[1]));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(x[2]), "v"(x[3]));
  // (mixlo keeps the destination's upper half, which mixhi then overwrites: no initialisation needed)
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h01), "v"(x[0]));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h01), "v"(x[1]));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h23), "v"(x[2]));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h23), "v"(x[3]));
  const rg_u4 r = {h01, h23, l01, l23};
  return __builtin_bit_cast(rg_h8, r);
}

template <typename T, typename S, bool STATS>
__global__ __launch_bounds__(S::THREADS, S::WPS) void corr_fwd_region_kernel(
    const T* __restrict__ fmap1t, CorrLevel lv0, CorrLevel lv1, const float* __restrict__ coords, const int64_t* __restrict__ ii,
    const int64_t* __restrict__ jj, T* __restrict__ out, int BE, int E, int Np, int n2, int C, int64_t oes, int64_t ols, int R,
    const int* __restrict__ order, int nchunks, unsigned f1t_bytes, int band_rows, unsigned long long* __restrict__ stats) {
  constexpr bool HALF = sizeof(T) == 2;
  constexpr unsigned ESZ = sizeof(T);
  constexpr int CS = HALF ? 32 : 16, PCH = 16 / (int)ESZ;      // channels per slab unit / per 16-byte piece
  constexpr int NW = S::NW, EW = S::EW, TMAX = S::TMAX, RC = S::RC, SLOTS = S::SLOTS, CHMAX = S::CHMAX, DMAX = S::DMAX, DM = S::DM;
  constexpr int NPASS = S::NPASS, SP = S::SP, NBCH = S::NBCH, BUNIT = S::BUNIT, BUFSZ = S::BUFSZ, DPW = S::DPW, BPW = S::BPW;
  constexpr int CAP = RC * 64 - 1;                              // region positions (one more slot stays zero)
  constexpr int OPU = 4 * DPW + BPW;                            // DMA instructions a wave may have per slab unit
  constexpr unsigned OFF_NONE = 0x80000000u;

  extern __shared__ __attribute__((aligned(1024))) unsigned char rg_lds[];     // 2 slab buffers
  __shared__ float s_xy[CHMAX][2 * PP];
  __shared__ __attribute__((aligned(8))) short s_box[2][CHMAX][4];   // unclipped union box of the 9 windows: x0, y0, width, height (by slot)
  __shared__ __attribute__((aligned(8))) short s_cb[2][CHMAX][4];    // clipped to the frame: x0, y0, x1, y1 (empty = dead at the level); SORTED order
  __shared__ __attribute__((aligned(8))) short s_reg[2][CHMAX][4];   // region of the round that starts at sorted index s
  __shared__ int s_be[CHMAX], s_prow[CHMAX], s_b[CHMAX], s_fj[CHMAX];
  __shared__ int s_fid[CHMAX];                                  // frame id, SORTED order
  __shared__ unsigned s_key[CHMAX];
  __shared__ int s_perm[CHMAX], s_end[CHMAX], s_rstart[CHMAX + 1];
  __shared__ int s_cnt[2];                                      // edges the rounds take / rounds
  // what a stage needs of its level, read from LDS when the stage is set up (held in scalar registers the two CorrLevel arguments
  // cost ~50 of them for the whole kernel): [l][0..1] fmap2, [2..3] s_b, [4..5] s_n, [6] s_h, [7] s_w, [8] cb_shift, [9] block bytes, [10] frame bytes
  __shared__ unsigned s_lv[2][12];

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = 2 * R + 2, Dm = D - 1;
  const unsigned lds0 = (unsigned)(uintptr_t)rg_lds;
  // debug (DEVO_RG_STATS=1): cycles of wave 0 per phase, summed over the workgroups: [0] prologue, [1] stage set-up, [2] waiting for a
  // stage's first slab, [3] products (with the next slab's DMA instructions in between), [4] left-over DMA instructions, [5] waiting for the
  // slab + barrier, [6] epilogue, [7] tap-by-tap edges + dead tail, [8] rounds, [9] workgroups, [10] slab iterations, [11] total
  unsigned long long st_t = STATS ? __builtin_readcyclecounter() : 0ULL, st_acc[STATS ? 16 : 1] = {};
  const unsigned long long st_begin = st_t;
  auto stamp = [&](int i) { if constexpr (STATS) { const unsigned long long t = __builtin_readcyclecounter(); st_acc[i] += t - st_t; st_t = t; } };
  unsigned st_iters = 0;
  // chunk of the plan: XCD x (= blockIdx % 8) owns a contiguous range of chunks
  int chunk;
  {
    const int g = blockIdx.x, per = (nchunks + 7) >> 3;
    chunk = (g & 7) * per + (g >> 3);
    if ((g >> 3) >= per || chunk >= nchunks) return;
  }
  const int nh = min(max(order[BE], 0), BE);                    // heavy slots in front: the per-edge kernel's
  const int nd = min(max(order[2 * BE + 1], 0), BE - nh);        // dead slots at the end: zeros
  const int nlive = BE - nh - nd;
  const int cstart = nh + (int)((long long)chunk * nlive / nchunks), cend = nh + (int)((long long)(chunk + 1) * nlive / nchunks);
  const int nch = min(cend - cstart, CHMAX);                    // (the launcher sizes nchunks so that nothing is cut off)
  const float inv0 = 1.0f / lv0.coord_div, inv1 = 1.0f / lv1.coord_div;
  const bool pow2 = ((__float_as_uint(lv0.coord_div) | __float_as_uint(lv1.coord_div)) & 0x807fffffu) == 0u;
  auto scaled = [&](float v, int l) -> float { return pow2 ? v * (l ? inv1 : inv0) : v / (l ? lv1.coord_div : lv0.coord_div); };
  const int64_t off0 = lv0.out_offset, off1 = lv1.out_offset;

  // ---- geometry: thread t < nch owns plan slot cstart + t
  short bx_[2][4], cb_[2][4];
  int fid_ = 0;
  if (tid < nch) {
    const int be = order[cstart + tid];
    const int b = be / E, e = be - b * E;
    const int64_t pi = ii[e], fj = jj[e];
    const float2* __restrict__ ce = reinterpret_cast<const float2*>(coords + (int64_t)be * (2 * PP));
    float cv[2 * PP];
#pragma unroll
    for (int p = 0; p < PP; p++) { const float2 v = ce[p]; cv[2 * p] = v.x; cv[2 * p + 1] = v.y; }
#pragma unroll
    for (int p = 0; p < 2 * PP; p++) s_xy[tid][p] = cv[p];
    bool slow = false, dead = true;
    int x0l0 = 0;
#pragma unroll
    for (int l = 0; l < 2; l++) {
      int xlo = 0x7fffffff, xhi = -0x7fffffff, ylo = 0x7fffffff, yhi = -0x7fffffff;
#pragma unroll
      for (int p = 0; p < PP; p++) {
        const int ox = min(max(floor_to_int(scaled(cv[p], l)) - R, -30000), 30000), oy = min(max(floor_to_int(scaled(cv[PP + p], l)) - R, -30000), 30000);
        xlo = min(xlo, ox); xhi = max(xhi, ox); ylo = min(ylo, oy); yhi = max(yhi, oy);
      }
      const int bw = xhi - xlo + D, bh = yhi - ylo + D;
      const int H2 = l ? lv1.H2 : lv0.H2, W2 = l ? lv1.W2 : lv0.W2;
      const int cx0 = max(xlo, 0), cy0 = max(ylo, 0), cx1 = min(xlo + bw, W2), cy1 = min(ylo + bh, H2);
      const bool live = cx1 > cx0 && cy1 > cy0;
      bx_[l][0] = (short)xlo; bx_[l][1] = (short)ylo; bx_[l][2] = (short)min(bw, 32767); bx_[l][3] = (short)(live ? min(bh, 32767) : 0);   // height 0 = dead at the level
      cb_[l][0] = (short)cx0; cb_[l][1] = (short)cy0; cb_[l][2] = (short)(live ? cx1 : cx0); cb_[l][3] = (short)(live ? cy1 : cy0);
      if (live) { dead = false; if ((long long)bw * bh > 16 * TMAX) slow = true; }
      if (l == 0) x0l0 = xlo;
#pragma unroll
      for (int q = 0; q < 4; q++) s_box[l][tid][q] = bx_[l][q];
    }
    fid_ = b * n2 + (int)fj;
    s_be[tid] = be; s_prow[tid] = b * Np + (int)pi; s_b[tid] = b; s_fj[tid] = (int)fj;
    // sort key: (frame, plan band of the patch centre, box origin x); dead edges behind the live ones, tap-by-tap edges last
    const int band = min(max((int)fminf(fmaxf(cv[PP + 4] * (pow2 ? inv0 : 1.0f / lv0.coord_div), 0.0f), (float)(lv0.H2 - 1)) / max(band_rows, 1), 0), 255);
    unsigned key = ((unsigned)min(fid_, 2047) << 20) | ((unsigned)band << 12) | (unsigned)min(max(x0l0 + 1024, 0), 4095);
    if (dead) key = 0xfffffff0u;
    if (slow) key = 0xffffffffu;
    s_key[tid] = key;
  }
  stamp(8);
  if (tid < 2) s_cnt[tid] = 0;
  if (tid < 2) {
    const CorrLevel& lv = tid ? lv1 : lv0;
    const unsigned long long fp = (unsigned long long)reinterpret_cast<uintptr_t>(lv.fmap2);
    s_lv[tid][0] = (unsigned)fp; s_lv[tid][1] = (unsigned)(fp >> 32);
    s_lv[tid][2] = (unsigned)(unsigned long long)lv.s_b; s_lv[tid][3] = (unsigned)((unsigned long long)lv.s_b >> 32);
    s_lv[tid][4] = (unsigned)(unsigned long long)lv.s_n; s_lv[tid][5] = (unsigned)((unsigned long long)lv.s_n >> 32);
    s_lv[tid][6] = (unsigned)lv.s_h; s_lv[tid][7] = (unsigned)lv.s_w; s_lv[tid][8] = (unsigned)lv.cb_shift;
    s_lv[tid][9] = (unsigned)lv.block_stride * ESZ; s_lv[tid][10] = lv.frame_bytes;
  }
  __syncthreads();
  if (tid < nch) {                                              // rank sort (ties by slot); frame ids and clipped boxes move to sorted order
    const unsigned k = s_key[tid];
    int rank = 0;
    for (int j = 0; j < nch; j++) { const unsigned kj = s_key[j]; rank += (kj < k || (kj == k && j < tid)) ? 1 : 0; }
    s_perm[rank] = tid;
    s_fid[rank] = fid_;
#pragma unroll
    for (int l = 0; l < 2; l++)
#pragma unroll
      for (int q = 0; q < 4; q++) s_cb[l][rank][q] = cb_[l][q];
    if (k != 0xffffffffu) atomicAdd(&s_cnt[0], 1);
  }
  __syncthreads();
  stamp(9);
  const int nround_edges = __builtin_amdgcn_readfirstlane(s_cnt[0]);   // sorted indices [0, nround_edges) go through rounds
  {
    // greedy round from every start s (thread s): edges s .. s_end[s] - 1 and their union region per level
    const int nrmin = (nround_edges + SLOTS - 1) / SLOTS, rlim = nrmin > 0 ? (nround_edges + nrmin - 1) / nrmin : 1;   // balanced rounds
    if (tid < nround_edges) {
      const int f0 = s_fid[tid];
      int X[2][4];
#pragma unroll
      for (int l = 0; l < 2; l++) { X[l][0] = 32767; X[l][1] = 32767; X[l][2] = -32768; X[l][3] = -32768; }
      int e = tid;
      for (; e < nround_edges && e - tid < rlim; e++) {
        if (s_fid[e] != f0) break;
        int N[2][4];
        bool fits = true;
#pragma unroll
        for (int l = 0; l < 2; l++) {
          const int a0 = s_cb[l][e][0], a1 = s_cb[l][e][1], a2 = s_cb[l][e][2], a3 = s_cb[l][e][3];
          const bool live = a2 > a0 && a3 > a1;
          N[l][0] = live ? min(X[l][0], a0) : X[l][0]; N[l][1] = live ? min(X[l][1], a1) : X[l][1];
          N[l][2] = live ? max(X[l][2], a2) : X[l][2]; N[l][3] = live ? max(X[l][3], a3) : X[l][3];
          if (max(N[l][2] - N[l][0], 0) * max(N[l][3] - N[l][1], 0) > CAP) fits = false;
        }
        if (!fits) break;                                       // (never at e == tid: one edge is <= 16 TMAX positions)
#pragma unroll
        for (int l = 0; l < 2; l++) { X[l][0] = N[l][0]; X[l][1] = N[l][1]; X[l][2] = N[l][2]; X[l][3] = N[l][3]; }
      }
      s_end[tid] = e;
#pragma unroll
      for (int l = 0; l < 2; l++) {
        const bool any = X[l][2] > X[l][0] && X[l][3] > X[l][1];
        s_reg[l][tid][0] = (short)(any ? X[l][0] : 0); s_reg[l][tid][1] = (short)(any ? X[l][1] : 0);
        s_reg[l][tid][2] = (short)(any ? X[l][2] : 0); s_reg[l][tid][3] = (short)(any ? X[l][3] : 0);
      }
    }
    __syncthreads();
    if (tid == 0) {
      int nr = 0;
      for (int s = 0; s < nround_edges; s = s_end[s]) s_rstart[nr++] = s;
      s_rstart[nr] = nround_edges;
      s_cnt[1] = nr;
    }
    __syncthreads();
  }
  const int nrounds = __builtin_amdgcn_readfirstlane(s_cnt[1]);
  stamp(10);

  const int m = lane & 15, kg = lane >> 4;                      // MFMA operand row / column, k group
  const int ep = lane % PP, ea0 = lane / PP;                    // epilogue: pixel, window row (lane 63 idles)
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(fmap1t), 0, f1t_bytes, 0x00020000);
  const int NS = C / CS;
  const bool paired = (ols == 2 && off1 == off0 + 1 && (oes & 1) == 0 && (off0 & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & (2 * ESZ - 1)) == 0);

  // Blend of one (edge, level) for this lane's pixel out of a scratch area: tap (a, c) at scr[base + a * pitch + c].
  // o[pass][cx] = the reference's blend4 of window row a0 + 7 pass (correlation_kernel.cu:227-230, same operation order).
  auto blend_rows = [&](const float* scr, int base, int pitch, float dx, float dy, bool have, float (&o)[NPASS][DM]) {
    float w00, w01, w10, w11;
    {
#pragma clang fp contract(off)
      w00 = (1.0f - dx) * (1.0f - dy); w01 = dx * (1.0f - dy); w10 = (1.0f - dx) * dy; w11 = dx * dy;
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ps++) {
      const int a = ea0 + 7 * ps;
      const bool act = have && lane < 63 && a < Dm;
      float r0[DMAX], r1[DMAX];
      const float* rp = scr + base + a * pitch;
#pragma unroll
      for (int j = 0; j < DMAX; j++) { r0[j] = (act && j < D) ? rp[j] : 0.0f; r1[j] = (act && j < D) ? rp[pitch + j] : 0.0f; }
#pragma unroll
      for (int cx = 0; cx < DM; cx++) {
        float v;
        {
#pragma clang fp contract(off)
          v = w00 * r0[cx]; v = v + w01 * r0[cx + 1]; v = v + w10 * r1[cx]; v = v + w11 * r1[cx + 1];
        }
        o[ps][cx] = v;
      }
    }
  };
  // the lane's pixel of edge t at level l: window origin and sub-pixel fractions
  auto pixel_geo = [&](int t, int l, int& ox, int& oy, float& dx, float& dy) {
    const float qx = scaled(s_xy[t][ep], l), qy = scaled(s_xy[t][PP + ep], l);
    ox = min(max(floor_to_int(qx) - R, -30000), 30000); oy = min(max(floor_to_int(qy) - R, -30000), 30000);
    dx = qx - floorf(qx); dy = qy - floorf(qy);
  };
  // both levels' outputs of the lane's window rows -> the edge's record.  32-bit element offsets from a wave-uniform base, formed where they
  // are used (hipcc otherwise hoists seven 64-bit offsets out of the stage loop: 14 registers for the whole kernel)
  auto store_rows = [&](int be, const float (&o0)[NPASS][DM], const float (&o1)[NPASS][DM]) {
    T* op = out + (int64_t)__builtin_amdgcn_readfirstlane(be) * oes;
    int ep_ = ep, ea_ = ea0;
    asm volatile("" : "+v"(ep_), "+v"(ea_));
    const unsigned ols32 = (unsigned)ols, o0_32 = (unsigned)off0, o1_32 = (unsigned)off1;       // (the launcher checks that they fit)
#pragma unroll
    for (int ps = 0; ps < NPASS; ps++) {
      const int a = ea_ + 7 * ps;
      if (lane < 63 && a < Dm) {
        unsigned el = (unsigned)(a * PP + ep_);
        const unsigned step = (unsigned)(Dm * PP);
#pragma unroll
        for (int cx = 0; cx < DM; cx++) {
          if (cx < Dm) {
            if (paired) {
              if constexpr (HALF) {
                const unsigned short u0 = __half_as_ushort(__float2half(o0[ps][cx])), u1 = __half_as_ushort(__float2half(o1[ps][cx]));
                *reinterpret_cast<unsigned*>(op + (size_t)(el * 2u + o0_32)) = (unsigned)u0 | ((unsigned)u1 << 16);
              } else {
                typedef float f2v __attribute__((ext_vector_type(2)));
                const f2v v = {o0[ps][cx], o1[ps][cx]};
                *reinterpret_cast<f2v*>(op + (size_t)(el * 2u + o0_32)) = v;
              }
            } else {
              store_streamed(op + (size_t)(el * ols32 + o0_32), from_f32<T>(o0[ps][cx]));
              store_streamed(op + (size_t)(el * ols32 + o1_32), from_f32<T>(o1[ps][cx]));
            }
            el += step;
          }
        }
      }
    }
  };

  // ===================================================== stages: (round 0, level 1), (round 0, level 0), (round 1, level 1), ...
  // What the DMA of a stage needs (wave-uniform except the per-lane offsets): computed one stage AHEAD, so that the first slab of
  // the next stage is in flight under the current stage's epilogue.  Slab `it` of a stage lands in buffer (it + 1) & 1; the epilogue's
  // scratch lives in buffer 0.
  struct DmaP {
    int nchk, U, NI;                  // 64-position pieces per plane, slab units per iteration, iterations (0: nothing to stage)
    unsigned PS, boff, bb;            // plane stride, offset of the patch slabs, byte stride of a channel block
    int sh;
    __amdgpu_buffer_rsrc_t rs;
    unsigned v0, v1;                  // this lane's offsets in the frame for its (at most two) pieces of a plane
  };
  static_assert(DPW <= 2 && BPW <= 2, "a wave has at most two pieces per plane / per patch slab");
  unsigned vB0 = OFF_NONE, vB1 = OFF_NONE, vBn0 = OFF_NONE, vBn1 = OFF_NONE;    // ... and in the patches: current round, next round
  auto dma_params = [&](int stage, DmaP& P) {
    const int r = stage >> 1, l = 1 - (stage & 1);
    auto lvw = [&](int i) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)s_lv[l][i]); };
    const int rs_ = __builtin_amdgcn_readfirstlane(s_rstart[r]), re_ = __builtin_amdgcn_readfirstlane(s_rstart[r + 1]);
    const int X0 = __builtin_amdgcn_readfirstlane((int)s_reg[l][rs_][0]), Y0 = __builtin_amdgcn_readfirstlane((int)s_reg[l][rs_][1]);
    const int X1 = __builtin_amdgcn_readfirstlane((int)s_reg[l][rs_][2]), Y1 = __builtin_amdgcn_readfirstlane((int)s_reg[l][rs_][3]);
    const int RW = X1 - X0, npos = RW * (Y1 - Y0);
    P.nchk = (npos + 64) >> 6;
    P.PS = (unsigned)P.nchk * 1024u;
    int U = 1;
    while (2 * U <= NS && NS % (2 * U) == 0 && (unsigned)(2 * U) * (4u * P.PS + (unsigned)BUNIT) <= (unsigned)BUFSZ) U *= 2;
    P.U = U; P.NI = npos > 0 ? NS / U : 0;
    P.boff = (unsigned)U * 4u * P.PS;
    P.sh = (int)lvw(8); P.bb = lvw(9);
    const int t_first = __builtin_amdgcn_readfirstlane(s_perm[rs_]);
    const long long sb = (long long)(((unsigned long long)lvw(3) << 32) | lvw(2)), sn = (long long)(((unsigned long long)lvw(5) << 32) | lvw(4));
    const T* fbase = reinterpret_cast<const T*>((uintptr_t)(((unsigned long long)lvw(1) << 32) | lvw(0))) +
                     (long long)__builtin_amdgcn_readfirstlane(s_b[t_first]) * sb + (long long)__builtin_amdgcn_readfirstlane(s_fj[t_first]) * sn;
    P.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(fbase), 0, lvw(10), 0x00020000);
    const int sh_ = (int)lvw(6), sw_ = (int)lvw(7);
    const float inv_rw = __builtin_amdgcn_rcpf((float)max(RW, 1));
    auto frame_off = [&](int j) -> unsigned {
      const int pos = 64 * (w + j * NW) + lane;
      const int py = (int)(((float)pos + 0.5f) * inv_rw), px = pos - py * RW;
      return pos < npos ? (unsigned)((Y0 + py) * sh_ + (X0 + px) * sw_) * ESZ : OFF_NONE;
    };
    P.v0 = frame_off(0);
    P.v1 = DPW > 1 ? frame_off(1) : OFF_NONE;
    if (l == 1) {                                                 // a new round: its patch slabs (piece c' = w + j NW: 16 (slot, pixel) pairs, 4 lanes each)
      auto patch_off = [&](int j) -> unsigned {
        const int pair = 16 * (w + j * NW) + (lane >> 2), slot = pair / PP, px = pair - slot * PP, idx = rs_ + slot;
        unsigned v = OFF_NONE;
        if (slot < SLOTS && idx < re_) v = (unsigned)((s_prow[s_perm[idx]] * PP + px) * C) * ESZ + (unsigned)(lane & 3) * 16u;
        return v;
      };
      vBn0 = patch_off(0);
      vBn1 = BPW > 1 ? patch_off(1) : OFF_NONE;
    }
  };
  // DMA instruction (unit u, number rem < OPU) of slab `it`: false = not this wave's / beyond the region
  auto issue_op = [&](const DmaP& P, unsigned pB0, unsigned pB1, int it, int u, int rem) -> bool {
    const unsigned unit = (unsigned)(it * P.U + u);
    const unsigned base = lds0 + (unsigned)(S::NBUF == 2 ? ((it + 1) & 1) : 0) * (unsigned)BUFSZ;
    if (rem < 4 * DPW) {
      const int q = rem / DPW, j = rem - q * DPW, c = w + j * NW;     // (DPW is a constant)
      if (c >= P.nchk) return false;
      const unsigned ch = unit * CS + (unsigned)q * PCH, blk = ch >> P.sh, so = blk * P.bb + (ch - (blk << P.sh)) * ESZ;
      rg_dma16(j ? P.v1 : P.v0, P.rs, so, base + ((unsigned)(u * 4 + q)) * P.PS + (unsigned)c * 1024u);
    } else {
      const int j = rem - 4 * DPW, c = w + j * NW;
      if (c >= NBCH) return false;
      rg_dma16(j ? pB1 : pB0, rsB, unit * 64u, base + P.boff + (unsigned)u * BUNIT + (unsigned)c * 1024u);
    }
    return true;
  };
  auto issue_slab = [&](const DmaP& P, unsigned pB0, unsigned pB1, int it) {
    for (int u = 0; u < P.U; u++)
      for (int rem = 0; rem < OPU; rem++) issue_op(P, pB0, pB1, it, u, rem);
  };
  // fp32 storage: the slab in buffer bi, 16-byte piece by 16-byte piece, 4 floats -> (hi0..3 | lo0..3) in place (the patch slabs
  // arrive split).  All threads; the caller puts barriers around it.
  auto split_slab = [&](const DmaP& P, int bi) {
    if constexpr (!HALF) {
      unsigned char* buf = rg_lds + (size_t)bi * BUFSZ;
      const int npieces = P.U * 4 * P.nchk * 64;
      for (int i = tid; i < npieces; i += S::THREADS) {
        rg_f4* p = reinterpret_cast<rg_f4*>(buf + (size_t)i * 16);
        *reinterpret_cast<rg_h8*>(p) = rg_split4(*p);
      }
    }
  };

  float res1[EW][NPASS][DM];                                    // level-1 outputs of the lane's rows (wait for level 0)
#pragma unroll
  for (int k = 0; k < EW; k++)
#pragma unroll
    for (int ps = 0; ps < NPASS; ps++)
#pragma unroll
      for (int cx = 0; cx < DM; cx++) res1[k][ps][cx] = 0.0f;

  const int nstages = 2 * nrounds;
  DmaP cur;
  if (nstages > 0) {
    dma_params(0, cur);
    vB0 = vBn0; vB1 = vBn1;
    if (cur.NI > 0) issue_slab(cur, vB0, vB1, 0);
  }
  for (int stage = 0; stage < nstages; stage++) {
    const int r = stage >> 1, l = 1 - (stage & 1);               // level 1 first: its results wait in registers
    if constexpr (!S::XPRE) {
      if (stage > 0) {                                            // (the buffers were the previous stage's scratch until now)
        dma_params(stage, cur);
        if (l == 1) { vB0 = vBn0; vB1 = vBn1; }
        if (cur.NI > 0) issue_slab(cur, vB0, vB1, 0);
      }
    }
    const int rs = __builtin_amdgcn_readfirstlane(s_rstart[r]), re = __builtin_amdgcn_readfirstlane(s_rstart[r + 1]);
    const int X0 = __builtin_amdgcn_readfirstlane((int)s_reg[l][rs][0]), Y0 = __builtin_amdgcn_readfirstlane((int)s_reg[l][rs][1]);
    const int X1 = __builtin_amdgcn_readfirstlane((int)s_reg[l][rs][2]), Y1 = __builtin_amdgcn_readfirstlane((int)s_reg[l][rs][3]);
    const int RW = X1 - X0, npos = RW * (Y1 - Y0);
    // ---- this wave's edges at this level: tiles, box, packed tile addresses (two 16-bit byte offsets per register)
    int tk[EW], nt[EW], bx0[EW], by0[EW], bwk[EW];
    unsigned apk[EW][TMAX / 2];
    rg_f4 acc[EW][TMAX];
#pragma unroll
    for (int k = 0; k < EW; k++) {
      const int idx = rs + k * NW + w;
      tk[k] = idx < re ? __builtin_amdgcn_readfirstlane(s_perm[idx]) : -1;
      nt[k] = 0; bx0[k] = 0; by0[k] = 0; bwk[k] = 1;
      int ncell = 1;
      if (tk[k] >= 0) {
        const int t = tk[k];
        bx0[k] = __builtin_amdgcn_readfirstlane((int)s_box[l][t][0]); by0[k] = __builtin_amdgcn_readfirstlane((int)s_box[l][t][1]);
        bwk[k] = __builtin_amdgcn_readfirstlane((int)s_box[l][t][2]);
        ncell = __builtin_amdgcn_readfirstlane(bwk[k] * (int)s_box[l][t][3]);        // (height 0: dead at this level)
        nt[k] = (ncell + 15) >> 4;
        ncell = max(ncell, 1);
      }
      const float inv_bw = __builtin_amdgcn_rcpf((float)bwk[k]);
#pragma unroll
      for (int t = 0; t < TMAX; t++) {
        const int i = min(16 * t + m, ncell - 1);
        const int iy = (int)(((float)i + 0.5f) * inv_bw), ix = i - iy * bwk[k];
        const int gx = bx0[k] + ix, gy = by0[k] + iy;
        const bool in = gx >= X0 && gx < X1 && gy >= Y0 && gy < Y1;
        const unsigned pos16 = (unsigned)(in ? (gy - Y0) * RW + (gx - X0) : npos) * 16u;       // (outside the frame: the zero slot)
        if (t & 1) apk[k][t >> 1] |= pos16 << 16; else apk[k][t >> 1] = pos16;
      }
#pragma unroll
      for (int t = 0; t < TMAX; t++) acc[k][t] = rg_f4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned kgPS = (unsigned)kg * cur.PS;
    stamp(1);
    if (cur.NI > 0) {                                             // (block-uniform) else: every edge of the round is dead at this level
      rg_wait_dma();                                              // slab 0 was requested a stage ago
      rg_barrier();
      stamp(2);
      for (int it = 0; it < cur.NI; it++) {
        const int bi = S::NBUF == 2 ? ((it + 1) & 1) : 0;
        if constexpr (S::NBUF == 1) {
          if (it > 0) { issue_slab(cur, vB0, vB1, it); rg_wait_dma(); rg_barrier(); }
        }
        if constexpr (!HALF) { split_slab(cur, bi); rg_barrier(); }
        // Half of the waves request the next slab BEFORE their products, the other half AFTER: the two waves of a SIMD (w, w + NW / 2)
        // are in opposite phases, so the matrix pipe works while the partner sits in its DMA instructions (a wave is held at each of
        // them until the memory pipeline takes it: the CU's fill rate, ~20 B/clk, prices them at ~150-250 cycles apiece).
        const unsigned char* buf = rg_lds + (size_t)bi * BUFSZ;
        const bool more = S::NBUF == 2 && it + 1 < cur.NI;
        const bool tr = STATS && chunk == 100 && stage == 1 && it < 4 && lane == 0;
        unsigned long long* trp = stats + 16 + (it * NW + w) * 4;
        if (tr) trp[0] = __builtin_readcyclecounter();
#pragma unroll 1
        for (int phase = 0; phase < 2; phase++) {
          if (phase == 1 && tr) trp[1] = __builtin_readcyclecounter();
          if ((phase == 0) == (w < NW / 2)) {
            if (more) issue_slab(cur, vB0, vB1, it + 1);
            continue;
          }
          if (S::NBUF == 1 && phase != (w < NW / 2 ? 1 : 0)) continue;
          for (int u = 0; u < cur.U; u++) {
            const unsigned char* ra = buf + (size_t)u * 4 * cur.PS + kgPS;
            const unsigned char* rb = buf + cur.boff + (size_t)u * BUNIT + (size_t)(min(m, PP - 1) * 64 + kg * 16);
#pragma unroll
            for (int k = 0; k < EW; k++) {
              if (nt[k] > 0) {                                   // (wave-uniform)
                const rg_u4 yu = *reinterpret_cast<const rg_u4*>(rb + (size_t)((k * NW + w) * PP * 64));
                rg_h8 b1, b2;
                if constexpr (HALF) { b1 = __builtin_bit_cast(rg_h8, yu); b2 = b1; }
                else {
                  b1 = __builtin_bit_cast(rg_h8, rg_u4{yu[0], yu[1], yu[0], yu[1]});                            // [yh | yh]
                  b2 = __builtin_bit_cast(rg_h8, rg_u4{yu[2], yu[3], 0u, 0u});                                  // [yl | 0]
                }
#pragma unroll
                for (int tb = 0; tb < TMAX; tb += 4) {            // batches of 4 tiles: the reads go out together (surplus tiles re-read the last cell)
                  if (tb < nt[k]) {
                    constexpr int NB = 4;
                    static_assert(TMAX % NB == 0, "tiles go in batches of 4");
                    rg_h8 av[NB];
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                      const unsigned pk = apk[k][(tb + j) >> 1];
                      av[j] = *reinterpret_cast<const rg_h8*>(ra + (((tb + j) & 1) ? (pk >> 16) : (pk & 0xffffu)));
                    }
#pragma unroll
                    for (int j = 0; j < NB; j++) acc[k][tb + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[j], b1, acc[k][tb + j], 0, 0, 0);
                    if constexpr (!HALF) {
#pragma unroll
                      for (int j = 0; j < NB; j++) acc[k][tb + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[j], b2, acc[k][tb + j], 0, 0, 0);
                    }
                  }
                }
              }
            }
          }
        }
        stamp(3);
        if (tr) trp[2] = __builtin_readcyclecounter();
        if constexpr (S::NBUF == 2) rg_wait_dma();
        rg_barrier();                                             // (single buffer: everybody has read the slab)
        if (tr) trp[3] = __builtin_readcyclecounter();
        stamp(5);
        if constexpr (STATS) st_iters++;
      }
    }
    // ---- the next stage's first slab goes out before this stage's epilogue (buffer 1; the scratch is in buffer 0)
    DmaP nxt = cur;
    if constexpr (S::XPRE) {
      if (stage + 1 < nstages) {
        dma_params(stage + 1, nxt);
        if (nxt.NI > 0) issue_slab(nxt, l == 0 ? vBn0 : vB0, l == 0 ? vBn1 : vB1, 0);   // (a new round brings its own patch slabs)
      }
    }
    stamp(11);
    // ---- epilogue of the level: tiles -> scratch -> blended rows
    float* scr = reinterpret_cast<float*>(rg_lds + (size_t)w * S::SCRW);
#pragma unroll
    for (int k = 0; k < EW; k++) {
      if (tk[k] >= 0) {                                          // (wave-uniform)
        float o[NPASS][DM];
        const bool have = nt[k] > 0;
        if (have) {
          if (m < PP) {
#pragma unroll
            for (int t = 0; t < TMAX; t++)
              if (t < nt[k]) *reinterpret_cast<rg_f4*>(scr + m * SP + 16 * t + 4 * kg) = acc[k][t];
          }
          rg_lds_fence();
        }
        stamp(12);
        int ox, oy; float dx, dy;
        pixel_geo(tk[k], l, ox, oy, dx, dy);
        blend_rows(scr, ep * SP + (oy - by0[k]) * bwk[k] + (ox - bx0[k]), bwk[k], dx, dy, have, o);
        if (l == 1) {
#pragma unroll
          for (int ps = 0; ps < NPASS; ps++)
#pragma unroll
            for (int cx = 0; cx < DM; cx++) res1[k][ps][cx] = o[ps][cx];
        } else {
          store_rows(s_be[tk[k]], o, res1[k]);
        }
        if (have) rg_lds_fence();
        stamp(13);
      }
    }
    rg_barrier();                                                // buffer 0 is a DMA target again
    stamp(14);
    if constexpr (S::XPRE) {
      if (l == 0) { vB0 = vBn0; vB1 = vBn1; }
      cur = nxt;
    }
    stamp(6);
  }
  rg_wait_dma();

  // ===================================================== edges the rounds could not take: tap by tap, one wave per edge
  for (int i = nround_edges + w; i < nch; i += NW) {
    const int t = __builtin_amdgcn_readfirstlane(s_perm[i]);
    float* scr = reinterpret_cast<float*>(rg_lds + (size_t)w * S::SCRW);
    float o1[NPASS][DM], o0[NPASS][DM];
    const int64_t prow = s_prow[t];
#pragma unroll 1
    for (int li = 0; li < 2; li++) {
      const int l = 1 - li;
      const CorrLevel& lv = l ? lv1 : lv0;
      const T* fbase = static_cast<const T*>(lv.fmap2) + (int64_t)s_b[t] * lv.s_b + (int64_t)s_fj[t] * lv.s_n;
      const int sh = lv.cb_shift;
      const int64_t bs = lv.block_stride;
      const int ntap = D * D;
      for (int idx = lane; idx < PP * ntap; idx += 64) {
        const int p = idx / ntap, rr = idx - p * ntap, a = rr / D, c = rr - a * D;
        const int gx = floor_to_int(scaled(s_xy[t][p], l)) - R + c, gy = floor_to_int(scaled(s_xy[t][PP + p], l)) - R + a;
        float sum = 0.0f;
        if (gx >= 0 && gx < lv.W2 && gy >= 0 && gy < lv.H2) {
          const T* fp = fbase + (int64_t)gy * lv.s_h + (int64_t)gx * lv.s_w;
          const T* gp = fmap1t + (prow * PP + p) * C;
          for (int k = 0; k < C; k++) {
            const int blk = k >> sh;
            float g;
            if constexpr (HALF) g = to_f32(gp[k]);
            else {                                               // (hi0..3 | lo0..3) per 4 channels
              const _Float16* hp = reinterpret_cast<const _Float16*>(gp) + (k >> 2) * 8 + (k & 3);
              g = (float)hp[0] + (float)hp[4];
            }
            sum += g * to_f32(fp[(int64_t)blk * bs + (k - (blk << sh))]);
          }
        }
        scr[p * SP + rr] = sum;
      }
      wave_lds_fence();
      int ox, oy; float dx, dy;
      pixel_geo(t, l, ox, oy, dx, dy);
      if (l == 1) blend_rows(scr, ep * SP, D, dx, dy, true, o1); else blend_rows(scr, ep * SP, D, dx, dy, true, o0);
      wave_lds_fence();
    }
    store_rows(s_be[t], o0, o1);
  }
  // ===================================================== zero records for this chunk's share of the dead tail
  {
    const int nel = Dm * Dm * PP;
    for (int i = chunk; i < nd; i += nchunks) {
      const int be = order[BE - nd + i];
      T* o = out + (int64_t)be * oes;
      for (int j = tid; j < nel; j += S::THREADS) { o[(int64_t)j * ols + off0] = from_f32<T>(0.0f); o[(int64_t)j * ols + off1] = from_f32<T>(0.0f); }
    }
  }
  stamp(7);
  if constexpr (STATS) if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 8; i++) atomicAdd(&stats[i], st_acc[i]);
#pragma unroll
    for (int i = 8; i < 16; i++) atomicAdd(&stats[392 + i], st_acc[i]);            // [24..31]: sub-phases (prologue: 8 loads, 9 sort, 10 rounds; epilogue: 11 request, 12 scratch, 13 blend + store, 14 barrier)
    atomicAdd(&stats[8], (unsigned long long)nrounds); atomicAdd(&stats[9], 1ULL); atomicAdd(&stats[10], (unsigned long long)st_iters);
    atomicAdd(&stats[11], __builtin_readcyclecounter() - st_begin);
  }
}

// fmap1 [N][C][9] -> [N][9][C], the patch operand of the region kernel: 16 contiguous bytes per (pixel, 4 | 8 channels).  fp32: every group
// of 4 channels is stored as fp16 (hi0..3 | lo0..3), x = hi + lo — the form the kernel multiplies (same 16 bytes).
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
    for (int i = threadIdx.x; i < (C / 4) * PP; i += 256) {      // (C % 16 == 0 for the region kernel)
      const int p = i / (C / 4), c4 = i - p * (C / 4);
      const rg_f4 x = {(float)s[(4 * c4) * PP + p], (float)s[(4 * c4 + 1) * PP + p], (float)s[(4 * c4 + 2) * PP + p], (float)s[(4 * c4 + 3) * PP + p]};
      reinterpret_cast<rg_h8*>(o)[i] = rg_split4(x);
    }
  }
}
