// altcorr lookup on the DENSE matrix-core shape (v_mfma_f32_16x16x32_f16), region-staged through LDS for a GROUP of edges
// (included by corr.hip inside namespace devo).  Semantics: correlation_kernel.cu:82-136 (raw window) + :221-232 (blend,
// permute); devo.py:215-217 (two levels stacked).
//
// Why another kernel: corr_fwd_mfma_kernel (one wave per edge) pulls every edge's own ~27 KB (fp16) box of BOTH levels through
// the CU's memory path — 1.05 GB per cfg2 launch — and a CU takes in only ~11 B/cycle from HBM / Infinity Cache and ~22-28
// B/cycle from L2 (tools/ubench/dma_fill.hip), which is where its 99 us come from.  At DEVO's patch density each pyramid pixel
// is wanted by 8 (level 0) to 90 (level 1) edges, so here
//   * a WORKGROUP (8 waves, one per CU) takes CHUNK consecutive edges of the locality plan (same target frame, neighbouring
//     image positions) and forms ROUNDS of <= SLOTS edges whose union REGION per level fits a slab buffer;
//   * PRODUCER / CONSUMER split: NLOAD loader waves do nothing but plan rounds and request slabs with LDS-DMA loads
//     (global_load_lds_dwordx4: no staging registers, no ds_write pass), NCOMP compute waves multiply and write results.  The
//     kernel is ONE stream of slabs (round 0 slab 0, 1, .., round 1 slab 0, ..) over two LDS buffers with one workgroup barrier
//     per slab: behind barrier g slab g has landed and slab g - 1 is consumed, so the loaders request slab g + 1 — also the
//     NEXT round's first, while the compute waves are still in the current round's epilogue — and the compute waves multiply
//     slab g.  A slab = one channel slab (32 channels fp16 / 16 channels fp32) of BOTH levels' regions and of the items' patches;
//   * LDS image of a slab: planar [4 pieces][position][16 B] per level (+ the patch slabs, raw): for fp16 the A operand of the
//     MFMA is one conflict-free ds_read_b128 per 16 positions;
//   * the products run on v_mfma_f32_16x16x32_f16: M = 16 consecutive box positions of an edge, N = 16 patch pixels (9 used),
//     K = 32 channels; a compute wave keeps the accumulators of its EPW edges (both levels, <= MAXT0 + MAXT1 tiles of 16
//     positions) in registers across the channel slabs, so a staged slab serves all edges of the round;
//   * fp16 storage: operands as stored, fp32 accumulation.  fp32 storage: the compute waves split every value they read exactly
//     into hi + lo fp16 parts (x * 16 = hi + lo, 22 significant bits; the remainder is below fp32's own rounding for |x| >= 2^-7
//     and 2e-9 absolute below that); the MFMA sees K = [hi(16) | lo(16)] against B1 = [hi | hi] and B2 = [lo | lo] of the patch,
//     i.e. the full product (a_hi + a_lo)(b_hi + b_lo) with fp32 accumulation — fp32-class results (measured against the fp64
//     oracle in tests/test_gpu_altcorr.py) at 1/4 of the matrix-pipe time of the exact fp32 MFMA.  Domain of the fp32 path:
//     |feature| < 4094 (fp16 range after the x16 prescale); DEVO_CORR_DENSE=0 selects the exact kernel;
//   * the epilogue (wave-private raw window areas outside the slab buffers) blends level 1 into registers, then level 0, and
//     writes the interleaved two-level record (torch.stack([c0, c1], -1)) with full-width stores.
// Edges whose box exceeds the tile capacity at either level ("heavy": patch pixels spread apart) are processed as 9
// single-pixel items (window = box), any coordinates work; rounds are formed greedily so that the regions fit their buffers.
#pragma once

typedef _Float16 dn_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 dn_h2 __attribute__((ext_vector_type(2)));
typedef float dn_f4 __attribute__((ext_vector_type(4)));

constexpr int DN_WAVES = 8;                                 // ONE 512-thread workgroup per CU (2 waves per SIMD)
constexpr int DN_THREADS = DN_WAVES * 64;
constexpr int DN_NLOAD = 2;                                 // loader waves
constexpr int DN_NCOMP = DN_WAVES - DN_NLOAD;               // compute waves
constexpr float DN_PRESCALE = 16.0f;                        // fp32 storage: x * 16 = hi + lo (both operands)
constexpr float DN_UNSCALE = 1.0f / 256.0f;

template <int RMAX, int NL> struct DnShape {
  static constexpr int MAXT0 = RMAX <= 3 ? 10 : 16;         // 16-position tiles per item, finest level of the call
  static constexpr int MAXT1 = NL == 2 ? (RMAX <= 3 ? 6 : 12) : 0;   // ... coarse level
  static constexpr int EPW = RMAX <= 3 ? 2 : 1;             // items per compute wave
  static constexpr int SLOTS = DN_NCOMP * EPW;              // items per round
  static constexpr int CHUNK = 3 * SLOTS;                   // edges per workgroup
  static constexpr int R0CH = RMAX <= 3 ? 9 : 8;            // 64-position chunks of the level-0 region
  static constexpr int R1CH = NL == 2 ? 3 : 0;              // ... of the coarse level's region
  static constexpr int NPCH = (SLOTS * 36 + 63) / 64;       // 1 KB chunks of the patch part (36 16-byte pieces per item)
  static constexpr int BUFSZ = (R0CH + R1CH) * 4096 + NPCH * 1024;
  static constexpr int BOXSP = 16 * MAXT0 + 4;              // row pitch of the raw result area (conflict-free b128 dumps)
  static constexpr int RAWSZ = 9 * BOXSP * 4;               // a compute wave's raw result area
  static constexpr int DMAX = 2 * RMAX + 2;
  static constexpr int NQ = ((DMAX - 1) * (DMAX - 1) + 6) / 7;   // epilogue rounds of 63 outputs
};

// inclusive prefix min over lanes 0..15 of a row (row_shr 1, 2, 4, 8; lanes without a source keep their own value)
__device__ __forceinline__ int dn_prefix_min16(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x111, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x112, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x114, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x118, 0xf, 0xf, false));
  return v;
}

// LDS-DMA: 16 bytes per lane from the lane's global address to LDS byte offset lds_off + 16 * lane (lds_off wave-uniform).
// Issued through asm so that hipcc's waitcnt insertion does not drain it at the next LDS access or barrier; completion is
// awaited with dn_wait_dma() (extra outstanding loads only make the compiler's own vmcnt waits stricter).
__device__ __forceinline__ void dn_dma16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_off) : "memory");
}
__device__ __forceinline__ void dn_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// workgroup barrier that leaves vector-memory operations (the DMA) in flight
__device__ __forceinline__ void dn_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, int RMAX, int NL>
__global__ __launch_bounds__(DN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void corr_fwd_dense_kernel(
    const T* __restrict__ fmap1, CorrLevel lv0, CorrLevel lv1, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, T* __restrict__ out, int BE, int E, int Np, int n2,
    int C, int64_t out_estride, int64_t out_lstride, int R, const int* __restrict__ order,
    unsigned long long* __restrict__ stats) {
  using S = DnShape<RMAX, NL>;
  constexpr bool HALF = sizeof(T) == 2;
  constexpr int SC = HALF ? 32 : 16;                        // channels per slab
  constexpr unsigned ESZ = sizeof(T);
  constexpr int MAXT0 = S::MAXT0, MAXT1 = S::MAXT1, EPW = S::EPW, SLOTS = S::SLOTS, CHUNK = S::CHUNK, BOXSP = S::BOXSP, NQ = S::NQ;
  constexpr int NPCH = S::NPCH, BUFSZ = S::BUFSZ, RAWSZ = S::RAWSZ;
  constexpr int PPIECES = 36;                               // 16-byte pieces of one patch slab (SC * 9 * ESZ = 576 B)
  constexpr int MT1 = MAXT1 > 0 ? MAXT1 : 1;
  constexpr int L0PW = (S::R0CH + DN_NLOAD - 1) / DN_NLOAD, L1PW = (S::R1CH + DN_NLOAD - 1) / DN_NLOAD, PPW = (NPCH + DN_NLOAD - 1) / DN_NLOAD;
  static_assert(16 * MAXT0 <= S::R0CH * 64 && 16 * MAXT1 <= S::R1CH * 64 + (NL == 1), "a single item must fit the region");
  static_assert(S::R0CH * 4096 <= 65536 && S::R1CH * 4096 <= 65536, "16-bit tile addresses");
  static_assert(SLOTS <= 16, "the round planner scans 16 candidates");
#define LVF(l, F) ((NL == 2 && (l)) ? lv1.F : lv0.F)

  __shared__ __attribute__((aligned(1024))) unsigned char s_arena[2 * BUFSZ];
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[DN_NCOMP * RAWSZ];
  __shared__ int s_ox[NL][CHUNK][PP], s_oy[NL][CHUNK][PP];
  __shared__ float s_dx[NL][CHUNK][PP], s_dy[NL][CHUNK][PP];
  __shared__ int s_box[NL][CHUNK][4];                       // xmin, ymin, bw, bh of the edge's union box
  __shared__ int s_be[CHUNK], s_frame[CHUNK], s_pi[CHUNK], s_heavy[CHUNK];
  __shared__ int s_item[2][SLOTS][3];                       // per round parity: edge (chunk index), first pixel, pixel count
  __shared__ int s_reg[2][NL][4];                           // per round parity: region x0, y0, width, height
  __shared__ int s_nit[2];                                  // items of the round (0 = no round)
  __shared__ int s_ctl[2];                                  // next edge, next pixel of a heavy edge

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = 2 * R + 2, Dm = D - 1;
  const unsigned arena0 = (unsigned)(uintptr_t)s_arena;     // LDS byte offset of the slab buffers
  // chunk of the plan this workgroup owns: XCD x gets a contiguous range of chunks (workgroup g runs on XCD g % 8)
  const int nchunks = (BE + CHUNK - 1) / CHUNK;
  int chunk;
  {
    const int g = blockIdx.x, x = g & 7, per = (nchunks + 7) >> 3;
    chunk = x * per + (g >> 3);
    if ((g >> 3) >= per || chunk >= nchunks) return;          // (grid = 8 * per)
  }
  const int cstart = chunk * CHUNK, nch = min(CHUNK, BE - cstart);

  // ---- edge geometry: thread k < nch owns edge k of the chunk
  if (tid < nch) {
    const int be = order ? order[cstart + tid] : cstart + tid;
    const int b = be / E, e = be - b * E;
    const float* __restrict__ ce = coords + (int64_t)be * (2 * PP);
    float cv[2 * PP];
#pragma unroll
    for (int p = 0; p < 2 * PP; p++) cv[p] = ce[p];
    s_be[tid] = be; s_frame[tid] = b * n2 + (int)jj[e]; s_pi[tid] = b * Np + (int)ii[e];
    bool heavy = false;
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const float dv = LVF(l, coord_div);
      int xlo = 0x7fffffff, xhi = -0x7fffffff, ylo = 0x7fffffff, yhi = -0x7fffffff;
#pragma unroll
      for (int p = 0; p < PP; p++) {
        const float qx = cv[p] / dv, qy = cv[PP + p] / dv;           // the reference's true division (coords / s)
        const int ox = floor_to_int(qx) - R, oy = floor_to_int(qy) - R;
        s_ox[l][tid][p] = ox; s_oy[l][tid][p] = oy;
        s_dx[l][tid][p] = qx - floorf(qx); s_dy[l][tid][p] = qy - floorf(qy);
        xlo = min(xlo, ox); xhi = max(xhi, ox); ylo = min(ylo, oy); yhi = max(yhi, oy);
      }
      const int bw = xhi - xlo + D, bh = yhi - ylo + D;
      s_box[l][tid][0] = xlo; s_box[l][tid][1] = ylo; s_box[l][tid][2] = bw; s_box[l][tid][3] = bh;
      heavy = heavy || (long long)bw * bh > 16 * (l ? MAXT1 : MAXT0);
    }
    s_heavy[tid] = heavy ? 1 : 0;
  }
  if (tid == 0) { s_ctl[0] = 0; s_ctl[1] = 0; }

  const int nslab = C / SC;                                   // >= 2 (launcher)
  const int am = lane & 15, kg = lane >> 4;

  // ---- round planner (loader wave 0, lanes 0..15): the longest run of items whose union regions fit; result -> tables[par]
  auto plan_round = [&](int par) {
    const int cur = s_ctl[0], hp = s_ctl[1];
    int nit = 0;
    if (cur < nch) {
      const bool hv = s_heavy[cur] != 0;
      // candidate item of lane i: edge cur + i (all 9 pixels), or pixel hp + i of the heavy edge cur
      const int ce_ = hv ? cur : min(cur + am, nch - 1), cp = hv ? min(hp + am, PP - 1) : 0;
      bool ok = hv ? (hp + am < PP) : (cur + am < nch && s_heavy[ce_] == 0 && s_frame[ce_] == s_frame[cur]);
      ok = ok && am < SLOTS && lane < 16;
      int x0[NL], y0[NL], x1[NL], y1[NL];
#pragma unroll
      for (int l = 0; l < NL; l++) {
        int bx, by, bw, bh;
        if (hv) { bx = s_ox[l][ce_][cp]; by = s_oy[l][ce_][cp]; bw = D; bh = D; }
        else { bx = s_box[l][ce_][0]; by = s_box[l][ce_][1]; bw = s_box[l][ce_][2]; bh = s_box[l][ce_][3]; }
        x0[l] = dn_prefix_min16(bx); y0[l] = dn_prefix_min16(by);
        x1[l] = -dn_prefix_min16(-(bx + bw)); y1[l] = -dn_prefix_min16(-(by + bh));
        ok = ok && (long long)(x1[l] - x0[l]) * (y1[l] - y0[l]) <= 64 * (l ? S::R1CH : S::R0CH);
      }
      const unsigned long long m = __ballot(ok) & 0xffffull;
      nit = max((int)__builtin_ctzll(~m), 1);                 // leading run of admissible items (a single item always fits)
      if (lane < nit) { s_item[par][lane][0] = ce_; s_item[par][lane][1] = cp; s_item[par][lane][2] = hv ? 1 : PP; }
      if (lane == nit - 1) {
#pragma unroll
        for (int l = 0; l < NL; l++) { s_reg[par][l][0] = x0[l]; s_reg[par][l][1] = y0[l]; s_reg[par][l][2] = x1[l] - x0[l]; s_reg[par][l][3] = y1[l] - y0[l]; }
        int ncur = cur, nhp = hp;
        if (hv) { nhp = hp + nit; if (nhp >= PP) { nhp = 0; ncur = cur + 1; } } else ncur = cur + nit;
        s_ctl[0] = ncur; s_ctl[1] = nhp;
      }
    }
    if (lane == 0) s_nit[par] = nit;
  };

  __syncthreads();                                            // geometry complete
  if (wave == 0) plan_round(0);
  __syncthreads();                                            // round 0 planned

  if (wave < DN_NLOAD) {
    // ============================================================================================ LOADER WAVES
    // Level-0 chunk c -> loader c % NLOAD (its j-th: c = wave + NLOAD * j); level-1 and patch chunks likewise.
    const char* const zero_src = reinterpret_cast<const char*>(g_corr_zero);
    struct Round {
      int nck[NL]; unsigned ps[NL]; const char* fbase[NL];
      unsigned g0[L0PW], g1[L1PW > 0 ? L1PW : 1], pp[PPW];    // lane's byte offset in the frame / in fmap1; ~0u = zeros
    };
    auto round_setup = [&](int par) -> Round {
      Round rd;
      const int nitems = s_nit[par];
      const int frame = s_frame[s_item[par][0][0]];
      int x0[NL], y0[NL], RW[NL], NP[NL];
#pragma unroll
      for (int l = 0; l < NL; l++) {
        x0[l] = s_reg[par][l][0]; y0[l] = s_reg[par][l][1]; RW[l] = s_reg[par][l][2];
        NP[l] = RW[l] * s_reg[par][l][3];
        rd.nck[l] = (NP[l] + 63) >> 6;
        rd.ps[l] = (unsigned)rd.nck[l] * 1024u;
        rd.fbase[l] = reinterpret_cast<const char*>(static_cast<const T*>(LVF(l, fmap2)) + (int64_t)(frame / n2) * LVF(l, s_b) +
                                                    (int64_t)(frame % n2) * LVF(l, s_n));
      }
      auto src_off = [&](int l, int pos) -> unsigned {
        const float inv_rw = __builtin_amdgcn_rcpf((float)RW[l]);
        const int ry = (int)(((float)pos + 0.5f) * inv_rw), rx = pos - ry * RW[l];
        const int gy = y0[l] + ry, gx = x0[l] + rx;
        const bool in = pos < NP[l] && gy >= 0 && gy < LVF(l, H2) && gx >= 0 && gx < LVF(l, W2);
        return in ? (unsigned)(gy * (int)LVF(l, s_h) + gx * (int)LVF(l, s_w)) * ESZ : 0xffffffffu;
      };
#pragma unroll
      for (int j = 0; j < L0PW; j++) rd.g0[j] = src_off(0, (wave + DN_NLOAD * j) * 64 + lane);
      if constexpr (NL == 2) {
#pragma unroll
        for (int j = 0; j < L1PW; j++) rd.g1[j] = src_off(NL - 1, (wave + DN_NLOAD * j) * 64 + lane);
      }
#pragma unroll
      for (int j = 0; j < PPW; j++) {
        const int id = (wave + DN_NLOAD * j) * 64 + lane;
        const int it = id / PPIECES, pc = id - it * PPIECES;
        rd.pp[j] = (it < nitems) ? (unsigned)s_pi[s_item[par][it][0]] * (unsigned)(C * PP) * ESZ + (unsigned)pc * 16u : 0xffffffffu;
      }
      return rd;
    };
    auto issue_slab = [&](const Round& rd, int s, int buf) {
      const unsigned base = arena0 + (unsigned)buf * (unsigned)BUFSZ;
      const unsigned c0 = (unsigned)(s * SC);
      auto pieces = [&](int l, unsigned* so) {
        const int sh = LVF(l, cb_shift);
        const unsigned bb = (unsigned)LVF(l, block_stride) * ESZ;
        auto piece = [&](unsigned c) -> unsigned { const unsigned blk = c >> sh; return blk * bb + (c - (blk << sh)) * ESZ; };
        so[0] = piece(c0); so[1] = piece(c0 + SC / 4); so[2] = piece(c0 + SC / 2); so[3] = piece(c0 + 3 * SC / 4);
      };
      unsigned so[4];
      pieces(0, so);
#pragma unroll
      for (int j = 0; j < L0PW; j++) {
        const int ck = wave + DN_NLOAD * j;
        if (ck < rd.nck[0]) {                                 // wave-uniform
          const bool in = rd.g0[j] != 0xffffffffu;
#pragma unroll
          for (int q = 0; q < 4; q++) dn_dma16(in ? rd.fbase[0] + rd.g0[j] + so[q] : zero_src, base + (unsigned)q * rd.ps[0] + (unsigned)ck * 1024u);
        }
      }
      unsigned pbase = base + 4u * rd.ps[0];
      if constexpr (NL == 2) {
        pieces(1, so);
#pragma unroll
        for (int j = 0; j < L1PW; j++) {
          const int ck = wave + DN_NLOAD * j;
          if (ck < rd.nck[1]) {                               // wave-uniform
            const bool in = rd.g1[j] != 0xffffffffu;
#pragma unroll
            for (int q = 0; q < 4; q++) dn_dma16(in ? rd.fbase[1] + rd.g1[j] + so[q] : zero_src, pbase + (unsigned)q * rd.ps[1] + (unsigned)ck * 1024u);
          }
        }
        pbase += 4u * rd.ps[1];
      }
#pragma unroll
      for (int j = 0; j < PPW; j++) {
        const int ck = wave + DN_NLOAD * j;
        if (ck < NPCH)                                        // wave-uniform
          dn_dma16(rd.pp[j] != 0xffffffffu ? reinterpret_cast<const char*>(fmap1) + rd.pp[j] + (unsigned)s * 576u : zero_src, pbase + (unsigned)ck * 1024u);
      }
    };

    Round rd = round_setup(0);
    issue_slab(rd, 0, 0);
    int g = 0, par = 0;
    for (;;) {
      int nnext = 0;
      for (int s = 0; s < nslab; s++) {
        dn_wait_dma();                                        // this wave's part of slab g has landed
        dn_barrier();                                         // barrier g: slab g complete, slab g - 1 consumed
        if (s + 1 < nslab) issue_slab(rd, s + 1, (g + 1) & 1);
        else {
          nnext = s_nit[par ^ 1];                             // planned behind this round's first barrier
          if (nnext > 0) { rd = round_setup(par ^ 1); issue_slab(rd, 0, (g + 1) & 1); }
        }
        if (s == 0 && wave == 0) plan_round(par ^ 1);         // (visible to everybody behind the next barrier; nslab >= 2)
        g++;
      }
      if (nnext == 0) break;
      par ^= 1;
    }
    return;
  }

  // ================================================================================================ COMPUTE WAVES
  const int cw = wave - DN_NLOAD;
  float* const raw = reinterpret_cast<float*>(s_raw + cw * RAWSZ);
  int g = 0, par = 0;
  for (;;) {
    const int nitems = s_nit[par];
    if (stats && wave == DN_NLOAD && lane == 0) {             // debug (DEVO_DN_STATS): rounds, items, region positions per level
      atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], (unsigned long long)nitems);
      atomicAdd(&stats[2], (unsigned long long)(s_reg[par][0][2] * s_reg[par][0][3]));
      if (NL == 2) atomicAdd(&stats[3], (unsigned long long)(s_reg[par][NL - 1][2] * s_reg[par][NL - 1][3]));
      if (s_item[par][0][2] != PP) atomicAdd(&stats[4], 1ull);
    }
    const unsigned long long t_round = stats ? __builtin_readcyclecounter() : 0ull;
    unsigned long long t_prev = t_round;
    auto stamp = [&](int ph) {                                // debug: cycles of phase ph (first compute wave)
      if (stats && wave == DN_NLOAD && lane == 0) { const unsigned long long t = __builtin_readcyclecounter(); atomicAdd(&stats[6 + ph], t - t_prev); t_prev = t; }
    };
    // ---- this wave's items: tile addresses (byte offsets of the lane's A operand inside the level's part of a slab buffer,
    //      two per register) and the lane's patch element
    unsigned ps[NL];
    int rx0[NL], ry0[NL], RW[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
      rx0[l] = s_reg[par][l][0]; ry0[l] = s_reg[par][l][1]; RW[l] = s_reg[par][l][2];
      ps[l] = (unsigned)((RW[l] * s_reg[par][l][3] + 63) >> 6) * 1024u;
    }
    const unsigned l1base = 4u * ps[0];                       // level 1's planes follow level 0's
    const unsigned pbase = l1base + (NL == 2 ? 4u * ps[NL - 1] : 0u);
    int nt0[EPW], nt1[EPW];
    unsigned ta0[EPW][MAXT0 / 2], ta1[EPW][(MT1 + 1) / 2], baddr[EPW];
#pragma unroll
    for (int e = 0; e < EPW; e++) {
      const int slot = e * DN_NCOMP + cw;
      nt0[e] = 0; nt1[e] = 0; baddr[e] = 0;
#pragma unroll
      for (int t = 0; t < MAXT0 / 2; t++) ta0[e][t] = 0;
#pragma unroll
      for (int t = 0; t < (MT1 + 1) / 2; t++) ta1[e][t] = 0;
      if (slot < nitems) {                                    // wave-uniform
        const int ek = s_item[par][slot][0], p0 = s_item[par][slot][1], np = s_item[par][slot][2];
#pragma unroll
        for (int l = 0; l < NL; l++) {
          int bx, by, bw, bh;
          if (np == PP) { bx = s_box[l][ek][0]; by = s_box[l][ek][1]; bw = s_box[l][ek][2]; bh = s_box[l][ek][3]; }
          else { bx = s_ox[l][ek][p0]; by = s_oy[l][ek][p0]; bw = D; bh = D; }
          const int npos = bw * bh;
          const float inv_bw = __builtin_amdgcn_rcpf((float)bw);
          const int ix0 = bx - rx0[l], iy0 = by - ry0[l];
          // fp16: the lane reads piece kg (8 channels) of its position; fp32: pieces 2 (kg & 1), 2 (kg & 1) + 1 (4 + 4 channels)
          const unsigned plane = HALF ? (unsigned)kg : (unsigned)(2 * (kg & 1));
          auto addr = [&](int t) -> unsigned {
            const int s = min(16 * t + am, npos - 1);
            const int py = (int)(((float)s + 0.5f) * inv_bw), px = s - py * bw;
            return plane * ps[l] + (unsigned)(((iy0 + py) * RW[l] + ix0 + px) * 16);
          };
          if (l == 0) {
            nt0[e] = (npos + 15) >> 4;
#pragma unroll
            for (int t = 0; t < MAXT0; t++) { const unsigned a = addr(t); if (t & 1) ta0[e][t >> 1] |= a << 16; else ta0[e][t >> 1] = a; }
          } else {
            nt1[e] = (npos + 15) >> 4;
#pragma unroll
            for (int t = 0; t < MAXT1; t++) { const unsigned a = addr(t); if (t & 1) ta1[e][t >> 1] |= a << 16; else ta1[e][t >> 1] = a; }
          }
        }
        // patch slab of the item: raw [SC channels][9 pixels]; the lane wants channels 8 * (k-group) .. + 7 of ITS pixel
        const int pxl = p0 + min(am, np - 1);
        const int cgrp = HALF ? kg : (kg & 1);
        baddr[e] = pbase + (unsigned)(slot * (PPIECES * 16)) + (unsigned)((8 * cgrp * PP + pxl) * (int)ESZ);
      }
    }
    dn_f4 acc0[EPW][MAXT0], acc1[EPW][MT1];
#pragma unroll
    for (int e = 0; e < EPW; e++) {
#pragma unroll
      for (int t = 0; t < MAXT0; t++) acc0[e][t] = dn_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < MT1; t++) acc1[e][t] = dn_f4{0.f, 0.f, 0.f, 0.f};
    }
    stamp(0);                                                 // item setup

    // fp32 storage: the lane's 8 channels of a position (two 16-byte pieces) -> its half of K: hi parts (k-groups 0, 1) or lo parts
    auto split8 = [&](const unsigned char* p, unsigned pstride) -> dn_h8 {
      float x[8];
      __builtin_memcpy(&x[0], p, 16); __builtin_memcpy(&x[4], p + pstride, 16);
      dn_h8 r;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const float xs = x[c] * DN_PRESCALE;
        const _Float16 h = (_Float16)xs;
        r[c] = kg < 2 ? h : (_Float16)(xs - (float)h);
      }
      return r;
    };

    for (int s = 0; s < nslab; s++) {
      dn_barrier();                                           // barrier g: slab g has landed (the loaders waited for their DMA)
      stamp(1);
      const unsigned char* buf = s_arena + (unsigned)(g & 1) * (unsigned)BUFSZ;
      // ---- products: every item of the wave against the slab
#pragma unroll
      for (int e = 0; e < EPW; e++) {
        if (nt0[e] > 0) {                                     // wave-uniform
          dn_h8 b1, b2;
          if constexpr (HALF) {
            const _Float16* pb = reinterpret_cast<const _Float16*>(buf + baddr[e]);
#pragma unroll
            for (int i = 0; i < 8; i++) b1[i] = pb[i * PP];
            b2 = b1;
          } else {
            const float* pb = reinterpret_cast<const float*>(buf + baddr[e]);
            float xb[8];
#pragma unroll
            for (int i = 0; i < 8; i++) xb[i] = pb[i * PP];
#pragma unroll
            for (int i = 0; i < 8; i++) {
              const float xs = xb[i] * DN_PRESCALE;
              const _Float16 h = (_Float16)xs;
              b1[i] = h; b2[i] = (_Float16)(xs - (float)h);
            }
          }
          auto tiles0 = [&](auto ntc) {
            constexpr int NT = decltype(ntc)::value;
            dn_h8 a[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
              const unsigned off = (t & 1) ? (ta0[e][t >> 1] >> 16) : (ta0[e][t >> 1] & 0xffffu);
              if constexpr (HALF) a[t] = *reinterpret_cast<const dn_h8*>(buf + off);
              else a[t] = split8(buf + off, ps[0]);
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
              acc0[e][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b1, acc0[e][t], 0, 0, 0);
              if constexpr (!HALF) acc0[e][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b2, acc0[e][t], 0, 0, 0);
            }
          };
          auto tiles1 = [&](auto ntc) {
            constexpr int NT = decltype(ntc)::value;
            dn_h8 a[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
              const unsigned off = l1base + ((t & 1) ? (ta1[e][t >> 1] >> 16) : (ta1[e][t >> 1] & 0xffffu));
              if constexpr (HALF) a[t] = *reinterpret_cast<const dn_h8*>(buf + off);
              else a[t] = split8(buf + off, ps[NL - 1]);
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
              acc1[e][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b1, acc1[e][t], 0, 0, 0);
              if constexpr (!HALF) acc1[e][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b2, acc1[e][t], 0, 0, 0);
            }
          };
          const int n0 = nt0[e];
          if constexpr (MAXT0 == 10) {
            if (n0 <= 6) tiles0(std::integral_constant<int, 6>{});
            else if (n0 <= 7) tiles0(std::integral_constant<int, 7>{});
            else if (n0 <= 8) tiles0(std::integral_constant<int, 8>{});
            else tiles0(std::integral_constant<int, 10>{});
          } else {
            if (n0 <= 9) tiles0(std::integral_constant<int, 9>{});
            else if (n0 <= 13) tiles0(std::integral_constant<int, 13>{});
            else tiles0(std::integral_constant<int, MAXT0>{});
          }
          if constexpr (NL == 2) {
            const int n1 = nt1[e];
            if constexpr (MAXT1 == 6) {
              if (n1 <= 4) tiles1(std::integral_constant<int, 4>{});
              else if (n1 <= 5) tiles1(std::integral_constant<int, 5>{});
              else tiles1(std::integral_constant<int, 6>{});
            } else {
              if (n1 <= 9) tiles1(std::integral_constant<int, 9>{});
              else tiles1(std::integral_constant<int, MT1>{});
            }
          }
        }
      }
      g++;
      stamp(2);
    }
    const int nnext = s_nit[par ^ 1];                         // (planned behind this round's first barrier)
    // ---- epilogue: per item and level, accumulators -> raw[pixel][box position] (wave-private), then the fused bilinear
    //      blend + axis swap + permutation (correlation_kernel.cu:221-232).  Lane (g, p) = (lane / 9, lane % 9), lanes 0..62:
    //      output t = 63 j + lane = q * 9 + p with q = 7 j + g = cx * Dm + a.  The coarse level first (into registers).
    {
      const int ep = lane % PP, eg = lane / PP;
      const int nq = Dm * Dm;
      const bool pairs = NL == 2 && out_lstride == 2 && lv0.out_offset == 0 && lv1.out_offset == 1 && (out_estride & 1) == 0;
#pragma unroll
      for (int e = 0; e < EPW; e++) {
        if (nt0[e] > 0) {                                     // wave-uniform
          const int slot = e * DN_NCOMP + cw;
          const int ek = s_item[par][slot][0], p0 = s_item[par][slot][1], np = s_item[par][slot][2];
          const int col = ep - p0;
          const bool act = lane < 63 && col >= 0 && col < np;
          const int be = s_be[ek];
          T* op = out + (int64_t)be * out_estride;
          float o1[NQ];
#pragma unroll
          for (int j = 0; j < NQ; j++) o1[j] = 0.0f;
#pragma unroll
          for (int lx = 0; lx < NL; lx++) {
            const int l = NL - 1 - lx;
            // parameters of this lane's pixel (loaded before the dump so that their latency hides under it)
            int bx, by, bw;
            if (np == PP) { bx = s_box[l][ek][0]; by = s_box[l][ek][1]; bw = s_box[l][ek][2]; }
            else { bx = s_ox[l][ek][p0]; by = s_oy[l][ek][p0]; bw = D; }
            const float dxp = s_dx[l][ek][ep], dyp = s_dy[l][ek][ep];
            const int oxp = s_ox[l][ek][ep], oyp = s_oy[l][ek][ep];
            if (am < np) {
              float* dst = raw + am * BOXSP + 4 * kg;
              if (l == 0) {
#pragma unroll
                for (int t = 0; t < MAXT0; t++)
                  if (t < nt0[e]) { dn_f4 v = acc0[e][t]; if constexpr (!HALF) v *= DN_UNSCALE; *reinterpret_cast<dn_f4*>(dst + 16 * t) = v; }
              } else {
#pragma unroll
                for (int t = 0; t < MAXT1; t++)
                  if (t < nt1[e]) { dn_f4 v = acc1[e][t]; if constexpr (!HALF) v *= DN_UNSCALE; *reinterpret_cast<dn_f4*>(dst + 16 * t) = v; }
              }
            }
            wave_lds_fence();
            float w00, w01, w10, w11;
            {
#pragma clang fp contract(off)
              w00 = (1.0f - dxp) * (1.0f - dyp); w01 = dxp * (1.0f - dyp); w10 = (1.0f - dxp) * dyp; w11 = dxp * dyp;
            }
            // inactive lanes read (harmlessly) from pixel 0's row
            const float* rw = raw + (act ? col * BOXSP + (oyp - by) * bw + (oxp - bx) : 0);
            const int bwa = act ? bw : 0;
            float rv[NQ][4];                                  // all taps first (no branch between the LDS reads), then the blends
#pragma unroll
            for (int j = 0; j < NQ; j++) {
              const int q = min(7 * j + eg, nq - 1);
              const int cx = q / Dm, a = q - cx * Dm;
              const float* r = rw + (act ? a * bw + cx : 0);
              rv[j][0] = r[0]; rv[j][1] = r[1]; rv[j][2] = r[bwa]; rv[j][3] = r[bwa + 1];
            }
#pragma unroll
            for (int j = 0; j < NQ; j++) {
              const int q = 7 * j + eg;
              float o;
              {
#pragma clang fp contract(off)
                o = w00 * rv[j][0]; o = o + w01 * rv[j][1]; o = o + w10 * rv[j][2]; o = o + w11 * rv[j][3];
              }
              if (act && q < nq) {
                const int t = q * PP + ep;
                if (NL == 2 && l == 1) o1[j] = o;
                else if (NL == 2) {
                  if (pairs) {
                    if constexpr (HALF) {
                      dn_h2 pr = {(_Float16)o, (_Float16)o1[j]};
                      __builtin_nontemporal_store(pr, reinterpret_cast<dn_h2*>(op + 2 * t));
                    } else {
                      typedef float f2 __attribute__((ext_vector_type(2)));
                      f2 pr = {o, o1[j]};
                      __builtin_nontemporal_store(pr, reinterpret_cast<f2*>(op + 2 * t));
                    }
                  } else {
                    store_streamed(op + (int64_t)t * out_lstride + lv0.out_offset, from_f32<T>(o));
                    store_streamed(op + (int64_t)t * out_lstride + lv1.out_offset, from_f32<T>(o1[j]));
                  }
                } else store_streamed(op + (int64_t)t * out_lstride + lv0.out_offset, from_f32<T>(o));
              }
            }
            wave_lds_fence();                                 // the next level / item reuses the raw area
          }
        }
      }
    }
    stamp(3);
    if (stats && wave == DN_NLOAD && lane == 0) atomicAdd(&stats[5], __builtin_readcyclecounter() - t_round);
    if (nnext == 0) break;                                    // block-uniform
    par ^= 1;
  }
#undef LVF
}
