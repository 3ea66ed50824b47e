// altcorr backward as two products on the fp32 matrix cores (correlation_kernel.cu:139-190,252-269), channels-last fmap2, C % 128 == 0.
//
// With G_e[p] the D x D window gradient of patch pixel p of edge e (the adjoint of the four-tap blend, zero where the tap lies outside the
// frame) and o_e[p] the window's origin, the reference's scatter is two contractions:
//     d_fmap1[k][c][p]        = sum over positions x of   G_e[p](x - o_e[p]) * fmap2[j][x][c]                 (one per edge)
//     d_fmap2[j][x][c]        = sum over (e, p) with j = jj[e] of   G_e[p](x - o_e[p]) * fmap1[k][c][p]       (one per frame position)
// corr_bwd_edge_kernel does the first per edge (M = the 9 patch pixels, N = channels, K = the positions of the edge's clipped box) and
// hands G and one record per (edge, pixel) window — appended to the target frame's window list — to corr_bwd_frame_kernel, which OWNS an 8 x 8 tile of a
// frame's gradient (M = 16 positions per wave, N = channels, K = the (edge, pixel) windows that overlap the tile) and stores it once:
// no atomics on d_fmap2 (the one-kernel path issues 49 M of them per level), no memset of it, every position written exactly once.
// Both use v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation).  The channel a result column stands for is permuted so that
// lane n of the 16 columns owns channels 8 n .. 8 n + 7 across its 8 accumulators: its operand loads and its stores are 32 contiguous
// bytes, 16 lanes = the 512 bytes of a channels-last pixel.
#pragma once

namespace devo {

typedef float bw_f4 __attribute__((ext_vector_type(4)));

// fmap1 [N][C][9] -> [N][9][C] (plain fp32): the B operand of the frame kernel, 32 contiguous bytes per lane.  The same launch zeroes
// d_fmap1 (the edge kernel adds into it) and the frames' list cursors: two memset launches less per call.
__global__ __launch_bounds__(256) void corr_bwd_patch_t_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ d1,
                                                               int* __restrict__ cursors, int frames, int N, int C) {
  extern __shared__ __attribute__((aligned(16))) float bpt_lds[];
  const int n = blockIdx.x;
  if (n == 0) for (int i = threadIdx.x; i < frames; i += 256) cursors[i] = 0;
  if (n >= N) return;
  const float* in = src + (int64_t)n * C * PP;
  float* o = dst + (int64_t)n * C * PP;
  float* z = d1 + (int64_t)n * C * PP;
  for (int i = threadIdx.x; i < C * PP; i += 256) { bpt_lds[i] = in[i]; z[i] = 0.0f; }
  __syncthreads();
  for (int i = threadIdx.x; i < C * PP; i += 256) { const int p = i / C, c = i - p * C; o[i] = bpt_lds[c * PP + p]; }
}

// ---------------------------------------------------------------------------------------------------------------- per edge
// One workgroup of 4 waves per edge.  The gradient block becomes G (each wave forms a quarter; LDS + a global copy), then every wave takes
// a QUARTER of the box positions for d_fmap1 on the matrix cores: a 107-position box is 27 K steps = 7 per wave = ONE request batch, so a
// wave's chain of dependent global round trips is indices / coordinates / gradient block -> features -> done (each costs 7-8 us while the
// whole launch is resident: with one wave per edge and four batches the kernel took 59 us, profiles/r03_corr_backward.txt).  The four
// partial d_fmap1 are added in LDS, in the tensor's [c][p] order, and leave as ONE set of coalesced atomics.
constexpr int BWE_KS = 4;                       // waves per edge = parts of the K range
constexpr int BWD_NB = 8;                       // K steps requested together
struct BwdPair { int gs_off, ox, oy, row; };    // one (edge, patch pixel) window for the frame kernel: offset of its G, origin, row of fmap1_t
template <int RMAX>
__global__ __launch_bounds__(64 * BWE_KS) __attribute__((amdgpu_waves_per_eu(1, 4))) void corr_bwd_edge_kernel(
    const float* __restrict__ fmap2, const float* __restrict__ coords, const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
    const float* __restrict__ grad, float* __restrict__ d1, long long BE, int E, int Np, int n2, int C, int H2, int W2, int64_t s_b,
    int64_t s_n, int R, float* __restrict__ gs, BwdPair* __restrict__ pairs, int* __restrict__ cursors, int cap) {
  constexpr int DMX = 2 * RMAX + 2;
  __shared__ float s_g[PP * DMX * DMX];
  __shared__ float s_grad[(DMX - 1) * (DMX - 1) * PP];
  __shared__ float s_frac[2][PP];                                      // blend weights dx, dy per patch pixel
  __shared__ int s_org[2][PP];                                        // window origins ox, oy
  __shared__ __attribute__((aligned(16))) float s_out[BWE_KS][128 * PP];   // the waves' partial d_fmap1 of a channel half, [c][p] order
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const long long be = blockIdx.x;
  const int D = 2 * R + 2, Dm = D - 1, DD = D * D;
  const int b = (int)(be / E), e = (int)(be % E);
  const int64_t pi = ii[e], fj = jj[e];
  float cx = 0.0f, cy = 0.0f;
  if (ln < PP) { cx = coords[(be * 2 + 0) * PP + ln]; cy = coords[(be * 2 + 1) * PP + ln]; }      // (every wave: the box without a barrier)
  {
    const float* g = grad + be * Dm * Dm * PP;
    const int ng = Dm * Dm * PP;
    for (int i = tid; i < ng; i += 64 * BWE_KS) s_grad[i] = g[i];
  }
  const int my_ox = floor_to_int(cx) - R, my_oy = floor_to_int(cy) - R;
  if (tid < PP) { s_org[0][tid] = my_ox; s_org[1][tid] = my_oy; s_frac[0][tid] = cx - floorf(cx); s_frac[1][tid] = cy - floorf(cy); }
  int xmin = 1 << 30, xmax = -(1 << 30), ymin = 1 << 30, ymax = -(1 << 30);
#pragma unroll
  for (int p = 0; p < PP; p++) {
    const int ox = __builtin_amdgcn_readlane(my_ox, p), oy = __builtin_amdgcn_readlane(my_oy, p);
    xmin = min(xmin, ox); xmax = max(xmax, ox); ymin = min(ymin, oy); ymax = max(ymax, oy);
  }
  const int x0 = max(xmin, 0), x1 = min(xmax + D, W2), y0 = max(ymin, 0), y1 = min(ymax + D, H2);
  const int bw = max(x1 - x0, 0), npos = bw * max(y1 - y0, 0);
  if (npos == 0) return;                                              // (uniform over the workgroup: nothing of this edge is inside the frame)
  const int frame = b * n2 + (int)fj;
  int slot = 0;                                                       // the edge's slot in its frame's window list (order: whoever comes first)
  if (tid == 0) slot = atomicAdd(&cursors[frame], 1);
  // this wave's quarter of the K steps: requested now, before the window gradients exist (it only needs the box)
  const int mm = ln & 15, kq = ln >> 4;
  const float* __restrict__ f2 = fmap2 + (int64_t)b * s_b + fj * s_n + 8 * mm;          // this lane's 8 channels (+ 128 per channel half)
  const float inv_bw = __builtin_amdgcn_rcpf((float)bw);
  const int nstep = (npos + 3) >> 2, per = (nstep + BWE_KS - 1) / BWE_KS, sa = wv * per, sb = min(nstep, sa + per);
  bw_f4 lo[BWD_NB], hi[BWD_NB];
  int pos_a[BWD_NB];                                                   // (gy - y0) << 16 | (gx - x0), -1: beyond this wave's part
  auto request = [&](int s0, int ch) {
#pragma unroll
    for (int u = 0; u < BWD_NB; u++) {
      const int q = 4 * (s0 + u) + kq;
      const bool ok = s0 + u < sb && q < npos;
      const int qq = ok ? q : 0;
      const int ry = (int)(((float)qq + 0.5f) * inv_bw), rx = qq - ry * bw;
      const float* src = f2 + ((int64_t)(y0 + ry) * W2 + (x0 + rx)) * C + ch;
      lo[u] = *reinterpret_cast<const bw_f4*>(src);
      hi[u] = *reinterpret_cast<const bw_f4*>(src + 4);
      pos_a[u] = ok ? (ry << 16 | rx) : -1;
    }
  };
  request(sa, 0);
  __syncthreads();                                                    // gradient block, origins, blend weights
  // window gradients (correlation_kernel.cu:259-269): the adjoint of the blend; taps outside the frame carry nothing (:182)
  const float inv_dd = __builtin_amdgcn_rcpf((float)DD), inv_d = __builtin_amdgcn_rcpf((float)D);
  for (int o = tid; o < PP * DD; o += 64 * BWE_KS) {
    const int p = (int)(((float)o + 0.5f) * inv_dd), r_ = o - p * DD;
    const int a = (int)(((float)r_ + 0.5f) * inv_d), c = r_ - a * D;
    const float dx = s_frac[0][p], dy = s_frac[1][p];
    const int pox = s_org[0][p], poy = s_org[1][p];
    auto G = [&](int aa, int cc) -> float { return (aa >= 0 && aa < Dm && cc >= 0 && cc < Dm) ? s_grad[(cc * Dm + aa) * PP + p] : 0.0f; };
    float s = 0.0f;
    s += (1.0f - dx) * (1.0f - dy) * G(a, c);
    s += dx * (1.0f - dy) * G(a, c - 1);
    s += (1.0f - dx) * dy * G(a - 1, c);
    s += dx * dy * G(a - 1, c - 1);
    const int gy = poy + a, gx = pox + c;
    if (!(gy >= 0 && gy < H2 && gx >= 0 && gx < W2)) s = 0.0f;
    s_g[o] = s;
    gs[be * (PP * DD) + o] = s;                                       // for the frame kernel
  }
  __syncthreads();
  // d_fmap1: rows = the 9 patch pixels (lane m = l % 16; rows 9..15 stay zero), columns = channels, K = 4 box positions per step.
  const int mp = min(mm, PP - 1);
  const int dox = x0 - s_org[0][mp], doy = y0 - s_org[1][mp];          // box corner relative to this row's window origin
  const float* g_mine = s_g + mp * DD;
  float* g1 = d1 + ((int64_t)b * Np + pi) * C * PP;
  for (int ch = 0; ch < C; ch += 128) {
    bw_f4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = bw_f4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int s0 = sa; s0 < sb; s0 += BWD_NB) {
      if (s0 > sa || ch > 0) request(s0, ch);
      float a[BWD_NB];
#pragma unroll
      for (int u = 0; u < BWD_NB; u++) {
        const int aa = (pos_a[u] >> 16) + doy, cc = (pos_a[u] & 0xffff) + dox;
        const bool in = pos_a[u] >= 0 && mm < PP && aa >= 0 && aa < D && cc >= 0 && cc < D;
        const float av = g_mine[in ? aa * D + cc : 0];
        a[u] = in ? av : 0.0f;                                        // (a zero A row / column makes the clamped loads harmless)
      }
      __builtin_amdgcn_sched_barrier(0);                              // all requests first (the scheduler would pair each with its products)
#pragma unroll
      for (int u = 0; u < BWD_NB; u++) {
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], lo[u][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], hi[u][j], acc[4 + j], 0, 0, 0);
      }
    }
    // lane (n = mm, kq) holds rows 4 kq + i = patch pixel p, columns = channels ch + 8 n + j.  Into LDS in the tensor's [c][p] order, the four
    // waves' partials added there: consecutive threads then add to consecutive addresses (scattered float atomics — one cache line per
    // lane — made the first version 5x slower than everything else in it).
    if (ch > 0) __syncthreads();                                      // (the previous half's sums have been read)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int p = 4 * kq + i;
      if (p < PP) {
#pragma unroll
        for (int j = 0; j < 8; j++) s_out[wv][(8 * mm + j) * PP + p] = acc[j][i];
      }
    }
    __syncthreads();
    for (int t = tid; t < 128 * PP; t += 64 * BWE_KS) {
      float v = s_out[0][t];
#pragma unroll
      for (int w = 1; w < BWE_KS; w++) v += s_out[w][t];
      atomicAdd(g1 + ch * PP + t, v);
    }
  }
  if (wv == 0) {
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (ln < PP) pairs[((int64_t)frame * cap + slot) * PP + ln] = BwdPair{(int)(be * (PP * DD)) + ln * DD, my_ox, my_oy, (b * Np + (int)pi) * PP + ln};
  }
}

// ---------------------------------------------------------------------------------------------------------------- per frame tile
// One workgroup (4 waves) owns an 8 x 8 tile of one frame (x 128 channels per blockIdx.x): wave w the rows 2 w, 2 w + 1.  Every wave scans
// a quarter of the frame's (edge, pixel) windows — all its loads in flight at once; those that overlap the tile go into one LDS list in
// list order (count, barrier, write); every wave multiplies the list, BWF_NB K steps requested together.
constexpr int BWF_SCAN = 9;                     // windows a lane examines per pass: 4 x 64 x 9 = 2304 per workgroup and pass
constexpr int BWF_NB = 8;
constexpr int BWF_LIST = 512;                   // windows in the LDS list (8 KB: 16 workgroups per CU)
// WV waves per workgroup = a tile of 8 x 2 WV positions: 4 (8 x 8) where a level has thousands of tiles, 1 (8 x 2) where it has few (DEVO's
// level 1: 30 x 40 — 300 tiles of 8 x 8 cannot fill the chip and each keeps several hundred windows; 8 x 2 tiles keep 40 % fewer each).
template <int WV>
#ifndef BWF_WAVES
#define BWF_WAVES 4, 4                          // (64 VGPRs: 58 instead of 64 us at level 0 in the same run; 6, 6 and 8, 8 spill)
#endif
__global__ __launch_bounds__(64 * WV) __attribute__((amdgpu_waves_per_eu(BWF_WAVES))) void corr_bwd_frame_kernel(
    const float* __restrict__ f1t, const float* __restrict__ gs, const BwdPair* __restrict__ pairs, const int* __restrict__ cursors,
    float* __restrict__ d2, int n2, int C, int H2, int W2, int64_t s_b, int64_t s_n, int D, int cap, int tiles_x) {
  __shared__ BwdPair s_ent[BWF_LIST];
  __shared__ int s_cnt[2][WV];                                        // by pass parity: a pass that keeps nothing has no barrier after its reads
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int ch = (blockIdx.x / tiles_x) * 128, tx = (blockIdx.x % tiles_x) * 8, ty = blockIdx.y * (2 * WV), frame = blockIdx.z;
  const int npair = min(cursors[frame], cap) * PP;
  const BwdPair* __restrict__ fp = pairs + (int64_t)frame * cap * PP;
  const int mm = ln & 15, kq = ln >> 4;
  const int px = tx + (mm & 7), py = ty + 2 * wv + (mm >> 3);        // this lane's position as an A-operand row
  bw_f4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; j++) acc[j] = bw_f4{0.0f, 0.0f, 0.0f, 0.0f};
  const float* fb = f1t + ch + 8 * mm;
  int par = 0;
  for (int base = 0; base < npair; base += 64 * WV * BWF_SCAN, par ^= 1) {
    // ---- scan: wave w takes windows base + w * 64 * SCAN ..; the kept ones go into ONE list in list order (count, barrier, write)
    BwdPair e[BWF_SCAN];
    const int q0 = base + wv * 64 * BWF_SCAN + ln;
#pragma unroll
    for (int u = 0; u < BWF_SCAN; u++) e[u] = fp[min(q0 + 64 * u, npair - 1)];
    unsigned long long bal[BWF_SCAN];
    int n_kept = 0;
#pragma unroll
    for (int u = 0; u < BWF_SCAN; u++) {
      const bool hit = q0 + 64 * u < npair && e[u].ox <= tx + 7 && e[u].ox + D > tx && e[u].oy <= ty + 2 * WV - 1 && e[u].oy + D > ty;
      bal[u] = __ballot(hit);
      n_kept += __popcll(bal[u]);
    }
    if (ln == 0) s_cnt[par][wv] = n_kept;                             // (pass p + 2 rewrites this half behind pass p + 1's barrier: every wave has read it)
    __syncthreads();
    int first = 0, total = 0;
#pragma unroll
    for (int w = 0; w < WV; w++) { const int c = s_cnt[par][w]; if (w < wv) first += c; total += c; }
    // the list holds BWF_LIST windows (a level-0 tile keeps ~25 of a pass, a level-1 tile several hundred): chunk by chunk
    for (int c0 = 0; c0 < total; c0 += BWF_LIST) {
      int off = first - c0;
#pragma unroll
      for (int u = 0; u < BWF_SCAN; u++) {
        const int slot = off + __popcll(bal[u] & ((1ull << ln) - 1ull));
        if (((bal[u] >> ln) & 1ull) && slot >= 0 && slot < BWF_LIST) s_ent[slot] = e[u];
        off += __popcll(bal[u]);
      }
      __syncthreads();
      const int cnt = min(total - c0, BWF_LIST);
      // ---- K = the kept windows, 4 per step, BWF_NB steps requested together
      const int nstep = (cnt + 3) >> 2;
      for (int s0 = 0; s0 < nstep; s0 += BWF_NB) {
        bw_f4 lo[BWF_NB], hi[BWF_NB];
        float a[BWF_NB];
#pragma unroll
        for (int u = 0; u < BWF_NB; u++) {
          const int k = 4 * (s0 + u) + kq;
          const bool ok = k < cnt;
          const BwdPair en = s_ent[ok ? k : 0];
          const float* src = fb + (int64_t)en.row * C;
          lo[u] = *reinterpret_cast<const bw_f4*>(src);
          hi[u] = *reinterpret_cast<const bw_f4*>(src + 4);
          const int aa = py - en.oy, cc = px - en.ox;
          const bool in = ok && aa >= 0 && aa < D && cc >= 0 && cc < D;
          const float av = gs[in ? en.gs_off + aa * D + cc : 0];
          a[u] = in ? av : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < BWF_NB; u++) {
          if (4 * (s0 + u) >= cnt) break;                             // (uniform: the tail of the last batch)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], lo[u][j], acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; j++) acc[4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], hi[u][j], acc[4 + j], 0, 0, 0);
        }
      }
      __syncthreads();                                                // the list is rewritten by the next chunk / pass
    }
  }
  // lane (n = mm, kq) holds rows 4 kq + i = tile positions, columns = channels ch + 8 n + j: 32 contiguous bytes per position
  float* out = d2 + (int64_t)(frame / n2) * s_b + (int64_t)(frame % n2) * s_n + ch + 8 * mm;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int m = 4 * kq + i;
    const int ox_ = tx + (m & 7), oy_ = ty + 2 * wv + (m >> 3);
    if (ox_ < W2 && oy_ < H2) {
      float* o = out + ((int64_t)oy_ * W2 + ox_) * C;
      *reinterpret_cast<bw_f4*>(o) = bw_f4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
      *reinterpret_cast<bw_f4*>(o + 4) = bw_f4{acc[4][i], acc[5][i], acc[6][i], acc[7][i]};
    }
  }
}

}  // namespace devo
