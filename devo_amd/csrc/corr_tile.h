// Tile geometry of the staged lookup kernel and the plan's bin of an edge — shared by corr.hip (lookup + plan kernels)
// and ba.hip (the fused reprojection can emit the plan's bins while it still holds the coordinates).
#pragma once
#include <hip/hip_runtime.h>

#ifndef DEVO_PLAN_BAND
#define DEVO_PLAN_BAND 16                        // rows per band of the locality plan
#endif

namespace devo {

constexpr int CORR_KC = 8;                       // channels staged per LDS chunk
constexpr int CORR_ROWPAD = CORR_KC + 4;         // LDS row stride of a staged position in floats
// Tile capacity of the staged kernel (NG = 1: r <= 3, NG = 3: r <= 5) — shared with the locality plan, whose HEAVY
// class must be exactly the set of edges whose union box does not fit.
constexpr int SPP = CORR_ROWPAD / 4;                 // 16-byte slots per staged position (odd: 3 for KC = 8)
static_assert(CORR_ROWPAD % 4 == 0 && (SPP & 1) == 1, "the conflict-free pitch needs an odd number of slots per position");
__host__ __device__ constexpr int tile_positions(int ng) { return ng == 1 ? 160 : 256; }
__host__ __device__ constexpr int tile_slots(int ng) { return ng == 1 ? 528 : 960; }
// Box row pitch in 16-byte slots: the smallest value >= SPP*w that is = 8 (mod 16).  With an odd SPP this makes the
// tap-centric ds_read_b128 pattern (lane groups {0-3,12-15,20-27}, ... = 4 window rows x 4 taps) bank-conflict free
// for every box width.
__host__ __device__ __forceinline__ int tile_pitch(int w) { const int s = SPP * w; return s + ((8 - s) & 15); }
__host__ __device__ __forceinline__ bool tile_fits(int w, int h, int ng) {
  return (long long)w * h <= tile_positions(ng) && (long long)h * tile_pitch(w) <= tile_slots(ng);
}


#ifndef DEVO_PLAN_BLOCKS
#define DEVO_PLAN_BLOCKS 1                          // 0: number the bins band by band (A/B builds)
#endif
constexpr int CORR_PLAN_BB = DEVO_PLAN_BLOCKS ? 4 : 1;   // bands per block of the plan's bin numbering (4 x 16 rows)
__host__ __device__ __forceinline__ int corr_plan_bx(int xw) {   // column bins per block (~64 px)
  const int b = DEVO_PLAN_BLOCKS ? 64 / (xw > 0 ? xw : 64) : 1;
  return b < 1 ? 1 : b;
}

__device__ __forceinline__ int corr_floor_to_int(float v) {
  // static_cast<int>(floor(v)) (correlation_kernel.cu:118-119), made safe for non-finite / huge inputs
  float f = floorf(v);
  f = fminf(fmaxf(f, -1.0e6f), 1.0e6f);
  return (f == f) ? (int)f : -1000000;
}

// GROUP PLAN (the lookup's group form, corr_mm.h NW > 1: level 1 of a two-level lookup read from LDS).  A GROUP = the edges of one
// target frame whose patch CENTRE falls into one tile of CORR_GRP_T x CORR_GRP_T level-1 cells; its REGION = the level-1 positions any
// compact patch of the group can touch: the centre cell c gives window origins in [c - R - 1, c - R] (the 9 pixels lie within one cell
// of the centre) and windows of D = 2 R + 2 positions, so rows [t0 - R - 1, t0 + T + R + 2) = T + 2 R + 3 of them, columns alike.
// bin = group; HEAVY = a box that breaks the promise (level 1 outside the region, level 0 beyond 128 positions); DEAD = outside the
// frame at both levels.  The ordering step also stores every bin's first slot (corr_plan.h): workgroups find their group's edges there.
constexpr int CORR_GRP_T = 6;
constexpr int CORR_PLAN_TAIL = 4104;                          // ints behind the 2 BE + 2 of an edge plan: bin starts (<= CORR_ORDER_MAXBINS + 1), padding
__host__ __device__ __forceinline__ int corr_grp_count(int cells) { return (cells + CORR_GRP_T - 1) / CORR_GRP_T; }
// bins of a group plan: one per (batch, frame, group) + the DEAD class; 0 = more than the ordering step's counters hold
inline long long corr_grp_nbins(long long B, int n2, int H2, int W2, int l1) {
  if (l1 < 2 || H2 / l1 < 1 || W2 / l1 < 1) return 0;
  const long long nb = B * n2 * corr_grp_count(H2 / l1) * corr_grp_count(W2 / l1) + 1;
  return nb <= 4096 ? nb : 0;
}

// Plan bin of an edge from its 9 window origins: -1 = HEAVY (the union box does not fit the tile), else
// (batch, target frame, 16-row band of the patch centre, column bin of the patch centre) — consecutive edges of the sorted
// plan land next to each other in the image.
// x[p], y[p]: integer pixel of patch pixel p at the plan's level.  `geom` = corr_plan_pack(): bands | column bins << 8 |
// column-bin width << 16.
//
// Group mode (l1 >= 2: the lookup has a second level at 1 / l1 of the plan level's resolution, W2 = the plan level's width): the GROUP
// PLAN above —
//   DEAD  (returns dead_bin, the bin behind all others): the union box lies outside the frame at BOTH levels: all outputs are 0;
//   HEAVY (-1): more than `heavy_cells` level-0 box positions, or a level-1 box that leaves its group's region.
__device__ __forceinline__ int corr_plan_bin(const int* x, const int* y, float centre_x, float centre_y, int b, int frame, int n2,
                                             int H2, int geom, int D, int ng, int W2 = 0, int l1 = 0, int heavy_cells = 0,
                                             int dead_bin = -1) {
  const int nb = geom & 0xff, nxb = (geom >> 8) & 0xff, xw = geom >> 16;
  int xlo = x[0], xhi = x[0], ylo = y[0], yhi = y[0];
#pragma unroll
  for (int p = 1; p < 9; p++) { xlo = min(xlo, x[p]); xhi = max(xhi, x[p]); ylo = min(ylo, y[p]); yhi = max(yhi, y[p]); }
  if (l1 >= 2) {
    const int R = (D - 2) / 2;
    auto fdiv = [](int v, int d) -> int { return v >= 0 ? v / d : -((-v + d - 1) / d); };      // floor(v / d)
    // window origins: floor(coordinate) - R at level 0, floor(coordinate / l1) - R = floor(floor(coordinate) / l1) - R at level 1
    const int x0 = xlo - R, y0 = ylo - R, w0 = xhi - xlo + D, h0 = yhi - ylo + D;
    const int x1 = fdiv(xlo, l1) - R, y1 = fdiv(ylo, l1) - R, w1 = fdiv(xhi, l1) - fdiv(xlo, l1) + D, h1 = fdiv(yhi, l1) - fdiv(ylo, l1) + D;
    const int H1 = H2 / l1, W1 = W2 / l1;
    const bool live0 = x0 < W2 && y0 < H2 && x0 + w0 > 0 && y0 + h0 > 0;
    const bool live1 = x1 < W1 && y1 < H1 && x1 + w1 > 0 && y1 + h1 > 0;
    if (!live0 && !live1) return dead_bin;
    // the group of the patch centre (clamped into the frame) and its region
    const int cy1 = min(max(fdiv(corr_floor_to_int(centre_y), l1), 0), H1 - 1), cx1 = min(max(fdiv(corr_floor_to_int(centre_x), l1), 0), W1 - 1);
    const int gy = cy1 / CORR_GRP_T, gx = cx1 / CORR_GRP_T, ngy = corr_grp_count(H1), ngx = corr_grp_count(W1);
    const int ry0 = gy * CORR_GRP_T - R - 1, rx0 = gx * CORR_GRP_T - R - 1, rs = CORR_GRP_T + 2 * R + 3;
    if (live0 && (long long)w0 * h0 > heavy_cells) return -1;
    if (live1 && !(x1 >= rx0 && y1 >= ry0 && x1 + w1 <= rx0 + rs && y1 + h1 <= ry0 + rs)) return -1;
    const int f = min(max(frame, 0), n2 - 1);
    return ((b * n2 + f) * ngy + gy) * ngx + gx;
  } else
  // HEAVY = clearly more passes of the matrix-core kernel than a compact patch needs at this radius — more than two 64-position
  // passes for r <= 3, more than four for r <= 5 (a compact r = 5 box is 14 x 14 = 196 positions: with the r <= 3 threshold EVERY
  // edge of BASELINE's stress configuration was HEAVY, i.e. unsorted, and its lookup ran at half speed) — which includes every
  // box the staged kernel's tile cannot hold: the long items start first
  if ((long long)(xhi - xlo + D) * (yhi - ylo + D) > (ng == 1 ? 128 : 256) || !tile_fits(xhi - xlo + D, yhi - ylo + D, ng)) return -1;
  int band = (int)(fminf(fmaxf(centre_y, 0.0f), (float)(H2 - 1))) / DEVO_PLAN_BAND;
  band = min(max(band, 0), nb - 1);
  int xb = (int)(fminf(fmaxf(centre_x, 0.0f), 1.0e6f)) / xw;
  xb = min(max(xb, 0), nxb - 1);
  const int f = min(max(frame, 0), n2 - 1);
  // Bins are numbered block by block (CORR_PLAN_BB bands x enough column bins for ~64 px), so that consecutive plan slots — the
  // edges one XCD works on at the same time — cover a compact 2-D tile of the frame instead of a strip as wide as the frame: the
  // same number of edges then touches fewer distinct rows + margins of the pyramid (stress configuration: 2.8 MB instead of 4.4 MB
  // for 512 edges, against 4 MB of L2 per XCD).
  // (pyramid mode: plain band-major numbering — the region kernel wants long runs of image neighbours along a band)
  const int bx = corr_plan_bx(xw), nbx = (nxb + bx - 1) / bx, nbb = (nb + CORR_PLAN_BB - 1) / CORR_PLAN_BB;
  const int in_frame = l1 >= 2 ? band * nxb + xb
                               : ((band / CORR_PLAN_BB) * nbx + xb / bx) * (CORR_PLAN_BB * bx) + (band % CORR_PLAN_BB) * bx + (xb % bx);
  return (b * n2 + f) * (nbb * nbx * CORR_PLAN_BB * bx) + in_frame;
}

// Group plans: tiles (16 box positions each) a level-0 box may have before the edge counts as HEAVY
__host__ __device__ constexpr int corr_region_tmax(int radius) { return radius <= 3 ? 8 : 16; }
struct CorrPlanMode { int W2, l1, heavy_cells, dead_bin; };     // l1 < 2: single-level plan (legacy classes)

// Bins of the plan: row bands of 16 rows per frame (coarser if there are many frames: the counting sort keeps one LDS
// counter per bin) x column bins of >= 8 px (the frame width is not part of the plan's interface: columns beyond
// nxb * xw share the last bin; widths up to 2 * H2 are covered).  nb == 0 = too many frames.
constexpr int CORR_ORDER_MAXBINS = 4096;        // (one LDS counter per bin next to the ordering kernel's staging buffer)
struct CorrPlanGeom { int nb, nxb, xw; };
inline long long corr_plan_bins_per_frame(const CorrPlanGeom& g) {
  const int bx = corr_plan_bx(g.xw), nbx = (g.nxb + bx - 1) / bx, nbb = (g.nb + CORR_PLAN_BB - 1) / CORR_PLAN_BB;
  return (long long)nbb * nbx * CORR_PLAN_BB * bx;         // (whole blocks: a little more than nb * nxb)
}
inline CorrPlanGeom corr_plan_geom(long long B, int n2, int H2) {
  CorrPlanGeom g{(H2 + DEVO_PLAN_BAND - 1) / DEVO_PLAN_BAND, 1, 8};
  while (B * n2 * ((g.nb + CORR_PLAN_BB - 1) / CORR_PLAN_BB * CORR_PLAN_BB) > CORR_ORDER_MAXBINS && g.nb > 1) g.nb = (g.nb + 1) / 2;
  if (B * n2 * ((g.nb + CORR_PLAN_BB - 1) / CORR_PLAN_BB * CORR_PLAN_BB) > CORR_ORDER_MAXBINS || g.nb > 255) { g.nb = 0; return g; }
  const int want = (2 * H2 + 7) / 8;                      // 8-px columns across a 2:1 frame
  for (int nx = want > 255 ? 255 : want; nx >= 1; nx--) {  // the most column bins whose block-padded count still fits
    g.nxb = nx;
    int xw = (2 * H2 + nx - 1) / nx;
    xw = (xw + 3) / 4 * 4;
    g.xw = xw < 8 ? 8 : (xw > 32764 ? 32764 : xw);
    if (B * n2 * corr_plan_bins_per_frame(g) <= CORR_ORDER_MAXBINS) break;
  }
  return g;
}
inline int corr_plan_pack(const CorrPlanGeom& g) { return g.nb | (g.nxb << 8) | (g.xw << 16); }
// all bins of a plan: the (frame, band, column) bins and, behind them, the DEAD class of the pyramid mode
inline long long corr_plan_nbins(long long B, int n2, const CorrPlanGeom& g) { return B * n2 * corr_plan_bins_per_frame(g) + 1; }

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (no LDS round trips like ds_bpermute shuffles):
// Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8; lanes without a source add 0), then the last lane of row 0 / 2
// into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31).  All 64 lanes must be active.
__device__ __forceinline__ int wave_inclusive_sum(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return x;
}

}  // namespace devo
