// fastba for gfx950: sparse Gauss-Newton bundle adjustment (inference variant), reprojection, transform
// and the temporal-neighbour graph helper.  Replaces devo/fastba/ba_cuda.cu + ba.cpp (module cuda_ba,
// ba.cpp:152-157) and fuses devo/projective_ops.py:53-105.
//
// Design.  The reference accumulates the normal equations with 340 global float
// atomics per edge onto a few thousand addresses, then runs ~20 ATen launches + cuSOLVER (host sync) per
// Gauss-Newton iteration.  Here the edge list is grouped by patch once per call (integer counting sort,
// bit-exact `unique`), and each iteration is four launches with no host synchronisation:
//   ba_accumulate : one WAVE per patch.  Lanes = the patch's edges.  Per-patch quantities (C, u, the
//                   source-frame E block, B_ii, v_i) are reduced with wavefront shuffles; the patch's
//                   column of E is completed in LDS, the Schur update  S -= Q e e^T, y -= Q u e  is
//                   applied patch-by-patch into an LDS-resident lower-triangular S (dense E is never
//                   materialised); one partial S per workgroup goes to HBM.
//   ba_reduce     : sums the partials in a fixed order, mirrors the triangle, applies the damping.
//   ba_solve      : one workgroup, LDS-resident blocked (6x6) Cholesky with the right-hand side carried as
//                   an extra row, blocked back-substitution.
//   ba_retract    : pose retraction Exp(dX) * G and per-patch depth update  dz = Q (u - e^T dX).
#include "common.h"
#include "se3_dev.h"

namespace devo {

constexpr int BA_MAXN = 32;          // optimised poses per call (6N <= 192 rows keeps S in LDS)
constexpr int ACC_WAVES = 8;         // waves per ba_accumulate workgroup
constexpr int ACC_THREADS = ACC_WAVES * 64;
constexpr int ACC_MAX_WG = 64;       // partial systems written per iteration

struct BaMeta { int n_seg; int fail; int pad[2]; };

// ------------------------------------------------------------------------------------------------- utilities
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ unsigned wave_or(unsigned v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v |= (unsigned)__shfl_xor((int)v, off);
  return v;
}
__device__ __forceinline__ void lds_add(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS traffic of ONE wave is processed in order; this only stops the compiler from moving accesses across.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// In-place exclusive scan of data[0..n) by ONE workgroup of 1024 threads; data[n] = total; *total_out too.
__global__ __launch_bounds__(1024) void k_excl_scan(int* data, int n, int* total_out) {
  __shared__ int s_part[1024];
  const int t = threadIdx.x;
  const int chunk = (n + 1023) / 1024;
  const int lo = min(n, t * chunk), hi = min(n, lo + chunk);
  int s = 0;
  for (int i = lo; i < hi; i++) s += data[i];
  s_part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (t >= off) ? s_part[t - off] : 0;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  int run = s_part[t] - s;                      // exclusive prefix of this thread's chunk
  for (int i = lo; i < hi; i++) { int v = data[i]; data[i] = run; run += v; }
  if (t == 1023) { data[n] = s_part[1023]; if (total_out) *total_out = s_part[1023]; }
}

__global__ void k_flag_ids(const int64_t* __restrict__ kk, int E, int Np, int* flags) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += blockDim.x * gridDim.x) {
    int64_t k = kk[e];
    if (k >= 0 && k < Np) flags[k] = 1;
  }
}
// rank[] = exclusive scan of flags (flags[] consumed).  ku = rank of the edge's patch, kx = sorted unique ids.
__global__ void k_rank_edges(const int64_t* __restrict__ kk, int E, int Np, const int* __restrict__ rank,
                             int* ku, int* kx, int* counts) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = blockDim.x * gridDim.x;
  for (int e = gid; e < E; e += gsz) {
    int64_t k = kk[e];
    int r = (k >= 0 && k < Np) ? rank[k] : 0;
    ku[e] = r;
    atomicAdd(&counts[r], 1);
  }
  for (int p = gid; p < Np; p += gsz)
    if (rank[p + 1] != rank[p]) kx[rank[p]] = p;
}
__global__ void k_scatter_edges(const int* __restrict__ ku, int E, const int* __restrict__ seg_start, int* cursor, int* perm) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += blockDim.x * gridDim.x) {
    int s = ku[e];
    perm[seg_start[s] + atomicAdd(&cursor[s], 1)] = e;
  }
}
// Restore a deterministic (ascending edge id) order inside every segment: rank sort, one wave per segment.
__global__ void k_sort_segments(const int* __restrict__ seg_start, const int* __restrict__ n_seg_p, const int* __restrict__ in, int* out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (blockDim.x * gridDim.x) >> 6;
  const int n_seg = *n_seg_p;
  for (int s = wave; s < n_seg; s += nwaves) {
    const int a = seg_start[s], m = seg_start[s + 1] - a;
    for (int i = lane; i < m; i += 64) {
      int x = in[a + i], r = 0;
      for (int j = 0; j < m; j++) r += (in[a + j] < x);
      out[a + r] = x;
    }
  }
}

// ------------------------------------------------------------------------------------------------- per-edge maths
struct EdgeTerms {
  float r[2], w[2], Jz[2];
  float Ji[2][6], Jj[2][6];
};

// ba_cuda.cu:239-330 for one edge (fx,fy,cx,cy of intrinsics row 0).
__device__ __forceinline__ void edge_terms(const float* __restrict__ poses, const float* __restrict__ patches, int P,
                                           float fx, float fy, float cx, float cy, const float* __restrict__ target,
                                           const float* __restrict__ weight, int ix, int jx, int kx, int e, EdgeTerms& T) {
  const float* pi = poses + (int64_t)ix * 7;
  const float* pj = poses + (int64_t)jx * 7;
  float ti[3] = {pi[0], pi[1], pi[2]}, qi[4] = {pi[3], pi[4], pi[5], pi[6]};
  float tj[3] = {pj[0], pj[1], pj[2]}, qj[4] = {pj[3], pj[4], pj[5], pj[6]};
  const int PPx = P * P, ctr = (P / 2) * P + (P / 2);     // centre pixel [1][1] (ba_cuda.cu:254-257)
  const float* pk = patches + (int64_t)kx * 3 * PPx;
  float Xi[4] = {(pk[ctr] - cx) / fx, (pk[PPx + ctr] - cy) / fy, 1.0f, pk[2 * PPx + ctr]};
  float tij[3], qij[4], Xj[4];
  fb_relSE3(ti, qi, tj, qj, tij, qij);
  fb_actSE3(tij, qij, Xi, Xj);
  const float X = Xj[0], Y = Xj[1], Z = Xj[2], W = Xj[3];
  const float d = (Z >= 0.2f) ? 1.0f / Z : 0.0f;
  const float d2 = d * d;
  const float x1 = fx * (X / Z) + cx, y1 = fy * (Y / Z) + cy;
  const float rx = target[(int64_t)e * 2] - x1, ry = target[(int64_t)e * 2 + 1] - y1;
  const bool inb = (sqrtf(rx * rx + ry * ry) < 128.0f) && (Z > 0.2f) && (x1 > -64.0f) && (y1 > -64.0f) &&
                   (x1 < 2 * cx + 64.0f) && (y1 < 2 * cy + 64.0f);
  const float mask = inb ? 1.0f : 0.0f;
  T.r[0] = rx; T.r[1] = ry;
  T.w[0] = mask * weight[(int64_t)e * 2]; T.w[1] = mask * weight[(int64_t)e * 2 + 1];
  T.Jz[0] = fx * (tij[0] * d - tij[2] * (X * d2));
  T.Jz[1] = fy * (tij[1] * d - tij[2] * (Y * d2));
  T.Jj[0][0] = fx * W * d; T.Jj[0][1] = 0.0f; T.Jj[0][2] = fx * -X * W * d2;
  T.Jj[0][3] = fx * -X * Y * d2; T.Jj[0][4] = fx * (1 + X * X * d2); T.Jj[0][5] = fx * -Y * d;
  T.Jj[1][0] = 0.0f; T.Jj[1][1] = fy * W * d; T.Jj[1][2] = fy * -Y * W * d2;
  T.Jj[1][3] = fy * (-1 - Y * Y * d2); T.Jj[1][4] = fy * (X * Y * d2); T.Jj[1][5] = fy * X * d;
  fb_adjSE3(tij, qij, T.Jj[0], T.Ji[0]);
  fb_adjSE3(tij, qij, T.Jj[1], T.Ji[1]);
}

// ------------------------------------------------------------------------------------------------- accumulate
// LDS (dynamic): S_lds [n6 * LD] (lower triangle used), y_lds [n6], per-wave column buffers [ACC_WAVES][n6].
__global__ __launch_bounds__(ACC_THREADS) void k_ba_accumulate(
    const float* __restrict__ poses, const float* __restrict__ patches, const float* __restrict__ intr,
    const float* __restrict__ target, const float* __restrict__ weight, const float* __restrict__ lmbda,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, const int64_t* __restrict__ kk,
    const int* __restrict__ perm, const int* __restrict__ seg_start, const BaMeta* __restrict__ meta, int P, int t0,
    int N, float* __restrict__ partials, float* __restrict__ patch_rec, float* __restrict__ edge_e) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n6 = 6 * N, LD = n6 + 1;
  float* S_lds = smem;
  float* y_lds = S_lds + n6 * LD;
  float* col_all = y_lds + n6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* col = col_all + wave * n6;

  for (int i = tid; i < n6 * LD + n6; i += ACC_THREADS) smem[i] = 0.0f;
  __syncthreads();

  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float lm = lmbda[0];
  const int n_seg = meta->n_seg;

  for (int s = blockIdx.x * ACC_WAVES + wave; s < n_seg; s += gridDim.x * ACC_WAVES) {
    const int a0 = seg_start[s], m = seg_start[s + 1] - a0;
    for (int i = lane; i < n6; i += 64) col[i] = 0.0f;
    wave_lds_sync();
    float Csum = 0.0f, usum = 0.0f;
    unsigned fmask = 0;           // frames (relative to t0) this patch touches

    for (int base = 0; base < m; base += 64) {
      const bool act = base + lane < m;
      const int e = act ? perm[a0 + base + lane] : 0;
      EdgeTerms T;
      int ix = -1, jx = -1;
      if (act) {
        const int fi = (int)ii[e], fj = (int)jj[e];
        edge_terms(poses, patches, P, fx, fy, cx, cy, target, weight, fi, fj, (int)kk[e], e, T);
        ix = fi - t0; jx = fj - t0;
        if (ix >= N) ix = -1;
        if (jx >= N) jx = -1;
      } else {
        T.w[0] = T.w[1] = 0.0f; T.r[0] = T.r[1] = 0.0f; T.Jz[0] = T.Jz[1] = 0.0f;
#pragma unroll
        for (int c = 0; c < 6; c++) { T.Ji[0][c] = T.Ji[1][c] = T.Jj[0][c] = T.Jj[1][c] = 0.0f; }
      }
      // ---- patch-level scalars C, u (ba_cuda.cu:321-322)
      const float wz0 = T.w[0] * T.Jz[0], wz1 = T.w[1] * T.Jz[1];
      const float wr0 = T.w[0] * T.r[0], wr1 = T.w[1] * T.r[1];
      Csum += wz0 * T.Jz[0] + wz1 * T.Jz[1];
      usum += wz0 * T.r[0] + wz1 * T.r[1];
      float ej[6], ei[6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        ej[c] = (jx >= 0) ? (wz0 * T.Jj[0][c] + wz1 * T.Jj[1][c]) : 0.0f;       // E_j += w Jz Jj   (:311)
        ei[c] = (ix >= 0) ? -(wz0 * T.Ji[0][c] + wz1 * T.Ji[1][c]) : 0.0f;      // E_i -= w Jz Ji   (:309)
      }
      if (act) {
        float* rec = edge_e + ((int64_t)(a0 + base + lane)) * 12;
#pragma unroll
        for (int c = 0; c < 6; c++) { rec[c] = ej[c]; rec[6 + c] = ei[c]; }
      }
      if (N > 0) {
        if (jx >= 0) fmask |= 1u << jx;
        if (ix >= 0) fmask |= 1u << ix;
        // ---- frame-j blocks are lane-private (one edge per target frame): straight into LDS
        if (jx >= 0) {
#pragma unroll
          for (int c = 0; c < 6; c++) {
            lds_add(&col[6 * jx + c], ej[c]);
            lds_add(&y_lds[6 * jx + c], wr0 * T.Jj[0][c] + wr1 * T.Jj[1][c]);                               // v_j += w r Jj (:316)
          }
#pragma unroll
          for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c <= a; c++)
              lds_add(&S_lds[(6 * jx + a) * LD + 6 * jx + c],
                      T.w[0] * T.Jj[0][a] * T.Jj[0][c] + T.w[1] * T.Jj[1][a] * T.Jj[1][c]);                 // B_jj (:299)
          if (ix >= 0) {
            // B_ij = -w Ji Jj^T and B_ji = its transpose (:300-303): keep the one in the lower triangle
            // (both when i == j: they land on the same diagonal block).
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
              for (int c = 0; c < 6; c++) {
                const float vij = -(T.w[0] * T.Ji[0][a] * T.Jj[0][c] + T.w[1] * T.Ji[1][a] * T.Jj[1][c]);  // block (i,j)[a][c]
                if (ix > jx) lds_add(&S_lds[(6 * ix + a) * LD + 6 * jx + c], vij);
                else if (ix < jx) lds_add(&S_lds[(6 * jx + c) * LD + 6 * ix + a], vij);
                else { lds_add(&S_lds[(6 * ix + a) * LD + 6 * ix + c], vij); lds_add(&S_lds[(6 * ix + c) * LD + 6 * ix + a], vij); }
              }
          }
        }
        // ---- frame-i blocks: in DEVO graphs all edges of a patch share the source frame -> reduce the
        //      6x6 / 6x1 blocks across the wavefront with shuffles and add once; otherwise lane-private.
        const unsigned long long bi = __ballot(ix >= 0);
        if (bi) {
          const int ix0 = __shfl(ix, __ffsll((long long)bi) - 1);
          const bool uniform = (__ballot(ix >= 0 && ix != ix0) == 0ULL);
          if (uniform) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
              const float vc = wave_sum(ei[c]);
              const float vv = wave_sum((ix >= 0) ? -(wr0 * T.Ji[0][c] + wr1 * T.Ji[1][c]) : 0.0f);        // v_i -= w r Ji (:314)
              if (lane == 0) { lds_add(&col[6 * ix0 + c], vc); lds_add(&y_lds[6 * ix0 + c], vv); }
            }
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
              for (int c = 0; c <= a; c++) {
                float v = (ix >= 0) ? (T.w[0] * T.Ji[0][a] * T.Ji[0][c] + T.w[1] * T.Ji[1][a] * T.Ji[1][c]) : 0.0f;   // B_ii (:297)
                v = wave_sum(v);
                if (lane == 0) lds_add(&S_lds[(6 * ix0 + a) * LD + 6 * ix0 + c], v);
              }
          } else if (ix >= 0) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
              lds_add(&col[6 * ix + c], ei[c]);
              lds_add(&y_lds[6 * ix + c], -(wr0 * T.Ji[0][c] + wr1 * T.Ji[1][c]));
            }
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
              for (int c = 0; c <= a; c++)
                lds_add(&S_lds[(6 * ix + a) * LD + 6 * ix + c],
                        T.w[0] * T.Ji[0][a] * T.Ji[0][c] + T.w[1] * T.Ji[1][a] * T.Ji[1][c]);
          }
        }
      }
    }
    // ---- finish the patch: C, u, Q
    Csum = wave_sum(Csum);
    usum = wave_sum(usum);
    const float Q = 1.0f / (Csum + lm);                                   // ba_cuda.cu:492
    fmask = wave_or(fmask);
    if (lane == 0) { patch_rec[(int64_t)s * 2] = Q; patch_rec[(int64_t)s * 2 + 1] = usum; }
    if (N > 0) {
      wave_lds_sync();                                                    // the patch's E column is complete in LDS
      // ---- Schur complement, patch by patch:  S -= Q e e^T (lower triangle),  y -= Q u e   (:511-512)
      for (int r = lane; r < n6; r += 64) {
        const float er = col[r];
        if (er == 0.0f) continue;
        const float qer = Q * er;
        lds_add(&y_lds[r], -qer * usum);
        for (unsigned mm = fmask; mm; mm &= mm - 1) {
          const int fb = __ffs((int)mm) - 1;
          if (6 * fb > r) break;
#pragma unroll
          for (int c = 0; c < 6; c++) {
            const int cc = 6 * fb + c;
            if (cc <= r) lds_add(&S_lds[r * LD + cc], -qer * col[cc]);
          }
        }
      }
      wave_lds_sync();
    }
  }
  __syncthreads();
  if (N > 0) {
    float* out = partials + (int64_t)blockIdx.x * (n6 * LD + n6);
    for (int i = tid; i < n6 * LD + n6; i += ACC_THREADS) out[i] = smem[i];
  }
}

// S = sum of partial lower triangles, mirrored; S_dd <- S_dd*(1+1e-4)+1 (ba_cuda.cu:517-518); y = sum.
__global__ void k_ba_reduce(const float* __restrict__ partials, int n_part, int N, float* __restrict__ S, float* __restrict__ y) {
  const int n6 = 6 * N, LD = n6 + 1, stride = n6 * LD + n6;
  const int total = n6 * n6 + n6;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += blockDim.x * gridDim.x) {
    if (o < n6 * n6) {
      int r = o / n6, c = o % n6;
      int rr = max(r, c), cc = min(r, c);
      float s = 0.0f;
      for (int p = 0; p < n_part; p++) s += partials[(int64_t)p * stride + rr * LD + cc];
      if (r == c) s = s + (1e-4f * s + 1.0f);
      S[o] = s;
    } else {
      int r = o - n6 * n6;
      float s = 0.0f;
      for (int p = 0; p < n_part; p++) s += partials[(int64_t)p * stride + n6 * LD + r];
      y[r] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------- solve
// One workgroup.  A (n6+1) x LD lower-triangular working matrix lives in LDS; row n6 holds y^T, so after the
// factorisation row n6 is z = L^{-1} y.  Blocked by the 6x6 pose blocks.
constexpr int SOLVE_THREADS = 256;
__global__ __launch_bounds__(SOLVE_THREADS) void k_ba_solve(const float* __restrict__ S, const float* __restrict__ y, int N,
                                                            float* __restrict__ dX, BaMeta* meta, int iter, int* status_flag) {
  extern __shared__ __attribute__((aligned(16))) float A[];
  __shared__ int s_fail;
  const int n6 = 6 * N, LD = n6 + 1, rows = n6 + 1;
  const int tid = threadIdx.x;
  if (tid == 0) s_fail = 0;
  for (int i = tid; i < n6 * n6; i += SOLVE_THREADS) { int r = i / n6, c = i % n6; A[r * LD + c] = S[i]; }
  for (int i = tid; i < n6; i += SOLVE_THREADS) A[n6 * LD + i] = y[i];
  __syncthreads();
  if (meta->fail) return;                        // an earlier iteration broke down: the reference call has thrown by now

  for (int jb = 0; jb < N; jb++) {
    const int j0 = 6 * jb;
    // (a) factor the 6x6 diagonal block (one lane; 6 dependent pivots)
    if (tid == 0) {
      float L[6][6];
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c <= a; c++) L[a][c] = A[(j0 + a) * LD + j0 + c];
      bool ok = true;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        float d = L[c][c];
#pragma unroll
        for (int k = 0; k < c; k++) d -= L[c][k] * L[c][k];
        if (!(d > 0.0f)) ok = false;
        float ld = sqrtf(d), inv = 1.0f / ld;
        L[c][c] = ld;
#pragma unroll
        for (int a = c + 1; a < 6; a++) {
          float v = L[a][c];
#pragma unroll
          for (int k = 0; k < c; k++) v -= L[a][k] * L[c][k];
          L[a][c] = v * inv;
        }
      }
      if (!ok) s_fail = 1;
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c <= a; c++) A[(j0 + a) * LD + j0 + c] = L[a][c];
    }
    __syncthreads();
    // (b) panel: every row below solves  x L_bb^T = A[row][block]
    for (int r = j0 + 6 + tid; r < rows; r += SOLVE_THREADS) {
      float x[6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        float v = A[r * LD + j0 + c];
#pragma unroll
        for (int k = 0; k < c; k++) v -= x[k] * A[(j0 + c) * LD + j0 + k];
        x[c] = v / A[(j0 + c) * LD + j0 + c];
      }
#pragma unroll
      for (int c = 0; c < 6; c++) A[r * LD + j0 + c] = x[c];
    }
    __syncthreads();
    // (c) trailing update of the lower triangle (and of the rhs row)
    const int rem = rows - (j0 + 6);
    for (int idx = tid; idx < rem * rem; idx += SOLVE_THREADS) {
      int r = j0 + 6 + idx / rem, c = j0 + 6 + idx % rem;
      if (c > r || c >= n6) continue;
      float v = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; k++) v += A[r * LD + j0 + k] * A[c * LD + j0 + k];
      A[r * LD + c] -= v;
    }
    __syncthreads();
  }
  if (s_fail) {
    if (tid == 0) { meta->fail = iter + 1; if (status_flag) *status_flag = iter + 1; }
    return;
  }
  // back substitution  L^T x = z  (z = row n6), bottom-up by blocks; x overwrites z
  float* z = A + n6 * LD;
  for (int jb = N - 1; jb >= 0; jb--) {
    const int j0 = 6 * jb;
    if (tid == 0) {
#pragma unroll
      for (int c = 5; c >= 0; c--) {
        float v = z[j0 + c];
#pragma unroll
        for (int k = c + 1; k < 6; k++) v -= A[(j0 + k) * LD + j0 + c] * z[j0 + k];
        z[j0 + c] = v / A[(j0 + c) * LD + j0 + c];
      }
    }
    __syncthreads();
    for (int r = tid; r < j0; r += SOLVE_THREADS) {
      float v = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; k++) v += A[(j0 + k) * LD + r] * z[j0 + k];
      z[r] -= v;
    }
    __syncthreads();
  }
  for (int i = tid; i < n6; i += SOLVE_THREADS) dX[i] = z[i];
}

// ------------------------------------------------------------------------------------------------- retract
// poses[t0+i] <- Exp(dX_i) * poses[t0+i]  (ba_cuda.cu:160-188);  d <- d + dz; d>20 -> 1; d >= 1e-4 (:191-211)
__global__ void k_ba_retract(float* __restrict__ poses, float* __restrict__ patches, const float* __restrict__ dX,
                             const float* __restrict__ patch_rec, const float* __restrict__ edge_e,
                             const int64_t* __restrict__ jj, const int* __restrict__ perm, const int* __restrict__ seg_start,
                             const int* __restrict__ kx, const int64_t* __restrict__ ii, const BaMeta* __restrict__ meta, int P,
                             int t0, int N) {
  if (meta->fail) return;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = blockDim.x * gridDim.x;
  const int n_seg = meta->n_seg;
  for (int s = gid; s < n_seg; s += gsz) {
    const float* rec = patch_rec + (int64_t)s * 2;
    const int a0 = seg_start[s], m = seg_start[s + 1] - a0;
    float acc = rec[1];                                            // u
    if (N > 0) {
      for (int q = 0; q < m; q++) {
        const int e = perm[a0 + q];
        const int ix = (int)ii[e] - t0, jx = (int)jj[e] - t0;
        const float* er = edge_e + (int64_t)(a0 + q) * 12;
        if (jx >= 0 && jx < N) {
#pragma unroll
          for (int c = 0; c < 6; c++) acc -= er[c] * dX[6 * jx + c];
        }
        if (ix >= 0 && ix < N) {
#pragma unroll
          for (int c = 0; c < 6; c++) acc -= er[6 + c] * dX[6 * ix + c];
        }
      }
    }
    const float dz = rec[0] * acc;                                 // Q (u - E^T dX)   (ba_cuda.cu:523)
    float* pd = patches + ((int64_t)kx[s] * 3 + 2) * P * P;
    float d = pd[0] + dz;                                          // reads pixel [0][0] (:198)
    d = (d > 20.0f) ? 1.0f : d;
    d = fmaxf(d, 1e-4f);
    for (int i = 0; i < P * P; i++) pd[i] = d;
  }
  for (int t = gid; t < N; t += gsz) {
    float* p = poses + (int64_t)(t0 + t) * 7;
    float tt[3] = {p[0], p[1], p[2]}, q[4] = {p[3], p[4], p[5], p[6]}, t1[3], q1[4];
    fb_retrSE3(dX + 6 * t, tt, q, t1, q1);
    p[0] = t1[0]; p[1] = t1[1]; p[2] = t1[2]; p[3] = q1[0]; p[4] = q1[1]; p[5] = q1[2]; p[6] = q1[3];
  }
}

// ------------------------------------------------------------------------------------------------- reproject / transform
__global__ void k_reproject(const float* __restrict__ poses, const float* __restrict__ patches, const float* __restrict__ intr,
                            const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, const int64_t* __restrict__ kk,
                            float* __restrict__ coords, int E, int P) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int PPx = P * P;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += blockDim.x * gridDim.x) {
    const float* pi = poses + ii[e] * 7;
    const float* pj = poses + jj[e] * 7;
    float ti[3] = {pi[0], pi[1], pi[2]}, qi[4] = {pi[3], pi[4], pi[5], pi[6]};
    float tj[3] = {pj[0], pj[1], pj[2]}, qj[4] = {pj[3], pj[4], pj[5], pj[6]};
    float tij[3], qij[4];
    fb_relSE3(ti, qi, tj, qj, tij, qij);
    const float* pk = patches + kk[e] * 3 * PPx;
    float* out = coords + (int64_t)e * 2 * PPx;
    for (int i = 0; i < PPx; i++) {
      float Xi[4] = {(pk[i] - cx) / fx, (pk[PPx + i] - cy) / fy, 1.0f, pk[2 * PPx + i]}, Xj[4];
      fb_actSE3(tij, qij, Xi, Xj);
      out[i] = fx * (Xj[0] / Xj[2]) + cx;
      out[PPx + i] = fy * (Xj[1] / Xj[2]) + cy;
    }
  }
}

// devo/projective_ops.py:53-105 fused: iproj (per-frame intrinsics of frame i) -> Gij = Gj * Gi^-1 (lietorch
// semantics: quaternions renormalised on load) -> act4 -> proj (intrinsics of frame j, Z clamped at 0.1).
__global__ void k_transform(const float* __restrict__ poses, const float* __restrict__ patches, const float* __restrict__ intr,
                            const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, const int64_t* __restrict__ kk,
                            float* __restrict__ c_pp2, float* __restrict__ c_2pp, float* __restrict__ valid,
                            float* __restrict__ Ji, float* __restrict__ Jj, float* __restrict__ Jz, int E, int P, int flags) {
  const bool depth = flags & 1, tonly = flags & 2;
  const int PPx = P * P, ctr = (P / 2) * P + P / 2, nc = depth ? 3 : 2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += blockDim.x * gridDim.x) {
    const int64_t fi = ii[e], fj = jj[e];
    SE3<float> Gi = SE3<float>::load(poses + fi * 7), Gj = SE3<float>::load(poses + fj * 7);
    SE3<float> G = Gj.mul(Gi.inv());
    if (tonly) G.q = Q4<float>{0.0f, 0.0f, 0.0f, 1.0f};
    const float fxi = intr[fi * 4], fyi = intr[fi * 4 + 1], cxi = intr[fi * 4 + 2], cyi = intr[fi * 4 + 3];
    const float fxj = intr[fj * 4], fyj = intr[fj * 4 + 1], cxj = intr[fj * 4 + 2], cyj = intr[fj * 4 + 3];
    const float* pk = patches + kk[e] * 3 * PPx;
    float Xc = 0, Yc = 0, Zc = 1, Hc = 0;
    for (int i = 0; i < PPx; i++) {
      const float w = pk[2 * PPx + i];
      V3<float> X0{(pk[i] - cxi) / fxi, (pk[PPx + i] - cyi) / fyi, 1.0f};
      V3<float> X1 = qrot(G.q, X0) + w * G.t;
      if (i == ctr) { Xc = X1.x; Yc = X1.y; Zc = X1.z; Hc = w; }
      const float d = 1.0f / fmaxf(X1.z, 0.1f);
      const float u = fxj * (d * X1.x) + cxj, v = fyj * (d * X1.y) + cyj;
      if (c_pp2) { float* o = c_pp2 + ((int64_t)e * PPx + i) * nc; o[0] = u; o[1] = v; if (depth) o[2] = d; }
      if (c_2pp) { c_2pp[(int64_t)e * 2 * PPx + i] = u; c_2pp[(int64_t)e * 2 * PPx + PPx + i] = v; }
    }
    if (valid) valid[e] = (Zc > 0.2f) ? 1.0f : 0.0f;
    if (Jj) {
      const float d = (fabsf(Zc) > 0.2f) ? 1.0f / Zc : 0.0f;
      float J[2][6] = {{fxj * d * Hc, 0.0f, -fxj * Xc * d * d * Hc, -fxj * Xc * d * d * Yc, fxj * d * Zc + fxj * Xc * d * d * Xc, -fxj * d * Yc},
                       {0.0f, fyj * d * Hc, -fyj * Yc * d * d * Hc, -fyj * d * Zc - fyj * Yc * d * d * Yc, fyj * Yc * d * d * Xc, fyj * d * Xc}};
#pragma unroll
      for (int r = 0; r < 2; r++) {
        float a[6];
        G.adjT(J[r], a);
#pragma unroll
        for (int c = 0; c < 6; c++) { Jj[((int64_t)e * 2 + r) * 6 + c] = J[r][c]; if (Ji) Ji[((int64_t)e * 2 + r) * 6 + c] = -a[c]; }
      }
      if (Jz) {
        Jz[(int64_t)e * 2] = fxj * d * G.t.x - fxj * Xc * d * d * G.t.z;
        Jz[(int64_t)e * 2 + 1] = fyj * d * G.t.y - fyj * Yc * d * d * G.t.z;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- neighbors
__device__ __forceinline__ unsigned hash64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return (unsigned)k;
}
__global__ void k_hash_group(const int64_t* __restrict__ ii, int E, unsigned long long* keys, unsigned cap_mask, int* slot_of, int* counts) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += blockDim.x * gridDim.x) {
    const unsigned long long key = (unsigned long long)ii[e];
    unsigned h = hash64(key) & cap_mask;
    for (;;) {
      unsigned long long prev = atomicCAS(&keys[h], ~0ULL, key);
      if (prev == ~0ULL || prev == key) break;
      h = (h + 1) & cap_mask;
    }
    slot_of[e] = (int)h;
    atomicAdd(&counts[h], 1);
  }
}
// ba.cpp:127-139: within the edges that share ii, order by (jj, edge index); previous / next or -1.
__global__ void k_neighbors(const int64_t* __restrict__ jj, int E, const int* __restrict__ slot_of, const int* __restrict__ start,
                            const int* __restrict__ perm, int64_t* __restrict__ ix, int64_t* __restrict__ jx) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += blockDim.x * gridDim.x) {
    const int s = slot_of[e], a = start[s], b = start[s + 1];
    const int64_t je = jj[e];
    int64_t pj = 0, nj = 0; int pe = -1, ne = -1;
    for (int q = a; q < b; q++) {
      const int o = perm[q];
      if (o == e) continue;
      const int64_t jo = jj[o];
      const bool less = (jo < je) || (jo == je && o < e);
      if (less) { if (pe < 0 || jo > pj || (jo == pj && o > pe)) { pe = o; pj = jo; } }
      else      { if (ne < 0 || jo < nj || (jo == nj && o < ne)) { ne = o; nj = jo; } }
    }
    ix[e] = pe; jx[e] = ne;
  }
}

// ------------------------------------------------------------------------------------------------- workspace
struct BaLayout {
  size_t meta, rank, counts, cursor, ku, perm_a, perm_b, kx, partials, S, y, dX, patch_rec, edge_ej, total;
  int max_seg, n_part;
};
static BaLayout ba_layout(int E, int Np, int N) {
  BaLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  L.max_seg = E < Np ? E : Np;
  if (L.max_seg < 1) L.max_seg = 1;
  int want = (L.max_seg + ACC_WAVES - 1) / ACC_WAVES;
  L.n_part = want < ACC_MAX_WG ? (want < 1 ? 1 : want) : ACC_MAX_WG;
  const size_t n6 = 6 * (size_t)N;
  L.meta = take(sizeof(BaMeta));
  L.rank = take(sizeof(int) * ((size_t)Np + 1));
  L.counts = take(sizeof(int) * ((size_t)L.max_seg + 1));
  L.cursor = take(sizeof(int) * (size_t)L.max_seg);
  L.ku = take(sizeof(int) * (size_t)(E > 0 ? E : 1));
  L.perm_a = take(sizeof(int) * (size_t)(E > 0 ? E : 1));
  L.perm_b = take(sizeof(int) * (size_t)(E > 0 ? E : 1));
  L.kx = take(sizeof(int) * (size_t)L.max_seg);
  L.partials = take(sizeof(float) * (size_t)L.n_part * (n6 * (n6 + 1) + n6 + 1));
  L.S = take(sizeof(float) * (n6 * n6 + 1));
  L.y = take(sizeof(float) * (n6 + 1));
  L.dX = take(sizeof(float) * (n6 + 1));
  L.patch_rec = take(sizeof(float) * 2 * (size_t)L.max_seg);
  L.edge_ej = take(sizeof(float) * 12 * (size_t)(E > 0 ? E : 1));
  L.total = off;
  return L;
}

static unsigned next_pow2(unsigned v) { unsigned p = 1; while (p < v) p <<= 1; return p; }

}  // namespace devo

using namespace devo;

extern "C" {

size_t devo_ba_workspace_bytes(int E, int Np, int N) {
  if (E < 0 || Np < 0 || N < 0 || N > BA_MAXN) return 0;
  return ba_layout(E, Np, N).total;
}

int devo_ba_forward(float* poses, float* patches, const float* intrinsics, const float* target, const float* weight,
                    const float* lmbda, const int64_t* ii, const int64_t* jj, const int64_t* kk, int E, int Nbuf, int Np,
                    int P, int t0, int t1, int iterations, void* ws, size_t ws_bytes, int* status_flag,
                    devo_stream_t stream) {
  const int N = t1 - t0;
  DEVO_REQUIRE(E >= 0 && Np > 0 && Nbuf > 0 && P > 0, "devo_ba_forward: bad sizes");
  DEVO_REQUIRE(N >= 0 && t0 >= 0 && t1 <= Nbuf, "devo_ba_forward: bad pose window [%d,%d) for %d poses", t0, t1, Nbuf);
  if (N > BA_MAXN) { set_error("devo_ba_forward: %d optimised poses > %d supported", N, BA_MAXN); return DEVO_ERR_UNSUPPORTED; }
  if (E == 0 || iterations <= 0) return DEVO_OK;
  const BaLayout L = ba_layout(E, Np, N);
  if (ws == nullptr || ws_bytes < L.total) { set_error("devo_ba_forward: workspace %zu < %zu bytes", ws_bytes, L.total); return DEVO_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)ws;
  BaMeta* meta = (BaMeta*)(w + L.meta);
  int* rank = (int*)(w + L.rank);
  int* counts = (int*)(w + L.counts);
  int* cursor = (int*)(w + L.cursor);
  int* ku = (int*)(w + L.ku);
  int* perm_a = (int*)(w + L.perm_a);
  int* perm_b = (int*)(w + L.perm_b);
  int* kx = (int*)(w + L.kx);
  float* partials = (float*)(w + L.partials);
  float* S = (float*)(w + L.S);
  float* y = (float*)(w + L.y);
  float* dX = (float*)(w + L.dX);
  float* patch_rec = (float*)(w + L.patch_rec);
  float* edge_ej = (float*)(w + L.edge_ej);

  // ---- graph preparation: kx = unique(kk) sorted, ku = inverse (ba_cuda.cu:435-437), edges grouped by patch
  // (meta, rank, counts, cursor are contiguous at the head of the workspace)
  if (hipMemsetAsync(w + L.meta, 0, L.ku - L.meta, st) != hipSuccess) { set_error("devo_ba_forward: memset failed"); return DEVO_ERR_LAUNCH; }
  if (status_flag && hipMemsetAsync(status_flag, 0, sizeof(int), st) != hipSuccess) { set_error("devo_ba_forward: memset failed"); return DEVO_ERR_LAUNCH; }
  const int eb = blocks_for(E, 256, 1024);
  hipLaunchKernelGGL(k_flag_ids, dim3(eb), dim3(256), 0, st, kk, E, Np, rank);
  hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, st, rank, Np, &meta->n_seg);
  hipLaunchKernelGGL(k_rank_edges, dim3(blocks_for(E > Np ? E : Np, 256, 1024)), dim3(256), 0, st, kk, E, Np, rank, ku, kx, counts);
  hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, st, counts, L.max_seg, (int*)nullptr);
  hipLaunchKernelGGL(k_scatter_edges, dim3(eb), dim3(256), 0, st, ku, E, counts, cursor, perm_a);
  hipLaunchKernelGGL(k_sort_segments, dim3(blocks_for((long long)L.max_seg * 64, 256, 1024)), dim3(256), 0, st, counts, &meta->n_seg, perm_a, perm_b);
  int rc = check_launch("devo_ba_forward(prepare)");
  if (rc) return rc;

  const size_t n6 = 6 * (size_t)N;
  const size_t acc_lds = sizeof(float) * (n6 * (n6 + 1) + n6 + ACC_WAVES * n6 + 4);
  const size_t solve_lds = sizeof(float) * ((n6 + 1) * (n6 + 1) + 4);
  if (acc_lds > 64 * 1024 || solve_lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)k_ba_accumulate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_ba_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds) != hipSuccess) {
      (void)hipGetLastError();
      set_error("devo_ba_forward: cannot reserve %zu / %zu bytes of LDS", acc_lds, solve_lds);
      return DEVO_ERR_LAUNCH;
    }
  }
  for (int it = 0; it < iterations; it++) {
    hipLaunchKernelGGL(k_ba_accumulate, dim3(L.n_part), dim3(ACC_THREADS), acc_lds, st, poses, patches, intrinsics, target,
                       weight, lmbda, ii, jj, kk, perm_b, counts, meta, P, t0, N, partials, patch_rec, edge_ej);
    if ((rc = check_launch("devo_ba_forward(accumulate)"))) return rc;
    if (N > 0) {
      hipLaunchKernelGGL(k_ba_reduce, dim3(blocks_for((long long)(n6 * n6 + n6), 256, 256)), dim3(256), 0, st, partials, L.n_part, N, S, y);
      if ((rc = check_launch("devo_ba_forward(reduce)"))) return rc;
      hipLaunchKernelGGL(k_ba_solve, dim3(1), dim3(SOLVE_THREADS), solve_lds, st, S, y, N, dX, meta, it, status_flag);
      if ((rc = check_launch("devo_ba_forward(solve)"))) return rc;
    }
    hipLaunchKernelGGL(k_ba_retract, dim3(blocks_for(L.max_seg > N ? L.max_seg : N, 256, 1024)), dim3(256), 0, st, poses, patches, dX, patch_rec,
                       edge_ej, jj, perm_b, counts, kx, ii, meta, P, t0, N);
    if ((rc = check_launch("devo_ba_forward(retract)"))) return rc;
  }
  return check_launch("devo_ba_forward");
}

size_t devo_neighbors_workspace_bytes(int E) {
  if (E <= 0) return 256;
  const size_t cap = next_pow2((unsigned)(2 * (size_t)E));
  return align_up(8 * cap) + align_up(4 * (cap + 1)) + align_up(4 * cap) + 2 * align_up(4 * (size_t)E);
}

int devo_ba_neighbors(const int64_t* ii, const int64_t* jj, int64_t* ix, int64_t* jx, int E, void* ws, size_t ws_bytes,
                      devo_stream_t stream) {
  if (E <= 0) return DEVO_OK;
  const size_t need = devo_neighbors_workspace_bytes(E);
  if (ws == nullptr || ws_bytes < need) { set_error("devo_ba_neighbors: workspace %zu < %zu bytes", ws_bytes, need); return DEVO_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  const size_t cap = next_pow2((unsigned)(2 * (size_t)E));
  char* w = (char*)ws;
  unsigned long long* keys = (unsigned long long*)w; w += align_up(8 * cap);
  int* counts = (int*)w; w += align_up(4 * (cap + 1));
  int* cursor = (int*)w; w += align_up(4 * cap);
  int* slot_of = (int*)w; w += align_up(4 * (size_t)E);
  int* perm = (int*)w;
  if (hipMemsetAsync(keys, 0xFF, 8 * cap, st) != hipSuccess ||
      hipMemsetAsync(counts, 0, (char*)slot_of - (char*)counts, st) != hipSuccess) { set_error("devo_ba_neighbors: memset failed"); return DEVO_ERR_LAUNCH; }
  const int eb = blocks_for(E, 256, 1024);
  hipLaunchKernelGGL(k_hash_group, dim3(eb), dim3(256), 0, st, ii, E, keys, (unsigned)(cap - 1), slot_of, counts);
  hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, st, counts, (int)cap, (int*)nullptr);
  hipLaunchKernelGGL(k_scatter_edges, dim3(eb), dim3(256), 0, st, slot_of, E, counts, cursor, perm);
  hipLaunchKernelGGL(k_neighbors, dim3(eb), dim3(256), 0, st, jj, E, slot_of, counts, perm, ix, jx);
  return check_launch("devo_ba_neighbors");
}

int devo_ba_reproject(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii, const int64_t* jj,
                      const int64_t* kk, float* coords, int E, int P, devo_stream_t stream) {
  if (E <= 0) return DEVO_OK;
  hipLaunchKernelGGL(k_reproject, dim3(blocks_for(E, 128, 4096)), dim3(128), 0, (hipStream_t)stream, poses, patches, intrinsics, ii, jj, kk, coords, E, P);
  return check_launch("devo_ba_reproject");
}

int devo_transform(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii, const int64_t* jj,
                   const int64_t* kk, float* coords_pp2, float* coords_2pp, float* valid, float* Ji, float* Jj, float* Jz, int E,
                   int P, int flags, devo_stream_t stream) {
  if (E <= 0) return DEVO_OK;
  DEVO_REQUIRE(!(Ji || Jz) || Jj, "devo_transform: Jj must be requested together with Ji / Jz");
  hipLaunchKernelGGL(k_transform, dim3(blocks_for(E, 128, 4096)), dim3(128), 0, (hipStream_t)stream, poses, patches, intrinsics, ii, jj,
                     kk, coords_pp2, coords_2pp, valid, Ji, Jj, Jz, E, P, flags);
  return check_launch("devo_transform");
}

}  // extern "C"
