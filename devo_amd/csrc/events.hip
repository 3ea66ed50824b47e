// Event stream -> voxel grid, and the voxel standardisation (utils/event_utils.py:180-232,
// utils/voxel_utils.py:6-28 == devo/devo.py:438-452) — the step in front of the feature encoders, on the GPU so that a
// raw event stream can be handed over as device arrays.
//   * k_voxelize: one event per thread votes into the 2x2x2 neighbouring voxels of (t, y, x); weights are evaluated in
//     fp64 like the reference (x, y, t are fp64 there) and added as fp32 with hardware float atomics (HBM/L2 atomics;
//     the summation ORDER differs from the reference's eight sequential index_add_ passes: results agree to fp32
//     rounding, not bit for bit).
//   * k_voxel_stats / k_voxel_normalise: count, sum and sum of squares of the non-zero voxels of every segment (fp64
//     accumulation; the reference sums in fp32), then  v <- (v != 0) * (v - mean) / std.
#include "common.h"

namespace devo {

__global__ void k_voxelize(const float* __restrict__ xs, const float* __restrict__ ys, const double* __restrict__ ts,
                           const signed char* __restrict__ ps, int64_t N, int H, int W, int bins, float* __restrict__ grid) {
  const double t0 = ts[0], dur = ts[N - 1] - t0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const double x = (double)xs[i], y = (double)ys[i];
    const double t = (ts[i] - t0) * (double)(bins - 1) / dur;
    const float pol = (ps[i] == 0) ? -1.0f : (float)ps[i];                  // event_utils.py:198
    const double fx = floor(x), fy = floor(y), ft = floor(t);
#pragma unroll
    for (int cx = 0; cx < 2; cx++)
#pragma unroll
      for (int cy = 0; cy < 2; cy++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          const double lx = fx + cx, ly = fy + cy, lt = ft + ct;
          if (lx >= 0 && ly >= 0 && lt >= 0 && lx <= W - 1 && ly <= H - 1 && lt <= bins - 1) {
            const double w = (double)pol * (1.0 - fabs(lx - x)) * (1.0 - fabs(ly - y)) * (1.0 - fabs(lt - t));
            atomicAdd(grid + ((int64_t)lt * H + (int64_t)ly) * W + (int64_t)lx, (float)w);
          }
        }
  }
}

// stats[s] = {count of non-zeros, sum, sum of squares} of segment s (len elements each); 3 doubles per segment,
// zeroed by the launcher; per-workgroup partials are combined with fp64 atomics.
__global__ __launch_bounds__(256) void k_voxel_stats(const float* __restrict__ v, int64_t len, double* __restrict__ stats) {
  __shared__ double s_red[3][4];
  const int seg = blockIdx.y;
  const float* p = v + (int64_t)seg * len;
  double c = 0.0, s = 0.0, q = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) {
    const float a = p[i];
    if (a != 0.0f) { c += 1.0; s += (double)a; q += (double)a * (double)a; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { c += __shfl_xor(c, off); s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_red[0][wave] = c; s_red[1][wave] = s; s_red[2][wave] = q; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double t = (s_red[threadIdx.x][0] + s_red[threadIdx.x][1]) + (s_red[threadIdx.x][2] + s_red[threadIdx.x][3]);
    atomicAdd(stats + 3 * seg + threadIdx.x, t);
  }
}

// all_nonempty = every segment has a non-zero (the reference normalises only then, voxel_utils.py:19)
__global__ void k_voxel_normalise(float* __restrict__ v, int64_t len, int nseg, const double* __restrict__ stats) {
  for (int s = 0; s < nseg; s++) if (!(stats[3 * s] > 0.0)) return;
  const int seg = blockIdx.y;
  const double cnt = stats[3 * seg];
  const float mean = (float)(stats[3 * seg + 1] / cnt);
  const float sd = sqrtf((float)(stats[3 * seg + 2] / cnt) - mean * mean);
  float* p = v + (int64_t)seg * len;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = p[i];
    p[i] = (a != 0.0f) ? (a - mean) / sd : 0.0f * ((a - mean) / sd);       // mask * (...): 0 * finite = 0, like the reference
  }
}

}  // namespace devo

using namespace devo;

extern "C" {

int devo_voxelize(const float* xs, const float* ys, const double* ts, const signed char* ps, int64_t N, int H, int W, int bins,
                  float* grid, devo_stream_t stream) {
  DEVO_REQUIRE(N >= 0 && H > 0 && W > 0 && bins > 0, "devo_voxelize: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(grid, 0, sizeof(float) * (size_t)bins * H * W, st) != hipSuccess) { set_error("devo_voxelize: memset failed"); return DEVO_ERR_LAUNCH; }
  if (N == 0) return DEVO_OK;                                  // empty stream: empty grid (event_utils.py:187)
  hipLaunchKernelGGL(k_voxelize, dim3(blocks_for(N, 256, 4096)), dim3(256), 0, st, xs, ys, ts, ps, N, H, W, bins, grid);
  return check_launch("devo_voxelize");
}

size_t devo_voxel_std_workspace_bytes(int nseg) { return sizeof(double) * 3 * (size_t)(nseg > 0 ? nseg : 1); }

int devo_voxel_std(float* vox, int nseg, int64_t len, void* ws, size_t ws_bytes, devo_stream_t stream) {
  DEVO_REQUIRE(nseg >= 0 && len >= 0, "devo_voxel_std: bad sizes");
  if (nseg == 0 || len == 0) return DEVO_OK;
  if (ws == nullptr || ws_bytes < devo_voxel_std_workspace_bytes(nseg)) { set_error("devo_voxel_std: workspace too small"); return DEVO_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, devo_voxel_std_workspace_bytes(nseg), st) != hipSuccess) { set_error("devo_voxel_std: memset failed"); return DEVO_ERR_LAUNCH; }
  const unsigned bx = (unsigned)blocks_for(len, 256 * 8, 512);
  hipLaunchKernelGGL(k_voxel_stats, dim3(bx, (unsigned)nseg), dim3(256), 0, st, vox, len, (double*)ws);
  hipLaunchKernelGGL(k_voxel_normalise, dim3(bx, (unsigned)nseg), dim3(256), 0, st, vox, len, nseg, (const double*)ws);
  return check_launch("devo_voxel_std");
}

}  // extern "C"
