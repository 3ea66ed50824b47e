// altcorr for gfx950: patch <-> frame local correlation lookup (forward + backward) and patchify.
// Replaces devo/altcorr/correlation_kernel.cu (reference module cuda_corr, correlation.cpp:57-63).
//
// Design: the reference launches one thread per (edge, pixel, tap) and walks the
// 128 channels with a stride of H*W elements, then blends/permutes with ~14 ATen kernels.  Here ONE
// workgroup owns one edge and works POSITION-centric: the 9 patch pixels' (2R+2)^2 windows overlap almost
// entirely, so the union bounding box (~10x10 px) is staged once through LDS with coalesced 16-byte loads
// from a channels-last pyramid, every lane owns one position of the box and keeps 9 accumulators (one per
// patch pixel) whose f1 operands are wave-uniform scalar (SGPR) loads, and the bilinear blend + axis swap
// + output permutation are fused into the epilogue (no raw D x D tensor, no temporaries, no stack copy).
#include "common.h"
#include <hip/hip_fp16.h>

namespace devo {

constexpr int PP = 9;          // patch pixels (P = 3)
constexpr int MAXD = 12;       // 2*R+2 for R <= 5
constexpr int KC = 32;         // channels staged per LDS chunk
constexpr int ROWPAD = KC + 4; // LDS row stride in floats: conflict-free ds_read_b128 (36*l mod 64 distinct per 16 lanes)
constexpr int NT = 128;        // threads per workgroup (2 waves): one chunk of 128 box positions
constexpr int MAXPOS = 512;    // largest bounding box handled by the staged path

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f32<double>(double v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ double from_f32<double>(float v) { return (double)v; }

__device__ __forceinline__ int floor_to_int(float v) {
  // static_cast<int>(floor(v)) (correlation_kernel.cu:118-119), made safe for non-finite / huge inputs
  float f = floorf(v);
  f = fminf(fmaxf(f, -1.0e6f), 1.0e6f);
  return (f == f) ? (int)f : -1000000;
}

struct EdgeGeom {
  int ox[PP], oy[PP];     // window origin (tap 0,0) of each patch pixel in frame coordinates
  float dx[PP], dy[PP];   // sub-pixel fractions
};

// Bilinear epilogue of correlation_kernel.cu:221-232 on the raw window held in LDS.
// Evaluated without FMA contraction in the reference's order: ((1-dx)(1-dy))*r00 + (dx(1-dy))*r01 + ...
__device__ __forceinline__ float blend4(float dx, float dy, float r00, float r01, float r10, float r11) {
#pragma clang fp contract(off)
  float o = ((1.0f - dx) * (1.0f - dy)) * r00;
  o = o + (dx * (1.0f - dy)) * r01;
  o = o + ((1.0f - dx) * dy) * r10;
  o = o + (dx * dy) * r11;
  return o;
}

template <typename T>
__device__ __forceinline__ void corr_epilogue(const float* sraw, const float* sdx, const float* sdy, T* outp,
                                              int D, int64_t lstride) {
  const int Dm = D - 1;
  const int total = Dm * Dm * PP;
  for (int l = threadIdx.x; l < total; l += blockDim.x) {
    int p = l % PP;              // i0*3 + j0
    int a = (l / PP) % Dm;       // y offset  (logical dim 3)
    int c = l / (PP * Dm);       // x offset  (logical dim 2: permute(0,1,3,2,4,5), correlation_kernel.cu:232)
    const float* r = sraw + p * D * D + a * D + c;
    float o = blend4(sdx[p], sdy[p], r[0], r[1], r[D], r[D + 1]);
    outp[(int64_t)l * lstride] = from_f32<T>(o);
  }
}

// -------------------------------------------------------------------------------------------------
// Fast path: fmap2 channels-last (channel stride 1), C % KC == 0.
// grid = B*E workgroups of NT threads.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void corr_fwd_cl_kernel(
    const T* __restrict__ fmap1, const T* __restrict__ fmap2, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, T* __restrict__ out, int E, int Np, int n2,
    int C, int H2, int W2, int64_t s_b, int64_t s_n, int64_t s_h, int64_t s_w, int64_t out_estride,
    int64_t out_lstride, int64_t out_offset, int R) {
  __shared__ __attribute__((aligned(16))) float s_f2[NT * ROWPAD];
  __shared__ float s_raw[PP * MAXD * MAXD];
  __shared__ float s_dx[PP], s_dy[PP];
  __shared__ int s_ox[PP], s_oy[PP];

  const int D = 2 * R + 2;
  const int be = blockIdx.x;
  const int b = be / E, e = be % E;
  const int tid = threadIdx.x;

  if (tid < PP) {
    float x = coords[((int64_t)be * 2 + 0) * PP + tid];
    float y = coords[((int64_t)be * 2 + 1) * PP + tid];
    float fx = floorf(x), fy = floorf(y);
    s_ox[tid] = floor_to_int(x) - R;
    s_oy[tid] = floor_to_int(y) - R;
    s_dx[tid] = x - fx;
    s_dy[tid] = y - fy;
  }
  for (int i = tid; i < PP * D * D; i += NT) s_raw[i] = 0.0f;
  __syncthreads();

  int xmin = s_ox[0], xmax = s_ox[0], ymin = s_oy[0], ymax = s_oy[0];
#pragma unroll
  for (int p = 1; p < PP; p++) {
    xmin = min(xmin, s_ox[p]); xmax = max(xmax, s_ox[p]);
    ymin = min(ymin, s_oy[p]); ymax = max(ymax, s_oy[p]);
  }
  const int bw = xmax - xmin + D, bh = ymax - ymin + D;
  const long long npos_ll = (long long)bw * bh;

  const int64_t pi = ii[e];
  const int64_t fj = jj[e];
  const T* __restrict__ f1 = fmap1 + ((int64_t)b * Np + pi) * C * PP;           // [C][9], wave-uniform
  const T* __restrict__ f2 = fmap2 + (int64_t)b * s_b + fj * s_n;
  T* outp = out + (int64_t)be * out_estride + out_offset;

  if (npos_ll <= MAXPOS) {
    const int npos = (int)npos_ll;
    for (int base = 0; base < npos; base += NT) {
      const int mypos = base + tid;
      const int py = mypos / bw, px = mypos - py * bw;
      float acc[PP];
#pragma unroll
      for (int p = 0; p < PP; p++) acc[p] = 0.0f;

      for (int kc = 0; kc < C; kc += KC) {
        // ---- stage [NT positions][KC channels] of frame fj into LDS (16 B per lane, 8 lanes per position)
        constexpr int VEC = 16 / sizeof(T);            // elements per 16-byte load
        constexpr int PARTS = KC / VEC;                // 16-byte loads per position
#pragma unroll
        for (int it = 0; it < PARTS; it++) {
          int q = tid + it * NT;
          int pos = q / PARTS, part = q - pos * PARTS;
          int gp = base + pos;
          int gy = ymin + gp / bw, gx = xmin + (gp % bw);
          bool ok = (gp < npos) && gy >= 0 && gy < H2 && gx >= 0 && gx < W2;
          float v[VEC];
          if (ok) {
            const T* src = f2 + (int64_t)gy * s_h + (int64_t)gx * s_w + kc + part * VEC;
            uint4 raw = *reinterpret_cast<const uint4*>(src);
            const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int u = 0; u < VEC; u++) v[u] = to_f32<T>(rv[u]);
          } else {
#pragma unroll
            for (int u = 0; u < VEC; u++) v[u] = 0.0f;
          }
          float* dst = s_f2 + pos * ROWPAD + part * VEC;
#pragma unroll
          for (int u = 0; u < VEC; u += 4) *reinterpret_cast<float4*>(dst + u) = make_float4(v[u], v[u + 1], v[u + 2], v[u + 3]);
        }
        __syncthreads();
        // ---- 9 accumulators per position; f1 operands are wave-uniform
        const float* row = s_f2 + tid * ROWPAD;
#pragma unroll
        for (int k = 0; k < KC; k += 4) {
          float4 v = *reinterpret_cast<const float4*>(row + k);
          const T* w = f1 + (int64_t)(kc + k) * PP;
#pragma unroll
          for (int p = 0; p < PP; p++) {
            acc[p] = fmaf(to_f32<T>(w[p]), v.x, acc[p]);
            acc[p] = fmaf(to_f32<T>(w[PP + p]), v.y, acc[p]);
            acc[p] = fmaf(to_f32<T>(w[2 * PP + p]), v.z, acc[p]);
            acc[p] = fmaf(to_f32<T>(w[3 * PP + p]), v.w, acc[p]);
          }
        }
        __syncthreads();
      }
      // ---- scatter this position's 9 sums into the per-pixel raw windows
      if (mypos < npos) {
        const int gy = ymin + py, gx = xmin + px;
#pragma unroll
        for (int p = 0; p < PP; p++) {
          int a = gy - s_oy[p], c = gx - s_ox[p];
          if (a >= 0 && a < D && c >= 0 && c < D) s_raw[p * D * D + a * D + c] = acc[p];
        }
      }
    }
  } else {
    // ---- patch pixels spread far apart: evaluate the 9 windows one tap at a time (rare)
    for (int o = tid; o < PP * D * D; o += NT) {
      int p = o / (D * D), a = (o / D) % D, c = o % D;
      int gy = s_oy[p] + a, gx = s_ox[p] + c;
      float s = 0.0f;
      if (gy >= 0 && gy < H2 && gx >= 0 && gx < W2) {
        const T* src = f2 + (int64_t)gy * s_h + (int64_t)gx * s_w;
        for (int k = 0; k < C; k++) s = fmaf(to_f32<T>(f1[k * PP + p]), to_f32<T>(src[k]), s);
      }
      s_raw[o] = s;
    }
  }
  __syncthreads();
  corr_epilogue<T>(s_raw, s_dx, s_dy, outp, D, out_lstride);
}

// -------------------------------------------------------------------------------------------------
// Generic path: arbitrary fmap2 strides (e.g. the reference's NCHW pyramid), any C.
// One workgroup per edge, one tap per lane, channel loop with the tensor's own strides.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void corr_fwd_generic_kernel(
    const T* __restrict__ fmap1, const T* __restrict__ fmap2, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, T* __restrict__ out, int E, int Np, int n2,
    int C, int H2, int W2, int64_t s_b, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int64_t out_estride,
    int64_t out_lstride, int64_t out_offset, int R) {
  __shared__ float s_raw[PP * MAXD * MAXD];
  __shared__ float s_dx[PP], s_dy[PP];
  __shared__ int s_ox[PP], s_oy[PP];
  const int D = 2 * R + 2;
  const int be = blockIdx.x;
  const int b = be / E, e = be % E;
  const int tid = threadIdx.x;
  if (tid < PP) {
    float x = coords[((int64_t)be * 2 + 0) * PP + tid];
    float y = coords[((int64_t)be * 2 + 1) * PP + tid];
    s_ox[tid] = floor_to_int(x) - R;
    s_oy[tid] = floor_to_int(y) - R;
    s_dx[tid] = x - floorf(x);
    s_dy[tid] = y - floorf(y);
  }
  __syncthreads();
  const T* __restrict__ f1 = fmap1 + ((int64_t)b * Np + ii[e]) * C * PP;
  const T* __restrict__ f2 = fmap2 + (int64_t)b * s_b + jj[e] * s_n;
  for (int o = tid; o < PP * D * D; o += NT) {
    int p = o / (D * D), a = (o / D) % D, c = o % D;
    int gy = s_oy[p] + a, gx = s_ox[p] + c;
    float s = 0.0f;
    if (gy >= 0 && gy < H2 && gx >= 0 && gx < W2) {
      const T* src = f2 + (int64_t)gy * s_h + (int64_t)gx * s_w;
      for (int k = 0; k < C; k++) s = fmaf(to_f32<T>(f1[k * PP + p]), to_f32<T>(src[(int64_t)k * s_c]), s);
    }
    s_raw[o] = s;
  }
  __syncthreads();
  corr_epilogue<T>(s_raw, s_dx, s_dy, out + (int64_t)be * out_estride + out_offset, D, out_lstride);
}

// -------------------------------------------------------------------------------------------------
// Backward (fp32).  One workgroup per edge, ONE CHANNEL PER LANE: for every box position the lane reads
// its channel of fmap2 (coalesced when channels-last), updates 9 register accumulators of d_fmap1 and
// emits ONE atomic per (position, channel) into d_fmap2 — consecutive lanes hit consecutive addresses.
// The reference issues 2*C scalar atomics per (pixel, tap) thread (correlation_kernel.cu:182-188).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void corr_bwd_kernel(
    const float* __restrict__ fmap1, const float* __restrict__ fmap2, const float* __restrict__ coords,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, const float* __restrict__ grad,
    float* __restrict__ d1, float* __restrict__ d2, int E, int Np, int n2, int C, int H2, int W2, int64_t s_b,
    int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int R) {
  __shared__ float s_g[PP * MAXD * MAXD];   // gradient of the raw D x D windows (correlation_kernel.cu:259-269)
  __shared__ float s_dx[PP], s_dy[PP];
  __shared__ int s_ox[PP], s_oy[PP];
  const int D = 2 * R + 2, Dm = D - 1;
  const int be = blockIdx.x;
  const int b = be / E, e = be % E;
  const int tid = threadIdx.x;
  if (tid < PP) {
    float x = coords[((int64_t)be * 2 + 0) * PP + tid];
    float y = coords[((int64_t)be * 2 + 1) * PP + tid];
    s_ox[tid] = floor_to_int(x) - R;
    s_oy[tid] = floor_to_int(y) - R;
    s_dx[tid] = x - floorf(x);
    s_dy[tid] = y - floorf(y);
  }
  __syncthreads();
  const float* g = grad + (int64_t)be * Dm * Dm * PP;       // logical [c][a][i0][j0]
  for (int o = tid; o < PP * D * D; o += NT) {
    int p = o / (D * D), a = (o / D) % D, c = o % D;
    float dx = s_dx[p], dy = s_dy[p], s = 0.0f;
    auto G = [&](int aa, int cc) -> float {
      return (aa >= 0 && aa < Dm && cc >= 0 && cc < Dm) ? g[((int64_t)cc * Dm + aa) * PP + p] : 0.0f;
    };
    s += (1.0f - dx) * (1.0f - dy) * G(a, c);
    s += dx * (1.0f - dy) * G(a, c - 1);
    s += (1.0f - dx) * dy * G(a - 1, c);
    s += dx * dy * G(a - 1, c - 1);
    int gy = s_oy[p] + a, gx = s_ox[p] + c;
    if (!(gy >= 0 && gy < H2 && gx >= 0 && gx < W2)) s = 0.0f;   // out-of-bounds taps contribute nothing (:182)
    s_g[o] = s;
  }
  __syncthreads();

  int xmin = s_ox[0], xmax = s_ox[0], ymin = s_oy[0], ymax = s_oy[0];
#pragma unroll
  for (int p = 1; p < PP; p++) {
    xmin = min(xmin, s_ox[p]); xmax = max(xmax, s_ox[p]);
    ymin = min(ymin, s_oy[p]); ymax = max(ymax, s_oy[p]);
  }
  // clip the box to the frame: positions outside carry zero gradient
  const int x0 = max(xmin, 0), x1 = min(xmax + D, W2), y0 = max(ymin, 0), y1 = min(ymax + D, H2);
  const int64_t pi = ii[e], fj = jj[e];
  const float* __restrict__ f1 = fmap1 + ((int64_t)b * Np + pi) * C * PP;
  const float* __restrict__ f2 = fmap2 + (int64_t)b * s_b + fj * s_n;
  float* g1 = d1 + ((int64_t)b * Np + pi) * C * PP;
  float* g2 = d2 + (int64_t)b * s_b + fj * s_n;

  for (int k = tid; k < C; k += NT) {
    float w[PP], acc[PP];
#pragma unroll
    for (int p = 0; p < PP; p++) { w[p] = f1[k * PP + p]; acc[p] = 0.0f; }
    for (int gy = y0; gy < y1; gy++) {
      for (int gx = x0; gx < x1; gx++) {
        float gv[PP];
        bool any = false;
#pragma unroll
        for (int p = 0; p < PP; p++) {
          int a = gy - s_oy[p], c = gx - s_ox[p];
          gv[p] = (a >= 0 && a < D && c >= 0 && c < D) ? s_g[p * D * D + a * D + c] : 0.0f;
          any |= (gv[p] != 0.0f);
        }
        if (!any) continue;                                       // wave-uniform: gv depends only on the position
        int64_t off = (int64_t)gy * s_h + (int64_t)gx * s_w + (int64_t)k * s_c;
        float v = f2[off], t = 0.0f;
#pragma unroll
        for (int p = 0; p < PP; p++) { acc[p] = fmaf(gv[p], v, acc[p]); t = fmaf(gv[p], w[p], t); }
        atomicAdd(g2 + off, t);
      }
    }
#pragma unroll
    for (int p = 0; p < PP; p++) atomicAdd(g1 + k * PP + p, acc[p]);
  }
}

// -------------------------------------------------------------------------------------------------
// patchify (correlation_kernel.cu:16-80): integer-offset gather of (2R+2)^2 windows, one lane per output.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ void patchify_fwd_kernel(const T* __restrict__ net, const float* __restrict__ coords, T* __restrict__ out,
                                    int M, int C, int H, int W, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int R,
                                    int64_t total) {
  const int D = 2 * R + 2;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < total; n += (int64_t)blockDim.x * gridDim.x) {
    int c = n % D; int64_t r = n / D;
    int a = r % D; r /= D;
    int k = r % C; r /= C;
    int m = r % M; int b = r / M;
    float x = coords[((int64_t)b * M + m) * 2], y = coords[((int64_t)b * M + m) * 2 + 1];
    int i = floor_to_int(y) + a - R, j = floor_to_int(x) + c - R;
    T v = from_f32<T>(0.0f);
    if (i >= 0 && i < H && j >= 0 && j < W) v = net[b * sb + k * sc + i * sh + j * sw];
    out[n] = v;
  }
}

template <typename T>
__global__ void patchify_bwd_kernel(const float* __restrict__ coords, const T* __restrict__ grad, T* __restrict__ dnet,
                                    int M, int C, int H, int W, int R, int64_t total) {
  const int D = 2 * R + 2;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < total; n += (int64_t)blockDim.x * gridDim.x) {
    int c = n % D; int64_t r = n / D;
    int a = r % D; r /= D;
    int k = r % C; r /= C;
    int m = r % M; int b = r / M;
    float x = coords[((int64_t)b * M + m) * 2], y = coords[((int64_t)b * M + m) * 2 + 1];
    int i = floor_to_int(y) + a - R, j = floor_to_int(x) + c - R;
    if (i >= 0 && i < H && j >= 0 && j < W) atomicAdd(dnet + (((int64_t)b * C + k) * H + i) * W + j, grad[n]);
  }
}

}  // namespace devo

using namespace devo;

template <typename T>
static int launch_corr_fwd(const void* fmap1, const void* fmap2, const float* coords, const int64_t* ii,
                           const int64_t* jj, void* out, int B, int E, int Np, int n2, int C, int H2, int W2,
                           const int64_t* f2s, int64_t oes, int64_t ols, int64_t ooff, int R, hipStream_t st) {
  const bool cl = (f2s[2] == 1) && (C % KC == 0) && (f2s[3] % (16 / sizeof(T)) == 0) && (f2s[4] % (16 / sizeof(T)) == 0) &&
                  (f2s[0] % (16 / sizeof(T)) == 0) && (f2s[1] % (16 / sizeof(T)) == 0) &&
                  ((reinterpret_cast<uintptr_t>(fmap2) & 15) == 0) && sizeof(T) <= 4;
  dim3 grid((unsigned)((long long)B * E)), block(NT);
  if (cl) {
    hipLaunchKernelGGL(corr_fwd_cl_kernel<T>, grid, block, 0, st, (const T*)fmap1, (const T*)fmap2, coords, ii, jj,
                       (T*)out, E, Np, n2, C, H2, W2, f2s[0], f2s[1], f2s[3], f2s[4], oes, ols, ooff, R);
  } else {
    hipLaunchKernelGGL(corr_fwd_generic_kernel<T>, grid, block, 0, st, (const T*)fmap1, (const T*)fmap2, coords, ii,
                       jj, (T*)out, E, Np, n2, C, H2, W2, f2s[0], f2s[1], f2s[2], f2s[3], f2s[4], oes, ols, ooff, R);
  }
  return check_launch("devo_corr_forward");
}

extern "C" {

int devo_corr_forward(const void* fmap1, const void* fmap2, const float* coords, const int64_t* ii,
                      const int64_t* jj, void* out, int B, int E, int Np, int n2, int C, int P, int H2, int W2,
                      const int64_t* f2s, int64_t out_estride, int64_t out_lstride, int64_t out_offset, int radius,
                      int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(P == 3, "devo_corr_forward: patch size P must be 3 (got %d)", P);
  DEVO_REQUIRE(radius >= 0 && 2 * radius + 2 <= MAXD, "devo_corr_forward: radius %d unsupported (max 5)", radius);
  DEVO_REQUIRE(B >= 0 && E >= 0 && C > 0 && H2 > 0 && W2 > 0, "devo_corr_forward: bad sizes");
  DEVO_REQUIRE(f2s != nullptr, "devo_corr_forward: fmap2 strides missing");
  if ((long long)B * E == 0) return DEVO_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case DEVO_F32: return launch_corr_fwd<float>(fmap1, fmap2, coords, ii, jj, out, B, E, Np, n2, C, H2, W2, f2s, out_estride, out_lstride, out_offset, radius, st);
    case DEVO_F16: return launch_corr_fwd<__half>(fmap1, fmap2, coords, ii, jj, out, B, E, Np, n2, C, H2, W2, f2s, out_estride, out_lstride, out_offset, radius, st);
    case DEVO_F64: return launch_corr_fwd<double>(fmap1, fmap2, coords, ii, jj, out, B, E, Np, n2, C, H2, W2, f2s, out_estride, out_lstride, out_offset, radius, st);
  }
  set_error("devo_corr_forward: unknown dtype %d", dtype);
  return DEVO_ERR_UNSUPPORTED;
}

int devo_corr_backward(const void* fmap1, const void* fmap2, const float* coords, const int64_t* ii,
                       const int64_t* jj, const float* grad, void* fmap1_grad, void* fmap2_grad, int B, int E, int Np,
                       int n2, int C, int P, int H2, int W2, const int64_t* f2s, int64_t f2_numel_span, int radius,
                       int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(P == 3, "devo_corr_backward: patch size P must be 3 (got %d)", P);
  DEVO_REQUIRE(radius >= 0 && 2 * radius + 2 <= MAXD, "devo_corr_backward: radius %d unsupported (max 5)", radius);
  if (dtype != DEVO_F32) { set_error("devo_corr_backward: fp32 only (the reference's grad accessor is float)"); return DEVO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(fmap1_grad, 0, sizeof(float) * (size_t)B * Np * C * PP, st) != hipSuccess ||
      hipMemsetAsync(fmap2_grad, 0, sizeof(float) * (size_t)f2_numel_span, st) != hipSuccess) {
    set_error("devo_corr_backward: memset failed");
    return DEVO_ERR_LAUNCH;
  }
  if ((long long)B * E == 0) return DEVO_OK;
  hipLaunchKernelGGL(corr_bwd_kernel, dim3((unsigned)((long long)B * E)), dim3(NT), 0, st, (const float*)fmap1,
                     (const float*)fmap2, coords, ii, jj, grad, (float*)fmap1_grad, (float*)fmap2_grad, E, Np, n2, C, H2,
                     W2, f2s[0], f2s[1], f2s[2], f2s[3], f2s[4], radius);
  return check_launch("devo_corr_backward");
}

int devo_patchify_forward(const void* net, const float* coords, void* out, int B, int M, int C, int H, int W,
                          const int64_t* ns, int radius, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(radius >= 0 && ns != nullptr, "devo_patchify_forward: bad arguments");
  const int D = 2 * radius + 2;
  int64_t total = (int64_t)B * M * C * D * D;
  if (total == 0) return DEVO_OK;
  hipStream_t st = (hipStream_t)stream;
  int blocks = blocks_for(total, 256, 8192);
#define PF(T) hipLaunchKernelGGL(patchify_fwd_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)net, coords, (T*)out, M, C, H, W, ns[0], ns[1], ns[2], ns[3], radius, total)
  if (dtype == DEVO_F32) PF(float); else if (dtype == DEVO_F16) PF(__half); else if (dtype == DEVO_F64) PF(double);
  else { set_error("devo_patchify_forward: unknown dtype %d", dtype); return DEVO_ERR_UNSUPPORTED; }
#undef PF
  return check_launch("devo_patchify_forward");
}

int devo_patchify_backward(const float* coords, const void* grad, void* net_grad, int B, int M, int C, int H, int W,
                           int radius, int dtype, devo_stream_t stream) {
  const int D = 2 * radius + 2;
  int64_t total = (int64_t)B * M * C * D * D;
  hipStream_t st = (hipStream_t)stream;
  size_t esz = dtype == DEVO_F64 ? 8 : 4;
  if (dtype != DEVO_F32 && dtype != DEVO_F64) { set_error("devo_patchify_backward: F32/F64 only"); return DEVO_ERR_UNSUPPORTED; }
  if (hipMemsetAsync(net_grad, 0, esz * (size_t)B * C * H * W, st) != hipSuccess) { set_error("devo_patchify_backward: memset failed"); return DEVO_ERR_LAUNCH; }
  if (total == 0) return DEVO_OK;
  int blocks = blocks_for(total, 256, 8192);
  if (dtype == DEVO_F32) hipLaunchKernelGGL(patchify_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, coords, (const float*)grad, (float*)net_grad, M, C, H, W, radius, total);
  else hipLaunchKernelGGL(patchify_bwd_kernel<double>, dim3(blocks), dim3(256), 0, st, coords, (const double*)grad, (double*)net_grad, M, C, H, W, radius, total);
  return check_launch("devo_patchify_backward");
}

}  // extern "C"
