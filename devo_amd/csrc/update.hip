// Update operator (devo/enet.py:32-99, devo/blocks.py:15-48): the element-wise / reduction
// pieces between the GEMMs, for gfx950.  The dense layers themselves are plain library GEMMs (hipBLASLt through
// torch.nn.functional.linear); what is written here are the ops the reference spreads over ~40 small ATen /
// torch_scatter launches:
//   * row LayerNorm (eps 1e-3) with up to two fused residual inputs and an optional fused ReLU        (enet.py:47,53-55,65,83)
//   * masked neighbour gather  net[:, ix] * (ix >= 0)                                                (enet.py:86-91)
//   * SoftAgg: per-group softmax over the edges + weighted sum, one pass over HBM per operand        (blocks.py:42-43)
//   * expand-and-add of the aggregated rows back onto the edges  net += h(y)[:, group]               (blocks.py:46, enet.py:93-94)
//   * gated residual  x + sigmoid(g) * r                                                             (blocks.py:28-29)
//   * the two 2-wide heads: ReLU -> Linear(dim, 2) (-> Sigmoid)                                      (enet.py:68-78,96-98)
// All kernels: fp32 or fp16 storage (DEVO runs the update under autocast), fp32 arithmetic, one wave per row where a
// row reduction is needed (dim = 384 -> 6 elements per lane), no atomics, fixed summation order.
#include "common.h"
#include <hip/hip_fp16.h>
#include <initializer_list>

namespace devo {

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<__half>(const __half* p) { return __half2float(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<__half>(__half* p, float v) { *p = __float2half(v); }

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

constexpr int UPD_MAXPER = 16;      // elements of a row per lane: dim <= 1024

template <typename T> struct Vec2;
template <> struct Vec2<float> { typedef float2 type; };
template <> struct Vec2<__half> { typedef __half2 type; };
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2(const __half* p) { return __half22float2(*reinterpret_cast<const __half2*>(p)); }
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void st2(__half* p, float2 v) { *reinterpret_cast<__half2*>(p) = __float22half2_rn(v); }

// out[r] = LN(x[r] + a[r] + b[r]) * gamma + beta  (a, b optional), optionally ReLU'd.
// One wave per row; two-pass mean / variance in registers (the row is read once).  PAIRS: the row is walked in
// 128-element strips, every lane owning 2 adjacent elements of a strip (8- / 4-byte accesses); otherwise scalars.
// Extra input terms (any may be absent): hy[group_of[row]] (SoftAgg expand, blocks.py:46) and sigmoid(gate[row]) * res[row]
// (GatedResidual, blocks.py:28-29) — so that  net + agg(net)  and  x + gate * res  never make a round trip to HBM
// before the LayerNorm that follows them (enet.py:52-57).
template <typename T, bool PAIRS>
__global__ __launch_bounds__(256) void k_layernorm(const T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ b,
                                                   const T* __restrict__ hy, const int* __restrict__ group_of,
                                                   const T* __restrict__ gate, int64_t ld_gate, const T* __restrict__ res,
                                                   const T* __restrict__ gamma, const T* __restrict__ beta,
                                                   T* __restrict__ out, int64_t rows, int dim, float eps, int relu) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  constexpr int W = PAIRS ? 2 : 1;
  const int per = (dim + 64 * W - 1) / (64 * W);
  const T* hrow = hy ? hy + (int64_t)group_of[row] * dim : nullptr;
  float v[UPD_MAXPER];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < UPD_MAXPER / W; k++) {
    const int c = (lane + 64 * k) * W;
    if (k < per && c < dim) {
      if (PAIRS) {
        float2 t = ld2(x + row * dim + c);
        if (a) { const float2 u = ld2(a + row * dim + c); t.x += u.x; t.y += u.y; }
        if (b) { const float2 u = ld2(b + row * dim + c); t.x += u.x; t.y += u.y; }
        if (hrow) { const float2 u = ld2(hrow + c); t.x += u.x; t.y += u.y; }
        if (gate) {
          const float2 gv = ld2(gate + row * ld_gate + c), rv = ld2(res + row * dim + c);
          t.x += rv.x / (1.0f + __expf(-gv.x)); t.y += rv.y / (1.0f + __expf(-gv.y));
        }
        v[2 * k] = t.x; v[2 * k + 1] = t.y;
        s += t.x + t.y;
      } else {
        float t = ld(x + row * dim + c);
        if (a) t += ld(a + row * dim + c);
        if (b) t += ld(b + row * dim + c);
        if (hrow) t += ld(hrow + c);
        if (gate) t += ld(res + row * dim + c) / (1.0f + __expf(-ld(gate + row * ld_gate + c)));
        v[k] = t;
        s += t;
      }
    }
  }
  const float mean = wsum(s) / (float)dim;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < UPD_MAXPER / W; k++) {
    const int c = (lane + 64 * k) * W;
    if (k < per && c < dim) {
#pragma unroll
      for (int u = 0; u < W; u++) { const float d = v[W * k + u] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wsum(q) / (float)dim + eps);           // biased variance, like torch.nn.LayerNorm
#pragma unroll
  for (int k = 0; k < UPD_MAXPER / W; k++) {
    const int c = (lane + 64 * k) * W;
    if (k < per && c < dim) {
      if (PAIRS) {
        const float2 gm = ld2(gamma + c), bt = ld2(beta + c);
        float2 o = make_float2((v[2 * k] - mean) * rstd * gm.x + bt.x, (v[2 * k + 1] - mean) * rstd * gm.y + bt.y);
        if (relu) { o.x = fmaxf(o.x, 0.0f); o.y = fmaxf(o.y, 0.0f); }
        st2(out + row * dim + c, o);
      } else {
        float o = (v[k] - mean) * rstd * ld(gamma + c) + ld(beta + c);
        if (relu) o = fmaxf(o, 0.0f);
        st(out + row * dim + c, o);
      }
    }
  }
}

// out[e] = idx[e] >= 0 ? src[idx[e]] : 0
template <typename T>
__global__ void k_masked_gather(const T* __restrict__ src, const int64_t* __restrict__ idx, T* __restrict__ out, int64_t E, int dim) {
  const int64_t n = E * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / dim;
    const int c = (int)(i - e * dim);
    const int64_t j = idx[e];
    st(out + i, j >= 0 ? ld(src + j * dim + c) : 0.0f);
  }
}

// adjoint of out = x + sigmoid(gate) * res with respect to gate and res (d x = d out):  d gate = d out * res * s (1 - s),  d res = d out * s
template <typename T>
__global__ void k_gated_residual_bwd(const T* __restrict__ gate, int64_t ld_gate, const T* __restrict__ res, const T* __restrict__ dout,
                                     T* __restrict__ dgate, T* __restrict__ dres, int64_t rows, int dim) {
  const int64_t n = rows * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / dim;
    const int c = (int)(i - r * dim);
    const float sg = 1.0f / (1.0f + __expf(-ld(gate + r * ld_gate + c)));
    const float d = ld(dout + i);
    st(dgate + i, d * ld(res + i) * sg * (1.0f - sg));
    st(dres + i, d * sg);
  }
}

// SoftAgg reduction: for group s (edges perm[seg[s] .. seg[s+1])) and channel c
//   y[s][c] = sum_e f[e][c] * exp(g[e][c] - max_e g[e][c]) / sum_e exp(g[e][c] - max)
// ONE pass over the group's rows (online softmax: the running sums are rescaled when the maximum grows), one thread per
// channel pair, workgroups stride over the groups (n_seg lives on the device).  f and g have row stride `ld_fg` (they
// may be the two halves of one [E, 2 dim] GEMM output).  group_of[e] = s is written for the expand step.
template <typename T>
__global__ __launch_bounds__(256) void k_softagg(const T* __restrict__ f, const T* __restrict__ g, int64_t ld_fg,
                                                 const int* __restrict__ perm, const int* __restrict__ seg,
                                                 const int* __restrict__ n_seg_p, T* __restrict__ y,
                                                 int* __restrict__ group_of, int dim) {
  const int n_seg = *n_seg_p;
  const int nslab = (dim + 511) / 512;
  for (int w = blockIdx.x; w < n_seg * nslab; w += gridDim.x) {
    const int s = w / nslab, c = ((w - s * nslab) * 256 + threadIdx.x) * 2;
    const int a0 = seg[s], a1 = seg[s + 1];
    if (c < dim) {                                           // dim is even (checked by the launcher)
      float m0 = -3.0e38f, m1 = -3.0e38f, den0 = 0.0f, den1 = 0.0f, num0 = 0.0f, num1 = 0.0f;
      for (int a = a0; a < a1; a++) {
        const int64_t e = perm[a];
        const float2 gv = ld2(g + e * ld_fg + c), fv = ld2(f + e * ld_fg + c);
        const float n0 = fmaxf(m0, gv.x), n1 = fmaxf(m1, gv.y);
        const float r0 = __expf(m0 - n0), r1 = __expf(m1 - n1), w0 = __expf(gv.x - n0), w1 = __expf(gv.y - n1);
        den0 = den0 * r0 + w0; num0 = num0 * r0 + fv.x * w0; m0 = n0;
        den1 = den1 * r1 + w1; num1 = num1 * r1 + fv.y * w1; m1 = n1;
      }
      st2(y + (int64_t)s * dim + c, make_float2(num0 / den0, num1 / den1));
    }
    if (group_of && (w - s * nslab) == 0)
      for (int a = a0 + threadIdx.x; a < a1; a += 256) group_of[perm[a]] = s;
  }
}

// Adjoint of k_softagg (training): with w_e = softmax over the group of g, y = sum_e f_e w_e,
//   d f_e = w_e * dy,   d g_e = w_e * dy * (f_e - y)       (channel-wise; sum_e' w_e' f_e' = y removes the cross term)
// Same work distribution; pass 1 recomputes the group's max / normaliser / y online, pass 2 writes the two gradients.
template <typename T>
__global__ __launch_bounds__(256) void k_softagg_bwd(const T* __restrict__ f, const T* __restrict__ g, int64_t ld_fg,
                                                     const int* __restrict__ perm, const int* __restrict__ seg,
                                                     const int* __restrict__ n_seg_p, const T* __restrict__ dy,
                                                     T* __restrict__ df, T* __restrict__ dg, int64_t ld_d, int dim) {
  const int n_seg = *n_seg_p;
  const int nslab = (dim + 511) / 512;
  for (int w = blockIdx.x; w < n_seg * nslab; w += gridDim.x) {
    const int s = w / nslab, c = ((w - s * nslab) * 256 + threadIdx.x) * 2;
    const int a0 = seg[s], a1 = seg[s + 1];
    if (c >= dim) continue;
    float m0 = -3.0e38f, m1 = -3.0e38f, den0 = 0.0f, den1 = 0.0f, num0 = 0.0f, num1 = 0.0f;
    for (int a = a0; a < a1; a++) {
      const int64_t e = perm[a];
      const float2 gv = ld2(g + e * ld_fg + c), fv = ld2(f + e * ld_fg + c);
      const float n0 = fmaxf(m0, gv.x), n1 = fmaxf(m1, gv.y);
      const float r0 = __expf(m0 - n0), r1 = __expf(m1 - n1), w0 = __expf(gv.x - n0), w1 = __expf(gv.y - n1);
      den0 = den0 * r0 + w0; num0 = num0 * r0 + fv.x * w0; m0 = n0;
      den1 = den1 * r1 + w1; num1 = num1 * r1 + fv.y * w1; m1 = n1;
    }
    const float i0 = 1.0f / den0, i1 = 1.0f / den1, y0 = num0 * i0, y1 = num1 * i1;
    const float2 dyv = ld2(dy + (int64_t)s * dim + c);
    for (int a = a0; a < a1; a++) {
      const int64_t e = perm[a];
      const float2 gv = ld2(g + e * ld_fg + c), fv = ld2(f + e * ld_fg + c);
      const float w0 = __expf(gv.x - m0) * i0 * dyv.x, w1 = __expf(gv.y - m1) * i1 * dyv.y;
      st2(df + e * ld_d + c, make_float2(w0, w1));
      st2(dg + e * ld_d + c, make_float2(w0 * (fv.x - y0), w1 * (fv.y - y1)));
    }
  }
}

// net[e] += hy[group_of[e]]
template <typename T>
__global__ void k_expand_add(T* __restrict__ net, const T* __restrict__ hy, const int* __restrict__ group_of, int64_t E, int dim) {
  const int64_t n = E * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / dim;
    const int c = (int)(i - e * dim);
    st(net + i, ld(net + i) + ld(hy + (int64_t)group_of[e] * dim + c));
  }
}

// out = x + sigmoid(gate) * res      (gate rows have stride ld_gate: it may be a column block of a wider GEMM output)
template <typename T>
__global__ void k_gated_residual(const T* __restrict__ x, const T* __restrict__ gate, int64_t ld_gate, const T* __restrict__ res,
                                 T* __restrict__ out, int64_t rows, int dim) {
  const int64_t n = rows * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / dim;
    const int c = (int)(i - r * dim);
    const float gt = 1.0f / (1.0f + __expf(-ld(gate + r * ld_gate + c)));
    st(out + i, ld(x + i) + gt * ld(res + i));
  }
}

// net[e] = x[e] + sigmoid(gate[e]) * res[e] (gate == nullptr: net = x as given, nothing stored), then
// delta[e] = Wd relu(net[e]) + bd ; weight[e] = sigmoid(Ww relu(net[e]) + bw)   (Wd, Ww: [2, dim]); one wave per edge
template <typename T>
__global__ __launch_bounds__(256) void k_heads(const T* __restrict__ x, const T* __restrict__ gate, int64_t ld_gate,
                                               const T* __restrict__ res, T* __restrict__ net_out, const T* __restrict__ Wd,
                                               const T* __restrict__ bd, const T* __restrict__ Ww, const T* __restrict__ bw,
                                               T* __restrict__ delta, T* __restrict__ weight, int64_t E, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  float d0 = 0.0f, d1 = 0.0f, w0 = 0.0f, w1 = 0.0f;
  for (int c = lane; c < dim; c += 64) {
    float nv = ld(x + e * dim + c);
    if (gate) {
      nv += ld(res + e * dim + c) / (1.0f + __expf(-ld(gate + e * ld_gate + c)));
      st(net_out + e * dim + c, nv);
      nv = ld(net_out + e * dim + c);                       // the heads see the stored (rounded) value, like separate kernels
    }
    const float v = fmaxf(nv, 0.0f);
    d0 += v * ld(Wd + c); d1 += v * ld(Wd + dim + c);
    w0 += v * ld(Ww + c); w1 += v * ld(Ww + dim + c);
  }
  d0 = wsum(d0); d1 = wsum(d1); w0 = wsum(w0); w1 = wsum(w1);
  if (lane == 0) {
    st(delta + e * 2, d0 + ld(bd)); st(delta + e * 2 + 1, d1 + ld(bd + 1));
    st(weight + e * 2, 1.0f / (1.0f + __expf(-(w0 + ld(bw)))));
    st(weight + e * 2 + 1, 1.0f / (1.0f + __expf(-(w1 + ld(bw + 1)))));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 16-byte forms of the row-wise kernels (dim a multiple of 4 floats / 8 halves, 16-byte aligned operands — the Update
// operator's 384-wide rows): one 16-byte access per lane instead of a 2- / 4-byte one, 32-bit index arithmetic.  Same
// element-wise arithmetic as the scalar forms above (which stay as the fallback for odd shapes).
template <typename T> struct ChunkOf { static constexpr int V = 16 / (int)sizeof(T); };
__device__ __forceinline__ void ldc(const float* p, float (&v)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ldc(const __half* p, float (&v)[8]) {
  const uint4 t = *reinterpret_cast<const uint4*>(p);
  const __half2* h = reinterpret_cast<const __half2*>(&t);
#pragma unroll
  for (int u = 0; u < 4; u++) { const float2 f = __half22float2(h[u]); v[2 * u] = f.x; v[2 * u + 1] = f.y; }
}
__device__ __forceinline__ void stc(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void stc(__half* p, const float (&v)[8]) {
  uint4 t;
  __half2* h = reinterpret_cast<__half2*>(&t);
#pragma unroll
  for (int u = 0; u < 4; u++) h[u] = __float22half2_rn(make_float2(v[2 * u], v[2 * u + 1]));
  *reinterpret_cast<uint4*>(p) = t;
}
template <typename T> __device__ __forceinline__ float round_as(float v);           // the value a store + load of type T gives back
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_as<__half>(float v) { return __half2float(__float2half(v)); }

// k_layernorm with 16-byte chunks (up to two per lane: dim <= 512 floats / 1024 halves); same fused input terms
template <typename T>
__global__ __launch_bounds__(256) void k_layernorm_v(const T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ b,
                                                     const T* __restrict__ hy, const int* __restrict__ group_of,
                                                     const T* __restrict__ gate, int64_t ld_gate, const T* __restrict__ res,
                                                     const T* __restrict__ gamma, const T* __restrict__ beta,
                                                     T* __restrict__ out, int64_t rows, int dim, float eps, int relu, int cpr) {
  constexpr int V = ChunkOf<T>::V;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* hrow = hy ? hy + (int64_t)group_of[row] * dim : nullptr;
  float v[2][V];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int c = (lane + 64 * k) * V;
    if (lane + 64 * k < cpr) {
      float t[V], u[V];
      ldc(x + row * dim + c, t);
      if (a) { ldc(a + row * dim + c, u);
#pragma unroll
        for (int i = 0; i < V; i++) t[i] += u[i]; }
      if (b) { ldc(b + row * dim + c, u);
#pragma unroll
        for (int i = 0; i < V; i++) t[i] += u[i]; }
      if (hrow) { ldc(hrow + c, u);
#pragma unroll
        for (int i = 0; i < V; i++) t[i] += u[i]; }
      if (gate) {
        float gv[V];
        ldc(gate + row * ld_gate + c, gv); ldc(res + row * dim + c, u);
#pragma unroll
        for (int i = 0; i < V; i++) t[i] += u[i] / (1.0f + __expf(-gv[i]));
      }
#pragma unroll
      for (int i = 0; i < V; i++) { v[k][i] = t[i]; s += t[i]; }
    }
  }
  const float mean = wsum(s) / (float)dim;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < 2; k++)
    if (lane + 64 * k < cpr) {
#pragma unroll
      for (int i = 0; i < V; i++) { const float d = v[k][i] - mean; q += d * d; }
    }
  const float rstd = rsqrtf(wsum(q) / (float)dim + eps);           // biased variance, like torch.nn.LayerNorm
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int c = (lane + 64 * k) * V;
    if (lane + 64 * k < cpr) {
      float gm[V], bt[V], o[V];
      ldc(gamma + c, gm); ldc(beta + c, bt);
#pragma unroll
      for (int i = 0; i < V; i++) {
        o[i] = (v[k][i] - mean) * rstd * gm[i] + bt[i];
        if (relu) o[i] = fmaxf(o[i], 0.0f);
      }
      stc(out + row * dim + c, o);
    }
  }
}

// Sum over the 16 lanes of a DPP row, result in all of them (row_ror 8, 4, 2, 1): no LDS round trips.
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
  return v;
}

// k_layernorm with a QUARTER wave per row (16 lanes x NC 16-byte chunks, cpr = 16 NC <= 128): four rows per wave, the two row
// reductions inside a DPP row.  A whole wave per 768-byte row spends its time in two 6-step shuffle chains.
template <typename T, int NC>
__global__ __launch_bounds__(256) void k_layernorm_q(const T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ b,
                                                     const T* __restrict__ hy, const int* __restrict__ group_of,
                                                     const T* __restrict__ gate, int64_t ld_gate, const T* __restrict__ res,
                                                     const T* __restrict__ gamma, const T* __restrict__ beta,
                                                     T* __restrict__ out, int64_t rows, int dim, float eps, int relu) {
  constexpr int V = ChunkOf<T>::V;
  const int l16 = threadIdx.x & 15;
  const int64_t row_raw = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = row_raw < rows;
  const int64_t row = live ? row_raw : rows - 1;               // (all lanes stay active for the DPP sums)
  const T* hrow = hy ? hy + (int64_t)group_of[row] * dim : nullptr;
  float v[NC][V];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < NC; k++) {
    const int c = (l16 + 16 * k) * V;
    float t[V], u[V];
    ldc(x + row * dim + c, t);
    if (a) { ldc(a + row * dim + c, u);
#pragma unroll
      for (int i = 0; i < V; i++) t[i] += u[i]; }
    if (b) { ldc(b + row * dim + c, u);
#pragma unroll
      for (int i = 0; i < V; i++) t[i] += u[i]; }
    if (hrow) { ldc(hrow + c, u);
#pragma unroll
      for (int i = 0; i < V; i++) t[i] += u[i]; }
    if (gate) {
      float gv[V];
      ldc(gate + row * ld_gate + c, gv); ldc(res + row * dim + c, u);
#pragma unroll
      for (int i = 0; i < V; i++) t[i] += u[i] / (1.0f + __expf(-gv[i]));
    }
#pragma unroll
    for (int i = 0; i < V; i++) { v[k][i] = t[i]; s += t[i]; }
  }
  const float mean = row16_sum(s) / (float)dim;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < NC; k++)
#pragma unroll
    for (int i = 0; i < V; i++) { const float d = v[k][i] - mean; q += d * d; }
  const float rstd = rsqrtf(row16_sum(q) / (float)dim + eps);      // biased variance, like torch.nn.LayerNorm
  if (!live) return;
#pragma unroll
  for (int k = 0; k < NC; k++) {
    const int c = (l16 + 16 * k) * V;
    float gm[V], bt[V], o[V];
    ldc(gamma + c, gm); ldc(beta + c, bt);
#pragma unroll
    for (int i = 0; i < V; i++) {
      o[i] = (v[k][i] - mean) * rstd * gm[i] + bt[i];
      if (relu) o[i] = fmaxf(o[i], 0.0f);
    }
    stc(out + row * dim + c, o);
  }
}

// Adjoint of k_layernorm_q (fp32, training): y = LN(x + a + b) [ReLU'd], g = dL/dy  ->  dx (= da = db) and the column sums
// dgamma += sum_rows g xhat, dbeta += sum_rows g.  A quarter wave per row as in the forward (mean and rstd are recomputed: the row is in
// registers anyway); every lane keeps the partial column sums of its 4 NC columns over its rows, the workgroup folds its 16 quarter waves
// through LDS and adds 2 dim values to global memory (devo/enet.py:44,52-56,62: the LayerNorms the reference differentiates through
// torch.autograd; ATen runs layer_norm_grad_input + cuComputePartGradGammaBeta + a reduction).
template <int NC>
__global__ __launch_bounds__(256) void k_layernorm_bwd_q(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ dout, float* __restrict__ dx,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int dim, float eps,
                                                         int relu, float* __restrict__ partials) {
  extern __shared__ float lnb_part[];                      // [16 quarter waves][2 dim]
  const int l16 = threadIdx.x & 15, qw = threadIdx.x >> 4;
  float gm[NC][4], bt[NC][4], ag[NC][4], ab[NC][4];
#pragma unroll
  for (int k = 0; k < NC; k++) {
    const int c = (l16 + 16 * k) * 4;
    ldc(gamma + c, gm[k]); ldc(beta + c, bt[k]);
#pragma unroll
    for (int i = 0; i < 4; i++) { ag[k][i] = 0.0f; ab[k][i] = 0.0f; }
  }
  const float inv_dim = 1.0f / (float)dim;
  const int64_t stride = (int64_t)gridDim.x * 16;
  const int64_t first = (int64_t)blockIdx.x * 16 + qw;
  const int64_t iters = (rows + stride - 1) / stride;      // every quarter wave runs the same number of rounds (the DPP sums need all lanes)
  for (int64_t it = 0; it < iters; it++) {
    const int64_t row_raw = first + it * stride;
    const bool live = row_raw < rows;
    const int64_t row = live ? row_raw : rows - 1;
    float v[NC][4], g[NC][4];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NC; k++) {
      const int c = (l16 + 16 * k) * 4;
      float u[4];
      ldc(x + row * dim + c, v[k]);
      if (a) { ldc(a + row * dim + c, u);
#pragma unroll
        for (int i = 0; i < 4; i++) v[k][i] += u[i]; }
      if (b) { ldc(b + row * dim + c, u);
#pragma unroll
        for (int i = 0; i < 4; i++) v[k][i] += u[i]; }
      ldc(dout + row * dim + c, g[k]);
#pragma unroll
      for (int i = 0; i < 4; i++) s += v[k][i];
    }
    const float mean = row16_sum(s) * inv_dim;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < NC; k++)
#pragma unroll
      for (int i = 0; i < 4; i++) { const float d = v[k][i] - mean; q += d * d; }
    const float rstd = rsqrtf(row16_sum(q) * inv_dim + eps);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < NC; k++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float xh = (v[k][i] - mean) * rstd;
        float gi = g[k][i];
        if (relu && !(xh * gm[k][i] + bt[k][i] > 0.0f)) gi = 0.0f;      // the forward's max(., 0): no gradient where it clipped
        if (!live) gi = 0.0f;
        ag[k][i] += gi * xh; ab[k][i] += gi;
        const float gg = gi * gm[k][i];
        s1 += gg; s2 += gg * xh;
        v[k][i] = xh; g[k][i] = gg;
      }
    s1 = row16_sum(s1) * inv_dim; s2 = row16_sum(s2) * inv_dim;
    if (live) {
#pragma unroll
      for (int k = 0; k < NC; k++) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = rstd * (g[k][i] - s1 - v[k][i] * s2);
        stc(dx + row * dim + (l16 + 16 * k) * 4, o);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NC; k++) {
    const int c = (l16 + 16 * k) * 4;
    *reinterpret_cast<float4*>(lnb_part + qw * 2 * dim + c) = make_float4(ag[k][0], ag[k][1], ag[k][2], ag[k][3]);
    *reinterpret_cast<float4*>(lnb_part + qw * 2 * dim + dim + c) = make_float4(ab[k][0], ab[k][1], ab[k][2], ab[k][3]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * dim; c += 256) {
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; w++) t += lnb_part[w * 2 * dim + c];
    if (partials) partials[(size_t)blockIdx.x * 2 * dim + c] = t;      // deterministic form: k_layernorm_colsum adds the workgroups' rows in order
    else atomicAdd(c < dim ? dgamma + c : dbeta + (c - dim), t);
  }
}
// dgamma | dbeta += the workgroups' partial column sums, in workgroup order (one thread per column: run-to-run identical bits)
__global__ __launch_bounds__(256) void k_layernorm_colsum(const float* __restrict__ partials, int nwg, int dim, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 2 * dim) return;
  float t = 0.0f;
  for (int w = 0; w < nwg; w++) t += partials[(size_t)w * 2 * dim + c];
  float* d = c < dim ? dgamma + c : dbeta + (c - dim);
  *d += t;
}

template <typename T>
__global__ void k_masked_gather_v(const T* __restrict__ src, const int64_t* __restrict__ idx, T* __restrict__ out, unsigned total, unsigned cpr) {
  constexpr int V = ChunkOf<T>::V;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned e = i / cpr, q = i - e * cpr;
    const int64_t j = idx[e];
    uint4 t = make_uint4(0u, 0u, 0u, 0u);
    if (j >= 0) t = *reinterpret_cast<const uint4*>(src + (j * cpr + q) * V);      // a copy: no conversion
    *reinterpret_cast<uint4*>(out + ((int64_t)e * cpr + q) * V) = t;
  }
}

template <typename T>
__global__ void k_expand_add_v(T* __restrict__ net, const T* __restrict__ hy, const int* __restrict__ group_of, unsigned total, unsigned cpr) {
  constexpr int V = ChunkOf<T>::V;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned e = i / cpr, q = i - e * cpr;
    float a[V], b[V];
    ldc(net + (int64_t)i * V, a);
    ldc(hy + ((int64_t)group_of[e] * cpr + q) * V, b);
#pragma unroll
    for (int u = 0; u < V; u++) a[u] += b[u];
    stc(net + (int64_t)i * V, a);
  }
}

template <typename T>
__global__ void k_gated_residual_v(const T* __restrict__ x, const T* __restrict__ gate, int64_t ld_gate, const T* __restrict__ res,
                                   T* __restrict__ out, unsigned total, unsigned cpr) {
  constexpr int V = ChunkOf<T>::V;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned r = i / cpr, q = i - r * cpr;
    float xv[V], gv[V], rv[V];
    ldc(x + (int64_t)i * V, xv); ldc(gate + (int64_t)r * ld_gate + q * V, gv); ldc(res + (int64_t)i * V, rv);
#pragma unroll
    for (int u = 0; u < V; u++) xv[u] += (1.0f / (1.0f + __expf(-gv[u]))) * rv[u];
    stc(out + (int64_t)i * V, xv);
  }
}

template <typename T>
__global__ void k_gated_residual_bwd_v(const T* __restrict__ gate, int64_t ld_gate, const T* __restrict__ res, const T* __restrict__ dout,
                                       T* __restrict__ dgate, T* __restrict__ dres, unsigned total, unsigned cpr) {
  constexpr int V = ChunkOf<T>::V;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned r = i / cpr, q = i - r * cpr;
    float gv[V], rv[V], dv[V], og[V], orr[V];
    ldc(gate + (int64_t)r * ld_gate + q * V, gv); ldc(res + (int64_t)i * V, rv); ldc(dout + (int64_t)i * V, dv);
#pragma unroll
    for (int u = 0; u < V; u++) {
      const float sg = 1.0f / (1.0f + __expf(-gv[u]));
      og[u] = dv[u] * rv[u] * sg * (1.0f - sg);
      orr[u] = dv[u] * sg;
    }
    stc(dgate + (int64_t)i * V, og); stc(dres + (int64_t)i * V, orr);
  }
}

// k_heads with one 16-byte chunk per lane (cpr <= 64: one wave covers a row)
template <typename T>
__global__ __launch_bounds__(256) void k_heads_v(const T* __restrict__ x, const T* __restrict__ gate, int64_t ld_gate,
                                                 const T* __restrict__ res, T* __restrict__ net_out, const T* __restrict__ Wd,
                                                 const T* __restrict__ bd, const T* __restrict__ Ww, const T* __restrict__ bw,
                                                 T* __restrict__ delta, T* __restrict__ weight, int64_t E, int dim, int cpr) {
  constexpr int V = ChunkOf<T>::V;
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  float d0 = 0.0f, d1 = 0.0f, w0 = 0.0f, w1 = 0.0f;
  if (lane < cpr) {
    const int c = lane * V;
    float nv[V];
    ldc(x + e * dim + c, nv);
    if (gate) {
      float gv[V], rv[V];
      ldc(gate + e * ld_gate + c, gv); ldc(res + e * dim + c, rv);
#pragma unroll
      for (int u = 0; u < V; u++) nv[u] += rv[u] / (1.0f + __expf(-gv[u]));
      stc(net_out + e * dim + c, nv);
#pragma unroll
      for (int u = 0; u < V; u++) nv[u] = round_as<T>(nv[u]);          // the heads see the stored (rounded) value, like separate kernels
    }
    float a0[V], a1[V], b0[V], b1[V];
    ldc(Wd + c, a0); ldc(Wd + dim + c, a1); ldc(Ww + c, b0); ldc(Ww + dim + c, b1);
#pragma unroll
    for (int u = 0; u < V; u++) {
      const float v = fmaxf(nv[u], 0.0f);
      d0 += v * a0[u]; d1 += v * a1[u]; w0 += v * b0[u]; w1 += v * b1[u];
    }
  }
  d0 = wsum(d0); d1 = wsum(d1); w0 = wsum(w0); w1 = wsum(w1);
  if (lane == 0) {
    st(delta + e * 2, d0 + ld(bd)); st(delta + e * 2 + 1, d1 + ld(bd + 1));
    st(weight + e * 2, 1.0f / (1.0f + __expf(-(w0 + ld(bw)))));
    st(weight + e * 2 + 1, 1.0f / (1.0f + __expf(-(w1 + ld(bw + 1)))));
  }
}

// k_heads with a quarter wave per edge (16 lanes x NC chunks), the four dot products summed inside a DPP row
template <typename T, int NC>
__global__ __launch_bounds__(256) void k_heads_q(const T* __restrict__ x, const T* __restrict__ gate, int64_t ld_gate,
                                                 const T* __restrict__ res, T* __restrict__ net_out, const T* __restrict__ Wd,
                                                 const T* __restrict__ bd, const T* __restrict__ Ww, const T* __restrict__ bw,
                                                 T* __restrict__ delta, T* __restrict__ weight, int64_t E, int dim) {
  constexpr int V = ChunkOf<T>::V;
  const int l16 = threadIdx.x & 15;
  const int64_t e_raw = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = e_raw < E;
  const int64_t e = live ? e_raw : E - 1;
  float d0 = 0.0f, d1 = 0.0f, w0 = 0.0f, w1 = 0.0f;
#pragma unroll
  for (int k = 0; k < NC; k++) {
    const int c = (l16 + 16 * k) * V;
    float nv[V];
    ldc(x + e * dim + c, nv);
    if (gate) {
      float gv[V], rv[V];
      ldc(gate + e * ld_gate + c, gv); ldc(res + e * dim + c, rv);
#pragma unroll
      for (int u = 0; u < V; u++) nv[u] += rv[u] / (1.0f + __expf(-gv[u]));
      if (live) stc(net_out + e * dim + c, nv);
#pragma unroll
      for (int u = 0; u < V; u++) nv[u] = round_as<T>(nv[u]);          // the heads see the stored (rounded) value, like separate kernels
    }
    float a0[V], a1[V], b0[V], b1[V];
    ldc(Wd + c, a0); ldc(Wd + dim + c, a1); ldc(Ww + c, b0); ldc(Ww + dim + c, b1);
#pragma unroll
    for (int u = 0; u < V; u++) {
      const float v = fmaxf(nv[u], 0.0f);
      d0 += v * a0[u]; d1 += v * a1[u]; w0 += v * b0[u]; w1 += v * b1[u];
    }
  }
  d0 = row16_sum(d0); d1 = row16_sum(d1); w0 = row16_sum(w0); w1 = row16_sum(w1);
  if (live && l16 == 0) {
    st(delta + e * 2, d0 + ld(bd)); st(delta + e * 2 + 1, d1 + ld(bd + 1));
    st(weight + e * 2, 1.0f / (1.0f + __expf(-(w0 + ld(bw)))));
    st(weight + e * 2 + 1, 1.0f / (1.0f + __expf(-(w1 + ld(bw + 1)))));
  }
}

// k_softagg with the rows of a group dealt to the four waves of the workgroup (a frame-pair group has ~100 rows: one thread
// per channel walking them one after the other is a chain of ~100 dependent row latencies), 16-byte chunks per lane (NQ <= 2
// chunks: dim <= 512 floats / 1024 halves), two rows in flight per wave; the four partial (max, normaliser, sum) triples are
// merged through LDS.  Dynamic LDS: 12 * dim floats.
template <typename T, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_softagg_v(const T* __restrict__ f, const T* __restrict__ g, int64_t ld_fg,
                                                   const int* __restrict__ perm, const int* __restrict__ seg,
                                                   const int* __restrict__ n_seg_p, T* __restrict__ y,
                                                   int* __restrict__ group_of, int dim, int cpr) {
  constexpr int V = ChunkOf<T>::V;
  extern __shared__ float s_part[];                          // [3][NW][dim]: max, normaliser, weighted sum of every wave
  const int n_seg = *n_seg_p;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = blockIdx.x; s < n_seg; s += gridDim.x) {
    const int a0 = seg[s], a1 = seg[s + 1];
    float m[2][V], den[2][V], num[2][V];
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
      for (int u = 0; u < V; u++) { m[k][u] = -3.0e38f; den[k][u] = 0.0f; num[k][u] = 0.0f; }
    auto fold = [&](int k, const float (&gv)[V], const float (&fv)[V]) {
#pragma unroll
      for (int u = 0; u < V; u++) {
        const float n = fmaxf(m[k][u], gv[u]);
        const float r = __expf(m[k][u] - n), w = __expf(gv[u] - n);
        den[k][u] = den[k][u] * r + w; num[k][u] = num[k][u] * r + fv[u] * w; m[k][u] = n;
      }
    };
    constexpr int RF = 2;                                      // rows in flight per wave: a, a + NW (4 measured slower: 17.0 / 14.3 us against 14.9 / 13.8)
    for (int a = a0 + wave; a < a1; a += RF * NW) {
      int64_t er[RF];
      bool have[RF];
#pragma unroll
      for (int r = 0; r < RF; r++) { have[r] = a + r * NW < a1; er[r] = perm[have[r] ? a + r * NW : a]; }
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int q = lane + 64 * k;
        if (q < cpr) {
          float gr[RF][V], fr[RF][V];
#pragma unroll
          for (int r = 0; r < RF; r++) { ldc(g + er[r] * ld_fg + q * V, gr[r]); ldc(f + er[r] * ld_fg + q * V, fr[r]); }
#pragma unroll
          for (int r = 0; r < RF; r++)
            if (have[r]) fold(k, gr[r], fr[r]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int q = lane + 64 * k;
      if (q < cpr) {
#pragma unroll
        for (int u = 0; u < V; u++) {
          const int c = q * V + u;
          s_part[(0 * NW + wave) * dim + c] = m[k][u];
          s_part[(1 * NW + wave) * dim + c] = den[k][u];
          s_part[(2 * NW + wave) * dim + c] = num[k][u];
        }
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += 64 * NW) {
      float M = s_part[c];
#pragma unroll
      for (int w = 1; w < NW; w++) M = fmaxf(M, s_part[w * dim + c]);
      float dn = 0.0f, nm = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const float r = __expf(s_part[w * dim + c] - M);
        dn += s_part[(NW + w) * dim + c] * r; nm += s_part[(2 * NW + w) * dim + c] * r;
      }
      st(y + (int64_t)s * dim + c, nm / dn);
    }
    if (group_of)
      for (int a = a0 + threadIdx.x; a < a1; a += 64 * NW) group_of[perm[a]] = s;
    __syncthreads();                                         // s_part is reused by the next group
  }
}

// Small groups (the edges of a patch: 15 at cfg2, ~21 in DEVO's sliding window): ONE WAVE per group, no LDS, no barrier — the group's rows are
// requested together (batches of RB rows: every lane holds its 16-byte chunk of each, raw), the batch's maximum is exact, one exponential per
// element (the online form of k_softagg_v pays two per row and element, and its four waves meet at two barriers around an LDS merge for 15
// rows); batches of a longer group are merged online.  Deterministic: rows in the order of perm.  Lanes beyond the row's chunks idle.
template <typename T>
__global__ __launch_bounds__(256) void k_softagg_w(const T* __restrict__ f, const T* __restrict__ g, int64_t ld_fg, const int* __restrict__ perm,
                                                   const int* __restrict__ seg, const int* __restrict__ n_seg_p, T* __restrict__ y,
                                                   int* __restrict__ group_of, int dim, int cpr) {
  constexpr int V = ChunkOf<T>::V, RB = sizeof(T) == 2 ? 16 : 8, NQ = sizeof(T) == 2 ? 1 : 2;   // (dim <= 512 halves / 512 floats: cpr <= 64 NQ)
  typedef uint4 raw_t;
  const int n_seg = *n_seg_p;
  const int lane = threadIdx.x & 63;
  const int w0 = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), nw = (int)gridDim.x * 4;
  for (int s = w0; s < n_seg; s += nw) {
    const int a0 = seg[s], a1 = seg[s + 1];
    float m[NQ][V], den[NQ][V], num[NQ][V];
#pragma unroll
    for (int k = 0; k < NQ; k++)
#pragma unroll
      for (int u = 0; u < V; u++) { m[k][u] = -3.0e38f; den[k][u] = 0.0f; num[k][u] = 0.0f; }
    for (int a = a0; a < a1; a += RB) {
      const int nr = min(RB, a1 - a);                              // wave-uniform
      const int mine = (lane < nr) ? perm[a + lane] : 0;
      if (group_of && lane < nr) group_of[mine] = s;
      raw_t gr[NQ][RB], fr[NQ][RB];
#pragma unroll
      for (int r = 0; r < RB; r++) {
        const int64_t e = (int64_t)__builtin_amdgcn_readlane(mine, r < nr ? r : 0);      // (rows beyond the batch repeat its first: loaded, not used)
#pragma unroll
        for (int k = 0; k < NQ; k++) {
          const int q = lane + 64 * k;
          if (q < cpr) {
            gr[k][r] = *reinterpret_cast<const raw_t*>(g + e * ld_fg + q * V);
            fr[k][r] = *reinterpret_cast<const raw_t*>(f + e * ld_fg + q * V);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int q = lane + 64 * k;
        if (q < cpr) {
          float bm[V];
#pragma unroll
          for (int u = 0; u < V; u++) bm[u] = m[k][u];
#pragma unroll
          for (int r = 0; r < RB; r++) {
            if (r < nr) {                                          // (converted again below: conversions are cheaper than 128 more registers)
              float gv[V];
              ldc(reinterpret_cast<const T*>(&gr[k][r]), gv);
#pragma unroll
              for (int u = 0; u < V; u++) bm[u] = fmaxf(bm[u], gv[u]);
            }
          }
#pragma unroll
          for (int u = 0; u < V; u++) {
            const float sc = __expf(m[k][u] - bm[u]);              // (first batch: exp(-3e38 - max) = 0 on zeros)
            den[k][u] *= sc; num[k][u] *= sc; m[k][u] = bm[u];
          }
#pragma unroll
          for (int r = 0; r < RB; r++) {
            if (r < nr) {
              float fv[V], gv[V];
              raw_t gt = gr[k][r];
              asm volatile("" : "+v"(gt.x), "+v"(gt.y), "+v"(gt.z), "+v"(gt.w));      // (opaque: or the compiler keeps the first pass's 128 converted values alive)
              ldc(reinterpret_cast<const T*>(&gt), gv);
              ldc(reinterpret_cast<const T*>(&fr[k][r]), fv);
#pragma unroll
              for (int u = 0; u < V; u++) {
                const float w = __expf(gv[u] - bm[u]);
                den[k][u] += w; num[k][u] += fv[u] * w;
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NQ; k++) {
      const int q = lane + 64 * k;
      if (q < cpr) {
        float o[V];
#pragma unroll
        for (int u = 0; u < V; u++) o[u] = num[k][u] / den[k][u];
        stc(y + (int64_t)s * dim + q * V, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- instance norm, channels-last
// The encoders' InstanceNorm2d (no affine, no running statistics: extractor.py:27-38; devo_amd/patchifier.py) on a channels-last activation
// [N, H, W, C]: y = relu((x - mean_nc) * rstd_nc) and, for the tail of a residual block (extractor.py:48-54), relu(res + y).  Through ATen one
// norm of an NHWC tensor is a copy to NCHW, a statistics kernel, a transform kernel, (a copy back) and the ReLU — 10 norms per frame, 0.6 of the
// encoders' 1.25 ms at 480 x 640 (tools/probe_patchifier_frame.py).  Here: a partial-sums launch (shifted by the image's first pixel: no
// cancellation for activations with a large mean) and an apply launch that adds the partials up in a fixed order (deterministic).
constexpr int IN_PART = 64;                                            // pixel chunks per image
// bias (optional, [C]): the convolution's bias, added here as ATen adds it behind a bias-free MIOpen call — v = round(x + b) — so that the
// caller can leave the bias out of the convolution (one elementwise launch per convolution less)
template <typename T>
__global__ __launch_bounds__(256) void k_instnorm_stats(const T* __restrict__ x, const T* __restrict__ bias, int HW, int C, float* __restrict__ part) {
  constexpr int V = ChunkOf<T>::V;
  const int cpr = C / V, n = blockIdx.y, p = blockIdx.x;
  const int cv = threadIdx.x % cpr, pl = threadIdx.x / cpr, npl = 256 / cpr;           // (threads beyond npl * cpr idle)
  const int per = (HW + IN_PART - 1) / IN_PART, i0 = p * per, i1 = min(HW, i0 + per);
  const T* xn = x + (int64_t)n * HW * C;
  float k[V], s1[V], s2[V], bv[V];
#pragma unroll
  for (int u = 0; u < V; u++) bv[u] = 0.f;
  if (bias) ldc(bias + cv * V, bv);
  ldc(xn + cv * V, k);                                                                   // the shift: this image's first pixel
#pragma unroll
  for (int u = 0; u < V; u++) { k[u] = round_as<T>(k[u] + bv[u]); s1[u] = 0.f; s2[u] = 0.f; }
  if (pl < npl)
    for (int i = i0 + pl; i < i1; i += npl) {
      float v[V];
      ldc(xn + (int64_t)i * C + cv * V, v);
#pragma unroll
      for (int u = 0; u < V; u++) { const float d = round_as<T>(v[u] + bv[u]) - k[u]; s1[u] += d; s2[u] += d * d; }
    }
  extern __shared__ float in_lds[];                                                      // [npl][2][C]
  if (pl < npl) {
#pragma unroll
    for (int u = 0; u < V; u++) { in_lds[(pl * 2) * C + cv * V + u] = s1[u]; in_lds[(pl * 2 + 1) * C + cv * V + u] = s2[u]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {                                       // fixed order over the pixel lanes
    float a = 0.f;
    const int which = c / C, ch = c - which * C;
    for (int q = 0; q < npl; q++) a += in_lds[(q * 2 + which) * C + ch];
    part[(((int64_t)n * IN_PART + p) * 2 + which) * C + ch] = a;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_instnorm_apply(const T* __restrict__ x, const T* __restrict__ bias, const T* __restrict__ res, T* __restrict__ y,
                                                        int HW, int C, const float* __restrict__ part, float eps, int relu) {
  constexpr int V = ChunkOf<T>::V;
  extern __shared__ float in_lds[];                                                      // mean [C] | rstd [C]
  const int n = blockIdx.y;
  const T* xn = x + (int64_t)n * HW * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};   // (four chains, sixteen loads in flight: a fixed order all the same)
#pragma unroll 2
    for (int p = 0; p < IN_PART; p += 8) {
      float va[8], vb[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { va[j] = part[(((int64_t)n * IN_PART + p + j) * 2) * C + c]; vb[j] = part[(((int64_t)n * IN_PART + p + j) * 2 + 1) * C + c]; }
#pragma unroll
      for (int j = 0; j < 8; j++) { a4[j & 3] += va[j]; b4[j & 3] += vb[j]; }
    }
    const float a = (a4[0] + a4[1]) + (a4[2] + a4[3]), b = (b4[0] + b4[1]) + (b4[2] + b4[3]);
    float k1[1];
    k1[0] = 0.f;
    {                                                                                    // the shift again (one element)
      const T* q = xn + c;
      k1[0] = round_as<T>((float)(*q) + (bias ? (float)bias[c] : 0.f));
    }
    const float m = a / (float)HW, var = fmaxf(b / (float)HW - m * m, 0.f);
    in_lds[c] = k1[0] + m;
    in_lds[C + c] = rsqrtf(var + eps);
  }
  __syncthreads();
  const int cpr = C / V;
  const int64_t total = (int64_t)HW * cpr;
  const T* rn = res ? res + (int64_t)n * HW * C : nullptr;
  T* yn = y + (int64_t)n * HW * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % cpr);
    float v[V], o[V];
    ldc(xn + i * V, v);
    if (bias) {
      float bv[V];
      ldc(bias + cv * V, bv);
#pragma unroll
      for (int u = 0; u < V; u++) v[u] = round_as<T>(v[u] + bv[u]);
    }
#pragma unroll
    for (int u = 0; u < V; u++) {
      float t = (v[u] - in_lds[cv * V + u]) * in_lds[C + cv * V + u];
      t = round_as<T>(t);                                                                // (the norm's output as ATen stores it, then the ReLU on that)
      o[u] = relu ? fmaxf(t, 0.f) : t;
    }
    if (rn) {
      float r[V];
      ldc(rn + i * V, r);
#pragma unroll
      for (int u = 0; u < V; u++) o[u] = fmaxf(round_as<T>(r[u] + o[u]), 0.f);           // relu(res + y): the residual block's tail
    }
    stc(yn + i * V, o);
  }
}

// y = act(round(x + bias)) and, with res, relu(round(res + y)) on a channels-last activation: what ATen runs as a bias add, a ReLU, a sum and a ReLU
// behind a convolution of the context encoder (no norm there: extractor.py:27-54 with norm_fn = 'none'), one launch
template <typename T>
__global__ __launch_bounds__(256) void k_bias_act_cl(const T* __restrict__ x, const T* __restrict__ bias, const T* __restrict__ res, T* __restrict__ y,
                                                     int64_t total, int cpr, int relu) {
  constexpr int V = ChunkOf<T>::V;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % cpr);
    float v[V], o[V];
    ldc(x + i * V, v);
    if (bias) {
      float bv[V];
      ldc(bias + cv * V, bv);
#pragma unroll
      for (int u = 0; u < V; u++) v[u] = round_as<T>(v[u] + bv[u]);
    }
#pragma unroll
    for (int u = 0; u < V; u++) o[u] = relu ? fmaxf(v[u], 0.f) : v[u];
    if (res) {
      float r[V];
      ldc(res + i * V, r);
#pragma unroll
      for (int u = 0; u < V; u++) o[u] = fmaxf(round_as<T>(r[u] + o[u]), 0.f);
    }
    stc(y + i * V, o);
  }
}

static unsigned grid_for(long long n, int per_block, int cap) {
  long long b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

}  // namespace devo

using namespace devo;

#define UPD_DISPATCH(DT, CALL_F32, CALL_F16)                                                  \
  do {                                                                                        \
    if ((DT) == DEVO_F32) { CALL_F32; }                                                       \
    else if ((DT) == DEVO_F16) { CALL_F16; }                                                  \
    else { set_error("update ops: fp32 / fp16 only (dtype %d)", (int)(DT)); return DEVO_ERR_UNSUPPORTED; } \
  } while (0)

// 16-byte forms apply when every row starts on a 16-byte boundary and the element count stays in 32 bits
static bool upd_vec_ok(int dtype, int64_t rows, int dim, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds = {}) {
  const int V = dtype == DEVO_F32 ? 4 : 8;
  if (dim % V || rows * (dim / V) >= (1LL << 31)) return false;
  for (const void* p : ptrs) if (reinterpret_cast<uintptr_t>(p) & 15) return false;
  for (int64_t l : lds) if (l % V) return false;
  return true;
}

extern "C" {

int devo_upd_layernorm(const void* x, const void* add1, const void* add2, const void* hy, const int* group_of, const void* gate,
                       int64_t ld_gate, const void* res, const void* gamma, const void* beta, void* out, int64_t rows, int dim,
                       float eps, int relu, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(rows >= 0 && dim > 0 && dim <= 64 * UPD_MAXPER, "devo_upd_layernorm: dim %d unsupported (1..%d)", dim, 64 * UPD_MAXPER);
  DEVO_REQUIRE((hy == nullptr) == (group_of == nullptr) && (gate == nullptr) == (res == nullptr) && (gate == nullptr || ld_gate >= dim),
               "devo_upd_layernorm: hy needs group_of, gate needs res");
  if (rows == 0) return DEVO_OK;
  hipStream_t st_ = (hipStream_t)stream;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  auto al8 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
  const bool pairs = (dim % 2 == 0) && (ld_gate % 2 == 0) && al8(x) && al8(add1) && al8(add2) && al8(hy) && al8(gate) && al8(res) &&
                     al8(gamma) && al8(beta) && al8(out);
#define LN_ARGS(TT) (const TT*)x, (const TT*)add1, (const TT*)add2, (const TT*)hy, group_of, (const TT*)gate, ld_gate, (const TT*)res, \
                    (const TT*)gamma, (const TT*)beta, (TT*)out, rows, dim, eps, relu
  const bool vec = upd_vec_ok(dtype, rows, dim, {x, add1, add2, hy, gate, res, gamma, beta, out}, {gate ? ld_gate : 0});
  if (vec && dtype == DEVO_F16 && dim == 384) {               // the Update operator's rows: quarter wave per row
    hipLaunchKernelGGL((k_layernorm_q<__half, 3>), dim3((unsigned)((rows + 15) / 16)), block, 0, st_, LN_ARGS(__half));
  } else if (vec && dtype == DEVO_F32 && dim == 384) {
    hipLaunchKernelGGL((k_layernorm_q<float, 6>), dim3((unsigned)((rows + 15) / 16)), block, 0, st_, LN_ARGS(float));
  } else if (vec && dim / (dtype == DEVO_F32 ? 4 : 8) <= 128) {
    const int cpr = dim / (dtype == DEVO_F32 ? 4 : 8);
    UPD_DISPATCH(dtype,
      hipLaunchKernelGGL(k_layernorm_v<float>, grid, block, 0, st_, LN_ARGS(float), cpr),
      hipLaunchKernelGGL(k_layernorm_v<__half>, grid, block, 0, st_, LN_ARGS(__half), cpr));
  } else if (pairs) {
    UPD_DISPATCH(dtype,
      hipLaunchKernelGGL((k_layernorm<float, true>), grid, block, 0, st_, LN_ARGS(float)),
      hipLaunchKernelGGL((k_layernorm<__half, true>), grid, block, 0, st_, LN_ARGS(__half)));
  } else {
    UPD_DISPATCH(dtype,
      hipLaunchKernelGGL((k_layernorm<float, false>), grid, block, 0, st_, LN_ARGS(float)),
      hipLaunchKernelGGL((k_layernorm<__half, false>), grid, block, 0, st_, LN_ARGS(__half)));
  }
#undef LN_ARGS
  return check_launch("devo_upd_layernorm");
}

int devo_upd_layernorm_backward(const float* x, const float* add1, const float* add2, const float* gamma, const float* beta, const float* dout,
                                float* dx, float* dgamma, float* dbeta, int64_t rows, int dim, float eps, int relu, float* partials,
                                devo_stream_t stream) {
  DEVO_REQUIRE(rows >= 0 && dim == 384, "devo_upd_layernorm_backward: rows of 384 values (the update operator's; got %d)", dim);
  if (rows == 0) return DEVO_OK;
  DEVO_REQUIRE(x && gamma && beta && dout && dx && dgamma && dbeta, "devo_upd_layernorm_backward: null tensor");
  DEVO_REQUIRE(upd_vec_ok(DEVO_F32, rows, dim, {x, add1, add2, gamma, beta, dout, dx}), "devo_upd_layernorm_backward: tensors must be 16-byte aligned");
  const unsigned nwg = (unsigned)std::min<int64_t>((rows + 15) / 16, 512);
  hipLaunchKernelGGL((k_layernorm_bwd_q<6>), dim3(nwg), dim3(256), (size_t)16 * 2 * dim * sizeof(float), (hipStream_t)stream, x, add1, add2, gamma, beta,
                     dout, dx, dgamma, dbeta, rows, dim, eps, relu, partials);
  if (partials) hipLaunchKernelGGL(k_layernorm_colsum, dim3((unsigned)((2 * dim + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)partials, (int)nwg, dim, dgamma, dbeta);
  return check_launch("devo_upd_layernorm_backward");
}

int devo_upd_masked_gather(const void* src, const int64_t* idx, void* out, int64_t E, int dim, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(E >= 0 && dim > 0, "devo_upd_masked_gather: bad sizes");
  if (E == 0) return DEVO_OK;
  hipStream_t st_ = (hipStream_t)stream;
  if (upd_vec_ok(dtype, E, dim, {src, out})) {
    const unsigned cpr = (unsigned)(dim / (dtype == DEVO_F32 ? 4 : 8)), total = (unsigned)(E * cpr);
    const dim3 vgrid(grid_for(total, 256, 16384)), vblock(256);
    UPD_DISPATCH(dtype,
      hipLaunchKernelGGL(k_masked_gather_v<float>, vgrid, vblock, 0, st_, (const float*)src, idx, (float*)out, total, cpr),
      hipLaunchKernelGGL(k_masked_gather_v<__half>, vgrid, vblock, 0, st_, (const __half*)src, idx, (__half*)out, total, cpr));
    return check_launch("devo_upd_masked_gather");
  }
  const dim3 grid(grid_for(E * dim, 256, 8192)), block(256);
  UPD_DISPATCH(dtype,
    hipLaunchKernelGGL(k_masked_gather<float>, grid, block, 0, st_, (const float*)src, idx, (float*)out, E, dim),
    hipLaunchKernelGGL(k_masked_gather<__half>, grid, block, 0, st_, (const __half*)src, idx, (__half*)out, E, dim));
  return check_launch("devo_upd_masked_gather");
}

size_t devo_instnorm_workspace_bytes(int N, int C) { return N > 0 && C > 0 ? (size_t)N * IN_PART * 2 * C * sizeof(float) : 0; }

int devo_bias_act_cl(const void* x, const void* bias, const void* res, void* y, int64_t pixels, int C, int relu, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(x && y && pixels > 0 && C > 0, "devo_bias_act_cl: null argument or bad sizes");
  DEVO_REQUIRE(dtype == DEVO_F32 || dtype == DEVO_F16, "devo_bias_act_cl: fp32 / fp16 only (dtype %d)", dtype);
  const int V = dtype == DEVO_F32 ? 4 : 8;
  DEVO_REQUIRE(C % V == 0, "devo_bias_act_cl: C = %d must be a multiple of %d", C, V);
  DEVO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
               "devo_bias_act_cl: 16-byte alignment");
  DEVO_REQUIRE(!res || relu, "devo_bias_act_cl: the residual tail is relu(res + relu(x + bias))");
  const int cpr = C / V;
  const int64_t total = pixels * cpr;
  const dim3 g(grid_for(total, 256 * 4, 2048));
  UPD_DISPATCH(dtype,
    hipLaunchKernelGGL(k_bias_act_cl<float>, g, dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)bias, (const float*)res, (float*)y, total, cpr, relu),
    hipLaunchKernelGGL(k_bias_act_cl<__half>, g, dim3(256), 0, (hipStream_t)stream, (const __half*)x, (const __half*)bias, (const __half*)res, (__half*)y, total, cpr, relu));
  return check_launch("devo_bias_act_cl");
}

int devo_instnorm_cl(const void* x, const void* res, void* y, int N, int HW, int C, float eps, int relu, void* workspace, size_t ws_bytes, int dtype,
                     devo_stream_t stream) {
  return devo_instnorm_bias_cl(x, nullptr, res, y, N, HW, C, eps, relu, workspace, ws_bytes, dtype, stream);
}

int devo_instnorm_bias_cl(const void* x, const void* bias, const void* res, void* y, int N, int HW, int C, float eps, int relu, void* workspace, size_t ws_bytes,
                          int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(x && y && workspace && N > 0 && HW > 0 && C > 0, "devo_instnorm_cl: null argument or bad sizes");
  DEVO_REQUIRE(dtype == DEVO_F32 || dtype == DEVO_F16, "devo_instnorm_cl: fp32 / fp16 only (dtype %d)", dtype);
  const int V = dtype == DEVO_F32 ? 4 : 8;
  DEVO_REQUIRE(C % V == 0 && C / V <= 256 && 256 / (C / V) >= 1, "devo_instnorm_cl: C = %d must be a multiple of %d, at most %d", C, V, 256 * V);
  DEVO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
               "devo_instnorm_cl: 16-byte alignment");
  DEVO_REQUIRE(ws_bytes >= devo_instnorm_workspace_bytes(N, C), "devo_instnorm_cl: workspace %zu < %zu bytes", ws_bytes, devo_instnorm_workspace_bytes(N, C));
  DEVO_REQUIRE(!res || relu, "devo_instnorm_cl: the residual tail is relu(res + relu(norm))");
  hipStream_t st_ = (hipStream_t)stream;
  const int npl = 256 / (C / V);
  const size_t lds1 = sizeof(float) * (size_t)npl * 2 * C, lds2 = sizeof(float) * 2 * (size_t)C;
  const dim3 g1(IN_PART, (unsigned)N), g2(grid_for((long long)HW * (C / V), 256 * 4, 1024), (unsigned)N);
  UPD_DISPATCH(dtype,
    hipLaunchKernelGGL(k_instnorm_stats<float>, g1, dim3(256), lds1, st_, (const float*)x, (const float*)bias, HW, C, (float*)workspace),
    hipLaunchKernelGGL(k_instnorm_stats<__half>, g1, dim3(256), lds1, st_, (const __half*)x, (const __half*)bias, HW, C, (float*)workspace));
  UPD_DISPATCH(dtype,
    hipLaunchKernelGGL(k_instnorm_apply<float>, g2, dim3(256), lds2, st_, (const float*)x, (const float*)bias, (const float*)res, (float*)y, HW, C, (const float*)workspace, eps, relu),
    hipLaunchKernelGGL(k_instnorm_apply<__half>, g2, dim3(256), lds2, st_, (const __half*)x, (const __half*)bias, (const __half*)res, (__half*)y, HW, C, (const float*)workspace, eps, relu));
  return check_launch("devo_instnorm_cl");
}

int devo_upd_softagg(const void* f, const void* g, int64_t ld_fg, const int* perm, const int* seg_start, const int* n_seg,
                     void* y, int* group_of, int64_t E, int dim, int dtype, devo_stream_t stream) {
  return devo_upd_softagg_hint(f, g, ld_fg, perm, seg_start, n_seg, y, group_of, E, dim, dtype, 0, stream);
}

// rows_per_group: the caller's estimate (E / groups; 0 = unknown).  Groups of many rows (the frame pairs: ~100 edges each, ~200 groups) are
// a chain of dependent row latencies per wave: they get 16 waves per workgroup instead of 4.
int devo_upd_softagg_hint(const void* f, const void* g, int64_t ld_fg, const int* perm, const int* seg_start, const int* n_seg,
                          void* y, int* group_of, int64_t E, int dim, int dtype, int rows_per_group, devo_stream_t stream) {
  DEVO_REQUIRE(E >= 0 && dim > 0 && dim % 2 == 0 && ld_fg >= dim && ld_fg % 2 == 0, "devo_upd_softagg: bad sizes (dim and the row stride must be even)");
  DEVO_REQUIRE(((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y)) & 7) == 0, "devo_upd_softagg: operands must be 8-byte aligned");
  if (E == 0) return DEVO_OK;
  hipStream_t st_ = (hipStream_t)stream;
  if (upd_vec_ok(dtype, E, dim, {f, g}, {ld_fg}) && dim / (dtype == DEVO_F32 ? 4 : 8) <= 128) {
    const int cpr = dim / (dtype == DEVO_F32 ? 4 : 8);
    // one workgroup per group: the caller's estimate sizes the grid (a workgroup without a group still has to be placed, read the count and
    // retire: with E workgroups — 4 096 here — for 225 frame pairs the launch was eight rounds of empty 1 024-thread workgroups); the
    // kernels stride over the groups, so an estimate that is too low costs balance, not results
    const dim3 vgrid(grid_for(rows_per_group > 0 ? E / rows_per_group + 1 : E, 1, 4096));
    if (rows_per_group >= 48 && sizeof(float) * 48 * (size_t)dim <= 160 * 1024) {
      const size_t lds = sizeof(float) * 48 * (size_t)dim;
      static PerDeviceOnce attr_done;
      if (attr_done.first()) {
        DEVO_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_softagg_v<float, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(&k_softagg_v<__half, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                     "devo_upd_softagg: cannot raise the dynamic LDS limit");
      }
      UPD_DISPATCH(dtype,
        hipLaunchKernelGGL((k_softagg_v<float, 16>), vgrid, dim3(1024), lds, st_, (const float*)f, (const float*)g, ld_fg, perm, seg_start, n_seg, (float*)y, group_of, dim, cpr),
        hipLaunchKernelGGL((k_softagg_v<__half, 16>), vgrid, dim3(1024), lds, st_, (const __half*)f, (const __half*)g, ld_fg, perm, seg_start, n_seg, (__half*)y, group_of, dim, cpr));
      return check_launch("devo_upd_softagg");
    }
    static const bool wave_form = [] { const char* e = getenv("DEVO_UPD_SOFTAGG_WAVE"); return !(e && e[0] == '0'); }();
    if (wave_form && rows_per_group > 0 && rows_per_group <= 40 && cpr <= (dtype == DEVO_F32 ? 128 : 64)) {     // one wave per group (a patch's edges)
      const dim3 wgrid(grid_for((E / rows_per_group + 1 + 3) / 4, 1, 4096));
      UPD_DISPATCH(dtype,
        hipLaunchKernelGGL((k_softagg_w<float>), wgrid, dim3(256), 0, st_, (const float*)f, (const float*)g, ld_fg, perm, seg_start, n_seg, (float*)y, group_of, dim, cpr),
        hipLaunchKernelGGL((k_softagg_w<__half>), wgrid, dim3(256), 0, st_, (const __half*)f, (const __half*)g, ld_fg, perm, seg_start, n_seg, (__half*)y, group_of, dim, cpr));
      return check_launch("devo_upd_softagg");
    }
    const dim3 vblock(256);
    const size_t lds = sizeof(float) * 12 * (size_t)dim;
    UPD_DISPATCH(dtype,
      hipLaunchKernelGGL((k_softagg_v<float, 4>), vgrid, vblock, lds, st_, (const float*)f, (const float*)g, ld_fg, perm, seg_start, n_seg, (float*)y, group_of, dim, cpr),
      hipLaunchKernelGGL((k_softagg_v<__half, 4>), vgrid, vblock, lds, st_, (const __half*)f, (const __half*)g, ld_fg, perm, seg_start, n_seg, (__half*)y, group_of, dim, cpr));
    return check_launch("devo_upd_softagg");
  }
  const dim3 grid(grid_for(E * ((dim + 511) / 512), 1, 4096)), block(256);
  UPD_DISPATCH(dtype,
    hipLaunchKernelGGL(k_softagg<float>, grid, block, 0, st_, (const float*)f, (const float*)g, ld_fg, perm, seg_start, n_seg, (float*)y, group_of, dim),
    hipLaunchKernelGGL(k_softagg<__half>, grid, block, 0, st_, (const __half*)f, (const __half*)g, ld_fg, perm, seg_start, n_seg, (__half*)y, group_of, dim));
  return check_launch("devo_upd_softagg");
}

int devo_upd_softagg_backward(const void* f, const void* g, int64_t ld_fg, const int* perm, const int* seg_start, const int* n_seg,
                              const void* dy, void* df, void* dg, int64_t ld_d, int64_t E, int dim, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(E >= 0 && dim > 0 && dim % 2 == 0 && ld_fg >= dim && ld_fg % 2 == 0 && ld_d >= dim && ld_d % 2 == 0,
               "devo_upd_softagg_backward: bad sizes (dim and the row strides must be even)");
  DEVO_REQUIRE(((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(df) |
                 reinterpret_cast<uintptr_t>(dg)) & 7) == 0, "devo_upd_softagg_backward: operands must be 8-byte aligned");
  if (E == 0) return DEVO_OK;
  hipStream_t st_ = (hipStream_t)stream;
  const dim3 grid(grid_for(E * ((dim + 511) / 512), 1, 4096)), block(256);
  UPD_DISPATCH(dtype,
    hipLaunchKernelGGL(k_softagg_bwd<float>, grid, block, 0, st_, (const float*)f, (const float*)g, ld_fg, perm, seg_start, n_seg, (const float*)dy, (float*)df, (float*)dg, ld_d, dim),
    hipLaunchKernelGGL(k_softagg_bwd<__half>, grid, block, 0, st_, (const __half*)f, (const __half*)g, ld_fg, perm, seg_start, n_seg, (const __half*)dy, (__half*)df, (__half*)dg, ld_d, dim));
  return check_launch("devo_upd_softagg_backward");
}

int devo_upd_expand_add(void* net, const void* hy, const int* group_of, int64_t E, int dim, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(E >= 0 && dim > 0, "devo_upd_expand_add: bad sizes");
  if (E == 0) return DEVO_OK;
  hipStream_t st_ = (hipStream_t)stream;
  if (upd_vec_ok(dtype, E, dim, {net, hy})) {
    const unsigned cpr = (unsigned)(dim / (dtype == DEVO_F32 ? 4 : 8)), total = (unsigned)(E * cpr);
    const dim3 vgrid(grid_for(total, 256, 16384)), vblock(256);
    UPD_DISPATCH(dtype,
      hipLaunchKernelGGL(k_expand_add_v<float>, vgrid, vblock, 0, st_, (float*)net, (const float*)hy, group_of, total, cpr),
      hipLaunchKernelGGL(k_expand_add_v<__half>, vgrid, vblock, 0, st_, (__half*)net, (const __half*)hy, group_of, total, cpr));
    return check_launch("devo_upd_expand_add");
  }
  const dim3 grid(grid_for(E * dim, 256, 8192)), block(256);
  UPD_DISPATCH(dtype,
    hipLaunchKernelGGL(k_expand_add<float>, grid, block, 0, st_, (float*)net, (const float*)hy, group_of, E, dim),
    hipLaunchKernelGGL(k_expand_add<__half>, grid, block, 0, st_, (__half*)net, (const __half*)hy, group_of, E, dim));
  return check_launch("devo_upd_expand_add");
}

int devo_upd_gated_residual(const void* x, const void* gate, int64_t ld_gate, const void* res, void* out, int64_t rows, int dim,
                            int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(rows >= 0 && dim > 0 && ld_gate >= dim, "devo_upd_gated_residual: bad sizes");
  if (rows == 0) return DEVO_OK;
  hipStream_t st_ = (hipStream_t)stream;
  if (upd_vec_ok(dtype, rows, dim, {x, gate, res, out}, {ld_gate})) {
    const unsigned cpr = (unsigned)(dim / (dtype == DEVO_F32 ? 4 : 8)), total = (unsigned)(rows * cpr);
    const dim3 vgrid(grid_for(total, 256, 16384)), vblock(256);
    UPD_DISPATCH(dtype,
      hipLaunchKernelGGL(k_gated_residual_v<float>, vgrid, vblock, 0, st_, (const float*)x, (const float*)gate, ld_gate, (const float*)res, (float*)out, total, cpr),
      hipLaunchKernelGGL(k_gated_residual_v<__half>, vgrid, vblock, 0, st_, (const __half*)x, (const __half*)gate, ld_gate, (const __half*)res, (__half*)out, total, cpr));
    return check_launch("devo_upd_gated_residual");
  }
  const dim3 grid(grid_for(rows * dim, 256, 8192)), block(256);
  UPD_DISPATCH(dtype,
    hipLaunchKernelGGL(k_gated_residual<float>, grid, block, 0, st_, (const float*)x, (const float*)gate, ld_gate, (const float*)res, (float*)out, rows, dim),
    hipLaunchKernelGGL(k_gated_residual<__half>, grid, block, 0, st_, (const __half*)x, (const __half*)gate, ld_gate, (const __half*)res, (__half*)out, rows, dim));
  return check_launch("devo_upd_gated_residual");
}

int devo_upd_gated_residual_backward(const void* gate, int64_t ld_gate, const void* res, const void* dout, void* dgate, void* dres,
                                     int64_t rows, int dim, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(rows >= 0 && dim > 0 && ld_gate >= dim, "devo_upd_gated_residual_backward: bad sizes");
  if (rows == 0) return DEVO_OK;
  hipStream_t st_ = (hipStream_t)stream;
  if (upd_vec_ok(dtype, rows, dim, {gate, res, dout, dgate, dres}, {ld_gate})) {
    const unsigned cpr = (unsigned)(dim / (dtype == DEVO_F32 ? 4 : 8)), total = (unsigned)(rows * cpr);
    const dim3 vgrid(grid_for(total, 256, 16384)), vblock(256);
    UPD_DISPATCH(dtype,
      hipLaunchKernelGGL(k_gated_residual_bwd_v<float>, vgrid, vblock, 0, st_, (const float*)gate, ld_gate, (const float*)res, (const float*)dout, (float*)dgate, (float*)dres, total, cpr),
      hipLaunchKernelGGL(k_gated_residual_bwd_v<__half>, vgrid, vblock, 0, st_, (const __half*)gate, ld_gate, (const __half*)res, (const __half*)dout, (__half*)dgate, (__half*)dres, total, cpr));
    return check_launch("devo_upd_gated_residual_backward");
  }
  const dim3 grid(grid_for(rows * dim, 256, 8192)), block(256);
  UPD_DISPATCH(dtype,
    hipLaunchKernelGGL(k_gated_residual_bwd<float>, grid, block, 0, st_, (const float*)gate, ld_gate, (const float*)res, (const float*)dout, (float*)dgate, (float*)dres, rows, dim),
    hipLaunchKernelGGL(k_gated_residual_bwd<__half>, grid, block, 0, st_, (const __half*)gate, ld_gate, (const __half*)res, (const __half*)dout, (__half*)dgate, (__half*)dres, rows, dim));
  return check_launch("devo_upd_gated_residual_backward");
}

int devo_upd_heads(const void* x, const void* gate, int64_t ld_gate, const void* res, void* net_out, const void* Wd, const void* bd,
                   const void* Ww, const void* bw, void* delta, void* weight, int64_t E, int dim, int dtype, devo_stream_t stream) {
  DEVO_REQUIRE(E >= 0 && dim > 0, "devo_upd_heads: bad sizes");
  DEVO_REQUIRE(gate == nullptr || (res != nullptr && net_out != nullptr && ld_gate >= dim), "devo_upd_heads: gate needs res and net_out");
  if (E == 0) return DEVO_OK;
  hipStream_t st_ = (hipStream_t)stream;
  const dim3 grid((unsigned)((E + 3) / 4)), block(256);
  const bool hvec = upd_vec_ok(dtype, E, dim, {x, gate, res, net_out, Wd, Ww}, {gate ? ld_gate : 0});
#define HEADS_ARGS(TT) (const TT*)x, (const TT*)gate, ld_gate, (const TT*)res, (TT*)net_out, (const TT*)Wd, (const TT*)bd, (const TT*)Ww, (const TT*)bw, (TT*)delta, (TT*)weight, E, dim
  if (hvec && dim == 384 && (dtype == DEVO_F16 || dtype == DEVO_F32)) {          // the Update operator's rows: quarter wave per edge
    const dim3 qgrid((unsigned)((E + 15) / 16));
    if (dtype == DEVO_F16) hipLaunchKernelGGL((k_heads_q<__half, 3>), qgrid, block, 0, st_, HEADS_ARGS(__half));
    else hipLaunchKernelGGL((k_heads_q<float, 6>), qgrid, block, 0, st_, HEADS_ARGS(float));
    return check_launch("devo_upd_heads");
  }
#undef HEADS_ARGS
  if (hvec && dim / (dtype == DEVO_F32 ? 4 : 8) <= 64) {
    const int cpr = dim / (dtype == DEVO_F32 ? 4 : 8);
    UPD_DISPATCH(dtype,
      hipLaunchKernelGGL(k_heads_v<float>, grid, block, 0, st_, (const float*)x, (const float*)gate, ld_gate, (const float*)res, (float*)net_out, (const float*)Wd, (const float*)bd, (const float*)Ww, (const float*)bw, (float*)delta, (float*)weight, E, dim, cpr),
      hipLaunchKernelGGL(k_heads_v<__half>, grid, block, 0, st_, (const __half*)x, (const __half*)gate, ld_gate, (const __half*)res, (__half*)net_out, (const __half*)Wd, (const __half*)bd, (const __half*)Ww, (const __half*)bw, (__half*)delta, (__half*)weight, E, dim, cpr));
    return check_launch("devo_upd_heads");
  }
  UPD_DISPATCH(dtype,
    hipLaunchKernelGGL(k_heads<float>, grid, block, 0, st_, (const float*)x, (const float*)gate, ld_gate, (const float*)res, (float*)net_out, (const float*)Wd, (const float*)bd, (const float*)Ww, (const float*)bw, (float*)delta, (float*)weight, E, dim),
    hipLaunchKernelGGL(k_heads<__half>, grid, block, 0, st_, (const __half*)x, (const __half*)gate, ld_gate, (const __half*)res, (__half*)net_out, (const __half*)Wd, (const __half*)bd, (const __half*)Ww, (const __half*)bw, (__half*)delta, (__half*)weight, E, dim));
  return check_launch("devo_upd_heads");
}

}  // extern "C"
