// SE(3) batched group ops for gfx950: one group element per lane, grid-stride, fp32 / fp64.
// Replaces the SE3 instantiations of the reference's Eigen-templated kernels
// (devo/lietorch/src/lietorch_gpu.cu:20-294) behind lietorch_backends.* (lietorch.cpp:286-316).
// These ops are launch/latency bound (batch = E .. 9E elements of 28-56 B): the design goal is one
// launch per op with fully coalesced 7-wide rows staged through registers, nothing else.
#include "common.h"
#include "se3_dev.h"
#include <stdarg.h>

namespace devo {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#define LOOP(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x)

template <typename T> DEVO_HD void store_grad7(T* p, const T* g6) {
#pragma unroll
  for (int k = 0; k < 6; k++) p[k] = g6[k];
  p[6] = T(0);
}

template <typename T> __global__ void k_exp(const T* a, T* X, int64_t n) {
  LOOP(i, n) { se3_exp<T>(a + i * 6).store(X + i * 7); }
}
template <typename T> __global__ void k_exp_bwd(const T* grad, const T* a, T* da, int64_t n) {
  LOOP(i, n) { row_times_left_jacobian<T>(grad + i * 7, a + i * 6, da + i * 6); }      // lietorch_gpu.cu:32-44
}
template <typename T> __global__ void k_log(const T* X, T* a, int64_t n) {
  LOOP(i, n) { se3_log<T>(SE3<T>::load(X + i * 7), a + i * 6); }
}
template <typename T> __global__ void k_log_bwd(const T* grad, const T* X, T* dX, int64_t n) {
  LOOP(i, n) {                                                                          // :58-70
    T a[6], o[6];
    se3_log<T>(SE3<T>::load(X + i * 7), a);
    row_times_left_jacobian_inverse<T>(grad + i * 6, a, o);
    store_grad7(dX + i * 7, o);
  }
}
template <typename T> __global__ void k_inv(const T* X, T* Y, int64_t n) {
  LOOP(i, n) { SE3<T>::load(X + i * 7).inv().store(Y + i * 7); }
}
template <typename T> __global__ void k_inv_bwd(const T* grad, const T* X, T* dX, int64_t n) {
  LOOP(i, n) {                                                                          // :85-97  -dY * Adj(X^-1)
    T o[6];
    SE3<T>::load(X + i * 7).inv().row_times_Adj(grad + i * 7, o);
#pragma unroll
    for (int k = 0; k < 6; k++) o[k] = -o[k];
    store_grad7(dX + i * 7, o);
  }
}
template <typename T> __global__ void k_mul(const T* X, const T* Y, T* Z, int64_t n) {
  LOOP(i, n) { SE3<T>::load(X + i * 7).mul(SE3<T>::load(Y + i * 7)).store(Z + i * 7); }
}
template <typename T> __global__ void k_mul_bwd(const T* grad, const T* X, T* dX, T* dY, int64_t n) {
  LOOP(i, n) {                                                                          // :112-125
    T g[6], o[6];
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = grad[i * 7 + k];
    SE3<T>::load(X + i * 7).row_times_Adj(g, o);
    store_grad7(dX + i * 7, g);
    store_grad7(dY + i * 7, o);
  }
}
template <typename T> __global__ void k_adj(const T* X, const T* a, T* b, int64_t n) {
  LOOP(i, n) { SE3<T>::load(X + i * 7).adj(a + i * 6, b + i * 6); }
}
template <typename T> __global__ void k_adj_bwd(const T* grad, const T* X, const T* a, T* dX, T* da, int64_t n) {
  LOOP(i, n) {                                                                          // :140-157
    SE3<T> G = SE3<T>::load(X + i * 7);
    T b[6], o[6], db[6];
#pragma unroll
    for (int k = 0; k < 6; k++) db[k] = grad[i * 6 + k];
    G.adj(a + i * 6, b);
    G.row_times_Adj(db, da + i * 6);
    row_times_small_adj<T>(db, b, o);
#pragma unroll
    for (int k = 0; k < 6; k++) o[k] = -o[k];
    store_grad7(dX + i * 7, o);
  }
}
template <typename T> __global__ void k_adjT(const T* X, const T* a, T* b, int64_t n) {
  LOOP(i, n) { SE3<T>::load(X + i * 7).adjT(a + i * 6, b + i * 6); }
}
template <typename T> __global__ void k_adjT_bwd(const T* grad, const T* X, const T* a, T* dX, T* da, int64_t n) {
  LOOP(i, n) {                                                                          // :173-188
    SE3<T> G = SE3<T>::load(X + i * 7);
    T Adb[6], o[6], av[6];
#pragma unroll
    for (int k = 0; k < 6; k++) av[k] = a[i * 6 + k];
    G.adj(grad + i * 6, Adb);
#pragma unroll
    for (int k = 0; k < 6; k++) da[i * 6 + k] = Adb[k];
    row_times_small_adj<T>(av, Adb, o);
#pragma unroll
    for (int k = 0; k < 6; k++) o[k] = -o[k];
    store_grad7(dX + i * 7, o);
  }
}
template <typename T> __global__ void k_act(const T* X, const T* p, T* q, int64_t n) {
  LOOP(i, n) {
    V3<T> r = SE3<T>::load(X + i * 7).act(V3<T>{p[i * 3], p[i * 3 + 1], p[i * 3 + 2]});
    q[i * 3] = r.x; q[i * 3 + 1] = r.y; q[i * 3 + 2] = r.z;
  }
}
template <typename T> __global__ void k_act_bwd(const T* grad, const T* X, const T* p, T* dX, T* dp, int64_t n) {
  LOOP(i, n) {                                                                          // :204-221
    SE3<T> G = SE3<T>::load(X + i * 7);
    V3<T> g{grad[i * 3], grad[i * 3 + 1], grad[i * 3 + 2]};
    V3<T> pp = G.act(V3<T>{p[i * 3], p[i * 3 + 1], p[i * 3 + 2]});
    V3<T> d = mulTv(qmat(G.q), g);
    dp[i * 3] = d.x; dp[i * 3 + 1] = d.y; dp[i * 3 + 2] = d.z;
    V3<T> r = cross(pp, g);                       // g * hat(-pp) = pp x g
    T o[6] = {g.x, g.y, g.z, r.x, r.y, r.z};
    store_grad7(dX + i * 7, o);
  }
}
template <typename T> __global__ void k_act4(const T* X, const T* p, T* q, int64_t n) {
  LOOP(i, n) {                                                                          // se3.h:53-56
    SE3<T> G = SE3<T>::load(X + i * 7);
    T w = p[i * 4 + 3];
    V3<T> r = qrot(G.q, V3<T>{p[i * 4], p[i * 4 + 1], p[i * 4 + 2]}) + w * G.t;
    q[i * 4] = r.x; q[i * 4 + 1] = r.y; q[i * 4 + 2] = r.z; q[i * 4 + 3] = w;
  }
}
template <typename T> __global__ void k_act4_bwd(const T* grad, const T* X, const T* p, T* dX, T* dp, int64_t n) {
  LOOP(i, n) {                                                                          // :238-256, se3.h:211-217
    SE3<T> G = SE3<T>::load(X + i * 7);
    V3<T> g{grad[i * 4], grad[i * 4 + 1], grad[i * 4 + 2]};
    T g3 = grad[i * 4 + 3], w = p[i * 4 + 3];
    V3<T> pp = qrot(G.q, V3<T>{p[i * 4], p[i * 4 + 1], p[i * 4 + 2]}) + w * G.t;
    V3<T> d = mulTv(qmat(G.q), g);                // dq * T(4x4): first 3 cols = g R, last = g.t + g3
    dp[i * 4] = d.x; dp[i * 4 + 1] = d.y; dp[i * 4 + 2] = d.z; dp[i * 4 + 3] = dot(g, G.t) + g3;
    V3<T> r = cross(pp, g);
    T o[6] = {w * g.x, w * g.y, w * g.z, r.x, r.y, r.z};
    store_grad7(dX + i * 7, o);
  }
}
template <typename T> __global__ void k_as_matrix(const T* X, T* M, int64_t n) {
  LOOP(i, n) {                                                                          // :258-269 row-major 4x4
    SE3<T> G = SE3<T>::load(X + i * 7);
    M3<T> R = qmat(G.q);
    T* m = M + i * 16;
    T tv[3] = {G.t.x, G.t.y, G.t.z};
#pragma unroll
    for (int r = 0; r < 3; r++) { m[r * 4] = R.m[r][0]; m[r * 4 + 1] = R.m[r][1]; m[r * 4 + 2] = R.m[r][2]; m[r * 4 + 3] = tv[r]; }
    m[12] = 0; m[13] = 0; m[14] = 0; m[15] = 1;
  }
}
template <typename T> __global__ void k_jinv(const T* X, const T* a, T* b, int64_t n) {
  LOOP(i, n) {                                                                          // :283-294
    T l[6];
    se3_log<T>(SE3<T>::load(X + i * 7), l);
    left_jacobian_inverse_times<T>(l, a + i * 6, b + i * 6);
  }
}


// The update step of the differentiable BA (devo/ba.py:172-182) as one kernel per direction: poses[fixedp .. fixedp + n_opt) <- Exp(dX_i) * pose_i
// (`poses.retr`), every patch's inverse depth <- clamp(d + dZ_k, dmin, dmax) over its P x P pixels, everything else copied.  The reference
// (and round 3) ran it as torch.stack / clamp / zeros / slice assignment + the Exp and Mul group ops: ~8 launches forward, ~12 backward.
// Work item i < N: pose i; item N + k: patch k.
__global__ void k_ba_apply_step(const float* __restrict__ poses, const float* __restrict__ patches, const float* __restrict__ dX, const float* __restrict__ dZ,
                                int N, int Np, int PP, int fixedp, int n_opt, float dmin, float dmax, float* __restrict__ poses_out, float* __restrict__ patches_out) {
  LOOP(i, (int64_t)N + Np) {
    if (i < N) {
      SE3<float> X = SE3<float>::load(poses + i * 7);
      if (i >= fixedp && i < fixedp + n_opt) X = se3_exp<float>(dX + (i - fixedp) * 6).mul(X);
      X.store(poses_out + i * 7);
    } else {
      const int64_t k = i - N;
      const float* p = patches + k * 3 * PP;
      float* o = patches_out + k * 3 * PP;
      const float dz = dZ[k];
      for (int j = 0; j < 2 * PP; j++) o[j] = p[j];
      for (int j = 2 * PP; j < 3 * PP; j++) o[j] = fminf(fmaxf(p[j] + dz, dmin), dmax);
    }
  }
}
// adjoint: g_poses_out [N,7] (lietorch's embedding: tangent in the first six), g_patches_out -> g_poses, g_patches, g_dX [6 n_opt], g_dZ [Np]
__global__ void k_ba_apply_step_bwd(const float* __restrict__ poses, const float* __restrict__ patches, const float* __restrict__ dX, const float* __restrict__ dZ,
                                    const float* __restrict__ g_poses_out, const float* __restrict__ g_patches_out, int N, int Np, int PP, int fixedp,
                                    int n_opt, float dmin, float dmax, float* __restrict__ g_poses, float* __restrict__ g_patches, float* __restrict__ g_dX,
                                    float* __restrict__ g_dZ) {
  LOOP(i, (int64_t)N + Np) {
    if (i < N) {
      if (i >= fixedp && i < fixedp + n_opt) {
        const float* a = dX + (i - fixedp) * 6;
        float g[7], o[6], ga7[7];
#pragma unroll
        for (int k = 0; k < 7; k++) g[k] = g_poses_out ? g_poses_out[i * 7 + k] : 0.0f;
        se3_exp<float>(a).row_times_Adj(g, o);                          // Mul's adjoint (lietorch_gpu.cu:112-125): d left = g, d right = g Adj(left)
        store_grad7(g_poses + i * 7, o);
        store_grad7(ga7, g);
        row_times_left_jacobian<float>(ga7, a, g_dX + (i - fixedp) * 6); // Exp's adjoint (:32-44)
      } else {
#pragma unroll
        for (int k = 0; k < 7; k++) g_poses[i * 7 + k] = g_poses_out ? g_poses_out[i * 7 + k] : 0.0f;
      }
    } else {
      const int64_t k = i - N;
      const float* p = patches + k * 3 * PP;
      const float* g = g_patches_out ? g_patches_out + k * 3 * PP : nullptr;
      float* o = g_patches + k * 3 * PP;
      const float dz = dZ[k];
      float s = 0.0f;
      for (int j = 0; j < 2 * PP; j++) o[j] = g ? g[j] : 0.0f;
      for (int j = 2 * PP; j < 3 * PP; j++) {
        const float d = p[j] + dz;
        const float gj = (g && d >= dmin && d <= dmax) ? g[j] : 0.0f;    // clamp passes the gradient inside [dmin, dmax] (ATen's clamp_backward mask)
        o[j] = gj; s += gj;
      }
      g_dZ[k] = s;
    }
  }
}

}  // namespace devo

using namespace devo;

#define SE3_DISPATCH(NAME, KERNEL, ...)                                                      \
  do {                                                                                       \
    if (n < 0) { set_error(NAME ": negative batch"); return DEVO_ERR_ARG; }                  \
    if (n == 0) return DEVO_OK;                                                              \
    hipStream_t st = (hipStream_t)s;                                                         \
    int blocks = blocks_for(n, 256, 4096);                                                   \
    if (dtype == DEVO_F32) { typedef float T; hipLaunchKernelGGL(KERNEL<T>, dim3(blocks), dim3(256), 0, st, __VA_ARGS__); } \
    else if (dtype == DEVO_F64) { typedef double T; hipLaunchKernelGGL(KERNEL<T>, dim3(blocks), dim3(256), 0, st, __VA_ARGS__); } \
    else { set_error(NAME ": dtype must be F32 or F64"); return DEVO_ERR_UNSUPPORTED; }      \
    return check_launch(NAME);                                                               \
  } while (0)

#define CP(x) ((const T*)(x))
#define MP(x) ((T*)(x))

extern "C" {

int devo_abi_version(void) { return DEVO_ABI_VERSION; }
const char* devo_last_error(void) { return g_err; }
int devo_stream_capturing(devo_stream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return st == hipStreamCaptureStatusNone ? 0 : 1;
}

int devo_se3_exp(const void* a, void* X, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_exp", k_exp, CP(a), MP(X), n); }
int devo_se3_exp_backward(const void* grad, const void* a, void* da, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_exp_backward", k_exp_bwd, CP(grad), CP(a), MP(da), n); }
int devo_se3_log(const void* X, void* a, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_log", k_log, CP(X), MP(a), n); }
int devo_se3_log_backward(const void* grad, const void* X, void* dX, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_log_backward", k_log_bwd, CP(grad), CP(X), MP(dX), n); }
int devo_se3_inv(const void* X, void* Y, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_inv", k_inv, CP(X), MP(Y), n); }
int devo_se3_inv_backward(const void* grad, const void* X, void* dX, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_inv_backward", k_inv_bwd, CP(grad), CP(X), MP(dX), n); }
int devo_se3_mul(const void* X, const void* Y, void* Z, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_mul", k_mul, CP(X), CP(Y), MP(Z), n); }
int devo_se3_mul_backward(const void* grad, const void* X, const void* Y, void* dX, void* dY, int64_t n, int dtype, devo_stream_t s) { (void)Y; SE3_DISPATCH("devo_se3_mul_backward", k_mul_bwd, CP(grad), CP(X), MP(dX), MP(dY), n); }
int devo_se3_adj(const void* X, const void* a, void* b, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_adj", k_adj, CP(X), CP(a), MP(b), n); }
int devo_se3_adj_backward(const void* grad, const void* X, const void* a, void* dX, void* da, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_adj_backward", k_adj_bwd, CP(grad), CP(X), CP(a), MP(dX), MP(da), n); }
int devo_se3_adjT(const void* X, const void* a, void* b, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_adjT", k_adjT, CP(X), CP(a), MP(b), n); }
int devo_se3_adjT_backward(const void* grad, const void* X, const void* a, void* dX, void* da, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_adjT_backward", k_adjT_bwd, CP(grad), CP(X), CP(a), MP(dX), MP(da), n); }
int devo_se3_act(const void* X, const void* p, void* q, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_act", k_act, CP(X), CP(p), MP(q), n); }
int devo_se3_act_backward(const void* grad, const void* X, const void* p, void* dX, void* dp, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_act_backward", k_act_bwd, CP(grad), CP(X), CP(p), MP(dX), MP(dp), n); }
int devo_se3_act4(const void* X, const void* p, void* q, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_act4", k_act4, CP(X), CP(p), MP(q), n); }
int devo_se3_act4_backward(const void* grad, const void* X, const void* p, void* dX, void* dp, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_act4_backward", k_act4_bwd, CP(grad), CP(X), CP(p), MP(dX), MP(dp), n); }
int devo_se3_as_matrix(const void* X, void* T44, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_as_matrix", k_as_matrix, CP(X), MP(T44), n); }
int devo_se3_jinv(const void* X, const void* a, void* b, int64_t n, int dtype, devo_stream_t s) { SE3_DISPATCH("devo_se3_jinv", k_jinv, CP(X), CP(a), MP(b), n); }

int devo_ba_apply_step(const float* poses, const float* patches, const float* dX, const float* dZ, int N, int Np, int P, int fixedp, int n_opt,
                       float dmin, float dmax, float* poses_out, float* patches_out, devo_stream_t s) {
  if (N < 0 || Np < 0 || P <= 0 || fixedp < 0 || n_opt < 0 || fixedp + n_opt > N) { set_error("devo_ba_apply_step: bad sizes"); return DEVO_ERR_ARG; }
  if (N + Np == 0) return DEVO_OK;
  if (!poses || !patches || !dZ || !poses_out || !patches_out || (n_opt > 0 && !dX)) { set_error("devo_ba_apply_step: null tensor"); return DEVO_ERR_ARG; }
  hipLaunchKernelGGL(k_ba_apply_step, dim3((unsigned)((N + Np + 255) / 256)), dim3(256), 0, (hipStream_t)s, poses, patches, dX, dZ, N, Np, P * P, fixedp, n_opt,
                     dmin, dmax, poses_out, patches_out);
  return check_launch("devo_ba_apply_step");
}
int devo_ba_apply_step_backward(const float* poses, const float* patches, const float* dX, const float* dZ, const float* g_poses_out, const float* g_patches_out,
                                int N, int Np, int P, int fixedp, int n_opt, float dmin, float dmax, float* g_poses, float* g_patches, float* g_dX,
                                float* g_dZ, devo_stream_t s) {
  if (N < 0 || Np < 0 || P <= 0 || fixedp < 0 || n_opt < 0 || fixedp + n_opt > N) { set_error("devo_ba_apply_step_backward: bad sizes"); return DEVO_ERR_ARG; }
  if (N + Np == 0) return DEVO_OK;
  if (!poses || !patches || !dZ || !g_poses || !g_patches || !g_dZ || (n_opt > 0 && (!dX || !g_dX))) { set_error("devo_ba_apply_step_backward: null tensor"); return DEVO_ERR_ARG; }
  hipLaunchKernelGGL(k_ba_apply_step_bwd, dim3((unsigned)((N + Np + 255) / 256)), dim3(256), 0, (hipStream_t)s, poses, patches, dX, dZ, g_poses_out, g_patches_out,
                     N, Np, P * P, fixedp, n_opt, dmin, dmax, g_poses, g_patches, g_dX, g_dZ);
  return check_launch("devo_ba_apply_step_backward");
}

}  // extern "C"
