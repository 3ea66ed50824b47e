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

int devo_abi_version(void) { return 1; }
const char* devo_last_error(void) { return g_err; }

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

}  // extern "C"
