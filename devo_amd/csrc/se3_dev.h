// SE(3) device maths for gfx950 — header-only, no Eigen.
// Semantics follow the reference's lietorch group classes (devo/lietorch/include/se3.h:36-217,
// so3.h:27-208, common.h:7) and, where marked "fastba", the plain-C helpers of
// devo/fastba/ba_cuda.cu:18-156.  Element layout: (tx,ty,tz, qx,qy,qz,qw).
#pragma once
#include <hip/hip_runtime.h>

namespace devo {

template <typename T> struct Consts;
template <> struct Consts<float>  { static constexpr float  eps = 1e-6f; };
template <> struct Consts<double> { static constexpr double eps = 1e-6;  };

#define DEVO_HD __device__ __forceinline__

template <typename T> DEVO_HD T t_sqrt(T x);
template <> DEVO_HD float  t_sqrt<float>(float x)   { return sqrtf(x); }
template <> DEVO_HD double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> DEVO_HD T t_sin(T x);
template <> DEVO_HD float  t_sin<float>(float x)   { return sinf(x); }
template <> DEVO_HD double t_sin<double>(double x) { return sin(x); }
template <typename T> DEVO_HD T t_cos(T x);
template <> DEVO_HD float  t_cos<float>(float x)   { return cosf(x); }
template <> DEVO_HD double t_cos<double>(double x) { return cos(x); }
template <typename T> DEVO_HD T t_atan(T x);
template <> DEVO_HD float  t_atan<float>(float x)   { return atanf(x); }
template <> DEVO_HD double t_atan<double>(double x) { return atan(x); }
template <typename T> DEVO_HD T t_abs(T x) { return x < T(0) ? -x : x; }

template <typename T> struct V3 { T x, y, z; };
template <typename T> struct Q4 { T x, y, z, w; };
template <typename T> struct M3 { T m[3][3]; };

template <typename T> DEVO_HD V3<T> v3(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> DEVO_HD V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> DEVO_HD V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> DEVO_HD V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <typename T> DEVO_HD V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> DEVO_HD T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> DEVO_HD V3<T> cross(V3<T> a, V3<T> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// ---- quaternion pieces ----------------------------------------------------------------------
template <typename T> DEVO_HD Q4<T> qnormalize(Q4<T> q) {                 // so3.h:31-37
  T n = t_sqrt<T>(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  T r = T(1) / n;
  return {q.x * r, q.y * r, q.z * r, q.w * r};
}
template <typename T> DEVO_HD Q4<T> qconj(Q4<T> q) { return {-q.x, -q.y, -q.z, q.w}; }
template <typename T> DEVO_HD Q4<T> qmul(Q4<T> a, Q4<T> b) {               // Eigen quaternion product
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
          a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
template <typename T> DEVO_HD V3<T> qrot(Q4<T> q, V3<T> p) {               // so3.h:51-56 / ba_cuda.cu:18-28
  V3<T> qv{q.x, q.y, q.z};
  V3<T> uv = cross(qv, p);
  uv = uv + uv;
  return p + q.w * uv + cross(qv, uv);
}
template <typename T> DEVO_HD M3<T> qmat(Q4<T> q) {                        // Eigen toRotationMatrix
  T tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3<T> R;
  R.m[0][0] = 1 - (tyy + tzz); R.m[0][1] = txy - twz;       R.m[0][2] = txz + twy;
  R.m[1][0] = txy + twz;       R.m[1][1] = 1 - (txx + tzz); R.m[1][2] = tyz - twx;
  R.m[2][0] = txz - twy;       R.m[2][1] = tyz + twx;       R.m[2][2] = 1 - (txx + tyy);
  return R;
}
template <typename T> DEVO_HD V3<T> mulv(const M3<T>& A, V3<T> v) {
  return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z,
          A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
          A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
template <typename T> DEVO_HD V3<T> mulTv(const M3<T>& A, V3<T> v) {       // A^T v
  return {A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z,
          A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
          A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z};
}
template <typename T> DEVO_HD M3<T> hat(V3<T> v) {                         // so3.h:104-112
  M3<T> H;
  H.m[0][0] = 0;    H.m[0][1] = -v.z; H.m[0][2] = v.y;
  H.m[1][0] = v.z;  H.m[1][1] = 0;    H.m[1][2] = -v.x;
  H.m[2][0] = -v.y; H.m[2][1] = v.x;  H.m[2][2] = 0;
  return H;
}
template <typename T> DEVO_HD M3<T> mm(const M3<T>& A, const M3<T>& B) {
  M3<T> C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
  return C;
}
template <typename T> DEVO_HD M3<T> lin3(T a, const M3<T>& A, T b, const M3<T>& B) {   // I + aA + bB
  M3<T> C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C.m[i][j] = (i == j ? T(1) : T(0)) + a * A.m[i][j] + b * B.m[i][j];
  return C;
}

// ---- SE3 element ----------------------------------------------------------------------------
template <typename T> struct SE3 {
  V3<T> t;
  Q4<T> q;
  DEVO_HD static SE3 load(const T* p) {                                    // se3.h:34 + so3.h:31-33
    SE3 X;
    X.t = {p[0], p[1], p[2]};
    X.q = qnormalize(Q4<T>{p[3], p[4], p[5], p[6]});
    return X;
  }
  DEVO_HD void store(T* p) const {
    p[0] = t.x; p[1] = t.y; p[2] = t.z; p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
  }
  DEVO_HD SE3 inv() const {                                                // se3.h:36-38
    SE3 Y;
    Y.q = qnormalize(qconj(q));
    Y.t = -qrot(Y.q, t);
    return Y;
  }
  DEVO_HD SE3 mul(const SE3& o) const {                                    // se3.h:45-47
    SE3 Z;
    Z.q = qnormalize(qmul(q, o.q));
    Z.t = t + qrot(q, o.t);
    return Z;
  }
  DEVO_HD V3<T> act(V3<T> p) const { return qrot(q, p) + t; }              // se3.h:49-51
  // Adj = [[R, [t]x R],[0, R]]   (se3.h:58-67)
  DEVO_HD void adj(const T* a, T* b) const {                               // b = Adj a
    M3<T> R = qmat(q);
    V3<T> tau{a[0], a[1], a[2]}, phi{a[3], a[4], a[5]};
    V3<T> Rphi = mulv(R, phi);
    V3<T> u = mulv(R, tau) + cross(t, Rphi);
    b[0] = u.x; b[1] = u.y; b[2] = u.z; b[3] = Rphi.x; b[4] = Rphi.y; b[5] = Rphi.z;
  }
  DEVO_HD void adjT(const T* a, T* b) const {                              // b = Adj^T a
    M3<T> R = qmat(q);
    V3<T> a0{a[0], a[1], a[2]}, a1{a[3], a[4], a[5]};
    V3<T> b0 = mulTv(R, a0);
    // ([t]x R)^T a0 = R^T [t]x^T a0 = R^T (a0 x t)
    V3<T> b1 = mulTv(R, cross(a0, t)) + mulTv(R, a1);
    b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b1.x; b[4] = b1.y; b[5] = b1.z;
  }
  DEVO_HD void row_times_Adj(const T* g, T* out) const { adjT(g, out); }   // g(1x6) * Adj == (Adj^T g^T)^T
};

// ---- exp / log and Jacobians ----------------------------------------------------------------
template <typename T> DEVO_HD Q4<T> so3_exp(V3<T> phi) {                   // so3.h:153-168
  T theta2 = dot(phi, phi), theta = t_sqrt<T>(theta2), imag, real;
  if (theta < Consts<T>::eps) {
    T theta4 = theta2 * theta2;
    imag = T(0.5) - T(1.0 / 48.0) * theta2 + T(1.0 / 3840.0) * theta4;
    real = T(1) - T(1.0 / 8.0) * theta2 + T(1.0 / 384.0) * theta4;
  } else {
    imag = t_sin<T>(T(0.5) * theta) / theta;
    real = t_cos<T>(T(0.5) * theta);
  }
  return qnormalize(Q4<T>{imag * phi.x, imag * phi.y, imag * phi.z, real});
}
template <typename T> DEVO_HD V3<T> so3_log(Q4<T> q) {                     // so3.h:115-151
  T sn = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w, f;
  const T eps = Consts<T>::eps;
  if (sn < eps * eps) {
    f = T(2) / w - T(2.0 / 3.0) * sn / (w * w * w);
  } else {
    T n = t_sqrt<T>(sn);
    if (t_abs(w) < eps) f = (w > T(0)) ? T(3.14159265358979323846) / n : -T(3.14159265358979323846) / n;
    else f = T(2) * t_atan<T>(n / w) / n;
  }
  return {f * q.x, f * q.y, f * q.z};
}
template <typename T> DEVO_HD M3<T> so3_left_jacobian(V3<T> phi) {         // so3.h:170-187
  M3<T> P = hat(phi), P2 = mm(P, P);
  T theta2 = dot(phi, phi), theta = t_sqrt<T>(theta2);
  bool small = theta < Consts<T>::eps;
  T c1 = small ? T(0.5) - T(1.0 / 24.0) * theta2 : (T(1) - t_cos<T>(theta)) / theta2;
  T c2 = small ? T(1.0 / 6.0) - T(1.0 / 120.0) * theta2 : (theta - t_sin<T>(theta)) / (theta2 * theta);
  return lin3(c1, P, c2, P2);
}
template <typename T> DEVO_HD M3<T> so3_left_jacobian_inverse(V3<T> phi) { // so3.h:189-205
  M3<T> P = hat(phi), P2 = mm(P, P);
  T theta2 = dot(phi, phi), theta = t_sqrt<T>(theta2), half = T(0.5) * theta;
  T c2 = (theta < Consts<T>::eps) ? T(1.0 / 12.0)
         : (T(1) - theta * t_cos<T>(half) / (T(2) * t_sin<T>(half))) / (theta * theta);
  return lin3(T(-0.5), P, c2, P2);
}
template <typename T> DEVO_HD SE3<T> se3_exp(const T* a) {                 // se3.h:134-142
  V3<T> tau{a[0], a[1], a[2]}, phi{a[3], a[4], a[5]};
  SE3<T> X;
  X.q = so3_exp(phi);
  X.t = mulv(so3_left_jacobian(phi), tau);
  return X;
}
template <typename T> DEVO_HD void se3_log(const SE3<T>& X, T* a) {        // se3.h:124-132
  V3<T> phi = so3_log(X.q);
  V3<T> tau = mulv(so3_left_jacobian_inverse(phi), X.t);
  a[0] = tau.x; a[1] = tau.y; a[2] = tau.z; a[3] = phi.x; a[4] = phi.y; a[5] = phi.z;
}
template <typename T> DEVO_HD M3<T> se3_calcQ(const T* a) {                // se3.h:144-173
  V3<T> tau{a[0], a[1], a[2]}, phi{a[3], a[4], a[5]};
  M3<T> Tau = hat(tau), Phi = hat(phi);
  T theta2 = dot(phi, phi), theta = t_sqrt<T>(theta2), theta4 = theta2 * theta2;
  bool small = theta < Consts<T>::eps;
  T c1 = small ? T(1.0 / 6.0) - T(1.0 / 120.0) * theta2 : (theta - t_sin<T>(theta)) / (theta2 * theta);
  T c2 = small ? T(1.0 / 24.0) - T(1.0 / 720.0) * theta2 : (theta2 + 2 * t_cos<T>(theta) - 2) / (2 * theta4);
  T c3 = small ? T(1.0 / 120.0) - T(1.0 / 2520.0) * theta2
               : (2 * theta - 3 * t_sin<T>(theta) + theta * t_cos<T>(theta)) / (2 * theta4 * theta);
  M3<T> PT = mm(Phi, Tau), TP = mm(Tau, Phi), PTP = mm(PT, Phi);
  M3<T> PPT = mm(Phi, PT), TPP = mm(TP, Phi), PTPP = mm(PTP, Phi), PPTP = mm(Phi, PTP);
  M3<T> Q;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      Q.m[i][j] = T(0.5) * Tau.m[i][j] + c1 * (PT.m[i][j] + TP.m[i][j] + PTP.m[i][j]) +
                  c2 * (PPT.m[i][j] + TPP.m[i][j] - 3 * PTP.m[i][j]) + c3 * (PTPP.m[i][j] + PPTP.m[i][j]);
  return Q;
}
// out(1x6) = g(1x6) * J where J = [[A, Bm],[0, A]]  (block form shared by J_l and J_l^{-1})
template <typename T> DEVO_HD void row_times_blockJ(const T* g, const M3<T>& A, const M3<T>& Bm, T* out) {
  V3<T> g0{g[0], g[1], g[2]}, g1{g[3], g[4], g[5]};
  V3<T> o0 = mulTv(A, g0);
  V3<T> o1 = mulTv(Bm, g0) + mulTv(A, g1);
  out[0] = o0.x; out[1] = o0.y; out[2] = o0.z; out[3] = o1.x; out[4] = o1.y; out[5] = o1.z;
}
template <typename T> DEVO_HD void row_times_left_jacobian(const T* g, const T* a, T* out) {   // se3.h:175-186
  V3<T> phi{a[3], a[4], a[5]};
  row_times_blockJ(g, so3_left_jacobian(phi), se3_calcQ(a), out);
}
template <typename T> DEVO_HD void neg_JQJ(const M3<T>& Ji, const M3<T>& Q, M3<T>& out) {
  M3<T> t = mm(mm(Ji, Q), Ji);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) out.m[i][j] = -t.m[i][j];
}
template <typename T> DEVO_HD void row_times_left_jacobian_inverse(const T* g, const T* a, T* out) {  // se3.h:188-201
  V3<T> phi{a[3], a[4], a[5]};
  M3<T> Ji = so3_left_jacobian_inverse(phi), B;
  neg_JQJ(Ji, se3_calcQ(a), B);
  row_times_blockJ(g, Ji, B, out);
}
template <typename T> DEVO_HD void left_jacobian_inverse_times(const T* a, const T* v, T* out) {     // J_l^{-1}(a) v
  V3<T> phi{a[3], a[4], a[5]};
  M3<T> Ji = so3_left_jacobian_inverse(phi), B;
  neg_JQJ(Ji, se3_calcQ(a), B);
  V3<T> v0{v[0], v[1], v[2]}, v1{v[3], v[4], v[5]};
  V3<T> o0 = mulv(Ji, v0) + mulv(B, v1), o1 = mulv(Ji, v1);
  out[0] = o0.x; out[1] = o0.y; out[2] = o0.z; out[3] = o1.x; out[4] = o1.y; out[5] = o1.z;
}
// out(1x6) = a(1x6) * ad(b),  ad(b) = [[Phi, Tau],[0, Phi]]  (se3.h:100-113)
template <typename T> DEVO_HD void row_times_small_adj(const T* a, const T* b, T* out) {
  V3<T> a0{a[0], a[1], a[2]}, a1{a[3], a[4], a[5]}, tau{b[0], b[1], b[2]}, phi{b[3], b[4], b[5]};
  // row * hat(v) = (hat(v)^T row^T)^T = -(v x row) = row x v
  V3<T> o0 = cross(a0, phi);
  V3<T> o1 = cross(a0, tau) + cross(a1, phi);
  out[0] = o0.x; out[1] = o0.y; out[2] = o0.z; out[3] = o1.x; out[4] = o1.y; out[5] = o1.z;
}

// ---- fastba helpers (plain float, exact restatement of the inference-BA group maths) ----------
DEVO_HD void fb_relSE3(const float* ti, const float* qi, const float* tj, const float* qj, float* tij, float* qij) {
  // ba_cuda.cu:56-67: q_ij = q_j * q_i^{-1} (no renormalisation), t_ij = t_j - R(q_ij) t_i
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  V3<float> r = qrot(Q4<float>{qij[0], qij[1], qij[2], qij[3]}, V3<float>{ti[0], ti[1], ti[2]});
  tij[0] = tj[0] - r.x; tij[1] = tj[1] - r.y; tij[2] = tj[2] - r.z;
}
DEVO_HD void fb_actSE3(const float* t, const float* q, const float* X, float* Y) {   // ba_cuda.cu:30-37
  V3<float> r = qrot(Q4<float>{q[0], q[1], q[2], q[3]}, V3<float>{X[0], X[1], X[2]});
  Y[0] = r.x + X[3] * t[0]; Y[1] = r.y + X[3] * t[1]; Y[2] = r.z + X[3] * t[2]; Y[3] = X[3];
}
DEVO_HD void fb_adjSE3(const float* t, const float* q, const float* X, float* Y) {   // ba_cuda.cu:39-54
  Q4<float> qi{-q[0], -q[1], -q[2], q[3]};
  V3<float> x0{X[0], X[1], X[2]}, x1{X[3], X[4], X[5]}, tt{t[0], t[1], t[2]};
  V3<float> y0 = qrot(qi, x0), y1 = qrot(qi, x1);
  V3<float> u = cross(x0, tt);
  V3<float> v = qrot(qi, u);
  Y[0] = y0.x; Y[1] = y0.y; Y[2] = y0.z; Y[3] = y1.x + v.x; Y[4] = y1.y + v.y; Y[5] = y1.z + v.z;
}
DEVO_HD void fb_expSO3(const float* phi, float* q) {                                  // ba_cuda.cu:70-92
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta_p4 = theta_sq * theta_sq, theta = sqrtf(theta_sq), imag, real;
  // ba_cuda.cu:79-81 compares with the double 1e-8 and evaluates the series in double (its literals are doubles), rounding once:
  // (double)theta_sq < 1e-8 <=> theta_sq <= 1e-8f (1e-8f = 9.99999994e-9 is the float just below the double 1e-8)
  if (theta_sq <= 1e-8f) {
    imag = (float)(0.5 - (1.0 / 48.0) * (double)theta_sq + (1.0 / 3840.0) * (double)theta_p4);
    real = (float)(1.0 - (1.0 / 8.0) * (double)theta_sq + (1.0 / 384.0) * (double)theta_p4);
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  q[0] = imag * phi[0]; q[1] = imag * phi[1]; q[2] = imag * phi[2]; q[3] = real;
}
DEVO_HD void fb_expSE3(const float* xi, float* t, float* q) {                          // ba_cuda.cu:108-135
  fb_expSO3(xi + 3, q);
  V3<float> tau{xi[0], xi[1], xi[2]}, phi{xi[3], xi[4], xi[5]};
  float theta_sq = dot(phi, phi), theta = sqrtf(theta_sq);
  V3<float> tt = tau;
  if (theta > 1e-4f) {
    float a = (1 - cosf(theta)) / theta_sq;
    V3<float> c1 = cross(phi, tau);
    tt = tt + a * c1;
    float b = (theta - sinf(theta)) / (theta * theta_sq);
    tt = tt + b * cross(phi, c1);
  }
  t[0] = tt.x; t[1] = tt.y; t[2] = tt.z;
}
DEVO_HD void fb_retrSE3(const float* xi, const float* t, const float* q, float* t1, float* q1) {   // ba_cuda.cu:138-156
  float dt[3], dq[4];
  fb_expSE3(xi, dt, dq);
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  V3<float> r = qrot(Q4<float>{dq[0], dq[1], dq[2], dq[3]}, V3<float>{t[0], t[1], t[2]});
  t1[0] = r.x + dt[0]; t1[1] = r.y + dt[1]; t1[2] = r.z + dt[2];
}

}  // namespace devo
