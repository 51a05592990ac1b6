// SE(3) pose algebra on device: one thread per pose, float64 like the reference's numpy host code.
//   RT_transform   /root/reference/lib/pair_matching/RT_transform.py:135-161 (R_transform :51-69, T_transform :82-103, quat2mat :393-443)
//   calc_RT_delta  RT_transform.py:16-48 (R_inv_transform :72-79, T_inv_transform :113-132, mat2quat :446-523)
//   ZoomTrans      deepim/operator_py/zoom_trans.py:22-76
//   Transform3D    deepim/operator_py/transform3d.py:42-327
// Keeping these on the GPU removes the asnumpy() sync + numpy + re-upload between refinement iterations
// (deepim/core/tester.py:523-532, lib/pair_matching/batch_updater_py_multi.py:211-312).
#include "common.h"

namespace dim {

enum { ROT_MODEL = 0, ROT_CAMERA = 1, ROT_CAMERA_NEW = 2, ROT_NAIVE = 3 };

__device__ inline void quat2mat_d(const double q[4], double M[9]) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double Nq = w * w + x * x + y * y + z * z;
  if (Nq < 2.220446049250313e-16) {
    M[0] = 1; M[1] = 0; M[2] = 0; M[3] = 0; M[4] = 1; M[5] = 0; M[6] = 0; M[7] = 0; M[8] = 1;
    return;
  }
  double s = 2.0 / Nq;
  double X = x * s, Y = y * s, Z = z * s;
  double wX = w * X, wY = w * Y, wZ = w * Z, xX = x * X, xY = x * Y, xZ = x * Z, yY = y * Y, yZ = y * Z, zZ = z * Z;
  M[0] = 1.0 - (yY + zZ); M[1] = xY - wZ;         M[2] = xZ + wY;
  M[3] = xY + wZ;         M[4] = 1.0 - (xX + zZ); M[5] = yZ - wX;
  M[6] = xZ - wY;         M[7] = yZ + wX;         M[8] = 1.0 - (xX + yY);
}

__device__ inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// rotation matrix -> (w,x,y,z), w >= 0.  The reference takes the dominant eigenvector of the
// Bar-Itzhack matrix (mat2quat, RT_transform.py:446-523); for a rotation matrix that vector is
// the usual quaternion, computed here with Shepperd's branch selection and normalised.
__device__ inline void mat2quat_d(const double M[9], double q[4]) {
  double tr = M[0] + M[4] + M[8];
  double w, x, y, z;
  if (tr > M[0] && tr > M[4] && tr > M[8]) {
    w = 1.0 + tr; x = M[7] - M[5]; y = M[2] - M[6]; z = M[3] - M[1];
  } else if (M[0] > M[4] && M[0] > M[8]) {
    x = 1.0 + M[0] - M[4] - M[8]; w = M[7] - M[5]; y = M[1] + M[3]; z = M[2] + M[6];
  } else if (M[4] > M[8]) {
    y = 1.0 - M[0] + M[4] - M[8]; w = M[2] - M[6]; x = M[1] + M[3]; z = M[5] + M[7];
  } else {
    z = 1.0 - M[0] - M[4] + M[8]; w = M[3] - M[1]; x = M[2] + M[6]; y = M[5] + M[7];
  }
  double n = sqrt(w * w + x * x + y * y + z * z);
  if (w < 0) n = -n;
  q[0] = w / n; q[1] = x / n; q[2] = y / n; q[3] = z / n;
}

// Euler angles of the reference's EULER deltas: euler2mat / mat2euler with their default axes "sxyz" (RT_transform.py:250-383; the only
// convention RT_transform :139-140 and calc_RT_delta :39-40 use): rotations about the STATIC x, y, z axes in that order,
// M = Rz(ak) Ry(aj) Rx(ai).
__device__ inline void euler2mat_d(double ai, double aj, double ak, double M[9]) {
  double si = sin(ai), ci = cos(ai), sj = sin(aj), cj = cos(aj), sk = sin(ak), ck = cos(ak);
  M[0] = cj * ck; M[1] = sj * si * ck - ci * sk; M[2] = sj * ci * ck + si * sk;
  M[3] = cj * sk; M[4] = sj * si * sk + ci * ck; M[5] = sj * ci * sk - si * ck;
  M[6] = -sj;     M[7] = cj * si;                M[8] = cj * ci;
}

// inverse; at the gimbal lock (cos(aj) < 4 eps) the third angle is set to zero like the reference does (:370-377)
__device__ inline void mat2euler_d(const double M[9], double e[3]) {
  double cy = sqrt(M[0] * M[0] + M[3] * M[3]);
  if (cy > 4.0 * 2.220446049250313e-16) {
    e[0] = atan2(M[7], M[8]);
    e[1] = atan2(-M[6], cy);
    e[2] = atan2(M[3], M[0]);
  } else {
    e[0] = atan2(-M[5], M[4]);
    e[1] = atan2(-M[6], cy);
    e[2] = 0.0;
  }
}

// pose_out[b] = RT_transform(pose_src[b], se3[b,0:4], se3[b,4:7]); EULER: se3 rows are [ai, aj, ak, t] (6 floats)
template <bool EULER>
__global__ void se3_compose_kernel(const float* __restrict__ pose_src, const float* __restrict__ se3, float* __restrict__ pose_out,
                                   double* __restrict__ pose_out_f64, int B, int rot_coord, double m0, double m1, double m2,
                                   double s0, double s1, double s2) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* ps = pose_src + 12 * b;
  const float* d = se3 + (EULER ? 6 : 7) * b;
  double Rd[9], Rs[9], Ro[9], To[3];
  if (EULER) {
    euler2mat_d(d[0], d[1], d[2], Rd);
  } else {
    double q[4] = {d[0], d[1], d[2], d[3]};
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);  // LA.norm
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
    quat2mat_d(q, Rd);
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rs[3 * i + j] = ps[4 * i + j];
  double Ts[3] = {ps[3], ps[7], ps[11]};
  const float* dt = d + (EULER ? 3 : 4);
  double td[3] = {(double)dt[0], (double)dt[1], (double)dt[2]};
  if (rot_coord == ROT_NAIVE) {
    // se3_mul(se3_mx, pose_src): R = Rd*Rs, T = Rd*Ts + t   (float32 result in the reference)
    mat3_mul(Rd, Rs, Ro);
    for (int i = 0; i < 3; ++i) To[i] = Rd[3 * i] * Ts[0] + Rd[3 * i + 1] * Ts[1] + Rd[3 * i + 2] * Ts[2] + td[i];
  } else {
    if (rot_coord == ROT_MODEL) mat3_mul(Rs, Rd, Ro);
    else mat3_mul(Rd, Rs, Ro);
    double t0 = td[0] * s0 + m0, t1 = td[1] * s1 + m1, t2 = td[2] * s2 + m2;
    double z2 = Ts[2] / exp(t2);
    To[2] = z2;
    if (rot_coord == ROT_CAMERA_NEW) {
      To[0] = Ts[2] * t0 + Ts[0];
      To[1] = Ts[2] * t1 + Ts[1];
    } else {
      To[0] = z2 * (t0 + Ts[0] / Ts[2]);
      To[1] = z2 * (t1 + Ts[1] / Ts[2]);
    }
  }
  float* po = pose_out + 12 * b;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      po[4 * i + j] = (float)Ro[3 * i + j];
      if (pose_out_f64) pose_out_f64[12 * b + 4 * i + j] = Ro[3 * i + j];
    }
    po[4 * i + 3] = (float)To[i];
    if (pose_out_f64) pose_out_f64[12 * b + 4 * i + 3] = To[i];
  }
}

// (rot_delta quat (B,4), trans_delta (B,3)) = calc_RT_delta(pose_src, pose_tgt, rot_type="QUAT")
// rot_mat (B,3,3), optional: the same delta as a rotation matrix (rot_type="MATRIX"); rot (quaternion) may then be null
// rot_euler (B,3), optional: the same delta as static-xyz Euler angles (rot_type="EULER")
__global__ void se3_delta_kernel(const float* __restrict__ pose_src, const float* __restrict__ pose_tgt, float* __restrict__ rot,
                                 float* __restrict__ rot_mat, float* __restrict__ rot_euler, float* __restrict__ trans, int B, int rot_coord,
                                 double m0, double m1, double m2, double s0, double s1, double s2) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* ps = pose_src + 12 * b;
  const float* pt = pose_tgt + 12 * b;
  double Rs[9], Rt[9], RsT[9], Rd[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Rs[3 * i + j] = ps[4 * i + j];
      RsT[3 * j + i] = ps[4 * i + j];
      Rt[3 * i + j] = pt[4 * i + j];
    }
  double Ts[3] = {ps[3], ps[7], ps[11]}, Tt[3] = {pt[3], pt[7], pt[11]};
  double dT[3];
  if (rot_coord == ROT_NAIVE) {
    // se3_mul(pose_tgt, se3_inverse(pose_src)) -- float32 intermediates in the reference (projection.py:20,39)
    mat3_mul(Rt, RsT, Rd);
    float inv_t[3];
    for (int i = 0; i < 3; ++i) inv_t[i] = (float)(-(RsT[3 * i] * Ts[0] + RsT[3 * i + 1] * Ts[1] + RsT[3 * i + 2] * Ts[2]));
    for (int i = 0; i < 3; ++i) {
      dT[i] = (double)(float)(Rt[3 * i] * inv_t[0] + Rt[3 * i + 1] * inv_t[1] + Rt[3 * i + 2] * inv_t[2] + Tt[i]);
      for (int j = 0; j < 3; ++j) Rd[3 * i + j] = (double)(float)Rd[3 * i + j];
    }
  } else {
    if (rot_coord == ROT_MODEL) mat3_mul(RsT, Rt, Rd);
    else mat3_mul(Rt, RsT, Rd);
    if (rot_coord == ROT_CAMERA_NEW) {
      dT[0] = (Tt[0] - Ts[0]) / Ts[2];
      dT[1] = (Tt[1] - Ts[1]) / Ts[2];
    } else {
      dT[0] = Tt[0] / Tt[2] - Ts[0] / Ts[2];
      dT[1] = Tt[1] / Tt[2] - Ts[1] / Ts[2];
    }
    dT[2] = log(Ts[2] / Tt[2]);
    dT[0] = (dT[0] - m0) / s0;
    dT[1] = (dT[1] - m1) / s1;
    dT[2] = (dT[2] - m2) / s2;
  }
  if (rot) {
    double q[4];
    mat2quat_d(Rd, q);
    for (int i = 0; i < 4; ++i) rot[4 * b + i] = (float)q[i];
  }
  if (rot_mat)
    for (int i = 0; i < 9; ++i) rot_mat[9 * b + i] = (float)Rd[i];
  if (rot_euler) {
    double e[3];
    mat2euler_d(Rd, e);
    for (int i = 0; i < 3; ++i) rot_euler[3 * b + i] = (float)e[i];
  }
  for (int i = 0; i < 3; ++i) trans[3 * b + i] = (float)dT[i];
}

// KT[b] = K * calc_se3(pose_src[b], pose_tgt[b])  (batch_updater_py_multi.py:306-312; RT_transform.calc_se3 :186-197 with the
// float32 intermediates of lib/utils/projection.py:20,39).  Feeds dim_depth_to_flow.
__global__ void pose_to_KT_kernel(const float* __restrict__ pose_src, const float* __restrict__ pose_tgt, double k00, double k01, double k02,
                                  double k10, double k11, double k12, double k20, double k21, double k22, float* __restrict__ KT, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* ps = pose_src + 12 * b;
  const float* pt = pose_tgt + 12 * b;
  // se3_inverse(pose_src) -> float32
  float Ri[9], Ti[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Ri[3 * i + j] = ps[4 * j + i];
  for (int i = 0; i < 3; ++i)
    Ti[i] = (float)(-1.0 * ((double)Ri[3 * i] * ps[3] + (double)Ri[3 * i + 1] * ps[7] + (double)Ri[3 * i + 2] * ps[11]));
  // se3_mul(pose_tgt, inverse) -> float32
  float M[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      M[4 * i + j] = (float)((double)pt[4 * i] * Ri[j] + (double)pt[4 * i + 1] * Ri[3 + j] + (double)pt[4 * i + 2] * Ri[6 + j]);
    M[4 * i + 3] = (float)(((double)pt[4 * i] * Ti[0] + (double)pt[4 * i + 1] * Ti[1] + (double)pt[4 * i + 2] * Ti[2]) + (double)pt[4 * i + 3]);
  }
  const double K[9] = {k00, k01, k02, k10, k11, k12, k20, k21, k22};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j)
      KT[12 * b + 4 * i + j] = (float)(K[3 * i] * M[j] + K[3 * i + 1] * M[4 + j] + K[3 * i + 2] * M[8 + j]);
}

// ZoomTrans forward and backward (one thread per sample)
__global__ void zoom_trans_kernel(const float* __restrict__ zoom_factor, const float* __restrict__ in, float* __restrict__ out,
                                  int B, int mode /*0 copy, 1 divide by wx, 2 multiply by wx*/) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float wx = zoom_factor[4 * b];
  float x = in[3 * b], y = in[3 * b + 1], z = in[3 * b + 2];
  if (mode == 1) { x = x / wx; y = y / wx; }
  else if (mode == 2) { x = x * wx; y = y * wx; }
  out[3 * b] = x; out[3 * b + 1] = y; out[3 * b + 2] = z;
}

// ---------------------------------------------------------------- Transform3D
// float32 per-sample target pose exactly like transform3dOperator.forward (numpy float32 scalars)
__device__ inline bool t3d_quat2mat_f(const float* q, float thr, float M[9]) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  float Nq = w * w + x * x + y * y + z * z;
  bool ok = (Nq - 1.f > -thr) && (Nq - 1.f < thr);
  if (!ok) {
    M[0] = 1; M[1] = 0; M[2] = 0; M[3] = 0; M[4] = 1; M[5] = 0; M[6] = 0; M[7] = 0; M[8] = 1;
    return false;
  }
  float s = 2.0f / Nq;
  float X = x * s, Y = y * s, Z = z * s;
  float wX = w * X, wY = w * Y, wZ = w * Z, xX = x * X, xY = x * Y, xZ = x * Z, yY = y * Y, yZ = y * Z, zZ = z * Z;
  M[0] = 1.0f - (yY + zZ); M[1] = xY - wZ;          M[2] = xZ + wY;
  M[3] = xY + wZ;          M[4] = 1.0f - (xX + zZ); M[5] = yZ - wX;
  M[6] = xZ - wY;          M[7] = yZ + wX;          M[8] = 1.0f - (xX + yY);
  return true;
}

struct T3DConst {
  float m[3], s[3];
  int rot_coord;
};

__device__ inline void t3d_target(const float* q, const float* td, const float* ps, const T3DConst& c, float Rt[9], float Tt[3]) {
  float Rd[9], Rs[9];
  t3d_quat2mat_f(q, 1e-2f, Rd);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rs[3 * i + j] = ps[4 * i + j];
  const float* A = (c.rot_coord == ROT_MODEL) ? Rs : Rd;
  const float* Bm = (c.rot_coord == ROT_MODEL) ? Rd : Rs;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rt[3 * i + j] = A[3 * i] * Bm[j] + A[3 * i + 1] * Bm[3 + j] + A[3 * i + 2] * Bm[6 + j];
  float Ts[3] = {ps[3], ps[7], ps[11]};
  if (c.rot_coord == ROT_NAIVE) {
    for (int i = 0; i < 3; ++i) Tt[i] = Rd[3 * i] * Ts[0] + Rd[3 * i + 1] * Ts[1] + Rd[3 * i + 2] * Ts[2] + td[i];
    return;
  }
  float t0 = td[0] * c.s[0] + c.m[0], t1 = td[1] * c.s[1] + c.m[1], t2 = td[2] * c.s[2] + c.m[2];
  float z2 = Ts[2] / expf(t2);
  Tt[2] = z2;
  if (c.rot_coord == ROT_CAMERA_NEW) {
    Tt[0] = Ts[2] * t0 + Ts[0];
    Tt[1] = Ts[2] * t1 + Ts[1];
  } else {
    Tt[0] = z2 * (t0 + Ts[0] / Ts[2]);
    Tt[1] = z2 * (t1 + Ts[1] / Ts[2]);
  }
}

// out[b,:,n] = R_tgt[b] * P[b,:,n] + T_tgt[b]       points layout (B,3,Npts)
__global__ __launch_bounds__(256) void transform3d_fwd_kernel(const float* __restrict__ pts, const float* __restrict__ rot,
                                                              const float* __restrict__ trans, const float* __restrict__ pose_src,
                                                              float* __restrict__ out, int Npts, T3DConst c) {
  const int b = blockIdx.y;
  __shared__ float sR[9], sT[3];
  if (threadIdx.x == 0) t3d_target(rot + 4 * b, trans + 3 * b, pose_src + 12 * b, c, sR, sT);
  __syncthreads();
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Npts) return;
  const float* p = pts + (long)b * 3 * Npts;
  float x = p[n], y = p[Npts + n], z = p[2 * Npts + n];
  float* o = out + (long)b * 3 * Npts;
  o[n] = sR[0] * x + sR[1] * y + sR[2] * z + sT[0];
  o[Npts + n] = sR[3] * x + sR[4] * y + sR[5] * z + sT[1];
  o[2 * Npts + n] = sR[6] * x + sR[7] * y + sR[8] * z + sT[2];
}

// backward: d_rot (B,4), d_trans (B,3).  transform3d.py:120-327.  One block per sample.
__global__ __launch_bounds__(256) void transform3d_bwd_kernel(const float* __restrict__ grad, const float* __restrict__ pts,
                                                              const float* __restrict__ rot, const float* __restrict__ trans,
                                                              const float* __restrict__ pose_src, float* __restrict__ d_rot,
                                                              float* __restrict__ d_trans, int Npts, T3DConst c) {
  const int b = blockIdx.x;
  const float* g = grad + (long)b * 3 * Npts;
  const float* p = pts + (long)b * 3 * Npts;
  // 12 sums: dT[i] = sum_n g[i,n];  D[i][j] = sum_n g[i,n] * p[j,n]   (NAIVE uses src-transformed points)
  float acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  const float* ps = pose_src + 12 * b;
  for (int n = threadIdx.x; n < Npts; n += blockDim.x) {
    float gx = g[n], gy = g[Npts + n], gz = g[2 * Npts + n];
    float x = p[n], y = p[Npts + n], z = p[2 * Npts + n];
    if (c.rot_coord == ROT_NAIVE) {
      float sx = ps[0] * x + ps[1] * y + ps[2] * z + ps[3];
      float sy = ps[4] * x + ps[5] * y + ps[6] * z + ps[7];
      float sz = ps[8] * x + ps[9] * y + ps[10] * z + ps[11];
      x = sx; y = sy; z = sz;
    }
    acc[0] += gx; acc[1] += gy; acc[2] += gz;
    acc[3] += gx * x; acc[4] += gx * y; acc[5] += gx * z;
    acc[6] += gy * x; acc[7] += gy * y; acc[8] += gy * z;
    acc[9] += gz * x; acc[10] += gz * y; acc[11] += gz * z;
  }
  __shared__ float red[4][12];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = 0; i < 12; ++i) {
    float v = acc[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float S[12];
  for (int i = 0; i < 12; ++i) S[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  const float* td = trans + 3 * b;
  // ---- translation (T_transform_backward :193-225)
  float Ts[3] = {ps[3], ps[7], ps[11]};
  if (c.rot_coord == ROT_NAIVE) {
    d_trans[3 * b] = S[0]; d_trans[3 * b + 1] = S[1]; d_trans[3 * b + 2] = S[2];
  } else {
    float t0 = td[0] * c.s[0] + c.m[0], t1 = td[1] * c.s[1] + c.m[1], t2 = td[2] * c.s[2] + c.m[2];
    float z2 = Ts[2] / expf(t2);
    if (c.rot_coord == ROT_CAMERA_NEW) {
      d_trans[3 * b] = S[0] * (c.s[0] * Ts[2]);
      d_trans[3 * b + 1] = S[1] * (c.s[1] * Ts[2]);
      d_trans[3 * b + 2] = S[2] * (-c.s[2] * z2);
    } else {
      float share = -c.s[2] * z2;
      d_trans[3 * b] = S[0] * (c.s[0] * z2);
      d_trans[3 * b + 1] = S[1] * (c.s[1] * z2);
      d_trans[3 * b + 2] = S[0] * (share * (t0 + Ts[0] / Ts[2])) + S[1] * (share * (t1 + Ts[1] / Ts[2])) + S[2] * (-c.s[2] * z2);
    }
  }
  // ---- rotation: Rm_tgt_diff = grad * P^T, chained through rot_coord, then quat2mat_backward :256-327
  float Dt[9] = {S[3], S[4], S[5], S[6], S[7], S[8], S[9], S[10], S[11]};
  float D[9];
  if (c.rot_coord == ROT_NAIVE) {
    for (int i = 0; i < 9; ++i) D[i] = Dt[i];
  } else {
    float RsT[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) RsT[3 * j + i] = ps[4 * i + j];
    const float* A = (c.rot_coord == ROT_MODEL) ? RsT : Dt;
    const float* Bm = (c.rot_coord == ROT_MODEL) ? Dt : RsT;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) D[3 * i + j] = A[3 * i] * Bm[j] + A[3 * i + 1] * Bm[3 + j] + A[3 * i + 2] * Bm[6 + j];
  }
  const float* q = rot + 4 * b;
  float w = q[0], x = q[1], y = q[2], z = q[3];
  float Nq = w * w + x * x + y * y + z * z;
  float* dq = d_rot + 4 * b;
  if (!((Nq - 1.f > -1e-4f) && (Nq - 1.f < 1e-4f))) {
    dq[0] = dq[1] = dq[2] = dq[3] = 0.f;
    return;
  }
  float Ns = sqrtf(Nq);
  float w_ = w / Ns, x_ = x / Ns, y_ = y / Ns, z_ = z / Ns;
  const float s = 2.0f;
  float wd = (-z_ * D[1] + y_ * D[2] + z_ * D[3] - x_ * D[5] - y_ * D[6] + x_ * D[7]) * s;
  float xd = (y_ * D[1] + z_ * D[2] + y_ * D[3] - 2 * x_ * D[4] - w_ * D[5] + z_ * D[6] + w_ * D[7] - 2 * x_ * D[8]) * s;
  float yd = (-2 * y_ * D[0] + x_ * D[1] + w_ * D[2] + x_ * D[3] + z_ * D[5] - w_ * D[6] + z_ * D[7] - 2 * y_ * D[8]) * s;
  float zd = (-2 * z_ * D[0] - w_ * D[1] + x_ * D[2] + w_ * D[3] - 2 * z_ * D[4] + y_ * D[5] + x_ * D[6] + y_ * D[7]) * s;
  float share = Ns * Ns * Ns * (w * wd + x * xd + y * yd + z * zd);
  dq[0] = Ns * wd - w * share;
  dq[1] = Ns * xd - x * share;
  dq[2] = Ns * yd - y * share;
  dq[3] = Ns * zd - z * share;
}

static int parse_rot(int rot_coord) { return (rot_coord >= 0 && rot_coord <= 3) ? rot_coord : -1; }

}  // namespace dim

using namespace dim;

extern "C" {

int dim_se3_compose(const float* pose_src, const float* se3, float* pose_out, double* pose_out_f64, int B, int rot_coord,
                    const float* T_means3, const float* T_stds3, void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(pose_src && se3 && pose_out && T_means3 && T_stds3, "null pointer");
  DIM_REQUIRE(parse_rot(rot_coord) >= 0, "unknown rot_coord %d", rot_coord);
  hipLaunchKernelGGL(se3_compose_kernel<false>, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), pose_src, se3, pose_out,
                     pose_out_f64, B, rot_coord, (double)T_means3[0], (double)T_means3[1], (double)T_means3[2], (double)T_stds3[0],
                     (double)T_stds3[1], (double)T_stds3[2]);
  return check_launch("se3_compose");
}

int dim_se3_compose_euler(const float* pose_src, const float* euler_trans6, float* pose_out, double* pose_out_f64, int B, int rot_coord,
                          const float* T_means3, const float* T_stds3, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(pose_src && euler_trans6 && pose_out && T_means3 && T_stds3, "null pointer");
  DIM_REQUIRE(parse_rot(rot_coord) >= 0, "unknown rot_coord %d", rot_coord);
  hipLaunchKernelGGL(se3_compose_kernel<true>, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), pose_src, euler_trans6, pose_out,
                     pose_out_f64, B, rot_coord, (double)T_means3[0], (double)T_means3[1], (double)T_means3[2], (double)T_stds3[0],
                     (double)T_stds3[1], (double)T_stds3[2]);
  return check_launch("se3_compose_euler");
}

int dim_se3_delta_euler(const float* pose_src, const float* pose_tgt, float* rot_euler, float* trans, int B, int rot_coord,
                        const float* T_means3, const float* T_stds3, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(pose_src && pose_tgt && rot_euler && trans && T_means3 && T_stds3, "null pointer");
  DIM_REQUIRE(parse_rot(rot_coord) >= 0, "unknown rot_coord %d", rot_coord);
  hipLaunchKernelGGL(se3_delta_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), pose_src, pose_tgt, nullptr, nullptr, rot_euler,
                     trans, B, rot_coord, (double)T_means3[0], (double)T_means3[1], (double)T_means3[2], (double)T_stds3[0],
                     (double)T_stds3[1], (double)T_stds3[2]);
  return check_launch("se3_delta_euler");
}

int dim_se3_delta(const float* pose_src, const float* pose_tgt, float* rot_quat, float* trans, int B, int rot_coord,
                  const float* T_means3, const float* T_stds3, void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(pose_src && pose_tgt && rot_quat && trans && T_means3 && T_stds3, "null pointer");
  DIM_REQUIRE(parse_rot(rot_coord) >= 0, "unknown rot_coord %d", rot_coord);
  hipLaunchKernelGGL(se3_delta_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), pose_src, pose_tgt, rot_quat, nullptr, nullptr, trans,
                     B, rot_coord, (double)T_means3[0], (double)T_means3[1], (double)T_means3[2], (double)T_stds3[0],
                     (double)T_stds3[1], (double)T_stds3[2]);
  return check_launch("se3_delta");
}

int dim_se3_delta_matrix(const float* pose_src, const float* pose_tgt, float* rot_mat, float* trans, int B, int rot_coord,
                         const float* T_means3, const float* T_stds3, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(pose_src && pose_tgt && rot_mat && trans && T_means3 && T_stds3, "null pointer");
  DIM_REQUIRE(parse_rot(rot_coord) >= 0, "unknown rot_coord %d", rot_coord);
  hipLaunchKernelGGL(se3_delta_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), pose_src, pose_tgt, nullptr, rot_mat, nullptr, trans,
                     B, rot_coord, (double)T_means3[0], (double)T_means3[1], (double)T_means3[2], (double)T_stds3[0],
                     (double)T_stds3[1], (double)T_stds3[2]);
  return check_launch("se3_delta_matrix");
}

int dim_pose_to_KT(const float* pose_src, const float* pose_tgt, const float* K9, float* KT, int B, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(pose_src && pose_tgt && K9 && KT, "null pointer");
  hipLaunchKernelGGL(pose_to_KT_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), pose_src, pose_tgt, (double)K9[0],
                     (double)K9[1], (double)K9[2], (double)K9[3], (double)K9[4], (double)K9[5], (double)K9[6], (double)K9[7], (double)K9[8],
                     KT, B);
  return check_launch("pose_to_KT");
}

int dim_zoom_trans(const float* zoom_factor, const float* in, float* out, int B, int mode, void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(zoom_factor && in && out, "null pointer");
  DIM_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (copy), 1 (divide) or 2 (multiply)");
  hipLaunchKernelGGL(zoom_trans_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), zoom_factor, in, out, B, mode);
  return check_launch("zoom_trans");
}

static T3DConst make_t3d(int rot_coord, const float* m, const float* s) {
  T3DConst c;
  for (int i = 0; i < 3; ++i) { c.m[i] = m[i]; c.s[i] = s[i]; }
  c.rot_coord = rot_coord;
  return c;
}

int dim_transform3d_fwd(const float* points, const float* rot, const float* trans, const float* pose_src, float* out, int B,
                        int Npts, int rot_coord, const float* T_means3, const float* T_stds3, void* stream) {
  if (B == 0 || Npts == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(points && rot && trans && pose_src && out && T_means3 && T_stds3, "null pointer");
  DIM_REQUIRE(parse_rot(rot_coord) >= 0, "unknown rot_coord %d", rot_coord);
  hipLaunchKernelGGL(transform3d_fwd_kernel, dim3(ceil_div(Npts, 256), B), dim3(256), 0, as_stream(stream), points, rot, trans,
                     pose_src, out, Npts, make_t3d(rot_coord, T_means3, T_stds3));
  return check_launch("transform3d_fwd");
}

int dim_transform3d_bwd(const float* out_grad, const float* points, const float* rot, const float* trans, const float* pose_src,
                        float* d_rot, float* d_trans, int B, int Npts, int rot_coord, const float* T_means3, const float* T_stds3,
                        void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(out_grad && points && rot && trans && pose_src && d_rot && d_trans && T_means3 && T_stds3, "null pointer");
  DIM_REQUIRE(parse_rot(rot_coord) >= 0, "unknown rot_coord %d", rot_coord);
  hipLaunchKernelGGL(transform3d_bwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), out_grad, points, rot, trans, pose_src,
                     d_rot, d_trans, Npts, make_t3d(rot_coord, T_means3, T_stds3));
  return check_launch("transform3d_bwd");
}

}  // extern "C"
