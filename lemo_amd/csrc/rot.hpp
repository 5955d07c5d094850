// Rotation-representation conversions on the fitting path, forward and analytic backward.
//   6-D -> R (Gram-Schmidt)      utils/utils.py:63-70 ; vposer_smpl.py:53-62
//   R -> quaternion -> axis-angle torchgeometry==0.1.2 rotation_matrix_to_angle_axis
//                                 (call sites utils/utils.py:80, vposer_smpl.py:160)
//   axis-angle -> R (Rodrigues)   human_body_prior/body_model/lbs.py:166-193 (angle = |aa + 1e-8|)
// All matrices are row-major float[9]: R[3*i+j].
#pragma once
#include "common.hpp"

namespace lemo {

// ---- 6-D -> R ---------------------------------------------------------------------------------
// x6 viewed as [3][2]: a1 = (x0,x2,x4), a2 = (x1,x3,x5); R columns = (b1,b2,b3).
__device__ __forceinline__ void rot6d_fwd(const float* x6, float* R) {
  const V3 a1 = v3(x6[0], x6[2], x6[4]), a2 = v3(x6[1], x6[3], x6[5]);
  const float n1 = fmaxf(sqrtf(dot(a1, a1)), 1e-12f);
  const V3 b1 = (1.f / n1) * a1;
  const float d = dot(b1, a2);
  const V3 u = a2 - d * b1;
  const float n2 = fmaxf(sqrtf(dot(u, u)), 1e-12f);
  const V3 b2 = (1.f / n2) * u;
  const V3 b3 = cross(b1, b2);
  R[0] = b1.x; R[1] = b2.x; R[2] = b3.x;
  R[3] = b1.y; R[4] = b2.y; R[5] = b3.y;
  R[6] = b1.z; R[7] = b2.z; R[8] = b3.z;
}

__device__ __forceinline__ void rot6d_bwd(const float* x6, const float* dR, float* dx6) {
  const V3 a1 = v3(x6[0], x6[2], x6[4]), a2 = v3(x6[1], x6[3], x6[5]);
  const float n1 = fmaxf(sqrtf(dot(a1, a1)), 1e-12f);
  const V3 b1 = (1.f / n1) * a1;
  const float d = dot(b1, a2);
  const V3 u = a2 - d * b1;
  const float n2 = fmaxf(sqrtf(dot(u, u)), 1e-12f);
  const V3 b2 = (1.f / n2) * u;
  V3 g1 = v3(dR[0], dR[3], dR[6]), g2 = v3(dR[1], dR[4], dR[7]);
  const V3 g3 = v3(dR[2], dR[5], dR[8]);
  // b3 = b1 x b2
  g1 = g1 + cross(b2, g3);
  g2 = g2 + cross(g3, b1);
  // b2 = u / |u|
  const V3 du = (1.f / n2) * (g2 - dot(g2, b2) * b2);
  // u = a2 - (b1.a2) b1
  const float dub1 = dot(du, b1);
  const V3 da2 = du - dub1 * b1;
  g1 = g1 - (dub1 * a2 + d * du);
  // b1 = a1 / |a1|
  const V3 da1 = (1.f / n1) * (g1 - dot(g1, b1) * b1);
  dx6[0] = da1.x; dx6[2] = da1.y; dx6[4] = da1.z;
  dx6[1] = da2.x; dx6[3] = da2.y; dx6[5] = da2.z;
}

// ---- R -> axis-angle through the tgm 0.1.2 quaternion ------------------------------------------
// Branch selection (eps = 1e-6):  d2 = R22 < eps ; d0_d1 = R00 > R11 ; d0_nd1 = R00 < -R11
//   c0 = d2 & d0_d1 ; c1 = d2 & !d0_d1 ; c2 = !d2 & d0_nd1 ; c3 = !d2 & !d0_nd1
// Each branch: q = 0.5 * qc / sqrt(tc) with qc linear in R (w first).
__device__ __forceinline__ int quat_branch(const float* R) {
  const bool d2 = R[8] < 1e-6f, d01 = R[0] > R[4], d0n1 = R[0] < -R[4];
  return d2 ? (d01 ? 0 : 1) : (d0n1 ? 2 : 3);
}

__device__ __forceinline__ void quat_candidates(const float* R, int br, float* qc, float* tc) {
  const float R00 = R[0], R01 = R[1], R02 = R[2], R10 = R[3], R11 = R[4], R12 = R[5], R20 = R[6], R21 = R[7], R22 = R[8];
  if (br == 0) {
    const float t = 1.f + R00 - R11 - R22;
    qc[0] = R21 - R12; qc[1] = t; qc[2] = R10 + R01; qc[3] = R02 + R20; *tc = t;
  } else if (br == 1) {
    const float t = 1.f - R00 + R11 - R22;
    qc[0] = R02 - R20; qc[1] = R10 + R01; qc[2] = t; qc[3] = R21 + R12; *tc = t;
  } else if (br == 2) {
    const float t = 1.f - R00 - R11 + R22;
    qc[0] = R10 - R01; qc[1] = R02 + R20; qc[2] = R21 + R12; qc[3] = t; *tc = t;
  } else {
    const float t = 1.f + R00 + R11 + R22;
    qc[0] = t; qc[1] = R21 - R12; qc[2] = R02 - R20; qc[3] = R10 - R01; *tc = t;
  }
}

__device__ __forceinline__ void rotmat_to_aa_fwd(const float* R, float* aa) {
  float qc[4], tc;
  quat_candidates(R, quat_branch(R), qc, &tc);
  const float r = 0.5f / sqrtf(tc);
  const float w = qc[0] * r, q1 = qc[1] * r, q2 = qc[2] * r, q3 = qc[3] * r;
  const float s2 = q1 * q1 + q2 * q2 + q3 * q3;
  const float s = sqrtf(s2);
  const float two_theta = 2.f * (w < 0.f ? atan2f(-s, -w) : atan2f(s, w));
  const float k = s2 > 0.f ? two_theta / s : 2.f;
  aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
}

// d(aa) -> dR  (recomputes the forward quantities)
__device__ __forceinline__ void rotmat_to_aa_bwd(const float* R, const float* g, float* dR) {
  float qc[4], tc;
  const int br = quat_branch(R);
  quat_candidates(R, br, qc, &tc);
  const float rs = 1.f / sqrtf(tc);
  const float r = 0.5f * rs;
  const float w = qc[0] * r, q1 = qc[1] * r, q2 = qc[2] * r, q3 = qc[3] * r;
  const float s2 = q1 * q1 + q2 * q2 + q3 * q3;
  const float s = sqrtf(s2);
  float dq[4] = {0.f, 0.f, 0.f, 0.f};
  if (s2 > 0.f) {
    const float t = (w < 0.f ? atan2f(-s, -w) : atan2f(s, w));
    const float k = 2.f * t / s;
    dq[1] = k * g[0]; dq[2] = k * g[1]; dq[3] = k * g[2];
    const float dk = g[0] * q1 + g[1] * q2 + g[2] * q3;
    const float dt = 2.f / s * dk;
    float ds = -2.f * t / s2 * dk;
    const float n2 = s2 + w * w;
    ds += w / n2 * dt;
    dq[0] = -s / n2 * dt;
    const float dss = ds / s;
    dq[1] += q1 * dss; dq[2] += q2 * dss; dq[3] += q3 * dss;
  } else {
    dq[1] = 2.f * g[0]; dq[2] = 2.f * g[1]; dq[3] = 2.f * g[2];
  }
  // q_j = 0.5 * qc_j / sqrt(tc)
  float dqc[4];
  float dtc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { dqc[i] = r * dq[i]; dtc += dq[i] * qc[i]; }
  dtc *= -0.25f * rs * rs * rs;
#pragma unroll
  for (int i = 0; i < 9; ++i) dR[i] = 0.f;
  // index helpers: R00=0 R01=1 R02=2 R10=3 R11=4 R12=5 R20=6 R21=7 R22=8
  if (br == 0) {
    const float dt = dtc + dqc[1];
    dR[0] += dt; dR[4] -= dt; dR[8] -= dt;
    dR[7] += dqc[0]; dR[5] -= dqc[0];
    dR[3] += dqc[2]; dR[1] += dqc[2];
    dR[2] += dqc[3]; dR[6] += dqc[3];
  } else if (br == 1) {
    const float dt = dtc + dqc[2];
    dR[0] -= dt; dR[4] += dt; dR[8] -= dt;
    dR[2] += dqc[0]; dR[6] -= dqc[0];
    dR[3] += dqc[1]; dR[1] += dqc[1];
    dR[7] += dqc[3]; dR[5] += dqc[3];
  } else if (br == 2) {
    const float dt = dtc + dqc[3];
    dR[0] -= dt; dR[4] -= dt; dR[8] += dt;
    dR[3] += dqc[0]; dR[1] -= dqc[0];
    dR[2] += dqc[1]; dR[6] += dqc[1];
    dR[7] += dqc[2]; dR[5] += dqc[2];
  } else {
    const float dt = dtc + dqc[0];
    dR[0] += dt; dR[4] += dt; dR[8] += dt;
    dR[7] += dqc[1]; dR[5] -= dqc[1];
    dR[2] += dqc[2]; dR[6] -= dqc[2];
    dR[3] += dqc[3]; dR[1] -= dqc[3];
  }
}

// ---- Rodrigues (lbs.py:166-193) ---------------------------------------------------------------
__device__ __forceinline__ void rodrigues_fwd(const float* aa, float* R) {
  const float ex = aa[0] + 1e-8f, ey = aa[1] + 1e-8f, ez = aa[2] + 1e-8f;
  const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
  const float rx = aa[0] / ang, ry = aa[1] / ang, rz = aa[2] / ang;
  const float sn = sinf(ang), c1 = 1.f - cosf(ang);
  // K = [[0,-rz,ry],[rz,0,-rx],[-ry,rx,0]] ; KK = K*K
  const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) {
      float kk = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) kk += K[3 * i + k] * K[3 * k + jx];
      R[3 * i + jx] = (i == jx ? 1.f : 0.f) + sn * K[3 * i + jx] + c1 * kk;
    }
}

__device__ __forceinline__ void rodrigues_bwd(const float* aa, const float* dR, float* daa) {
  const float ex = aa[0] + 1e-8f, ey = aa[1] + 1e-8f, ez = aa[2] + 1e-8f;
  const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
  const float inv = 1.f / ang;
  const float rx = aa[0] * inv, ry = aa[1] * inv, rz = aa[2] * inv;
  const float sn = sinf(ang), cs = cosf(ang), c1 = 1.f - cs;
  const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
  float KK[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) {
      float kk = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) kk += K[3 * i + k] * K[3 * k + jx];
      KK[3 * i + jx] = kk;
    }
  float dsn = 0.f, dc1 = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) { dsn += dR[i] * K[i]; dc1 += dR[i] * KK[i]; }
  // dK = sn*dR + c1*(dR K^T + K^T dR)
  float dK[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) a += dR[3 * i + k] * K[3 * jx + k] + K[3 * k + i] * dR[3 * k + jx];
      dK[3 * i + jx] = sn * dR[3 * i + jx] + c1 * a;
    }
  const float gx = dK[7] - dK[5], gy = dK[2] - dK[6], gz = dK[3] - dK[1];
  float dang = dsn * cs + dc1 * sn;
  dang -= (gx * aa[0] + gy * aa[1] + gz * aa[2]) * inv * inv;
  daa[0] = gx * inv + dang * ex * inv;
  daa[1] = gy * inv + dang * ey * inv;
  daa[2] = gz * inv + dang * ez * inv;
}

}  // namespace lemo
