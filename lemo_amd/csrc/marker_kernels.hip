// Marker-image encode / decode around the fitting loop: device versions of the two numpy
// functions the reference runs on the host per clip,
//     reconstruct_global_body   utils/utils.py:184-203   (trajectory integration: local -> global markers)
//     get_local_markers_4chan   utils/utils.py:209-265   (global -> heading-normalised local + (dx, dz, dr))
// so that the clip never leaves the GPU between the infilling network and the fitting engine.
// Every rotation in both functions is about the up axis: the reference's sequential quaternion products
// (utils/Quaternions.py:93-118, 402-407) are prefix sums of angles, done here as log-step scans in float64
// like the numpy code; one block per clip (T <= 256 frames).
#include "kernels.hpp"

namespace lemo {

#define MK_T 256

// inclusive Hillis-Steele scan over s[0..n) (n <= MK_T, blockDim = MK_T); all threads call
__device__ __forceinline__ void scan_inclusive(double* s, int n) {
  const int t = threadIdx.x;
  for (int off = 1; off < n; off <<= 1) {
    const double add = (t < n && t >= off) ? s[t - off] : 0.0;
    __syncthreads();
    if (t < n) s[t] += add;
    __syncthreads();
  }
}

// q = (cos a/2, 0, sin a/2, 0) applied as q v q^-1: rotation about +y by a
__device__ __forceinline__ void rot_y(double a, double x, double z, double& xo, double& zo) {
  const double c = cos(a), s = sin(a);
  xo = c * x + s * z;
  zo = -s * x + c * z;
}

__global__ void __launch_bounds__(MK_T)
reconstruct_global_body_kernel(const float* __restrict__ in, int T, int J, double rot0, float* __restrict__ out) {
  __shared__ double ang[MK_T], tx[MK_T], tz[MK_T];
  const int t = threadIdx.x, E = J + 2;
  // frame i is rotated by R_y(theta_i), theta_i = -rot0 - sum_{k<i} r_k ; after frame i the heading becomes
  // theta_{i+1} and the translation advances by R_y(theta_{i+1}) (x_i, 0, z_i)          (utils.py:193-200)
  if (t < T) ang[t] = -(double)in[((size_t)t * E + J + 1) * 3 + 2];
  __syncthreads();
  scan_inclusive(ang, T);                                   // ang[i] = -sum_{k<=i} r_k
  double th_next = 0.0;
  if (t < T) {
    th_next = ang[t] - rot0;                                // theta_{t+1}
    const double x = in[((size_t)t * E + J + 1) * 3 + 0], z = in[((size_t)t * E + J + 1) * 3 + 1];
    double dx, dz;
    rot_y(th_next, x, z, dx, dz);
    tx[t] = dx; tz[t] = dz;
  }
  __syncthreads();
  scan_inclusive(tx, T);
  scan_inclusive(tz, T);                                    // t?[i] = translation after frame i
  for (int w = t; w < T * J; w += MK_T) {
    const int i = w / J, m = w - i * J;
    const double th = (i == 0 ? 0.0 : ang[i - 1]) - rot0;
    const double ox = i == 0 ? 0.0 : tx[i - 1], oz = i == 0 ? 0.0 : tz[i - 1];
    const float* p = in + ((size_t)i * E + 1 + m) * 3;      // slot 0 is the (dropped) reference joint
    // (x, y, z) -> swap y/z -> rotate about the middle axis, shift x and the last axis -> swap back
    double rx, rz;
    rot_y(th, (double)p[0], (double)p[1], rx, rz);
    float* o = out + ((size_t)i * J + m) * 3;
    o[0] = (float)(rx + ox); o[1] = (float)(rz + oz); o[2] = p[2];
  }
}

__global__ void __launch_bounds__(MK_T)
local_markers_4chan_kernel(const float* __restrict__ body, const float* __restrict__ contact, int T, int M1,
                           int sdr_l, int sdr_r, int hip_l, int hip_r, float* __restrict__ image, double* __restrict__ rot0_out) {
  __shared__ double fx[MK_T], fz[MK_T];                     // forward direction (x, z) before / after the filter
  __shared__ double qw[MK_T], qy[MK_T];                     // per-frame heading quaternion (w, 0, y, 0)
  __shared__ float redm[MK_T / 64];
  __shared__ float s_min;
  const int t = threadIdx.x;
  // ---- "put on floor": min of the up coordinate over the whole clip (utils.py:214)
  float mn = 3.4e38f;
  for (int w = t; w < T * M1; w += MK_T) mn = fminf(mn, body[(size_t)w * 3 + 2]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
  if ((t & 63) == 0) redm[t >> 6] = mn;
  __syncthreads();
  if (t == 0) { float m = redm[0]; for (int i = 1; i < MK_T / 64; ++i) m = fminf(m, redm[i]); s_min = m; }
  __syncthreads();
  const double floor_z = s_min;
  // ---- forward direction per frame: across = (sdr_r - sdr_l) + (hip_r - hip_l), forward = across x up
  // (in the swapped (x, up, y) frame: forward = (-across_y, 0, across_x))              (utils.py:228-236)
  if (t < T) {
    const float* b = body + (size_t)t * M1 * 3;
    const double ax = ((double)b[sdr_r * 3] - b[sdr_l * 3]) + ((double)b[hip_r * 3] - b[hip_l * 3]);
    const double au = ((double)b[sdr_r * 3 + 2] - b[sdr_l * 3 + 2]) + ((double)b[hip_r * 3 + 2] - b[hip_l * 3 + 2]);
    const double ay = ((double)b[sdr_r * 3 + 1] - b[sdr_l * 3 + 1]) + ((double)b[hip_r * 3 + 1] - b[hip_l * 3 + 1]);
    const double n = sqrt(ax * ax + au * au + ay * ay);
    fx[t] = -(ay / n); fz[t] = ax / n;
  }
  __syncthreads();
  // gaussian_filter1d(sigma = 20, truncate 4 -> radius 80, mode 'nearest') over time, then normalise
  double gx = 0.0, gz = 0.0;
  if (t < T) {
    double wsum = 0.0;
    for (int k = -80; k <= 80; ++k) wsum += exp(-0.5 * (double)(k * k) / 400.0);
    for (int k = -80; k <= 80; ++k) {
      const double w = exp(-0.5 * (double)(k * k) / 400.0) / wsum;
      int i = t + k; i = i < 0 ? 0 : (i > T - 1 ? T - 1 : i);
      gx += w * fx[i]; gz += w * fz[i];
    }
    const double n = sqrt(gx * gx + gz * gz);               // the middle component is exactly 0
    gx /= n; gz /= n;
    // Quaternions.between(forward, (0,0,1)) (Quaternions.py:396-399): (w, a) = (|f| + f.z, f x target), normalised
    double w = sqrt(gx * gx + gz * gz) + gz, y = -gx;
    const double qn = sqrt(w * w + y * y);
    qw[t] = w / qn; qy[t] = y / qn;
  }
  __syncthreads();
  const int d = 3 * M1 + 4, Tm = T - 1;
  // ---- channel 0: local, heading-normalised pelvis + markers, then the contact labels
  for (int w = t; w < Tm * M1; w += MK_T) {
    const int i = w / M1, m = w - i * M1;
    const float* p = body + ((size_t)i * M1 + m) * 3, *r = body + (size_t)i * M1 * 3;
    const double x = (double)p[0] - r[0], y = (double)p[1] - r[1], u = (double)p[2] - floor_z;
    // rotate (x, u, y) about the middle axis by the frame's heading quaternion: angle a with cos a/2 = qw, sin a/2 = qy
    const double c = qw[i] * qw[i] - qy[i] * qy[i], s = 2.0 * qw[i] * qy[i];
    const double rx = c * x + s * y, ry = -s * x + c * y;
    float* o = image + (size_t)i * d + 3 * m;
    o[0] = (float)rx; o[1] = (float)ry; o[2] = (float)u;
  }
  for (int w = t; w < Tm * 4; w += MK_T) image[(size_t)(w >> 2) * d + 3 * M1 + (w & 3)] = contact[w];
  // ---- channels 1-3: root velocity in the rotated frame and heading change, repeated over d
  if (t < Tm) {
    const float* r0 = body + (size_t)t * M1 * 3, *r1 = r0 + (size_t)M1 * 3;
    const double vx = (double)r1[0] - r0[0], vy = (double)r1[1] - r0[1];
    const double c = qw[t + 1] * qw[t + 1] - qy[t + 1] * qy[t + 1], s = 2.0 * qw[t + 1] * qy[t + 1];
    const double gvx = c * vx + s * vy, gvy = -s * vx + c * vy;
    // rot[t+1] * -rot[t] is a rotation about the up axis; Pivots.from_quaternions = atan2 of its image of (0,0,1)
    const double pw = qw[t + 1] * qw[t] + qy[t + 1] * qy[t], py = qy[t + 1] * qw[t] - qw[t + 1] * qy[t];
    const double gr = atan2(2.0 * pw * py, pw * pw - py * py);
    for (int k = 0; k < d; ++k) {
      image[((size_t)1 * Tm + t) * d + k] = (float)gvx;
      image[((size_t)2 * Tm + t) * d + k] = (float)gvy;
      image[((size_t)3 * Tm + t) * d + k] = (float)gr;
    }
  }
  if (t == 0) rot0_out[0] = atan2(2.0 * qw[0] * qy[0], qw[0] * qw[0] - qy[0] * qy[0]);
}

int reconstruct_global_body(const float* in, int T, int J, double rot0, float* out, hipStream_t s) {
  if (T < 1 || T > MK_T || J < 1) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(reconstruct_global_body_kernel, dim3(1), dim3(MK_T), 0, s, in, T, J, rot0, out);
  return (int)hipGetLastError();
}

int local_markers_4chan(const float* body, const float* contact, int T, int M1, float* image, double* rot0, hipStream_t s) {
  if (T < 2 || T > MK_T || M1 < 59) return LEMO_ERR_SHAPE;   // direction markers 26/27/56/57 (+1 pelvis), utils.py:228
  hipLaunchKernelGGL(local_markers_4chan_kernel, dim3(1), dim3(MK_T), 0, s, body, contact, T, M1, 27, 57, 28, 58, image, rot0);
  return (int)hipGetLastError();
}

}  // namespace lemo
