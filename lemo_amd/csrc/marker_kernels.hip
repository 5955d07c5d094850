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
reconstruct_global_body_kernel(const float* __restrict__ in, int T, int J, double rot0_host, const double* __restrict__ rot0_dev,
                               float* __restrict__ out) {
  __shared__ double ang[MK_T], tx[MK_T], tz[MK_T];
  const int t = threadIdx.x, E = J + 2;
  const double rot0 = rot0_dev ? rot0_dev[0] : rot0_host;     // the pivot may live on the device (no host round trip)
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

// ---- decode of the infilling network's output (opt_amass_temp.py:273-325; twin fitting_temp_slide.py:895-940) ----------
// One launch per clip replaces: sigmoid -> {0,1} contact labels (:273-275), the reshuffle of the image rows into
// [T, traj + pelvis + 67 markers, 3] (:280-297), de-normalisation by the preprocess statistics (f64 arithmetic, f32
// store, like numpy's in-place assignment into the f32 array, :303-311), the reorder into reconstruct_global_body's
// input format (:317-319), the trajectory integration itself and the removal of the pelvis row (:320-321) -- and, for
// the PROX twin, the shift back from the floor and the rigid transform back to PROX world coordinates (:934-939).
//   rec   [d][T]     channel 0 of the network output, un-padded (d = 3 + 3 J67 + 4: pelvis, markers, 4 contact logits)
//   traj  [3][T]     row 0 of channels 1..3 of the (un-masked) input image: normalised dx, dz, dr
//   stats [2 d + 4]  doubles: Xmean_local[d], Xstd_local[d], Xmean_global_xy, Xstd_global_xy, Xmean_global_r, Xstd_global_r
//   rot0  [1]        double, device: rot_0_pivot of the clip
//   post  [13]       floats or NULL: z shift, then row-major M[9] and t[3] of  out = (p + (0,0,shift)) . M + t
__global__ void __launch_bounds__(MK_T)
decode_clip_kernel(const float* __restrict__ rec, const float* __restrict__ traj, const double* __restrict__ stats,
                   const double* __restrict__ rot0p, const float* __restrict__ post, int T, int J, float* __restrict__ lbl,
                   float* __restrict__ markers) {
  __shared__ double ang[MK_T], tx[MK_T], tz[MK_T];
  const int t = threadIdx.x, d = 3 * J + 4;                  // J = 1 pelvis + markers
  const double* mean = stats, *sd = stats + d;
  const double mxy = stats[2 * d], sxy = stats[2 * d + 1], mr = stats[2 * d + 2], sr = stats[2 * d + 3];
  const double rot0 = rot0p[0];
  for (int w = t; w < T * 4; w += MK_T) {                    // F.sigmoid(...) > 0.5 -> 1 else 0
    const int i = w >> 2, k = w & 3;
    const float s = 1.f / (1.f + expf(-rec[(size_t)(d - 4 + k) * T + i]));
    lbl[w] = s > 0.5f ? 1.f : 0.f;
  }
  float vx = 0.f, vz = 0.f;
  if (t < T) {
    vx = (float)((double)traj[t] * sxy + mxy);
    vz = (float)((double)traj[T + t] * sxy + mxy);
    ang[t] = -(double)(float)((double)traj[2 * T + t] * sr + mr);
  }
  __syncthreads();
  scan_inclusive(ang, T);
  if (t < T) {
    double dx, dz;
    rot_y(ang[t] - rot0, (double)vx, (double)vz, dx, dz);
    tx[t] = dx; tz[t] = dz;
  }
  __syncthreads();
  scan_inclusive(tx, T);
  scan_inclusive(tz, T);
  const int Jm = J - 1;
  for (int w = t; w < T * Jm; w += MK_T) {
    const int i = w / Jm, m = w - i * Jm + 1;                // skip the pelvis (row 0 of the local body)
    const double th = (i == 0 ? 0.0 : ang[i - 1]) - rot0;
    const double ox = i == 0 ? 0.0 : tx[i - 1], oz = i == 0 ? 0.0 : tz[i - 1];
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = (float)((double)rec[(size_t)(3 * m + c) * T + i] * sd[3 * m + c] + mean[3 * m + c]);
    double rx, rz;
    rot_y(th, (double)p[0], (double)p[1], rx, rz);
    float gx = (float)(rx + ox), gy = (float)(rz + oz), gz = p[2];
    if (post) {
      gz += post[0];
      const float* M = post + 1, *tt = post + 10;
      const float ax = gx * M[0] + gy * M[3] + gz * M[6] + tt[0];
      const float ay = gx * M[1] + gy * M[4] + gz * M[7] + tt[1];
      const float az = gx * M[2] + gy * M[5] + gz * M[8] + tt[2];
      gx = ax; gy = ay; gz = az;
    }
    float* o = markers + ((size_t)i * Jm + (m - 1)) * 3;
    o[0] = gx; o[1] = gy; o[2] = gz;
  }
}

int decode_clip(const float* rec, const float* traj, const double* stats, const double* rot0, const float* post, int T, int J,
                float* lbl, float* markers, hipStream_t s) {
  if (T < 1 || T > MK_T || J < 2) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(decode_clip_kernel, dim3(1), dim3(MK_T), 0, s, rec, traj, stats, rot0, post, T, J, lbl, markers);
  return (int)hipGetLastError();
}

int reconstruct_global_body(const float* in, int T, int J, double rot0, float* out, hipStream_t s, const double* rot0_dev) {
  if (T < 1 || T > MK_T || J < 1) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(reconstruct_global_body_kernel, dim3(1), dim3(MK_T), 0, s, in, T, J, rot0, rot0_dev, out);
  return (int)hipGetLastError();
}

int local_markers_4chan(const float* body, const float* contact, int T, int M1, float* image, double* rot0, hipStream_t s) {
  if (T < 2 || T > MK_T || M1 < 59) return LEMO_ERR_SHAPE;   // direction markers 26/27/56/57 (+1 pelvis), utils.py:228
  hipLaunchKernelGGL(local_markers_4chan_kernel, dim3(1), dim3(MK_T), 0, s, body, contact, T, M1, 27, 57, 28, 58, image, rot0);
  return (int)hipGetLastError();
}

}  // namespace lemo
