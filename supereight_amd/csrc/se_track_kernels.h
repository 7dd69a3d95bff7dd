// the consumer side of the raycast -- image pyramid + ICP tracking
// (se_denseslam/src/preprocessing.cpp, tracking.cpp) on the device, so that vertex_ / normal_ never
// leave HBM.  Same arithmetic contract as the hot path (se_device.h).  Since r03 the whole ICP loop is device-resident, since r04
// with ONE launch per iteration (k_icp_iter: the previous iteration's final sums + 6x6 solve + SE3 exponential + pose update +
// convergence test as a redundant prologue of every workgroup, then track + reduce), the state in device memory, one host read per frame.
#pragma once
#include "se_device.h"

#define SE_TRACK_SEGMENTS 32   // summation order of the reduction, see k_icp_iter (r02: 16, r03: 128, r04: 32 -- 256 workgroups, each of which repeats the previous
                               // iteration's final sums and pose update in its prologue: 32 KB of partials per workgroup instead of 128 KB)
#define SE_TRACK_LANES 256

struct TrackData { int result; float error; float J[6]; };   // se_denseslam/include/se/commons.h:249-253

// bilateralFilterKernel (preprocessing.cpp:41-89), radius 2.  The reference's expf is its C library's;
// here it is the correctly rounded one (double exp, rounded once), which differs from a given libm
// in the last bit of a small fraction of arguments -- the one stage with a stated tolerance.
struct Gauss5 { float g[5]; };
__global__ void k_bilateral_filter(float* __restrict__ out, const float* __restrict__ in, int width, int height, Gauss5 G, float e_d) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= width || y >= height) return;
  const int pos = x + y * width;
  const float center = in[pos];
  if (center == 0) { out[pos] = 0; return; }
  const float e_d_squared_2 = e_d * e_d * 2;
  float sum = 0.0f, t = 0.0f;
#pragma unroll
  for (int i = -2; i <= 2; ++i)
#pragma unroll
    for (int j = -2; j <= 2; ++j) {
      const int ux = min(max(x + i, 0), width - 1), uy = min(max(y + j, 0), height - 1);
      const float curPix = in[ux + uy * width];
      if (curPix > 0) {
        const float mod = (curPix - center) * (curPix - center);
        const float factor = G.g[i + 2] * G.g[j + 2] * (float)exp((double)(-mod / e_d_squared_2));
        t += factor * curPix;
        sum += factor;
      }
    }
  out[pos] = t / sum;
}

// halfSampleRobustImageKernel (preprocessing.cpp:190-226)
__global__ void k_half_sample(float* __restrict__ out, int ow, int oh, const float* __restrict__ in, int iw, float e_d, int r) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= ow || y >= oh) return;
  const int cx = 2 * x, cy = 2 * y;
  float sum = 0.0f, t = 0.0f;
  const float center = in[cx + cy * iw];
  for (int i = -r + 1; i <= r; ++i)
    for (int j = -r + 1; j <= r; ++j) {
      const int ux = min(max(cx + j, 0), 2 * ow - 1), uy = min(max(cy + i, 0), 2 * oh - 1);
      const float current = in[ux + uy * iw];
      if (fabsf(current - center) < e_d) { sum += 1.0f; t += current; }
    }
  out[x + y * ow] = t / sum;
}

// scaled_depth_[0..2] in ONE launch (r04): the copy of float_depth_ into level 0 (DenseSLAMSystem.cpp:149-152) and the two
// halfSampleRobustImageKernel passes with r = 1 (DenseSLAMSystem.cpp:154-159; preprocessing.cpp:190-226).  A thread owns a 4x4 block of level 0:
// it stores it, forms the block's four level-1 pixels and from those its one level-2 pixel -- per output pixel the same values in the
// same order as k_half_sample (window rows outer, columns inner; centre = the window's first pixel), so the images are bit-identical;
// three launches and two kernel boundaries of an ICP frame's host-bound preamble become one.  l1 / l2 may be null (fewer levels),
// `store0` = 0 when level 0 is already in place (the bilateral filter wrote it).
__device__ __forceinline__ float se_half_sample4(float a00, float a01, float a10, float a11, float e_d) {
  const float center = a00;
  float sum = 0.0f, t = 0.0f;
  if (fabsf(a00 - center) < e_d) { sum += 1.0f; t += a00; }
  if (fabsf(a01 - center) < e_d) { sum += 1.0f; t += a01; }
  if (fabsf(a10 - center) < e_d) { sum += 1.0f; t += a10; }
  if (fabsf(a11 - center) < e_d) { sum += 1.0f; t += a11; }
  return t / sum;
}
__global__ __launch_bounds__(256) void k_depth_pyramid(float* __restrict__ l0, float* __restrict__ l1, float* __restrict__ l2, const float* __restrict__ in,
                                                        int W, int H, float e_d, int store0) {
  const int bx = blockIdx.x * blockDim.x + threadIdx.x, by = blockIdx.y;
  const int x0 = 4 * bx, y0 = 4 * by;
  if (x0 >= W || y0 >= H) return;
  float v[4][4];
  if ((W & 3) == 0 && y0 + 3 < H) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 q = *reinterpret_cast<const float4*>(in + x0 + (size_t)(y0 + i) * W);
      v[i][0] = q.x; v[i][1] = q.y; v[i][2] = q.z; v[i][3] = q.w;
      if (store0) *reinterpret_cast<float4*>(l0 + x0 + (size_t)(y0 + i) * W) = q;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool inside = x0 + j < W && y0 + i < H;
        v[i][j] = inside ? in[x0 + j + (size_t)(y0 + i) * W] : 0.f;
        if (store0 && inside) l0[x0 + j + (size_t)(y0 + i) * W] = v[i][j];
      }
  }
  if (!l1) return;
  const int W1 = W >> 1, H1 = H >> 1;
  float h[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      h[a][b] = se_half_sample4(v[2 * a][2 * b], v[2 * a][2 * b + 1], v[2 * a + 1][2 * b], v[2 * a + 1][2 * b + 1], e_d);
      const int x1 = 2 * bx + b, y1 = 2 * by + a;
      if (x1 < W1 && y1 < H1) l1[x1 + (size_t)y1 * W1] = h[a][b];
    }
  if (!l2) return;
  const int W2 = W >> 2, H2 = H >> 2;
  if (bx < W2 && by < H2) l2[bx + (size_t)by * W2] = se_half_sample4(h[0][0], h[0][1], h[1][0], h[1][1], e_d);
}

// depth2vertexKernel (preprocessing.cpp:91-111): (depth * invK * Vector4f(x, y, 1, 0)).head<3>()
struct InvK { float m[12]; };
// all pyramid levels in one launch (blockIdx.z = level): the per-level launches were ~2 us of kernel and ~4 us of launch boundary each (k_vertex_normal_levels)
struct PyrLevels { float* vertex[8]; float* normal[8]; const float* depth[8]; int w[8], h[8]; InvK K[8]; };
__device__ __forceinline__ f3 ld3(const float* p, int i) { return {p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]}; }
__device__ __forceinline__ f3 f3_cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float f3_dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// depth2vertexKernel (preprocessing.cpp:91-111) + vertex2normalKernel<NegY> (preprocessing.cpp:113-159; an invalid pixel only gets .x = INVALID, as in
// the reference) of every level in ONE launch (r04): a pixel's vertex is a function of its own depth and
// coordinates, so the normal's four neighbour vertices are formed from the depth image (5 x 4 B instead of 5 x 12 B read back from the vertex image
// a launch later) by the same expression -- bit-identical to the two kernels in sequence, one launch and one boundary less per tracked frame.
__device__ __forceinline__ f3 se_depth_vertex(const float* __restrict__ depth, int x, int y, int W, const InvK& K) {
  const float d = depth[x + y * W];
  f3 v = {0.f, 0.f, 0.f};
  if (d > 0) {
    v.x = (((d * K.m[0]) * (float)x + (d * K.m[1]) * (float)y) + (d * K.m[2]) * 1.f) + (d * K.m[3]) * 0.f;
    v.y = (((d * K.m[4]) * (float)x + (d * K.m[5]) * (float)y) + (d * K.m[6]) * 1.f) + (d * K.m[7]) * 0.f;
    v.z = (((d * K.m[8]) * (float)x + (d * K.m[9]) * (float)y) + (d * K.m[10]) * 1.f) + (d * K.m[11]) * 0.f;
  }
  return v;
}
__global__ void k_vertex_normal_levels(PyrLevels L, int negy) {
  const int l = blockIdx.z, width = L.w[l], height = L.h[l];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= width || y >= height) return;
  const float* __restrict__ depth = L.depth[l];
  const InvK& K = L.K[l];
  const f3 center = se_depth_vertex(depth, x, y, width, K);
  float* v = L.vertex[l] + 3 * (size_t)(x + y * width);
  v[0] = center.x; v[1] = center.y; v[2] = center.z;
  float* o = L.normal[l] + 3 * (size_t)(x + y * width);
  if (center.z == 0.f) { o[0] = -2.f; return; }
  const int plx = max(x - 1, 0), prx = min(x + 1, width - 1);
  int puy, pdy;
  if (negy) { puy = max(y - 1, 0); pdy = min(y + 1, height - 1); }
  else { pdy = max(y - 1, 0); puy = min(y + 1, height - 1); }
  const f3 left = se_depth_vertex(depth, plx, y, width, K), right = se_depth_vertex(depth, prx, y, width, K);
  const f3 up = se_depth_vertex(depth, x, puy, width, K), down = se_depth_vertex(depth, x, pdy, width, K);
  if (left.z == 0 || right.z == 0 || up.z == 0 || down.z == 0) { o[0] = -2.f; return; }
  const f3 n = f3_normalized(f3_cross(f3_sub(right, left), f3_sub(up, down)));
  o[0] = n.x; o[1] = n.y; o[2] = n.z;
}

struct TrackArgs {
  float view[12];  // K * raycast_pose^-1, rows 0..2
  float dist_threshold, normal_threshold;
  int inW, inH, refW, refH;
  int level;             // pyramid level of this iteration (IcpState::stop[level])
  float icp_threshold;
};

// trackKernel (tracking.cpp:226-302) for one pixel of the pyramid level.  A rejected pixel only gets `result`, as in the reference.
__device__ __forceinline__ void se_track_pixel(TrackData& row, int px, int py, const float* __restrict__ inVertex, const float* __restrict__ inNormal,
                                               const float* __restrict__ refVertex, const float* __restrict__ refNormal, const float* T, const TrackArgs& a) {
  const f3 inN = ld3(inNormal, px + py * a.inW);
  if (inN.x == -2.f) { row.result = -1; return; }
  const f3 projectedVertex = m34_mul_h(T, ld3(inVertex, px + py * a.inW));
  const f3 projectedPos = m34_mul_h(a.view, projectedVertex);
  const float ppx = projectedPos.x / projectedPos.z + 0.5f, ppy = projectedPos.y / projectedPos.z + 0.5f;
  if (ppx < 0 || ppx > a.refW - 1 || ppy < 0 || ppy > a.refH - 1) { row.result = -2; return; }
  const int rx = cvt_i32(ppx), ry = cvt_i32(ppy);
  const f3 referenceNormal = ld3(refNormal, rx + ry * a.refW);
  if (referenceNormal.x == -2.f) { row.result = -3; return; }
  const f3 diff = f3_sub(ld3(refVertex, rx + ry * a.refW), projectedVertex);
  const float R3[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  const f3 projectedNormal = m3_mul(R3, inN);
  if (sqrtf(f3_sqnorm(diff)) > a.dist_threshold) { row.result = -4; return; }
  if (f3_dot(projectedNormal, referenceNormal) < a.normal_threshold) { row.result = -5; return; }
  row.result = 1;
  row.error = f3_dot(referenceNormal, diff);
  row.J[0] = referenceNormal.x; row.J[1] = referenceNormal.y; row.J[2] = referenceNormal.z;
  const f3 c = f3_cross(projectedVertex, referenceNormal);
  row.J[3] = c.x; row.J[4] = c.y; row.J[5] = c.z;
}

// one pixel's contribution to the 32 sums (tracking.cpp:113-170)
__device__ __forceinline__ void se_accumulate_row(float* s, const TrackData& row) {
  if (row.result < 1) {
    s[29] += row.result == -4 ? 1 : 0;
    s[30] += row.result == -5 ? 1 : 0;
    s[31] += row.result > -4 ? 1 : 0;
    return;
  }
  s[0] += row.error * row.error;
#pragma unroll
  for (int i = 0; i < 6; ++i) s[1 + i] += row.error * row.J[i];
  int k = 7;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) s[k++] += row.J[i] * row.J[j];
  s[28] += 1;
}

// ---- device-resident ICP (r03): the whole loop of DenseSLAMSystem::tracking (DenseSLAMSystem.cpp:165-186) runs without the host.
// State that one iteration hands to the next lives in device memory; the host enqueues every iteration of every level up
// front plus k_icp_finish, and reads ONE record per frame.  An iteration whose level has already met the convergence test
// returns at once (the reference's `break`, tracking.cpp:180-183).
struct IcpState {
  float pose[16];      // current estimate, row-major 4x4 (Ttrack of the next iteration)
  float old_pose[16];  // pose_ on entry (checkPoseKernel restores it)
  float last_pose[16]; // the pose the last iteration that ran tracked with (k_icp_finish_rows rebuilds tracking_result_ from it)
  float reduce0[32];   // row 0 of reduction_output_ of the last iteration that ran
  int stop[8];         // per pyramid level: the update norm fell below icp_threshold -> the level's remaining iterations are skipped
  int iterations;      // iterations that ran
  int tracked;
};
struct IcpHostRecord { float pose[16]; float reduce0[32]; int iterations; int tracked; unsigned seq;
  // (frame seq << 32) | (launches of the frame that have stored their state << 8) | stop flags of the 8 levels: written by every k_icp_iter launch as
  // one 8-byte store; the host reads it to stop enqueuing a level's remaining iterations once the level has converged (se_hip_track)
  unsigned long long progress; };

// sin / cos of a float, DEFINED (oracle and device alike, see oracle/se_oracle.cpp so_sincos) as the correctly rounded result:
// evaluated in double -- Cody-Waite reduction by pi/2 in two parts, the fdlibm kernel polynomials -- and rounded once.  Plain
// IEEE double arithmetic in source order (no FMA contraction), so host and device agree bit for bit; it equals
// (float)sin((double)x) of glibc for every one of 7.3e6 random arguments tried, and glibc's own sinf / cosf on 99.6 - 99.8 %
// (they are 1 ulp off the correctly rounded value elsewhere).  Valid for |x| < 2^20 (ICP angle updates are << pi).
__host__ __device__ inline void se_sincos_f32(float xf, float* s, float* c) {
  const double INV_PIO2 = 6.36619772367581382433e-01, PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double x = (double)xf;
  const double kd = rint(x * INV_PIO2);
  const double r = (x - kd * PIO2_1) - kd * PIO2_1T;
  const double z = r * r;
  const double ks = r + (z * r) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
  const double kc = 1.0 - (0.5 * z - z * (z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))));
  const int n = (int)((long long)kd & 3);
  const double sv = (n == 0) ? ks : (n == 1) ? kc : (n == 2) ? -ks : -kc;
  const double cv = (n == 0) ? kc : (n == 1) ? -ks : (n == 2) ? -kc : ks;
  *s = (float)sv; *c = (float)cv;
}

// updatePoseKernel's arithmetic (tracking.cpp:42-65, 304-318): makeJTJ + LLT solve.  Eigen::LLT is defined as the unblocked
// Cholesky with left-to-right inner sums (oracle: solve6).
__device__ inline bool se_solve6(const float* vals /*b[6], upper triangle[21]*/, float x[6]) {
  float Cm[6][6], L[6][6];
  int k = 6;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) { Cm[r][c] = vals[k]; Cm[c][r] = vals[k]; ++k; }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float d = Cm[j][j];
    if (j > 0) { float sn = 0; for (int q = 0; q < j; ++q) sn += L[j][q] * L[j][q]; d -= sn; }
    if (!(d > 0.f)) { for (int i = 0; i < 6; ++i) x[i] = 0.f; return false; }
    d = sqrtf(d);
    L[j][j] = d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float v = Cm[i][j];
      if (j > 0) { float sp = 0; for (int q = 0; q < j; ++q) sp += L[i][q] * L[j][q]; v -= sp; }
      L[i][j] = v / d;
    }
  }
  float yv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { float v = vals[i]; for (int q = 0; q < i; ++q) v -= L[i][q] * yv[q]; yv[i] = v / L[i][i]; }
#pragma unroll
  for (int i = 5; i >= 0; --i) { float v = yv[i]; for (int q = i + 1; q < 6; ++q) v -= L[q][i] * x[q]; x[i] = v / L[i][i]; }
  return true;
}
// Sophus::SE3f::exp (tracking.cpp:310), Sophus 1.0 closed form with epsilon 1e-5f (oracle: se3_exp); T = row-major 4x4
__device__ inline void se_se3_exp(const float a[6], float T[16]) {
  const float eps = 1e-5f;
  const float ox = a[3], oy = a[4], oz = a[5];
  const float theta_sq = (ox * ox + oy * oy) + oz * oz, theta = sqrtf(theta_sq), half_theta = 0.5f * theta;
  float imag_factor, real_factor;
  if (theta_sq < eps * eps) {
    const float theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_po4;
    real_factor = 1.f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_po4;
  } else {
    float sh, ch;
    se_sincos_f32(half_theta, &sh, &ch);
    imag_factor = sh / theta;
    real_factor = ch;
  }
  float qw = real_factor, qx = imag_factor * ox, qy = imag_factor * oy, qz = imag_factor * oz;
  const float qn = sqrtf(((qw * qw + qx * qx) + qy * qy) + qz * qz);
  qw /= qn; qx /= qn; qy /= qn; qz /= qn;
  const float tx = 2.f * qx, ty = 2.f * qy, tz = 2.f * qz;
  const float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const float R[3][3] = {{1.f - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1.f - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1.f - (txx + tyy)}};
  float V[3][3];
  if (theta < eps) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = R[i][j];
  } else {
    const float Om[3][3] = {{0, -oz, oy}, {oz, 0, -ox}, {-oy, ox, 0}};
    float Om2[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Om2[i][j] = (Om[i][0] * Om[0][j] + Om[i][1] * Om[1][j]) + Om[i][2] * Om[2][j];
    float st, ct;
    se_sincos_f32(theta, &st, &ct);
    const float ca = (1.f - ct) / theta_sq, cb = (theta - st) / (theta_sq * theta);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) V[i][j] = ((i == j ? 1.f : 0.f) + ca * Om[i][j]) + cb * Om2[i][j];
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i][j];
    T[i * 4 + 3] = (V[i][0] * a[0] + V[i][1] * a[1]) + V[i][2] * a[2];
  }
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

// ---- the same arithmetic spread over the lanes of ONE wave (r04).  The single-lane versions above cost ~1 000 dependent VALU instructions (3 us) in
// the prologue of every ICP launch.  Here all 64 lanes of wave 0 run the code; what is the same for every lane is computed redundantly (free on a SIMD),
// and the pieces that are independent of each other go to different lanes: the rows of the Cholesky factor (lane r owns row r: one column step = one
// multiply-add chain, ONE division), the two sincos calls, the four quaternion divisions, the sixteen entries of the pose product.  Every float operation
// has the operands and the order it has in se_solve6 / se_se3_exp (sums left to right, true divisions, the same sqrtf): the results are bit-identical,
// which tests/test_gpu_tracking.py checks against the oracle frame by frame.  SE_ICP_WAVE_SOLVE 0 = the single-lane code (A/B).
#ifndef SE_ICP_WAVE_SOLVE
#define SE_ICP_WAVE_SOLVE 1
#endif
__device__ __forceinline__ float se_lane(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
// vals: b[6], upper triangle[21] (LDS); x: the solution, in every lane.  (Eigen::LLT as in se_solve6.)
__device__ __forceinline__ void se_solve6_wave(const float* vals, float x[6], int lane) {
  const int r = lane < 6 ? lane : 5;     // (lanes 6..63 mirror lane 5: defined arithmetic, never read)
  float Lr[6], diag = 1.f;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    if (ok) {
      const int lo = r < j ? r : j, hi = r < j ? j : r;
      float w = vals[6 + lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];     // C[r][j]
      if (j > 0) {
        float sp = 0.f;
#pragma unroll
        for (int q = 0; q < j; ++q) sp += Lr[q] * se_lane(Lr[q], j);     // L[r][q] * L[j][q]; lane j: L[j][q]^2
        w -= sp;
      }
      const float d = se_lane(w, j);
      if (!(d > 0.f)) ok = false;      // (same in every lane)
      else {
        const float sd = sqrtf(d);
        const float off = w / sd;
        Lr[j] = (r == j) ? sd : off;     // (rows above the diagonal hold values nobody reads)
        if (r == j) diag = sd;
      }
    }
  }
  if (!ok) {
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = 0.f;
    return;
  }
  // forward substitution: lane i holds b[i] - sum_{q<i} L[i][q] y[q], subtracted in the order q = 0, 1, ...
  float v = vals[r], y[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    y[q] = se_lane(v / diag, q);
    if (q < 5) v -= Lr[q] * y[q];
  }
  // backward substitution is a chain (x[i] needs every x[q > i], subtracted in the order q = i+1, ..., 5): every lane runs it on the factor's columns
  float Lc[6][6], dg[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    dg[i] = se_lane(diag, i);
#pragma unroll
    for (int q = i + 1; q < 6; ++q) Lc[q][i] = se_lane(Lr[i], q);
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float u = y[i];
#pragma unroll
    for (int q = i + 1; q < 6; ++q) u -= Lc[q][i] * x[q];
    x[i] = u / dg[i];
  }
}
// se_se3_exp over a wave: T (rows 0..2; row 3 is 0 0 0 1) in every lane
__device__ __forceinline__ void se_se3_exp_wave(const float a[6], float T[16], int lane) {
  const float eps = 1e-5f;
  const float ox = a[3], oy = a[4], oz = a[5];
  const float theta_sq = (ox * ox + oy * oy) + oz * oz, theta = sqrtf(theta_sq), half_theta = 0.5f * theta;
  // lane 0: sincos(half_theta), lane 1: sincos(theta) -- evaluated whether or not the small-angle branches below use them
  float sv, cv;
  se_sincos_f32((lane & 1) ? theta : half_theta, &sv, &cv);
  const float sh = se_lane(sv, 0), ch = se_lane(cv, 0), st = se_lane(sv, 1), ct = se_lane(cv, 1);
  float imag_factor, real_factor;
  if (theta_sq < eps * eps) {
    const float theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_po4;
    real_factor = 1.f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_po4;
  } else {
    imag_factor = sh / theta;
    real_factor = ch;
  }
  float qw = real_factor, qx = imag_factor * ox, qy = imag_factor * oy, qz = imag_factor * oz;
  const float qn = sqrtf(((qw * qw + qx * qx) + qy * qy) + qz * qz);
  {
    const int c = lane & 3;
    const float comp = c == 0 ? qw : c == 1 ? qx : c == 2 ? qy : qz;
    const float nq = comp / qn;
    qw = se_lane(nq, 0); qx = se_lane(nq, 1); qy = se_lane(nq, 2); qz = se_lane(nq, 3);
  }
  const float tx = 2.f * qx, ty = 2.f * qy, tz = 2.f * qz;
  const float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const float R[3][3] = {{1.f - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1.f - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1.f - (txx + tyy)}};
  float V[3][3];
  if (theta < eps) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = R[i][j];
  } else {
    const float Om[3][3] = {{0, -oz, oy}, {oz, 0, -ox}, {-oy, ox, 0}};
    float Om2[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Om2[i][j] = (Om[i][0] * Om[0][j] + Om[i][1] * Om[1][j]) + Om[i][2] * Om[2][j];
    const float ca = (1.f - ct) / theta_sq, cb = (theta - st) / (theta_sq * theta);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) V[i][j] = ((i == j ? 1.f : 0.f) + ca * Om[i][j]) + cb * Om2[i][j];
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i][j];
    T[i * 4 + 3] = (V[i][0] * a[0] + V[i][1] * a[1]) + V[i][2] * a[2];
  }
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

struct Pose16 { float m[16]; };   // row-major 4x4
// One ICP iteration = ONE launch, no host in between, no atomics (r04; r03: two launches, k_icp_track + k_icp_update, ~15 us per
// iteration of which 4.7 us were the single-workgroup update kernel and 3.6 us the two kernel boundaries).
//  k_icp_iter(j) = [prologue: the rest of iteration j-1] + [trackKernel and the first two stages of reduceKernel of iteration j].
//   Prologue (tracking.cpp:205-224, 304-318), executed by EVERY workgroup redundantly -- the result is the same everywhere, so no
//   workgroup has to wait for another and only workgroup (0, 0) stores it: per strip the SE_TRACK_SEGMENTS partial rows of iteration
//   j-1 are added in order (8 x 32 lanes in parallel), then the strips in order; one lane solves the 6x6 system, applies exp(x) to
//   the pose and evaluates the convergence test, all in the oracle's order.  State is double-buffered (launch j reads state[j & 1]
//   and writes state[(j + 1) & 1]) so that workgroup (0, 0) never overwrites what a later-starting workgroup still has to read; the
//   partial rows alternate likewise.
//   Pixel phase (tracking.cpp:62-302), grid = (SE_TRACK_SEGMENTS, 8).  The reference leaves the summation order of the reduction to
//   OpenMP; here (and in the oracle) it is fixed: strip b = rows y = b (mod 8) as in the reference, split into SE_TRACK_SEGMENTS
//   contiguous segments, one workgroup each; lane t computes the TrackData of pixels t, t+256, ... of its segment and accumulates
//   them in that order; the 256 partials are combined by a binary tree.  A lane has ~5 pixels on the 640x480 level: their loads are
//   issued together (inputs, then the gathers at the projected positions), the rows are accumulated afterwards, in order.
//  k_icp_finish = the prologue once more (the last iteration's sums and update) + checkPoseKernel + the host record.
// (r03's first single-launch attempt used a last-workgroup ticket and was slower than r02: every workgroup paid an agent-scope fence
//  and an atomic on one word.)
struct IcpShared { float strip[8][32]; float pose[16]; float delta[16]; float prev[16]; int conv; };
// the rest of iteration `prev`: final sums -> sh.strip[0], pose update -> sh.pose, convergence -> sh.conv; P = the pose it tracked with
__device__ __forceinline__ void se_icp_finalize(IcpShared& sh, const float* __restrict__ partial, const float* P, float icp_threshold) {
  const int t = threadIdx.x, bb = t >> 5, i = t & 31;
  if (t < 256) {
    float total = 0.f;
    for (int gg = 0; gg < SE_TRACK_SEGMENTS; ++gg) total += partial[(bb * SE_TRACK_SEGMENTS + gg) * 32 + i];
    sh.strip[bb][i] = total;
  }
  __syncthreads();
  if (t < 32) {
    float row0 = sh.strip[0][t];
    for (int b2 = 1; b2 < 8; ++b2) row0 += sh.strip[b2][t];
    sh.strip[0][t] = row0;
  }
  __syncthreads();
#if SE_ICP_WAVE_SOLVE
  if (t < 64) {      // wave 0, every lane
    float x[6], D[16];
    se_solve6_wave(&sh.strip[0][1], x, t);
    se_se3_exp_wave(x, D, t);
    // updatePoseKernel: pose = delta * pose (4x4 product, inner sums left to right) -- lane e forms entry e; delta's row and pose's column reach it through LDS
    if (t == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) { sh.delta[q] = D[q]; sh.prev[q] = P[q]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (t < 16) {
      const int r = t >> 2, c = t & 3;
      sh.pose[t] = ((sh.delta[r * 4 + 0] * sh.prev[0 * 4 + c] + sh.delta[r * 4 + 1] * sh.prev[1 * 4 + c]) + sh.delta[r * 4 + 2] * sh.prev[2 * 4 + c]) + sh.delta[r * 4 + 3] * sh.prev[3 * 4 + c];
    }
    if (t == 0) {
      float xn = 0.f;
      for (int q = 0; q < 6; ++q) xn += x[q] * x[q];
      sh.conv = sqrtf(xn) < icp_threshold ? 1 : 0;
    }
  }
#else
  if (t == 0) {
    float x[6], D[16], N[16];
    se_solve6(&sh.strip[0][1], x);
    se_se3_exp(x, D);
    // updatePoseKernel: pose = delta * pose (4x4 product, inner sums left to right)
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c)
        N[r * 4 + c] = ((D[r * 4 + 0] * P[0 * 4 + c] + D[r * 4 + 1] * P[1 * 4 + c]) + D[r * 4 + 2] * P[2 * 4 + c]) + D[r * 4 + 3] * P[3 * 4 + c];
#pragma unroll
    for (int q = 0; q < 16; ++q) sh.pose[q] = N[q];
    float xn = 0.f;
    for (int q = 0; q < 6; ++q) xn += x[q] * x[q];
    sh.conv = sqrtf(xn) < icp_threshold ? 1 : 0;
  }
#endif
  __syncthreads();
}
// state[(j + 1) & 1] as iteration j-1's update leaves it (or a plain copy when that iteration did not run)
// (first: the frame's first launch -- there is no state yet, `p0` is pose_ on entry; this replaces r03's k_icp_begin launch)
// Returns (in the lanes of wave 0) the new stop flags as a bit mask.
__device__ __forceinline__ unsigned se_icp_store_state(IcpState* __restrict__ so, const IcpState* __restrict__ si, const IcpShared& sh, bool prev_ran, int prev_level,
                                                       bool first, const Pose16& p0) {
  const int t = threadIdx.x;
  if (first) {
    if (t < 16) { so->old_pose[t] = p0.m[t]; so->pose[t] = p0.m[t]; so->last_pose[t] = p0.m[t]; }
    if (t < 32) so->reduce0[t] = 0.f;
    if (t < 8) so->stop[t] = 0;
    if (t == 0) { so->iterations = 0; so->tracked = 0; }
    return 0u;
  }
  if (t < 16) { so->old_pose[t] = si->old_pose[t]; so->pose[t] = prev_ran ? sh.pose[t] : si->pose[t]; so->last_pose[t] = prev_ran ? si->pose[t] : si->last_pose[t]; }
  if (t < 32) so->reduce0[t] = prev_ran ? sh.strip[0][t] : si->reduce0[t];
  int stop = 0;
  if (t < 8) { stop = (prev_ran && t == prev_level && sh.conv) ? 1 : si->stop[t]; so->stop[t] = stop; }
  if (t == 0) { so->iterations = si->iterations + (prev_ran ? 1 : 0); so->tracked = si->tracked; }
  return t < 64 ? (unsigned)(__ballot(stop != 0) & 0xffull) : 0u;
}
#define SE_TRACK_BATCH 5   // pixels of one lane whose loads are in flight together (640x480: ceil(1200 / 256))
__global__ __launch_bounds__(SE_TRACK_LANES) void k_icp_iter(const IcpState* __restrict__ si, IcpState* __restrict__ so, const float* __restrict__ inVertex,
                                                              const float* __restrict__ inNormal, const float* __restrict__ refVertex,
                                                              const float* __restrict__ refNormal, const float* __restrict__ partial_prev,
                                                              float* __restrict__ partial, TrackArgs a, int prev_level, Pose16 p0,
                                                              unsigned long long* __restrict__ progress, unsigned seq, int launch) {
  __shared__ float lanes[SE_TRACK_LANES / 2][33];   // what the upper half of a tree stride hands to the lower half (+1: bank-conflict padding)
  __shared__ IcpShared sh;
  const int b = blockIdx.y, g = blockIdx.x, t = threadIdx.x;
  const bool first = prev_level < 0;                                    // the frame's first launch: the state is `p0`, nothing has converged
  const bool prev_ran = !first && si->stop[prev_level] == 0;            // (launch j-1 returned before its pixel phase otherwise: nothing to finish)
  float T[12];
  int stop_cur = first ? 0 : si->stop[a.level];
  // this lane's first pixels: their inputs do not depend on the pose, so their loads fly under the prologue
  // (r06, measured and dropped: requesting the previous iteration's partial rows and its pose here as well, before the stop flags are known, so that the
  // prologue's loads are one round trip instead of two -- tracked loop 4 115 -> 4 072 frames/s, same poses: the flags arrive with the kernel arguments'
  // first use and the 32 extra loads per lane delay the pixel loads behind them; tools/track_ab.py)
  const int W = a.inW, H = a.inH;
  const int rows = (H - b + 7) / 8;
  const long npx = (long)rows * W;
  const long seg_len = (npx + SE_TRACK_SEGMENTS - 1) / SE_TRACK_SEGMENTS;
  const long lo = g * seg_len, hi = min(npx, (g + 1) * seg_len);
  f3 inN0[SE_TRACK_BATCH], inV0[SE_TRACK_BATCH];
  if (!stop_cur) {
#pragma unroll
    for (int j = 0; j < SE_TRACK_BATCH; ++j) {
      const long i = lo + t + (long)j * SE_TRACK_LANES;
      const long ii = i < hi ? i : (lo < npx ? lo : 0);   // (a lane without a pixel reads some pixel of the strip)
      const int y = b + 8 * (int)(ii / W), x = (int)(ii % W);
      inN0[j] = ld3(inNormal, x + y * a.inW);
      inV0[j] = ld3(inVertex, x + y * a.inW);
    }
  }
  if (prev_ran) {
    float P[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) P[i] = si->pose[i];
    se_icp_finalize(sh, partial_prev, P, a.icp_threshold);
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = sh.pose[i];
    if (prev_level == a.level && sh.conv) stop_cur = 1;
  } else {
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = first ? p0.m[i] : si->pose[i];
  }
  if (g == 0 && b == 0) {
    const unsigned stops = se_icp_store_state(so, si, sh, prev_ran, prev_level, first, p0);
    // (pinned host memory; one 8-byte store, nothing else to order it with: the host only uses it to stop enqueuing launches that would return at once)
    if (t == 0 && progress) *(volatile unsigned long long*)progress = ((unsigned long long)seq << 32) | ((unsigned long long)(launch + 1) << 8) | stops;
  }
  if (stop_cur) return;                 // the level has converged: the reference's `break`
  const float R3[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.f;
  for (long base = lo + t; base < hi; base += (long)SE_TRACK_BATCH * SE_TRACK_LANES) {
    // trackKernel (tracking.cpp:226-302) for SE_TRACK_BATCH pixels, staged so that their loads overlap: same operations per pixel as se_track_pixel
    f3 inN[SE_TRACK_BATCH], inV[SE_TRACK_BATCH], rN[SE_TRACK_BATCH], rV[SE_TRACK_BATCH], pv[SE_TRACK_BATCH];
    int res[SE_TRACK_BATCH], ridx[SE_TRACK_BATCH];
#pragma unroll
    for (int j = 0; j < SE_TRACK_BATCH; ++j) {
      const long i = base + (long)j * SE_TRACK_LANES;
      const bool live = i < hi;
      const long ii = live ? i : (lo < npx ? lo : 0);
      const int y = b + 8 * (int)(ii / W), x = (int)(ii % W);
      if (base == lo + t) { inN[j] = inN0[j]; inV[j] = inV0[j]; }     // (fetched before the prologue)
      else { inN[j] = ld3(inNormal, x + y * a.inW); inV[j] = ld3(inVertex, x + y * a.inW); }
      res[j] = live ? 0 : 2;              // 2: no pixel
    }
#pragma unroll
    for (int j = 0; j < SE_TRACK_BATCH; ++j) {
      if (res[j] == 0 && inN[j].x == -2.f) res[j] = -1;
      pv[j] = m34_mul_h(T, inV[j]);
      const f3 projectedPos = m34_mul_h(a.view, pv[j]);
      const float ppx = projectedPos.x / projectedPos.z + 0.5f, ppy = projectedPos.y / projectedPos.z + 0.5f;
      if (res[j] == 0 && (ppx < 0 || ppx > a.refW - 1 || ppy < 0 || ppy > a.refH - 1)) res[j] = -2;
      ridx[j] = res[j] == 0 ? cvt_i32(ppx) + cvt_i32(ppy) * a.refW : 0;
    }
#pragma unroll
    for (int j = 0; j < SE_TRACK_BATCH; ++j) { rN[j] = ld3(refNormal, ridx[j]); rV[j] = ld3(refVertex, ridx[j]); }
#pragma unroll
    for (int j = 0; j < SE_TRACK_BATCH; ++j) {
      if (res[j] == 2) continue;
      TrackData row;
      row.error = 0.f;
#pragma unroll
      for (int q = 0; q < 6; ++q) row.J[q] = 0.f;
      row.result = res[j];
      if (res[j] == 0) {
        const f3 referenceNormal = rN[j];
        if (referenceNormal.x == -2.f) row.result = -3;
        else {
          const f3 diff = f3_sub(rV[j], pv[j]);
          const f3 projectedNormal = m3_mul(R3, inN[j]);
          if (sqrtf(f3_sqnorm(diff)) > a.dist_threshold) row.result = -4;
          else if (f3_dot(projectedNormal, referenceNormal) < a.normal_threshold) row.result = -5;
          else {
            row.result = 1;
            row.error = f3_dot(referenceNormal, diff);
            row.J[0] = referenceNormal.x; row.J[1] = referenceNormal.y; row.J[2] = referenceNormal.z;
            const f3 c = f3_cross(pv[j], referenceNormal);
            row.J[3] = c.x; row.J[4] = c.y; row.J[5] = c.z;
          }
        }
      }
      se_accumulate_row(acc, row);     // (tracking_result_ itself is written once per frame, by k_icp_finish_rows)
    }
  }
  // the binary tree over the 256 lanes, lane t += lane t + st for st = 128, 64, ..., 1 (the order the oracle fixes).  The sums stay in registers:
  // for the two strides that cross waves the upper half hands its 32 values over through LDS, the six strides inside wave 0 are lane shuffles
  // (r04; before, every stride read both operands from LDS and wrote the sum back: 768 LDS instructions on wave 0 instead of 256).
#pragma unroll
  for (int st = SE_TRACK_LANES / 2; st >= 64; st >>= 1) {
    if (t >= st && t < 2 * st)
#pragma unroll
      for (int i = 0; i < 32; ++i) lanes[t - st][i] = acc[i];
    __syncthreads();
    if (t < st)
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] += lanes[t][i];
    __syncthreads();
  }
  if (t < 64) {
#pragma unroll
    for (int st = 32; st > 0; st >>= 1)
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] += __shfl_down(acc[i], st, 64);     // (lanes >= st add something irrelevant: only lanes < st are read next)
    if (t == 0) {
      float4* dst = reinterpret_cast<float4*>(partial + (size_t)(b * SE_TRACK_SEGMENTS + g) * 32);
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
    }
  }
}

// The last iteration's prologue work (see k_icp_iter) + checkPoseKernel (tracking.cpp:320-334) + the one record the host reads per
// frame (pinned memory, sequence word last).  One workgroup of SE_TRACK_LANES threads.
__device__ __forceinline__ void se_icp_finish_wg(IcpShared& sh, const IcpState* __restrict__ si, IcpState* __restrict__ so, const float* __restrict__ partial_prev,
                                                 IcpHostRecord* __restrict__ host, int W, int H, unsigned seq, float icp_threshold, int prev_level, const Pose16& p0) {
  const int t = threadIdx.x;
  if (prev_level < 0) {   // no iteration was enqueued at all (every pyramid entry 0): the record is the entry pose, sums zero -> rejected, as in the reference
    se_icp_store_state(so, si, sh, false, prev_level, true, p0);
    if (t != 0) return;
    for (int i = 0; i < 16; ++i) host->pose[i] = p0.m[i];
    for (int i = 0; i < 32; ++i) host->reduce0[i] = 0.f;
    host->iterations = 0; host->tracked = 0;
    __threadfence_system();
    *(volatile unsigned*)&host->seq = seq;
    return;
  }
  const bool prev_ran = si->stop[prev_level] == 0;
  if (prev_ran) {
    float P[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) P[i] = si->pose[i];
    se_icp_finalize(sh, partial_prev, P, icp_threshold);
  }
  se_icp_store_state(so, si, sh, prev_ran, prev_level, false, p0);
  if (t != 0) return;
  const float* v = prev_ran ? sh.strip[0] : si->reduce0;     // reduction_output_ of the last iteration that ran
  const bool bad = ((double)sqrtf(v[0] / v[28]) > 2e-2) || (v[28] / (W * H) < 0.15f);
  for (int i = 0; i < 16; ++i) host->pose[i] = bad ? si->old_pose[i] : (prev_ran ? sh.pose[i] : si->pose[i]);
  for (int i = 0; i < 32; ++i) host->reduce0[i] = v[i];
  host->iterations = si->iterations + (prev_ran ? 1 : 0);
  host->tracked = bad ? 0 : 1;
  so->tracked = bad ? 0 : 1;     // (se_icp_store_state's own store to this word was this thread's, earlier in program order)
  __threadfence_system();
  *(volatile unsigned*)&host->seq = seq;
}
// (stand-alone: a frame whose pyramid holds no iteration at all)
__global__ __launch_bounds__(SE_TRACK_LANES) void k_icp_finish(const IcpState* __restrict__ si, IcpState* __restrict__ so, const float* __restrict__ partial_prev,
                                                                IcpHostRecord* __restrict__ host, int W, int H, unsigned seq, float icp_threshold, int prev_level, Pose16 p0) {
  __shared__ IcpShared sh;
  se_icp_finish_wg(sh, si, so, partial_prev, host, W, H, seq, icp_threshold, prev_level, p0);
}
// tracking_result_ (what renderTrackKernel shows and se_hip_download_track returns) as the reference leaves it after the frame's last
// ICP iteration: once per tracked frame (k_icp_finish_rows) over the finest level that ran, with the pose that iteration tracked with -- instead
// of 32 bytes per pixel stored by every one of the 19 iterations (10 MB each on the 640x480 level).  `result` of every pixel and
// error / J of the accepted ones are the reference's; the reference's rejected pixels keep whatever an earlier iteration or
// frame left in error / J, which nothing reads.
// The frame's last launch: workgroup (0, 0) is k_icp_finish -- it is dispatched first, so the host's record is on its way while the other
// workgroups (grid rows 1..inH) write tracking_result_ with the pose the last iteration that ran tracked with, read from the state
// as it was BEFORE the finish (si: its `pose` if the last enqueued iteration ran, else `last_pose`).  r04; before, finish and rows were two launches
// and the rows launch kept the stream busy when the caller's integration() arrived (its scan then took the side queue and an event join).
__global__ __launch_bounds__(SE_TRACK_LANES) void k_icp_finish_rows(const IcpState* __restrict__ si, IcpState* __restrict__ so, const float* __restrict__ partial_prev,
                                                                     IcpHostRecord* __restrict__ host, unsigned seq, int prev_level, Pose16 p0,
                                                                     TrackData* __restrict__ output, const float* __restrict__ inVertex, const float* __restrict__ inNormal,
                                                                     const float* __restrict__ refVertex, const float* __restrict__ refNormal, TrackArgs a) {
  __shared__ IcpShared sh;
  if (blockIdx.y == 0) {
    if (blockIdx.x == 0) se_icp_finish_wg(sh, si, so, partial_prev, host, a.refW, a.refH, seq, a.icp_threshold, prev_level, p0);
    return;
  }
  const int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y - 1;
  if (px >= a.inW || py >= a.inH) return;
  const bool prev_ran = si->stop[prev_level] == 0;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = prev_ran ? si->pose[i] : si->last_pose[i];
  TrackData row;
  se_track_pixel(row, px, py, inVertex, inNormal, refVertex, refNormal, T, a);
  TrackData& dst = output[px + py * a.refW];
  dst.result = row.result;
  if (row.result == 1) { dst.error = row.error; for (int j = 0; j < 6; ++j) dst.J[j] = row.J[j]; }
}

// renderTrackKernel (rendering.cpp:154-213)
__global__ void k_render_track(unsigned char* __restrict__ out, const TrackData* __restrict__ data, int n) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n) return;
  unsigned char r, g, b;
  switch (data[pos].result) {
    case 1: r = 128; g = 128; b = 128; break;
    case -1: r = 0; g = 0; b = 0; break;
    case -2: r = 255; g = 0; b = 0; break;
    case -3: r = 0; g = 255; b = 0; break;
    case -4: r = 0; g = 0; b = 255; break;
    case -5: r = 255; g = 255; b = 0; break;
    default: r = 255; g = 128; b = 128; break;
  }
  unsigned char* o = out + 4 * (size_t)pos;
  o[0] = r; o[1] = g; o[2] = b; o[3] = 0;
}
