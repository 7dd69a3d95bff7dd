// the consumer side of the raycast -- image pyramid + ICP tracking
// (se_denseslam/src/preprocessing.cpp, tracking.cpp) on the device, so that vertex_ / normal_ never
// leave HBM.  Same arithmetic contract as the hot path (se_device.h).  The per-iteration 6x6 solve and
// SE3 exponential stay on the host, as updatePoseKernel does in the reference (tracking.cpp:304-318).
#pragma once
#include "se_device.h"

#define SE_TRACK_SEGMENTS 16   // summation order of the reduction, see k_track_reduce
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

// depth2vertexKernel (preprocessing.cpp:91-111): (depth * invK * Vector4f(x, y, 1, 0)).head<3>()
struct InvK { float m[12]; };
__global__ void k_depth2vertex(float* __restrict__ vertex, const float* __restrict__ depth, int W, int H, InvK K) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W || y >= H) return;
  float* v = vertex + 3 * (size_t)(x + y * W);
  const float d = depth[x + y * W];
  if (d > 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      v[i] = (((d * K.m[i * 4 + 0]) * (float)x + (d * K.m[i * 4 + 1]) * (float)y) + (d * K.m[i * 4 + 2]) * 1.f) + (d * K.m[i * 4 + 3]) * 0.f;
  } else { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; }
}

__device__ __forceinline__ f3 ld3(const float* p, int i) { return {p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]}; }
__device__ __forceinline__ f3 f3_cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float f3_dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// vertex2normalKernel<NegY> (preprocessing.cpp:113-159); an invalid pixel only gets .x = INVALID, as in the reference
__global__ void k_vertex2normal(float* __restrict__ out, const float* __restrict__ in, int width, int height, int negy) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= width || y >= height) return;
  float* o = out + 3 * (size_t)(x + y * width);
  const f3 center = ld3(in, x + width * y);
  if (center.z == 0.f) { o[0] = -2.f; return; }
  const int plx = max(x - 1, 0), prx = min(x + 1, width - 1);
  int puy, pdy;
  if (negy) { puy = max(y - 1, 0); pdy = min(y + 1, height - 1); }
  else { pdy = max(y - 1, 0); puy = min(y + 1, height - 1); }
  const f3 left = ld3(in, plx + width * y), right = ld3(in, prx + width * y), up = ld3(in, x + width * puy), down = ld3(in, x + width * pdy);
  if (left.z == 0 || right.z == 0 || up.z == 0 || down.z == 0) { o[0] = -2.f; return; }
  const f3 n = f3_normalized(f3_cross(f3_sub(right, left), f3_sub(up, down)));
  o[0] = n.x; o[1] = n.y; o[2] = n.z;
}

struct TrackArgs {
  float T[12];     // pose (Ttrack), rows 0..2
  float view[12];  // K * raycast_pose^-1, rows 0..2
  float dist_threshold, normal_threshold;
  int inW, inH, refW, refH;
};

// trackKernel (tracking.cpp:226-302): one thread per pixel of the pyramid level
__global__ __launch_bounds__(256) void k_track(TrackData* __restrict__ output, const float* __restrict__ inVertex, const float* __restrict__ inNormal,
                                               const float* __restrict__ refVertex, const float* __restrict__ refNormal, TrackArgs a) {
  const int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
  if (px >= a.inW || py >= a.inH) return;
  TrackData& row = output[px + py * a.refW];
  const f3 inN = ld3(inNormal, px + py * a.inW);
  if (inN.x == -2.f) { row.result = -1; return; }
  const f3 projectedVertex = m34_mul_h(a.T, ld3(inVertex, px + py * a.inW));
  const f3 projectedPos = m34_mul_h(a.view, projectedVertex);
  const float ppx = projectedPos.x / projectedPos.z + 0.5f, ppy = projectedPos.y / projectedPos.z + 0.5f;
  if (ppx < 0 || ppx > a.refW - 1 || ppy < 0 || ppy > a.refH - 1) { row.result = -2; return; }
  const int rx = cvt_i32(ppx), ry = cvt_i32(ppy);
  const f3 referenceNormal = ld3(refNormal, rx + ry * a.refW);
  if (referenceNormal.x == -2.f) { row.result = -3; return; }
  const f3 diff = f3_sub(ld3(refVertex, rx + ry * a.refW), projectedVertex);
  const float R3[9] = {a.T[0], a.T[1], a.T[2], a.T[4], a.T[5], a.T[6], a.T[8], a.T[9], a.T[10]};
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

// reduceKernel (tracking.cpp:62-224).  The reference leaves the summation order to an OpenMP
// reduction; here it is fixed: strip b = rows y = b (mod 8) as in the reference, split into
// SE_TRACK_SEGMENTS contiguous segments (one workgroup each); lane t accumulates pixels t, t+256, ...
// of its segment in order; the 256 partials are combined by a binary tree; k_track_reduce_final adds
// the segments and then the strips in order.  grid = (SE_TRACK_SEGMENTS, 8).
__global__ __launch_bounds__(SE_TRACK_LANES) void k_track_reduce(float* __restrict__ partial, const TrackData* __restrict__ J, int JW, int W, int H) {
  __shared__ float lanes[SE_TRACK_LANES][33];   // +1: bank-conflict padding
  const int b = blockIdx.y, g = blockIdx.x, t = threadIdx.x;
  const int rows = (H - b + 7) / 8;
  const long npx = (long)rows * W;
  const long seg_len = (npx + SE_TRACK_SEGMENTS - 1) / SE_TRACK_SEGMENTS;
  const long lo = g * seg_len, hi = min(npx, (g + 1) * seg_len);
  float s[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) s[i] = 0.f;
  for (long i = lo + t; i < hi; i += SE_TRACK_LANES) {
    const int y = b + 8 * (int)(i / W), x = (int)(i % W);
    const TrackData row = J[x + y * JW];
    se_accumulate_row(s, row);
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) lanes[t][i] = s[i];
  __syncthreads();
  for (int st = SE_TRACK_LANES / 2; st > 0; st >>= 1) {
    if (t < st)
#pragma unroll
      for (int i = 0; i < 32; ++i) lanes[t][i] += lanes[t + st][i];
    __syncthreads();
  }
  if (t < 32) partial[(b * SE_TRACK_SEGMENTS + g) * 32 + t] = lanes[0][t];
}
// The sums also go straight to pinned host memory, followed by a sequence word: updatePoseKernel's 6x6 solve runs on the
// host once per ICP iteration, and polling that word costs a few microseconds where a device-to-host copy plus a stream
// synchronisation cost ~10 (19 iterations per frame).
__global__ void k_track_reduce_final(float* __restrict__ out /*8*32*/, const float* __restrict__ partial, float* host_out, unsigned* host_seq, unsigned seq) {
  const int i = threadIdx.x;
  if (i < 32) {
    float row0 = 0.f;
    float rows[8];
    for (int b = 0; b < 8; ++b) {
      float total = 0.f;
      for (int g = 0; g < SE_TRACK_SEGMENTS; ++g) total += partial[(b * SE_TRACK_SEGMENTS + g) * 32 + i];
      rows[b] = total;
      if (b == 0) row0 = total; else row0 += total;
    }
    rows[0] = row0;
    for (int b = 0; b < 8; ++b) { out[b * 32 + i] = rows[b]; if (host_out) host_out[b * 32 + i] = rows[b]; }
  }
  if (host_seq) {
    __threadfence_system();
    __syncthreads();
    if (i == 0) { *(volatile unsigned*)host_seq = seq; }
  }
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
