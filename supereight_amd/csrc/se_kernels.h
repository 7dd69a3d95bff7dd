// gfx950 kernels of the dense-fusion hot path.  Each kernel names the reference code whose
// result it reproduces (paths relative to the reference checkout); none of it is translated:
// the pointer octree is replaced by the dense index pyramid of se_device.h, sort/unique
// allocation by atomic insertion, the per-thread active-list build by an in-kernel predicate.
#pragma once
#include "se_device.h"

#define SE_WG 256
#define SE_STACK 10  // ray stack slots = octree levels above the leaves (<= 9 for 4096^3)

// ------------------------------------------------------------------------------------------
// index insertion (replaces Octree::allocate / allocate_level, se_core/include/se/octree.hpp:792-856)
// ------------------------------------------------------------------------------------------
// Creates every missing ancestor of the octant (x,y,z)@level.  A thread that loses the CAS
// stops: the winner keeps walking up, so all ancestors exist when the kernel ends.
__device__ __forceinline__ void se_ensure_ancestors(const DevMap& m, int level, int x, int y, int z) {
  for (int l = level - 1; l >= 1; --l) {
    x >>= 1; y >>= 1; z >>= 1;
    uint32_t* e = m.tab + tab_index(m, l, x, y, z);
    const uint32_t old = atomicCAS(e, 0u, SE_PENDING);
    if (old != 0u) break;
    const uint32_t nid = atomicAdd(&m.ctr[C_NODES], 1u);
    if (nid >= m.cap_nodes) { m.ctr[C_OVERFLOW] = 1u; atomicExch(e, 0u); break; }
    m.npos[nid] = pack_pos(x, y, z);
    m.nlevel[nid] = (uint8_t)l;
    atomicExch(e, nid + 1u);
  }
}

// Inserts the octant (x,y,z)@level (block if level == leaf_level, else an internal node with no
// children yet) if absent.  Returns true if this thread created it.
__device__ __forceinline__ bool se_insert_octant(const DevMap& m, int level, int x, int y, int z) {
  uint32_t* e = m.tab + tab_index(m, level, x, y, z);
  const uint32_t old = atomicCAS(e, 0u, SE_PENDING);
  if (old != 0u) return false;
  if (level == m.leaf_level) {
    const uint32_t bid = atomicAdd(&m.ctr[C_BLOCKS], 1u);
    if (bid >= m.cap_blocks) { m.ctr[C_OVERFLOW] = 1u; atomicExch(e, 0u); return false; }
    m.bpos[bid] = pack_pos(x, y, z);
    m.bactive[bid] = 1;  // allocate_level: active(true), octree.hpp:841
    atomicExch(e, bid + 1u);
  } else {
    const uint32_t nid = atomicAdd(&m.ctr[C_NODES], 1u);
    if (nid >= m.cap_nodes) { m.ctr[C_OVERFLOW] = 1u; atomicExch(e, 0u); return false; }
    m.npos[nid] = pack_pos(x, y, z);
    m.nlevel[nid] = (uint8_t)level;
    atomicExch(e, nid + 1u);
  }
  se_ensure_ancestors(m, level, x, y, z);
  return true;
}

__device__ __forceinline__ void se_append_key(const DevMap& m, int level, int x, int y, int z) {
  const unsigned long long idx = atomicAdd(&m.newkeys[0], 1ull);
  if (idx < m.cap_keys) m.newkeys[1 + idx] = se_make_key(x, y, z, level, m.max_level);
  else m.ctr[C_OVERFLOW] = 2u;
}

template <bool STATS> __device__ __forceinline__ void se_stat_add(const DevMap& m, int which, unsigned long long v) {
  if (STATS) {
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&m.stats[which], v);
  }
}

struct AllocArgs {
  float kpose[12];  // rows 0..2 of pose * K^-1
  float cam[3];     // pose translation
  float band;
  float inv_voxel;  // 1 / voxelSize
  float voxel;
  int num_steps;    // SDF: ceil(band * inv_voxel)
  int W, H, row_begin, row_end;
  int depth_fine, depth_mid, depth_coarse;  // OFusion: step_to_depth() of the three step sizes
};

// ------------------------------------------------------------------------------------------
// SDF allocation scan: buildAllocationList (se_denseslam/src/kfusion/alloc_impl.hpp:54-118)
// fused with Octree::allocate.  One thread per pixel; every band step probes the leaf grid;
// a miss inserts the block (one winner per block), a hit sets VoxelBlock::active_.
// ------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(SE_WG) void k_alloc_scan_sdf(DevMap m, const float* __restrict__ depthmap, AllocArgs a) {
  const int npix = (a.row_end - a.row_begin) * a.W;
  const int pid = blockIdx.x * SE_WG + threadIdx.x;
  unsigned long long probes = 0, newk = 0;
  if (pid < npix) {
    const int x = pid % a.W;
    const int y = a.row_begin + pid / a.W;
    const float depth = depthmap[x + y * a.W];
    if (!(depth == 0)) {
      const f3 worldVertex = m34_mul_h(a.kpose, {(x + 0.5f) * depth, (y + 0.5f) * depth, depth});
      const f3 camera = {a.cam[0], a.cam[1], a.cam[2]};
      const f3 direction = f3_normalized(f3_sub(camera, worldVertex));
      const f3 origin = f3_sub(worldVertex, f3_scale(a.band * 0.5f, direction));
      const f3 step = f3_div(f3_scale_r(direction, a.band), (float)a.num_steps);
      f3 voxelPos = origin;
      const float fsize = (float)m.size;
      int lbx = -1, lby = -1, lbz = -1;  // last block handled (probing it again changes nothing)
      for (int i = 0; i < a.num_steps; ++i) {
        const f3 s = f3_scale_r(voxelPos, a.inv_voxel);
        const float vx = floorf(s.x), vy = floorf(s.y), vz = floorf(s.z);
        if ((vx < fsize) && (vy < fsize) && (vz < fsize) && (vx >= 0) && (vy >= 0) && (vz >= 0)) {
          const int bx = (int)vx >> 3, by = (int)vy >> 3, bz = (int)vz >> 3;
          ++probes;
          if (bx != lbx || by != lby || bz != lbz) {
            lbx = bx; lby = by; lbz = bz;
            const uint32_t e = m.tab[tab_index(m, m.leaf_level, bx, by, bz)];
            if (e == 0u) {
              if (se_insert_octant(m, m.leaf_level, bx, by, bz)) { se_append_key(m, m.leaf_level, bx, by, bz); ++newk; }
            } else if (e != SE_PENDING) {
              m.bactive[e - 1u] = 1;  // n->active(true), alloc_impl.hpp:109
            }
          }
        }
        voxelPos = f3_add(voxelPos, step);
      }
    }
  }
  se_stat_add<STATS>(m, S_PROBES, probes);
  se_stat_add<STATS>(m, S_NEWKEYS, newk);
}

// ------------------------------------------------------------------------------------------
// OFusion allocation scan: buildOctantList (se_denseslam/src/bfusion/alloc_impl.hpp:56-129)
// fused with Octree::allocate.  Marches from behind the surface to the camera with the
// three-stage step size; coarse steps insert childless octants at levels leaf-1 / leaf-2.
// ------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(SE_WG) void k_alloc_scan_ofusion(DevMap m, const float* __restrict__ depthmap, AllocArgs a) {
  const int npix = (a.row_end - a.row_begin) * a.W;
  const int pid = blockIdx.x * SE_WG + threadIdx.x;
  unsigned long long probes = 0, newk = 0;
  if (pid < npix) {
    const int x = pid % a.W;
    const int y = a.row_begin + pid / a.W;
    const float depth = depthmap[x + y * a.W];
    if (!(depth == 0)) {
      int tree_depth = m.max_level;
      float stepsize = a.voxel;
      const f3 worldVertex = m34_mul_h(a.kpose, {(x + 0.5f) * depth, (y + 0.5f) * depth, depth});
      const f3 camera = {a.cam[0], a.cam[1], a.cam[2]};
      const f3 direction = f3_normalized(f3_sub(camera, worldVertex));
      const f3 origin = f3_sub(worldVertex, f3_scale(a.band * 0.5f, direction));
      const float dist = sqrtf(f3_sqnorm(f3_sub(camera, origin)));
      f3 step = f3_scale_r(direction, stepsize);
      f3 voxelPos = origin;
      const float fsize = (float)m.size;
      const float hf_band = a.band, half = a.band * 0.5f;
      for (float travelled = 0.f; travelled < dist; travelled += stepsize) {
        const f3 s = f3_scale_r(voxelPos, a.inv_voxel);
        const float vx = floorf(s.x), vy = floorf(s.y), vz = floorf(s.z);
        if ((vx < fsize) && (vy < fsize) && (vz < fsize) && (vx >= 0) && (vy >= 0) && (vz >= 0)) {
          ++probes;
          const int lvl = tree_depth < m.leaf_level ? tree_depth : m.leaf_level;  // fetch_octant stops at the leaves
          const int sh = m.max_level - lvl;
          const int ox = (int)vx >> sh, oy = (int)vy >> sh, oz = (int)vz >> sh;
          const uint32_t e = m.tab[tab_index(m, lvl, ox, oy, oz)];
          if (e == 0u) {
            if (se_insert_octant(m, lvl, ox, oy, oz)) { se_append_key(m, lvl, ox, oy, oz); ++newk; }
          } else if (tree_depth >= m.leaf_level && e != SE_PENDING) {
            m.bactive[e - 1u] = 1;
          }
        }
        // compute_stepsize / step_to_depth (alloc_impl.hpp:37-51); the three depths are
        // evaluated on the host with the C library's log2f
        if (travelled < hf_band) { stepsize = a.voxel; tree_depth = a.depth_fine; }
        else if (travelled < hf_band + half) { stepsize = 10.f * a.voxel; tree_depth = a.depth_mid; }
        else { stepsize = 30.f * a.voxel; tree_depth = a.depth_coarse; }
        step = f3_scale_r(direction, stepsize);
        voxelPos = f3_add(voxelPos, step);
      }
    }
  }
  se_stat_add<STATS>(m, S_PROBES, probes);
  se_stat_add<STATS>(m, S_NEWKEYS, newk);
}

// Octree::allocate for key lists gathered from other ranks (multi-GPU): one thread per key.
__global__ __launch_bounds__(SE_WG) void k_alloc_commit(DevMap m, const unsigned long long* __restrict__ lists, int nlists,
                                                         long long stride_words) {
  const int li = blockIdx.y;
  if (li >= nlists) return;
  const unsigned long long* list = lists + (long long)li * stride_words;
  unsigned long long n = list[0];
  if (n > (unsigned long long)(stride_words - 1)) n = (unsigned long long)(stride_words - 1);
  for (unsigned long long i = blockIdx.x * (unsigned long long)SE_WG + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * SE_WG) {
    const unsigned long long key = list[1 + i];
    const int level = (int)(key & 0x1FFull);
    if (level < 1 || level > m.leaf_level) continue;
    const unsigned long long code = key & ~0x1FFull;
    const int sh = m.max_level - level;
    const int x = (int)(se_compact21(code) >> sh), y = (int)(se_compact21(code >> 1) >> sh), z = (int)(se_compact21(code >> 2) >> sh);
    if ((unsigned)x >= (1u << level) || (unsigned)y >= (1u << level) || (unsigned)z >= (1u << level)) continue;
    se_insert_octant(m, level, x, y, z);
  }
}

// unique_multiscale keeps keys[0] whatever its level (se_core/include/se/algorithms/unique.hpp:64-79):
// when the smallest key of a frame's list (after filter_ancestors) is a coarse octant, the
// reference walks it down to the leaves along child 0.  k_min_key finds the smallest key greater
// than `lower` (3 passes resolve the ancestor chain), k_zero_chain inserts that chain.
__global__ __launch_bounds__(SE_WG) void k_min_key(DevMap m, unsigned long long* out, const unsigned long long* lower_ptr, int has_lower) {
  unsigned long long n = m.newkeys[0];
  if (n > m.cap_keys) n = m.cap_keys;
  const unsigned long long lower = has_lower ? *lower_ptr : 0ull;
  unsigned long long best = ~0ull;
  for (unsigned long long i = blockIdx.x * (unsigned long long)SE_WG + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * SE_WG) {
    const unsigned long long k = m.newkeys[1 + i];
    if ((!has_lower || k > lower) && k < best) best = k;
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_down(best, o); if (v < best) best = v; }
  if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(out, best);
}
__global__ void k_zero_chain(DevMap m, const unsigned long long* chain /* k0,k1,k2 candidates */) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long cur = chain[0];
  if (cur == ~0ull) return;
  for (int j = 1; j < 3; ++j) {
    const unsigned long long nx = chain[j];
    if (nx == ~0ull) break;
    // descendant(nx, cur): octant_ops.hpp:81-88
    const int lvl = (int)(cur & 0x1FFull);
    const int sh = 3 * (m.max_level - lvl);
    if (((nx & ~0x1FFull) >> sh) == ((cur & ~0x1FFull) >> sh)) cur = nx; else break;
  }
  const int level = (int)(cur & 0x1FFull);
  if (level < 1 || level >= m.leaf_level) return;
  const unsigned long long code = cur & ~0x1FFull;
  for (int l = level + 1; l <= m.leaf_level; ++l) {
    const int sh = m.max_level - l;
    const int x = (int)(se_compact21(code) >> sh), y = (int)(se_compact21(code >> 1) >> sh), z = (int)(se_compact21(code >> 2) >> sh);
    se_insert_octant(m, l, x, y, z);
  }
}

// ------------------------------------------------------------------------------------------
// integration: projective_map (se_core/include/se/functors/projective_functor.hpp:45-176)
// ------------------------------------------------------------------------------------------
struct IntegArgs {
  float R[9], t[3];        // Tcw = SE3f(pose).inverse()
  float K3[9];             // K.topLeftCorner<3,3>()
  float cam[12];           // rows 0..2 of K * Tcw.matrix()  (in_frustum)
  float delta[3];          // R * (voxel, 0, 0)
  float cdelta[3];         // K3 * delta
  float voxel;
  float mu;                // SDF: mu; OFusion: noiseFactor
  float maxweight;
  float timestamp;         // OFusion
  int W, H;
  const float* bspline;    // OFusion: 1000-entry B-spline CDF table
  const float* logodds;    // OFusion: log2f(s/(1-s)) for every (Q1 index, Q2 index) pair, see se_hip_api.hip
};

#define SE_LO_DIM 1002  // 0..999 table entries, 1000 = "0" (t < -3), 1001 = "1" (t > 3)

// sdf_update::operator() (se_denseslam/src/kfusion/mapping_impl.hpp:35-65)
__device__ __forceinline__ void se_sdf_update(const IntegArgs& a, const float* __restrict__ depthmap, f3 pos, float px_, float py_,
                                              float& vx, float& vy, bool& dirty) {
  const int px = cvt_i32(px_), py = cvt_i32(py_);
  const float depthSample = depthmap[px + a.W * py];
  if (depthSample <= 0) return;
  const float diff = (depthSample - pos.z) * sqrtf(1 + sqf(pos.x / pos.z) + sqf(pos.y / pos.z));
  if (diff > -a.mu) {
    const float sdf = fminf(1.f, diff / a.mu);
    vx = clampf((vy * vx + sdf) / (vy + 1.f), -1.f, 1.f);
    vy = fminf(vy + 1, a.maxweight);
    dirty = true;
  }
}

// bspline_memoized index (se_denseslam/src/bfusion/mapping_impl.hpp:126-137)
__device__ __forceinline__ int se_bspline_index(float t) {
  const float inverseRange = 1 / 6.f;
  if (t >= -3.0f && t <= 3.0f) return (int)(unsigned)(((t + 3.f) * inverseRange) * (1000.f - 1) + 0.5f);
  if (t > 3) return 1001;
  return 1000;
}
// bfusion_update::operator() (se_denseslam/src/bfusion/mapping_impl.hpp:157-191)
__device__ __forceinline__ void se_bfusion_update(const IntegArgs& a, const float* __restrict__ depthmap, f3 pos, float px_, float py_,
                                                  float& vx, float& vy, bool& dirty) {
  const int px = cvt_i32(px_), py = cvt_i32(py_);
  const float depthSample = depthmap[px + a.W * py];
  if (depthSample <= 0) return;
  const float diff = (pos.z - depthSample) * sqrtf(1 + sqf(pos.x / pos.z) + sqf(pos.y / pos.z));
  const float sigma = clampf(a.mu * sqf(pos.z), 2 * a.voxel, 0.05f);
  const float tt = diff / sigma;
  // HNew(): sample = Q(t) - 0.5 Q(t-3); the clamp to [0.03, 0.97] and log2f(s/(1-s)) of
  // updateLogs() are folded into the (i1, i2) table built by the host with the C library
  const int i1 = se_bspline_index(tt), i2 = se_bspline_index(tt - 3);
  const float q1 = i1 < 1000 ? a.bspline[i1] : (i1 == 1001 ? 1.f : 0.f);
  const float q2 = i2 < 1000 ? a.bspline[i2] : (i2 == 1001 ? 1.f : 0.f);
  const float sample = q1 - q2 * 0.5f;
  if (sample == 0.5f) return;
  const float lo = a.logodds[i1 * SE_LO_DIM + i2];
  const double delta_t = (double)a.timestamp - (double)vy;
  const float dtf = (float)delta_t;
  float fraction = 1.f / (1.f + (dtf / 4.f));   // applyWindow, CAPITAL_T = 4
  fraction = std_max(0.5f, fraction);
  vx = vx * fraction;
  vx = clampf(vx + lo, -1000.f, 1000.f);
  vy = a.timestamp;
  dirty = true;
}

// in_frustum (se_core/include/se/algorithms/filter.hpp:38-49): min corner only, no z > 0 test
__device__ __forceinline__ bool se_in_frustum(const IntegArgs& a, int bx8, int by8, int bz8) {
  const f3 p = {(float)bx8 * a.voxel, (float)by8 * a.voxel, (float)bz8 * a.voxel};
  const f3 vc = m34_mul_h(a.cam, p);
  const int px = cvt_i32(vc.x / vc.z), py = cvt_i32(vc.y / vc.z);
  return px >= 0 && px < a.W && py >= 0 && py < a.H;
}

// update_node (projective_functor.hpp:113-137): one thread per (node, child corner).
// unpack_morton(node->code_) is applied to the full key in the reference, so the level bits
// leak into the corner position: bit0 -> x+1, bit1 -> y+1, bit2 -> z+1, bit3 -> x+2.
template <bool OFUSION>
__device__ __forceinline__ void se_update_node_corner(const DevMap& m, const float* __restrict__ depthmap, const IntegArgs& a, uint32_t tid) {
  const uint32_t n = tid >> 3;
  const int i = tid & 7;
  const int level = m.nlevel[n];
  const uint32_t np = m.npos[n];
  const int sh = m.max_level - level;
  const unsigned side = (unsigned)m.size >> level;
  const int vx0 = ((int)(np & 1023u) << sh) + (level & 1) + (((level >> 3) & 1) << 1);
  const int vy0 = ((int)((np >> 10) & 1023u) << sh) + ((level >> 1) & 1);
  const int vz0 = ((int)(np >> 20) << sh) + ((level >> 2) & 1);
  const float s = 0.5f * a.voxel * side;
  const f3 delta = m3_mul(a.R, {s, s, s});
  const f3 delta_c = m3_mul(a.K3, delta);
  const f3 base_cam = f3_add(m3_mul(a.R, f3_scale(a.voxel, {(float)vx0, (float)vy0, (float)vz0})), {a.t[0], a.t[1], a.t[2]});
  const f3 basepix_hom = m3_mul(a.K3, base_cam);
  const f3 dir = {(float)((i & 1) > 0), (float)((i & 2) > 0), (float)((i & 4) > 0)};
  const f3 vox_cam = f3_add(base_cam, f3_mul(dir, delta));
  const f3 pix_hom = f3_add(basepix_hom, f3_mul(dir, delta_c));
  if (vox_cam.z < 0.0001f) return;
  const float inverse_depth = 1.f / pix_hom.z;
  const float pixx = pix_hom.x * inverse_depth + 0.5f;
  const float pixy = pix_hom.y * inverse_depth + 0.5f;
  if (pixx < 0.5f || pixx > a.W - 1.5f || pixy < 0.5f || pixy > a.H - 1.5f) return;
  float vx = m.nx[tid], vy = m.ny[tid];
  bool dirty = false;
  if (OFUSION) se_bfusion_update(a, depthmap, vox_cam, pixx, pixy, vx, vy, dirty);
  else se_sdf_update(a, depthmap, vox_cam, pixx, pixy, vx, vy, dirty);
  if (dirty) { m.nx[tid] = vx; m.ny[tid] = vy; }
}

// projective_functor::apply (projective_functor.hpp:139-160) in one launch.
// Blocks: one wave per block, lane = x + 8*y, all 8 z-slices in flight: every slice is one
// coalesced 256-byte row of each SoA plane and the 16 loads of a lane are issued before the first
// use.  build_active_list's predicate (active || in_frustum) is evaluated per block at the top
// (wave-uniform), update_block's visibility flag is a wave ballot.  Internal nodes (8 corner
// values each) are swept by the same grid afterwards, one thread per corner.
template <bool OFUSION, bool STATS>
__global__ __launch_bounds__(SE_WG) void k_integrate(DevMap m, const float* __restrict__ depthmap, IntegArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * SE_WG + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * SE_WG) >> 6;
  const uint32_t nblocks = min(m.ctr[C_BLOCKS], m.cap_blocks);
  const uint32_t nnodes = min(m.ctr[C_NODES], m.cap_nodes);
  const int lx = lane & 7, ly = lane >> 3;
  unsigned long long swept = 0;
  for (uint32_t b = wave; b < nblocks; b += nwaves) {
    const uint32_t bp = m.bpos[b];
    const int bx = (int)(bp & 1023u) << 3, by = (int)((bp >> 10) & 1023u) << 3, bz = (int)(bp >> 20) << 3;
    if (!m.bactive[b] && !se_in_frustum(a, bx, by, bz)) continue;
    if (STATS && lane == 0) ++swept;
    float* px = m.vx + (size_t)b * 512 + lane;
    float* py = m.vy + (size_t)b * 512 + lane;
    float vx[8], vy[8];
#pragma unroll
    for (int zi = 0; zi < 8; ++zi) { vx[zi] = px[zi * 64]; vy[zi] = py[zi * 64]; }
    bool visible = false;
    const int y = by + ly;
    const float fx = (float)lx;
#pragma unroll
    for (int zi = 0; zi < 8; ++zi) {
      const int z = bz + zi;
      // update_block: projective_functor.hpp:73-111
      const f3 start = f3_add(m3_mul(a.R, {bx * a.voxel, y * a.voxel, z * a.voxel}), {a.t[0], a.t[1], a.t[2]});
      const f3 camerastart = m3_mul(a.K3, start);
      const f3 camera_voxel = f3_add(camerastart, f3_scale(fx, {a.cdelta[0], a.cdelta[1], a.cdelta[2]}));
      const f3 pos = f3_add(start, f3_scale(fx, {a.delta[0], a.delta[1], a.delta[2]}));
      if (pos.z < 0.0001f) continue;
      const float inverse_depth = 1.f / camera_voxel.z;
      const float pixx = camera_voxel.x * inverse_depth + 0.5f;
      const float pixy = camera_voxel.y * inverse_depth + 0.5f;
      if (pixx < 0.5f || pixx > a.W - 1.5f || pixy < 0.5f || pixy > a.H - 1.5f) continue;
      visible = true;
      bool dirty = false;
      if (OFUSION) se_bfusion_update(a, depthmap, pos, pixx, pixy, vx[zi], vy[zi], dirty);
      else se_sdf_update(a, depthmap, pos, pixx, pixy, vx[zi], vy[zi], dirty);
      if (dirty) { px[zi * 64] = vx[zi]; py[zi * 64] = vy[zi]; }
    }
    const bool any = __ballot(visible) != 0ull;
    if (lane == 0) m.bactive[b] = any ? 1 : 0;  // block->active(is_visible)
  }
  if (STATS && lane == 0 && swept) atomicAdd(&m.stats[S_SWEPT], swept);
  for (uint32_t tid = blockIdx.x * SE_WG + threadIdx.x; tid < nnodes * 8u; tid += gridDim.x * SE_WG)
    se_update_node_corner<OFUSION>(m, depthmap, a, tid);
}

// ------------------------------------------------------------------------------------------
// raycast: raycastKernel (se_denseslam/src/rendering.cpp:51-90) = ray_iterator first leaf
// (se_core/include/se/ray_iterator.hpp) + field-specific march (kfusion|bfusion/rendering_impl.hpp)
// + Octree::grad (se_core/include/se/octree.hpp:652-737)
// ------------------------------------------------------------------------------------------
struct RayArgs {
  float view3[9];  // (pose * K^-1).topLeftCorner<3,3>()
  float org[3];    // its translation column
  float nearp, farp, mu, step, largestep;
  float inv_voxel;   // size / dim   (VolumeTemplate, volume_template.hpp:77-102)
  float grad_scale;  // 0.5f * dim / size
  float epsilon;     // exp2f(-log2(size))
  int min_scale;     // CAST_STACK_DEPTH - log2(size / 8)
  int W, H, row_begin, row_end;
};

struct BlkCache { int bx, by, bz; uint32_t e; };

// leaf-grid entry of the block holding voxel (x,y,z); 0 if outside the volume or not allocated
__device__ __forceinline__ uint32_t se_block_of(const DevMap& m, int x, int y, int z, BlkCache& c) {
  if (!in_volume(m, x, y, z)) return 0u;
  const int bx = x >> 3, by = y >> 3, bz = z >> 3;
  if (bx == c.bx && by == c.by && bz == c.bz) return c.e;
  const uint32_t e = m.tab[tab_index(m, m.leaf_level, bx, by, bz)];
  c.bx = bx; c.by = by; c.bz = bz; c.e = e;
  return e;
}
__device__ __forceinline__ size_t se_voxel_index(uint32_t e, int x, int y, int z) {
  return (size_t)(e - 1u) * 512 + (size_t)((x & 7) + ((y & 7) << 3) + ((z & 7) << 6));
}

// Octree::interp (octree.hpp:541-563) with gather_points (interp_gather.hpp:107-237): every corner
// is read from the block that contains it; a missing block yields empty().x, the all-cross case
// goes through get_fine -> initValue().x (identical values for both field types).
__device__ __forceinline__ float se_interp(const DevMap& m, f3 pos, BlkCache& c) {
  const float flx = floorf(pos.x), fly = floorf(pos.y), flz = floorf(pos.z);
  const int bx = cvt_i32(flx), by = cvt_i32(fly), bz = cvt_i32(flz);
  const float fx = pos.x - flx, fy = pos.y - fly, fz = pos.z - flz;
  const int lx = max(bx, 0), ly = max(by, 0), lz = max(bz, 0);
  const int cm = (((lx & 7) == 7) << 2) | (((ly & 7) == 7) << 1) | ((lz & 7) == 7);
  float p[8];
  if (cm == 0) {
    const uint32_t e = se_block_of(m, lx, ly, lz, c);
    if (e == 0u) {
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = m.empty_x;
    } else {
      const float* b = m.vx + se_voxel_index(e, lx, ly, lz);
      p[0] = b[0]; p[1] = b[1]; p[2] = b[8]; p[3] = b[9]; p[4] = b[64]; p[5] = b[65]; p[6] = b[72]; p[7] = b[73];
    }
  } else {
    const float missing = (cm == 7) ? m.init_x : m.empty_x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int x = lx + (k & 1), y = ly + ((k >> 1) & 1), z = lz + (k >> 2);
      const uint32_t e = se_block_of(m, x, y, z, c);
      p[k] = e ? m.vx[se_voxel_index(e, x, y, z)] : missing;
    }
  }
  return (((p[0] * (1 - fx) + p[1] * fx) * (1 - fy) + (p[2] * (1 - fx) + p[3] * fx) * fy) * (1 - fz) +
          ((p[4] * (1 - fx) + p[5] * fx) * (1 - fy) + (p[6] * (1 - fx) + p[7] * fx) * fy) * fz);
}

// value of voxel (x,y,z) as the cached Octree::get(x,y,z,block) sees it: stored value or initValue().x
__device__ __forceinline__ float se_sample_x(const DevMap& m, int x, int y, int z, BlkCache& c) {
  const uint32_t e = se_block_of(m, x, y, z, c);
  return e ? m.vx[se_voxel_index(e, x, y, z)] : m.init_x;
}

// Octree::grad (octree.hpp:652-737), same term order
__device__ __forceinline__ f3 se_grad(const DevMap& m, f3 pos, BlkCache& c) {
  const float flx = floorf(pos.x), fly = floorf(pos.y), flz = floorf(pos.z);
  const int bx = cvt_i32(flx), by = cvt_i32(fly), bz = cvt_i32(flz);
  const float fx = pos.x - flx, fy = pos.y - fly, fz = pos.z - flz;
  const int hi = m.size - 1;
  const int llx = max(bx - 1, 0), lly = max(by - 1, 0), llz = max(bz - 1, 0);
  const int lux = max(bx, 0), luy = max(by, 0), luz = max(bz, 0);
  const int ulx = min(bx + 1, hi), uly = min(by + 1, hi), ulz = min(bz + 1, hi);
  const int uux = min(bx + 2, hi), uuy = min(by + 2, hi), uuz = min(bz + 2, hi);
  const int lox = lux, loy = luy, loz = luz, upx = ulx, upy = uly, upz = ulz;
#define G(X, Y, Z) se_sample_x(m, X, Y, Z, c)
  f3 g;
  g.x = (((G(ulx, loy, loz) - G(llx, loy, loz)) * (1 - fx) + (G(uux, loy, loz) - G(lux, loy, loz)) * fx) * (1 - fy) +
         ((G(ulx, upy, loz) - G(llx, upy, loz)) * (1 - fx) + (G(uux, upy, loz) - G(lux, upy, loz)) * fx) * fy) * (1 - fz) +
        (((G(ulx, loy, upz) - G(llx, loy, upz)) * (1 - fx) + (G(uux, loy, upz) - G(lux, loy, upz)) * fx) * (1 - fy) +
         ((G(ulx, upy, upz) - G(llx, upy, upz)) * (1 - fx) + (G(uux, upy, upz) - G(lux, upy, upz)) * fx) * fy) * fz;
  g.y = (((G(lox, uly, loz) - G(lox, lly, loz)) * (1 - fx) + (G(upx, uly, loz) - G(upx, lly, loz)) * fx) * (1 - fy) +
         ((G(lox, uuy, loz) - G(lox, luy, loz)) * (1 - fx) + (G(upx, uuy, loz) - G(upx, luy, loz)) * fx) * fy) * (1 - fz) +
        (((G(lox, uly, upz) - G(lox, lly, upz)) * (1 - fx) + (G(upx, uly, upz) - G(upx, lly, upz)) * fx) * (1 - fy) +
         ((G(lox, uuy, upz) - G(lox, luy, upz)) * (1 - fx) + (G(upx, uuy, upz) - G(upx, luy, upz)) * fx) * fy) * fz;
  g.z = (((G(lox, loy, ulz) - G(lox, loy, llz)) * (1 - fx) + (G(upx, loy, ulz) - G(upx, loy, llz)) * fx) * (1 - fy) +
         ((G(lox, upy, ulz) - G(lox, upy, llz)) * (1 - fx) + (G(upx, upy, ulz) - G(upx, upy, llz)) * fx) * fy) * (1 - fz) +
        (((G(lox, loy, uuz) - G(lox, loy, luz)) * (1 - fx) + (G(upx, loy, uuz) - G(upx, loy, luz)) * fx) * (1 - fy) +
         ((G(lox, upy, uuz) - G(lox, upy, luz)) * (1 - fx) + (G(upx, upy, uuz) - G(upx, upy, luz)) * fx) * fy) * fz;
#undef G
  return g;  // the caller applies (0.5f * dim / size)
}

// ray_iterator constructor + first next() (se_core/include/se/ray_iterator.hpp:53-226) on the
// index pyramid.  The node pointer of the reference becomes the packed position of the parent at
// its level; its stack lives in LDS, one column per lane.  Returns tcmin(); *tmax_out = tmax().
__device__ __forceinline__ float se_first_leaf(const DevMap& m, const RayArgs& a, f3 origin, f3 direction, float* tmax_out,
                                               uint32_t (*s_par)[SE_WG], float (*s_tmax)[SE_WG]) {
  const int tid = threadIdx.x;
  f3 pos = {1.0f, 1.0f, 1.0f};
  int idx = 0;
  uint32_t parent = 0u;  // root
  float scale_exp2 = 0.5f;
  int scale = 22;
  const float eps = a.epsilon;
  f3 d;
  d.x = fabsf(direction.x) < eps ? copysignf(eps, direction.x) : direction.x;
  d.y = fabsf(direction.y) < eps ? copysignf(eps, direction.y) : direction.y;
  d.z = fabsf(direction.z) < eps ? copysignf(eps, direction.z) : direction.z;
  const f3 scaled_origin = f3_add(f3_div(origin, m.dim), {1.f, 1.f, 1.f});
  const f3 t_coef = f3_scale(-1.f, {1.f / fabsf(d.x), 1.f / fabsf(d.y), 1.f / fabsf(d.z)});
  f3 t_bias = f3_mul(t_coef, scaled_origin);
  int octant_mask = 7;
  if (d.x > 0.0f) { octant_mask ^= 1; t_bias.x = 3.0f * t_coef.x - t_bias.x; }
  if (d.y > 0.0f) { octant_mask ^= 2; t_bias.y = 3.0f * t_coef.y - t_bias.y; }
  if (d.z > 0.0f) { octant_mask ^= 4; t_bias.z = 3.0f * t_coef.z - t_bias.z; }
  float t_min = fmaxf(fmaxf(2.0f * t_coef.x - t_bias.x, 2.0f * t_coef.y - t_bias.y), 2.0f * t_coef.z - t_bias.z);
  float t_max = fminf(fminf(t_coef.x - t_bias.x, t_coef.y - t_bias.y), t_coef.z - t_bias.z);
  float h = t_max;
  t_min = fmaxf(t_min, a.nearp / m.dim);
  t_max = fminf(t_max, a.farp / m.dim);
  *tmax_out = t_max * m.dim;
  if (1.5f * t_coef.x - t_bias.x > t_min) { idx ^= 1; pos.x = 1.5f; }
  if (1.5f * t_coef.y - t_bias.y > t_min) { idx ^= 2; pos.y = 1.5f; }
  if (1.5f * t_coef.z - t_bias.z > t_min) { idx ^= 4; pos.z = 1.5f; }
#pragma unroll
  for (int i = 0; i < SE_STACK; ++i) { s_par[i][tid] = 0u; s_tmax[i][tid] = 0.f; }

  f3 t_corner = {0.f, 0.f, 0.f};
  float tc_max = 0.f;
  for (int guard = 0; guard < 4096 && scale < 23; ++guard) {
    t_corner = f3_sub(f3_mul(pos, t_coef), t_bias);
    tc_max = fminf(fminf(t_corner.x, t_corner.y), t_corner.z);
    const int cidx = idx ^ octant_mask ^ 7;
    const int clevel = 23 - scale;  // level of the child
    const uint32_t child = (parent << 1) | (uint32_t)(cidx & 1) | ((uint32_t)((cidx >> 1) & 1) << 10) | ((uint32_t)(cidx >> 2) << 20);
    const bool exists = m.tab[tab_index_packed(m, clevel, child)] != 0u;
    if (scale == a.min_scale && exists) break;  // leaf found: t_min is its entry distance
    if (exists && t_min <= t_max) {
      // descend (ray_iterator.hpp:172-199)
      const float tv_max = fminf(t_max, tc_max);
      const float half = scale_exp2 * 0.5f;
      const f3 t_center = f3_add(f3_scale(half, t_coef), t_corner);
      if (tc_max < h) { s_par[22 - scale][tid] = parent; s_tmax[22 - scale][tid] = t_max; }
      h = tc_max;
      parent = child;
      idx = 0;
      scale--;
      scale_exp2 = half;
      idx ^= (t_center.x > t_min) ? 1 : 0;
      idx ^= (t_center.y > t_min) ? 2 : 0;
      idx ^= (t_center.z > t_min) ? 4 : 0;
      pos.x += scale_exp2 * (float)((idx & 1) != 0);
      pos.y += scale_exp2 * (float)((idx & 2) != 0);
      pos.z += scale_exp2 * (float)((idx & 4) != 0);
      t_max = tv_max;
      continue;
    }
    // advance_ray (ray_iterator.hpp:116-167)
    const int step_mask = (t_corner.x <= tc_max) | ((t_corner.y <= tc_max) << 1) | ((t_corner.z <= tc_max) << 2);
    pos.x -= scale_exp2 * (float)((step_mask & 1) != 0);
    pos.y -= scale_exp2 * (float)((step_mask & 2) != 0);
    pos.z -= scale_exp2 * (float)((step_mask & 4) != 0);
    t_min = tc_max;
    idx ^= step_mask;
    if ((idx & step_mask) != 0) {
      unsigned differing_bits = 0;
      if ((step_mask & 1) != 0) differing_bits |= __float_as_int(pos.x) ^ __float_as_int(pos.x + scale_exp2);
      if ((step_mask & 2) != 0) differing_bits |= __float_as_int(pos.y) ^ __float_as_int(pos.y + scale_exp2);
      if ((step_mask & 4) != 0) differing_bits |= __float_as_int(pos.z) ^ __float_as_int(pos.z + scale_exp2);
      scale = (__float_as_int((float)differing_bits) >> 23) - 127;
      scale_exp2 = __int_as_float((scale - 23 + 127) << 23);
      const int slot = 22 - scale;
      if (slot >= 0 && slot < SE_STACK) { parent = s_par[slot][tid]; t_max = s_tmax[slot][tid]; }
      if (scale >= 0 && scale < 31) {
        const int shx = __float_as_int(pos.x) >> scale;
        const int shy = __float_as_int(pos.y) >> scale;
        const int shz = __float_as_int(pos.z) >> scale;
        pos.x = __int_as_float(shx << scale);
        pos.y = __int_as_float(shy << scale);
        pos.z = __int_as_float(shz << scale);
        idx = (shx & 1) | ((shy & 1) << 1) | ((shz & 1) << 2);
      }
      h = 0.0f;
    }
  }
  return t_min * m.dim;
}

// One thread per pixel; a wave covers an 8x8 pixel tile so that its rays stay in neighbouring
// blocks.  Output: packed float3 vertex / normal images (se::Image<Eigen::Vector3f>).
template <bool OFUSION, bool STATS>
__global__ __launch_bounds__(SE_WG) void k_raycast(DevMap m, RayArgs a, float* __restrict__ vertex, float* __restrict__ normal) {
  __shared__ uint32_t s_par[SE_STACK][SE_WG];
  __shared__ float s_tmax[SE_STACK][SE_WG];
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * (SE_WG / 64) + (threadIdx.x >> 6);
  const int tiles_x = (a.W + 7) >> 3;
  const int px = ((tile % tiles_x) << 3) + (lane & 7);
  const int py = a.row_begin + ((tile / tiles_x) << 3) + (lane >> 3);
  unsigned long long n_get = 0, n_interp = 0, n_grad = 0, n_hit = 0;
  if (px < a.W && py < a.row_end) {
    const f3 dir = f3_normalized(m3_mul(a.view3, {(float)px, (float)py, 1.f}));
    const f3 org = {a.org[0], a.org[1], a.org[2]};
    float tfar;
    const float t_min = se_first_leaf(m, a, org, dir, &tfar, s_par, s_tmax);
    float hx = 0.f, hy = 0.f, hz = 0.f, hw = 0.f;
    BlkCache c = {-1, -1, -1, 0u};
    if (t_min > 0.f) {
      const float tnear = t_min;
      if (!OFUSION) {
        // raycast(const Volume<SDF>&...) (se_denseslam/src/kfusion/rendering_impl.hpp:34-74)
        if (tnear < tfar) {
          float t = tnear;
          float stepsize = a.largestep;
          f3 position = f3_add(org, f3_scale_r(dir, t));
          float f_t = se_interp(m, f3_scale(a.inv_voxel, position), c);
          if (STATS) ++n_interp;
          float f_tt = 0;
          if (f_t > 0) {
            for (int guard = 0; t < tfar && guard < 65536; t += stepsize, ++guard) {
              if (STATS) ++n_get;
              // VolumeTemplate::get -> get_fine (volume_template.hpp:77-83)
              const int ix = cvt_i32(a.inv_voxel * position.x), iy = cvt_i32(a.inv_voxel * position.y), iz = cvt_i32(a.inv_voxel * position.z);
              const uint32_t e = se_block_of(m, ix, iy, iz, c);
              float dx = m.init_x, dy = m.init_y;
              if (e) { const size_t vi = se_voxel_index(e, ix, iy, iz); dx = m.vx[vi]; dy = m.vy[vi]; }
              if (dy == 0) {
                stepsize = a.largestep;
                position = f3_add(position, f3_scale(stepsize, dir));
                continue;
              }
              f_tt = dx;
              if ((double)f_tt <= 0.1 && f_tt >= -0.5f) {
                f_tt = se_interp(m, f3_scale(a.inv_voxel, position), c);
                if (STATS) ++n_interp;
              }
              if (f_tt < 0) break;
              stepsize = fmaxf(f_tt * a.mu, a.step);
              position = f3_add(position, f3_scale(stepsize, dir));
              f_t = f_tt;
            }
            if (f_tt < 0) {
              t = t + stepsize * f_tt / (f_t - f_tt);
              const f3 r = f3_add(org, f3_scale_r(dir, t));
              hx = r.x; hy = r.y; hz = r.z; hw = t;
            }
          }
        }
      } else {
        // raycast(const Volume<OFusion>&...) (se_denseslam/src/bfusion/rendering_impl.hpp:35-68)
        if (tnear < tfar) {
          float t = tnear;
          const float stepsize = a.step;
          float f_t = se_interp(m, f3_scale(a.inv_voxel, f3_add(org, f3_scale_r(dir, t))), c);
          if (STATS) ++n_interp;
          float f_tt = 0;
          if (f_t <= 0.f) {
            for (int guard = 0; t < tfar && guard < 65536; t += stepsize, ++guard) {
              const f3 pos = f3_add(org, f3_scale_r(dir, t));
              if (STATS) ++n_get;
              const int ix = cvt_i32(a.inv_voxel * pos.x), iy = cvt_i32(a.inv_voxel * pos.y), iz = cvt_i32(a.inv_voxel * pos.z);
              const uint32_t e = se_block_of(m, ix, iy, iz, c);
              float dx = m.init_x, dy = m.init_y;
              if (e) { const size_t vi = se_voxel_index(e, ix, iy, iz); dx = m.vx[vi]; dy = m.vy[vi]; }
              if (dx > -100.f && dy > 0.f) {
                f_tt = se_interp(m, f3_scale(a.inv_voxel, pos), c);
                if (STATS) ++n_interp;
              }
              if (f_tt > 0.f) break;
              f_t = f_tt;
            }
            if (f_tt > 0.f) {
              t = t - stepsize * (f_tt - 0.f) / (f_tt - f_t);
              const f3 r = f3_add(org, f3_scale_r(dir, t));
              hx = r.x; hy = r.y; hz = r.z; hw = t;
            }
          }
        }
      }
    }
    float* v = vertex + 3 * (size_t)(px + py * a.W);
    float* n = normal + 3 * (size_t)(px + py * a.W);
    if ((double)hw > 0.0) {
      if (STATS) { ++n_hit; ++n_grad; }
      v[0] = hx; v[1] = hy; v[2] = hz;
      const f3 g = se_grad(m, f3_scale(a.inv_voxel, {hx, hy, hz}), c);
      const f3 surfNorm = f3_scale(a.grad_scale, g);
      if (sqrtf(f3_sqnorm(surfNorm)) == 0) {
        n[0] = -2.f; n[1] = 0.f; n[2] = 0.f;  // INVALID (commons.h:71)
      } else {
        const f3 nn = OFUSION ? f3_normalized(surfNorm) : f3_normalized(f3_scale(-1.f, surfNorm));
        n[0] = nn.x; n[1] = nn.y; n[2] = nn.z;
      }
    } else {
      v[0] = 0.f; v[1] = 0.f; v[2] = 0.f;
      n[0] = -2.f; n[1] = 0.f; n[2] = 0.f;
    }
  }
  if (STATS) {
    se_stat_add<true>(m, S_GETS, n_get);
    se_stat_add<true>(m, S_INTERPS, n_interp);
    se_stat_add<true>(m, S_GRADS, n_grad);
    se_stat_add<true>(m, S_HITS, n_hit);
  }
}

// pool initialisation: every voxel / node value starts at voxel_traits<T>::initValue()
__global__ void k_fill(float* __restrict__ p, float v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// mm2metersKernel (se_denseslam/src/preprocessing.cpp:161-188) on the device
__global__ void k_mm2meters(float* __restrict__ out, int ow, int oh, const unsigned short* __restrict__ in, int iw, int ratio) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x < ow && y < oh) out[x + ow * y] = in[x * ratio + iw * y * ratio] / 1000.0f;
}
