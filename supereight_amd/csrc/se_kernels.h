// gfx950 kernels of the dense-fusion hot path.  Each kernel names the reference code whose
// result it reproduces (paths relative to the reference checkout); none of it is translated:
// the pointer octree is replaced by the dense index pyramid of se_device.h, sort/unique
// allocation by atomic insertion, the per-thread active-list build by an in-kernel predicate.
#pragma once
#include "se_device.h"

#ifndef SE_WG
#define SE_WG 256
#endif
// Workgroup sizes of the two image-space kernels (the LDS staging of the occupancy bits is per
// workgroup).  Measured on MI355X, 640x480 -> 512^3, same box: raycast 65 / 55.5 / 56.0 us with 64 / 128 / 256
// threads; the allocation scan is insensitive (37-40 us overlapped).
#ifndef SE_WG_RAY
#define SE_WG_RAY 128
#endif
#ifndef SE_WG_SCAN
#define SE_WG_SCAN 128
#endif
#ifndef SE_TILE_W
#define SE_TILE_W 8     // raycast: a wave covers a SE_TILE_W x SE_TILE_H pixel tile (product 64)
#define SE_TILE_H 8
#endif
#ifndef SE_FUSED_RAY_PRIO
#define SE_FUSED_RAY_PRIO 1   // k_raycast_scan at <= 512^3: raycast waves start at issue priority 1 instead of 0 (A/B: profiles/r04x_fused_ray_prio_ab.log)
#endif
#define SE_SPEC_OF 8  // OFusion march: samples fetched per memory round trip
#ifndef SE_COST_BATCH
#define SE_COST_BATCH 5   // raycast scheduling: cost of a tile = trips of its slowest ray + SE_COST_BATCH * its march batches (fitted against per-wave clocks in r02)
#endif

// ------------------------------------------------------------------------------------------
// index insertion (replaces Octree::allocate / allocate_level, se_core/include/se/octree.hpp:792-856)
// ------------------------------------------------------------------------------------------
// Wave-aggregated counters (r06; north-star: "wave-ballot compaction of newly-allocated blocks").  Pool slots and key-list slots are handed out by
// global counters, and one word takes ~90 atomics per microsecond (se_mark_dilated's note): frame 0 at 2048^3 inserts 402 k blocks, the stress stream
// 1-3 k per frame at 1024^3.  The lanes of a wave that reach the same insertion point together (`mine` = this lane takes a slot; any divergent context:
// the ballot sees the lanes that are active here) therefore share ONE atomic: the first of them adds their number, every lane takes base + its rank among
// them (mbcnt).  Which lane gets which slot was never specified (pool order is the reference's race too, memory_pool.hpp:71); sets are unchanged.
#ifndef SE_WAVE_AGG
#define SE_WAVE_AGG 1      // 0: one atomic per lane (the r05 form; A/B: profiles/r06a)
#endif
#ifndef SE_DEFER_MARK
#define SE_DEFER_MARK 1    // 0: beam-start marks in place even when the occupancy bits are deferred (the r05 form); 2: no marks at all (scan bisect; needs SE_HIP_BEAM=0 SE_HIP_OF_LEAP=0)
#endif
template <typename T>
__device__ __forceinline__ T se_wave_take(T* counter, bool mine) {
  if (!SE_WAVE_AGG) return mine ? atomicAdd(counter, (T)1) : (T)0;
  const unsigned long long wm = __ballot(mine);
  if (!mine) return (T)0;
  const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(wm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wm, 0u));
  const int lead = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(wm));
  T base = (T)0;
  if (rank == 0u) base = atomicAdd(counter, (T)__builtin_popcountll(wm));
  if (sizeof(T) == 8) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)base & 0xFFFFFFFFull), lead);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)base >> 32), lead);
    return (T)((((unsigned long long)hi) << 32) | lo) + (T)rank;
  }
  return (T)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)base, lead) + (T)rank;
}

// Creates every missing ancestor of the octant (x,y,z)@level.  A thread that loses the CAS
// stops: the winner keeps walking up, so all ancestors exist when the kernel ends.
// `alive`: this lane created the octant (the other lanes of the wave only take part in the shared counter update).
__device__ __forceinline__ void se_ensure_ancestors(const DevMap& m, int level, int x, int y, int z, bool alive = true) {
  for (int l = level - 1; l >= 1; --l) {
    x >>= 1; y >>= 1; z >>= 1;
    uint32_t* e = m.tab + tab_index(m, l, x, y, z);
    bool won = false;
    if (alive) won = atomicCAS(e, 0u, SE_PENDING) == 0u;
    alive = won;
    if (__ballot(won) == 0ull) break;
    const uint32_t nid = se_wave_take(&m.ctr[C_NODES], won);
    if (!won) continue;
    if (nid >= m.cap_nodes) { atomicSub(&m.ctr[C_NODES], 1u); m.ctr[C_OVERFLOW] = 1u; atomicExch(e, 0u); alive = false; continue; }   // counter stays bounded; reported by the next API call (SE_HIP_E_CAPACITY)
    m.npos[nid] = pack_pos(x, y, z);
    m.nlevel[nid] = (uint8_t)l;
    occ_set(m, l, x, y, z);
    atomicExch(e, nid + 1u);
  }
}

// Inserts the octant (x,y,z)@level (block if level == leaf_level, else an internal node with no
// children yet) if absent.  Returns true if this thread created it.
// A new block marks its cell and the 26 around it in a dilated bitmap of level C (cbits: the coarse grid, fbits: the block grid itself; se_device.h).
// Per (y, z) neighbour the three x-neighbours are adjacent bits of one word, or of two where they straddle a boundary: atomic ORs whose result nobody
// reads -- the inserting lane never waits for them.  A rolled loop: the call sits in the rare insertion branch of the scans' unrolled flush code.
// How NOT to do it, both measured: (1) looking at each word before writing it, word by word, is 27 dependent round trips on one lane and doubled the
// scan's duration (profiles/r05l, r05n; three batched reads per bitmap still cost the stress stream 11 % of its frame rate, profiles/r05s); (2) ORs
// without any test are thousands of atomics per frame onto the few hot words of the COARSE bitmap at 2048^3 -- one word takes ~90 atomics per
// microsecond -- and the scan beside the raycast went from 272 to 442 us (profiles/r05r_configs.log).  Hence se_mark_coarse below: the coarse dilation
// is done once per coarse cell (one read of the cell's own "has a block" bit decides), the fine one per block.
__device__ __forceinline__ void se_mark_dilated(uint32_t* bits, int C, int cx, int cy, int cz) {
  const int n = 1 << C;
  const int lo = max(cx - 1, 0), hi = min(cx + 1, n - 1);
#pragma clang loop unroll(disable)
  for (int k = 0; k < 9; ++k) {
    const int uy = cy + (k % 3) - 1, uz = cz + (k / 3) - 1;
    if ((unsigned)uy >= (unsigned)n || (unsigned)uz >= (unsigned)n) continue;
    const uint32_t i0 = (((uint32_t)uz << (2 * C)) | ((uint32_t)uy << C)) + (uint32_t)lo;
    const unsigned long long run = ((1ull << (hi - lo + 1)) - 1ull) << (i0 & 31u);     // bits lo .. hi relative to word i0 >> 5
    atomicOr(bits + (i0 >> 5), (uint32_t)run);
    if ((uint32_t)(run >> 32)) atomicOr(bits + (i0 >> 5) + 1, (uint32_t)(run >> 32));
  }
}
__device__ __forceinline__ void se_mark_coarse(const DevMap& m, int bx, int by, int bz) {
  const int C = m.clevel, sh = m.leaf_level - C;
  const int cx = bx >> sh, cy = by >> sh, cz = bz >> sh;
  // cown = the undilated coarse occupancy ("a block exists IN this coarse cell"), the second half of the cbits allocation: whoever finds the bit set
  // knows that an earlier block of the same cell has dilated it already (two racing first blocks both do it: idempotent)
  uint32_t* cown = m.cbits + ((size_t)1 << (3 * C)) / 32 + (C < 2 ? 1 : 0);
  const uint32_t idx = ((uint32_t)cz << (2 * C)) | ((uint32_t)cy << C) | (uint32_t)cx, bit = 1u << (idx & 31u);
  if (!(*(volatile uint32_t*)(cown + (idx >> 5)) & bit)) {
    atomicOr(cown + (idx >> 5), bit);
    se_mark_dilated(m.cbits, C, cx, cy, cz);
  }
  if (m.fbits) { const int fl = se_flevel(m), fs = m.leaf_level - fl; se_mark_dilated(m.fbits, fl, bx >> fs, by >> fs, bz >> fs); }
}

// `want` = false: the lane only accompanies the others of its wave (shared counter updates, se_wave_take); the callers pass the lanes that found the
// entry empty.  A block's beam-start marks (se_mark_coarse) are left to se_occ_commit (m.defer_mark: every allocation scan -- only a raycast reads the
// marks, and the sweep in front of it walks the scan's key list anyway) -- r05 marked in place, up to 18 atomics and a dependent read per new block on the
// scan's critical path (profiles/r06a_insert_ab.log); a block inserted beside a raycast holds initValue() and is invisible to it either way.
__device__ __forceinline__ bool se_insert_octant(const DevMap& m, int level, int x, int y, int z, bool want = true) {
  uint32_t* e = m.tab + tab_index(m, level, x, y, z);
  bool won = false;
  if (want) won = atomicCAS(e, 0u, SE_PENDING) == 0u;
  if (__ballot(won) == 0ull) return false;
  const bool leaf = level == m.leaf_level;
  if (__ballot(won && leaf) != 0ull) {
    const uint32_t idx = se_wave_take(&m.ctr[C_BLOCKS], won && leaf);
    if (won && leaf) {
      if (idx >= m.cap_blocks) { atomicSub(&m.ctr[C_BLOCKS], 1u); m.ctr[C_OVERFLOW] = 1u; atomicExch(e, 0u); won = false; }
      else {
        const uint32_t bp = pack_pos(x, y, z);
        const uint32_t slot = block_slot(m, idx, bp);
        m.bpos[idx] = bp;
        m.bactive[slot] = 1;  // allocate_level: active(true), octree.hpp:841
        occ_set(m, level, x, y, z);
        { const uint32_t lin = block_linear(m, x, y, z); atomicOr(&m.lbits[lin >> 5], 1u << (lin & 31u)); }   // (never deferred: a reader that sees the bit early finds PENDING or a brick of initValue())
        if (SE_DEFER_MARK == 0 || (SE_DEFER_MARK == 1 && !m.defer_mark)) se_mark_coarse(m, x, y, z);
        atomicExch(e, slot + 1u);
      }
    }
  }
  if (__ballot(won && !leaf) != 0ull) {
    const uint32_t nid = se_wave_take(&m.ctr[C_NODES], won && !leaf);
    if (won && !leaf) {
      if (nid >= m.cap_nodes) { atomicSub(&m.ctr[C_NODES], 1u); m.ctr[C_OVERFLOW] = 1u; atomicExch(e, 0u); won = false; }
      else {
        m.npos[nid] = pack_pos(x, y, z);
        m.nlevel[nid] = (uint8_t)level;
        occ_set(m, level, x, y, z);
        atomicExch(e, nid + 1u);
      }
    }
  }
  se_ensure_ancestors(m, level, x, y, z, won);
  return won;
}

#define SE_KEY_ACTIVATE (1ull << 63)  // list entry = "set VoxelBlock::active_ of this existing block"
// (`mine` = false: the lane only accompanies its wave, see se_wave_take)
__device__ __forceinline__ void se_append_key(const DevMap& m, int level, int x, int y, int z, unsigned long long flag = 0ull, bool mine = true) {
  if (__ballot(mine) == 0ull) return;
  const unsigned long long idx = se_wave_take(&m.newkeys[0], mine);
  if (!mine) return;
  if (idx < m.cap_keys) m.newkeys[1 + idx] = se_make_key(x, y, z, level, m.max_level) | flag;
  else m.ctr[C_OVERFLOW] = 2u;
}
// n->active(true) of the allocation scans.  A row-sharded replica only sees its own rays, so a block
// it switches from inactive to active is also reported to the peers (which cannot see that ray).
__device__ __forceinline__ bool se_set_active_once(const DevMap& m, uint32_t slot) {
  // one winner per block: the thread that flips the byte from 0 (bytes are only ever written whole, by this scan and by the sweep)
  uint32_t* w = (uint32_t*)m.bactive + (slot >> 2);
  const uint32_t bit = 1u << (8u * (slot & 3u));
  return (atomicOr(w, bit) & (0xFFu << (8u * (slot & 3u)))) == 0u;
}
__device__ __forceinline__ void se_mark_active(const DevMap& m, uint32_t slot, bool sharded, int bx, int by, int bz) {
  if (sharded) { if (se_set_active_once(m, slot)) se_append_key(m, m.leaf_level, bx, by, bz, SE_KEY_ACTIVATE); }   // one key per block, not one per ray
  else m.bactive[slot] = 1;
}

template <bool STATS> __device__ __forceinline__ void se_stat_add(const DevMap& m, int which, unsigned long long v) {
  if (STATS) {
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&m.stats[which], v);
  }
}

// Host-resident input (r06).  se_hip_upload_depth / se_hip_upload_depth_mm leave the caller's image in pinned host memory; the first kernel that
// needs float_depth_ reads it from there (zero copy over PCIe) and materialises the device image `out` on the way: the allocation scan, one thread per
// pixel, is that kernel on the frame path -- no DMA packet, no conversion launch, no second queue between the sensor buffer and the scan.  kind 1 =
// uint16 millimetres of an image `ratio` times the computation size (mm2metersKernel, se_denseslam/src/preprocessing.cpp:161-188: in[x * ratio + in_w * y
// * ratio] / 1000.0f), 2 = float metres (float_depth_ itself), 0 = nothing pending: read the device image.
struct DepthSrc { const void* host; float* out; int kind, in_w, ratio; };
__device__ __forceinline__ float se_depth_at(const DepthSrc& ds, const float* __restrict__ depthmap, int x, int y, int W) {
  if (ds.kind == 0) return depthmap[x + y * W];
  float d;
  if (ds.kind == 1) d = ((const unsigned short*)ds.host)[x * ds.ratio + ds.in_w * y * ds.ratio] / 1000.0f;
  else d = ((const float*)ds.host)[x + y * W];
  ds.out[x + y * W] = d;
  return d;
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
  int of_lvl[3];            // min(depth, leaf level) of the three stages (fetch_octant stops at the leaves) ...
  uint32_t of_off[3];       // ... and the offset of that level in the index pyramid (DevMap::off is never indexed dynamically on the device)
  int sharded;      // this replica scans only part of the image: report re-activated blocks to the peers
};

// ------------------------------------------------------------------------------------------
// SDF allocation scan: buildAllocationList (se_denseslam/src/kfusion/alloc_impl.hpp:54-118)
// fused with Octree::allocate.  One thread per pixel; every band step is classified by the block it
// falls into; a block that does not exist is inserted (one winner per block), one that does gets
// VoxelBlock::active_ set.
//
// In steady state 6.6 M band steps find ~40 new blocks, and all but a handful of the blocks they cross
// are already active (the previous sweep left every visible block active), so the kernel is organised
// around that: pass 1 walks the band with the reference's float arithmetic and only records the
// distinct blocks in LDS (no memory access, no branch on memory); pass 2 fetches the `active` bytes of
// all of them in one round trip (dense bricks: the flag's index is the block's grid index, no look-up)
// and goes to the index only for the few that are not active yet.  The result -- block set, active
// flags, key list as a set -- is what probing the index at every step gives.
// ------------------------------------------------------------------------------------------
#ifndef SE_SCAN_SLOTS
#define SE_SCAN_SLOTS 8     // distinct blocks of one ray buffered before their flags are fetched
#endif
template <bool STATS, bool DENSE>
__device__ __forceinline__ void se_scan_flush(const DevMap& m, const AllocArgs& a, const uint32_t* s_blk, int nb, unsigned long long& newk) {
  const int L = m.leaf_level;
  const uint32_t mask = (1u << L) - 1u;
  uint32_t lin[SE_SCAN_SLOTS];
  uint32_t val[SE_SCAN_SLOTS];
#pragma unroll
  for (int k = 0; k < SE_SCAN_SLOTS; ++k) lin[k] = s_blk[k * SE_WG_SCAN];
#pragma unroll
  for (int k = 0; k < SE_SCAN_SLOTS; ++k) {
    const uint32_t l = k < nb ? lin[k] : lin[0];
    val[k] = DENSE ? (uint32_t)m.bactive[l] : m.tab[m.leaf_off + l];   // dense: active flag; pooled: index entry
  }
#pragma unroll
  for (int k = 0; k < SE_SCAN_SLOTS; ++k) {
    if (k >= nb) continue;
    const int bx = (int)(lin[k] & mask), by = (int)((lin[k] >> L) & mask), bz = (int)(lin[k] >> (2 * L));
    uint32_t e;
    if (DENSE) {
      if (val[k]) continue;                        // exists and is active already
      // (r06, measured and dropped: the index entries of all the not-yet-active slots fetched together in front of this loop instead of one by one behind the
      // previous slot's insertion -- stress stream scan 21.4 -> 27.3 us at 512^3: the 64 rays of a wave meet the same new blocks at neighbouring slots, and an
      // entry read early is still 0 for a block a neighbouring lane inserts one slot earlier, so more lanes go through the insertion path to lose its CAS)
      e = m.tab[m.leaf_off + lin[k]];
    } else {
      e = val[k];
    }
    if (e == 0u) {
      if (se_insert_octant(m, L, bx, by, bz)) { se_append_key(m, L, bx, by, bz); ++newk; }
    } else if (e != SE_PENDING) {
      // n->active(true), alloc_impl.hpp:109.  A row-sharded replica only sees its own rays, so a block it
      // switches from inactive to active is also reported to the peers -- once, by the thread that flipped it.
      if (a.sharded) { if (se_set_active_once(m, e - 1u)) se_append_key(m, L, bx, by, bz, SE_KEY_ACTIVATE); }
      else m.bactive[e - 1u] = 1;
    }
  }
}
// r03, measured and dropped (profiles/r03_ab2_scan_pooled.log): (a) recognising a step that stays in the previous step's block by six
// float compares against the block's bounds, the floor / range test / convert / shift sequence only on a crossing -- the 64 rays
// of a wave cross block boundaries at different steps, so the wave executes both paths on nearly every step: 34.6 vs 32.2 us at
// 512^3, 62.5 vs 57.4 us at 1024^3, 260 vs 247 us at 2048^3 beside the raycast; (b) two / four lanes per pixel, each walking
// half / a quarter of the band after replaying the additions in front of it (shorter dependent chains, a single round of
// waves): 34.0 / 38.3 us, 54.8 / 59.0 us, 282 / 289 us.  What stays is the straight-line loop body below.
// (int)floorf(x), saturating, in one instruction
__device__ __forceinline__ int se_cvt_flr(float x) { int r; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
// (the body of the kernel, so that k_raycast_scan can run it as part of a raycast launch: `bid` = workgroup index within the scan's grid,
// s_blk_all = SE_SCAN_SLOTS * SE_WG_SCAN words of LDS)
template <bool STATS, bool DENSE>
__device__ __forceinline__ void se_scan_sdf_wg(const DevMap& m, const float* __restrict__ depthmap, const AllocArgs& a, uint32_t* s_blk_all, int bid, const DepthSrc& ds) {
  uint32_t* s_blk = s_blk_all + threadIdx.x;
  unsigned long long probes = 0, newk = 0;
  int x, y;
  bool in_image;
  {
    // a wave scans an 8x8 pixel tile: its rays cross the same 1-2 blocks per step (64 pixels of a row measured the same) ...
    const int lane = threadIdx.x & 63;
    const int tile = bid * (SE_WG_SCAN / 64) + (threadIdx.x >> 6);
    if (ds.kind == 0) {
      const int tiles_x = (a.W + 7) >> 3;
      x = (tile % tiles_x) * 8 + (lane & 7);
      y = a.row_begin + (tile / tiles_x) * 8 + (lane >> 3);
    } else {
      // ... and 64 pixels of a row when the image is still in host memory (DepthSrc): the wave's read is one contiguous 128 / 256-byte piece -- host
      // memory is not cached on the device, the 16-byte row pieces of an 8x8 tile would each cross PCIe as a request of their own
      const int tiles_x = (a.W + 63) >> 6;
      x = (tile % tiles_x) * 64 + lane;
      y = a.row_begin + tile / tiles_x;
    }
    in_image = x < a.W && y < a.row_end;
  }
  if (in_image) {
    const float depth = se_depth_at(ds, depthmap, x, y, a.W);
    if (!(depth == 0)) {
      const f3 worldVertex = m34_mul_h(a.kpose, {(x + 0.5f) * depth, (y + 0.5f) * depth, depth});
      const f3 camera = {a.cam[0], a.cam[1], a.cam[2]};
      const f3 direction = f3_normalized(f3_sub(camera, worldVertex));
      const f3 origin = f3_sub(worldVertex, f3_scale(a.band * 0.5f, direction));
      const f3 step = f3_div(f3_scale_r(direction, a.band), (float)a.num_steps);
      f3 voxelPos = origin;
      uint32_t last = 0xFFFFFFFFu;   // last block recorded (probing it again changes nothing)
      int nb = 0;
      // floor(s) as an integer in one instruction (v_cvt_flr_i32_f32 saturates, so a coordinate beyond +-2^31 stays outside);
      // 0 <= floor(s.a) < size on all three axes <=> the OR of the three integers has no bit at or above log2(size) (size is a
      // power of two; a negative integer has its sign bit set).  The reference's float tests are false for NaN, the conversion
      // gives 0: a ray with a non-finite origin or step (depth = inf / NaN) is skipped up front, as the reference's tests would
      // skip every one of its steps -- with finite origin and step no position is NaN.
      const bool finite = fabsf(origin.x) < INFINITY && fabsf(origin.y) < INFINITY && fabsf(origin.z) < INFINITY &&
                          fabsf(step.x) < INFINITY && fabsf(step.y) < INFINITY && fabsf(step.z) < INFINITY;
      const uint32_t hi_mask = ~(uint32_t)(m.size - 1);
      const int L = m.leaf_level;
      for (int i = 0; finite && i < a.num_steps; ++i) {
        const f3 s = f3_scale_r(voxelPos, a.inv_voxel);
        const int ix = se_cvt_flr(s.x), iy = se_cvt_flr(s.y), iz = se_cvt_flr(s.z);
        if ((((uint32_t)ix | (uint32_t)iy | (uint32_t)iz) & hi_mask) == 0u) {
          ++probes;
          const uint32_t lin = ((((uint32_t)iz >> 3) << L | ((uint32_t)iy >> 3)) << L) | ((uint32_t)ix >> 3);
          if (lin != last) {
            last = lin;
            if (nb == SE_SCAN_SLOTS) { se_scan_flush<STATS, DENSE>(m, a, s_blk, nb, newk); nb = 0; }
            s_blk[nb * SE_WG_SCAN] = lin;
            ++nb;
          }
        }
        voxelPos = f3_add(voxelPos, step);
      }
      if (nb) se_scan_flush<STATS, DENSE>(m, a, s_blk, nb, newk);
    }
  }
  se_stat_add<STATS>(m, S_PROBES, probes);
  se_stat_add<STATS>(m, S_NEWKEYS, newk);
}
template <bool STATS, bool DENSE>
__global__ __launch_bounds__(SE_WG_SCAN) void k_alloc_scan_sdf(DevMap m, const float* __restrict__ depthmap, AllocArgs a, DepthSrc ds) {
  __shared__ uint32_t s_blk_all[SE_SCAN_SLOTS * SE_WG_SCAN];
  se_scan_sdf_wg<STATS, DENSE>(m, depthmap, a, s_blk_all, (int)blockIdx.x, ds);
}

// ------------------------------------------------------------------------------------------
// OFusion allocation scan: buildOctantList (se_denseslam/src/bfusion/alloc_impl.hpp:56-129)
// fused with Octree::allocate.  Marches from behind the surface to the camera with the
// three-stage step size; coarse steps insert childless octants at levels leaf-1 / leaf-2.
// ------------------------------------------------------------------------------------------
template <bool STATS>
__device__ __forceinline__ void se_scan_ofusion_wg(const DevMap& m, const float* __restrict__ depthmap, const AllocArgs& a, int bid, const DepthSrc& ds) {
  const int npix = (a.row_end - a.row_begin) * a.W;
  const int pid = bid * SE_WG_SCAN + threadIdx.x;
  unsigned long long probes = 0, newk = 0;
  if (pid < npix) {
    const int x = pid % a.W;
    const int y = a.row_begin + pid / a.W;
    const float depth = se_depth_at(ds, depthmap, x, y, a.W);
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
      int llvl = -1, lox = -1, loy = -1, loz = -1;  // last octant handled (probing it again changes nothing)
      for (float travelled = 0.f; travelled < dist; travelled += stepsize) {
        const f3 s = f3_scale_r(voxelPos, a.inv_voxel);
        const float vx = floorf(s.x), vy = floorf(s.y), vz = floorf(s.z);
        if ((vx < fsize) && (vy < fsize) && (vz < fsize) && (vx >= 0) && (vy >= 0) && (vz >= 0)) {
          ++probes;
          const int lvl = tree_depth < m.leaf_level ? tree_depth : m.leaf_level;  // fetch_octant stops at the leaves
          const int sh = m.max_level - lvl;
          const int ox = (int)vx >> sh, oy = (int)vy >> sh, oz = (int)vz >> sh;
          if (lvl != llvl || ox != lox || oy != loy || oz != loz) {
            llvl = lvl; lox = ox; loy = oy; loz = oz;
            const uint32_t e = m.tab[tab_index(m, lvl, ox, oy, oz)];
            if (e == 0u) {
              if (se_insert_octant(m, lvl, ox, oy, oz)) { se_append_key(m, lvl, ox, oy, oz); ++newk; }
            } else if (tree_depth >= m.leaf_level && e != SE_PENDING) {
              se_mark_active(m, e - 1u, a.sharded != 0, ox, oy, oz);
            }
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
template <bool STATS>
__global__ __launch_bounds__(SE_WG_SCAN) void k_alloc_scan_ofusion(DevMap m, const float* __restrict__ depthmap, AllocArgs a, DepthSrc ds) {
  se_scan_ofusion_wg<STATS>(m, depthmap, a, (int)blockIdx.x, ds);
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
    const unsigned long long raw = list[1 + i];
    const bool activate = (raw & SE_KEY_ACTIVATE) != 0ull;
    const unsigned long long key = raw & ~SE_KEY_ACTIVATE;
    const int level = (int)(key & 0x1FFull);
    if (level < 1 || level > m.leaf_level) continue;
    const unsigned long long code = key & ~0x1FFull;
    const int sh = m.max_level - level;
    const int x = (int)(se_compact21(code) >> sh), y = (int)(se_compact21(code >> 1) >> sh), z = (int)(se_compact21(code >> 2) >> sh);
    if ((unsigned)x >= (1u << level) || (unsigned)y >= (1u << level) || (unsigned)z >= (1u << level)) continue;
    if (activate) {
      if (level != m.leaf_level) continue;
      const uint32_t e = m.tab[leaf_index(m, x, y, z)];
      if (e != 0u && e != SE_PENDING) m.bactive[e - 1u] = 1;
    } else {
      se_insert_octant(m, level, x, y, z);
    }
  }
}

// Deferred occupancy update: when the allocation scan of frame f+1 runs concurrently with the raycast
// of frame f it inserts into tab[] / the lists but leaves occ[] (which that raycast is walking) and the
// beam-start bitmaps cbits / fbits alone; this kernel then sets the bits of every inserted octant and of
// all its ancestors, and marks the new blocks' neighbourhoods (se_mark_coarse).
struct OccLists { const unsigned long long* lists; int nlists; long long stride_words; };   // [count, keys...] per list
// Done by `nwg` workgroups of the launch only, those from index `first` on: every participating thread reads each list's count word first
// (nlists dependent round trips), which the ~10^4 other waves of a sweep launch need not pay for -- and which workgroup 0 of a sweep, whose
// raycast scheduler is the longest dependent chain of the launch, must not have in front of it (first = 1 there: r05, 6-10 us on the stress stream).
__device__ __forceinline__ void se_occ_commit(const DevMap& m, const OccLists L, unsigned nwg, unsigned first = 0u) {
  if (blockIdx.x < first || blockIdx.x >= first + nwg) return;
  const unsigned wg = blockIdx.x - first;
  for (int li = 0; li < L.nlists; ++li) {
    const unsigned long long* list = L.lists + (long long)li * L.stride_words;
    unsigned long long n = list[0];
    if (n > (unsigned long long)(L.stride_words - 1)) n = (unsigned long long)(L.stride_words - 1);
    for (unsigned long long i = wg * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)nwg * blockDim.x) {
      const unsigned long long raw = list[1 + i];
      if (raw & SE_KEY_ACTIVATE) continue;
      const int level = (int)(raw & 0x1FFull);
      if (level < 1 || level > m.leaf_level) continue;
      const unsigned long long code = raw & ~0x1FFull;
      const int sh = m.max_level - level;
      int x = (int)(se_compact21(code) >> sh), y = (int)(se_compact21(code >> 1) >> sh), z = (int)(se_compact21(code >> 2) >> sh);
      if ((unsigned)x >= (1u << level) || (unsigned)y >= (1u << level) || (unsigned)z >= (1u << level)) continue;
      if (SE_DEFER_MARK == 1 && level == m.leaf_level) se_mark_coarse(m, x, y, z);   // the beam-start bitmaps of a deferred insertion (se_insert_octant)
      for (int l = level; l >= 1; --l) {
        const uint32_t c = occ_code(l, x, y, z);
        atomicOr(&m.occ[c >> 5], 1u << (c & 31u));
        x >>= 1; y >>= 1; z >>= 1;
      }
    }
  }
}
__global__ __launch_bounds__(SE_WG) void k_occ_commit(DevMap m, OccLists L) { se_occ_commit(m, L, gridDim.x); }

// unique_multiscale keeps keys[0] whatever its level (se_core/include/se/algorithms/unique.hpp:64-79):
// when the smallest key of a frame's list (after filter_ancestors) is a coarse octant, the
// reference walks it down to the leaves along child 0.  k_zero_chain finds the three smallest
// distinct keys of the frame's key list(s) (they resolve the ancestor chain that filter_ancestors
// collapses) and inserts that chain.
// One workgroup: three passes "smallest key greater than the previous" over the list(s), then the
// chain insertion by thread 0.
__global__ __launch_bounds__(SE_WG) void k_zero_chain(DevMap m, const unsigned long long* __restrict__ lists, int nlists, long long stride_words) {
  __shared__ unsigned long long s_best[SE_WG / 64];
  __shared__ unsigned long long s_chain[3];
  unsigned long long lower = 0ull;
  for (int pass = 0; pass < 3; ++pass) {
    unsigned long long best = ~0ull;
    for (int li = 0; li < nlists; ++li) {
      const unsigned long long* list = lists + (long long)li * stride_words;
      unsigned long long n = list[0];
      if (n > (unsigned long long)(stride_words - 1)) n = (unsigned long long)(stride_words - 1);
      for (unsigned long long i = threadIdx.x; i < n; i += SE_WG) {
        const unsigned long long k = list[1 + i];
        if (k & SE_KEY_ACTIVATE) continue;
        if ((pass == 0 || k > lower) && k < best) best = k;
      }
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_down(best, o); if (v < best) best = v; }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long b = s_best[0];
      for (int w = 1; w < SE_WG / 64; ++w) if (s_best[w] < b) b = s_best[w];
      s_chain[pass] = b;
    }
    __syncthreads();
    lower = s_chain[pass];
    if (lower == ~0ull) break;
  }
  if (threadIdx.x != 0) return;
  unsigned long long cur = s_chain[0];
  if (cur == ~0ull) return;
  for (int j = 1; j < 3; ++j) {
    const unsigned long long nx = s_chain[j];
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
#define SE_SHARD_SUB 64   // sub-segments (and record counters) of a sharded-sweep send segment
struct IntegArgs {
  float R[9], t[3];        // Tcw = SE3f(pose).inverse()
  float K3[9];             // K.topLeftCorner<3,3>()
  float cam[12];           // rows 0..2 of K * Tcw.matrix()  (in_frustum)
  float delta[3];          // R * (voxel, 0, 0)
  float cdelta[3];         // K3 * delta
  float voxel;
  float mu;                // SDF: mu; OFusion: noiseFactor
  int stats;               // count the swept blocks (se_hip_enable_stats)
  float maxweight;
  float timestamp;         // OFusion
  int W, H;
  const float* bspline;    // OFusion: 1000-entry B-spline CDF table
  const float* logodds;    // OFusion: log2f(s/(1-s)) for every (Q1 index, Q2 index) pair, see se_hip_api.hip
  int fast_div;            // the operands of the sweep's divisions are in the range in which their shared-reciprocal form is the IEEE division (see se_rcp_refined)
  int commit_occ;          // publish the occupancy bits of this frame's (side-stream) allocation scan / commit first
  OccLists occ_lists;      // the key lists whose insertions are published
  uint32_t* ctr_mirror;    // pinned host copy of ctr[] (launch-geometry estimate of the next sweep), may be null
  unsigned long long* zero_count;   // count word of the key list the NEXT allocation scan appends to (the handle's two own lists
                                    // alternate): cleared here, so that no fill kernel sits in front of that scan; may be null
  // raycast scheduling hint (see RayArgs::tile_cost): this launch also turns the previous raycast's per-tile costs into
  // the three priority thresholds of the next one (top 40 % / 15 % / 5 % of the tiles by default); null = hint off
  const unsigned short* tile_cost;
  int n_tiles;
  int* prio_thr;
  int prio_permille[3];    // share of the tiles (per mille) that get priority >= 1 / >= 2 / 3
  uint32_t* ray_order;     // out: the tile PAIRS (one raycast workgroup each) by descending previous cost (RayArgs::ray_order); may be null
  // Sharded sweep (se_hip_set_sweep_shard; SURVEY 8(e) option 4, measured in DESIGN 7): of R replicas, this one updates only
  // the blocks it owns, owner = (bx + by + bz) mod R in block units, and packs each of them into its send segment --
  // [counts][records: position | active << 31][vx bricks][vy bricks] -- which k_apply_bricks writes into the other replicas'
  // maps after the all-gather.  shard_world <= 1: off.  SE_SHARD_SUB counters, each over 1/SE_SHARD_SUB of the records.
  int shard_world, shard_rank;
  unsigned long long* shard_count;
  uint32_t* shard_recs;
  float* shard_vx;
  float* shard_vy;
  uint32_t shard_cap;
};

// One workgroup turns the previous raycast's per-tile costs into the schedule of the next one:
//  * 256-bin histogram of the tile pairs (tiles 2p, 2p+1: the two waves of one raycast workgroup) by their mean cost;
//  * priority thresholds = smallest cost v with #(pairs of mean cost >= v) <= fraction * pairs;
//  * counting sort of the pairs -> order[] = pairs, costliest first (ties in arrival order: scheduling only, never results).
// The function is the critical path of a small sweep launch (it runs in workgroup 0 while the others sweep blocks: r01-r04 it was ~22 us of dependent
// round trips, longer than a 512^3 sweep's blocks), so it is built for latency: the costs are loaded once, all loads of a thread issued together, and
// kept in registers for both passes; one LDS atomic per pair and pass; a chunk of 8 neighbouring tiles goes to a lane far from the lanes that get the
// chunks around it (neighbouring tiles have near-equal costs, and 64 lanes adding to one LDS word are served one after the other); the suffix sums over
// the bins are one bin per thread.  cost[] is padded (16-byte loads); hist = 784 words of LDS.
__device__ __forceinline__ void se_ray_schedule(const unsigned short* __restrict__ cost, int n, int* __restrict__ thr, unsigned* hist, const int* permille,
                                                uint32_t* __restrict__ order) {
  static_assert(SE_WG == 256, "se_ray_schedule: one histogram bin per thread");
  // this workgroup's four waves are a chain of short dependent steps; the block-sweeping waves they share their SIMDs with have arithmetic to issue
  // at every cycle -- at equal priority each step of the chain queues behind them
  __builtin_amdgcn_s_setprio(3);
  unsigned* phist = hist + 256;   // pairs per mean cost
  unsigned* base = hist + 512;    // first slot of each cost in order[]
  unsigned* wsum = hist + 768;    // [4] wave totals of the suffix sums
  int* tmin = (int*)(hist + 776); // [3] thresholds
  const unsigned tid = threadIdx.x;
  // CH chunks of 8 tiles per thread cover 20 480 tiles (1280x960: 19 200); tiles beyond that keep the image-order schedule (order[] = identity there)
  constexpr int CH = 10;
  const int nch = min((n + 7) >> 3, CH * 256);
  const int n_pairs = (n + 1) >> 1;
  // chunk of work item i = i * S mod nch, S a prime that does not divide nch (a bijection of 0 .. nch - 1)
  const int S = (nch % 37) ? 37 : ((nch % 41) ? 41 : 43);
  uint4 q[CH];
  int ck[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int i = (int)tid + 256 * j;
    ck[j] = i < nch ? (int)(((long long)i * S) % nch) : -1;
    q[j] = ck[j] >= 0 ? *(const uint4*)(cost + 8 * ck[j]) : make_uint4(0u, 0u, 0u, 0u);
  }
  phist[tid] = 0u;
  if (tid < 3u) tmin[tid] = 256;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    if (ck[j] < 0) continue;
    const unsigned w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t0 = 8 * ck[j] + 2 * e;
      if (t0 >= n) continue;
      const unsigned c0 = w[e] & 0xFFFFu, c1 = (t0 + 1 < n) ? (w[e] >> 16) : c0;
      atomicAdd(&phist[min((c0 + c1) >> 1, 255u)], 1u);
    }
  }
  __syncthreads();
  // suffix sums over the 256 bins, one bin per thread: b = #(pairs with mean cost >= v)
  const unsigned ln = tid & 63u, wv = tid >> 6;
  const unsigned ph = phist[tid];
  unsigned b = ph;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned ub = __shfl_down(b, o);
    if (ln + o < 64u) b += ub;
  }
  if (ln == 0u) wsum[wv] = b;
  __syncthreads();
  for (unsigned w2 = wv + 1; w2 < 4u; ++w2) b += wsum[w2];
  base[tid] = b - ph;
  if (tid >= 1u) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned lim = (unsigned)((long long)n_pairs * permille[k] / 1000);
      if (b <= lim) atomicMin(&tmin[k], (int)tid);
    }
  }
  __syncthreads();
  if (tid < 3u) thr[tid] = tmin[tid];
  if (order) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (ck[j] < 0) continue;
      const unsigned w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t0 = 8 * ck[j] + 2 * e;
        if (t0 >= n) continue;
        const unsigned c0 = w[e] & 0xFFFFu, c1 = (t0 + 1 < n) ? (w[e] >> 16) : c0;
        const unsigned cc = min((c0 + c1) >> 1, 255u);
        order[base[cc] + atomicSub(&phist[cc], 1u) - 1u] = (uint32_t)(4 * ck[j] + e);   // phist[cc] counts down: a unique slot of cc's range
      }
    }
  }
}

#define SE_LO_DIM 1002  // 0..999 table entries, 1000 = "0" (t < -3), 1001 = "1" (t > 3)

// sdf_update::operator() (se_denseslam/src/kfusion/mapping_impl.hpp:35-65)
__device__ __forceinline__ void se_sdf_apply(const IntegArgs& a, float depthSample, f3 pos, float& vx, float& vy, bool& dirty) {
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
// ---- r05: the sweep's divisions and its square root without the steps that are the identity on its operands ------------------------
// A block voxel pays five IEEE divisions and one IEEE square root (update_block's 1 / cam.z, sdf_update's pos.x / pos.z, pos.y / pos.z,
// diff / mu and the weighted average, kfusion/mapping_impl.hpp:35-65): 71 of the ~95 vector instructions of a slice.  AMDGPU lowers
// n / d (f32, denormals on) to
//     ds = v_div_scale(d, d, n);  ns = v_div_scale(n, d, n);  r0 = v_rcp(ds);  e0 = fma(-ds, r0, 1);  r1 = fma(e0, r0, r0);
//     q0 = ns * r1;  e1 = fma(-ds, q0, ns);  q1 = fma(e1, r1, q0);  e2 = fma(-ds, q1, ns);  q = v_div_fmas(e2, r1, q1);  v_div_fixup(q, d, n)
// v_div_scale hands its operand back unchanged, v_div_fmas is a plain fma and v_div_fixup returns q unless an operand is zero, infinite, NaN
// or denormal, the numerator is below 2^-103 (the rescaling keeps the residuals e1, e2 exact: their grain is ulp(q) ulp(d), which must not
// fall below 2^-149), the quotient's exponent leaves [-126, 96) or 1 / d is denormal.  Outside those cases the
// division IS the eight operations in the middle, three of which (r0, e0, r1) depend on the divisor alone -- so 1 / z, x / z and y / z share
// them (cam.z == pos.z: K's last row is (0, 0, 1)), and diff / mu takes them from a per-wave constant: 17 + 5 instructions instead of 44,
// the SAME operations on the SAME operands, hence the same bits, by construction (no lemma, no table).
// IntegArgs::fast_div is set by the host only when the operands are in that range for every voxel of the volume (se_hip_integrate_sweep:
// finite pose, |pos| < 2^40 everywhere, 2^-30 <= mu <= 2^30, K of the form getCameraMatrix builds); otherwise the kernel instantiation
// with the compiler's divisions runs.  What the range check does not cover is harmless where it happens:
//   * z: a valid voxel has 1e-4 <= pos.z (update_block's own test) -- normal, and 1 / z too; lanes with smaller, NaN or infinite z yield
//     "not valid" or a NaN diff either way (no update);
//   * x / z, y / z with |x| < 2^-100: the quotient is below 2^-86 in both forms, its square underflows to +0 -- the only use of it;
//     a zero numerator gives +-0, squared +0;
//   * diff / mu: diff is +0 or at least 2^-38 in magnitude (a difference of two floats, one of them >= 1e-4, times a factor >= 1); the
//     quotient is only looked at through fminf(1, .), which maps every value >= 1, +inf and NaN (an overflowed q0) to 1, and only when
//     diff > -mu.
// The weighted average (y x + sdf) / (y + 1) keeps the compiler's division: its numerator is a stored voxel value nobody bounds, and its divisor -- an integer
// in 1 .. 256 since the weights are bytes (r06) -- is a different one per voxel: there is no reciprocal to share.
struct SeRcp { float nd, r1; };   // -d, and the once-refined reciprocal of d
__device__ __forceinline__ SeRcp se_rcp_refined(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e0 = __builtin_fmaf(-d, r0, 1.f);
  return {-d, __builtin_fmaf(e0, r0, r0)};
}
__device__ __forceinline__ float se_div_refined(float n, const SeRcp r) {
  const float q0 = n * r.r1;
  const float e1 = __builtin_fmaf(r.nd, q0, n);
  const float q1 = __builtin_fmaf(e1, r.r1, q0);
  const float e2 = __builtin_fmaf(r.nd, q1, n);
  return __builtin_fmaf(e2, r.r1, q1);
}
__device__ __forceinline__ float se_inv_refined(const SeRcp r) {   // 1 / d: q0 = 1 * r1 is r1
  const float e1 = __builtin_fmaf(r.nd, r.r1, 1.f);
  const float q1 = __builtin_fmaf(e1, r.r1, r.r1);
  const float e2 = __builtin_fmaf(r.nd, q1, 1.f);
  return __builtin_fmaf(e2, r.r1, q1);
}
// sqrtf(s) for s >= 1 (the functor's 1 + (x/z)^2 + (y/z)^2), +inf or NaN.  AMDGPU's lowering is: scale by 2^32 if s < 2^-96, r = v_sqrt(s),
// the two neighbours of r tested by exact residuals fma(-r', r, s), scale back, s itself for 0 / +inf.  Here the scaling never triggers and
// +inf comes out of v_sqrt unchanged (both residuals are NaN, both tests false): what is left are the same nine operations.
__device__ __forceinline__ float se_sqrt_ge1(float s) {
  const float r = __builtin_amdgcn_sqrtf(s);
  const float dn = __uint_as_float(__float_as_uint(r) - 1u), up = __uint_as_float(__float_as_uint(r) + 1u);
  const float vp = __builtin_fmaf(-dn, r, s), vs = __builtin_fmaf(-up, r, s);
  float o = (vp <= 0.f) ? dn : r;
  o = (vs > 0.f) ? up : o;
  return o;
}
// sqrt(1 + (x/z)^2 + (y/z)^2) of sdf_update / bfusion_update (mapping_impl.hpp), the factor that turns a depth difference into a distance along the ray
template <bool FAST>
__device__ __forceinline__ float se_ray_factor(f3 pos, const SeRcp rz) {
  if (FAST) return se_sqrt_ge1(1 + sqf(se_div_refined(pos.x, rz)) + sqf(se_div_refined(pos.y, rz)));
  return sqrtf(1 + sqf(pos.x / pos.z) + sqf(pos.y / pos.z));
}

// Branch-free forms for the block sweep: every expression of the functor is evaluated for every lane
// (results of lanes that the reference skips are discarded by the final selects), so that the 8
// z-slices of a lane are 8 independent dependency chains the compiler can interleave.  `root` = se_ray_factor of the voxel.
template <bool FAST>
__device__ __forceinline__ bool se_sdf_apply_nb(const IntegArgs& a, const SeRcp rmu, bool valid, float depthSample, float posz, float root, float& vx, float& vy) {
  const float diff = (depthSample - posz) * root;
  const bool upd = valid && !(depthSample <= 0) && (diff > -a.mu);
  const float sdf = fminf(1.f, FAST ? se_div_refined(diff, rmu) : diff / a.mu);
  const float nx = clampf((vy * vx + sdf) / (vy + 1.f), -1.f, 1.f);
  const float ny = fminf(vy + 1, a.maxweight);
  vx = upd ? nx : vx;
  vy = upd ? ny : vy;
  return upd;
}
__device__ __forceinline__ bool se_bfusion_apply_nb(const IntegArgs& a, bool valid, float depthSample, float posz, float root, float& vx, float& vy) {
  const float diff = (posz - depthSample) * root;
  const float sigma = clampf(a.mu * sqf(posz), 2 * a.voxel, 0.05f);
  const float tt = diff / sigma;
  const int i1 = se_bspline_index(tt), i2 = se_bspline_index(tt - 3);
  const float q1 = a.bspline[i1 < 1000 ? i1 : 0], q2 = a.bspline[i2 < 1000 ? i2 : 0];
  const float s1 = i1 < 1000 ? q1 : (i1 == 1001 ? 1.f : 0.f), s2 = i2 < 1000 ? q2 : (i2 == 1001 ? 1.f : 0.f);
  const float sample = s1 - s2 * 0.5f;
  const bool upd = valid && !(depthSample <= 0) && !(sample == 0.5f);
  const float lo = a.logodds[i1 * SE_LO_DIM + i2];
  const double delta_t = (double)a.timestamp - (double)vy;
  const float dtf = (float)delta_t;
  float fraction = 1.f / (1.f + (dtf / 4.f));
  fraction = std_max(0.5f, fraction);
  const float nx = clampf(vx * fraction + lo, -1000.f, 1000.f);
  vx = upd ? nx : vx;
  vy = upd ? a.timestamp : vy;
  return upd;
}
__device__ __forceinline__ void se_sdf_update(const IntegArgs& a, const float* __restrict__ depthmap, f3 pos, float px_, float py_,
                                              float& vx, float& vy, bool& dirty) {
  const int px = cvt_i32(px_), py = cvt_i32(py_);
  se_sdf_apply(a, depthmap[px + a.W * py], pos, vx, vy, dirty);
}

// bfusion_update::operator() (se_denseslam/src/bfusion/mapping_impl.hpp:157-191)
__device__ __forceinline__ void se_bfusion_apply(const IntegArgs& a, float depthSample, f3 pos, float& vx, float& vy, bool& dirty) {
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
__device__ __forceinline__ void se_bfusion_update(const IntegArgs& a, const float* __restrict__ depthmap, f3 pos, float px_, float py_,
                                                  float& vx, float& vy, bool& dirty) {
  const int px = cvt_i32(px_), py = cvt_i32(py_);
  se_bfusion_apply(a, depthmap[px + a.W * py], pos, vx, vy, dirty);
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
// Blocks: one wave per block, lane = x + 8*y, 8 z-slices per lane: every slice is one coalesced
// 256-byte row of each SoA plane and the 16 voxel loads of a lane are issued before the first use.
// build_active_list's predicate (active || in_frustum) is evaluated per block at the top
// (wave-uniform), update_block's visibility flag is a wave ballot.  Internal nodes (8 corner values
// each) are swept by the same grid afterwards, one thread per corner.
// Per block, over the 8 slices of the lane without branches: project the voxel, decide `valid`, form the pixel index and ISSUE the depth
// gather; then the voxel's ray factor sqrt(1 + (x/z)^2 + (y/z)^2) -- the bulk of the arithmetic, none of it needs the depth sample or the
// voxel, so it runs under the latency of the 8 gathers and the 16 voxel loads; last the update proper.
// Measured alternatives on MI355X (640x480 -> 512^3, ~10 k swept blocks, r01-r04 versions 26-29 us): a 512-thread workgroup per block
// 41-49 us; software prefetch of the next block 32-44 us; 16-byte loads and stores (4 consecutive x per lane) +2 %; two slices per
// v_pk_* instruction (r03) +-0; skipping the y plane of weight-saturated blocks (r04) +-0 (logs under profiles/, history in profiles/README.md).
// FAST: the divisions of stages 1-3 in their shared-reciprocal form (se_rcp_refined above; IntegArgs::fast_div).
// SHARD: the owner-computes variant (IntegArgs::shard_world > 1) -- a template parameter because its packing code costs the
// plain sweep 11 VGPRs.
#ifndef SE_SWEEP_WAVES
#define SE_SWEEP_WAVES 5    // > 0: force the register budget of that many waves per SIMD
#endif
#if SE_SWEEP_WAVES > 0
#define SE_SWEEP_OCC __attribute__((amdgpu_waves_per_eu(SE_SWEEP_WAVES, SE_SWEEP_WAVES)))
#else
#define SE_SWEEP_OCC
#endif
template <bool OFUSION, bool FAST, bool SHARD>
__global__ __launch_bounds__(SE_WG) SE_SWEEP_OCC void k_integrate(DevMap m, const float* __restrict__ depthmap, IntegArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * SE_WG + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * SE_WG) >> 6;
  const uint32_t nblocks = min(m.ctr[C_BLOCKS], m.cap_blocks);
  const uint32_t nnodes = min(m.ctr[C_NODES], m.cap_nodes);
  const int lx = lane & 7, ly = lane >> 3;
  const float fx = (float)lx;
  unsigned long long swept = 0;
  // counters as this sweep sees them -> pinned host memory (posted write; sizes the next sweep's grid
  // without a device-to-host copy between this kernel and the raycast)
  // (by the LAST workgroup: workgroup 0 runs the raycast scheduler below, whose loads would queue up behind the acknowledgement of a store that
  // crosses PCIe)
  if (a.ctr_mirror && blockIdx.x == gridDim.x - 1 && threadIdx.x < C_COUNT) a.ctr_mirror[threadIdx.x] = m.ctr[threadIdx.x];
  if (a.zero_count && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *a.zero_count = 0ull;
  __shared__ unsigned s_hist[784];
  // workgroup 0 computes the next raycast's schedule and therefore takes no blocks -- with its share of them on top it was the last workgroup of
  // the launch to finish.  (r01-r04 it WAS the launch: ~22 us of dependent round trips in se_ray_schedule under a 512^3 sweep whose blocks take 18-21 us)
  const bool scheduler = a.prio_thr != nullptr && gridDim.x > 1;
  const int wskip = scheduler ? SE_WG / 64 : 0;
  if (a.commit_occ) se_occ_commit(m, a.occ_lists, min(gridDim.x - (scheduler ? 1u : 0u), 64u), scheduler ? 1u : 0u);   // nothing in this kernel reads occ[]; the raycast that follows does
  if (scheduler && blockIdx.x == 0) se_ray_schedule(a.tile_cost, a.n_tiles, a.prio_thr, s_hist, a.prio_permille, a.ray_order);
  const SeRcp rmu = se_rcp_refined(a.mu);   // (FAST, SDF: the divisor-only part of diff / mu, once per wave)
  // (the block position of the next iteration is loaded an iteration ahead, and the first one beside the counters rather than behind them: in bounds
  // whatever the counter says.  Voxel loads in front of the active test -- one dependent round trip less per block -- measured +-0 at 512^3, -3 % at
  // 1024^3 and +2 % on the stress stream, where half the blocks are inactive and their bricks would be read for nothing: not done.
  // profiles/r05e_sched_ab.log)
  const uint32_t stride = (uint32_t)(nwaves - wskip);
  uint32_t b = (scheduler && blockIdx.x == 0) ? 0xFFFFFFFFu : (uint32_t)(wave - wskip);
  uint32_t bp_next = m.bpos[min(b, m.cap_blocks - 1u)];
  for (; b < nblocks; b += stride) {
    const uint32_t bp = bp_next;
    bp_next = m.bpos[min(b + stride, m.cap_blocks - 1u)];
    const int bx = (int)(bp & 1023u) << 3, by = (int)((bp >> 10) & 1023u) << 3, bz = (int)(bp >> 20) << 3;
    if (SHARD && (unsigned)((bx >> 3) + (by >> 3) + (bz >> 3)) % (unsigned)a.shard_world != (unsigned)a.shard_rank) continue;
    const uint32_t slot = block_slot(m, b, bp);
    float* px = m.vx + (size_t)slot * 1024 + lane;
    float* py = px + 512;
    uint2* pyb = (uint2*)(m.vx + (size_t)slot * 1024 + 512) + lane;        // SDF: the weights are bytes, this lane's eight z slices consecutive (se_device.h)
    uint2 wy = {0u, 0u};
    float vx[8], vy[8];
    const unsigned char act = m.bactive[slot];
    if (!act && !se_in_frustum(a, bx, by, bz)) continue;
    if (a.stats && lane == 0) ++swept;
    {
#pragma unroll
      for (int zi = 0; zi < 8; ++zi) { vx[zi] = px[zi * 64]; if (OFUSION) vy[zi] = py[zi * 64]; }
      if (!OFUSION) {
        wy = *pyb;
#pragma unroll
        for (int zi = 0; zi < 8; ++zi) vy[zi] = (float)(((zi < 4 ? wy.x : wy.y) >> (8 * (zi & 3))) & 255u);
      }
    }
    bool visible = false;
    const int y = by + ly;
    // update_block (projective_functor.hpp:73-111)
    int pidx[8];
    bool valid[8], upd[8];
    float ds[8], posz[8], root[8];
    // stages 1 + 2 per slice: projection, validity, pixel, the depth gather is issued, then the ray factor (which needs neither the depth sample
    // nor the voxel: it runs under the latency of the gathers and of the 16 voxel loads above)
#pragma unroll
    for (int zi = 0; zi < 8; ++zi) {
      const int z = bz + zi;
      const f3 start = f3_add(m3_mul(a.R, {bx * a.voxel, y * a.voxel, z * a.voxel}), {a.t[0], a.t[1], a.t[2]});
      const f3 pos = f3_add(start, f3_scale(fx, {a.delta[0], a.delta[1], a.delta[2]}));
      float cvx, cvy, inverse_depth;
      SeRcp rz = {0.f, 0.f};
      if (FAST) {
        // K3 = [fx 0 cx; 0 fy cy; 0 0 1] (checked by the host): K3 * start without its products by zero -- (a + 0 * s) + b and a + b differ only in
        // the sign of an exactly-zero sum, which `pix = cam * inverse_depth + 0.5` does not see; row 2 gives cam.z == pos.z
        cvx = (a.K3[0] * start.x + a.K3[2] * start.z) + fx * a.cdelta[0];
        cvy = (a.K3[4] * start.y + a.K3[5] * start.z) + fx * a.cdelta[1];
        rz = se_rcp_refined(pos.z);
        inverse_depth = se_inv_refined(rz);
      } else {
        const f3 camera_voxel = f3_add(m3_mul(a.K3, start), f3_scale(fx, {a.cdelta[0], a.cdelta[1], a.cdelta[2]}));
        cvx = camera_voxel.x; cvy = camera_voxel.y;
        inverse_depth = 1.f / camera_voxel.z;
      }
      const float pixx = cvx * inverse_depth + 0.5f;
      const float pixy = cvy * inverse_depth + 0.5f;
      valid[zi] = !(pos.z < 0.0001f) && !(pixx < 0.5f || pixx > a.W - 1.5f || pixy < 0.5f || pixy > a.H - 1.5f);
      visible = visible || valid[zi];
      // sdf_update / bfusion_update: pixel.cast<int>().  A valid pixel lies in [0.5, W - 1.5] x [0.5, H - 1.5], where the hardware conversion (one
      // instruction) equals the x86 cast; an invalid one reads pixel 0 and is discarded (NaN cannot be valid: it needs camera_voxel.z == 0 == pos.z)
      pidx[zi] = valid[zi] ? se_cvt_hw(pixx) + a.W * se_cvt_hw(pixy) : 0;
      ds[zi] = depthmap[pidx[zi]];
      posz[zi] = pos.z;
      root[zi] = se_ray_factor<FAST>(pos, rz);
    }
    // stage 3: the functor
#pragma unroll
    for (int zi = 0; zi < 8; ++zi)
      upd[zi] = OFUSION ? se_bfusion_apply_nb(a, valid[zi], ds[zi], posz[zi], root[zi], vx[zi], vy[zi])
                        : se_sdf_apply_nb<FAST>(a, rmu, valid[zi], ds[zi], posz[zi], root[zi], vx[zi], vy[zi]);
    // a voxel the functor left alone is written back unchanged only if a neighbour in the same 256-byte
    // row changed (wave-uniform test): no extra traffic for untouched rows, no branch per voxel otherwise
#pragma unroll
    for (int zi = 0; zi < 8; ++zi) {
      if (__ballot(upd[zi]) != 0ull) {
        px[zi * 64] = vx[zi];
        if (OFUSION) py[zi * 64] = vy[zi];
      }
    }
    if (!OFUSION) {
      // the lane's eight weights go back as they came, in one 8-byte store, if the functor touched any voxel of the block
      bool u = false;
#pragma unroll
      for (int zi = 0; zi < 8; ++zi) u = u || upd[zi];
      if (__ballot(u) != 0ull) {
        uint2 o;
        o.x = (uint32_t)(int)vy[0] | ((uint32_t)(int)vy[1] << 8) | ((uint32_t)(int)vy[2] << 16) | ((uint32_t)(int)vy[3] << 24);
        o.y = (uint32_t)(int)vy[4] | ((uint32_t)(int)vy[5] << 8) | ((uint32_t)(int)vy[6] << 16) | ((uint32_t)(int)vy[7] << 24);
        *pyb = o;
      }
    }
    const bool any = __ballot(visible) != 0ull;
    if (lane == 0) m.bactive[slot] = any ? 1 : 0;  // block->active(is_visible)
    if (SHARD) {   // the other replicas get this block's flag and, if anything of it was in view, its voxels
      // record slots are handed out by SE_SHARD_SUB counters, each over its own 1/SE_SHARD_SUB of the segment (one
      // counter for the ~10 k blocks of a frame is 50-70 us of serialised atomics: one word takes ~90 of them per us)
      const uint32_t sub = (uint32_t)wave & (SE_SHARD_SUB - 1), subcap = a.shard_cap / SE_SHARD_SUB;
      uint32_t rec = 0u;
      if (lane == 0) rec = (uint32_t)atomicAdd(a.shard_count + sub, 1ull);
      rec = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec);
      const bool fits = rec < subcap;
      rec += sub * subcap;
      if (fits) {
        if (lane == 0) a.shard_recs[rec] = bp | (any ? 0x80000000u : 0u);
        if (any) {
          float* sx = a.shard_vx + (size_t)rec * 512 + lane;
          float* sy = a.shard_vy + (size_t)rec * 512 + lane;
#pragma unroll
          for (int zi = 0; zi < 8; ++zi) { sx[zi * 64] = vx[zi]; sy[zi * 64] = vy[zi]; }
        }
      } else if (lane == 0) {
        m.ctr[C_OVERFLOW] = 3u;   // brick exchange segment too small: the peers miss this block's update
      }
    }
  }
  if (a.stats && lane == 0 && swept) atomicAdd(&m.stats[S_SWEPT], swept);
  for (uint32_t tid = blockIdx.x * SE_WG + threadIdx.x; tid < nnodes * 8u; tid += gridDim.x * SE_WG)
    se_update_node_corner<OFUSION>(m, depthmap, a, tid);
}

// The receiving side of the sharded sweep: every record of the other replicas' segments (grid.y = segment) is written
// into this replica's map -- the active flag always, the 512 voxels if the owner saw any of them.  One wave per record.
__global__ __launch_bounds__(SE_WG) void k_apply_bricks(DevMap m, const unsigned char* __restrict__ recv, size_t seg_bytes, uint32_t cap, int self) {
  const int seg = blockIdx.y;
  if (seg == self) return;
  const unsigned char* base = recv + (size_t)seg * seg_bytes;
  const unsigned long long* counts = (const unsigned long long*)base;
  const uint32_t* recs = (const uint32_t*)(base + SE_SHARD_SUB * 8);
  const float* bvx = (const float*)(base + SE_SHARD_SUB * 8 + (size_t)cap * 4);
  const float* bvy = bvx + (size_t)cap * 512;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * SE_WG + threadIdx.x) >> 6, nwaves = (gridDim.x * SE_WG) >> 6;
  const uint32_t subcap = cap / SE_SHARD_SUB;
  // wave w walks sub-segment w mod SE_SHARD_SUB (the grid has a multiple of SE_SHARD_SUB waves)
  const uint32_t sub = wave & (SE_SHARD_SUB - 1);
  const uint32_t n = (uint32_t)min(counts[sub], (unsigned long long)subcap);
  // the sender ran out of record slots: it raised its own C_OVERFLOW, and so does every receiver of the truncated segment --
  // all replicas report SE_HIP_E_CAPACITY for the same frame instead of the sender alone
  if (counts[sub] > (unsigned long long)subcap && lane == 0 && wave < SE_SHARD_SUB) m.ctr[C_OVERFLOW] = 3u;
  for (uint32_t j = wave / SE_SHARD_SUB; j < n; j += nwaves / SE_SHARD_SUB) {
    const uint32_t i = sub * subcap + j;
    const uint32_t r = recs[i], bp = r & 0x3FFFFFFFu;
    const int bx = (int)(bp & 1023u), by = (int)((bp >> 10) & 1023u), bz = (int)(bp >> 20);
    const uint32_t e = m.dense ? block_linear(m, bx, by, bz) + 1u : m.tab[leaf_index(m, bx, by, bz)];
    if (e == 0u || e == SE_PENDING) continue;   // (cannot happen: the block sets are equal after the key exchange)
    const uint32_t slot = e - 1u;
    if (lane == 0) m.bactive[slot] = (unsigned char)(r >> 31);
    if (r >> 31) {
      const float* sx = bvx + (size_t)i * 512 + lane;
      const float* sy = bvy + (size_t)i * 512 + lane;
      float* dx = m.vx + (size_t)slot * SE_BRICK_STRIDE + lane;
#pragma unroll
      for (int zi = 0; zi < 8; ++zi) { dx[zi * 64] = sx[zi * 64]; se_st_y(m, (size_t)slot * SE_BRICK_STRIDE + lane + zi * 64, sy[zi * 64]); }
    }
  }
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
  float near_n, far_n;       // nearp / dim, farp / dim (ray_iterator.hpp:101-102)
  float scaled_origin[3];    // origin / dim + 1 (ray_iterator.hpp:79)
  float inv_voxel;   // size / dim   (VolumeTemplate, volume_template.hpp:77-102)
  float grad_scale;  // 0.5f * dim / size
  float epsilon;     // exp2f(-log2(size))
  int min_scale;     // CAST_STACK_DEPTH - log2(size / 8)
  int W, H, row_begin, row_end;
  int cache_levels;  // occupancy levels 1..cache_levels are staged in LDS
  int cache_words;   // = occ_words_upto(cache_levels)
  uint32_t cache_codes;  // heap codes below this value are staged (= 2 * 8^cache_levels)
  int has_deep;          // there are non-leaf levels beyond the staged ones (volumes > 512^3)
  int stack_depth;   // ray stack slots (= leaf level)
  // Beam start (r05; results do not depend on it, see se_beam_start): sample spacing along the tile's centre ray, edge of a coarse cell, its inverse, 1 / dim
  int beam;          // 0 off, 1 coarse stage, 2 coarse + fine stage
  float beam_dt, beam_cell, beam_inv_cell, inv_dim;
  float beam_dt2, beam_cellf, beam_inv_cellf;     // second stage: the block grid itself (DevMap::fbits)
  // OFusion march: leap over block-free space (se_of_leap; results do not depend on it): the block grid dilated by one block, its level, the spacing of the
  // check points along the ray (0.9 block edges); leap_bits = nullptr switches it off
  const uint32_t* leap_bits;
  int leap_level;
  float leap_dt;
  // Scheduling (results do not depend on it).  tile_cost[] = cost of every wave tile (8x8 pixels) in the previous
  // raycast launch, trips + 5 * march batches of its slowest ray.  All waves of a 640x480 launch are resident from the
  // first microsecond, every SIMD works through the 4-5 tiles the dispatcher gives it, and the launch lasts as long as the
  // unluckiest SIMD: with workgroups in image order the cost sums per SIMD spread 3x (115..343 around a median of 179,
  // r02 per-wave records) and the SIMD finish times 24..42 us follow them (correlation 0.85).  The dispatcher hands workgroup
  // i to compute unit i mod n_cus (observed; a speed assumption only), so the integration sweep sorts the workgroups' tile
  // pairs by previous cost (ray_order) and workgroup i takes the pair at position (i / n_cus, i mod n_cus) of a snake deal:
  // every compute unit gets one pair of each cost stratum.  On top, the waves of the costliest tiles -- the silhouettes
  // and depth edges of the previous frame -- run with a raised issue priority: their dependent-load chains are the longest.
  const uint32_t* ray_order;   // ceil(n_tiles / 2) pair indices, costliest first (identity until the first sweep)
  int n_cus;
  // Host gate: the first thread of the launch writes gate_seq to a pinned host word.  This launch starts only when the
  // integration sweep in front of it on the same queue has completed, so a host that sees the word knows that sweep is
  // done and may release the next frame's allocation scan on the other queue -- the dependency sweep(f) -> scan(f+1)
  // without an event-record packet between sweep(f) and raycast(f) on this queue.
  uint32_t* gate;
  uint32_t gate_seq;
  unsigned short* tile_cost;
  int cost_shift;      // tile costs are stored >> cost_shift so that the 256 bins of se_ray_schedule keep their resolution in volumes
                       // > 512^3, whose traversals are 2-4x as long (costs beyond 255 all fell into the last bin)
  const int* prio_thr; // cost thresholds of s_setprio 1 / 2 / 3, refreshed by the integration sweep (se_ray_schedule)
  unsigned long long* wlog;   // -DSE_WAVE_PROBE builds only (tools/wave_timeline.py): per-wave clock records, pinned host memory; null otherwise
};

struct BlkCache { int bx, by, bz; uint32_t e; };
// voxel_traits<T>::initValue() / empty() as register values.  Reading them through the by-value
// DevMap inside `cond ? load : m.init_x` lets the compiler fold both into one FLAT load of a
// selected address (and park the constants in scratch to have an address).
struct FieldConst { float init_x, init_y, empty_x; };
__device__ __forceinline__ FieldConst se_field_const(const DevMap& m) {
  FieldConst f = {m.init_x, m.init_y, m.empty_x};
  asm volatile("" : "+s"(f.init_x), "+s"(f.init_y), "+s"(f.empty_x));
  return f;
}

// slot+1 of the block holding voxel (x,y,z); 0 if outside the volume or (pooled mode) not allocated.
// Dense mode needs no memory access: every brick of the grid exists and unallocated ones hold initValue().
template <bool DENSE>
__device__ __forceinline__ uint32_t se_block_of(const DevMap& m, int x, int y, int z, BlkCache& c) {
  if (!in_volume(m, x, y, z)) return 0u;
  const int bx = x >> 3, by = y >> 3, bz = z >> 3;
  if (DENSE) return block_linear(m, bx, by, bz) + 1u;
  if (bx == c.bx && by == c.by && bz == c.bz) return c.e;
  uint32_t e = m.tab[leaf_index(m, bx, by, bz)];
  e = e == SE_PENDING ? 0u : e;   // (see se_block_entry)
  c.bx = bx; c.by = by; c.bz = bz; c.e = e;
  return e;
}
__device__ __forceinline__ size_t se_voxel_index(uint32_t e, int x, int y, int z) {
  return (size_t)(e - 1u) * SE_BRICK_STRIDE + (size_t)((x & 7) + ((y & 7) << 3) + ((z & 7) << 6));
}

// Memory-level parallelism is what the raycast kernel lives on: a wave is a chain of dependent L2
// round trips, so interp and grad are written as "all block look-ups -> all voxel loads -> math"
// with unconditional loads (a missing block reads a dummy address and the value is replaced
// afterwards) instead of one branch per sample.

// leaf-grid entry of block (bx,by,bz) or 0 outside the volume; `hint` is a block whose entry is
// already known (the block of the preceding get) and saves the load.
template <bool DENSE>
__device__ __forceinline__ uint32_t se_block_entry(const DevMap& m, int bx, int by, int bz, const BlkCache& hint) {
  const int nb = m.size >> 3;
  uint32_t e = 0u;
  if (DENSE) return ((unsigned)bx < (unsigned)nb && (unsigned)by < (unsigned)nb && (unsigned)bz < (unsigned)nb) ? block_linear(m, bx, by, bz) + 1u : 0u;
  if (bx == hint.bx && by == hint.by && bz == hint.bz) e = hint.e;
  else if ((unsigned)bx < (unsigned)nb && (unsigned)by < (unsigned)nb && (unsigned)bz < (unsigned)nb) e = m.tab[leaf_index(m, bx, by, bz)];
  // Pooled bricks: the allocation scan of the NEXT frame may be inserting blocks while this raycast runs (overlap mode).  An
  // entry in flight reads as PENDING -> "not allocated"; one already published points at a brick that still holds
  // initValue() in every voxel (bricks are pre-filled and never recycled), which is exactly what "not allocated" reads as
  // (initValue().x == empty().x for both field types) -- so the race cannot change a result.
  return e == SE_PENDING ? 0u : e;
}

// Octree::interp (octree.hpp:541-563) with gather_points (interp_gather.hpp:107-237): every corner
// is read from the block that contains it; a missing block yields empty().x, the all-cross case
// goes through get_fine -> initValue().x (identical values for both field types).
template <bool DENSE>
__device__ __forceinline__ float se_interp_generic(const DevMap& m, const FieldConst fc, f3 pos, BlkCache& c) {
  const float flx = floorf(pos.x), fly = floorf(pos.y), flz = floorf(pos.z);
  const int bx = cvt_i32(flx), by = cvt_i32(fly), bz = cvt_i32(flz);
  const float fx = pos.x - flx, fy = pos.y - fly, fz = pos.z - flz;
  const int lx = max(bx, 0), ly = max(by, 0), lz = max(bz, 0);
  const bool cx = (lx & 7) == 7, cy = (ly & 7) == 7, cz = (lz & 7) == 7;
  const float missing = (cx && cy && cz) ? fc.init_x : fc.empty_x;
  // phase 1: the (up to 8) blocks the cell touches
  uint32_t e[8];
  e[0] = se_block_entry<DENSE>(m, lx >> 3, ly >> 3, lz >> 3, c);
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const bool need = (!(k & 1) || cx) && (!(k & 2) || cy) && (!(k & 4) || cz);
    e[k] = 0u;
    if (need) e[k] = se_block_entry<DENSE>(m, (lx + (k & 1)) >> 3, (ly + ((k >> 1) & 1)) >> 3, (lz + (k >> 2)) >> 3, c);
  }
  // phase 2: the 8 corner values.  Corner (i, j, k) lies in block (i && cx, j && cy, k && cz) of the 2x2x2 candidates: the entry is picked axis by
  // axis (12 selects for the cell; r04 -- a chain of seven compares and selects per corner before) and the voxel offset is a sum of per-axis terms
  uint32_t ex1[4], exy[2][2][2];     // ex1[q] = the x-upper corner's entry for (y, z) block pair q; exy[i][j][s] after the y step
#pragma unroll
  for (int q = 0; q < 4; ++q) ex1[q] = cx ? e[2 * q + 1] : e[2 * q];
#pragma unroll
  for (int sz = 0; sz < 2; ++sz) {
    exy[0][0][sz] = e[4 * sz];
    exy[1][0][sz] = ex1[2 * sz];
    exy[0][1][sz] = cy ? e[4 * sz + 2] : e[4 * sz];
    exy[1][1][sz] = cy ? ex1[2 * sz + 1] : ex1[2 * sz];
  }
  const uint32_t ox[2] = {(uint32_t)lx & 7u, (uint32_t)(lx + 1) & 7u};
  const uint32_t oy[2] = {((uint32_t)ly & 7u) << 3, ((uint32_t)(ly + 1) & 7u) << 3};
  const uint32_t oz[2] = {((uint32_t)lz & 7u) << 6, ((uint32_t)(lz + 1) & 7u) << 6};
  float p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = k & 1, j = (k >> 1) & 1, kz = k >> 2;
    const uint32_t ek = kz ? (cz ? exy[i][j][1] : exy[i][j][0]) : exy[i][j][0];
    const size_t vi = ek ? (size_t)(ek - 1u) * SE_BRICK_STRIDE + (size_t)(ox[i] + oy[j] + oz[kz]) : 0;
    const float v = m.vx[vi];
    p[k] = ek ? v : missing;
  }
  return (((p[0] * (1 - fx) + p[1] * fx) * (1 - fy) + (p[2] * (1 - fx) + p[3] * fx) * fy) * (1 - fz) +
          ((p[4] * (1 - fx) + p[5] * fx) * (1 - fy) + (p[6] * (1 - fx) + p[7] * fx) * fy) * fz);
}

// Dense grid, sample stencil entirely inside the volume (always, for a point the march can reach): the
// voxel index (bz << 2l | by << l | bx) * 512 + (x & 7) + 8 (y & 7) + 64 (z & 7) is a sum of one term per axis,
// so a stencil costs a few integer operations per axis plus two additions per sample instead of a full index
// computation per sample (the gradient's 32 samples were a quarter of the kernel's vector instructions).
// Same loads, same arithmetic on the values as the generic forms, which stay the fallback.
struct AxisTerm { uint32_t blk, loc; };
__device__ __forceinline__ AxisTerm se_axis_x(int x) { return {(uint32_t)(x >> 3), (uint32_t)(x & 7)}; }
__device__ __forceinline__ AxisTerm se_axis_y(const DevMap& m, int y) { return {(uint32_t)(y >> 3) << m.leaf_level, (uint32_t)(y & 7) << 3}; }
__device__ __forceinline__ AxisTerm se_axis_z(const DevMap& m, int z) { return {(uint32_t)(z >> 3) << (2 * m.leaf_level), (uint32_t)(z & 7) << 6}; }
__device__ __forceinline__ size_t se_axis_index(AxisTerm a, AxisTerm b, AxisTerm c) {
  return (size_t)(a.blk + b.blk + c.blk) * SE_BRICK_STRIDE + (size_t)(a.loc + b.loc + c.loc);
}

// Dense grid: the address of every corner of an interpolation cell follows from the position alone, so
// the SDF march can fetch the corners of a sample together with the sample itself.  Same cell
// arithmetic as se_interp; the block of corner k is ((lx + (k & 1)) >> 3, ...), the block gather_points
// picks in every cross case.
struct InterpCell { float fx, fy, fz, missing; uint32_t ok; };
__device__ __forceinline__ InterpCell se_interp_cell_dense(const DevMap& m, const FieldConst fc, f3 pos, size_t vi[8]) {
  const float flx = floorf(pos.x), fly = floorf(pos.y), flz = floorf(pos.z);
  const int bx = cvt_i32(flx), by = cvt_i32(fly), bz = cvt_i32(flz);
  InterpCell cell;
  cell.fx = pos.x - flx; cell.fy = pos.y - fly; cell.fz = pos.z - flz;
  const int lx = max(bx, 0), ly = max(by, 0), lz = max(bz, 0);
  const bool cx = (lx & 7) == 7, cy = (ly & 7) == 7, cz = (lz & 7) == 7;
  cell.missing = (cx && cy && cz) ? fc.init_x : fc.empty_x;
  cell.ok = 0u;
  const int nb = m.size >> 3;
  const int top = m.size - 1;
  if (lx < top && ly < top && lz < top) {   // the whole cell inside the volume: per-axis terms (see se_axis_index)
    const AxisTerm X[2] = {se_axis_x(lx), se_axis_x(lx + 1)}, Y[2] = {se_axis_y(m, ly), se_axis_y(m, ly + 1)}, Z[2] = {se_axis_z(m, lz), se_axis_z(m, lz + 1)};
#pragma unroll
    for (int k = 0; k < 8; ++k) vi[k] = se_axis_index(X[k & 1], Y[(k >> 1) & 1], Z[k >> 2]);
    cell.ok = 0xFFu;
    return cell;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int x = lx + (k & 1), y = ly + ((k >> 1) & 1), z = lz + (k >> 2);
    const bool ok = (unsigned)(x >> 3) < (unsigned)nb && (unsigned)(y >> 3) < (unsigned)nb && (unsigned)(z >> 3) < (unsigned)nb;
    vi[k] = ok ? se_voxel_index(block_linear(m, x >> 3, y >> 3, z >> 3) + 1u, x, y, z) : 0;
    cell.ok |= ok ? (1u << k) : 0u;
  }
  return cell;
}
__device__ __forceinline__ float se_interp_blend(const InterpCell& c, const float v[8]) {
  float p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = ((c.ok >> k) & 1u) ? v[k] : c.missing;
  const float fx = c.fx, fy = c.fy, fz = c.fz;
  return (((p[0] * (1 - fx) + p[1] * fx) * (1 - fy) + (p[2] * (1 - fx) + p[3] * fx) * fy) * (1 - fz) +
          ((p[4] * (1 - fx) + p[5] * fx) * (1 - fy) + (p[6] * (1 - fx) + p[7] * fx) * fy) * fz);
}

// Octree::grad (octree.hpp:652-737), same term order.  The 48 get() calls of the reference touch
// 32 distinct voxels: per axis the four clamped coordinates {base-1, base, base+1, base+2} (indices
// 0..3 below; "lower" = 1, "upper" = 2), every sample having at least two axes on a central index.
// They lie in at most 2x2x2 blocks.  A voxel of a missing block reads as initValue().x (the cached
// Octree::get(x,y,z,block) falls back to the tree walk, octree.hpp:379-408).
template <bool DENSE>
__device__ __forceinline__ f3 se_grad_generic(const DevMap& m, const FieldConst fc, f3 pos, BlkCache& c) {
  const float flx = floorf(pos.x), fly = floorf(pos.y), flz = floorf(pos.z);
  const int bx = cvt_i32(flx), by = cvt_i32(fly), bz = cvt_i32(flz);
  const float fx = pos.x - flx, fy = pos.y - fly, fz = pos.z - flz;
  const int hi = m.size - 1;
  const int X[4] = {max(bx - 1, 0), max(bx, 0), min(bx + 1, hi), min(bx + 2, hi)};
  const int Y[4] = {max(by - 1, 0), max(by, 0), min(by + 1, hi), min(by + 2, hi)};
  const int Z[4] = {max(bz - 1, 0), max(bz, 0), min(bz + 1, hi), min(bz + 2, hi)};
  const int xb0 = X[0] >> 3, yb0 = Y[0] >> 3, zb0 = Z[0] >> 3;
  const int xb1 = X[3] >> 3, yb1 = Y[3] >> 3, zb1 = Z[3] >> 3;
  // phase 1: entries of the 2x2x2 candidate blocks (coincident ones are the same cache line)
  uint32_t e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = se_block_entry<DENSE>(m, (k & 1) ? xb1 : xb0, (k & 2) ? yb1 : yb0, (k & 4) ? zb1 : zb0, c);
  // phase 2: the 32 voxels.  A voxel's block among the 2x2x2 candidates is (x block != xb0, y block != yb0, z block != zb0): the brick base is
  // picked axis by axis (x: 16 selects, y: 32, z: one per voxel; r04 -- seven compares and selects per voxel before) and the offset inside the
  // brick is a sum of per-axis terms.  `base` = brick offset + 1 in voxels' units (0: no brick), so that "missing" stays one test.
  bool ux[4], uy[4], uz[4];
  uint32_t ox[4], oy[4], oz[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ux[i] = (X[i] >> 3) != xb0; uy[i] = (Y[i] >> 3) != yb0; uz[i] = (Z[i] >> 3) != zb0;
    ox[i] = (uint32_t)X[i] & 7u; oy[i] = ((uint32_t)Y[i] & 7u) << 3; oz[i] = ((uint32_t)Z[i] & 7u) << 6;
  }
  uint32_t ex[4][4];        // [xi][q]: entry of x-voxel xi in the (y, z) block pair q
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int q = 0; q < 4; ++q) ex[xi][q] = ux[xi] ? e[2 * q + 1] : e[2 * q];
  float V[4][4][4];
#pragma unroll
  for (int yi = 0; yi < 4; ++yi)
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      if (!((xi == 1 || xi == 2) || (yi == 1 || yi == 2))) continue;      // (no voxel of this column is needed)
      const uint32_t exy0 = uy[yi] ? ex[xi][1] : ex[xi][0], exy1 = uy[yi] ? ex[xi][3] : ex[xi][2];
      const uint32_t oxy = ox[xi] + oy[yi];
#pragma unroll
      for (int zi = 0; zi < 4; ++zi) {
        const int central = (xi == 1 || xi == 2) + (yi == 1 || yi == 2) + (zi == 1 || zi == 2);
        if (central < 2) continue;
        const uint32_t ek = uz[zi] ? exy1 : exy0;
        const size_t vi = ek ? (size_t)(ek - 1u) * SE_BRICK_STRIDE + (size_t)(oxy + oz[zi]) : 0;
        const float v = m.vx[vi];
        V[zi][yi][xi] = ek ? v : fc.init_x;
      }
    }
  f3 g;
  g.x = (((V[1][1][2] - V[1][1][0]) * (1 - fx) + (V[1][1][3] - V[1][1][1]) * fx) * (1 - fy) +
         ((V[1][2][2] - V[1][2][0]) * (1 - fx) + (V[1][2][3] - V[1][2][1]) * fx) * fy) * (1 - fz) +
        (((V[2][1][2] - V[2][1][0]) * (1 - fx) + (V[2][1][3] - V[2][1][1]) * fx) * (1 - fy) +
         ((V[2][2][2] - V[2][2][0]) * (1 - fx) + (V[2][2][3] - V[2][2][1]) * fx) * fy) * fz;
  g.y = (((V[1][2][1] - V[1][0][1]) * (1 - fx) + (V[1][2][2] - V[1][0][2]) * fx) * (1 - fy) +
         ((V[1][3][1] - V[1][1][1]) * (1 - fx) + (V[1][3][2] - V[1][1][2]) * fx) * fy) * (1 - fz) +
        (((V[2][2][1] - V[2][0][1]) * (1 - fx) + (V[2][2][2] - V[2][0][2]) * fx) * (1 - fy) +
         ((V[2][3][1] - V[2][1][1]) * (1 - fx) + (V[2][3][2] - V[2][1][2]) * fx) * fy) * fz;
  g.z = (((V[2][1][1] - V[0][1][1]) * (1 - fx) + (V[2][1][2] - V[0][1][2]) * fx) * (1 - fy) +
         ((V[2][2][1] - V[0][2][1]) * (1 - fx) + (V[2][2][2] - V[0][2][2]) * fx) * fy) * (1 - fz) +
        (((V[3][1][1] - V[1][1][1]) * (1 - fx) + (V[3][1][2] - V[1][1][2]) * fx) * (1 - fy) +
         ((V[3][2][1] - V[1][2][1]) * (1 - fx) + (V[3][2][2] - V[1][2][2]) * fx) * fy) * fz;
  return g;  // the caller applies (0.5f * dim / size)
}

// Dense-grid forms of interp / grad on the per-axis terms above; the generic forms are the fallback.
template <bool DENSE>
__device__ __forceinline__ float se_interp(const DevMap& m, const FieldConst fc, f3 pos, BlkCache& c) {
  if (!DENSE) return se_interp_generic<DENSE>(m, fc, pos, c);
  const float flx = floorf(pos.x), fly = floorf(pos.y), flz = floorf(pos.z);
  const int bx = cvt_i32(flx), by = cvt_i32(fly), bz = cvt_i32(flz);
  const int lx = max(bx, 0), ly = max(by, 0), lz = max(bz, 0);
  const int top = m.size - 1;
  if (!(lx < top && ly < top && lz < top)) return se_interp_generic<DENSE>(m, fc, pos, c);   // a corner outside the volume
  const float fx = pos.x - flx, fy = pos.y - fly, fz = pos.z - flz;
  const AxisTerm X[2] = {se_axis_x(lx), se_axis_x(lx + 1)}, Y[2] = {se_axis_y(m, ly), se_axis_y(m, ly + 1)}, Z[2] = {se_axis_z(m, lz), se_axis_z(m, lz + 1)};
  float p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = m.vx[se_axis_index(X[k & 1], Y[(k >> 1) & 1], Z[k >> 2])];
  return (((p[0] * (1 - fx) + p[1] * fx) * (1 - fy) + (p[2] * (1 - fx) + p[3] * fx) * fy) * (1 - fz) +
          ((p[4] * (1 - fx) + p[5] * fx) * (1 - fy) + (p[6] * (1 - fx) + p[7] * fx) * fy) * fz);
}

template <bool DENSE>
__device__ __forceinline__ f3 se_grad(const DevMap& m, const FieldConst fc, f3 pos, BlkCache& c) {
  if (!DENSE) return se_grad_generic<DENSE>(m, fc, pos, c);
  const float flx = floorf(pos.x), fly = floorf(pos.y), flz = floorf(pos.z);
  const int bx = cvt_i32(flx), by = cvt_i32(fly), bz = cvt_i32(flz);
  const int hi = m.size - 1;
  // indices 0, 1 are clamped from below only (octree.hpp:658-663): they leave the volume when the point does
  if (!(bx <= hi && by <= hi && bz <= hi)) return se_grad_generic<DENSE>(m, fc, pos, c);
  const float fx = pos.x - flx, fy = pos.y - fly, fz = pos.z - flz;
  const AxisTerm X[4] = {se_axis_x(max(bx - 1, 0)), se_axis_x(max(bx, 0)), se_axis_x(min(bx + 1, hi)), se_axis_x(min(bx + 2, hi))};
  const AxisTerm Y[4] = {se_axis_y(m, max(by - 1, 0)), se_axis_y(m, max(by, 0)), se_axis_y(m, min(by + 1, hi)), se_axis_y(m, min(by + 2, hi))};
  const AxisTerm Z[4] = {se_axis_z(m, max(bz - 1, 0)), se_axis_z(m, max(bz, 0)), se_axis_z(m, min(bz + 1, hi)), se_axis_z(m, min(bz + 2, hi))};
  float V[4][4][4];
#pragma unroll
  for (int zi = 0; zi < 4; ++zi)
#pragma unroll
    for (int yi = 0; yi < 4; ++yi)
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        const int central = (xi == 1 || xi == 2) + (yi == 1 || yi == 2) + (zi == 1 || zi == 2);
        if (central < 2) continue;
        V[zi][yi][xi] = m.vx[se_axis_index(X[xi], Y[yi], Z[zi])];
      }
  f3 g;
  g.x = (((V[1][1][2] - V[1][1][0]) * (1 - fx) + (V[1][1][3] - V[1][1][1]) * fx) * (1 - fy) +
         ((V[1][2][2] - V[1][2][0]) * (1 - fx) + (V[1][2][3] - V[1][2][1]) * fx) * fy) * (1 - fz) +
        (((V[2][1][2] - V[2][1][0]) * (1 - fx) + (V[2][1][3] - V[2][1][1]) * fx) * (1 - fy) +
         ((V[2][2][2] - V[2][2][0]) * (1 - fx) + (V[2][2][3] - V[2][2][1]) * fx) * fy) * fz;
  g.y = (((V[1][2][1] - V[1][0][1]) * (1 - fx) + (V[1][2][2] - V[1][0][2]) * fx) * (1 - fy) +
         ((V[1][3][1] - V[1][1][1]) * (1 - fx) + (V[1][3][2] - V[1][1][2]) * fx) * fy) * (1 - fz) +
        (((V[2][2][1] - V[2][0][1]) * (1 - fx) + (V[2][2][2] - V[2][0][2]) * fx) * (1 - fy) +
         ((V[2][3][1] - V[2][1][1]) * (1 - fx) + (V[2][3][2] - V[2][1][2]) * fx) * fy) * fz;
  g.z = (((V[2][1][1] - V[0][1][1]) * (1 - fx) + (V[2][1][2] - V[0][1][2]) * fx) * (1 - fy) +
         ((V[2][2][1] - V[0][2][1]) * (1 - fx) + (V[2][2][2] - V[0][2][2]) * fx) * fy) * (1 - fz) +
        (((V[3][1][1] - V[1][1][1]) * (1 - fx) + (V[3][1][2] - V[1][1][2]) * fx) * (1 - fy) +
         ((V[3][2][1] - V[1][2][1]) * (1 - fx) + (V[3][2][2] - V[1][2][2]) * fx) * fy) * fz;
  return g;  // the caller applies (0.5f * dim / size)
}

struct RaySpan { float tcmin, tmax; int trips; };
// se::ray_iterator (se_core/include/se/ray_iterator.hpp:53-250) up to the first leaf, on the occupancy
// bits.  Same float arithmetic on t and pos as the reference; what is restated is the integer side:
//  * a node is its heap code (occ_code), the child test is one bit of one word;
//  * `idx` is not carried: pos is the node's corner in [1, 2)^3, an exact multiple of the node size, so
//    idx bit a is bit `scale` of the mantissa of pos.a (which is how the reference's pop recomputes it);
//  * advance: subtracting scale_exp2 flips exactly bit `scale` unless it borrows, so old ^ new of the
//    three coordinates is the reference's differing_bits, and "leaves the parent" (idx & step_mask) is
//    "some bit above `scale` changed".
struct SeNoHook { __device__ __forceinline__ void operator()() const {} };
// `before_loop` runs between the ray set-up and the traversal loop (the raycast kernel finishes the LDS staging of the
// occupancy bits there, so that the staging loads fly under the set-up arithmetic); `live` = false skips the loop.
template <bool SHALLOW, typename Hook = SeNoHook>   // SHALLOW: the caller guarantees a.has_deep == 0 (every non-leaf level is staged in LDS)
__device__ __forceinline__ RaySpan se_first_leaf(const DevMap& m, const RayArgs& a, f3 origin, f3 direction,
                                                 const uint32_t* s_occ, uint32_t* s_par, float* s_tmax, Hook before_loop = Hook(), bool live = true) {
  const int tid = threadIdx.x;
  f3 pos = {1.0f, 1.0f, 1.0f};
  uint32_t parent = 1u;  // root
  float scale_exp2 = 0.5f;
  int scale = 22;
  const float eps = a.epsilon;
  f3 d;
  d.x = fabsf(direction.x) < eps ? copysignf(eps, direction.x) : direction.x;
  d.y = fabsf(direction.y) < eps ? copysignf(eps, direction.y) : direction.y;
  d.z = fabsf(direction.z) < eps ? copysignf(eps, direction.z) : direction.z;
  const f3 scaled_origin = {a.scaled_origin[0], a.scaled_origin[1], a.scaled_origin[2]};   // origin / dim + 1: the same for every ray, formed on the host
  const f3 t_coef = f3_scale(-1.f, {1.f / fabsf(d.x), 1.f / fabsf(d.y), 1.f / fabsf(d.z)});
  f3 t_bias = f3_mul(t_coef, scaled_origin);
  uint32_t om = 0u;     // octant_mask ^ 7
  if (d.x > 0.0f) { om ^= 1u; t_bias.x = 3.0f * t_coef.x - t_bias.x; }
  if (d.y > 0.0f) { om ^= 2u; t_bias.y = 3.0f * t_coef.y - t_bias.y; }
  if (d.z > 0.0f) { om ^= 4u; t_bias.z = 3.0f * t_coef.z - t_bias.z; }
  float t_min = fmaxf(fmaxf(2.0f * t_coef.x - t_bias.x, 2.0f * t_coef.y - t_bias.y), 2.0f * t_coef.z - t_bias.z);
  float t_max = fminf(fminf(t_coef.x - t_bias.x, t_coef.y - t_bias.y), t_coef.z - t_bias.z);
  float h = t_max;
  t_min = fmaxf(t_min, a.near_n);   // nearp / dim
  t_max = fminf(t_max, a.far_n);    // farp / dim
  const float tmax_m = t_max * m.dim;
  if (1.5f * t_coef.x - t_bias.x > t_min) pos.x = 1.5f;
  if (1.5f * t_coef.y - t_bias.y > t_min) pos.y = 1.5f;
  if (1.5f * t_coef.z - t_bias.z > t_min) pos.z = 1.5f;
  // a stack slot that was never pushed reads as the first node of its level (code 1 << 3i), t_max 0
  for (int i = 0; i < a.stack_depth; ++i) { s_par[i * SE_WG_RAY + tid] = 1u << (3 * i); s_tmax[i * SE_WG_RAY + tid] = 0.f; }

  uint32_t gw_index = 0xFFFFFFFFu, gw_word = 0u;  // last occupancy word fetched from global memory (levels between cache and leaf)
  uint32_t leaf_byte = 0u;                        // the 8 leaf-level sibling bits of the current parent
  // the 64 leaf bits below the node two levels above the leaves that the ray is in (its children are the leaf
  // parents, byte c of the 8-byte group = the sibling byte of child c): fetched once when that node is entered,
  // a trip or more before the first leaf test needs it, instead of one dependent byte load per leaf parent
  unsigned long long gp_bits = 0ull;
  uint32_t gp_code = 0u;
  const uint8_t* occ_bytes = (const uint8_t*)m.occ;
  // One trip = one node.  Some lanes of a wave descend while others advance on nearly every trip, so
  // both updates are computed for every lane and selected (straight-line code, no exec-mask juggling);
  // only the rare events are branches: leaving a parent (stack read), entering a leaf parent (sibling
  // byte load) and the global occupancy word of volumes > 512^3.
  before_loop();
  const int max_trips = live ? 4096 : 0;
  const bool has_deep = SHALLOW ? false : (a.has_deep != 0);
  int guard = 0;
  // The occupancy word of a trip depends only on (parent, pos, scale), which are final at the end of the previous
  // trip: its LDS read is issued there, so the float work at the top of the trip runs under the LDS latency (a wave
  // in the tail of a launch is alone on its SIMD, with nobody else to cover that latency).
  uint32_t cidx = 0u, child = 0u, word = 0u;
  bool at_leaves = false, deep = false;
#define SE_TRIP_PREFETCH()                                                                                                   \
  do {                                                                                                                       \
    const uint32_t us_ = (uint32_t)scale;                                                                                    \
    cidx = (__builtin_amdgcn_ubfe(__float_as_uint(pos.x), us_, 1u) | (__builtin_amdgcn_ubfe(__float_as_uint(pos.y), us_, 1u) << 1) | \
            (__builtin_amdgcn_ubfe(__float_as_uint(pos.z), us_, 1u) << 2)) ^ om;                                              \
    child = (parent << 3) | cidx;                                                                                            \
    at_leaves = scale == a.min_scale;                                                                                        \
    deep = has_deep && child >= a.cache_codes;                                                                               \
    word = s_occ[(at_leaves || deep) ? 0u : (child >> 5)];                                                                   \
  } while (0)
  if (guard < max_trips) SE_TRIP_PREFETCH();
  for (; guard < max_trips && scale < 23; ++guard) {
    const f3 t_corner = f3_sub(f3_mul(pos, t_coef), t_bias);
    const float tc_max = fminf(fminf(t_corner.x, t_corner.y), t_corner.z);
    const uint32_t us = (uint32_t)scale;
    const uint32_t ox = __float_as_uint(pos.x), oy = __float_as_uint(pos.y), oz = __float_as_uint(pos.z);
    // descend (ray_iterator.hpp:172-199) / advance_ray (ray_iterator.hpp:116-167) candidates
    const float half = scale_exp2 * 0.5f;
    const f3 t_center = f3_add(f3_scale(half, t_coef), t_corner);
    const f3 dpos = {pos.x + ((t_center.x > t_min) ? half : 0.f), pos.y + ((t_center.y > t_min) ? half : 0.f), pos.z + ((t_center.z > t_min) ? half : 0.f)};
    const f3 apos = {pos.x - ((t_corner.x <= tc_max) ? scale_exp2 : 0.f), pos.y - ((t_corner.y <= tc_max) ? scale_exp2 : 0.f),
                     pos.z - ((t_corner.z <= tc_max) ? scale_exp2 : 0.f)};
    const uint32_t differing_bits = (ox ^ __float_as_uint(apos.x)) | (oy ^ __float_as_uint(apos.y)) | (oz ^ __float_as_uint(apos.z));
    // occupancy test: leaf level -> the sibling byte fetched when this parent was entered;
    // staged levels -> LDS; levels in between (volumes > 512^3) -> global word, cached per word
    asm volatile("" : "+v"(word));  // keep the LDS load an LDS load, and its first use here
    if (has_deep) {
      if (deep && !at_leaves) {
        const uint32_t w = child >> 5;
        if (w != gw_index) { gw_index = w; gw_word = m.occ[w]; }
        word = gw_word;
      }
    }
    word = at_leaves ? leaf_byte : word;
    const uint32_t shift = at_leaves ? cidx : (child & 31u);
    const bool exists = (word >> shift) & 1u;
    if (at_leaves && exists) break;  // leaf found: t_min is its entry distance
    const bool desc = exists && t_min <= t_max;
    if (desc && tc_max < h) { s_par[(22 - scale) * SE_WG_RAY + tid] = parent; s_tmax[(22 - scale) * SE_WG_RAY + tid] = t_max; }
    const bool pop = !desc && differing_bits > (1u << us);
    // select
    t_max = desc ? fminf(t_max, tc_max) : t_max;
    t_min = desc ? t_min : tc_max;
    h = desc ? tc_max : h;
    parent = desc ? child : parent;
    scale = desc ? scale - 1 : scale;
    scale_exp2 = desc ? half : scale_exp2;
    pos.x = desc ? dpos.x : apos.x; pos.y = desc ? dpos.y : apos.y; pos.z = desc ? dpos.z : apos.z;
    // (consumer first: the wait for gp_bits it contains must not cover a load that another lane issues in this same trip)
    if (desc && scale == a.min_scale) {   // entered a leaf parent: its sibling byte is first used next trip
      if ((child >> 3) != gp_code) {      // (only a parent restored from a never-pushed stack slot, see the stack note above)
        gp_code = child >> 3;
        gp_bits = *(const unsigned long long*)(occ_bytes + ((size_t)gp_code << 3));
      }
      leaf_byte = (uint32_t)(gp_bits >> (cidx << 3)) & 0xFFu;
    }
    if (desc && scale == a.min_scale + 1) { gp_code = child; gp_bits = *(const unsigned long long*)(occ_bytes + ((size_t)child << 3)); }
    if (pop) {
      // the highest differing bit is the scale of the first ancestor the ray is still inside
      scale = 31 - __clz(differing_bits);            // == (float_as_int((float)differing_bits) >> 23) - 127, differing_bits < 2^24
      scale_exp2 = __int_as_float((scale - 23 + 127) << 23);
      const int slot = 22 - scale;
      if (slot >= 0 && slot < a.stack_depth) { parent = s_par[slot * SE_WG_RAY + tid]; t_max = s_tmax[slot * SE_WG_RAY + tid]; }
      if (scale < 23) {
        const uint32_t keep = 0xFFFFFFFFu << scale;
        pos.x = __uint_as_float(__float_as_uint(pos.x) & keep);
        pos.y = __uint_as_float(__float_as_uint(pos.y) & keep);
        pos.z = __uint_as_float(__float_as_uint(pos.z) & keep);
      }
      h = 0.0f;
    }
    if (scale < 23) SE_TRIP_PREFETCH();
  }
#undef SE_TRIP_PREFETCH
  return {t_min * m.dim, tmax_m, guard};
}

// The same iterator without its stack (r04).  For a ray that is regular at set-up (it enters the volume before it
// leaves it, nothing is NaN) the reference's stack of (parent, t_max) and its `h` carry no information
// (tests/cpp/first_leaf_equiv.cpp is the CPU model of this function, checked ray by ray against the oracle's iterator):
//  * parent: heap codes are child = parent * 8 + idx, so the node a pop returns to is the current code shifted right
//    by three bits per level popped;
//  * t_max = min(far / dim, root exit, tc_max of every ancestor cell on the path).  t = pos * t_coef - t_bias is
//    monotone in pos, so every ancestor's tc_max is >= the tc_max of any cell inside it, and t_min only ever takes the
//    tc_max of cells inside the current ancestors: `t_min <= t_max` is `t_min <= min(far / dim, root exit)` -- unless the
//    ray descended from a cell whose own tc_max was already below t_min.  That does happen, once in ~10^6 rays: the child
//    slot is chosen by t_center = half * t_coef + t_corner, a different rounding of the same plane than the child's own
//    t_corner, so a ray within an ulp of a cell edge can be put into a child it has, by t_corner, already left.  Exactly
//    that event (descent with tc_max < t_min) sets `redo`, as does an irregular set-up, and the caller re-runs those
//    rays through se_first_leaf above (whole waves skip it: __any);
//  * h only decides whether a slot is rewritten; a slot that is read was written by the first descent below the same
//    parent (monotonicity again: a cell whose exit equals its parent's exit is left together with the parent).
// What a trip then needs from memory is the 8 sibling bits of the current parent (byte `parent` of the heap-ordered
// occupancy bits): one LDS byte per descent or pop instead of a word per trip; leaf parents take theirs from the 64
// leaf bits of their own parent, fetched one level earlier as before.
// t_start (r05, units of dim like every t here; 0 = none): the search enters the tree at max(t_min, t_start) instead of t_min.  The caller guarantees
// (se_beam_start) that no allocated block lies within centimetres of the ray before t_start, so the first leaf and the plane through which the ray
// enters it are the same -- and t_min at the leaf is that plane's time, `plane * t_coef - t_bias` of the cell left last, whatever cells came before.
// Should the search end without ever having advanced (t_min still t_start: it would be returning t_start as an entry time) the ray is handed back.
// (r06, measured and dropped: the search as a walk over the leaf grid alone -- one geometric descent to the leaf-size cell at t_min by the iterator's own
// t_center comparisons, then advance_ray at the leaf scale, eight cells computed ahead per round trip and their leaf-bitmap words fetched together (the cell
// sequence does not depend on what is loaded), `t_parent <= t_lim` standing in for the iterator's descent rule.  On the CPU model (fl::flat in
// tests/cpp/first_leaf_equiv.cpp) that is 6.9 steps of ~30 instructions instead of 10.4 trips of 70-110 per ray at 512^3, 15.6 against 13.9 at 1024^3.  On
// the device (profiles/r06m_flat_walk_ab.log): raycast beside the scan 35.1 -> 35.0 us at 512^3, 69.2 -> 65.3 us at 1024^3, 163 -> 192 us at 2048^3, the
// stress streams 6-12 % slower -- every wave pays the descent, a full batch and an L2 round trip where the hierarchical search reads LDS, and the last waves
// of a launch are bound by their march, not by their search.  It is also not exact as it stands: where two plane crossings fall within rounding of each
// other the iterator's placement by t_center and the walk's order by t_corner differ, about one ray in 10^6-10^7 enters its block through the other plane
// (8 ulp in t_min); that would need a near-tie test per cell on top.)
template <bool SHALLOW, typename Hook = SeNoHook>
__device__ __forceinline__ RaySpan se_first_leaf_lite(const DevMap& m, const RayArgs& a, f3 origin, f3 direction, const uint32_t* s_occ,
                                                      bool& redo, Hook before_loop = Hook(), bool live = true, float t_start = 0.f) {
  f3 pos = {1.0f, 1.0f, 1.0f};
  uint32_t parent = 1u;  // root
  float scale_exp2 = 0.5f;
  int scale = 22;
  const float eps = a.epsilon;
  f3 d;
  d.x = fabsf(direction.x) < eps ? copysignf(eps, direction.x) : direction.x;
  d.y = fabsf(direction.y) < eps ? copysignf(eps, direction.y) : direction.y;
  d.z = fabsf(direction.z) < eps ? copysignf(eps, direction.z) : direction.z;
  const f3 scaled_origin = {a.scaled_origin[0], a.scaled_origin[1], a.scaled_origin[2]};
  const f3 t_coef = f3_scale(-1.f, {1.f / fabsf(d.x), 1.f / fabsf(d.y), 1.f / fabsf(d.z)});
  f3 t_bias = f3_mul(t_coef, scaled_origin);
  uint32_t om = 0u;     // octant_mask ^ 7
  if (d.x > 0.0f) { om ^= 1u; t_bias.x = 3.0f * t_coef.x - t_bias.x; }
  if (d.y > 0.0f) { om ^= 2u; t_bias.y = 3.0f * t_coef.y - t_bias.y; }
  if (d.z > 0.0f) { om ^= 4u; t_bias.z = 3.0f * t_coef.z - t_bias.z; }
  float t_min = fmaxf(fmaxf(2.0f * t_coef.x - t_bias.x, 2.0f * t_coef.y - t_bias.y), 2.0f * t_coef.z - t_bias.z);
  const float h0 = fminf(fminf(t_coef.x - t_bias.x, t_coef.y - t_bias.y), t_coef.z - t_bias.z);
  t_min = fmaxf(t_min, a.near_n);
  const float t_lim = fminf(h0, a.far_n);   // t_max_init: all the t_max this loop knows
  const float tmax_m = t_lim * m.dim;
  redo = live && !(t_min < h0);             // irregular set-up (a ray that misses the volume, any NaN): full iterator
  const bool jumped = t_start > t_min;
  t_min = jumped ? t_start : t_min;
  if (1.5f * t_coef.x - t_bias.x > t_min) pos.x = 1.5f;
  if (1.5f * t_coef.y - t_bias.y > t_min) pos.y = 1.5f;
  if (1.5f * t_coef.z - t_bias.z > t_min) pos.z = 1.5f;
  const uint8_t* occ_bytes = (const uint8_t*)m.occ;
  const uint8_t* s_occ8 = (const uint8_t*)s_occ;
  const uint32_t staged_parents = a.cache_codes >> 3;   // parents below this code have their sibling byte in LDS
  unsigned long long gp_bits = 0ull;   // the 64 leaf bits below the node two levels above the leaves the ray is in
  uint32_t gp_code = 0u;
  before_loop();
  // sibling byte of parent P: staged -> LDS; leaf parent -> byte (P & 7) of gp_bits; else (volumes > 512^3) global
  // (r06, measured and dropped: fetching the 8 sibling bytes of a node's children as one 8-byte group when the node is entered, as the leaf bits are, so that
  // neither a descent into a level-5 / level-6 node nor a pop back to one waits for memory inside the trip: k_raycast 67.2 -> 69.5 us at 1024^3, 205 -> 212 us
  // at 2048^3 -- the load is needed one trip later, which a wave alone on its SIMD reaches long before the data; profiles/r06c_groups_ab.log)
  // (r06, measured: for the > 512^3 instantiations the compiler turns this select of two loads into ONE load of a selected address -- a FLAT load, which counts
  // on both memory counters, so every trip begins with s_waitcnt vmcnt(0) lgkmcnt(0).  Forcing two loads -- the LDS value pinned by an empty asm, or the global
  // arm as a wavefront-scope atomic load (SE_SIB_SPLIT) -- was slower, not faster: k_raycast 68.4 -> 70.0 us at 1024^3, the last waves' search phase 36 -> 40 us,
  // because the pinned LDS read is waited for at once instead of at the top of the next trip; profiles/r06f_flatfix_ab.log)
#ifndef SE_SIB_SPLIT
#define SE_SIB_SPLIT 0
#endif
  auto sib_of = [&](uint32_t P) -> uint32_t {
    if (SE_SIB_SPLIT && !SHALLOW && !(P < staged_parents)) return (uint32_t)__hip_atomic_load(occ_bytes + P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return (SHALLOW || P < staged_parents) ? (uint32_t)s_occ8[P] : (uint32_t)occ_bytes[P];
  };
#define SE_SIB_OF(P) sib_of(P)
  uint32_t sib = s_occ8[1];   // the root's children: word 0 of the occupancy bits is always staged
  int guard = 0;
  const int max_trips = (!live || redo) ? 0 : 4096;
  for (; guard < max_trips && scale < 23; ++guard) {
    const f3 t_corner = f3_sub(f3_mul(pos, t_coef), t_bias);
    const float tc_max = fminf(fminf(t_corner.x, t_corner.y), t_corner.z);
    const uint32_t us = (uint32_t)scale;
    const uint32_t ox = __float_as_uint(pos.x), oy = __float_as_uint(pos.y), oz = __float_as_uint(pos.z);
    const uint32_t cidx = (__builtin_amdgcn_ubfe(ox, us, 1u) | (__builtin_amdgcn_ubfe(oy, us, 1u) << 1) | (__builtin_amdgcn_ubfe(oz, us, 1u) << 2)) ^ om;
    const bool exists = (sib >> cidx) & 1u;
    if (exists && scale == a.min_scale) break;   // leaf found: t_min is its entry distance
    if (exists && t_min <= t_lim) {
      if (tc_max < t_min) { redo = true; break; }
      // descend (ray_iterator.hpp:172-199)
      const float half = scale_exp2 * 0.5f;
      const f3 t_center = f3_add(f3_scale(half, t_coef), t_corner);
      pos.x += (t_center.x > t_min) ? half : 0.f;
      pos.y += (t_center.y > t_min) ? half : 0.f;
      pos.z += (t_center.z > t_min) ? half : 0.f;
      parent = (parent << 3) | cidx;
      scale -= 1;
      scale_exp2 = half;
      if (scale == a.min_scale && !(parent < staged_parents)) {   // entered a leaf parent
        if ((parent >> 3) != gp_code) { gp_code = parent >> 3; gp_bits = *(const unsigned long long*)(occ_bytes + ((size_t)gp_code << 3)); }
        sib = (uint32_t)(gp_bits >> ((parent & 7u) << 3)) & 0xFFu;
      } else {
        if (scale == a.min_scale + 1) { gp_code = parent; gp_bits = *(const unsigned long long*)(occ_bytes + ((size_t)parent << 3)); }
        sib = SE_SIB_OF(parent);
      }
    } else {
      // advance_ray (ray_iterator.hpp:116-167)
      pos.x -= (t_corner.x <= tc_max) ? scale_exp2 : 0.f;
      pos.y -= (t_corner.y <= tc_max) ? scale_exp2 : 0.f;
      pos.z -= (t_corner.z <= tc_max) ? scale_exp2 : 0.f;
      t_min = tc_max;
      const uint32_t differing_bits = (ox ^ __float_as_uint(pos.x)) | (oy ^ __float_as_uint(pos.y)) | (oz ^ __float_as_uint(pos.z));
      if (differing_bits > (1u << us)) {
        // pop: the highest differing bit is the scale of the first ancestor the ray is still inside
        const int ns = 31 - __clz(differing_bits);
        parent >>= 3 * (ns - scale);
        scale = ns;
        scale_exp2 = __int_as_float((scale - 23 + 127) << 23);
        if (scale < 23) {
          const uint32_t keep = 0xFFFFFFFFu << scale;
          pos.x = __uint_as_float(__float_as_uint(pos.x) & keep);
          pos.y = __uint_as_float(__float_as_uint(pos.y) & keep);
          pos.z = __uint_as_float(__float_as_uint(pos.z) & keep);
          sib = SE_SIB_OF(parent);
        }
      }
    }
  }
#undef SE_SIB_OF
  if (jumped && t_min == t_start) redo = true;
  return {t_min * m.dim, tmax_m, guard};
}

// raycast(const Volume<T>&, origin, direction, tnear, tfar, mu, step, largestep)
// (se_denseslam/src/kfusion/rendering_impl.hpp:34-74, bfusion/rendering_impl.hpp:35-68): writes the hit
// (position, distance) or leaves it zero.  Shared by the raycast kernel and the volume renderer.
struct RayCounters { unsigned long long n_get, n_interp; unsigned n_batch; };

// ---- lean dense-grid march (r04) ---------------------------------------------------------------------------------
// The raycast is bound by VALU issue slots, and by r03's counters two thirds of the march's instructions
// were address arithmetic and range bookkeeping, not the reference's float operations.  For the dense grid:
//  * float -> int by the hardware conversion (one instruction) instead of the compare-and-select restatement of x86's
//    cvttss2si: the two agree for |f| < 2^31 and differ only for NaN (0 instead of INT_MIN) -- rays whose direction or
//    origin is not finite never get here (k_raycast: their result is "no hit" whatever the volume holds, because the
//    first interp returns NaN) and the host only selects these kernels for poses within 2^30 voxels of the volume;
//  * "inside the volume" is one compare of the OR of the three integers (size is a power of two, a negative or saturated
//    value has high bits set);
//  * a voxel's index is the sum of one term per axis, (block part << 10) | local part, so a sample costs ~4 integer
//    operations per axis and the 8 / 32 voxels of interp / grad one 3-input add each; maps of <= 4 GiB (O32: 512^3 dense)
//    carry the terms as byte offsets and load through a 32-bit offset from the scalar base, the brick's y plane with an
//    immediate offset from the same address (y = x + 512 floats, se_device.h);
//  * the eight corners of sample 0 are fetched with the batch only while the march is inside the band (previous value
//    < 1): in free or unobserved space they were ~100 instructions and 8 loads per batch that nobody used.  A sample
//    that turns out to need them without having them pays one extra round trip (se_interp), results unchanged;
//  * `(double)f_tt <= 0.1` is `f_tt < 0.1f` (0.1f is the smallest float above 0.1): no double-precision compare.
#ifndef SE_MARCH_SKIP
#define SE_MARCH_SKIP 4    // SDF march in unobserved space: positions asked of the leaf bitmap per round trip (se_march_skip); 0 = off.  Measured 0 / 4 / 8
                           // (profiles/r04p_march_skip_ab.log): 59.8 / 59.8 / 61.6 us per frame at 512^3, 193.8 / 188.4 / 189.5 at 1024^3, stress 69.7 / 68.6 / 71.3
#endif
#ifndef SE_POOLED_CELL
#define SE_POOLED_CELL 1   // pooled SDF march: the interpolation cell of sample 0 fetched with the batch when it lies inside the sample's brick; 0 = off (A/B)
#endif
#ifndef SE_MARCH_PROBE
#define SE_MARCH_PROBE 1   // dense maps > 512^3: leaf-bitmap probe in front of brick reads while the march is in unobserved space (se_cast_ray_sdf_lean)
#endif
template <bool O32> struct SeDense;
template <> struct SeDense<true> {     // byte-offset terms, 32 bit
  typedef uint32_t idx_t;
  static __device__ __forceinline__ idx_t tx(const DevMap&, uint32_t x) { return ((x >> 3) << 12) | ((x & 7u) << 2); }
  static __device__ __forceinline__ idx_t ty(const DevMap& m, uint32_t y) { return ((y >> 3) << (12 + m.leaf_level)) | ((y & 7u) << 5); }
  static __device__ __forceinline__ idx_t tz(const DevMap& m, uint32_t z) { return ((z >> 3) << (12 + 2 * m.leaf_level)) | ((z & 7u) << 8); }
  static __device__ __forceinline__ idx_t sum(idx_t a, idx_t b, idx_t c) { return a + b + c; }
  static __device__ __forceinline__ float ldx(const DevMap& m, idx_t i) { return *(const float*)((const char*)m.vx + (size_t)i); }
  static __device__ __forceinline__ float ldy(const DevMap& m, idx_t i) { return *(const float*)((const char*)m.vx + (size_t)i + 2048); }
  // SDF: the weight is a byte, 2048 + voxel bytes into the brick (se_device.h)
  static __device__ __forceinline__ float ldyb(const DevMap& m, idx_t i) { return (float)*((const uint8_t*)m.vx + (size_t)((i & 0xFFFFF000u) + 2048u + SE_YB((i >> 2) & 511u))); }
};
template <> struct SeDense<false> {    // (block sum, local sum) in 32 bit each, widened at the load
  struct idx_t { uint32_t blk, loc; };
  static __device__ __forceinline__ idx_t tx(const DevMap&, uint32_t x) { return {x >> 3, x & 7u}; }
  static __device__ __forceinline__ idx_t ty(const DevMap& m, uint32_t y) { return {(y >> 3) << m.leaf_level, (y & 7u) << 3}; }
  static __device__ __forceinline__ idx_t tz(const DevMap& m, uint32_t z) { return {(z >> 3) << (2 * m.leaf_level), (z & 7u) << 6}; }
  static __device__ __forceinline__ idx_t sum(idx_t a, idx_t b, idx_t c) { return {a.blk + b.blk + c.blk, a.loc + b.loc + c.loc}; }
  static __device__ __forceinline__ float ldx(const DevMap& m, idx_t i) { return m.vx[((size_t)i.blk << 10) | i.loc]; }
  static __device__ __forceinline__ float ldy(const DevMap& m, idx_t i) { return m.vx[(((size_t)i.blk << 10) | i.loc) + 512]; }
  static __device__ __forceinline__ float ldyb(const DevMap& m, idx_t i) { return (float)*((const uint8_t*)m.vx + (((size_t)i.blk << 12) + 2048u + SE_YB(i.loc))); }
};
template <bool O32> struct SeCell { float fx, fy, fz; typename SeDense<O32>::idx_t vi[8]; bool inside; };
// the interpolation cell of a point given in voxel units (Octree::interp, octree.hpp:541-563): fractions and the eight
// corner indices; inside = the whole cell lies in the volume (else the caller takes se_interp_generic)
template <bool O32>
__device__ __forceinline__ SeCell<O32> se_cell_lean(const DevMap& m, f3 pos) {
  typedef SeDense<O32> A;
  SeCell<O32> cell;
  const float flx = floorf(pos.x), fly = floorf(pos.y), flz = floorf(pos.z);
  const int bx = se_cvt_hw(flx), by = se_cvt_hw(fly), bz = se_cvt_hw(flz);
  cell.fx = pos.x - flx; cell.fy = pos.y - fly; cell.fz = pos.z - flz;
  const int lx = max(bx, 0), ly = max(by, 0), lz = max(bz, 0);
  const int top = m.size - 1;
  cell.inside = max(max(lx, ly), lz) < top;
  const typename A::idx_t X[2] = {A::tx(m, lx), A::tx(m, lx + 1)}, Y[2] = {A::ty(m, ly), A::ty(m, ly + 1)}, Z[2] = {A::tz(m, lz), A::tz(m, lz + 1)};
#pragma unroll
  for (int k = 0; k < 8; ++k) cell.vi[k] = A::sum(X[k & 1], Y[(k >> 1) & 1], Z[k >> 2]);
  return cell;
}
template <bool O32>
__device__ __forceinline__ float se_interp_lean(const DevMap& m, const FieldConst fc, f3 pos, BlkCache& c) {
  const SeCell<O32> cell = se_cell_lean<O32>(m, pos);
  if (!cell.inside) return se_interp_generic<true>(m, fc, pos, c);   // a corner outside the volume
  float p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = SeDense<O32>::ldx(m, cell.vi[k]);
  const float fx = cell.fx, fy = cell.fy, fz = cell.fz;
  return (((p[0] * (1 - fx) + p[1] * fx) * (1 - fy) + (p[2] * (1 - fx) + p[3] * fx) * fy) * (1 - fz) +
          ((p[4] * (1 - fx) + p[5] * fx) * (1 - fy) + (p[6] * (1 - fx) + p[7] * fx) * fy) * fz);
}
// Octree::grad on the dense grid (same terms and order as se_grad); falls back to the generic form when an index leaves the volume
template <bool O32>
__device__ __forceinline__ f3 se_grad_lean(const DevMap& m, const FieldConst fc, f3 pos, BlkCache& c) {
  typedef SeDense<O32> A;
  const float flx = floorf(pos.x), fly = floorf(pos.y), flz = floorf(pos.z);
  const int bx = se_cvt_hw(flx), by = se_cvt_hw(fly), bz = se_cvt_hw(flz);
  const int hi = m.size - 1;
  if (!(bx <= hi && by <= hi && bz <= hi)) return se_grad_generic<true>(m, fc, pos, c);
  const float fx = pos.x - flx, fy = pos.y - fly, fz = pos.z - flz;
  const typename A::idx_t X[4] = {A::tx(m, max(bx - 1, 0)), A::tx(m, max(bx, 0)), A::tx(m, min(bx + 1, hi)), A::tx(m, min(bx + 2, hi))};
  const typename A::idx_t Y[4] = {A::ty(m, max(by - 1, 0)), A::ty(m, max(by, 0)), A::ty(m, min(by + 1, hi)), A::ty(m, min(by + 2, hi))};
  const typename A::idx_t Z[4] = {A::tz(m, max(bz - 1, 0)), A::tz(m, max(bz, 0)), A::tz(m, min(bz + 1, hi)), A::tz(m, min(bz + 2, hi))};
  float V[4][4][4];
#pragma unroll
  for (int zi = 0; zi < 4; ++zi)
#pragma unroll
    for (int yi = 0; yi < 4; ++yi)
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        const int central = (xi == 1 || xi == 2) + (yi == 1 || yi == 2) + (zi == 1 || zi == 2);
        if (central < 2) continue;
        V[zi][yi][xi] = A::ldx(m, A::sum(X[xi], Y[yi], Z[zi]));
      }
  f3 g;
  g.x = (((V[1][1][2] - V[1][1][0]) * (1 - fx) + (V[1][1][3] - V[1][1][1]) * fx) * (1 - fy) +
         ((V[1][2][2] - V[1][2][0]) * (1 - fx) + (V[1][2][3] - V[1][2][1]) * fx) * fy) * (1 - fz) +
        (((V[2][1][2] - V[2][1][0]) * (1 - fx) + (V[2][1][3] - V[2][1][1]) * fx) * (1 - fy) +
         ((V[2][2][2] - V[2][2][0]) * (1 - fx) + (V[2][2][3] - V[2][2][1]) * fx) * fy) * fz;
  g.y = (((V[1][2][1] - V[1][0][1]) * (1 - fx) + (V[1][2][2] - V[1][0][2]) * fx) * (1 - fy) +
         ((V[1][3][1] - V[1][1][1]) * (1 - fx) + (V[1][3][2] - V[1][1][2]) * fx) * fy) * (1 - fz) +
        (((V[2][2][1] - V[2][0][1]) * (1 - fx) + (V[2][2][2] - V[2][0][2]) * fx) * (1 - fy) +
         ((V[2][3][1] - V[2][1][1]) * (1 - fx) + (V[2][3][2] - V[2][1][2]) * fx) * fy) * fz;
  g.z = (((V[2][1][1] - V[0][1][1]) * (1 - fx) + (V[2][1][2] - V[0][1][2]) * fx) * (1 - fy) +
         ((V[2][2][1] - V[0][2][1]) * (1 - fx) + (V[2][2][2] - V[0][2][2]) * fx) * fy) * (1 - fz) +
        (((V[3][1][1] - V[1][1][1]) * (1 - fx) + (V[3][1][2] - V[1][1][2]) * fx) * (1 - fy) +
         ((V[3][2][1] - V[1][2][1]) * (1 - fx) + (V[3][2][2] - V[1][2][2]) * fx) * fy) * fz;
  return g;
}
// Walking through unobserved space (r04).  While the last value had weight 0 the reference's march takes `largestep` after `largestep` until a get()
// returns a weight again; 43 % of all samples of a frame are such steps, most of them in blocks that were never allocated, and each pair of them
// was a dependent memory round trip.  The next SE_MARCH_SKIP positions -- formed by the same float additions the loop performs -- are asked of the
// leaf bitmap together (L2-resident; outside the volume and "no block here" both read as initValue(), weight 0), and the leading ones that have no
// block are consumed exactly as the loop would consume them (t < tfar tested before each, t += largestep after), without touching a brick or the index.
// The first position that has a block ends the run; the regular batch takes over there.  Results cannot change: an absent block's voxels ARE
// initValue() (dense: pre-filled bricks; pooled: no brick), also while the next frame's scan is inserting blocks beside this launch (a fresh brick
// holds initValue() too).  Returns true if the march is over (t reached tfar).
// (r05, measured and dropped: 8 / 12 positions per round trip once a run has outlasted the first four -- for the rays past a depth edge on their way to
// the far wall, ~27 blocks: fused launch 36.2 -> 44.9 / 48.1 us at 512^3, 69.9 -> 88.5 / 90.1 us at 1024^3, although the stand-alone raycast of the
// same frame is unchanged; the same out of line: 72 us and 32 B of scratch.  profiles/r05al_skip_long_ab.log, r05ap_skip_long2_ab.log)
template <bool STATS>
__device__ __forceinline__ bool se_march_skip(const DevMap& m, const RayArgs& a, f3 dir, float tfar, f3& position, float& t, RayCounters& rc) {
#if SE_MARCH_SKIP > 0
  const f3 sd = f3_scale(a.largestep, dir);
  f3 q[SE_MARCH_SKIP];
  uint32_t lin[SE_MARCH_SKIP], w[SE_MARCH_SKIP];
  bool in[SE_MARCH_SKIP];
  q[0] = position;
#pragma unroll
  for (int i = 1; i < SE_MARCH_SKIP; ++i) q[i] = f3_add(q[i - 1], sd);
#pragma unroll
  for (int i = 0; i < SE_MARCH_SKIP; ++i) {
    const int ix = se_cvt_hw(a.inv_voxel * q[i].x), iy = se_cvt_hw(a.inv_voxel * q[i].y), iz = se_cvt_hw(a.inv_voxel * q[i].z);
    in[i] = (uint32_t)(ix | iy | iz) < (uint32_t)m.size;
    lin[i] = in[i] ? block_linear(m, ix >> 3, iy >> 3, iz >> 3) : 0u;
  }
#pragma unroll
  for (int i = 0; i < SE_MARCH_SKIP; ++i) w[i] = m.lbits[lin[i] >> 5];
  bool run = true;
#pragma unroll
  for (int i = 0; i < SE_MARCH_SKIP; ++i) {
    const bool present = in[i] && ((w[i] >> (lin[i] & 31u)) & 1u);
    run = run && !present;
    if (run) {
      if (!(t < tfar)) return true;
      if (STATS) ++rc.n_get;
      position = f3_add(position, sd);
      t += a.largestep;
    }
  }
#endif
  return false;
}
// one get(): voxel coordinates, inside-the-volume flag and index of the sample at q (metres)
template <bool O32> struct SeSample { typename SeDense<O32>::idx_t vi; bool in; };
template <bool O32>
__device__ __forceinline__ SeSample<O32> se_sample_lean(const DevMap& m, const RayArgs& a, f3 q) {
  typedef SeDense<O32> A;
  const int ix = se_cvt_hw(a.inv_voxel * q.x), iy = se_cvt_hw(a.inv_voxel * q.y), iz = se_cvt_hw(a.inv_voxel * q.z);
  SeSample<O32> s;
  s.in = (uint32_t)(ix | iy | iz) < (uint32_t)m.size;
  const uint32_t ux = s.in ? (uint32_t)ix : 0u, uy = s.in ? (uint32_t)iy : 0u, uz = s.in ? (uint32_t)iz : 0u;   // (outside: voxel 0, value replaced)
  s.vi = A::sum(A::tx(m, ux), A::ty(m, uy), A::tz(m, uz));
  return s;
}
// raycast(const Volume<SDF>&, ...) (se_denseslam/src/kfusion/rendering_impl.hpp:34-74) on the dense grid; same float
// operations in the same order as se_cast_ray's generic form below (which stays the path of pooled bricks)
template <bool STATS, bool O32>
__device__ __forceinline__ void se_cast_ray_sdf_lean(const DevMap& m, const RayArgs& a, const FieldConst fc, f3 org, f3 dir, float tnear, float tfar,
                                                     BlkCache& c, float& hx, float& hy, float& hz, float& hw, RayCounters& rc) {
  typedef SeDense<O32> A;
  if (!(tnear < tfar)) return;
  float t = tnear;
  float stepsize = a.largestep;
  f3 position = f3_add(org, f3_scale_r(dir, t));
  float f_t = se_interp_lean<O32>(m, fc, f3_scale(a.inv_voxel, position), c);
  if (STATS) ++rc.n_interp;
  float f_tt = 0;
  if (!(f_t > 0)) return;
  float S = a.largestep;
  bool done = false;
  // (r05, measured and dropped: 4 samples per round trip instead of 2 from the 3rd / 6th batch of a ray on, outside the truncation band -- aimed at the
  // silhouette rays whose 12-23 round trips end the launch: fused launch 43.2 / 41.8 instead of 35.8 us at 512^3, 75 / 73 instead of 71 us at 1024^3,
  // profiles/r05w_deep_ab.log.  Nor does fetching the SECOND sample's eight corners with the batch while the march creeps along at `step` inside the band (a
  // grazing ray consumes both samples and wants both interpolated: two round trips per batch become one): 36.9 / 36.3 -> 36.7 / 36.8 us at 512^3, 69.0 / 68.5
  // -> 72.4 / 71.8 us at 1024^3, profiles/r05ae_prefetch1_ab.log.  Every earlier attempt to trade instructions for round trips in this loop lost too:
  // profiles/DESIGN_r01-r04.md 4.4)
  bool band = f_t < 1.f;   // the last value seen was inside the truncation band: the next sample probably wants its interpolated value
  bool unobs = false;      // the last consumed sample had weight 0 (unobserved space)
  for (int guard = 0; t < tfar && !done && guard < 65536; ++guard) {
    ++rc.n_batch;
    if (SE_MARCH_SKIP > 0 && unobs) {   // (S == stepsize == largestep in this state)
      if (se_march_skip<STATS>(m, a, dir, tfar, position, t, rc)) break;
      if (!(t < tfar)) break;
    }
    const f3 q0 = position;
    // (r06, measured and dropped: inside the truncation band of a surface met at a shallow angle the step shrinks by a constant factor from sample to sample --
    // a ray at 10 degrees to a wall takes ~16 samples from the band's edge to the surface, and those rays end the launch -- so the speculative second sample
    // was placed at S * (S / S_prev), within a voxel or two of the next position, as a prefetch of its lines: +-0 at 512^3 / 1024^3 / 2048^3, with or without
    // the next z slice of its voxel, profiles/r06e_extrapolate_ab.log.  Likewise a prefetch of the brick lines around the depth image's own estimate of the
    // hit point, issued when the march starts: +-0, profiles/r06d_prefetch_ab.log.  A batch of those rays is two dependent round trips and ~400 instructions
    // of a wave alone on its SIMD whatever the cache state.)
    // (r06, measured and dropped: inside the band a batch of ONE sample taken from its interpolation cell -- the voxel get() reads is the cell's corner 0, and
    // the speculative second sample is rarely consumed where the step is the shrinking distance to the surface -- ~60 instructions and three loads fewer per
    // in-band batch: fused launch 36.5 -> 39.0 us at 512^3, 17.3 k -> 16.6 k frames/s, stress stream -3.5 %: where the march creeps at the minimum step the
    // second sample IS consumed, and a batch of one doubles the loop's fixed cost there; profiles/r06j_band_solo_ab.log)
    const f3 q1 = f3_add(q0, f3_scale(S, dir));
    SeSample<O32> s0 = se_sample_lean<O32>(m, a, q0), s1 = se_sample_lean<O32>(m, a, q1);
    float x0, y0, x1, y1;
    bool probed = false;
    if constexpr (SE_MARCH_PROBE && !O32) {
     if (unobs) {
      probed = true;
      // Volumes > 512^3: the march is latency-bound on cold brick lines, and 43 % of its samples lie in blocks that were
      // never allocated -- whose bricks the dense grid backs with real memory nobody else touches.  While the march walks through
      // unobserved space (the last value had weight 0) it asks the leaf bitmap first (L2-resident) and reads a brick only where a
      // block exists; an absent block reads as initValue(), which is what its brick holds.
      const uint32_t w0 = m.lbits[s0.vi.blk >> 5], w1 = m.lbits[s1.vi.blk >> 5];
      s0.in = s0.in && ((w0 >> (s0.vi.blk & 31u)) & 1u);
      s1.in = s1.in && ((w1 >> (s1.vi.blk & 31u)) & 1u);
      x0 = fc.init_x; y0 = fc.init_y; x1 = fc.init_x; y1 = fc.init_y;
      if (s0.in) { x0 = A::ldx(m, s0.vi); y0 = A::ldyb(m, s0.vi); }
      if (s1.in) { x1 = A::ldx(m, s1.vi); y1 = A::ldyb(m, s1.vi); }
     }
    }
    if (!probed) { x0 = A::ldx(m, s0.vi); y0 = A::ldyb(m, s0.vi); x1 = A::ldx(m, s1.vi); y1 = A::ldyb(m, s1.vi); }
    SeCell<O32> cell0;
    float cv0[8];
    bool have0 = false;
    if (band) {
      cell0 = se_cell_lean<O32>(m, f3_scale(a.inv_voxel, q0));
      have0 = cell0.inside;
      if (have0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) cv0[k] = A::ldx(m, cell0.vi[k]);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!(t < tfar)) { done = true; break; }
      if (STATS) ++rc.n_get;
      const bool ok = i ? s1.in : s0.in;
      const float dx = ok ? (i ? x1 : x0) : fc.init_x, dy = ok ? (i ? y1 : y0) : fc.init_y;
      unobs = dy == 0;
      if (dy == 0) {
        stepsize = a.largestep;
        position = f3_add(position, f3_scale(stepsize, dir));
        band = false;
      } else {
        f_tt = dx;
        if (f_tt < 0.1f && f_tt >= -0.5f) {   // (double)f_tt <= 0.1
          if (i == 0 && have0) {
            const float fx = cell0.fx, fy = cell0.fy, fz = cell0.fz;
            f_tt = (((cv0[0] * (1 - fx) + cv0[1] * fx) * (1 - fy) + (cv0[2] * (1 - fx) + cv0[3] * fx) * fy) * (1 - fz) +
                    ((cv0[4] * (1 - fx) + cv0[5] * fx) * (1 - fy) + (cv0[6] * (1 - fx) + cv0[7] * fx) * fy) * fz);
          } else {
            f_tt = se_interp_lean<O32>(m, fc, f3_scale(a.inv_voxel, position), c);
          }
          if (STATS) ++rc.n_interp;
        }
        if (f_tt < 0) { done = true; break; }
        stepsize = fmaxf(f_tt * a.mu, a.step);
        position = f3_add(position, f3_scale(stepsize, dir));
        f_t = f_tt;
        band = f_tt < 1.f;
      }
      t += stepsize;
      if (stepsize != S) { S = stepsize; break; }
    }
  }
  if (f_tt < 0) {
    t = t + stepsize * f_tt / (f_t - f_tt);
    const f3 r = f3_add(org, f3_scale_r(dir, t));
    hx = r.x; hy = r.y; hz = r.z; hw = t;
  }
}
// ---- the same for pooled bricks (the north-star layout: bump-allocated bricks behind the index).  A sample's block entry comes from
// tab[]; while the march is in unobserved space it asks the leaf bitmap first, so that a block that was never allocated costs a bit test
// in an L2-resident array instead of a line of the 8 / 64 MB leaf index.  The entry of the previous sample's block is kept (most steps
// stay in or next to it).  interp / grad of a hit go through the generic forms (eight / 2x2x2 block look-ups), as before.
struct SePSample { uint32_t e, loc; int bx, by, bz; };
struct SePCache { uint32_t lin, e; };
template <bool PROBE>
__device__ __forceinline__ SePSample se_sample_pooled(const DevMap& m, const RayArgs& a, f3 q, SePCache& c, bool unobs) {
  const int ix = se_cvt_hw(a.inv_voxel * q.x), iy = se_cvt_hw(a.inv_voxel * q.y), iz = se_cvt_hw(a.inv_voxel * q.z);
  const bool in = (uint32_t)(ix | iy | iz) < (uint32_t)m.size;
  SePSample s;
  s.bx = ix >> 3; s.by = iy >> 3; s.bz = iz >> 3;
  s.loc = ((uint32_t)ix & 7u) | (((uint32_t)iy & 7u) << 3) | (((uint32_t)iz & 7u) << 6);
  s.e = 0u;
  if (in) {
    const uint32_t lin = block_linear(m, s.bx, s.by, s.bz);
    if (lin == c.lin) s.e = c.e;
    else {
      bool present = true;
      if (PROBE && unobs) present = (m.lbits[lin >> 5] >> (lin & 31u)) & 1u;
      uint32_t e = 0u;
      if (present) { e = m.tab[m.leaf_off + lin]; e = e == SE_PENDING ? 0u : e; }   // (PENDING: see se_block_entry)
      c.lin = lin; c.e = e;
      s.e = e;
    }
  }
  return s;
}
__device__ __forceinline__ size_t se_pooled_index(const SePSample& s) { return ((size_t)(s.e ? s.e - 1u : 0u) << 10) | s.loc; }
template <bool STATS>
__device__ __forceinline__ void se_cast_ray_sdf_pooled(const DevMap& m, const RayArgs& a, const FieldConst fc, f3 org, f3 dir, float tnear, float tfar,
                                                       BlkCache& c, float& hx, float& hy, float& hz, float& hw, RayCounters& rc) {
  if (!(tnear < tfar)) return;
  float t = tnear;
  float stepsize = a.largestep;
  f3 position = f3_add(org, f3_scale_r(dir, t));
  float f_t = se_interp_generic<false>(m, fc, f3_scale(a.inv_voxel, position), c);
  if (STATS) ++rc.n_interp;
  float f_tt = 0;
  if (!(f_t > 0)) return;
  float S = a.largestep;
  bool done = false, unobs = false, band = f_t < 1.f;
  SePCache pc = {0xFFFFFFFFu, 0u};
  for (int guard = 0; t < tfar && !done && guard < 65536; ++guard) {
    ++rc.n_batch;
    if (SE_MARCH_SKIP > 0 && unobs) {
      if (se_march_skip<STATS>(m, a, dir, tfar, position, t, rc)) break;
      if (!(t < tfar)) break;
    }
    const f3 q0 = position;
    const f3 q1 = f3_add(q0, f3_scale(S, dir));
    const SePSample s0 = se_sample_pooled<true>(m, a, q0, pc, unobs), s1 = se_sample_pooled<true>(m, a, q1, pc, unobs);
    const size_t i0 = se_pooled_index(s0), i1 = se_pooled_index(s1);
    // (SDF weights are bytes, 2048 + voxel bytes into the brick: se_device.h)
    const uint8_t* yb = (const uint8_t*)m.vx;
    const float x0 = m.vx[i0], y0 = (float)yb[((i0 >> 10) << 12) + 2048u + SE_YB((uint32_t)i0 & 511u)], x1 = m.vx[i1], y1 = (float)yb[((i1 >> 10) << 12) + 2048u + SE_YB((uint32_t)i1 & 511u)];
    // (r06) inside the truncation band the interpolation cell of sample 0 rides with the batch, as on the dense grid -- when the cell lies inside sample 0's
    // own brick (no corner on another block: two cells in three), its eight addresses follow from the entry the sample has just looked up.  Same values,
    // same blend as se_interp_generic; one round trip instead of two for the sample that decides the step
    bool have0 = false;
    float cv0[8], cfx = 0.f, cfy = 0.f, cfz = 0.f;
    if (SE_POOLED_CELL && band && s0.e) {
      const f3 pv = f3_scale(a.inv_voxel, q0);
      const float flx = floorf(pv.x), fly = floorf(pv.y), flz = floorf(pv.z);
      const int lx = max(cvt_i32(flx), 0), ly = max(cvt_i32(fly), 0), lz = max(cvt_i32(flz), 0);
      if ((lx & 7) != 7 && (ly & 7) != 7 && (lz & 7) != 7 && (lx >> 3) == s0.bx && (ly >> 3) == s0.by && (lz >> 3) == s0.bz) {
        have0 = true;
        cfx = pv.x - flx; cfy = pv.y - fly; cfz = pv.z - flz;
        const size_t base = ((size_t)(s0.e - 1u) << 10) + (size_t)((lx & 7) + ((ly & 7) << 3) + ((lz & 7) << 6));
#pragma unroll
        for (int k = 0; k < 8; ++k) cv0[k] = m.vx[base + (size_t)((k & 1) + ((k >> 1) & 1) * 8 + (k >> 2) * 64)];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!(t < tfar)) { done = true; break; }
      if (STATS) ++rc.n_get;
      const SePSample& sm = i ? s1 : s0;
      const float dx = sm.e ? (i ? x1 : x0) : fc.init_x, dy = sm.e ? (i ? y1 : y0) : fc.init_y;
      unobs = dy == 0;
      if (dy == 0) {
        stepsize = a.largestep;
        position = f3_add(position, f3_scale(stepsize, dir));
        band = false;
      } else {
        f_tt = dx;
        if (f_tt < 0.1f && f_tt >= -0.5f) {   // (double)f_tt <= 0.1
          if (i == 0 && have0) {
            f_tt = (((cv0[0] * (1 - cfx) + cv0[1] * cfx) * (1 - cfy) + (cv0[2] * (1 - cfx) + cv0[3] * cfx) * cfy) * (1 - cfz) +
                    ((cv0[4] * (1 - cfx) + cv0[5] * cfx) * (1 - cfy) + (cv0[6] * (1 - cfx) + cv0[7] * cfx) * cfy) * cfz);
          } else {
            c.bx = sm.bx; c.by = sm.by; c.bz = sm.bz; c.e = sm.e;      // the block of this sample as the look-up hint
            f_tt = se_interp_generic<false>(m, fc, f3_scale(a.inv_voxel, position), c);
          }
          if (STATS) ++rc.n_interp;
        }
        if (f_tt < 0) { done = true; break; }
        stepsize = fmaxf(f_tt * a.mu, a.step);
        position = f3_add(position, f3_scale(stepsize, dir));
        f_t = f_tt;
        band = f_tt < 1.f;
      }
      t += stepsize;
      if (stepsize != S) { S = stepsize; break; }
    }
  }
  if (f_tt < 0) {
    t = t + stepsize * f_tt / (f_t - f_tt);
    const f3 r = f3_add(org, f3_scale_r(dir, t));
    hx = r.x; hy = r.y; hz = r.z; hw = t;
  }
}
// ---- OFusion march: leaping over block-free space (r05) -----------------------------------------------------------------------------------
// raycast(Volume<OFusion>) steps one voxel at a time from the first leaf to the far plane (bfusion/rendering_impl.hpp:44-52), whatever it walks through: a
// ray that enters the blocks around a foreground object, misses the object and hits the wall two metres behind it takes 240 steps through space in which
// no block exists -- every one of them a get() that returns initValue() (y = 0: "no interp, f_tt unchanged, f_t = f_tt, t += step").  On the per-wave
// timeline (profiles/r05ah_wave_timelines.txt) these rays ARE the launch: 9 % of the waves run 25-32 batches of eight such samples, 45 us after the other
// 91 % are done.  A step through block-free space has no effect but the float addition t += step, so those additions are all that is done for it: check
// points 0.9 block edges apart on the ray are asked of the block grid dilated by one block (leap_bits: a clear bit = no block within one block of that
// cell, so every point within a block edge of a clear check point lies in a block that does not exist), sixteen per round trip, as long as they come back
// clear; the samples up to one step short of the last clear check point are consumed by K additions and f_t = f_tt.  Conservative: a bit set concurrently
// by the next frame's scan shortens the leap, and a block that appears behind the test holds initValue() in every voxel (the argument of
// se_march_skip).  Returns the number of samples consumed; `t` is advanced exactly as the loop would have advanced it.
#ifndef SE_OF_LEAP
#define SE_OF_LEAP 1
#endif
// 0 if the dilated block grid is clear at q (no block within one block of it), non-zero if not or if q lies outside the volume
__device__ __forceinline__ uint32_t se_leap_word(const RayArgs& a, f3 q) {
  const int F = a.leap_level;
  const int cx = se_cvt_flr(q.x * a.beam_inv_cellf), cy = se_cvt_flr(q.y * a.beam_inv_cellf), cz = se_cvt_flr(q.z * a.beam_inv_cellf);
  const bool in = (uint32_t)(cx | cy | cz) < (1u << F);
  const uint32_t idx = in ? (((uint32_t)cz << (2 * F)) | ((uint32_t)cy << F) | (uint32_t)cx) : 0u;
  const uint32_t w = a.leap_bits[idx >> 5];        // (unconditional: issued with the batch's value loads)
  return in ? ((w >> (idx & 31u)) & 1u) : 1u;
}
__device__ __forceinline__ int se_of_leap(const RayArgs& a, f3 org, f3 dir, float tfar, float& t, float& hold) {
  constexpr int M = 12;
  const int F = a.leap_level;
  float t_clear = t;
  for (int it = 0; it < 8; ++it) {
    uint32_t idx[M], w[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const f3 q = f3_add(org, f3_scale_r(dir, t_clear + (float)j * a.leap_dt));
      const int cx = se_cvt_flr(q.x * a.beam_inv_cellf), cy = se_cvt_flr(q.y * a.beam_inv_cellf), cz = se_cvt_flr(q.z * a.beam_inv_cellf);
      const bool in = (uint32_t)(cx | cy | cz) < (1u << F);     // outside the volume counts as "not clear": the march ends at the volume's face anyway
      idx[j] = in ? (((uint32_t)cz << (2 * F)) | ((uint32_t)cy << F) | (uint32_t)cx) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int j = 0; j < M; ++j) w[j] = idx[j] != 0xFFFFFFFFu ? a.leap_bits[idx[j] >> 5] : 0xFFFFFFFFu;
    int n = 0;
    bool run = true;
#pragma unroll
    for (int j = 0; j < M; ++j) { run = run && !((w[j] >> (idx[j] & 31u)) & 1u); n += run ? 1 : 0; }
    if (n < M) hold = t_clear + (float)(n + 1) * a.leap_dt;   // one block behind the first check point that is not clear: no further test before the march has got there
    if (n == 0) break;
    t_clear += (float)(n - 1) * a.leap_dt;
    if (n < M || !(t_clear < tfar)) break;
  }
  // samples t, t + step, ... (accumulated in float as the loop does) up to one step short of t_clear: at most a few hundred additions
  const int K = (int)((t_clear - t) * a.inv_voxel) - 1;
  int i = 0;
  for (; i + 8 <= K; i += 8) { t += a.step; t += a.step; t += a.step; t += a.step; t += a.step; t += a.step; t += a.step; t += a.step; }
  for (; i < K; ++i) t += a.step;
  return K > 0 ? K : 0;
}
template <bool STATS>
__device__ __forceinline__ void se_cast_ray_of_pooled(const DevMap& m, const RayArgs& a, const FieldConst fc, f3 org, f3 dir, float tnear, float tfar,
                                                      BlkCache& c, float& hx, float& hy, float& hz, float& hw, RayCounters& rc) {
  if (!(tnear < tfar)) return;
  float t = tnear;
  const float stepsize = a.step;
  float f_t = se_interp_generic<false>(m, fc, f3_scale(a.inv_voxel, f3_add(org, f3_scale_r(dir, t))), c);
  if (STATS) ++rc.n_interp;
  float f_tt = 0;
  if (!(f_t <= 0.f)) return;
  bool done = false, quiet = false;   // quiet: no sample of the last batch was observed
  float leap_hold = 0.f;              // no leap test before t has reached this value (the check point the last test found blocked)
  SePCache pc = {0xFFFFFFFFu, 0u};
  for (int guard = 0; t < tfar && !done && guard < 65536; ++guard) {
    ++rc.n_batch;
    if (SE_OF_LEAP && quiet && a.leap_bits && t >= leap_hold) {
      if (se_of_leap(a, org, dir, tfar, t, leap_hold) > 0) { f_t = f_tt; if (!(t < tfar)) break; }
    }
    float tt[SE_SPEC_OF];
    f3 q[SE_SPEC_OF];
    SePSample sm[SE_SPEC_OF];
    float qx[SE_SPEC_OF], qy[SE_SPEC_OF];
    tt[0] = t;
#pragma unroll
    for (int i = 1; i < SE_SPEC_OF; ++i) tt[i] = tt[i - 1] + stepsize;
#pragma unroll
    for (int i = 0; i < SE_SPEC_OF; ++i) { q[i] = f3_add(org, f3_scale_r(dir, tt[i])); sm[i] = se_sample_pooled<false>(m, a, q[i], pc, false); }
    quiet = (SE_OF_LEAP && a.leap_bits ? se_leap_word(a, q[SE_SPEC_OF - 1]) : 1u) == 0u;   // (see se_cast_ray_of_lean)
#pragma unroll
    for (int i = 0; i < SE_SPEC_OF; ++i) { const size_t vi = se_pooled_index(sm[i]); qx[i] = m.vx[vi]; qy[i] = m.vx[vi + 512]; }
    bool stop = false;
#pragma unroll
    for (int i = 0; i < SE_SPEC_OF; ++i) {
      if (stop) continue;
      t = tt[i];
      if (!(t < tfar)) { done = true; stop = true; continue; }
      if (STATS) ++rc.n_get;
      const float dx = sm[i].e ? qx[i] : fc.init_x, dy = sm[i].e ? qy[i] : fc.init_y;
      if (dx > -100.f && dy > 0.f) {
        quiet = false;
        c.bx = sm[i].bx; c.by = sm[i].by; c.bz = sm[i].bz; c.e = sm[i].e;
        f_tt = se_interp_generic<false>(m, fc, f3_scale(a.inv_voxel, q[i]), c);
        if (STATS) ++rc.n_interp;
      }
      if (f_tt > 0.f) { done = true; stop = true; continue; }
      f_t = f_tt;
    }
    if (!stop) t = tt[SE_SPEC_OF - 1] + stepsize;
  }
  if (f_tt > 0.f) {
    t = t - stepsize * (f_tt - 0.f) / (f_tt - f_t);
    const f3 r = f3_add(org, f3_scale_r(dir, t));
    hx = r.x; hy = r.y; hz = r.z; hw = t;
  }
}

// raycast(const Volume<OFusion>&, ...) (se_denseslam/src/bfusion/rendering_impl.hpp:35-68) on the dense grid with the lean addressing
// above; same float operations in the same order as the generic form in se_cast_ray
// (r05, measured and dropped: once the values of a batch are here, fetching the interpolation corners of the samples that want them two / four samples per
// round trip instead of one -- aimed at batches inside observed blocks, 3.6 us each on the per-wave timeline once the leap had removed the empty-space
// batches: 84 / 164 bytes of scratch per lane under the 96-register budget and interpolations behind a hit: fused launch 72.5 -> 107 / 115 us.
// profiles/r05ak_of_group_ab.log)
// (r06, the same idea the other way round, measured and dropped: once a batch has met an observed voxel, the following samples taken two / three at a time,
// each WITH the eight corners of its cell in the batch's own round trip, back to the wide batch when none of them is observed -- 12 bytes of scratch, identical
// images, and the fused launch 69.6 -> 128 / 111 us at 512^3, the stress stream 89 -> 180 / 146 us: the lanes of a wave are in the two modes at different
// times, every trip of the loop then runs both bodies, and a short batch advances its lanes a quarter as far.  With the switch made per wave (short batches
// while >= 16 / 32 / 40 lanes are inside observed space) 75.9 - 78.7 us: still behind.  profiles/r06q_of_observed_batch_ab.log)
template <bool STATS, bool O32>
__device__ __forceinline__ void se_cast_ray_of_lean(const DevMap& m, const RayArgs& a, const FieldConst fc, f3 org, f3 dir, float tnear, float tfar,
                                                    BlkCache& c, float& hx, float& hy, float& hz, float& hw, RayCounters& rc) {
  typedef SeDense<O32> A;
  if (!(tnear < tfar)) return;
  float t = tnear;
  const float stepsize = a.step;
  float f_t = se_interp_lean<O32>(m, fc, f3_scale(a.inv_voxel, f3_add(org, f3_scale_r(dir, t))), c);
  if (STATS) ++rc.n_interp;
  float f_tt = 0;
  if (!(f_t <= 0.f)) return;
  bool done = false, quiet = false;   // quiet: no sample of the last batch was observed
  float leap_hold = 0.f;              // no leap test before t has reached this value (the check point the last test found blocked)
  for (int guard = 0; t < tfar && !done && guard < 65536; ++guard) {
    ++rc.n_batch;
    if (SE_OF_LEAP && quiet && a.leap_bits && t >= leap_hold) {
      if (se_of_leap(a, org, dir, tfar, t, leap_hold) > 0) { f_t = f_tt; if (!(t < tfar)) break; }
    }
    float tt[SE_SPEC_OF];
    f3 q[SE_SPEC_OF];
    SeSample<O32> sm[SE_SPEC_OF];
    float qx[SE_SPEC_OF], qy[SE_SPEC_OF];
    tt[0] = t;
#pragma unroll
    for (int i = 1; i < SE_SPEC_OF; ++i) tt[i] = tt[i - 1] + stepsize;
#pragma unroll
    for (int i = 0; i < SE_SPEC_OF; ++i) { q[i] = f3_add(org, f3_scale_r(dir, tt[i])); sm[i] = se_sample_lean<O32>(m, a, q[i]); }
    // (the leap is only tried from block-free space: the batch's last sample asks the dilated grid along with the values -- a quiet batch inside
    // allocated, unobserved blocks would otherwise pay a test that cannot succeed, batch after batch: r05an)
    const uint32_t lw = SE_OF_LEAP && a.leap_bits ? se_leap_word(a, q[SE_SPEC_OF - 1]) : 1u;
#pragma unroll
    for (int i = 0; i < SE_SPEC_OF; ++i) { qx[i] = A::ldx(m, sm[i].vi); qy[i] = A::ldy(m, sm[i].vi); }
    quiet = lw == 0u;
    bool stop = false;
#pragma unroll
    for (int i = 0; i < SE_SPEC_OF; ++i) {
      if (stop) continue;
      t = tt[i];
      if (!(t < tfar)) { done = true; stop = true; continue; }
      if (STATS) ++rc.n_get;
      const float dx = sm[i].in ? qx[i] : fc.init_x, dy = sm[i].in ? qy[i] : fc.init_y;
      if (dx > -100.f && dy > 0.f) {
        quiet = false;
        f_tt = se_interp_lean<O32>(m, fc, f3_scale(a.inv_voxel, q[i]), c);
        if (STATS) ++rc.n_interp;
      }
      if (f_tt > 0.f) { done = true; stop = true; continue; }
      f_t = f_tt;
    }
    if (!stop) t = tt[SE_SPEC_OF - 1] + stepsize;
  }
  if (f_tt > 0.f) {
    t = t - stepsize * (f_tt - 0.f) / (f_tt - f_t);
    const f3 r = f3_add(org, f3_scale_r(dir, t));
    hx = r.x; hy = r.y; hz = r.z; hw = t;
  }
}

template <bool OFUSION, bool STATS, bool DENSE, bool O32 = false>
__device__ __forceinline__ void se_cast_ray(const DevMap& m, const RayArgs& a, const FieldConst fc, f3 org, f3 dir, float t_min, float tfar,
                                            BlkCache& c, float& hx, float& hy, float& hz, float& hw, RayCounters& rc) {
  if (!OFUSION && DENSE) se_cast_ray_sdf_lean<STATS, O32>(m, a, fc, org, dir, t_min, tfar, c, hx, hy, hz, hw, rc);
  else if (OFUSION && DENSE) se_cast_ray_of_lean<STATS, O32>(m, a, fc, org, dir, t_min, tfar, c, hx, hy, hz, hw, rc);
  else if (!OFUSION) se_cast_ray_sdf_pooled<STATS>(m, a, fc, org, dir, t_min, tfar, c, hx, hy, hz, hw, rc);
  else se_cast_ray_of_pooled<STATS>(m, a, fc, org, dir, t_min, tfar, c, hx, hy, hz, hw, rc);
}

// ---- beam start (r05) ------------------------------------------------------------------------------------------------------------
// Every ray used to begin its first-leaf search at the near plane and walk ~20 octree cells of empty space to the surface (19.7 trips per ray at
// 512^3, ~45 % of the kernel's vector instructions).  The 64 rays of a wave form a thin beam (an 8x8 pixel tile: 5 cm across at 4 m), so the wave
// first asks, cooperatively, how far the WHOLE beam is clear: lane i tests the point at t_i = near + i dt on the tile's centre ray against cbits,
// the coarse bitmap in which every cell within one cell of an allocated block is set (se_device.h).  A clear bit at p means: no block within
// `cell` (infinity norm) of p.  Every point of every ray of the tile with t in [t_i - dt/2, t_i + dt/2] lies within
//     (t_i + dt/2) (rmax + eps) + dt/2
// of p -- rmax = the largest distance between a ray's unit direction and the centre ray's (over the wave's 64 lanes, +5 %), eps the iterator's
// clamp of near-zero direction components (ray_iterator.hpp:63-75: the traversed line differs from the true one by at most eps t) -- and the
// sample counts as clear only if that bound is below 0.9 cell.  t_safe = the end of the clear run from the near plane; the rays enter the tree there
// (se_first_leaf_lite).  CPU model of both stages against the reference iterator, ray by ray: tests/cpp/first_leaf_equiv.cpp (17.7 -> 13.5 -> 10.4 trips
// per ray on the benchmark stream, 62 -> 24 -> 17 for OFusion on the stress stream).  Conservative by construction: centimetres of margin against rounding, a bit set concurrently by the next frame's scan
// only shortens the run, NaN anywhere fails the bound -> no jump.
__device__ __forceinline__ float se_beam_start(const DevMap& m, const RayArgs& a, f3 org, f3 dir, float tile_cx, float tile_cy) {
  const f3 dc = f3_normalized(m3_mul(a.view3, {tile_cx, tile_cy, 1.f}));
  float dev = sqrtf(f3_sqnorm(f3_sub(dir, dc)));
  for (int o = 32; o > 0; o >>= 1) dev = fmaxf(dev, __shfl_xor(dev, o));
  const float rad = dev * 1.05f + a.epsilon;
  const int lane = threadIdx.x & 63;
  const float ti = a.nearp + (float)lane * a.beam_dt;
  const f3 p = f3_add(org, f3_scale_r(dc, ti));
  const int C = m.clevel;
  const int cx = se_cvt_flr(p.x * a.beam_inv_cell), cy = se_cvt_flr(p.y * a.beam_inv_cell), cz = se_cvt_flr(p.z * a.beam_inv_cell);
  // A sample outside the volume sees no blocks -- unless it lies in the one-cell shell around it, within a cell of whatever is allocated on that face:
  // the shell takes the dilated bit of the boundary cell it touches (which covers every cell within one of the sample's own)
#ifndef SE_BEAM_SHELL
#define SE_BEAM_SHELL 1    // 0: samples outside the volume count as clear (the r05 form; A/B)
#endif
  const int nC = 1 << C;
  const bool in = SE_BEAM_SHELL ? ((uint32_t)(cx + 1) <= (uint32_t)nC && (uint32_t)(cy + 1) <= (uint32_t)nC && (uint32_t)(cz + 1) <= (uint32_t)nC) : ((uint32_t)(cx | cy | cz) < (uint32_t)nC);
  const uint32_t idx = in ? (((uint32_t)min(max(cz, 0), nC - 1) << (2 * C)) | ((uint32_t)min(max(cy, 0), nC - 1) << C) | (uint32_t)min(max(cx, 0), nC - 1)) : 0u;
  const uint32_t w = m.cbits[idx >> 5];
  const bool occupied = in && ((w >> (idx & 31u)) & 1u);
  const bool clear = !occupied && ((ti + 0.5f * a.beam_dt) * rad + 0.5f * a.beam_dt <= 0.9f * a.beam_cell);
  const unsigned long long blocked = __ballot(!clear);
  const int j = blocked ? (int)__builtin_ctzll(blocked) : 64;
  float t_safe = j < 1 ? 0.f : a.nearp + ((float)j - 0.5f) * a.beam_dt;
  if (a.beam >= 2) {
    // second stage, from the end of the coarse run on: the same test against fbits, the level-min(leaf, 6) grid dilated by one of its cells -- the coarse
    // stage stops 15-45 cm in front of the first block near the beam (one coarse cell of dilation, one of quantisation, half a sample), this one 7-15 cm
    const float t1 = fmaxf(t_safe, a.nearp);
    const float tf = t1 + ((float)lane + 0.5f) * a.beam_dt2;
    const f3 pf = f3_add(org, f3_scale_r(dc, tf));
    const int F = se_flevel(m);
    const int fx = se_cvt_flr(pf.x * a.beam_inv_cellf), fy = se_cvt_flr(pf.y * a.beam_inv_cellf), fz = se_cvt_flr(pf.z * a.beam_inv_cellf);
    const int nF = 1 << F;
    const bool inf = SE_BEAM_SHELL ? ((uint32_t)(fx + 1) <= (uint32_t)nF && (uint32_t)(fy + 1) <= (uint32_t)nF && (uint32_t)(fz + 1) <= (uint32_t)nF) : ((uint32_t)(fx | fy | fz) < (uint32_t)nF);   // (the shell: as above)
    const uint32_t fidx = inf ? (((uint32_t)min(max(fz, 0), nF - 1) << (2 * F)) | ((uint32_t)min(max(fy, 0), nF - 1) << F) | (uint32_t)min(max(fx, 0), nF - 1)) : 0u;
    const uint32_t fw = m.fbits[fidx >> 5];
    const bool clear2 = !(inf && ((fw >> (fidx & 31u)) & 1u)) && ((tf + 0.5f * a.beam_dt2) * rad + 0.5f * a.beam_dt2 <= 0.9f * a.beam_cellf);
    const unsigned long long blocked2 = __ballot(!clear2);
    const int j2 = blocked2 ? (int)__builtin_ctzll(blocked2) : 64;
    if (j2 > 0) t_safe = t1 + (float)j2 * a.beam_dt2;
  }
  // (never beyond the far plane: the iterator descends into a node only while t_min <= far / dim, but returns the leaves of a node it is already in
  // whatever their distance -- up to a node's width behind the far plane; a search started there would never descend and miss them)
  return fminf(t_safe, a.farp) * a.inv_dim;
}

// One thread per pixel; a wave covers an 8x8 pixel tile so that its rays stay in neighbouring
// blocks.  Output: packed float3 vertex / normal images (se::Image<Eigen::Vector3f>).
// (r05, fused launch with a 6th / 7th wave slot per SIMD so that scan waves start beside the raycast's first round instead of behind it: 73 / 70 VGPRs without
// scratch for the SDF instantiation, +-0 at 512^3 on both streams; the OFusion instantiation spills: 89 -> 114 / 160 us.  profiles/r05ag_occ_ab.log)
// Register budget of the dense-grid instantiations: 5 waves per SIMD (<= 96 VGPRs), stated instead of hoped for -- 640x480 is 5 120 waves, exactly five per
// SIMD, and an instantiation that lands on 97 registers (the OFusion march did, depending on an unrelated unroll factor) runs a second round of workgroups.
// The pooled instantiations keep what the compiler gives them (135 VGPRs for OFusion: forcing 96 means scratch in the march).
#ifndef SE_RAY_WAVES_PER_EU
#define SE_RAY_WAVES_PER_EU 5   // (experiments: 6 / 7 / 8, profiles/r05ag_occ_ab.log, profiles/DESIGN_r01-r04.md)
#endif
#define SE_RAY_OCC __attribute__((amdgpu_waves_per_eu(DENSE ? SE_RAY_WAVES_PER_EU : 1)))
// The kernel's body (k_raycast_scan runs it for the first workgroups of a fused launch): `bid` = workgroup index within the raycast's grid.
template <bool OFUSION, bool STATS, bool DENSE, bool SHALLOW, bool O32>   // O32: the voxel planes span <= 4 GiB (dense 512^3): 32-bit byte offsets
__device__ __forceinline__ void se_raycast_wg(const DevMap& m, const RayArgs& a, float* __restrict__ vertex, float* __restrict__ normal, uint32_t* smem, const int bid) {
  // LDS: [occupancy words of levels 1..cache_levels][ray stack: parent codes][ray stack: t_max]
  uint32_t* s_occ = smem;
  uint32_t* s_par = smem + a.cache_words;
  float* s_tmax = (float*)(s_par + a.stack_depth * SE_WG_RAY);
  if (a.gate && bid == 0 && threadIdx.x == 0) *(volatile uint32_t*)a.gate = a.gate_seq;
#ifdef SE_WAVE_PROBE
  const unsigned long long w_t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long w_t1 = w_t0, w_t2 = w_t0;
  int dbg_trips = 0, dbg_batches = 0;
#endif
  const unsigned long long tk0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
  // LDS staging of the occupancy words, split in two: the global loads are issued here, the LDS writes and the barrier
  // follow the ray set-up inside se_first_leaf (a per-level copy of only the used words was slower)
  constexpr int kStage = 2048 / SE_WG_RAY;      // occupancy levels <= 5 are 2048 words
  uint32_t st[kStage];
#pragma unroll
  for (int j = 0; j < kStage; ++j) { const int i = threadIdx.x + j * SE_WG_RAY; st[j] = i < a.cache_words ? m.occ[i] : 0u; }
  unsigned long long tk1 = tk0;
  unsigned long long tk2 = tk1, tk3 = tk1;
  const FieldConst fc = se_field_const(m);
  const int lane = threadIdx.x & 63;
  // Workgroup -> tile pair: the snake deal of the cost-sorted pairs over the compute units (see RayArgs::ray_order)
  const int tiles_x = (a.W + SE_TILE_W - 1) / SE_TILE_W;
  const int tiles_y = (a.row_end - a.row_begin + SE_TILE_H - 1) / SE_TILE_H;
  const int n_tiles = tiles_x * tiles_y, n_pairs = (n_tiles + 1) >> 1;
  int tile;
  {
    const int rnd = bid / a.n_cus, cu = bid - rnd * a.n_cus;
    const int pos = rnd * a.n_cus + ((rnd & 1) ? a.n_cus - 1 - cu : cu);
    const int pair = pos < n_pairs ? (int)a.ray_order[pos] : n_pairs;   // (positions of the last, partial round beyond the list: no pair)
    // (r05, measured and dropped: XCD x = workgroup index mod 8 taking the x-th contiguous band of tile pairs, so that a brick is cached by one L2 instead
    // of up to eight -- without the cost-sorted deal the fused launch takes 41.0 instead of 36.2 us at 512^3, OFusion 100 instead of 88 us: the launch
    // ends with its slowest compute unit, balance beats locality.  profiles/r05v_xcd_ab.log)
    tile = 2 * pair + (threadIdx.x >> 6);
  }
  int tx = tile % tiles_x, ty = tile / tiles_x;
  if (tile >= n_tiles) { tx = tiles_x; ty = 1 << 20; tile = 0; }   // no tile: fails the tests below
  const int px = tx * SE_TILE_W + (lane % SE_TILE_W);
  const int py = a.row_begin + ty * SE_TILE_H + (lane / SE_TILE_W);
  const bool tile_in_image = tx < tiles_x && ty < tiles_y;
  const int tile_slot = __builtin_amdgcn_readfirstlane(tile_in_image ? ty * tiles_x + tx : 0);
  unsigned my_cost = 0u;
  if (a.tile_cost) {
    // (clamped like the histogram bins the thresholds come from; 256 = "no tile gets this priority")
    const int prev = min(__builtin_amdgcn_readfirstlane((int)a.tile_cost[tile_slot]), 255);
    if (prev >= a.prio_thr[2]) __builtin_amdgcn_s_setprio(3);
    else if (prev >= a.prio_thr[1]) __builtin_amdgcn_s_setprio(2);
    else if (prev >= a.prio_thr[0]) __builtin_amdgcn_s_setprio(1);
  }
  unsigned long long n_get = 0, n_interp = 0, n_grad = 0, n_hit = 0;
  const bool in_image = px < a.W && py < a.row_end;
  const f3 dir = f3_normalized(m3_mul(a.view3, {(float)px, (float)py, 1.f}));
  const f3 org = {a.org[0], a.org[1], a.org[2]};
  auto finish_staging = [&]() {
#pragma unroll
    for (int j = 0; j < kStage; ++j) { const int i = threadIdx.x + j * SE_WG_RAY; if (i < a.cache_words) s_occ[i] = st[j]; }
    for (int i = threadIdx.x + kStage * SE_WG_RAY; i < a.cache_words; i += SE_WG_RAY) s_occ[i] = m.occ[i];   // (SE_HIP_RAY_CACHE_LEVELS > 5)
    __syncthreads();
    if (STATS) tk1 = __builtin_amdgcn_s_memtime();
  };
  // every thread of the workgroup goes through the set-up and the barrier; only rays inside the image enter the loop
  bool redo = false;
  const float t_start = a.beam ? se_beam_start(m, a, org, dir, (float)(tx * SE_TILE_W) + 0.5f * (SE_TILE_W - 1), (float)(a.row_begin + ty * SE_TILE_H) + 0.5f * (SE_TILE_H - 1)) : 0.f;
  RaySpan span = se_first_leaf_lite<SHALLOW>(m, a, org, dir, s_occ, redo, finish_staging, in_image, t_start);
  if (__any(redo)) {   // (about one ray in 10^6, plus rays that miss the volume: the reference iterator with its stack)
    const RaySpan full = se_first_leaf<SHALLOW>(m, a, org, dir, s_occ, s_par, s_tmax, SeNoHook(), redo);
    if (redo) span = {full.tcmin, full.tmax, span.trips + full.trips};
  }
#ifdef SE_WAVE_PROBE
  w_t1 = __builtin_amdgcn_s_memrealtime();
#endif
  if (in_image) {
    const float t_min = span.tcmin, tfar = span.tmax;
    if (STATS) tk2 = __builtin_amdgcn_s_memtime();
    float hx = 0.f, hy = 0.f, hz = 0.f, hw = 0.f;
    BlkCache c = {-1, -1, -1, 0u};
    if (t_min > 0.f) {
      RayCounters rc = {0ull, 0ull, 0u};
      se_cast_ray<OFUSION, STATS, DENSE, O32>(m, a, fc, org, dir, t_min, tfar, c, hx, hy, hz, hw, rc);
      if (STATS) { n_get += rc.n_get; n_interp += rc.n_interp; }
      my_cost = (unsigned)SE_COST_BATCH * rc.n_batch;
    }
    my_cost += (unsigned)span.trips;
#ifdef SE_WAVE_PROBE
    dbg_trips = span.trips; dbg_batches = (int)((my_cost - (unsigned)span.trips) / (unsigned)SE_COST_BATCH);
#endif
    if (STATS) tk3 = __builtin_amdgcn_s_memtime();
    float* v = vertex + 3 * (size_t)(px + py * a.W);
    float* n = normal + 3 * (size_t)(px + py * a.W);
    if (hw > 0.f) {   // (hit.w() > 0.0)
      if (STATS) { ++n_hit; ++n_grad; }
      v[0] = hx; v[1] = hy; v[2] = hz;
      const f3 g = DENSE ? se_grad_lean<O32>(m, fc, f3_scale(a.inv_voxel, {hx, hy, hz}), c) : se_grad<DENSE>(m, fc, f3_scale(a.inv_voxel, {hx, hy, hz}), c);
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
#ifdef SE_WAVE_PROBE
  // lane 0 of every wave: clocks (100 MHz) at entry / after the first-leaf search / at exit, HW_ID + XCC_ID, the wave maxima of trips and march batches
  {
    for (int o = 32; o > 0; o >>= 1) { dbg_trips = max(dbg_trips, __shfl_xor(dbg_trips, o)); dbg_batches = max(dbg_batches, __shfl_xor(dbg_batches, o)); }
    w_t2 = __builtin_amdgcn_s_memrealtime();
    if (a.wlog && (threadIdx.x & 63) == 0) {
      const size_t wi = (size_t)bid * (SE_WG_RAY / 64) + (threadIdx.x >> 6);
      if (wi < 32768) {
        a.wlog[4 * wi] = w_t0;
        a.wlog[4 * wi + 1] = w_t2;
        a.wlog[4 * wi + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg(63492);
        a.wlog[4 * wi + 3] = ((unsigned long long)(uint32_t)min((unsigned long long)(w_t1 - w_t0), 0xFFFFull) << 48) | ((unsigned long long)(uint32_t)(tile_slot & 0xFFFF) << 32) |
                             ((unsigned long long)(dbg_trips & 0x3FF) << 20) | ((unsigned long long)(dbg_batches & 0x3FF) << 10);
      }
    }
  }
#endif
  if (a.tile_cost) {
    // wave maximum through the (now idle) first stack slot of this wave's lane 0; LDS operations of one wave are ordered
    uint32_t* slot = s_par + (threadIdx.x & ~63);
    if (lane == 0) *slot = 0u;
    atomicMax(slot, my_cost);
    if (lane == 0 && tile_in_image) a.tile_cost[tile_slot] = (unsigned short)min(*slot >> a.cost_shift, 65535u);
  }
  if (STATS) {
    const unsigned long long tk4 = __builtin_amdgcn_s_memtime();
    se_stat_add<true>(m, S_GETS, n_get);
    se_stat_add<true>(m, S_INTERPS, n_interp);
    se_stat_add<true>(m, S_GRADS, n_grad);
    se_stat_add<true>(m, S_HITS, n_hit);
    // per-wave phase clocks (shader cycles): LDS staging, first-leaf search, march, gradient + store
    if (lane == 0) {
      atomicAdd(&m.stats[S_T_STAGE], tk1 - tk0);
      atomicAdd(&m.stats[S_T_ITER], tk2 - tk1);
      atomicAdd(&m.stats[S_T_MARCH], tk3 - tk2);
      atomicAdd(&m.stats[S_T_GRAD], tk4 - tk3);
      atomicMax(&m.stats[S_T_WAVEMAX], tk4 - tk0);
      atomicMax(&m.stats[13], tk2 - tk1);
      atomicMax(&m.stats[14], tk3 - tk2);
      atomicMax(&m.stats[15], tk4 - tk3);
    }
  }
}

template <bool OFUSION, bool STATS, bool DENSE, bool SHALLOW, bool O32 = false>
__global__ __launch_bounds__(SE_WG_RAY) SE_RAY_OCC void k_raycast(DevMap m, RayArgs a, float* __restrict__ vertex, float* __restrict__ normal) {
  extern __shared__ uint32_t smem[];
  se_raycast_wg<OFUSION, STATS, DENSE, SHALLOW, O32>(m, a, vertex, normal, smem, (int)blockIdx.x);
}
// r04: raycast of frame f and allocation scan of frame f+1 in ONE launch on ONE queue.  The two always ran side by side (the scan on a second
// queue behind a host gate), and the price of the second queue was the wait in front of the next sweep: 5.3 us of a 67 us frame for an event that
// has long fired when the queue reaches it.  Here the first `ray_wgs` workgroups are the raycast's, the rest the scan's: they share the
// chip exactly as before -- the raycast's workgroups are dispatched first, the scan's fill in as those retire -- and sweep(f) -> [raycast(f), scan(f+1)]
// -> sweep(f+1) are consecutive launches of one queue with nothing between them.  The scan defers its occupancy bits (the raycast reads occ[]), the
// sweep behind publishes them, as in the two-queue schedule.  Needs both frames' inputs at launch time: se_hip_frame defers a frame's raycast to the
// next call (any other API call flushes it first), so only a caller that streams frames without looking at each result gets this path.
static_assert(SE_WG_RAY == SE_WG_SCAN, "the fused raycast + scan launch uses one workgroup size");
template <bool OFUSION, bool DENSE, bool SHALLOW, bool O32>
__global__ __launch_bounds__(SE_WG_RAY) SE_RAY_OCC void k_raycast_scan(DevMap m, RayArgs a, float* __restrict__ vertex, float* __restrict__ normal, int ray_wgs,
                                                                         DevMap ms, const float* __restrict__ depthmap, AllocArgs sa, DepthSrc ds) {
  extern __shared__ uint32_t smem[];
  // Dispatch order = blockIdx order: the raycast's workgroups first -- all of them resident at once (the host only fuses launches whose raycast fits the
  // chip in one round, frame_can_fuse) -- the scan's fill in as those retire.  (Letting scan workgroups in earlier delays raycast waves: profiles/r05p;
  // fusing launches of several raycast rounds, 1280x960, lost to the two-queue schedule: profiles/r04p.)
  const int is_scan = (int)blockIdx.x >= ray_wgs;
  const int bid = is_scan ? (int)blockIdx.x - ray_wgs : (int)blockIdx.x;
  if (!is_scan) {
    // the raycast's waves are the launch's critical path: they start at issue priority 1, the scan's (priority 0) take the slots they leave.  Same box,
    // two repetitions (profiles/r04x_fused_ray_prio_ab.log): +1.1 % frames/s at 512^3, +2 % OFusion, 0 on the stress stream, -1 % at 1024^3 (there the scan,
    // 2x as long, ends the launch when it is held back) -> volumes whose every level is staged (<= 512^3) only
    if (SE_FUSED_RAY_PRIO && SHALLOW) __builtin_amdgcn_s_setprio(1);
    se_raycast_wg<OFUSION, false, DENSE, SHALLOW, O32>(m, a, vertex, normal, smem, bid);
    return;
  }
  if (OFUSION) se_scan_ofusion_wg<false>(ms, depthmap, sa, bid, ds);
  else se_scan_sdf_wg<false, DENSE>(ms, depthmap, sa, smem, bid, ds);   // (its SE_SCAN_SLOTS * SE_WG_SCAN words fit the raycast's LDS allocation: checked by the host)
}

// map read-back: packs the bricks named by slots[] (already in the caller's order) contiguously
__global__ __launch_bounds__(SE_WG) void k_gather_bricks(const float* __restrict__ plane, const uint32_t* __restrict__ slots, size_t n, float* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)SE_WG + threadIdx.x; i < n * 512; i += (size_t)gridDim.x * SE_WG)
    out[i] = plane[(size_t)slots[i >> 9] * SE_BRICK_STRIDE + (i & 511)];
}
// ... the y plane, whatever its storage (se_ld_y)
__global__ __launch_bounds__(SE_WG) void k_gather_bricks_y(DevMap m, const uint32_t* __restrict__ slots, size_t n, float* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)SE_WG + threadIdx.x; i < n * 512; i += (size_t)gridDim.x * SE_WG)
    out[i] = se_ld_y(m, (size_t)slots[i >> 9] * SE_BRICK_STRIDE + (i & 511));
}
// se_hip_load_map: values of the octants of a map file -> device planes (the octants were inserted by k_alloc_commit before)
__global__ __launch_bounds__(SE_WG) void k_load_nodes(DevMap m, const unsigned long long* __restrict__ keys, const float* __restrict__ x, const float* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)SE_WG + threadIdx.x; i < n * 8; i += (size_t)gridDim.x * SE_WG) {
    const unsigned long long key = keys[i >> 3];
    const int level = (int)(key & 0x1FFull);
    uint32_t nid = 0u;   // key 0 = the root
    if (level != 0) {
      if (level >= m.leaf_level) continue;
      const unsigned long long code = key & ~0x1FFull;
      const int sh = m.max_level - level;
      const int px = (int)(se_compact21(code) >> sh), py = (int)(se_compact21(code >> 1) >> sh), pz = (int)(se_compact21(code >> 2) >> sh);
      if ((unsigned)px >= (1u << level) || (unsigned)py >= (1u << level) || (unsigned)pz >= (1u << level)) continue;   // (k_alloc_commit skipped it too)
      const uint32_t e = m.tab[tab_index(m, level, px, py, pz)];
      if (e == 0u || e == SE_PENDING) continue;
      nid = e - 1u;
    }
    m.nx[(size_t)nid * 8 + (i & 7)] = x[i];
    m.ny[(size_t)nid * 8 + (i & 7)] = y[i];
  }
}
__global__ __launch_bounds__(SE_WG) void k_load_blocks(DevMap m, const int* __restrict__ coords, const float* __restrict__ x, const float* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)SE_WG + threadIdx.x; i < n * 512; i += (size_t)gridDim.x * SE_WG) {
    const size_t b = i >> 9;
    const int bx = coords[3 * b] >> 3, by = coords[3 * b + 1] >> 3, bz = coords[3 * b + 2] >> 3;
    const uint32_t e = m.tab[leaf_index(m, bx, by, bz)];
    if (e == 0u || e == SE_PENDING) continue;
    m.vx[(size_t)(e - 1u) * SE_BRICK_STRIDE + (i & 511)] = x[i];
    se_st_y(m, (size_t)(e - 1u) * SE_BRICK_STRIDE + (i & 511), y[i]);
  }
}
// pool initialisation: every voxel / node value starts at voxel_traits<T>::initValue()
// both planes of interleaved bricks ([512 x][512 y] per brick): n = floats in total
__global__ void k_fill_bricks(float* __restrict__ p, float vx, float vy, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (i & 512) ? vy : vx;
}
__global__ void k_fill(float* __restrict__ p, float v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// mm2metersKernel (se_denseslam/src/preprocessing.cpp:161-188) on the device, for the consumers of float_depth_ that come before a frame's allocation
// scan (tracking, renderDepth) and for row-sharded replicas, whose scan sees only its own rows: materialises a host-resident input (DepthSrc)
__global__ void k_depth_from_host(DepthSrc ds, int ow, int oh) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x < ow && y < oh) se_depth_at(ds, nullptr, x, y, ow);
}

// ------------------------------------------------------------------------------------------
// renderVolumeKernel / renderDepthKernel / renderTrackKernel
// (se_denseslam/src/rendering.cpp:111-283), RGBW 8-bit images
// ------------------------------------------------------------------------------------------
// ray_iterator's tmin() / tmax() (ray_iterator.hpp:53-102, 232-240): entry / exit of the volume cube
// clamped to the near / far planes -- what renderVolumeKernel uses (it does not advance to a leaf)
__device__ __forceinline__ RaySpan se_ray_range(const DevMap& m, const RayArgs& a, f3 origin, f3 direction) {
  const float eps = a.epsilon;
  f3 d;
  d.x = fabsf(direction.x) < eps ? copysignf(eps, direction.x) : direction.x;
  d.y = fabsf(direction.y) < eps ? copysignf(eps, direction.y) : direction.y;
  d.z = fabsf(direction.z) < eps ? copysignf(eps, direction.z) : direction.z;
  const f3 scaled_origin = f3_add(f3_div(origin, m.dim), {1.f, 1.f, 1.f});
  const f3 t_coef = f3_scale(-1.f, {1.f / fabsf(d.x), 1.f / fabsf(d.y), 1.f / fabsf(d.z)});
  f3 t_bias = f3_mul(t_coef, scaled_origin);
  if (d.x > 0.0f) t_bias.x = 3.0f * t_coef.x - t_bias.x;
  if (d.y > 0.0f) t_bias.y = 3.0f * t_coef.y - t_bias.y;
  if (d.z > 0.0f) t_bias.z = 3.0f * t_coef.z - t_bias.z;
  float t_min = fmaxf(fmaxf(2.0f * t_coef.x - t_bias.x, 2.0f * t_coef.y - t_bias.y), 2.0f * t_coef.z - t_bias.z);
  float t_max = fminf(fminf(t_coef.x - t_bias.x, t_coef.y - t_bias.y), t_coef.z - t_bias.z);
  t_min = fmaxf(t_min, a.nearp / m.dim);
  t_max = fminf(t_max, a.farp / m.dim);
  return {t_min * m.dim, t_max * m.dim, 0};
}

struct ShadeArgs { float light[3], ambient[3]; int render; };

// renderVolumeKernel (rendering.cpp:215-283)
template <bool OFUSION, bool DENSE>
__global__ __launch_bounds__(SE_WG_RAY) void k_render_volume(DevMap m, RayArgs a, ShadeArgs sh, const float* __restrict__ vertex,
                                                         const float* __restrict__ normal, unsigned char* __restrict__ out) {
  const FieldConst fc = se_field_const(m);
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * (SE_WG_RAY / 64) + (threadIdx.x >> 6);
  const int tiles_x = (a.W + 7) >> 3;
  const int px = ((tile % tiles_x) << 3) + (lane & 7);
  const int py = ((tile / tiles_x) << 3) + (lane >> 3);
  if (px >= a.W || py >= a.H) return;
  const int pix = px + py * a.W;
  f3 test, surfNorm;
  if (sh.render) {
    const f3 dir = f3_normalized(m3_mul(a.view3, {(float)px, (float)py, 1.f}));
    const f3 org = {a.org[0], a.org[1], a.org[2]};
    const RaySpan sp = se_ray_range(m, a, org, dir);
    float hx = 0.f, hy = 0.f, hz = 0.f, hw = 0.f;
    BlkCache c = {-1, -1, -1, 0u};
    RayCounters rc = {0ull, 0ull, 0u};
    if (sp.tcmin > 0.f) se_cast_ray<OFUSION, false, DENSE>(m, a, fc, org, dir, sp.tcmin, sp.tmax, c, hx, hy, hz, hw, rc);
    if (hw > 0) {
      test = {hx, hy, hz};
      const f3 g = se_grad<DENSE>(m, fc, f3_scale(a.inv_voxel, test), c);
      surfNorm = f3_scale(a.grad_scale, g);
      if (!OFUSION) surfNorm = f3_scale(-1.f, surfNorm);
    } else {
      test = {0.f, 0.f, 0.f};
      surfNorm = {-2.f, 0.f, 0.f};
    }
  } else {
    test = {vertex[3 * pix], vertex[3 * pix + 1], vertex[3 * pix + 2]};
    surfNorm = {normal[3 * pix], normal[3 * pix + 1], normal[3 * pix + 2]};
  }
  unsigned char* o = out + 4 * (size_t)pix;
  if (surfNorm.x != -2.f && sqrtf(f3_sqnorm(surfNorm)) > 0) {
    const f3 diff = f3_normalized(f3_sub(test, {sh.light[0], sh.light[1], sh.light[2]}));
    const f3 sn = f3_normalized(surfNorm);
    const float dirv = fmaxf((sn.x * diff.x + sn.y * diff.y) + sn.z * diff.z, 0.f);
    float col[3] = {dirv + sh.ambient[0], dirv + sh.ambient[1], dirv + sh.ambient[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      col[i] = std_max(col[i], 0.f);     // se::math::clamp on arrays: max then min (math_utils.h:111-116)
      col[i] = std_min(col[i], 1.f);
      col[i] *= 255.f;
      o[i] = (unsigned char)col[i];
    }
    o[3] = 0;
  } else {
    o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0;
  }
}

// gs2rgb (se_denseslam/include/se/commons.h:105-163), double arithmetic as in the reference
__device__ __forceinline__ void se_gs2rgb(double h, unsigned char* rgbw) {
  double r = 0, g = 0, b = 0;
  const double v = 0.75, mm = 0.25, sv = 0.6667;
  h *= 6.0;
  const int sextant = (int)h;
  const double fract = h - sextant;
  const double vsf = v * sv * fract;
  const double mid1 = mm + vsf, mid2 = v - vsf;
  switch (sextant) {
    case 0: r = v; g = mid1; b = mm; break;
    case 1: r = mid2; g = v; b = mm; break;
    case 2: r = mm; g = v; b = mid1; break;
    case 3: r = mm; g = mid2; b = v; break;
    case 4: r = mid1; g = mm; b = v; break;
    case 5: r = v; g = mm; b = mid2; break;
    default: r = 0; g = 0; b = 0; break;
  }
  rgbw[0] = (unsigned char)(r * 255); rgbw[1] = (unsigned char)(g * 255); rgbw[2] = (unsigned char)(b * 255); rgbw[3] = 0;
}
// renderDepthKernel (rendering.cpp:111-152)
__global__ void k_render_depth(unsigned char* __restrict__ out, const float* __restrict__ depth, int n, float nearp, float farp) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n) return;
  const float rangeScale = 1 / (farp - nearp);
  unsigned char* o = out + 4 * (size_t)pos;
  const float d0 = depth[pos];
  if (d0 < nearp) { o[0] = 255; o[1] = 255; o[2] = 255; o[3] = 0; }
  else if (d0 > farp) { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; }
  else { const float d = (d0 - nearp) * rangeScale; se_gs2rgb(d, o); }
}
