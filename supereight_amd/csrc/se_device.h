// Device-side data layout and arithmetic helpers of the gfx950 dense-fusion path.
//
// HBM layout (one replica per GPU):
//   tab[]      dense direct-mapped index pyramid: for every octree level l in [1, leaf] one
//              (2^l)^3 uint32 grid, entry = 0 (absent) | id+1 | SE_PENDING (being created in the
//              running alloc kernel).  Replaces the pointer octree of
//              se_core/include/se/octree.hpp (fetch / fetch_octant / children walk).
//   occ[]      occupancy bit pyramid: one bit per possible octant of every level, Morton order
//              (the 8 children of an octant share one byte).  It is what the ray traversal walks;
//              its top levels (37 KB for 512^3) are staged in LDS by the raycast kernel.
//   vx[]       voxel bricks, one 4 KB slot per block: the 512 x values (floats), then the 512 y values (OFusion: floats; SDF: bytes, see below), voxel index x + 8y + 64z
//              (se_core/include/se/node.hpp:139-144).  SDF: x = tsdf, y = weight.
//              OFusion: x = log-odds, y = last-update time (the reference stores y as double;
//              every value it ever holds is a float, so float storage is lossless).
//              SDF weights (r06) are stored as BYTES: sdf_update leaves y = min(y + 1, maxweight = 100) (kfusion/mapping_impl.hpp:60,
//              DenseSLAMSystem.cpp:235), an integer in 0 .. 100 in every map this library or the reference produces -- one byte holds it without
//              loss, and the sweep, which is bound by HBM bandwidth from 1024^3 on, moves 5 bytes per voxel and direction instead of 8.  The y plane
//              of an SDF brick is 512 bytes at byte offset 2048 of the brick (the float plane's place; DevMap::ybyte, se_ld_y / se_st_y), ordered so
//              that the eight z slices of one (x, y) column are consecutive bytes -- byte SE_YB(v) = ((v & 63) << 3) | (v >> 6) for voxel v = x + 8y + 64z:
//              the sweep's lane (x, y) reads and writes its eight weights as ONE 8-byte access, the wave 512 consecutive bytes.  Every
//              export converts back to float.  A map file whose weights are not such integers is refused by se_hip_load_map.
//   bpos[]     compact list of allocated blocks: packed position in block units
//              (x | y<<10 | z<<20); bactive[] the VoxelBlock::active_ flag, indexed by voxel slot.
//   Voxel slot of a block: pooled mode = its list index (pool of max_blocks bricks behind tab[]);
//   dense mode (default while (N/8)^3 * 4 KB <= 64 GiB and a third of the free HBM, i.e. N <= 2048 on 288 GB) = its
//   linear grid index, every brick pre-initialised to initValue(): raycast addresses voxels
//   straight from coordinates, one dependent memory access shorter per sample.
//   nx[], ny[] Node::value_[8] of internal nodes, npos[]/nlevel[] their position and level.
//
// Arithmetic contract: IEEE-754 binary32, the operation order of the reference's source
// text, no FMA contraction (the library is built with -ffp-contract=off), correctly rounded
// division and sqrt (hipcc default), denormals preserved.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SE_PENDING 0xFFFFFFFFu
// Floats from one brick to the next: one array of 4 KB bricks [512 x | y plane at float 512 = byte 2048] -- a voxel's two values share a 4 KB page
// (one address translation per get() instead of two; dense maps scatter bricks over 8 / 64 GiB), see DESIGN.md 3.  The lean march paths
// (SeDense, se_pooled_index) spell the layout out as shifts: 4 096 bytes per brick, y plane 2 048 bytes behind x.
#define SE_BRICK_STRIDE 1024
static_assert(SE_BRICK_STRIDE == 1024, "the march paths of se_kernels.h address bricks as [512 x | 512 y] = 4 KB");
#define SE_MAX_LEVELS 12

enum { C_BLOCKS = 0, C_NODES = 1, C_OVERFLOW = 2, C_COUNT = 8 };
enum { S_PROBES = 0, S_NEWKEYS = 1, S_SWEPT = 2, S_NODES = 3, S_GETS = 4, S_INTERPS = 5, S_GRADS = 6, S_HITS = 7,
       S_T_ITER = 8, S_T_MARCH = 9, S_T_GRAD = 10, S_T_WAVEMAX = 11, S_T_STAGE = 12, S_COUNT = 16 };

struct DevMap {
  uint32_t* tab;
  uint32_t off[SE_MAX_LEVELS];
  uint32_t* occ;                  // occupancy bits in heap order: octant (level l, Morton index c) is bit (1 << 3l) | c
  uint32_t* lbits;                // one bit per cell of the block grid, in block_linear order: 'a block is allocated here' (the raycast's march asks it
                                  // before it touches a brick or the index: 32 KB at 512^3, 256 KB at 1024^3, 2 MB at 2048^3 -- L2-resident)
  uint32_t* cbits;                // beam start of the raycast: one bit per cell of the COARSE grid (level clevel, linear x + (y << clevel) + (z << 2 clevel)),
                                  // set for every cell within one cell (27-neighbourhood) of an allocated block: 4 KB at level 5.  A clear bit = no block anywhere
                                  // within one coarse cell of any point of this cell.  Only ever set (never cleared: blocks are never freed).  The second
                                  // half of the allocation holds the same grid undilated ("a block exists in this cell": se_mark_coarse).
  int clevel;
  uint32_t* fbits;                // the same on a finer grid, level se_flevel(m) = min(leaf_level, 6) (7.5 cm cells at 4.8 m: the margin an 8x8-pixel beam of a 640x480 camera
                                  // needs at working distance; r05 used the block grid itself, whose margin at 1024^3 -- 3.75 cm -- the beam does not fit): set for every cell
                                  // within one cell of an allocated block; 32 KB (second stage of the beam start, OFusion leap); null if that level is <= clevel
  int size, max_level, leaf_level;
  int defer_occ;                  // 1: insertions do not touch occ[] (a commit kernel sets the bits later)
  int defer_mark;                 // 1: insertions do not mark cbits / fbits (se_occ_commit does, from the key list, before the next raycast: every allocation scan)
  int dense;                      // 1: voxel slot of a block = its linear grid index (no look-up needed to address voxels)
  uint32_t leaf_off;              // = off[leaf_level]; kept separately so that hot kernels never index off[] dynamically
  float dim;
  float* vx;                      // the bricks: 1024 floats (4 KB) per slot, x plane first (layout note above)
  int ybyte;                      // 1: the y plane of a brick is 512 bytes (SDF weights), 0: 512 floats (OFusion) -- see the layout note above
  uint32_t* bpos;
  uint8_t* bactive;
  float* nx;
  float* ny;
  uint32_t* npos;
  uint8_t* nlevel;
  uint32_t* ctr;
  unsigned long long* stats;
  unsigned long long* newkeys;  // [0] = count, [1..] keys
  unsigned long long cap_keys;
  uint32_t cap_blocks, cap_nodes;
  float init_x, init_y, empty_x;
};

// byte of voxel v (< 512) within the 512-byte weight plane of an SDF brick
#define SE_YB(v) ((((v) & 63u) << 3) | ((v) >> 6))
// y of the voxel at float index vi = slot * SE_BRICK_STRIDE + voxel (voxel < 512) -- for the kernels that are not instantiated per field type
__device__ __forceinline__ float se_ld_y(const DevMap& m, size_t vi) {
  if (m.ybyte) return (float)((const uint8_t*)(m.vx + (vi & ~(size_t)(SE_BRICK_STRIDE - 1)) + 512))[SE_YB((uint32_t)vi & 511u)];
  return m.vx[vi + 512];
}
__device__ __forceinline__ void se_st_y(const DevMap& m, size_t vi, float y) {
  if (m.ybyte) ((uint8_t*)(m.vx + (vi & ~(size_t)(SE_BRICK_STRIDE - 1)) + 512))[SE_YB((uint32_t)vi & 511u)] = (uint8_t)(int)y;
  else m.vx[vi + 512] = y;
}

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 f3_add(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 f3_sub(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 f3_scale(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ f3 f3_scale_r(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 f3_div(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ f3 f3_mul(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ float f3_sqnorm(f3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
// Eigen's normalized(): v / sqrt(squaredNorm) when squaredNorm > 0
__device__ __forceinline__ f3 f3_normalized(f3 a) {
  const float z = f3_sqnorm(a);
  if (z > 0.f) return f3_div(a, sqrtf(z));
  return a;
}
// 3x3 (row-major r[9]) * vector, accumulation left to right
__device__ __forceinline__ f3 m3_mul(const float* r, f3 p) {
  f3 o;
  o.x = (r[0] * p.x + r[1] * p.y) + r[2] * p.z;
  o.y = (r[3] * p.x + r[4] * p.y) + r[5] * p.z;
  o.z = (r[6] * p.x + r[7] * p.y) + r[8] * p.z;
  return o;
}
// 3x4 (row-major a[12]) * homogeneous point
__device__ __forceinline__ f3 m34_mul_h(const float* a, f3 p) {
  f3 o;
  o.x = ((a[0] * p.x + a[1] * p.y) + a[2] * p.z) + a[3] * 1.f;
  o.y = ((a[4] * p.x + a[5] * p.y) + a[6] * p.z) + a[7] * 1.f;
  o.z = ((a[8] * p.x + a[9] * p.y) + a[10] * p.z) + a[11] * 1.f;
  return o;
}
// float -> int32 as x86 cvttss2si does it (what the reference's casts compile to)
__device__ __forceinline__ int cvt_i32(float f) {
  if (!(f > -2147483904.f && f < 2147483648.f)) return (int)0x80000000;
  return (int)f;
}
// float -> int32 by the hardware conversion (truncating, saturating, NaN -> 0): equals cvt_i32 for |f| < 2^31 and non-NaN f; one instruction.
// Used where the operand cannot be NaN: the stage calls refuse a pose / intrinsics with a NaN or an infinity (finite_pose, se_hip_api.hip), and with
// finite origin and direction every march position org + dir * t (t finite: tnear < tfar <= far) is finite.
__device__ __forceinline__ int se_cvt_hw(float f) { int r; asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f)); return r; }
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }  // std::min(a,b)
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }  // std::max(a,b)
__device__ __forceinline__ float clampf(float f, float a, float b) { return std_max(a, std_min(f, b)); }
__device__ __forceinline__ float sqf(float a) { return a * a; }

__device__ __forceinline__ unsigned long long se_expand21_d(unsigned long long v) {
  unsigned long long x = v & 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__device__ __forceinline__ uint32_t pack_pos(int x, int y, int z) { return (uint32_t)x | ((uint32_t)y << 10) | ((uint32_t)z << 20); }
__device__ __forceinline__ uint32_t tab_index(const DevMap& m, int l, int x, int y, int z) {
  return m.off[l] + (((((uint32_t)z << l) | (uint32_t)y) << l) | (uint32_t)x);
}
// (r06, measured and dropped: ordering the dense brick grid in tiles of 8 x 8 x 8 blocks -- 2 MiB of consecutive bricks per tile -- instead of rows of the
// whole grid, so that the bricks a surface patch touches share large pages and DRAM rows: the sweep at 2048^3 stayed at 745 us, the scan lost 10 % to the
// index conversion, 512^3 lost 2 % to the longer per-axis terms of the march; profiles/r06i_tiled_ab.log)
__device__ __forceinline__ uint32_t block_linear(const DevMap& m, int bx, int by, int bz) {
  const int l = m.leaf_level;
  return ((((uint32_t)bz << l) | (uint32_t)by) << l) | (uint32_t)bx;
}
// voxel slot of the block at list position `idx` with packed position `bp`
__device__ __forceinline__ uint32_t block_slot(const DevMap& m, uint32_t idx, uint32_t bp) {
  return m.dense ? block_linear(m, (int)(bp & 1023u), (int)((bp >> 10) & 1023u), (int)(bp >> 20)) : idx;
}
__device__ __forceinline__ uint32_t leaf_index(const DevMap& m, int bx, int by, int bz) {
  const int l = m.leaf_level;
  return m.leaf_off + (((((uint32_t)bz << l) | (uint32_t)by) << l) | (uint32_t)bx);
}
__device__ __forceinline__ uint32_t tab_index_packed(const DevMap& m, int l, uint32_t p) {
  return m.off[l] + (((((p >> 20) & 1023u) << l) | ((p >> 10) & 1023u)) << l | (p & 1023u));
}
// Morton code of an octant position (<= 10 bits per axis), x lowest
__device__ __forceinline__ uint32_t morton30(int x, int y, int z) {
  return (uint32_t)(se_expand21_d((unsigned long long)x) | (se_expand21_d((unsigned long long)y) << 1) | (se_expand21_d((unsigned long long)z) << 2));
}
// Occupancy bit of an octant = its heap code (1 << 3l) | morton(x, y, z): the root is code 1 and the
// children of code n are 8n .. 8n+7, so the eight sibling bits of a parent are byte n of the array and
// a ray traversal carries one integer per node with no per-level offsets (l is a per-lane value there;
// indexing by-value DevMap arrays with it would spill them to scratch).  Levels <= L occupy the first
// 2 * 8^L bits.
__host__ __device__ __forceinline__ size_t occ_words_upto(int l) { return l < 2 ? 1 : ((size_t)2 << (3 * l)) / 32; }
__device__ __forceinline__ uint32_t occ_code(int l, int x, int y, int z) { return (1u << (3 * l)) | morton30(x, y, z); }
__device__ __forceinline__ void occ_set(const DevMap& m, int l, int x, int y, int z) {
  if (m.defer_occ) return;
  const uint32_t code = occ_code(l, x, y, z);
  atomicOr(&m.occ[code >> 5], 1u << (code & 31u));
}
// level of the fbits grid (derived, not stored: the raycast kernels run at the scalar-register limit)
#define SE_FLEVEL_MAX 6
__host__ __device__ __forceinline__ int se_flevel(const DevMap& m) { return m.leaf_level < SE_FLEVEL_MAX ? m.leaf_level : SE_FLEVEL_MAX; }
__device__ __forceinline__ bool in_volume(const DevMap& m, int x, int y, int z) {
  return (unsigned)x < (unsigned)m.size && (unsigned)y < (unsigned)m.size && (unsigned)z < (unsigned)m.size;
}

// 21-bit-per-axis Morton spread (se_core/include/se/utils/morton_utils.hpp:37-45); only used when a
// key leaves the device-side index (new-key lists, downloads).
__host__ __device__ __forceinline__ unsigned long long se_expand21(unsigned long long v) {
  unsigned long long x = v & 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__host__ __device__ __forceinline__ unsigned long long se_compact21(unsigned long long v) {
  unsigned long long x = v & 0x1249249249249249ull;
  x = (x | x >> 2) & 0x10c30c30c30c30c3ull;
  x = (x | x >> 4) & 0x100f00f00f00f00full;
  x = (x | x >> 8) & 0x1f0000ff0000ffull;
  x = (x | x >> 16) & 0x1f00000000ffffull;
  x = (x | x >> 32) & 0x1fffffull;
  return x;
}
// key of the octant at `level` whose position in units of its own side is (x,y,z):
// morton(voxel coords) | level  (se_core/include/se/octant_ops.hpp:49-53)
__host__ __device__ __forceinline__ unsigned long long se_make_key(int x, int y, int z, int level, int max_level) {
  const int sh = max_level - level;
  const unsigned long long code = se_expand21((unsigned long long)x << sh) | (se_expand21((unsigned long long)y << sh) << 1) |
                                  (se_expand21((unsigned long long)z << sh) << 2);
  return code | (unsigned long long)level;
}
