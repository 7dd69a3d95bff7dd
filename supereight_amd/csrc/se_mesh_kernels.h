// Map export, second half: marching cubes over the allocated blocks,
// se::algorithms::marching_cube (se_core/include/se/algorithms/meshing.hpp:161-208) as called by
// DenseSLAMSystem::dump_mesh (se_denseslam/src/DenseSLAMSystem.cpp:302-322): inside(v) = v.x < 0,
// select(v) = v.x.  One wave per block, lane = x + 8y, the 8 z-cells of a lane in turn; every cell reads
// its 8 corners through get_fine (a missing block reads initValue(), whose y == 0 ends the cell).
// Triangle table: include/se_mc_table.h (the standard published table, content-identical to the reference's edge_tables.h:66).
#pragma once
#include "../../include/se_mc_table.h"
#include "se_device.h"

__constant__ signed char SE_MC_TRI[256][SE_MC_WIDTH];

struct MeshArgs {
  float* out;                       // 9 floats per triangle, or null: count only
  unsigned long long* counter;      // [0] = triangles counted, [1] = triangles written
  unsigned long long capacity;      // triangles `out` can hold
};

// Octree::get_fine (octree.hpp:357-377)
__device__ __forceinline__ void se_get_fine(const DevMap& m, int x, int y, int z, float& vx, float& vy) {
  vx = m.init_x; vy = m.init_y;
  if (!in_volume(m, x, y, z)) return;
  uint32_t e = m.dense ? block_linear(m, x >> 3, y >> 3, z >> 3) + 1u : m.tab[leaf_index(m, x >> 3, y >> 3, z >> 3)];
  if (e == 0u || e == SE_PENDING) return;
  const size_t vi = (size_t)(e - 1u) * SE_BRICK_STRIDE + (size_t)((x & 7) + ((y & 7) << 3) + ((z & 7) << 6));
  vx = m.vx[vi]; vy = se_ld_y(m, vi);
}

// compute_intersection (meshing.hpp:45-55): s + (0.0 - v1) * (d - s) / (v2 - v1), coefficient-wise
__device__ __forceinline__ f3 se_mc_vertex(const DevMap& m, int x, int y, int z, int edge) {
  const int C[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1}, {0, 1, 0}, {1, 1, 0}, {1, 1, 1}, {0, 1, 1}};
  const int E[12][2] = {{0, 1}, {1, 2}, {2, 3}, {0, 3}, {4, 5}, {5, 6}, {6, 7}, {4, 7}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  const int a = E[edge][0], b = E[edge][1];
  const int sx = x + C[a][0], sy = y + C[a][1], sz = z + C[a][2];
  const int dx = x + C[b][0], dy = y + C[b][1], dz = z + C[b][2];
  const float voxelSize = m.dim / m.size;
  const f3 s = {sx * voxelSize, sy * voxelSize, sz * voxelSize};
  const f3 d = {dx * voxelSize, dy * voxelSize, dz * voxelSize};
  float v1, v2, w;
  se_get_fine(m, sx, sy, sz, v1, w);
  se_get_fine(m, dx, dy, dz, v2, w);
  const float k = (float)(0.0 - (double)v1);
  return {s.x + (k * (d.x - s.x)) / (v2 - v1), s.y + (k * (d.y - s.y)) / (v2 - v1), s.z + (k * (d.z - s.z)) / (v2 - v1)};
}
__device__ __forceinline__ bool se_mc_reject(f3 v, float dim) { return v.x <= 0 || v.y <= 0 || v.z <= 0 || v.x > dim || v.y > dim || v.z > dim; }

__global__ __launch_bounds__(SE_WG) void k_mesh(DevMap m, MeshArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * SE_WG + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * SE_WG) >> 6;
  const uint32_t nblocks = min(m.ctr[C_BLOCKS], m.cap_blocks);
  const int C[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1}, {0, 1, 0}, {1, 1, 0}, {1, 1, 1}, {0, 1, 1}};
  for (uint32_t b = wave; b < nblocks; b += nwaves) {
    const uint32_t bp = m.bpos[b];
    const int bx = (int)(bp & 1023u) << 3, by = (int)((bp >> 10) & 1023u) << 3, bz = (int)(bp >> 20) << 3;
    const int x = bx + (lane & 7), y = by + (lane >> 3);
    // top = (coordinates + 8).cwiseMin(size - 1): cells whose +1 corner would leave the volume are skipped
    if (x >= min(bx + 8, m.size - 1) || y >= min(by + 8, m.size - 1)) continue;
    for (int z = bz; z < min(bz + 8, m.size - 1); ++z) {
      // compute_index (meshing.hpp:121-155)
      float px[8], py[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) se_get_fine(m, x + C[i][0], y + C[i][1], z + C[i][2], px[i], py[i]);
      bool known = true;
      unsigned index = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) { known = known && !(py[i] == 0.f); index |= (px[i] < 0.f) ? (1u << i) : 0u; }
      if (!known) index = 0;
      const signed char* edges = SE_MC_TRI[index];
      for (unsigned e = 0; e < 16 && edges[e] != -1; e += 3) {
        const f3 v1 = se_mc_vertex(m, x, y, z, edges[e]);
        const f3 v2 = se_mc_vertex(m, x, y, z, edges[e + 1]);
        const f3 v3 = se_mc_vertex(m, x, y, z, edges[e + 2]);
        if (se_mc_reject(v1, m.dim) || se_mc_reject(v2, m.dim) || se_mc_reject(v3, m.dim)) continue;
        if (!a.out) { atomicAdd(&a.counter[0], 1ull); continue; }
        const unsigned long long slot = atomicAdd(&a.counter[1], 1ull);
        if (slot >= a.capacity) continue;
        float* o = a.out + 9 * slot;
        o[0] = v1.x; o[1] = v1.y; o[2] = v1.z; o[3] = v2.x; o[4] = v2.y; o[5] = v2.z; o[6] = v3.x; o[7] = v3.y; o[8] = v3.z;
      }
    }
  }
}
