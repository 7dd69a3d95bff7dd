// TEST PROGRAM for include/se/octree.hpp's interp / grad (r04): builds a 64^3 SDF map from a fixed recipe -- blocks at the listed
// coordinates, voxel (x, y, z) = ((x * 73856093u ^ y * 19349663u ^ z * 83492791u) & 0xFFFF) / 65536 - 0.5 -- and prints, for every
// position on stdin, the bit patterns of interp(pos, .x) and grad(pos).  tests/test_host_octree_cpp.py builds the same map in the
// oracle's float tree and compares bit for bit.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "se/octree.hpp"

struct V3 { float v[3]; float& operator()(int i) { return v[i]; } float operator()(int i) const { return v[i]; } };
static uint64_t morton(unsigned x, unsigned y, unsigned z) {
  uint64_t k = 0;
  for (int i = 0; i < 21; ++i) k |= ((uint64_t)((x >> i) & 1) << (3 * i)) | ((uint64_t)((y >> i) & 1) << (3 * i + 1)) | ((uint64_t)((z >> i) & 1) << (3 * i + 2));
  return k;
}
static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main(int argc, char** argv) {
  const int N = 64, L = 3;   // leaves at level 3 (8^3 blocks of 8^3 voxels)
  se::Octree<SDF> t;
  t.init(N, 1.5f);
  // every ancestor of every listed block, then the blocks, in key order per level (add order does not matter to finalize())
  int nb = 0; int bc[64][3];
  for (int i = 1; i + 2 < argc; i += 3) { bc[nb][0] = std::atoi(argv[i]); bc[nb][1] = std::atoi(argv[i + 1]); bc[nb][2] = std::atoi(argv[i + 2]); ++nb; }
  t.add_node(0, N);
  for (int level = 1; level < L; ++level) {
    const unsigned side = N >> level;
    for (int i = 0; i < nb; ++i) {
      const unsigned x = bc[i][0] & ~(side - 1), y = bc[i][1] & ~(side - 1), z = bc[i][2] & ~(side - 1);
      const uint64_t code = morton(x, y, z) | (uint64_t)level;
      bool seen = false;
      for (auto& n : t.getNodesBuffer()) if (n->code_ == code) seen = true;
      if (!seen) t.add_node(code, side);
    }
  }
  for (int i = 0; i < nb; ++i) {
    auto* b = t.add_block(morton(bc[i][0], bc[i][1], bc[i][2]) | (uint64_t)L, bc[i], true);
    for (int z = 0; z < 8; ++z) for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {
      const uint32_t X = bc[i][0] + x, Y = bc[i][1] + y, Z = bc[i][2] + z;
      const float v = (float)((X * 73856093u ^ Y * 19349663u ^ Z * 83492791u) & 0xFFFFu) / 65536.0f - 0.5f;
      b->voxel_block_[x + 8 * y + 64 * z] = {v, 1.f};
    }
  }
  t.finalize();
  float px, py, pz;
  while (std::scanf("%f %f %f", &px, &py, &pz) == 3) {
    const V3 p = {{px, py, pz}};
    const float f = t.interp(p, [](const SDF& v) { return v.x; });
    const V3 g = t.grad(p);
    std::printf("%08x %08x %08x %08x\n", bits(f), bits(g(0)), bits(g(1)), bits(g(2)));
  }
  return 0;
}
