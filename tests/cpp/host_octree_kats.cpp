// CPU check of the host se::Octree of include/se/octree.hpp (what DenseSLAMSystem::getMap() hands out), with the vectors of
// the reference's serialise tests (se_core/test/io/io_unittest.cpp:57-128: code 24 / side 256 / 8 x 5.f; a block at
// (40, 48, 52) -- here (40, 48, 56), a valid block position -- holding 512 x 5.f, resp. 512 x {5.f, 2.}) and the read interface of se_core/include/se/octree.hpp:340-478.
// Usage: host_octree_kats <out-dir>   -> writes sdf.bin, ofusion.bin, prints "ok".
#include <cstdio>
#include <cstdlib>
#include <string>
#include <se/octree.hpp>

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

static se::key_t morton(int x, int y, int z) {
  se::key_t k = 0;
  for (int i = 0; i < 21; ++i) k |= ((se::key_t)((x >> i) & 1) << (3 * i)) | ((se::key_t)((y >> i) & 1) << (3 * i + 1)) | ((se::key_t)((z >> i) & 1) << (3 * i + 2));
  return k;
}

template <typename T, typename V> static void run(const std::string& file, V voxel, V corner) {
  se::Octree<T> t;
  const int size = 512, leaf = 6;                       // 512 / 8 = 2^6 blocks per edge
  t.init(size, 5.f);
  const int bx = 40, by = 48, bz = 56;   // (the reference vector is (40, 48, 52); a block that is to be FOUND again must sit on a multiple of 8)
  // the ancestors of the block, root first: level l keeps the top l bits of the coordinates (octant_ops.hpp:49-53)
  for (int l = 0; l < leaf; ++l) {
    const int mask = ~((size >> l) - 1) & (size - 1);
    se::Node<T>* n = t.add_node(morton(bx & mask, by & mask, bz & mask) | (se::key_t)l, (unsigned)(size >> l));
    for (auto& v : n->value_) v = corner;
  }
  const int c[3] = {bx, by, bz};
  se::VoxelBlock<T>* b = t.add_block(morton(bx, by, bz) | (se::key_t)leaf, c, true);
  for (auto& v : b->voxel_block_) v = voxel;
  t.finalize();
  CHECK(t.size() == 512 && t.dim() == 5.f && t.root() != nullptr);
  CHECK(t.getNodesBuffer().size() == 6 && t.getBlockBuffer().size() == 1);
  // fetch / fetch_octant (octree.hpp:441-478)
  CHECK(t.fetch(41, 49, 57) == b && t.fetch(40, 48, 56) == b && t.fetch(47, 55, 63) == b);
  CHECK(t.fetch(48, 48, 56) == nullptr && t.fetch(0, 0, 0) == nullptr);
  CHECK(t.fetch_octant(41, 49, 57, leaf) == b);
  for (int l = 1; l < leaf; ++l) {
    se::Node<T>* n = t.fetch_octant(41, 49, 57, l);
    CHECK(n != nullptr && (int)(n->code_ & 0x1FF) == l && n->side_ == (unsigned)(size >> l));
    unsigned char bits = n->children_mask_;
    CHECK(bits != 0 && (bits & (bits - 1)) == 0);        // one child on the path
  }
  CHECK(t.fetch_octant(300, 300, 300, 1) == nullptr);
  // get: voxel inside the block, the coarse node's corner value where the path ends (octree.hpp:340-355); get_fine: initValue
  CHECK(t.get(41, 49, 57).x == voxel.x && t.get(41, 49, 57).y == voxel.y);
  CHECK(b->data(41, 49, 57).x == voxel.x && b->data(0).x == voxel.x && b->active());
  CHECK(t.get(300, 300, 300).x == corner.x && t.get(300, 300, 300).y == corner.y);
  CHECK(t.get(48, 48, 56).x == corner.x);
  CHECK(t.get_fine(300, 300, 300).x == voxel_traits<T>::initValue().x && t.get_fine(300, 300, 300).y == voxel_traits<T>::initValue().y);
  CHECK(t.get_fine(41, 49, 57).x == voxel.x);
  t.save(file);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  run<SDF>(dir + "/sdf.bin", SDF{5.f, 2.f}, SDF{3.f, 4.f});
  typedef voxel_traits<OFusion>::value_type OV;
  run<OFusion>(dir + "/ofusion.bin", OV{5.f, 2.}, OV{3.f, 4.});
  std::printf("ok\n");
  return 0;
}
