/*
 * se::Octree<FieldType> -- host-side mirror of the reference's map container for what DenseSLAMSystem::getMap() hands to an
 * application (se_denseslam/include/se/DenseSLAMSystem.h:295; se_apps/src/benchmark.cpp:179-181 calls save() on it).
 *
 * The map of this build lives in HBM (index pyramid + SoA bricks); getMap() materialises it on the host
 * in the reference's own shape: a pointer octree of Node / VoxelBlock objects with the member layout of
 * se_core/include/se/node.hpp:45-137 (value_[8], code_, side_, children_mask_, child_ptr_[8]; coordinates_,
 * voxel_block_[512] x-fastest, active_), paged in two buffers in key order.  The read-only part of the reference's
 * interface is provided with the reference's semantics:
 *   size(), dim(), get(x,y,z), get_fine(x,y,z), fetch(x,y,z), fetch_octant(x,y,z,depth)   (octree.hpp:340-478)
 *   interp(pos, select), grad(pos), grad(pos, select)                                     (octree.hpp:541-563, 565-737; r04)
 *     -- what the reference's planning / collision users sample a map with.  `pos` is any 3-vector type with operator()(int)
 *     (Eigen::Vector3f in the reference; Eigen is not a dependency of this header), grad returns the same type.
 *   getBlockBuffer() / getNodesBuffer()-style access, save(filename)                        (octree.hpp:898-914)
 * Integration, allocation and ray casting stay on the device; this object is a snapshot.
 */
#ifndef SE_HIP_OCTREE_HPP
#define SE_HIP_OCTREE_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

/* field types: se_denseslam/include/se/volume_traits.hpp:41-72 */
struct SDF { float x; float y; };
struct OFusion {};
template <typename T> struct voxel_traits;
template <> struct voxel_traits<SDF> {
  typedef SDF value_type;
  static inline value_type empty() { return {1.f, -1.f}; }
  static inline value_type initValue() { return {1.f, 0.f}; }
};
template <> struct voxel_traits<OFusion> {
  typedef struct { float x; double y; } value_type;
  static inline value_type empty() { return {0.f, 0.}; }
  static inline value_type initValue() { return {0.f, 0.}; }
};

namespace se {
typedef uint64_t key_t;

template <typename T> class Node {
 public:
  typedef typename voxel_traits<T>::value_type value_type;
  value_type value_[8];
  key_t code_ = 0;
  unsigned int side_ = 0;
  unsigned char children_mask_ = 0;
  Node* child_ptr_[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  Node() { for (auto& v : value_) v = voxel_traits<T>::initValue(); }
  virtual ~Node() {}
  virtual bool isLeaf() { return false; }
  Node*& child(const int x, const int y, const int z) { return child_ptr_[x + y * 2 + z * 4]; }
  Node*& child(const int offset) { return child_ptr_[offset]; }
};

template <typename T> class VoxelBlock : public Node<T> {
 public:
  typedef typename voxel_traits<T>::value_type value_type;
  static constexpr unsigned int side = 8;
  static constexpr unsigned int sideSq = side * side;
  int coordinates_[3] = {0, 0, 0};
  value_type voxel_block_[side * sideSq];   /* x + 8 y + 64 z (node.hpp:139-144) */
  bool active_ = false;
  VoxelBlock() { for (auto& v : voxel_block_) v = voxel_traits<T>::initValue(); }
  bool isLeaf() { return true; }
  const int* coordinates() const { return coordinates_; }
  value_type data(int x, int y, int z) const { return voxel_block_[(x - coordinates_[0]) + (y - coordinates_[1]) * side + (z - coordinates_[2]) * sideSq]; }
  value_type data(int i) const { return voxel_block_[i]; }
  bool active() const { return active_; }
  value_type* getBlockRawPtr() { return voxel_block_; }
};

template <typename T> class Octree {
 public:
  typedef typename voxel_traits<T>::value_type value_type;
  static constexpr unsigned int blockSide = 8;
  Octree() {}
  Octree(const Octree&) = delete;
  Octree& operator=(const Octree&) = delete;

  inline int size() const { return size_; }
  inline float dim() const { return dim_; }
  inline Node<T>* root() const { return root_; }
  std::vector<std::unique_ptr<VoxelBlock<T>>>& getBlockBuffer() { return block_buffer_; }
  std::vector<std::unique_ptr<Node<T>>>& getNodesBuffer() { return nodes_buffer_; }

  /* Octree::fetch (octree.hpp:441-458) */
  VoxelBlock<T>* fetch(const int x, const int y, const int z) const {
    Node<T>* n = root_;
    if (!n) return nullptr;
    for (unsigned edge = size_ / 2; edge >= blockSide; edge /= 2) {
      n = n->child((x & edge) > 0u, (y & edge) > 0u, (z & edge) > 0u);
      if (!n) return nullptr;
    }
    return static_cast<VoxelBlock<T>*>(n);
  }
  /* Octree::fetch_octant (octree.hpp:461-478) */
  Node<T>* fetch_octant(const int x, const int y, const int z, const int depth) const {
    Node<T>* n = root_;
    if (!n) return nullptr;
    int d = 1;
    for (unsigned edge = size_ / 2; edge >= blockSide && d <= depth; edge /= 2, ++d) {
      n = n->child((x & edge) > 0u, (y & edge) > 0u, (z & edge) > 0u);
      if (!n) return nullptr;
    }
    return n;
  }
  /* Octree::get (octree.hpp:340-355): the value of the deepest allocated octant on the way down -- a coarse node's
   * value_ for its child corner, or the voxel; Octree::get_fine (octree.hpp:357-377): the voxel or initValue() */
  value_type get(const int x, const int y, const int z) const {
    Node<T>* n = root_;
    if (!n) return voxel_traits<T>::initValue();
    for (unsigned edge = size_ / 2; edge >= blockSide; edge /= 2) {
      const int childid = ((x & edge) > 0) + 2 * ((y & edge) > 0) + 4 * ((z & edge) > 0);
      Node<T>* tmp = n->child(childid);
      if (!tmp) return n->value_[childid];
      n = tmp;
    }
    return static_cast<VoxelBlock<T>*>(n)->data(x, y, z);
  }
  value_type get_fine(const int x, const int y, const int z) const {
    /* outside [0, size)^3 the reference's bit walk lands in whatever block the low bits name and reads past its array; here such a
     * voxel is "not allocated" (the answer the reference gives wherever it is defined) */
    if ((unsigned)x >= (unsigned)size_ || (unsigned)y >= (unsigned)size_ || (unsigned)z >= (unsigned)size_) return voxel_traits<T>::initValue();
    VoxelBlock<T>* b = fetch(x, y, z);
    return b ? b->data(x, y, z) : voxel_traits<T>::initValue();
  }

  /* Octree::get(x, y, z, cached) (octree.hpp:379-408): the voxel from `cached` if that block contains it, else the tree walk of get_fine */
  value_type get(const int x, const int y, const int z, VoxelBlock<T>* cached) const {
    if (cached) {
      const int* lo = cached->coordinates();
      if (x >= lo[0] && x <= lo[0] + (int)blockSide - 1 && y >= lo[1] && y <= lo[1] + (int)blockSide - 1 && z >= lo[2] && z <= lo[2] + (int)blockSide - 1)
        return cached->data(x, y, z);
    }
    return get_fine(x, y, z);
  }

  /* Octree::interp (octree.hpp:541-563) with gather_points (interpolation/interp_gather.hpp:44-237): trilinear blend of the eight
   * voxels around pos (voxel units).  Each corner comes from the block that holds it; the corners of a block that is not allocated
   * read as empty() -- except when the cell straddles a block boundary on all three axes, where the reference goes through
   * get_fine() and a missing block reads as initValue(). */
  template <typename Vec3, typename FieldSelect>
  float interp(const Vec3& pos, FieldSelect select) const {
    const float fl[3] = {floor_(pos(0)), floor_(pos(1)), floor_(pos(2))};
    const float f[3] = {pos(0) - fl[0], pos(1) - fl[1], pos(2) - fl[2]};
    const int lower[3] = {imax(to_int(fl[0]), 0), imax(to_int(fl[1]), 0), imax(to_int(fl[2]), 0)};
    const bool cross[3] = {lower[0] % (int)blockSide == (int)blockSide - 1, lower[1] % (int)blockSide == (int)blockSide - 1,
                           lower[2] % (int)blockSide == (int)blockSide - 1};
    const bool all_cross = cross[0] && cross[1] && cross[2];
    float p[8];
    for (int k = 0; k < 8; ++k) {
      const int d[3] = {k & 1, (k >> 1) & 1, k >> 2};
      const int x = lower[0] + d[0], y = lower[1] + d[1], z = lower[2] + d[2];
      if (all_cross) { p[k] = select(get_fine(x, y, z)); continue; }
      /* the block gather_points fetches for this corner: the base block, stepped across only on the axes that straddle */
      VoxelBlock<T>* b = fetch(lower[0] + (cross[0] ? d[0] : 0), lower[1] + (cross[1] ? d[1] : 0), lower[2] + (cross[2] ? d[2] : 0));
      const int* lo = b ? b->coordinates() : nullptr;
      const bool inside = b && x >= lo[0] && x < lo[0] + (int)blockSide && y >= lo[1] && y < lo[1] + (int)blockSide && z >= lo[2] && z < lo[2] + (int)blockSide;
      p[k] = inside ? select(b->data(x, y, z)) : select(voxel_traits<T>::empty());
    }
    return (((p[0] * (1 - f[0]) + p[1] * f[0]) * (1 - f[1]) + (p[2] * (1 - f[0]) + p[3] * f[0]) * f[1]) * (1 - f[2]) +
            ((p[4] * (1 - f[0]) + p[5] * f[0]) * (1 - f[1]) + (p[6] * (1 - f[0]) + p[7] * f[0]) * f[1]) * f[2]);
  }

  /* Octree::grad (octree.hpp:652-737): central differences of the trilinearly blended field, same term order; the result is scaled by
   * 0.5 * dim / size.  grad(pos) without a selector (octree.hpp:565-650) selects .x, as the reference does. */
  template <typename Vec3, typename FieldSelect>
  Vec3 grad(const Vec3& pos, FieldSelect select) const {
    const float fl[3] = {floor_(pos(0)), floor_(pos(1)), floor_(pos(2))};
    const float fx = pos(0) - fl[0], fy = pos(1) - fl[1], fz = pos(2) - fl[2];
    const int base[3] = {to_int(fl[0]), to_int(fl[1]), to_int(fl[2])};
    const int hi = size_ - 1;
    int X[4], Y[4], Z[4];   /* lower_lower, lower_upper (= lower), upper_lower (= upper), upper_upper */
    X[0] = imax(base[0] - 1, 0); X[1] = imax(base[0], 0); X[2] = imin(base[0] + 1, hi); X[3] = imin(base[0] + 2, hi);
    Y[0] = imax(base[1] - 1, 0); Y[1] = imax(base[1], 0); Y[2] = imin(base[1] + 1, hi); Y[3] = imin(base[1] + 2, hi);
    Z[0] = imax(base[2] - 1, 0); Z[1] = imax(base[2], 0); Z[2] = imin(base[2] + 1, hi); Z[3] = imin(base[2] + 2, hi);
    VoxelBlock<T>* n = fetch(base[0], base[1], base[2]);
    auto G = [&](int xi, int yi, int zi) { return select(get(X[xi], Y[yi], Z[zi], n)); };
    float g[3];
    g[0] = (((G(2, 1, 1) - G(0, 1, 1)) * (1 - fx) + (G(3, 1, 1) - G(1, 1, 1)) * fx) * (1 - fy) +
            ((G(2, 2, 1) - G(0, 2, 1)) * (1 - fx) + (G(3, 2, 1) - G(1, 2, 1)) * fx) * fy) * (1 - fz) +
           (((G(2, 1, 2) - G(0, 1, 2)) * (1 - fx) + (G(3, 1, 2) - G(1, 1, 2)) * fx) * (1 - fy) +
            ((G(2, 2, 2) - G(0, 2, 2)) * (1 - fx) + (G(3, 2, 2) - G(1, 2, 2)) * fx) * fy) * fz;
    g[1] = (((G(1, 2, 1) - G(1, 0, 1)) * (1 - fx) + (G(2, 2, 1) - G(2, 0, 1)) * fx) * (1 - fy) +
            ((G(1, 3, 1) - G(1, 1, 1)) * (1 - fx) + (G(2, 3, 1) - G(2, 1, 1)) * fx) * fy) * (1 - fz) +
           (((G(1, 2, 2) - G(1, 0, 2)) * (1 - fx) + (G(2, 2, 2) - G(2, 0, 2)) * fx) * (1 - fy) +
            ((G(1, 3, 2) - G(1, 1, 2)) * (1 - fx) + (G(2, 3, 2) - G(2, 1, 2)) * fx) * fy) * fz;
    g[2] = (((G(1, 1, 2) - G(1, 1, 0)) * (1 - fx) + (G(2, 1, 2) - G(2, 1, 0)) * fx) * (1 - fy) +
            ((G(1, 2, 2) - G(1, 2, 0)) * (1 - fx) + (G(2, 2, 2) - G(2, 2, 0)) * fx) * fy) * (1 - fz) +
           (((G(1, 1, 3) - G(1, 1, 1)) * (1 - fx) + (G(2, 1, 3) - G(2, 1, 1)) * fx) * (1 - fy) +
            ((G(1, 2, 3) - G(1, 2, 1)) * (1 - fx) + (G(2, 2, 3) - G(2, 2, 1)) * fx) * fy) * fz;
    const float scale = 0.5f * dim_ / size_;
    Vec3 out = pos;
    out(0) = scale * g[0]; out(1) = scale * g[1]; out(2) = scale * g[2];
    return out;
  }
  template <typename Vec3>
  Vec3 grad(const Vec3& pos) const { return grad(pos, [](const value_type& v) { return v.x; }); }

  /* Octree::save (octree.hpp:898-914; io/se_serialise.hpp:54-86): int size, float dim, size_t n, nodes {code, side, value_[8]},
   * size_t n, blocks {code, coordinates, voxel_block_[512]} -- written field by field so that the value_type padding of the
   * reference's build (OFusion: 4 bytes after x) is reproduced whatever this compiler lays out */
  void save(const std::string& filename) {
    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) return;
    std::fwrite(&size_, sizeof(int), 1, f);
    std::fwrite(&dim_, sizeof(float), 1, f);
    uint64_t n = nodes_buffer_.size();
    std::fwrite(&n, 8, 1, f);
    for (auto& p : nodes_buffer_) {
      const int side = (int)p->side_;
      std::fwrite(&p->code_, 8, 1, f); std::fwrite(&side, 4, 1, f);
      for (const auto& v : p->value_) put(f, v);
    }
    n = block_buffer_.size();
    std::fwrite(&n, 8, 1, f);
    for (auto& p : block_buffer_) {
      std::fwrite(&p->code_, 8, 1, f); std::fwrite(p->coordinates_, 4, 3, f);
      for (const auto& v : p->voxel_block_) put(f, v);
    }
    std::fclose(f);
  }

  /* ---- builder used by DenseSLAMSystem::getMap(): octants are appended in key order (the order save() writes and
   * se_hip_save_map writes), then finalize() links children to parents level by level */
  void init(int size, float dim) { size_ = size; dim_ = dim; max_level_ = 0; for (int s = size; s > 1; s >>= 1) ++max_level_; root_ = nullptr; nodes_buffer_.clear(); block_buffer_.clear(); }
  Node<T>* add_node(key_t code, unsigned side) {
    nodes_buffer_.emplace_back(new Node<T>());
    Node<T>* n = nodes_buffer_.back().get();
    n->code_ = code; n->side_ = side;
    return n;
  }
  VoxelBlock<T>* add_block(key_t code, const int coords[3], bool active) {
    block_buffer_.emplace_back(new VoxelBlock<T>());
    VoxelBlock<T>* b = block_buffer_.back().get();
    b->code_ = code; b->side_ = blockSide; b->active_ = active;
    b->coordinates_[0] = coords[0]; b->coordinates_[1] = coords[1]; b->coordinates_[2] = coords[2];
    return b;
  }
  void finalize() {
    for (int level = 0; level <= max_level_; ++level)
      for (auto& n : nodes_buffer_) if ((int)(n->code_ & 0x1FFull) == level) link(n.get(), n->code_);
    for (auto& b : block_buffer_) link(b.get(), b->code_);
  }

 private:
  /* floor / float -> int as the reference's build evaluates them (math::floorf + Eigen cast<int>: truncation; out of int range: INT_MIN, x86) */
  static float floor_(float v) { return std::floor(v); }
  static int to_int(float v) { return (v > -2147483904.f && v < 2147483648.f) ? (int)v : (int)0x80000000; }
  static int imax(int a, int b) { return a > b ? a : b; }
  static int imin(int a, int b) { return a < b ? a : b; }
  static void put(FILE* f, const SDF& v) { std::fwrite(&v.x, 4, 1, f); std::fwrite(&v.y, 4, 1, f); }
  template <typename V> static void put(FILE* f, const V& v) { const uint32_t pad = 0; std::fwrite(&v.x, 4, 1, f); std::fwrite(&pad, 4, 1, f); std::fwrite(&v.y, 8, 1, f); }
  static int coord_of(key_t code, int axis) {   /* compact the bits 3i + axis of the Morton code */
    int v = 0;
    for (int i = 0; i < 21; ++i) v |= (int)((code >> (3 * i + axis)) & 1ull) << i;
    return v;
  }
  void link(Node<T>* n, key_t code) {
    const int level = (int)(code & 0x1FFull);
    if (level == 0) { root_ = n; return; }
    const key_t morton = code & ~0x1FFull;
    const int x = coord_of(morton, 0), y = coord_of(morton, 1), z = coord_of(morton, 2);
    Node<T>* p = root_;
    unsigned edge = size_ / 2;
    for (int d = 1; d < level && p; ++d, edge /= 2) p = p->child((x & edge) > 0u, (y & edge) > 0u, (z & edge) > 0u);
    if (!p) return;   /* (not ancestor-closed: cannot happen for a map produced by the allocation kernels) */
    const int id = ((x & edge) > 0) + 2 * ((y & edge) > 0) + 4 * ((z & edge) > 0);
    p->child(id) = n;
    p->children_mask_ = (unsigned char)(p->children_mask_ | (1 << id));
  }
  int size_ = 0, max_level_ = 0;
  float dim_ = 0.f;
  Node<T>* root_ = nullptr;
  std::vector<std::unique_ptr<Node<T>>> nodes_buffer_;
  std::vector<std::unique_ptr<VoxelBlock<T>>> block_buffer_;
};
}  // namespace se

#endif /* SE_HIP_OCTREE_HPP */
