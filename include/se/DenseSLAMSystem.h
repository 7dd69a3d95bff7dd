/*
 * DenseSLAMSystem -- C++ mirror of the reference's pipeline class for the hot path
 *   se_denseslam/include/se/DenseSLAMSystem.h:58-411  (class DenseSLAMSystem)
 * on top of the C ABI of include/se_hip.h.  Header-only: an application written against the
 * reference class includes this header instead, defines SE_FIELD_TYPE (SDF or OFusion, as the
 * reference requires: DenseSLAMSystem.h:54) and links libse_hip.so.
 *
 * What is kept (same names, argument meaning and "did the stage run" return values):
 *   the two constructors, preprocessing() (mm -> metres, preprocessing.cpp:161-188, fused into the
 *   upload; the optional bilateral filter runs on the device in front of tracking()), integration(),
 *   tracking(), raycasting(), setPose()/getPose()/getPosition()/getInitPos(), getIntegrated(), getTracked(),
 *   getModelDimensions()/getModelResolution()/getComputationResolution(), renderVolume()/renderTrack()/
 *   renderDepth(), dump_mesh(), setViewPose()/getViewPose(), synchroniseDevices().
 * What differs, because the map lives in HBM:
 *   getMap(std::shared_ptr<se::Octree<FieldType>>&) materialises the device map as a host pointer octree with the reference's
 *   node / block layout and read interface (include/se/octree.hpp) -- a snapshot, not the live map; getMap(MapSnapshot&) is
 *   the flat form (blocks sorted by Morton key); getVertexNormal() downloads vertex_/normal_.
 *   tracking() runs the reference's ICP on the device; poses can still be injected
 *   with setPose(), as the reference's GUI does with ground truth (se_apps/src/mainQt.cpp:257-265).
 *   renderVolume / renderTrack / renderDepth shade on the device and copy the RGBW image out.
 * Errors of the C ABI are reported like the reference reports its own failures: message on
 * std::cerr; constructors additionally throw std::runtime_error (the reference would dereference
 * an unallocated map).
 */
#ifndef SE_HIP_DENSESLAMSYSTEM_H
#define SE_HIP_DENSESLAMSYSTEM_H

#include <algorithm>
#include <cstdint>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../se_hip.h"

#include "config.h"       /* the reference's Configuration, field for field (+ hip_device, hip_max_blocks) */
#include "eigen_pods.h"
#include "octree.hpp"     /* SDF / OFusion field types, se::Octree<FieldType> host mirror for getMap() */

#ifndef SE_FIELD_TYPE
#error "define SE_FIELD_TYPE to SDF or OFusion before including se/DenseSLAMSystem.h (as the reference requires)"
#endif
typedef SE_FIELD_TYPE FieldType;

struct MapSnapshot {
  int n_blocks = 0, n_nodes = 0;
  std::vector<int32_t> coords;   /* [n][3] block min corners (voxels), sorted by Morton key */
  std::vector<float> x, y;       /* [n][512], voxel index x + 8y + 64z */
  std::vector<uint8_t> active;
};

class DenseSLAMSystem {
 public:
  DenseSLAMSystem(const Eigen::Vector2i& inputSize, const Eigen::Vector3i& volumeResolution,
                  const Eigen::Vector3f& volumeDimensions, const Eigen::Vector3f& initPose, std::vector<int>& pyramid,
                  const Configuration& config)
      : DenseSLAMSystem(inputSize, volumeResolution, volumeDimensions, toMatrix4f(initPose), pyramid, config) {}

  DenseSLAMSystem(const Eigen::Vector2i& inputSize, const Eigen::Vector3i& volumeResolution,
                  const Eigen::Vector3f& volumeDimensions, const Eigen::Matrix4f& initPose, std::vector<int>& pyramid,
                  const Configuration& config)
      : computation_size_(inputSize), volume_resolution_(volumeResolution), volume_dimension_(volumeDimensions) {
    iterations_.assign(pyramid.begin(), pyramid.end());
    if (iterations_.empty()) iterations_ = {10, 5, 4};
    init_pose_ = Eigen::Vector3f(initPose(0, 3), initPose(1, 3), initPose(2, 3));
    mu_ = config.mu;
    pose_ = initPose;
    raycast_pose_ = initPose;
    se_hip_config c{};
    c.width = inputSize.x(); c.height = inputSize.y();
    c.volume_resolution = volumeResolution.x(); c.volume_dimension = volumeDimensions.x();
    c.field_type = is_sdf() ? SE_HIP_FIELD_SDF : SE_HIP_FIELD_OFUSION;
    c.device = config.hip_device; c.max_blocks = config.hip_max_blocks;
    if (se_hip_create(&c, &h_) != SE_HIP_OK) throw std::runtime_error(std::string("DenseSLAMSystem: ") + se_hip_last_error());
    /* The class never hands out device pointers: every way an application can look at vertex_ / normal_ (tracking, render*, getVertexNormal)
     * is an API call that launches a held-back raycast first, so the one-queue streaming schedule is safe by construction here. */
    if (config.hip_streaming) se_hip_set_streaming(h_, 1);
    if (config.hip_pinned_input) se_hip_set_pinned_input(h_, 1);
    live().push_back(h_);
  }
  ~DenseSLAMSystem() {
    std::vector<se_hip_pipeline*>& l = live();
    l.erase(std::remove(l.begin(), l.end(), h_), l.end());
    se_hip_destroy(h_);
  }
  DenseSLAMSystem(const DenseSLAMSystem&) = delete;
  DenseSLAMSystem& operator=(const DenseSLAMSystem&) = delete;

  /* DenseSLAMSystem.h:147 / DenseSLAMSystem.cpp:128-141 */
  bool preprocessing(const unsigned short* inputDepth, const Eigen::Vector2i& inputSize, const bool filterInput) {
    // bilateralFilterKernel feeds only the tracking pyramid (scaled_depth_[0]); it runs inside se_hip_track
    return ok(se_hip_filter_depth(h_, filterInput ? 1 : 0)) && ok(se_hip_upload_depth_mm(h_, inputDepth, inputSize.x(), inputSize.y()));
  }
  /* DenseSLAMSystem.h:154-157: declared with "TODO Implement this." in the reference and never defined; the RGB image has no
   * consumer in the pipeline (se-denseslam is depth-only), so the overload forwards to the depth-only form. */
  bool preprocessing(const unsigned short* inputDepth, const unsigned char* /*inputRGB*/, const Eigen::Vector2i& inputSize, const bool filterInput) {
    return preprocessing(inputDepth, inputSize, filterInput);
  }
  /* the reference's float_depth_ handed over directly (metres) */
  bool preprocessing(const float* depthMetres) { return ok(se_hip_upload_depth(h_, depthMetres)); }
  /* page-locked host memory for a frame reader's input buffer (Configuration::hip_pinned_input: such images are not copied by preprocessing()) */
  static void* allocateInput(size_t bytes) { return se_hip_host_alloc(bytes); }
  static void freeInput(void* host) { se_hip_host_free(host); }
  /* ... or left where a device-side producer put it (HBM, metres, valid until the next call): no copy at all */
  bool preprocessingDevice(const float* deviceDepthMetres) { return ok(se_hip_set_depth_device(h_, deviceDepthMetres)); }

  /* DenseSLAMSystem.h:173 / DenseSLAMSystem.cpp:143-189: ICP against the last raycast, on the device */
  bool tracking(const Eigen::Vector4f& k, float icp_threshold, unsigned tracking_rate, unsigned frame) {
    const int r = se_hip_track(h_, k.data(), icp_threshold, tracking_rate, frame, iterations_.data(), (int32_t)iterations_.size(), pose_.data());
    ok(r);
    tracked_ = r > 0;
    return tracked_;
  }

  /* DenseSLAMSystem.h:193 / DenseSLAMSystem.cpp:206-268 */
  bool integration(const Eigen::Vector4f& k, unsigned int integration_rate, float mu, unsigned int frame) {
    const int r = se_hip_integrate(h_, pose_.data(), k.data(), integration_rate, mu, frame);
    ok(r);
    integrated_ = r > 0;
    return r > 0;
  }
  /* DenseSLAMSystem.h:212 / DenseSLAMSystem.cpp:191-204 */
  bool raycasting(const Eigen::Vector4f& k, float mu, unsigned int frame) {
    /* held back until the next integration()'s allocation scan (one launch for both) on a streaming handle, se_hip_raycast otherwise */
    const int r = se_hip_raycast_deferred(h_, pose_.data(), k.data(), mu, frame);
    ok(r);
    if (r > 0) raycast_pose_ = pose_;
    return r > 0;
  }

  /* DenseSLAMSystem.h:241-286 / DenseSLAMSystem.cpp:274-300: RGBW images of the computation size */
  void renderVolume(unsigned char* out, const Eigen::Vector2i& outputSize, int frame, int raycast_rendering_rate,
                    const Eigen::Vector4f& k, float largestep) {
    (void)outputSize;
    ok(se_hip_render_volume(h_, out, viewPose_->data(), k.data(), mu_, largestep, (uint32_t)frame, (uint32_t)raycast_rendering_rate));
  }
  void renderTrack(unsigned char* out, const Eigen::Vector2i& outputSize) { (void)outputSize; ok(se_hip_render_track(h_, out)); }
  void renderDepth(unsigned char* out, const Eigen::Vector2i& outputSize) { (void)outputSize; ok(se_hip_render_depth(h_, out)); }

  /* DenseSLAMSystem.h:224 / DenseSLAMSystem.cpp:302-322: marching cubes of the map into a VTK file */
  void dump_mesh(const std::string filename) { ok(se_hip_dump_mesh(h_, filename.c_str())); }
  /* DenseSLAMSystem.h:219: declared, called by se_apps/src/benchmark.cpp:187, and EMPTY in the reference
   * (DenseSLAMSystem.cpp:270-272) -- kept empty so that an application behaves the same; saveMap() is the useful thing */
  void dump_volume(const std::string) {}
  /* the whole map in Octree::save's byte layout, and back (Octree::load without its defects, see se_hip.h) */
  bool saveMap(const std::string& filename) { return ok(se_hip_save_map(h_, filename.c_str())); }
  bool loadMap(const std::string& filename) { return ok(se_hip_load_map(h_, filename.c_str())); }
  void setViewPose(Eigen::Matrix4f* value = NULL) { viewPose_ = value ? value : &pose_; }   /* DenseSLAMSystem.h:363-372 */
  Eigen::Matrix4f* getViewPose() { return viewPose_; }

  /* DenseSLAMSystem.h:295: the reference shares its live se::Octree; here the device map is materialised as a host
   * se::Octree<FieldType> (include/se/octree.hpp) with the reference's node / block member layout and read interface */
  void getMap(std::shared_ptr<se::Octree<FieldType> >& out) {
    out = std::make_shared<se::Octree<FieldType> >();
    out->init(volume_resolution_.x(), volume_dimension_.x());
    int nb = 0, nn = 0;
    if (!ok(se_hip_counts(h_, &nb, &nn))) return;
    std::vector<uint64_t> code(nn);
    std::vector<uint32_t> side(nn);
    std::vector<float> nx((size_t)nn * 8), ny((size_t)nn * 8);
    if (!ok(se_hip_download_nodes(h_, code.data(), side.data(), nx.data(), ny.data()))) return;
    for (int i = 0; i < nn; ++i) {
      se::Node<FieldType>* n = out->add_node(code[i], side[i]);
      for (int j = 0; j < 8; ++j) { n->value_[j].x = nx[(size_t)i * 8 + j]; n->value_[j].y = ny[(size_t)i * 8 + j]; }
    }
    MapSnapshot snap;
    getMap(snap);
    int max_level = 0;
    for (int s = volume_resolution_.x(); s > 1; s >>= 1) ++max_level;
    for (int i = 0; i < snap.n_blocks; ++i) {
      const int* c = &snap.coords[(size_t)i * 3];
      uint64_t key = 0;
      for (int b = 0; b < max_level; ++b)
        key |= ((uint64_t)((c[0] >> b) & 1) << (3 * b)) | ((uint64_t)((c[1] >> b) & 1) << (3 * b + 1)) | ((uint64_t)((c[2] >> b) & 1) << (3 * b + 2));
      se::VoxelBlock<FieldType>* blk = out->add_block(key | (uint64_t)(max_level - 3), c, snap.active[i] != 0);
      for (int v = 0; v < 512; ++v) { blk->voxel_block_[v].x = snap.x[(size_t)i * 512 + v]; blk->voxel_block_[v].y = snap.y[(size_t)i * 512 + v]; }
    }
    out->finalize();
  }
  void getMap(MapSnapshot& out) {
    int nb = 0, nn = 0;
    if (!ok(se_hip_counts(h_, &nb, &nn))) return;
    out.n_blocks = nb; out.n_nodes = nn;
    out.coords.resize((size_t)nb * 3); out.x.resize((size_t)nb * 512); out.y.resize((size_t)nb * 512); out.active.resize(nb);
    if (nb) ok(se_hip_download_blocks(h_, out.coords.data(), out.x.data(), out.y.data(), out.active.data()));
  }
  /* vertex_ / normal_ of the last raycasting(): width*height packed xyz */
  bool getVertexNormal(std::vector<float>& vertex, std::vector<float>& normal) {
    const size_t n = (size_t)computation_size_.x() * computation_size_.y() * 3;
    vertex.resize(n); normal.resize(n);
    return ok(se_hip_download_vertex_normal(h_, vertex.data(), normal.data()));
  }

  bool getTracked() { return tracked_; }
  bool getIntegrated() { return integrated_; }
  Eigen::Vector3f getPosition() {
    return Eigen::Vector3f(pose_(0, 3) - init_pose_.x(), pose_(1, 3) - init_pose_.y(), pose_(2, 3) - init_pose_.z());
  }
  Eigen::Vector3f getInitPos() { return init_pose_; }
  Eigen::Matrix4f getPose() { return pose_; }
  /* DenseSLAMSystem.h:353-356: the translation is relative to the initial position */
  void setPose(const Eigen::Matrix4f pose) {
    pose_ = pose;
    pose_(0, 3) += init_pose_.x(); pose_(1, 3) += init_pose_.y(); pose_(2, 3) += init_pose_.z();
  }
  Eigen::Vector3f getModelDimensions() { return volume_dimension_; }
  Eigen::Vector3i getModelResolution() { return volume_resolution_; }
  Eigen::Vector2i getComputationResolution() { return computation_size_; }
  se_hip_pipeline* handle() { return h_; }
  /* every live pipeline of the process (what the argument-less synchroniseDevices() waits for) */
  static std::vector<se_hip_pipeline*>& live() { static std::vector<se_hip_pipeline*> l; return l; }

 private:
  static bool is_sdf();
  static Eigen::Matrix4f toMatrix4f(const Eigen::Vector3f& t) {  /* se_core/include/se/utils/math_utils.h:86-93 */
    Eigen::Matrix4f m = Eigen::Matrix4f::Identity();
    m(0, 3) = t.x(); m(1, 3) = t.y(); m(2, 3) = t.z();
    return m;
  }
  bool ok(int status) {
    if (status < 0) { std::cerr << "DenseSLAMSystem: " << se_hip_last_error() << std::endl; return false; }
    return true;
  }
  se_hip_pipeline* h_ = nullptr;
  Eigen::Vector2i computation_size_;
  Eigen::Vector3i volume_resolution_;
  Eigen::Vector3f volume_dimension_;
  Eigen::Vector3f init_pose_;
  Eigen::Matrix4f pose_, raycast_pose_;
  Eigen::Matrix4f* viewPose_ = &pose_;
  float mu_ = 0.1f;
  std::vector<int32_t> iterations_;
  bool tracked_ = false, integrated_ = false;
};

namespace se_hip_detail {
template <typename T> struct is_sdf_tag { static const bool value = false; };
template <> struct is_sdf_tag<SDF> { static const bool value = true; };
}  // namespace se_hip_detail
inline bool DenseSLAMSystem::is_sdf() { return se_hip_detail::is_sdf_tag<FieldType>::value; }

/* void synchroniseDevices(): declared and never defined in the reference (DenseSLAMSystem.h:418).  Its body here: wait for
 * everything every live DenseSLAMSystem of this process has enqueued on its device. */
inline void synchroniseDevices() { for (se_hip_pipeline* h : DenseSLAMSystem::live()) se_hip_sync(h); }
inline void synchroniseDevices(DenseSLAMSystem& s) { se_hip_sync(s.handle()); }

#endif /* SE_HIP_DENSESLAMSYSTEM_H */
